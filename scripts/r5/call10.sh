#!/bin/bash
# Round 5, GPU call 10: call-by-call window edits, the tracker with the one-lane solve as the default (tests, timing, stamps)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
timeout 900 python -m pytest tests/test_resident_gpu.py tests/test_tracker_gpu.py tests/test_nonfinite_gpu.py tests/test_abi.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -12
for rep in 1 2; do echo -n "tracker: "; timeout 120 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-260; done
echo "stamps:"; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trstamps.so timeout 120 python scripts/bench_tracker.py 2>&1 | grep "tr stamps" | tail -2
} 2>&1 | tee gpurun_out/r5_call10.log
