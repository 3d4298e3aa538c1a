#!/bin/bash
# Round 5, GPU call 9: the tracker's 8x8 solve by one lane (unpivoted, registers) against the wave version: tests on its library, A/B x3, stamps
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlane.so timeout 600 python -m pytest tests/test_tracker_gpu.py tests/test_nonfinite_gpu.py tests/test_adapter_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -6
for rep in 1 2 3; do
  echo -n "main:   "; timeout 120 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-220
  echo -n "trlane: "; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlane.so timeout 120 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-220
done
echo "stamps, main:";   LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trstamps.so timeout 120 python scripts/bench_tracker.py 2>&1 | grep "tr stamps" | tail -2
echo "stamps, trlane:"; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlane_stamps.so timeout 120 python scripts/bench_tracker.py 2>&1 | grep "tr stamps" | tail -2
} 2>&1 | tee gpurun_out/r5_call9.log
