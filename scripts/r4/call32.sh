#!/bin/bash
# Round 4, GPU call 32 (the last 80 s of the round's budget): the tracker's leader with every load of its step / accept blocks in front of the first store
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
timeout 45 python -m pytest tests/test_tracker_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
echo -n "base: "; timeout 25 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-260
echo -n "prev: "; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_prev.so timeout 25 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-260
} 2>&1 | tee gpurun_out/r4_call32.log
