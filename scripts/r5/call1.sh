#!/bin/bash
# Round 5, GPU call 1 (after scripts/r5/prepare.sh): the tracker-leader branch - its tests, its timing against main on the same box, the leader's stamps of both
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
python -c "import torch" 2>/dev/null          # page the image in before anything is timed
LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlead.so timeout 600 python -m pytest tests/test_tracker_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4
for rep in 1 2 3; do
  echo -n "main:   "; timeout 120 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-200
  echo -n "trlead: "; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlead.so timeout 120 python scripts/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-200
done
echo "stamps, main:";   LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trstamps.so timeout 120 python scripts/bench_tracker.py 2>&1 | grep "tr stamps" | tail -3
echo "stamps, trlead:"; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_trlead_stamps.so timeout 120 python scripts/bench_tracker.py 2>&1 | grep "tr stamps" | tail -3
} 2>&1 | tee gpurun_out/r5_call1.log
# HIP graph of 300 GN iterations against the same launches enqueued one by one (C3)
timeout 200 python scripts/r5/graph_gn.py C3 300 2>&1 | tail -4 | tee -a gpurun_out/r5_call1.log
