#!/bin/bash
# Round 5, GPU call 8: fused k_reduce_solve for 9 <= F <= 12 (2 K-splits per Schur tile) against the two launches, C5 x3 on one box; the tests of F > 8; device marginalizeFrame in the sequence tests
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
for rep in 1 2 3; do
for L in nofuse9 main; do
  if [ $L = main ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L C5: "; timeout 300 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --config C5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels'])"
done; done
unset LDSO_HIP_LIB
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ba_gpu.py tests/test_adapter_sequence_gpu.py tests/test_adapter_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert|^E " | tail -12
} 2>&1 | tee gpurun_out/r5_call8.log
