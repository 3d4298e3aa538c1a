#!/bin/bash
# Round 4, GPU call 21: H_M delta of the extras workgroup with four threads per row (one memory level) against the row-per-thread walk
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for L in prev base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L C3: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels']['k_reduce_solve'])"
done; done
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_p2p_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED" | tail -3
} 2>&1 | tee gpurun_out/r4_call21.log
