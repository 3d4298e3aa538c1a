#!/bin/bash
# Round 4, GPU call 29: canbreak on four lanes (square roots and comparisons side by side, conjunction by ballot); against the library of call 21 (prev)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_stamps.so timeout 200 python scripts/r4/round_cycles.py C3 | tail -3
for rep in 1 2; do
for L in prev base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L C3: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels']['k_reduce_solve'])"
done; done
unset LDSO_HIP_LIB
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
} 2>&1 | tee gpurun_out/r4_call29.log
