#!/bin/bash
# Round 4, GPU call 22: cycle stamps inside a round of the factorisation and inside the pair precalc; phase 1 with fewer instructions (LD_P1_SHORT) against the current form
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_stamps.so timeout 200 python scripts/r4/round_cycles.py C3
for rep in 1 2 3; do
for L in p1short base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L C3: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels']['k_reduce_solve'])"
done; done
} 2>&1 | tee gpurun_out/r4_call22.log
