#!/bin/bash
# Round 5, GPU call 2 (after scripts/r5/prepare.sh): the factorisation switches of branch next/solve-variants at C3, three repetitions on one box, parity printed
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
python -c "import torch" 2>/dev/null
for rep in 1 2 3; do
for L in sv_base sv_keeper3 sv_skip sv_keeper3skip sv_keeper2skip; do
  export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so
  echo -n "$L C3: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels']['k_reduce_solve'])"
done; done
unset LDSO_HIP_LIB
} 2>&1 | tee gpurun_out/r5_call2.log
