#!/bin/bash
# bench.py headline line for several builds of the library (scripts/build_variant.sh): value, ms/step, k_reduce_solve, parity
for L in "$@"; do
  for rep in 1 2; do
    if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
    timeout 120 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', j['value'], j['ms_per_step'], j['kernels']['k_reduce_solve']['avg_us'], j['kernels']['k_linearize']['avg_us'], j['parity_vs_oracle']['ok'])"
  done
done
