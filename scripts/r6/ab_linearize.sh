#!/bin/bash
# A/B of library variants on ONE box: the batched linearisation (B = 32, 8) and the C5 / C3 lines per variant.
#   scripts/r6/ab_linearize.sh <variant> ...      ("main" = ldso_amd/libldso_hip.so, else ldso_amd/libldso_hip_<variant>.so)
mkdir -p gpurun_out
for L in "$@"; do
  if [ "$L" = "main" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  for rep in 1 2; do
    timeout 300 python scripts/bench_batched.py --B 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['B32']
print('$L B32', b['gn_iters_per_s_aggregate'], b['ms_per_batch_iteration'], 'k_linearize_us', b['k_linearize']['avg_launch_us'], 'frac', b['k_linearize'].get('frac_of_8TBps'), 'finite', b.get('state_finite'))"
  done
  for cfg in C5 C3; do
    timeout 300 python bench.py --no-extras --no-cpu-baseline --config $cfg --min-timed-s 0.5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L $cfg', j['value'], j['ms_per_step'], {k: v.get('avg_us') for k, v in j.get('kernels', {}).items()}, 'frac', j['roofline'].get('frac_live'), j['parity_vs_oracle'])"
  done
done
