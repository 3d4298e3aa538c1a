#!/bin/bash
# scripts/r6/ab_small.sh <cfg list> -- <variants>: bench lines per variant on one box
CFGS=(); while [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "main" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  for cfg in "${CFGS[@]}"; do
    if [ "$cfg" = "B32" ]; then
      timeout 300 python scripts/bench_batched.py --B 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['B32']
print('$L B32', b['gn_iters_per_s_aggregate'], b['ms_per_batch_iteration'], 'k_linearize_us', b['k_linearize']['avg_launch_us'], 'frac', b['k_linearize'].get('frac_of_8TBps'), 'parity', b.get('parity_vs_oracle', {}).get('ok'))"
    else
      timeout 300 python bench.py --no-extras --no-cpu-baseline --config $cfg --min-timed-s 0.5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L $cfg', j['value'], j['ms_per_step'], {k: v.get('avg_us') for k, v in j.get('kernels', {}).items()}, 'frac', j['roofline'].get('frac_live'), j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'])"
    fi
  done
done
done
