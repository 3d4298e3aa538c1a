#!/bin/bash
# batched linearisation against points per wavefront (LDSO_BATCH_PPW): scripts/r6/ppw_sweep.sh <variant> <ppw> ...
L=$1; shift
if [ "$L" = "main" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
for P in "$@"; do
  LDSO_BATCH_PPW=$P timeout 300 python scripts/bench_batched.py --B 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['B32']
print('$L ppw $P B32', b['gn_iters_per_s_aggregate'], b['ms_per_batch_iteration'], 'k_linearize_us', b['k_linearize']['avg_launch_us'], 'frac', b['k_linearize'].get('frac_of_8TBps'), 'parity', b.get('parity_vs_oracle', {}).get('ok'))"
done
