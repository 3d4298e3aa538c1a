#!/bin/bash
for C in "$@"; do
  LDSO_BATCH_C0=$C timeout 300 python scripts/bench_batched.py --B 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['B32']
print('c0 $C B32', b['gn_iters_per_s_aggregate'], b['ms_per_batch_iteration'], 'k_linearize_us', b['k_linearize']['avg_launch_us'], 'frac', b['k_linearize'].get('frac_of_8TBps'), 'chunk', b['k_linearize'].get('points_per_workgroup'), 'parity', b.get('parity_vs_oracle', {}).get('ok'))"
done
