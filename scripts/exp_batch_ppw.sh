#!/bin/bash
# batched k_linearize against the chunking (points per wavefront): LDSO_BATCH_PPW = 1 (the single-window chunks), 2, 4, 7, default
for P in 1 2 4 6 8 ""; do
  echo "== LDSO_BATCH_PPW=$P"
  LDSO_BATCH_PPW=$P python scripts/bench_batched.py --B 8 32 --min-timed-s 0.1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k in ('B8', 'B32'):
    print(k, 'it/s', d[k]['gn_iters_per_s_aggregate'], 'lin us', d[k]['k_linearize']['avg_launch_us'], 'GB/s', d[k]['k_linearize']['achieved_GBps'])
"
done
