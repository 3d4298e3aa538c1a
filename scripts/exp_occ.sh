#!/bin/bash
run() { python scripts/bench_batched.py --B 8 32 --min-timed-s 0.2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', ' '.join('%s it/s %.0f lin %.1f us' % (k, d[k]['gn_iters_per_s_aggregate'], d[k]['k_linearize']['avg_launch_us']) for k in ('B8', 'B32')))"; }
unset LDSO_HIP_LIB; run base
for V in "$@"; do export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$V.so; run $V; done
