#!/bin/bash
# k_linearize experiments: headline C3 / C5 lines and the batched line
python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.3 --config C3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('C3', d['value'], 'it/s', d['ms_per_step'], 'ms; lin live us', d['roofline']['avg_launch_us_live'], d['kernels'])"
for E in "" 1; do
LDSO_LIN_DESC=$E python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.3 --config C5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('C5 desc=$E', d['value'], 'it/s', d['ms_per_step'], 'ms; lin live us', d['roofline']['avg_launch_us_live'], d['kernels'])"
done
python scripts/bench_batched.py --B 8 32 --min-timed-s 0.2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k in ('B8', 'B32'):
    print(k, 'it/s', d[k]['gn_iters_per_s_aggregate'], 'lin us', d[k]['k_linearize']['avg_launch_us'], 'GB/s', d[k]['k_linearize']['achieved_GBps'])"
