#!/bin/bash
for L in "$@"; do
  if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.3 --config C5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$L C5', d['value'], 'it/s', d['ms_per_step'], 'ms; lin live us', d['roofline']['avg_launch_us_live'])"
done
unset LDSO_HIP_LIB
LDSO_LIN_DESC=1 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.3 --config C5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('base+DESC C5', d['value'], 'it/s', d['ms_per_step'], 'ms; lin live us', d['roofline']['avg_launch_us_live'])"
