#!/bin/bash
# A/B: register budget of k_reduce_batch (workgroups per CU the allocation leaves room for), B = 32 different windows
for L in "$@"; do
  if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['B32'] if 'B32' in j else j
print('$L', json.dumps(b)[:400])"
done
