#!/bin/bash
# tracker latency for several builds of the library (scripts/build_variant.sh)
for L in "$@"; do
  if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  timeout 200 python scripts/dbg_tracker_coop.py 2>&1 | grep "^COOP" | python -c "import sys,json; j=json.loads(sys.stdin.read()[5:]); print('$L', j['gpu_track_ms'], j['gpu_track_batch20_ms'], j['lm_iterations'])"
done
