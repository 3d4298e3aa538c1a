#!/bin/bash
# whole-batch k_linearize_batch under rocprofv3 for two libraries, alternating
export TMPDIR=/tmp
ROOT=$PWD
for rep in 1 2; do
for L in main lm0; do
  if [ "$L" = "main" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  rm -rf /tmp/abp_$L; ( cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abp_$L -o stats --output-format csv -- python scripts/bench_batched.py --B 32 --min-timed-s 0.05 > /tmp/abp_$L.log 2>&1 )
  python - "$L" <<'PY'
import csv, glob, sys
f = glob.glob(f'/tmp/abp_{sys.argv[1]}/**/stats_kernel_trace.csv', recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if 'k_linearize_batch' in r['Kernel_Name']:
        g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']); by.setdefault(g, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(sys.argv[1], {g: (len(v), round(sum(v) / len(v), 2), round(min(v), 2)) for g, v in sorted(by.items())})
PY
done
done
