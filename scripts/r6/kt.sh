#!/bin/bash
# kernel names launched by a script: scripts/r6/kt.sh <script.py> [args]
ROOT=$PWD
export TMPDIR=/tmp
rm -rf /tmp/kt
cd /tmp
rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python $ROOT/"$@" 2>&1 | grep -v rocprofv3 | tail -3
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Kernel_Name"][:100], r["Grid_Size_X"], r["Workgroup_Size_X"])
PY
