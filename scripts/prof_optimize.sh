#!/bin/bash
# Kernel timeline of ldso_ba_optimize (scripts/time_optimize.py) with rocprofv3 --kernel-trace: start, duration and gap of the last kernels.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/optprof
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace -d "$OUT" -o opt --output-format csv -- python ${SCRIPT:-scripts/time_optimize.py} > "$OUT/run.log" 2>&1
python - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/**/opt_kernel_trace.csv", recursive=True))
if not f:
    print(open("$OUT/run.log").read()[-2000:]); raise SystemExit(1)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-${1:-45}:]
t0 = int(sel[0]["Start_Timestamp"]); prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %6.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r["Kernel_Name"][:60]))
    prev = e
PY
