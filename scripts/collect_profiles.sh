#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the bench command per configuration (C3 = the headline, C4, C5), of the default command (all lines,
#      incl. the batched windows) and of the tracker script                    -> gpurun_out/prof/stats_<name>/
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (C3, C5)          -> gpurun_out/prof/pmc_<cfg>_{FETCH,WRITE}_SIZE/
# Counter passes carry --kernel-trace only (no other trace domain).  scripts/prof_summarize.py writes the summaries into
# gpurun_out/profiles_<tag>/ (copy them into profiles/).
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
B="python $PWD/bench.py --no-cpu-baseline"
cd /tmp
for CFG in C3 C4 C5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG > "$OUT/stats_$CFG.log" 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_default" -o stats --output-format csv -- $B --steps 300 --warmup 30 > "$OUT/stats_default.log" 2>&1
( cd "$OLDPWD" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_tracker" -o stats --output-format csv -- python scripts/bench_tracker.py > "$OUT/stats_tracker.log" 2>&1 )
for CFG in C3 C5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${CFG}_$C" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/pmc_${CFG}_$C.log" 2>&1
  done
done
cd - > /dev/null
python scripts/prof_summarize.py "$OUT" "$TAG"
