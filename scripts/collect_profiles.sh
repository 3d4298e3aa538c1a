#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command            -> gpurun_out/prof/stats/
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes       -> gpurun_out/prof/pmc_{FETCH,WRITE}_SIZE/
# then scripts/prof_summarize.py writes the summaries (copy them into profiles/).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python $PWD/bench.py --steps 300 --warmup 30 --no-cpu-baseline"
PMCB="python $PWD/bench.py --steps 40 --warmup 5 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -o pmc --output-format csv -- $PMCB > "$OUT/pmc_$C.log" 2>&1
done
cd - > /dev/null
python scripts/prof_summarize.py "$OUT" "$TAG"
