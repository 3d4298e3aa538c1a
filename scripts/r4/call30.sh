#!/bin/bash
# Round 4, GPU call 30: the round's final kernels under rocprofv3 --kernel-trace --stats (C3 = the headline, C4, C5, B = 32, the sharded iteration at C3), then the
# driver's bench command itself (the line kept as profiles/r04_bench_line_final.json)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
B="python $ROOT/bench.py --no-cpu-baseline --min-timed-s 0.05"
BB="python $ROOT/scripts/bench_batched.py --B 32 --min-timed-s 0.05"
cd /tmp
for CFG in C3 C4 C5; do
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG > "$OUT/stats_$CFG.log" 2>&1
done
( cd "$ROOT" && timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_B32" -o stats --output-format csv -- $BB > "$OUT/stats_B32.log" 2>&1 )
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dist_C3" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config C3 --force-dist-path > "$OUT/stats_dist_C3.log" 2>&1
cd "$ROOT"
python scripts/prof_summarize.py "$OUT" r04 2>&1 | tee gpurun_out/r4_call30_summary.log
grep -hE '^\{' "$OUT"/stats_C3.log "$OUT"/stats_C4.log "$OUT"/stats_C5.log "$OUT"/stats_B32.log "$OUT"/stats_dist_C3.log > gpurun_out/r4_call30_lines.jsonl
rm -rf "$OUT"/stats_*/          # the raw traces stay on the box
timeout 400 python bench.py > gpurun_out/r4_call30_bench_default.json 2> gpurun_out/r4_call30_bench_default.err
tail -c 600 gpurun_out/r4_call30_bench_default.json
