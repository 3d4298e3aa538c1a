#!/bin/bash
# re-collect the C5 evidence only (after the pair-mode kernel): stats, PMC traffic, SQ counters -> gpurun_out/prof (summaries by scripts/prof_summarize.py)
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
ROOT=$PWD
B="python $ROOT/bench.py --no-cpu-baseline --min-timed-s 0.05"
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
cd /tmp
CFG=C5
rm -rf "$OUT/stats_bench_$CFG" "$OUT/pmc_${CFG}_FETCH_SIZE" "$OUT/pmc_${CFG}_WRITE_SIZE" "$OUT/sq_${CFG}_A" "$OUT/sq_${CFG}_B"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG > "$OUT/stats_$CFG.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${CFG}_$C" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/pmc_${CFG}_$C.log" 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc $SQA -d "$OUT/sq_${CFG}_A" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/sq_${CFG}_A.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQB -d "$OUT/sq_${CFG}_B" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/sq_${CFG}_B.log" 2>&1
cd "$ROOT"
python scripts/prof_summarize.py "$OUT" "$TAG" | grep -E "C5|k_linearize_one<2" | head
