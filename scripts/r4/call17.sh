#!/bin/bash
# Round 4, GPU call 17: SQ counters of k_linearize_one<2> at C5 (call 10 collected C3 and B32 only)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$PWD/gpurun_out/prof
rm -rf $OUT gpurun_out/profiles_r04c5; mkdir -p $OUT
B="python $ROOT/bench.py --no-cpu-baseline --min-timed-s 0.05"
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $SQA -d "$OUT/sq_C5_A" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config C5 > "$OUT/sq_C5_A.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQB -d "$OUT/sq_C5_B" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config C5 > "$OUT/sq_C5_B.log" 2>&1
cd "$ROOT"
python scripts/prof_summarize.py "$OUT" r04c5 2>&1 | tail -5 | cut -c1-400
rm -rf $OUT
