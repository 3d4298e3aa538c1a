#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the bench command per configuration (C3 = the headline, C4, C5), of the batched windows (B = 32: 32 different
#      windows per launch) and of the tracker script                              -> gpurun_out/prof/stats_<name>/
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (C3, C5, B32)     -> gpurun_out/prof/pmc_<cfg>_{FETCH,WRITE}_SIZE/
#   3. SQ counters of k_linearize in two 8-slot passes (C3, C5, B32)              -> gpurun_out/prof/sq_<cfg>_{A,B}/
# Counter passes carry --kernel-trace only (no other trace domain).  scripts/prof_summarize.py writes the summaries into
# gpurun_out/profiles_<tag>/ (copy them into profiles/).  The timed region of bench.py is cut to 50 ms here (--min-timed-s): the default 2 s
# would put 100 000 kernel records into every trace.
set -u
TAG=${1:-r04}
WHAT=${2:-all}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
ROOT=$PWD
B="python $ROOT/bench.py --no-cpu-baseline --min-timed-s 0.05"
BB="python $ROOT/scripts/bench_batched.py --B 32 --min-timed-s 0.05"
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$OUT/sq_counters_available.txt"
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  for CFG in C3 C4 C5; do
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG > "$OUT/stats_$CFG.log" 2>&1
  done
  ( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bench_B32" -o stats --output-format csv -- $BB > "$OUT/stats_B32.log" 2>&1 )
  ( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_tracker" -o stats --output-format csv -- python scripts/bench_tracker.py > "$OUT/stats_tracker.log" 2>&1 )
  # the sharded iteration on one GPU (the multi-GPU step with one rank owning every point): k_reduce + k_gn_export -> exchange -> k_gn_solve -> k_linearize,
  # once with the no-op all-reduce (= what RCCL brackets) and once through the peer-write exchange (k_p2p_push / k_p2p_sum)
  # (DIST_CFGS / DIST_P2P in the environment trim this part: the sharded iteration did not change in round 5 and GPU minutes were short)
  for CFG in ${DIST_CFGS-C3 C4}; do
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dist_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG --force-dist-path > "$OUT/stats_dist_$CFG.log" 2>&1
    if [ "${DIST_P2P-1}" = 1 ]; then
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dist_p2p_$CFG" -o stats --output-format csv -- $B --steps 300 --warmup 30 --no-extras --config $CFG --force-dist-path --allreduce p2p > "$OUT/stats_dist_p2p_$CFG.log" 2>&1
    fi
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for CFG in C3 C4 C5; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${CFG}_$C" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/pmc_${CFG}_$C.log" 2>&1
    done
  done
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_B32_$C" -o pmc --output-format csv -- $BB --steps 20 > "$OUT/pmc_B32_$C.log" 2>&1 )
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sq ]; then
  for CFG in C3 C5; do
    timeout 300 rocprofv3 --kernel-trace --pmc $SQA -d "$OUT/sq_${CFG}_A" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/sq_${CFG}_A.log" 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc $SQB -d "$OUT/sq_${CFG}_B" -o pmc --output-format csv -- $B --steps 40 --warmup 5 --no-extras --config $CFG > "$OUT/sq_${CFG}_B.log" 2>&1
  done
  ( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --pmc $SQA -d "$OUT/sq_B32_A" -o pmc --output-format csv -- $BB --steps 20 > "$OUT/sq_B32_A.log" 2>&1 )
  ( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --pmc $SQB -d "$OUT/sq_B32_B" -o pmc --output-format csv -- $BB --steps 20 > "$OUT/sq_B32_B.log" 2>&1 )
fi
cd "$ROOT"
python scripts/prof_summarize.py "$OUT" "$TAG"
