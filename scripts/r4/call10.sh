#!/bin/bash
# Round 4, GPU call 10: the rocprofv3 evidence of the round (kernel stats, PMC traffic, SQ counters, the sharded iteration)
cd "$(dirname "$0")/../.."
rm -rf gpurun_out/prof gpurun_out/profiles_r04
timeout 1500 bash scripts/collect_profiles.sh r04 all 2>&1 | tail -80 | cut -c1-300 | tee gpurun_out/r4_call10.log
du -sh gpurun_out/prof gpurun_out/profiles_r04 | tee -a gpurun_out/r4_call10.log
# the raw traces are large: only the summaries travel back
rm -rf gpurun_out/prof
