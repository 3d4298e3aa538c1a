#!/bin/bash
# Round 5, GPU call 11: the batched linearisation with 4-wave workgroups (2 per CU at the unconstrained register count, 3 per CU at <= 168 VGPRs) against 8 x 1, B = 32
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
for rep in 1 2; do
for cfg in "main:" "b4x2:2" "b4x2:3" "b4x2:4" "b4x3:1" "b4x3:2" "b4x3:3"; do
  L=${cfg%%:*}; PPW=${cfg#*:}
  if [ $L = main ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  if [ -z "$PPW" ]; then unset LDSO_BATCH_PPW; else export LDSO_BATCH_PPW=$PPW; fi
  echo -n "$L ppw8=$PPW B32: "; timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.3 2>&1 | tail -1 | cut -c1-400
done; done
unset LDSO_HIP_LIB LDSO_BATCH_PPW
} 2>&1 | tee gpurun_out/r5_call11.log
