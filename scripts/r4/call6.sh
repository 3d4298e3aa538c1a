#!/bin/bash
# Round 4, GPU call 6: cleaned k_linearize parity, the key-frame sequence, adapter wall time, k_reduce_batch register budgets / chunking against the record layout
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
echo "== parity of the cleaned kernel"
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_nonfinite_gpu.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -8 | cut -c1-400
echo "== adapter sequence"
timeout 600 python -m pytest tests/test_adapter_sequence_gpu.py -m gpu -q -s --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -25 | cut -c1-1200
echo "== adapter wall time"
timeout 300 python scripts/time_adapter.py C3 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -12 | cut -c1-1500
echo "== k_reduce_batch budgets (B32: window-it/s, lin us)"
for L in base rb3 rb4 rb6 nodense base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L: "; timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
done
unset LDSO_HIP_LIB
for PPW in 2 3 6 8; do
  echo -n "ppw $PPW: "; LDSO_BATCH_PPW=$PPW timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
done
} 2>&1 | tee gpurun_out/r4_call6.log
