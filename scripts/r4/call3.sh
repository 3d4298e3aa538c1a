#!/bin/bash
# Round 4, GPU call 3: record layout with the element-bit_cast fix: full suite, then A/B main / record layout / +global taps / +pipeline on one box.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== full -m gpu suite, record layout (default library)"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -60 | cut -c1-500
for L in main base gt pipe main base gt pipe; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  for C in C3 C5; do
  echo "== $L $C (value, ms per step, live k_linearize us, parity ok, energy log rel)"
  timeout 200 python bench.py --config $C --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{|parity check' | python -c "
import sys,json
s=sys.stdin.read()
try:
    j=json.loads(s); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us_live'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels'])
except Exception: print('FAILED', s[:300])"
  done
  echo "== $L B32 (window-it/s, k_linearize us, finite)"
  timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
done
for L in gt pipe; do
export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so
echo "== $L parity"
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_nonfinite_gpu.py -m gpu -q 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -12 | cut -c1-300
done
} 2>&1 | tee gpurun_out/r4_call3.log
