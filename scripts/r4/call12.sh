#!/bin/bash
# Round 4, GPU call 12: 16-column panels in the matrix-core factorisation (default) against the 4-column panels (libldso_hip_c4.so)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
run() {
  for C in C3 C4; do
    echo -n "$1 $C: "; timeout 200 env "${@:2}" python bench.py --config $C --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{|parity check' | python -c "
import sys,json
s=sys.stdin.read()
try:
    j=json.loads(s); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['rel'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels'])
except Exception: print('FAILED', s[:400])"
  done
}
{
echo "== parity (16-column panels)"
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_nonfinite_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -30 | cut -c1-500
for rep in 1 2; do
run "C=4" LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_c4.so
run "C=16" LDSO_DUMMY=1
done
echo -n "B32 C=4: "; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_c4.so timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
echo -n "B32 C=16: "; timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
} 2>&1 | tee gpurun_out/r4_call12.log
