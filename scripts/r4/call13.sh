#!/bin/bash
# Round 4, GPU call 13: 16-column panels with bounded scalar-register pressure against the 4-column panels
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
run() {
  for C in C3; do
    echo -n "$1 $C: "; timeout 200 env "${@:2}" python bench.py --config $C --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{|parity check' | python -c "
import sys,json
s=sys.stdin.read()
try:
    j=json.loads(s); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['rel'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels'])
except Exception: print('FAILED', s[:400])"
  done
}
{
for rep in 1 2; do
run "C=4" LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_c4.so
run "C=16" LDSO_DUMMY=1
done
echo "== parity (16-column panels)"
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -6 | cut -c1-500
} 2>&1 | tee gpurun_out/r4_call13.log
