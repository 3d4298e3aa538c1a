#!/bin/bash
# Round 4, GPU call 9: kernel-argument touch in k_reduce_solve / k_gn_solve / k_reduce (A/B), the adapter test that failed in call 8
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
run() {
  for C in C3 C5; do
    echo -n "$1 $C: "; timeout 200 env "${@:2}" python bench.py --config $C --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{|parity check' | python -c "
import sys,json
s=sys.stdin.read()
try:
    j=json.loads(s); print(j['value'], j['ms_per_step'], 'lin live', j['roofline']['avg_launch_us_live'], j['parity_vs_oracle']['ok'], j['kernels'])
except Exception: print('FAILED', s[:300])"
  done
}
{
echo "== adapter test"
timeout 300 python -m pytest tests/test_adapter_gpu.py -m gpu -q --tb=short -k "linearized" 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -30 | cut -c1-700
for rep in 1 2; do
run "no touch in solve/reduce" LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_notouchs.so
run "touch (default)" LDSO_DUMMY=1
done
echo -n "dist path, no touch: "; LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_notouchs.so timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --force-dist-path 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['kernels'])"
echo -n "dist path, touch: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --force-dist-path 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['kernels'], j['parity_vs_oracle']['ok'])"
} 2>&1 | tee gpurun_out/r4_call9.log
