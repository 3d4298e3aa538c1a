#!/bin/bash
# Round 4, GPU call 8: k_linearize_one (descriptor + chunk geometry in the kernel arguments, argument lines touched at entry) against the table-driven kernels
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
run() {   # label, env..., 
  for C in C3 C5; do
    echo -n "$1 $C: "; timeout 200 env "${@:2}" python bench.py --config $C --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{|parity check' | python -c "
import sys,json
s=sys.stdin.read()
try:
    j=json.loads(s); print(j['value'], j['ms_per_step'], 'lin live', j['roofline']['avg_launch_us_live'], 'b2b', j['roofline']['avg_launch_us_back_to_back_100'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'])
except Exception: print('FAILED', s[:300])"
  done
}
{
echo "== parity (default library: k_linearize_one)"
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_golden_ref.py tests/test_fullsize_gpu.py tests/test_nonfinite_gpu.py tests/test_adapter_gpu.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -8 | cut -c1-500
for rep in 1 2; do
run "r3path (table, no SINGLE)" LDSO_LIN_NO_ONE=1 LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_nosingle.so
run "table + SINGLE" LDSO_LIN_NO_ONE=1
run "one, no touch" LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_notouch.so
run "one + touch (default)" LDSO_DUMMY=1
run "one + touch + kernarg preload" LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_kpre.so
done
echo "== adapter wall time"
timeout 300 python scripts/time_adapter.py C3 2>&1 | grep '^{'
echo "== B32 / B8"
timeout 200 python scripts/bench_batched.py --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print({k:(v['gn_iters_per_s_aggregate'], v['k_linearize']['avg_launch_us']) for k,v in j.items() if isinstance(v,dict)})"
} 2>&1 | tee gpurun_out/r4_call8.log
