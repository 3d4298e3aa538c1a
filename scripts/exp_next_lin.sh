#!/bin/bash
# First GPU call of the next round: the two prepared k_linearize_batch experiments (compiled out by default; the default code object is
# instruction-identical to the one measured in round 3):
#   LD_SCALAR_POINT=1   per-point record entries (22 of 31 loads per point) through scalar loads         (ba_linearize.hip: PT())
#   LD_GLOBAL_TAPS=1    the 2x2 taps through global instead of flat addresses
# Build the variants where hipcc is (before gpurun):   bash scripts/exp_next_lin.sh build
# On the GPU box:                                      bash scripts/exp_next_lin.sh run
set -u
cd "$(dirname "$0")/.."
if [ "${1:-run}" = build ]; then
  bash scripts/build_variant.sh sp ba_linearize.hip "-DLD_SCALAR_POINT=1" | tail -1
  bash scripts/build_variant.sh gt ba_linearize.hip "-DLD_GLOBAL_TAPS=1" | tail -1
  bash scripts/build_variant.sh spgt ba_linearize.hip "-DLD_SCALAR_POINT=1 -DLD_GLOBAL_TAPS=1" | tail -1
  # with 22 loads off vmcnt the one-point-ahead record prefetch (LD_PREFETCH, no gain in round 3) may start to pay
  bash scripts/build_variant.sh spgtpf ba_linearize.hip "-DLD_SCALAR_POINT=1 -DLD_GLOBAL_TAPS=1 -DLD_PREFETCH=1" | tail -1
  exit 0
fi
mkdir -p gpurun_out
for L in base sp gt spgt spgtpf base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  echo "== $L: C3 (value, ms per step, live k_linearize us, parity) then B = 32"
  timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us_live'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'])"
  timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
  # parity of the variant proper: the batched windows must still equal the individual runs bit for bit
  [ "$L" != base ] && timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q -k "batch" 2>&1 | tail -1
done 2>&1 | tee gpurun_out/exp_next_lin.log
