#!/bin/bash
# A/B on the GPU box: se3_exp coefficient series + barrier-free leader hand-overs (base) against the closed-form exponential (oldexp)
# and the previous commit's tracker / control step (prev).  Variants built with scripts/build_variant.sh-style links beforehand.
mkdir -p gpurun_out
{
echo "== tracker (ms per track, ms per track in a batch of 20, LM iterations)"
timeout 600 bash scripts/bench_tracker_variants.sh base oldexp prev base prev
for L in base prev base prev; do
  if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  echo "== bench $L"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --min-timed-s 1.0 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
done
unset LDSO_HIP_LIB
} > gpurun_out/exp_exp.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/exp_exp.log
tail -3 gpurun_out/pytest_gpu.log >> gpurun_out/exp_exp.log
cat gpurun_out/exp_exp.log
