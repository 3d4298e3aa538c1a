#!/bin/bash
# Round 5, GPU call 4: overlapped GN iterations (two streams, k_reduce_solve parked on a device counter behind k_linearize) against one stream, C3 / C4, x3 on one box;
# then the tests that run enqueue_gn
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
for rep in 1 2 3; do
for ov in 0 1; do
  echo -n "overlap=$ov C3: "; LDSO_GN_OVERLAP=$ov timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['parity_vs_oracle']['rel'], j['kernels'])"
done; done
for ov in 0 1; do
  echo -n "overlap=$ov C4: "; LDSO_GN_OVERLAP=$ov timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --config C4 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels'])"
done
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ba_gpu.py tests/test_adapter_gpu.py tests/test_bench_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8
} 2>&1 | tee gpurun_out/r5_call4.log
