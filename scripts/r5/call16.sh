#!/bin/bash
# Round 5, GPU call 16: ldso_ba_enqueue_gn with the cached HIP graph against launch by launch (C3, driver's flags, x3 on one box; C5 once), the adapter with prefetching write-back, the tests that enqueue
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
for rep in 1 2 3; do
for gr in 0 1; do
  echo -n "graphs=$gr C3 steps 20: "; LDSO_GN_GRAPHS=$gr timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --steps 20 --warmup 5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['parity_vs_oracle']['rel'])"
done; done
for gr in 0 1; do
  echo -n "graphs=$gr C3 steps 300: "; LDSO_GN_GRAPHS=$gr timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'])"
  echo -n "graphs=$gr C5 steps 20: "; LDSO_GN_GRAPHS=$gr timeout 300 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 --steps 20 --warmup 5 --config C5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'])"
done
timeout 300 python scripts/time_adapter.py C3 2>&1 | tail -1 | python -c "import sys,json; a=json.loads(sys.stdin.read()); print('adapter', a['gpu_backend_optimize_ms'], a['split_ms'], 'first', a['first_call_full_upload_ms'], a['first_call_split_ms'], 'seq', a['resident_window']['keyframe_sequence_resident']['optimize_ms_median'], a['resident_window']['keyframe_sequence_resident']['split_ms_median'])"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ba_gpu.py tests/test_bench_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py tests/test_resident_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -8
} 2>&1 | tee gpurun_out/r5_call16.log
