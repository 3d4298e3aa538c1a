#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -8 | cut -c1-500
for i in 1 2; do timeout 300 python scripts/time_adapter.py C3 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['gpu_backend_optimize_ms'], j['split_ms'], j['flatten_upload_split_ms'], j['reference_FullSystem_optimize_ms'])"; done
} 2>&1 | tee gpurun_out/r4_call16.log
