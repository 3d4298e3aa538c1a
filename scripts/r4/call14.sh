#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py tests/test_ba_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -8 | cut -c1-500
timeout 300 python scripts/time_adapter.py C3 2>&1 | grep '^{'
} 2>&1 | tee gpurun_out/r4_call14.log
