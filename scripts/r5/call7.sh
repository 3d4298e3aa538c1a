#!/bin/bash
# Round 5, GPU call 7: resident window tests + the adapter suites with the delta path as the default
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
timeout 900 python -m pytest tests/test_resident_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py tests/test_bench_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert|^E " | tail -25
grep -E "resident" gpurun_out/observed_tolerances.jsonl | tail -3
} 2>&1 | tee gpurun_out/r5_call7.log
