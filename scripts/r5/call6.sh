#!/bin/bash
# Round 5, GPU call 6: the resident window - C-ABI test, the adapter tests (delta path is the default now), timing of optimize() per key frame with and without it
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
{
timeout 900 python -m pytest tests/test_resident_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert|^E " | tail -15
grep -E "resident" gpurun_out/observed_tolerances.jsonl | tail -3
timeout 300 python scripts/time_keyframe.py 2>&1 | tail -12
} 2>&1 | tee gpurun_out/r5_call6.log
timeout 600 python scripts/time_adapter.py C3 2>&1 | tail -3 | tee -a gpurun_out/r5_call6.log
