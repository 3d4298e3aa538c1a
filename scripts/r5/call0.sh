#!/bin/bash
# Round 5, GPU call 0: the new parity tests (F = 12 reference-direct adapter test, common-state comparisons, tightened stage energy), then calls 1 and 2
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_adapter_gpu.py tests/test_ba_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12 | tee gpurun_out/r5_call0.log
grep -E "common_state|stage_relinearise" gpurun_out/observed_tolerances.jsonl | tail -40 | tee -a gpurun_out/r5_call0.log
bash scripts/r5/call1.sh
bash scripts/r5/call2.sh
