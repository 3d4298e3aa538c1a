#!/bin/bash
# Round 5, GPU call 3: whole GPU suite on main (keeper wave 3, tracker leader merged), HIP graph replay (fixed timing), the driver's bench command
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12 | tee gpurun_out/r5_call3.log
timeout 200 python scripts/r5/graph_gn.py C3 300 2>&1 | tail -4 | tee -a gpurun_out/r5_call3.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_call3_bench.json 2> gpurun_out/r5_call3_bench.err
python - <<'PY' | tee -a gpurun_out/r5_call3.log
import json
j = json.loads([l for l in open("gpurun_out/r5_call3_bench.json") if l.startswith("{")][-1])
print("C3", j["value"], j["ms_per_step"], j["roofline"]["step_frac"], j["roofline"]["per_kernel"])
print("C5", j["c5"]["value"], j["c5"]["kernels"]); print("adapter", j["adapter"]["gpu_backend_optimize_ms"], j["adapter"]["split_ms"]); print("tracker", j["tracker"]["gpu_track_ms"], j["tracker"]["gpu_track_batch20_ms"])
PY
