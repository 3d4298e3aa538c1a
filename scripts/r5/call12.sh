#!/bin/bash
# Round 5, GPU call 12: the driver's sequence on the round's library - whole GPU suite, smoke, bench.py --gpus 1 --steps 20 --warmup 5
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
rm -f gpurun_out/observed_tolerances.jsonl
{
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05_driver_command.json 2> gpurun_out/bench_r05_driver_command.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/bench_r05_driver_command.json") if l.startswith("{")][-1])
print("C3", j["value"], j["ms_per_step"], "step_frac", j["roofline"]["step_frac"], "frac", j["roofline"]["frac"], j["parity_vs_oracle"]["ok"])
for k in j["roofline"]["per_kernel"]: print("   ", k)
print("C4", j["c4"]["value"], "C5", j["c5"]["value"], j["c5"]["kernels"])
print("B", {k: (v["gn_iters_per_s_aggregate"], v["k_linearize"]["frac_of_8TBps"]) for k, v in j["batched"].items() if k.startswith("B")})
a = j["adapter"]; print("adapter", a["gpu_backend_optimize_ms"], a["split_ms"], "first", a["first_call_full_upload_ms"], a["resident_window"])
print("tracker", j["tracker"]["gpu_track_ms"], j["tracker"]["gpu_track_batch20_ms"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
} 2>&1 | tee gpurun_out/r5_call12.log
