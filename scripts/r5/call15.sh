#!/bin/bash
# Round 5, GPU call 15: the driver's bench command with the round's profiles committed (roofline.frac from profiles/r05_*)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05_driver_command.json 2> gpurun_out/bench_r05_driver_command.err
python - <<'PY' | tee gpurun_out/r5_call15.log
import json
j = json.loads([l for l in open("gpurun_out/bench_r05_driver_command.json") if l.startswith("{")][-1])
print("C3", j["value"], j["ms_per_step"], "step_frac", j["roofline"]["step_frac"], "frac", j["roofline"]["frac"], j["roofline"]["avg_launch_us_source"], j["roofline"]["rocprofv3_profile"], j["parity_vs_oracle"]["ok"])
for k in j["roofline"]["per_kernel"]: print("   ", k)
print("C4", j["c4"]["value"], "C5", j["c5"]["value"])
a = j["adapter"]; print("adapter", a["gpu_backend_optimize_ms"], "first", a["first_call_full_upload_ms"], a["resident_window"]["keyframe_sequence_resident"]["optimize_ms_median"])
print("tracker", j["tracker"]["gpu_track_ms"], j["tracker"]["gpu_track_batch20_ms"], "cpu", j["cpu_baseline"]["value"])
PY
