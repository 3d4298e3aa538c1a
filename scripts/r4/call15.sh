#!/bin/bash
# Round 4, GPU call 15: the driver's own sequence on the round's final library - pytest -x -q -m gpu, smoke, bench
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -6 | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line_final.json 2> gpurun_out/r04_bench_line_final.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_line_final.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "live", j["roofline"]["avg_launch_us_live"], "parity", j["parity_vs_oracle"]["ok"], "cpu", j["cpu_baseline"]["value"])
print("c4", j["c4"]["value"], "c5", j["c5"]["value"], "batched", {k: v.get("gn_iters_per_s_aggregate") for k, v in j["batched"].items() if isinstance(v, dict)}, "adapter", j["adapter"]["gpu_backend_optimize_ms"], j["adapter"]["split_ms"], "tracker", j["tracker"]["gpu_track_ms"])
PY
} 2>&1 | tee gpurun_out/r4_call15.log
