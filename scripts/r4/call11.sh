#!/bin/bash
# Round 4, GPU call 11: the round's library - full suite, smoke, the driver's bench command
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/observed_tolerances.jsonl
{
echo "== full -m gpu suite"
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -15 | cut -c1-600
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | cut -c1-300
echo "== bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line_final.json 2> gpurun_out/r04_bench_line_final.err; tail -3 gpurun_out/r04_bench_line_final.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04_bench_line_final.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "avg_launch_us_live", "traffic")}, "kernels", j["kernels"])
print("parity", j["parity_vs_oracle"])
print("cpu", {k: j["cpu_baseline"][k] for k in ("value", "cores", "kind", "single_thread_value")}, "port", j["cpu_baseline"].get("port", {}).get("value"))
for k in ("c4", "c5"):
    print(k, j[k].get("value"), j[k].get("ms_per_step"), j[k].get("roofline", {}).get("frac"), j[k].get("roofline", {}).get("avg_launch_us_live"), j[k].get("kernels"), j[k].get("error"))
print("batched", {k: (v.get("gn_iters_per_s_aggregate"), v["k_linearize"]["avg_launch_us"], v["k_linearize"]["frac_of_8TBps"]) for k, v in j["batched"].items() if isinstance(v, dict) and "k_linearize" in v} if "error" not in j["batched"] else j["batched"])
print("adapter", j.get("adapter"))
print("tracker", {k: j["tracker"].get(k) for k in ("gpu_track_ms", "gpu_track_batch20_ms", "cpu_oracle_track_ms")} if "error" not in j.get("tracker", {}) else j.get("tracker"))
print("tracer", j.get("tracer")); print("initializer", j.get("initializer"))
PY
} 2>&1 | tee gpurun_out/r4_call11.log
