#!/bin/bash
# Round 4, GPU call 7: full suite on the round's library, the default bench line, C5 through the descriptor kernel, batch chunking against the new k_reduce_batch budget
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
rm -f gpurun_out/observed_tolerances.jsonl
{
echo "== full -m gpu suite"
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -25 | cut -c1-600
echo "== bench.py (default command)"
timeout 600 python bench.py > gpurun_out/r4_bench_line.json 2> gpurun_out/r4_bench_line.err; tail -3 gpurun_out/r4_bench_line.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r4_bench_line.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "avg_launch_us_live")}, "kernels", j["kernels"])
print("cpu", {k: j["cpu_baseline"][k] for k in ("value", "cores", "kind", "single_thread_value")})
for k in ("c4", "c5"):
    print(k, j[k].get("value"), j[k].get("ms_per_step"), j[k].get("roofline", {}).get("avg_launch_us_live"), j[k].get("kernels"), j[k].get("error"))
print("batched", {k: (v.get("gn_iters_per_s_aggregate"), v["k_linearize"]["avg_launch_us"]) for k, v in j["batched"].items() if isinstance(v, dict) and "k_linearize" in v} if "error" not in j["batched"] else j["batched"])
print("adapter", j.get("adapter"))
print("tracker", j.get("tracker"))
PY
echo "== C5, descriptor kernel for two slot groups"
LDSO_LIN_DESC=1 timeout 200 python bench.py --config C5 --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us_live'], j['parity_vs_oracle']['ok'], j['kernels'])"
echo "== B32 chunking (points per wavefront) with the new k_reduce_batch budget"
for PPW in 4 6 8; do
  echo -n "ppw $PPW: "; LDSO_BATCH_PPW=$PPW timeout 200 python scripts/bench_batched.py --B 32 --min-timed-s 0.5 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'], b['state_finite'])"
done
echo "== observed tolerances"
python - <<'PY'
import json, collections
d = collections.OrderedDict()
for l in open("gpurun_out/observed_tolerances.jsonl"):
    r = json.loads(l); k = r["name"]
    if k not in d or r["observed"] > d[k][0]: d[k] = (r["observed"], r["limit"])
for k, (o, l) in d.items(): print(f"{k:60s} observed {o:.3e} limit {l:.1e}")
PY
} 2>&1 | tee gpurun_out/r4_call7.log
