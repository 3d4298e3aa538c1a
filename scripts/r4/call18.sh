#!/bin/bash
# Round 4, GPU call 18: last validation of the final commit (suite + smoke + a short bench)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -4 | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['parity_vs_oracle']['ok'])"
} 2>&1 | tee gpurun_out/r4_call18.log
