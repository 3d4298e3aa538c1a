#!/bin/bash
# Round 4, GPU call 2: what fails with the record layout (full error output), and the new multi-rank tests against the known-good main library.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== bench C3, record layout"
timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.2 2>&1 | grep -vE "amdgpu.ids" | tail -25 | cut -c1-600
echo "== golden + ba, record layout, first failure"
timeout 600 python -X faulthandler -m pytest tests/test_golden_ref.py tests/test_golden.py tests/test_ba_gpu.py -m gpu -x -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -60 | cut -c1-400
export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_main.so
echo "== new tests, main library"
timeout 900 python -m pytest tests/test_p2p_gpu.py tests/test_bench_gpu.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -60 | cut -c1-600
echo "== bench --force-dist-path, main library"
timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.2 --force-dist-path 2>&1 | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['parity_vs_oracle'])"
} 2>&1 | tee gpurun_out/r4_call2.log
