#!/bin/bash
# Round 4, GPU call 5: kernel stats of the batched pipeline (main vs record layout), the key-frame sequence through the adapter, adapter wall time
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/prof4
export TMPDIR=/tmp
{
echo "== adapter sequence"
timeout 600 python -m pytest tests/test_adapter_sequence_gpu.py -m gpu -q -s --tb=short 2>&1 | grep -vE "amdgpu.ids|ThreadReduce|pyramid levels" | tail -30 | cut -c1-900
echo "== adapter wall time"
timeout 300 python scripts/time_adapter.py C3 2>&1 | grep '^{'
for L in main base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo "== $L B32 kernel stats"
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof4/stats_B32_$L -o stats --output-format csv -- python $ROOT/scripts/bench_batched.py --B 32 --min-timed-s 0.05 > $ROOT/gpurun_out/prof4/stats_B32_$L.log 2>&1
  f=$(find gpurun_out/prof4/stats_B32_$L -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min", round(float(r["MinNs"]) / 1e3, 2), "max", round(float(r["MaxNs"]) / 1e3, 2), "pct", r["Percentage"])
PY
  grep '^{' gpurun_out/prof4/stats_B32_$L.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); b=j['B32']; print(b['gn_iters_per_s_aggregate'], b['k_linearize']['avg_launch_us'])"
done
} 2>&1 | tee gpurun_out/r4_call5.log
