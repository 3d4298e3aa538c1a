#!/bin/bash
# Round 4, GPU call 28: the calibration part of gn_tail on wave 3 (flag hand-over to wave 0); against the library of call 21 (prev); the whole GPU suite and smoke()
# back substitution back to v_readlane; against the library of call 21 (prev); then the whole GPU suite and smoke()
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out
{
LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_stamps.so timeout 200 python scripts/r4/round_cycles.py C3 | tail -4
for rep in 1 2; do
for L in prev base; do
  if [ "$L" = base ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$ROOT/ldso_amd/libldso_hip_$L.so; fi
  echo -n "$L C3: "; timeout 200 python bench.py --no-cpu-baseline --no-extras --min-timed-s 0.5 2>&1 | grep -E '^\{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity_vs_oracle']['ok'], j['parity_vs_oracle']['energy_log_10_iterations_max_rel'], j['kernels']['k_reduce_solve'])"
done; done
unset LDSO_HIP_LIB
timeout 900 python -m pytest tests/ -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r4_call28.log
