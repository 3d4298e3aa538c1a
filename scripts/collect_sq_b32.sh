#!/bin/bash
# three 8-slot SQ passes on the batched windows (B = 32): issue / wait buckets, instruction mix, vector-memory pipeline
export TMPDIR=/tmp
ROOT=$PWD; OUT=$PWD/gpurun_out/prof; mkdir -p $OUT
BB="python $ROOT/scripts/bench_batched.py --B 32 --min-timed-s 0.05 --steps 20"
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
SQC="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM"
SQD="SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
cd $ROOT
for P in A B C D; do
  V=SQ$P
  rm -rf "$OUT/sq_B32_$P"
  timeout 300 rocprofv3 --kernel-trace --pmc ${!V} -d "$OUT/sq_B32_$P" -o pmc --output-format csv -- $BB > "$OUT/sq_B32_$P.log" 2>&1
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
out = os.path.join(os.getcwd(), "gpurun_out", "prof")
for ps in "ABCD":
    c = glob.glob(os.path.join(out, f"sq_B32_{ps}", "**", "*counter_collection.csv"), recursive=True)
    if not c: print("pass", ps, "missing"); continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(c[0])):
        if r["Kernel_Name"].startswith("void k_linearize_batch"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(ps, {k: round(sum(v) / len(v), 1) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()}.get("SQ_WAVES", ""))
PY
