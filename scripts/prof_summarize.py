"""Summarise the rocprofv3 outputs of scripts/collect_profiles.sh into the files kept under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(out), "profiles_" + tag)
os.makedirs(dst, exist_ok=True)


def find(pattern):
    c = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return c[0] if c else None


for d in sorted(glob.glob(os.path.join(out, "stats_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("stats_"):]
    ks = find(os.path.basename(d) + "/**/*kernel_stats.csv")
    if not ks:
        print("no kernel stats for", name)
        continue
    rows = list(csv.reader(open(ks)))
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(rows)
    print("==", name)
    for r in rows[1:7]:
        print("  ", r[0][:70], "calls", r[1], "avg ns", r[3])

# B = 32: the stats file averages whole-batch, half-batch (the pipelined iteration runs the batch as two halves) and single-window launches of
# k_linearize_batch under one name; the per-launch trace tells them apart by their grid size -> {tag}_bench_B32_linearize_by_grid.json
# (the bench line's k_linearize figure = the whole-batch launches: ldso_ba_batch_time_linearize)
kt = find("stats_bench_B32/**/*kernel_trace.csv")
if kt:
    by = defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if "k_linearize_batch" not in r.get("Kernel_Name", ""):
            continue
        try:
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            by[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        except (KeyError, ValueError):
            continue
    if by:
        res = {"note": "k_linearize_batch launches of `python scripts/bench_batched.py --B 32` by workgroup count (rocprofv3 --kernel-trace): the largest grid = all 32 windows "
                       "in one launch (the timing loop of ldso_ba_batch_time_linearize and the bench line's roofline figure), about half of it = one half-batch of the "
                       "pipelined iteration (overlapped with the other half's k_reduce_batch / k_gn_solve_batch), ~250 = a single window (set-up)",
               "by_workgroups": {str(k): {"launches": len(v), "avg_us": round(sum(v) / len(v), 3), "min_us": round(min(v), 3), "max_us": round(max(v), 3)} for k, v in sorted(by.items())}}
        json.dump(res, open(os.path.join(dst, f"{tag}_bench_B32_linearize_by_grid.json"), "w"), indent=1)
        print("== B32 k_linearize_batch by grid", {k: v["avg_us"] for k, v in res["by_workgroups"].items()})

# calibration of the gfx950 FETCH_SIZE unit: scripts/micro/pmc_calib.hip (see profiles/r01_pmc_traffic.json "calibration")
for cfg in ("C3", "C4", "C5", "B32"):
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) around `python bench.py --steps 40 --warmup 5 "
                   f"--no-cpu-baseline --no-extras --config {cfg}`; KB per launch as reported; corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE "
                   "tallies 128-B requests at 64 B; calibrated with scripts/micro/pmc_calib.hip: 256 MB of 4-byte coalesced loads report 0.5000x, stores 1.000x)",
           "config": cfg, "kernels": {}}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        cc = find(f"pmc_{cfg}_{C}/**/*counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(list)
        bygrid = defaultdict(lambda: defaultdict(list))          # the batched linearisation: launches of different sizes under one name (whole batch / half batch / one window)
        keep = []
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != C:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name].append(float(r["Counter_Value"]))
            if name.startswith("k_linearize_batch"):
                try:
                    bygrid[name][int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)].append(float(r["Counter_Value"]))
                except (KeyError, ValueError):
                    pass
            if len(keep) < 499 and name.startswith("k_"):
                keep.append(r)
        for name, g in bygrid.items():
            d = res["kernels"].setdefault(name, {})
            big = max(g)
            d[f"{C}_KB_mean_whole_batch"] = round(sum(g[big]) / len(g[big]), 3); d[f"launches_{C}_whole_batch"] = len(g[big]); d["whole_batch_workgroups"] = big
        if keep:
            with open(os.path.join(dst, f"{tag}_pmc_{cfg}_{C}_counter_collection.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(keep[0].keys())); w.writeheader(); w.writerows(keep)
        for name, v in acc.items():
            if not name.startswith("k_"):
                continue
            d = res["kernels"].setdefault(name, {})
            d[f"{C}_KB_mean"] = round(sum(v) / len(v), 3)
            d[f"launches_{C}"] = len(v)
    if not res["kernels"]:
        continue
    for name, d in res["kernels"].items():
        if "FETCH_SIZE_KB_mean" in d and "WRITE_SIZE_KB_mean" in d:
            d["hbm_bytes_per_launch_corrected"] = int(round((2 * d["FETCH_SIZE_KB_mean"] + d["WRITE_SIZE_KB_mean"]) * 1024))
        if "FETCH_SIZE_KB_mean_whole_batch" in d and "WRITE_SIZE_KB_mean_whole_batch" in d:
            d["hbm_bytes_per_launch_corrected_whole_batch"] = int(round((2 * d["FETCH_SIZE_KB_mean_whole_batch"] + d["WRITE_SIZE_KB_mean_whole_batch"]) * 1024))
    json.dump(res, open(os.path.join(dst, f"{tag}_pmc_traffic_{cfg}.json"), "w"), indent=1)
    print("== pmc", cfg)
    for name, d in res["kernels"].items():
        print("  ", name, d)

# SQ counters of the k_linearize kernels (two 8-slot passes per configuration): sums over the waves of a launch, averaged over the launches.
# Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE/XCD.
sq = {"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes A / B) around the bench command of each configuration (C3, C5: bench.py --config; "
              "B32: scripts/bench_batched.py --B 32 = 32 different windows per launch); per-launch means over the launches of the kernel named in `kernel`; "
              "derived: valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of wave time with a VALU instruction executing), wait_any / wait_inst_any / "
              "active_any = the three disjoint buckets of SQ_WAVE_CYCLES, valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES.", "configs": {}}
for cfg in ("C3", "C5", "B32"):
    per = {}
    for ps in ("A", "B"):
        cc = find(f"sq_{cfg}_{ps}/**/*counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        rows = list(csv.DictReader(open(cc)))
        big = 0
        if cfg == "B32":          # the whole-batch launches only (the largest grid): half-batch and single-window launches carry the same kernel name
            for r in rows:
                if "k_linearize_batch" in r["Kernel_Name"]:
                    try: big = max(big, int(r["Grid_Size"]))
                    except (KeyError, ValueError): pass
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("k_linearize"):
                if big and name.startswith("k_linearize_batch") and int(r["Grid_Size"]) != big:
                    continue
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, cs in acc.items():
            d = per.setdefault(name, {})
            for c, v in cs.items():
                d[c] = round(sum(v) / len(v), 1); d["launches_" + ps] = len(v)
    if not per:
        continue
    # the GN-iteration kernel = the one with the most launches
    name = max(per, key=lambda k: per[k].get("launches_A", 0))
    if cfg == "B32" and any(k.startswith("k_linearize_batch") for k in per):          # (its whole-batch launches only, see above: fewer than the set-up launches of k_linearize_one)
        name = [k for k in per if k.startswith("k_linearize_batch")][0]
    d = per[name]
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    der = {}
    if wc:
        der = {"valu_busy": round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4), "wait_any": round(d.get("SQ_WAIT_ANY", 0) / wc, 4),
               "wait_inst_any": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 4), "active_any": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)}
    if d.get("SQ_WAVES"):
        der["valu_insts_per_wave"] = round(d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"], 1)
        der["vmem_rd_insts_per_wave"] = round(d.get("SQ_INSTS_VMEM_RD", 0) / d["SQ_WAVES"], 1)
        der["lds_insts_per_wave"] = round(d.get("SQ_INSTS_LDS", 0) / d["SQ_WAVES"], 1)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    sq["configs"][cfg] = {"kernel": name, "counters": d, "derived": der, "other_kernels": {k: v for k, v in per.items() if k != name}}
    print("== sq", cfg, name, der)
if sq["configs"]:
    json.dump(sq, open(os.path.join(dst, f"{tag}_sq_linearize.json"), "w"), indent=1)
