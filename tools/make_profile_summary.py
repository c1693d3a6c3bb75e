"""Condense tools/gpu.sh runs (gpurun_out/{stats,pmc,sq}/TAG_*) into profiles/TAG_*:
  TAG_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  TAG_pmc_counters.csv   FETCH_SIZE / WRITE_SIZE per dispatch (one counter per pass)
  TAG_traffic.json       HBM bytes of the steady-state dominant launch (bench.py reports it as roofline.traffic)
  TAG_sq.json            SQ counters of the longest dominant launch, normalised
usage: python tools/make_profile_summary.py [--traffic-only|--sq-only|--stats-only] TAG [--kernel SUBSTR]"""
import csv
import json
import os
import shutil
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
tag = args[0]
ksub = args[1] if len(args) > 1 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
only = [f for f in flags if f.endswith("-only")]


def want(what):
    return not only or f"--{what}-only" in only


def dominant(rows, counter=None):
    """name of the kernel with the most total time among the fused / welch kernels"""
    tot = {}
    for r in rows:
        k = r["Kernel_Name"]
        if ksub:  # an explicit kernel-name substring picks among ALL kernels
            if ksub not in k:
                continue
        elif "fused_kernel" not in k and "fused3_kernel" not in k and "welch_kernel" not in k and "bigfft_" not in k:
            continue
        tot[k] = tot.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return max(tot, key=tot.get) if tot else None


def bench_line(path):
    try:
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    except OSError:
        pass
    return None


if want("stats"):
    src = os.path.join(go, "stats", f"{tag}_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, f"{tag}_kernel_stats.csv"))
        b = bench_line(os.path.join(go, "stats", f"{tag}_bench.log"))
        if b:
            json.dump(b, open(os.path.join(prof, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
        print("stats ->", f"profiles/{tag}_kernel_stats.csv")

if want("traffic"):
    rows_out, steady, kern, grid = [], {}, None, None
    ok = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(go, "pmc", f"{tag}_{c}_counter_collection.csv")
        if not os.path.exists(p):
            ok = False
            break
        rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == c]
        kern = kern or dominant(rows)
        vals = []
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            rows_out.append((c, name, r["Grid_Size"], float(r["Counter_Value"])))
            if r["Kernel_Name"] == kern:
                vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        gmax = max(g for g, _ in vals)  # steady-state launches: the largest grids (all stages active)
        ss = [v for g, v in vals if g >= 0.95 * gmax]
        steady[c] = sum(ss) / len(ss)
        grid = gmax
    if ok:
        with open(os.path.join(prof, f"{tag}_pmc_counters.csv"), "w") as f:
            f.write("counter,kernel,grid_size,value_KiB\n")
            for r in rows_out:
                f.write(f"{r[0]},{r[1]},{r[2]},{r[3]}\n")
        b = bench_line(os.path.join(go, "pmc", f"{tag}_FETCH_SIZE_bench.log")) or {}
        cfg = b.get("config", {})
        alg = int(cfg.get("algorithmic_bytes_per_sample", 4.0) * cfg.get("samples_per_step_per_channel", 1 << 26) * cfg.get("channels", 1))
        traffic = 2 * steady["FETCH_SIZE"] * 1024 + steady["WRITE_SIZE"] * 1024
        short = kern.split("(")[0].replace("void ", "").replace("psdk::", "")
        json.dump({
            "round": tag, "kernel": short.split("<")[0], "kernel_full": short, "grid_size": grid,
            "workload": cfg.get("workload", "bench.py default") + "; one span per launch (--coalesce 1 --passes 1), "
                        "steady-state launches (all stages active)",
            "fft_size": cfg.get("fft_size"), "channels": cfg.get("channels"),
            "samples_per_launch": cfg.get("samples_per_step_per_channel", 1 << 26) * cfg.get("channels", 1),
            "FETCH_SIZE_KiB": steady["FETCH_SIZE"], "WRITE_SIZE_KiB": steady["WRITE_SIZE"],
            "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio": traffic / alg,
            "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts half of streaming reads; calibration "
                    "in profiles/README.md). Excess over algorithmic = inter-stage streams written once and read once "
                    "(8/7 geometric tail) + per-workgroup partials + run-boundary re-reads.",
        }, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
        print(json.load(open(os.path.join(prof, f"{tag}_traffic.json"))))

if want("sq"):
    agg, kern = {}, None
    i = 1
    while True:
        p = os.path.join(go, "sq", f"{tag}_set{i}_counter_collection.csv")
        if not os.path.exists(p):
            break
        rows = list(csv.DictReader(open(p)))
        kern = kern or dominant(rows)
        by = {}
        for r in rows:
            if r["Kernel_Name"] != kern:
                continue
            d = by.setdefault(r["Dispatch_Id"], {"_dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                 "_grid": int(r["Grid_Size"]), "_wg": int(r["Workgroup_Size"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
        if by:
            b = max(by.values(), key=lambda x: x["_dur"])
            for k, v in b.items():
                agg.setdefault(k if not k.startswith("_") else f"{k}_set{i}", v)
        i += 1
    if agg:
        wc = agg.get("SQ_WAVE_CYCLES")
        bc = agg.get("SQ_BUSY_CYCLES")
        waves = agg.get("SQ_WAVES")
        out = {"round": tag, "kernel": kern.split("(")[0].replace("void ", ""), "raw": agg, "derived": {}}
        d = out["derived"]
        if wc:
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                      "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if k in agg:
                    d[k + "_per_wave_cycle"] = agg[k] / wc
        if waves:
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if k in agg:
                    d[k + "_per_wave"] = agg[k] / waves
        if bc and "SQ_LDS_IDX_ACTIVE" in agg:
            d["LDS_IDX_ACTIVE_per_busy_cycle"] = agg["SQ_LDS_IDX_ACTIVE"] / bc
        if "SQ_LDS_BANK_CONFLICT" in agg and agg.get("SQ_LDS_IDX_ACTIVE"):
            d["LDS_bank_conflict_frac"] = agg["SQ_LDS_BANK_CONFLICT"] / agg["SQ_LDS_IDX_ACTIVE"]
        json.dump(out, open(os.path.join(prof, f"{tag}_sq.json"), "w"), indent=1)
        print(json.dumps(d, indent=1))
