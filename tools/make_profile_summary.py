"""Condense a tools/scripts_gpu_record.sh run (gpurun_out/record/) into profiles/<tag>_*:
kernel stats (copied), one CSV of the FETCH_SIZE / WRITE_SIZE passes, and the traffic JSON that
bench.py reports as roofline.traffic.  usage: python tools/make_profile_summary.py r01d"""
import csv, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rec = os.path.join(root, "gpurun_out", "record")
prof = os.path.join(root, "profiles")
shutil.copy(os.path.join(rec, "stats_kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats.csv"))
rows = []
steady = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for r in csv.DictReader(open(os.path.join(rec, f"pmc_{c}_counter_collection.csv"))):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.append((c, name, r["Grid_Size"], float(r["Counter_Value"])))
        if "fused_kernel<1024, 0, false>" in r["Kernel_Name"]:
            vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    # steady-state launches: the ones with the largest grids (all stages active)
    gmax = max(g for g, _ in vals)
    ss = [v for g, v in vals if g >= 0.95 * gmax]
    steady[c] = sum(ss) / len(ss)
with open(os.path.join(prof, f"{tag}_pmc_counters.csv"), "w") as f:
    f.write("counter,kernel,grid_size,value_KB\n")
    for r in rows:
        f.write(f"{r[0]},{r[1]},{r[2]},{r[3]}\n")
traffic = 2 * steady["FETCH_SIZE"] * 1024 + steady["WRITE_SIZE"] * 1024
json.dump({
    "round": tag, "kernel": "fused_kernel",
    "workload": "bench.py default (1 channel, N=1024, 2^26 samples/step), steady-state launches (all stages active)",
    "FETCH_SIZE_KiB": steady["FETCH_SIZE"], "WRITE_SIZE_KiB": steady["WRITE_SIZE"],
    "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 4 << 26,
    "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts half of streaming reads; calibration in "
            "profiles/README.md). Excess over algorithmic = inter-stage streams written once and read once "
            "(8/7 geometric tail) + per-workgroup partials.",
}, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
print(json.load(open(os.path.join(prof, f"{tag}_traffic.json"))))
