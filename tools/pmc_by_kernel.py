#!/usr/bin/env python3
"""usage: tools/pmc_by_kernel.py TAG -- FETCH_SIZE / WRITE_SIZE of `tools/gpu.sh pmc TAG ...` summed per kernel (calls, average and
largest launch; KiB, FETCH_SIZE to be doubled on gfx950): which kernel of a multi-kernel path moves what."""
import csv, sys, collections, glob
tag = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc/{tag}_{c}_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] = max(a[2], float(r["Counter_Value"]))
    for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:4]:
        print(c, k, "calls", a[0], "avg", round(a[1] / a[0]), "max", round(a[2]))
