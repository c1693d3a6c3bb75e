#!/usr/bin/env python3
"""usage: tools/trace_gaps.py KERNEL_TRACE.csv [last_fraction]
Busy time against wall time over the last part of a rocprofv3 kernel trace (one stream of kernels): how much of the timed region the GPU
spent between kernels, and the distribution of those gaps -- tells a launch- / host-bound loop from a kernel-bound one."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = sorted(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(rows, rows[1:]))
q = lambda p: gaps[int(p * (len(gaps) - 1))] / 1e3
print(f"{len(rows)} kernels: busy {busy / 1e6:.2f} ms of {wall / 1e6:.2f} ms wall ({busy / wall:.3f}); gaps us: median {q(.5):.1f}, p90 {q(.9):.1f}, p99 {q(.99):.1f}, max {q(1):.1f}")
