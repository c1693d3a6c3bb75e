"""One-off wide run of tests/test_gpu_parity.py::test_randomized_feed_stress over many seeds (GPU box).
usage: python tools/stress_campaign.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
import test_gpu_parity as T

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sizes = [64, 256, 512, 1024, 1024, 2048, 4096, 8192, 16384]
sizes = [int(v) for v in os.environ.get("FUZZ_SIZES", "").split(",") if v] or sizes  # e.g. FUZZ_SIZES=8192,16384
bad = 0
t0 = time.time()
fn = getattr(T.test_randomized_feed_stress, "__wrapped__", T.test_randomized_feed_stress)
for seed in range(first, first + count):
    n = sizes[seed % len(sizes)]
    try:
        fn(pkg, ora, None, n, seed)
        print(f"seed {seed} n={n} ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed} n={n} FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
