"""One-off fuzz of tests/test_gpu_parity.py::test_several_channels_short_scattered_spans over random shapes (GPU box): several channels
fed in lockstep in short scattered spans at the library's own depth -- N, channel count, span length and span count at random.
usage: python tools/multichannel_campaign.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
import test_gpu_parity as T

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fn = getattr(T.test_several_channels_short_scattered_spans, "__wrapped__", T.test_several_channels_short_scattered_spans)
bad, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([256, 512, 1024, 1024, 2048, 4096]))
    nch = int(rng.integers(2, 9))
    lo = int(np.ceil(np.log2(4 * (n + 288))))  # a span is read in place from 4 (n + 288) samples
    piece_log2 = int(rng.integers(lo, lo + 3))
    npieces = int(rng.integers(12, 90))
    try:
        fn(pkg, ora, None, n, nch, piece_log2, npieces)
        print(f"seed {seed} n={n} channels={nch} spans={npieces} x 2^{piece_log2} ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed} n={n} channels={nch} spans={npieces} x 2^{piece_log2} FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
