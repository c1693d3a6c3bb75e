"""One-off fuzz of the zero-copy device-span planner (seam, pair-aligned split, alignment fallbacks,
drain skips, deferral) against the oracle: random N, detrend none/mean, plain or finite averaging,
1-7 device spans of random (mostly long, sometimes tiny / unaligned) lengths, optional mid-stream
read-outs.  usage: python tools/span_campaign.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as entry
import test_gpu_parity as T
from conftest import test_signal as make_signal

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sizes = [64, 256, 512, 1024, 1024, 2048, 4096, 8192, 16384]
sizes = [int(v) for v in os.environ.get("FUZZ_SIZES", "").split(",") if v] or sizes  # e.g. FUZZ_SIZES=8192,16384
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = sizes[seed % len(sizes)]
    detrend = ["none", "mean"][int(rng.integers(0, 2))]
    avg = None if rng.random() < 0.6 else (int(rng.integers(1, 60)), int(rng.integers(1, 3000)))
    nspans = int(rng.integers(1, 8)) if rng.random() < 0.6 else int(rng.integers(8, 24))
    lens = []
    for _ in range(nspans):
        r = rng.random()
        if r < 0.15:
            lens.append(int(rng.integers(1, 4 * n)))                      # short: copied
        elif r < 0.5:
            lens.append(int(rng.integers(4 * (n + 288), 40 * n)))         # just long enough for in-place
        else:
            lens.append(int(rng.integers(40 * n, 600 * n)) & ~3 if rng.random() < 0.7 else int(rng.integers(40 * n, 600 * n)))
    total = sum(lens)
    x = make_signal(pkg, total, seed=seed, tone=0.2)
    ofs = int(rng.integers(0, 4))                                          # device pointer alignment of the stream start
    xd = torch.zeros(total + 8, dtype=torch.float32, device="cuda")
    xd[ofs:ofs + total] = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    try:
        g = pkg.PsdCascadeBank(n)
        co = int(rng.choice([1, 4, 16, -2, -3, -4, -8, -11, -16]))  # negative: spans are held until the round is full; positive: PSDC_OPT_EAGER
        g.configure(coalesce=co, eager=co > 0, merge=bool(seed % 4 == 0))
        g.set_detrend(pkg.Detrend[detrend.upper()])
        if avg:
            g.set_avg(pkg.AvgOpts(*avg))
        chunks, a = [], 0
        for m in lens:
            g.process_device(0, xd.data_ptr() + 4 * (ofs + a), m)
            chunks.append(x[a:a + m])
            a += m
            if rng.random() < 0.2:
                g.num_stages(0)                                            # mid-stream read-out (drains the pipeline)
        g.sync()
        T.check_against_oracle(pkg, ora, g, chunks, n, detrend=detrend, avg=pkg.AvgOpts(*avg) if avg else None, what=f"seed {seed}")
        g.close()
        print(f"seed {seed} n={n} {detrend} avg={avg} coalesce={co} spans={lens} ofs={ofs} ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed} n={n} {detrend} avg={avg} coalesce={co} spans={lens} ofs={ofs} FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
