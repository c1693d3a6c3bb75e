"""Fuzz of the sizes that are not powers of two (and the generic kernels under caller-built windows): random N = 16 k <= 8192,
detrend, a Hann window or a caller's table with a random admissible overlap, host-fed in odd chunks and device-fed, against the
f64 oracle (which evaluates the DFT by its definition: few segments at large N).  usage: python tools/anyn_campaign.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as entry
import test_gpu_parity as T
from conftest import test_signal as make_signal

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    r = rng.random()
    if r < 0.5:
        n = 16 * int(rng.integers(2, 65))          # <= 1024
    elif r < 0.85:
        n = 16 * int(rng.integers(65, 257))        # <= 4096
    else:
        n = 16 * int(rng.integers(257, 513))       # <= 8192
    detrend = ["none", "midpoint", "span", "mean"][int(rng.integers(0, 4))]
    custom = rng.random() < 0.4
    if custom:
        hop = 8 * int(rng.integers(1, n // 8 + 1))  # (N - overlap) % 8 == 0, overlap < N
        i = np.arange(n, dtype=np.float64)
        w = (0.54 - 0.46 * np.cos(2 * np.pi * i / n)).astype(np.float32)
        m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
        win = pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), n - hop)
        window, owin = win, win.as_tuple()
    else:
        hop, window, owin = n // 2, pkg.Window.HANN, "hann"
    nseg = int(rng.integers(20, 400 if n <= 1024 else 60 if n <= 4096 else 24))
    total = n + hop * nseg + int(rng.integers(0, hop))
    x = make_signal(pkg, total, seed=seed, tone=0.2, dc=0.1)
    g = pkg.PsdCascadeBank(n, window=window)
    g.set_detrend(pkg.Detrend[detrend.upper()])
    cut = int(rng.integers(1, total))
    g.process(0, x[:cut])
    d = torch.from_numpy(x[cut:].copy()).cuda()
    g.process_device(0, d.data_ptr(), total - cut)
    tag = f"seed {seed} n={n} {detrend} {'custom overlap ' + str(n - hop) if custom else 'hann'} segs~{nseg}"
    try:
        T.check_against_oracle(pkg, ora, g, [x], n, detrend=detrend, window=owin, what=tag, justify=False)
        print(tag, f"ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(tag, "FAIL", str(e)[:300], flush=True)
    g.close()
print("failures:", bad)
