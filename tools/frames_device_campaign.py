"""Fuzz of psdc_process_adcdac_frames_device with the frames read in place (N >= 256): random batches per frame
(1..31), call sizes from a couple of frames to thousands, mid-stream read-outs, coalescing depths, detrends; the four
traces must track their oracle cascades and the Loss counters.  usage: python tools/frames_device_campaign.py [first] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as entry
import test_gpu_parity as T
import test_gpu_frames_inplace as F

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = [2048, 4096, 256, 8192, 16384, 512, 1024, 4096][seed % 8]
    batches = int(rng.integers(1, 32))
    per_frame = batches * 8
    nframes = int(rng.integers(30 * n, 260 * n)) // per_frame
    detrend = ["none", "midpoint", "span", "mean"][int(rng.integers(0, 4))]
    avg = None if rng.random() < 0.55 else pkg.AvgOpts(int(rng.integers(1, 60)), int(rng.integers(1, 3000)))  # (the EWMA + FRAMES kernel variants)
    try:
        buf, fs, traces = F.make_frames(pkg, ora, nframes, batches, seed=seed, seq0=int(rng.integers(0, 2**32)))
        d = torch.from_numpy(buf.reshape(-1)).cuda()
        g = pkg.PsdCascadeBank(n, 4)
        g.set_detrend(pkg.Detrend[detrend.upper()])
        if avg is not None:
            g.set_avg(avg)
        co = int(rng.integers(1, 9))
        g.configure(coalesce=co, eager=rng.random() < 0.5)
        pos = 0
        while pos < nframes:
            big = rng.random() < 0.7
            m = int(min(nframes - pos, rng.integers(4 * (n + 288) // per_frame + 1, 90 * n // per_frame) if big else rng.integers(1, 60)))
            assert g.process_adcdac_frames_device(d.data_ptr() + pos * fs, fs, m) == m
            pos += m
            if rng.random() < 0.15:
                g.num_stages(int(rng.integers(0, 4)))
        assert g.loss() == {"received": nframes * batches, "dropped": 0}, g.loss()
        for c in range(4):
            T.check_against_oracle(pkg, ora, g, [traces[c]], n, detrend=detrend, avg=avg, channel=c, what=f"seed {seed} trace {c}")
        g.close()
        print(f"seed {seed} n={n} batches={batches} frames={nframes} {detrend} avg={avg} coalesce {co} ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed} n={n} batches={batches} frames={nframes} FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
