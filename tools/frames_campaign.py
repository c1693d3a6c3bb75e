"""One-off fuzz of psdc_process_adcdac_frames: random batches per frame (1..31), frame counts per
call, FFT sizes, optional error frames; the four traces must track their oracle cascades and the
loss counters the oracle's Loss restatement.  usage: python tools/frames_campaign.py [first] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
import test_gpu_parity as T

pkg, ora = entry.load_package(), entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = [64, 256, 512, 1024, 4096][seed % 5]
    batches = int(rng.integers(1, 32))
    nframes = int(rng.integers(40, 4000))
    raw = rng.integers(-32768, 32768, size=(4, nframes * batches * 8)).astype(np.int16)
    seq0 = int(rng.integers(0, 2**32 - 10))
    data, fs = pkg.make_adcdac_frames(raw, batches, seq0=seq0)
    try:
        g = pkg.PsdCascadeBank(n, 4)
        pos = 0
        while pos < nframes:
            m = int(min(nframes - pos, rng.integers(1, 1500)))
            assert g.process_adcdac_frames(data[pos * fs:(pos + m) * fs], fs) == m
            pos += m
            if rng.random() < 0.1:
                g.num_stages(int(rng.integers(0, 4)))
        assert g.loss() == {"received": nframes * batches, "dropped": 0}, g.loss()
        traces = [[] for _ in range(4)]
        for f in range(nframes):
            st, seq, nb, tr = ora.adcdac_decode(data[f * fs:(f + 1) * fs])
            assert st == 0 and nb == batches and seq == (seq0 + f * batches) % 2**32
            for c in range(4):
                traces[c].append(tr[c])
        for c in range(4):
            T.check_against_oracle(pkg, ora, g, [np.concatenate(traces[c])], n, channel=c, what=f"seed {seed} trace {c}")
        g.close()
        print(f"seed {seed} n={n} batches={batches} frames={nframes} ok ({time.time() - t0:.0f}s)", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed} n={n} batches={batches} frames={nframes} FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
