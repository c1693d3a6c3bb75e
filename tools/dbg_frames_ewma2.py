import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
import __graft_entry__ as e
import test_gpu_frames_inplace as F
pkg, ora = e.load_package(), e.load_oracle()
n, batches = 8192, 22
nframes = (220 * n) // (8 * batches) + 5
buf, fs, traces = F.make_frames(pkg, ora, nframes, batches, seed=1)
d = torch.from_numpy(buf.reshape(-1)).cuda()
g = pkg.PsdCascadeBank(n, 4); import os
A = tuple(int(v) for v in os.environ.get('DBG_AVG', '5,100').split(','))
g.set_avg(pkg.AvgOpts(*A))
assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
sg = np.asarray(g.stage_spectrum(0, 0), dtype=np.float64)
x = traces[0].astype(np.float64)
w = np.sin(np.pi * np.arange(n) / n) ** 2
nseg = 1 + (x.size - n) // (n // 2)
K = int(os.environ.get('DBG_K', '60'))
P = []
for i in range(nseg - K, nseg):
    seg = x[i * (n // 2): i * (n // 2) + n] * w
    P.append(np.abs(np.fft.rfft(seg)) ** 2)
P = np.array(P).T  # bins x K
coef, *_ = np.linalg.lstsq(P, sg, rcond=None)
gamma = A[0] / (A[0] + 1.0)
print("segments", nseg, "fitted weights of the last 14 segments (newest last) vs gamma^k:")
for k in range(min(K, 24), 0, -1):
    print(f"  seg -{k}: fitted {coef[-k]:.4f}   expected {gamma ** (k - 1):.4f}")
