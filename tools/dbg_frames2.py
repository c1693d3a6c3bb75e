import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
pkg, ora = entry.load_package(), entry.load_oracle()
n, batches = 2048, 22
per_frame = batches * 8
nframes = (40 * n) // per_frame + 3
T = nframes * per_frame
lsb = float(np.float32(4.096) * np.float32(2.5) / np.float32(32768))
def total_power(m):  # impulse at offset m within the hop: sum over the two segments that hold it of w^2
    return np.sin(np.pi * m / n) ** 4 + np.cos(np.pi * m / n) ** 4
ms = np.arange(0, n // 2)
table = total_power(ms + 0.0)
for i in range(0, 16):
    pos = 9 * 1024 + 100 + i
    raw = np.zeros((4, T), dtype=np.int16)
    raw[0, pos] = 10000
    data, fs = pkg.make_adcdac_frames(raw, batches)
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    p = float(np.mean(g.stage_spectrum(0, 0)[8:1000])) / (10000 * lsb) ** 2
    # which offsets m in [0, 1024) are compatible?  (symmetric about 512: m and 1024 - m, and 512 - m ...)
    cand = np.flatnonzero(np.abs(table - p) < 2e-6)
    print(f"impulse at {pos} (offset in hop {pos % 1024}): power {p:.7f} expected {total_power(pos % 1024):.7f} -> compatible offsets {cand[:8]}")
    g.close()
