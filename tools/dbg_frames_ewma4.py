"""which sample positions does the failing kernel variant mishandle?  one impulse per run, position swept over a chunk"""
import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
import __graft_entry__ as e
pkg, ora = e.load_package(), e.load_oracle()
n, batches = int(os.environ.get("DBG_N", "8192")), 22
per_frame = 8 * batches
nframes = (40 * n) // per_frame + 5
per = nframes * per_frame
S = 21 * n
w = np.sin(np.pi * np.arange(n) / n) ** 2
lsb = 4.096 * 2.5 / 32768.0
wire = np.zeros((4, per), dtype=np.int16)
wire[2:] = np.int16(-32768)  # DAC zero on the wire
data, fs = pkg.make_adcdac_frames(wire, batches, seq0=0)
buf = np.frombuffer(data, dtype=np.uint8).copy()
d = torch.from_numpy(buf).cuda()
def poke(pos, val):
    f, r = divmod(pos, per_frame); b, i = divmod(r, 8)
    off = f * fs + 8 + b * 64 + 0 * 16 + i * 2        # trace 0 (ADC0)
    d[off:off + 2] = torch.tensor(list(np.array([val], dtype='<i2').tobytes()), dtype=torch.uint8, device='cuda')
bad = []
step = int(os.environ.get("DBG_STEP", "1"))
for p in range(0, n, step):
    poke(S + p, 20000)
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    tot = float(np.sum(np.asarray(g.stage_spectrum(0, 0), dtype=np.float64)))
    g.close()
    poke(S + p, 0)
    # the impulse sits in the segments starting at S + p - q (q = p mod hop and q + hop)
    q = p % (n // 2)
    exp = (20000 * lsb) ** 2 * (w[q] ** 2 + w[q + n // 2] ** 2) * (n // 2 + 1)
    if abs(tot - exp) > 1e-3 * max(exp, 1e-12) + 1e-9:
        bad.append((p, tot / exp if exp else float('inf')))
print("positions mishandled:", len(bad), "of", n // step)
print(bad[:64])
ps = np.array([b[0] for b in bad])
if ps.size:
    print("p mod 4 histogram", np.bincount(ps % 4, minlength=4), " p mod 2048 range", (ps % 2048).min(), (ps % 2048).max(), " p//2048 histogram", np.bincount(ps // 2048, minlength=4))
    print("lane = (p mod 2048)//4 histogram of low 6 bits:", np.bincount(((ps % 2048) // 4) % 64, minlength=64))
