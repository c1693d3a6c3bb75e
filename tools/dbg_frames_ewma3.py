import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
import __graft_entry__ as e
pkg, ora = e.load_package(), e.load_oracle()
n, batches = 8192, 22
per_frame = 8 * batches
nframes = (60 * n) // per_frame + 5
per = nframes * per_frame
raw = np.full((4, per), 1000, dtype=np.int16)
raw[0] = (np.arange(per) % 7919).astype(np.int16)          # a sawtooth: every sample position identifiable
wire = raw.copy(); wire[2:] = (wire[2:].view(np.uint16) ^ np.uint16(0x8000)).view(np.int16)
data, fs = pkg.make_adcdac_frames(wire, batches, seq0=0)
d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
lsb = np.float32(4.096 * 2.5 / 32768.0)
for c in (0, 1):
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    x = raw[c].astype(np.float32) * lsb
    ref = ora.PsdCascade(n, "f64"); ref.process(x)
    sg = np.asarray(g.stage_spectrum(c, 0), dtype=np.float64); sr = np.asarray(ref.stage_spectrum(0), dtype=np.float64)
    err = sg - sr
    print(f"trace {c}: max |err| / max ref {np.max(np.abs(err)) / sr.max():.3e}; ref bins>10 max {sr[10:].max():.3e}; gpu bins>10 max {sg[10:].max():.3e}, mean {sg[10:].mean():.3e}")
    g.close()
# where are the wrong samples?  the constant trace's spectrum beyond DC IS the (windowed) error's power spectrum: its
# autocorrelation shows the extent and the spacing of the wrong samples
g = pkg.PsdCascadeBank(n, 4)
nf1 = (3 * n) // per_frame + 1          # few segments: a clean picture
assert g.process_adcdac_frames_device(d.data_ptr(), fs, nf1) == nf1
sg = np.asarray(g.stage_spectrum(1, 0), dtype=np.float64)
print("segments", g.stage_info(1, 0)["count"])
sg[:6] = 0
ac = np.fft.irfft(sg)
ac = ac / ac[0]
big = np.flatnonzero(np.abs(ac[:n // 2]) > 0.05)
print("lags with |autocorr| > 0.05:", big[:80], "count", big.size)
