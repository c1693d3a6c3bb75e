"""Create / feed / read / destroy handles in a loop and watch the device's free memory and the process RSS."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
x = pkg.noise_host(1 << 20, 5)
d = torch.from_numpy(x).cuda()
frames, fs = pkg.make_adcdac_frames((np.random.default_rng(1).standard_normal((4, 22 * 8 * 200)) * 3000).astype(np.int16), 22)
dframes = torch.from_numpy(np.frombuffer(frames, dtype=np.uint8).copy()).cuda()  # the same frames resident in HBM (read in place)
nfr = len(frames) // fs
def hamming(n):
    i = np.arange(n, dtype=np.float64)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * i / n)).astype(np.float32)
    m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
    return pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), n // 2)
def cycle(i):
    n = (256, 1024, 4096, 16384, 400)[i % 5]
    g = pkg.PsdCascadeBank(n, 4, window=hamming(n) if i % 7 == 3 else pkg.Window.HANN)
    g.set_detrend(pkg.Detrend(i % 4))
    if i % 3 == 0:
        g.set_avg(pkg.AvgOpts(50, 5000))
    g.process(0, x)
    g.process_device(1, d.data_ptr(), x.size)
    g.process_device(1, d.data_ptr(), x.size)
    g.process_adcdac_frames(frames, fs)
    g.process_adcdac_frames_device(dframes.data_ptr(), fs, nfr)
    g.process_adcdac_frames_device(dframes.data_ptr(), fs, nfr)
    rec = g.pack_readout()
    pkg.unpack_stitch(rec, 2)
    c = g.clone()
    p, br = g.psd(0)
    c.psd(1)
    c.close()
    g.close()
for i in range(8):
    cycle(i)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(400):
    cycle(i)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"device free: {free0 >> 20} MiB -> {free1 >> 20} MiB ({(free0 - free1) >> 20} MiB lost over 400 cycles); max RSS {rss0 >> 10} -> {rss1 >> 10} MiB")
assert free0 - free1 < (64 << 20) and rss1 - rss0 < (256 << 10), "leak"
print("ok")
