"""Create / feed / read / destroy handles in a loop and watch the device's free memory and the process RSS."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
x = pkg.noise_host(1 << 20, 5)
d = torch.from_numpy(x).cuda()
frames, fs = pkg.make_adcdac_frames((np.random.default_rng(1).standard_normal((4, 22 * 8 * 200)) * 3000).astype(np.int16), 22)
def cycle(i):
    n = (256, 1024, 4096, 16384)[i % 4]
    g = pkg.PsdCascadeBank(n, 4)
    g.set_detrend(pkg.Detrend(i % 4))
    if i % 3 == 0:
        g.set_avg(pkg.AvgOpts(50, 5000))
    g.process(0, x)
    g.process_device(1, d.data_ptr(), x.size)
    g.process_device(1, d.data_ptr(), x.size)
    g.process_adcdac_frames(frames, fs)
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
