"""Config 5 by passes: per stage the largest deviation of the white-noise PSD from 2 in units of 1/sqrt(count)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
n, chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 1 << 28
passes = int(sys.argv[1])
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x7654321
d = [torch.empty(chunk, dtype=torch.float32, device="cuda") for _ in range(2)]
g = pkg.PsdCascadeBank(n)
t0 = time.perf_counter()
for i in range(passes):
    if i >= 2:
        g.sync()
    if os.environ.get("DBG_TORCH_NOISE"):
        d[i & 1].uniform_(-0.5, 0.5).mul_(12 ** 0.5)
        torch.cuda.synchronize()
    else:
        pkg.fill_noise_device(d[i & 1].data_ptr(), chunk, seed=seed, first_index=i * chunk)
    g.process_device(0, d[i & 1].data_ptr(), chunk)
g.sync()
print("passes", passes, "time", round(time.perf_counter() - t0, 2))
p, br = g.psd()
for k, b in enumerate(br):
    if not b.include:
        continue
    seg = p[b.start:b.start + len(b.bins)].astype(np.float64)
    dev = np.abs(seg * 0.5 - 1.0) * np.sqrt(b.count)
    j = int(np.argmax(dev))
    print(f"dec 8^{int(round(np.log(b.decimation) / np.log(8)))} count {b.count} bins {b.bins.start}..{b.bins.stop} mean {seg.mean():.5f} "
          f"max dev {dev[j]:.2f} sigma-units at bin {b.bins.start + j} value {seg[j]:.4f}; >5: {(dev > 5).sum()}")
    if dev[j] > 6:
        print("      around:", np.round(seg[max(0, j - 3):j + 4], 3), "outliers at", (np.nonzero(dev > 5)[0] + b.bins.start)[:12])
