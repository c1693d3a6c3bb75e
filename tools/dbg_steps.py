"""Per-step wall time of process_device + sync (looks for stalls: allocations, re-planning)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T = 1 << 26
x = torch.empty(T, dtype=torch.float32, device="cuda")
pkg.fill_noise_device(x.data_ptr(), T, seed=1)
torch.cuda.synchronize()
for rep in range(3):
    bank = pkg.PsdCascadeBank(n, 1)
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        bank.process_device(0, x.data_ptr(), T)
        bank.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(ts))
    out = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 2 * med]
    print(f"n={n} rep {rep}: median {med:.3f} ms, outliers {out}")
    bank.close()
