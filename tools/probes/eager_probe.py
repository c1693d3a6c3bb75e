import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as e
pkg = e.load_package()
n, m, ns = 1024, 1 << 22, 40
x = torch.empty(m * ns, dtype=torch.float32, device="cuda")
pkg.fill_noise_device(x.data_ptr(), m * ns, seed=1)
torch.cuda.synchronize()
def run(sleep, eager):
    g = pkg.PsdCascadeBank(n, 1)
    g.configure(eager=eager)
    for i in range(ns):
        g.process_device(0, x.data_ptr() + 4 * m * i, m)
        if sleep and i % 3 == 0:
            time.sleep(0.004)
    s = [g.stage_spectrum(0, k) for k in range(g.num_stages(0))]
    g.close()
    return s
for eager in (False, True):
    a, b = run(False, eager), run(True, eager)
    print("eager" if eager else "default", "bit-identical across host timing:", [bool(np.array_equal(p, q)) for p, q in zip(a, b)],
          "max rel diff", max(float(np.max(np.abs(p.astype(np.float64) - q) / q)) for p, q in zip(a, b) if q.min() > 0))
