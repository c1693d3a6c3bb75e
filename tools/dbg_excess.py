"""Diagnostic: where does the GPU exceed the f32 reference arithmetic on the level-steps torture signal?
Per detrend and N: rms relative error (GPU vs f64, f32 oracle vs f64) of stage 0 and the worst bins."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import __graft_entry__ as entry
pkg, ora = entry.load_package(), entry.load_oracle()

def signal(n):
    rng = np.random.default_rng(n)
    total = 600 * n + 8 * 77
    x = pkg.noise_host(total, seed=900 + n).astype(np.float64)
    edges = np.sort(rng.integers(0, total, size=9))
    levels = [0.0, 1000.0, -500.0, 3.0e4, 3.0e4 + 7.0, 0.25, -2.0e3, 1.0e5, 0.0, 12.0]
    for lv, a, b in zip(levels, np.r_[0, edges], np.r_[edges, total]):
        x[a:b] += lv
    return x.astype(np.float32)

for n in (1024, 4096):
    x = signal(n)
    for det in ("none", "midpoint", "mean"):
        g = pkg.PsdCascadeBank(n)
        g.set_detrend(pkg.Detrend[det.upper()])
        d = torch.from_numpy(x).cuda()
        g.process_device(0, d.data_ptr(), x.size)
        r64, r32 = ora.PsdCascade(n, "f64"), ora.PsdCascade(n, "f32")
        for o in (r64, r32):
            o.set_detrend(det)
            o.process(x)
        for k in range(2):
            a, b, c = g.stage_spectrum(0, k).astype(np.float64), r64.stage_spectrum(k), r32.stage_spectrum(k).astype(np.float64)
            eg, ef = np.abs(a - b) / b, np.abs(c - b) / b
            top = np.argsort(eg)[-4:][::-1]
            print(f"N={n} {det:8s} stage {k}: rms rel err gpu {np.sqrt(np.mean(eg**2)):.3g} f32ref {np.sqrt(np.mean(ef**2)):.3g} | "
                  f"median gpu {np.median(eg):.3g} f32ref {np.median(ef):.3g} | worst gpu bins "
                  + ", ".join(f"{i}:{eg[i]:.2g}/{ef[i]:.2g}" for i in top))
        g.close()
