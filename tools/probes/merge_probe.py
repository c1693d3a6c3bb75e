import sys, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as e
pkg = e.load_package()
for n, nch in ((1024, 1), (4096, 2), (256, 1)):
    total = 300 * n + 1234
    rng = np.random.default_rng(n + nch)
    xd = [torch.from_numpy(pkg.noise_host(total, 5 + c)).cuda() for c in range(nch)]
    torch.cuda.synchronize()
    one, many = pkg.PsdCascadeBank(n, nch), pkg.PsdCascadeBank(n, nch)
    one.configure(profile=True); many.configure(profile=True)
    for c in range(nch):
        one.process_device(c, xd[c].data_ptr(), total)
    pos = [0] * nch
    ncalls = 0
    while min(pos) < total:
        c = int(rng.integers(0, nch))
        if pos[c] >= total:
            continue
        m = int(min(total - pos[c], rng.choice([rng.integers(1, 40), rng.integers(1, 3 * n), rng.integers(3 * n, 20 * n)])))
        if pos[c] == 0:
            m = 6 * n + 8
        many.process_device(c, xd[c].data_ptr() + 4 * pos[c], m)
        pos[c] += m; ncalls += 1
    one.sync(); many.sync()
    print(n, nch, "calls", ncalls, "launches one/many", one.profile_read()["launches"], many.profile_read()["launches"])
    for c in range(nch):
        for k in range(one.num_stages(c)):
            a, b = one.stage_spectrum(c, k), many.stage_spectrum(c, k)
            print("  ch", c, "stage", k, one.stage_info(c, k)["count"], "equal", bool(np.array_equal(a, b)), "max rel", float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(a, 1e-30))) if a.max() > 0 else 0)
