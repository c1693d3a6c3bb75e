import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import __graft_entry__ as entry
pkg, ora = entry.load_package(), entry.load_oracle()
n, batches = 2048, 22
per_frame = batches * 8
nframes = (40 * n) // per_frame + 3
lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)
for kind in ("ramp", "noise"):
    if kind == "ramp":
        raw = np.stack([((np.arange(nframes * per_frame) % 20000) - 10000 + 1000 * c).astype(np.int16) for c in range(4)])
    else:
        raw = np.random.default_rng(1).integers(-3000, 3000, size=(4, nframes * per_frame)).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, batches)
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for coalesce in (1,):
        g = pkg.PsdCascadeBank(n, 4)
        g.configure(coalesce=coalesce)
        assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
        for c in (0, 2):
            x = raw[c].astype(np.float32) * lsb if c < 2 else (raw[c].view(np.uint16) ^ np.uint16(0x8000)).view(np.int16).astype(np.float32) * lsb
            r = ora.PsdCascade(n, "f64"); r.process(x)
            for k in range(g.num_stages(c)):
                gi, ri = g.stage_info(c, k), r.stage_info(k)
                gb, rb = g.stage_buf(c, k), r.stage_buf(k)
                sp_g, sp_r = g.stage_spectrum(c, k).astype(np.float64), r.stage_spectrum(k)
                rel = np.max(np.abs(sp_g - sp_r)) / np.max(np.abs(sp_r)) if ri["count"] else 0
                print(kind, "ch", c, "stage", k, gi == ri, "buf maxdiff", float(np.max(np.abs(gb - rb))) if rb.size else 0, "of", float(np.max(np.abs(rb))) if rb.size else 0,
                      "spec maxdiff/max", rel)
                if k == 1 and rb.size:
                    bad = np.flatnonzero(np.abs(gb - rb) > 1e-3 * np.max(np.abs(rb)))
                    print("   stage-1 pending: bad idx", bad[:10], "of", rb.size, "first vals", gb[:4], rb[:4])
        g.close()
