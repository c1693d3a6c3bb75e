import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
import __graft_entry__ as e
import test_gpu_frames_inplace as F
pkg, ora = e.load_package(), e.load_oracle()
n, batches = 8192, 22
nframes = (220 * n) // (8 * batches) + 5
buf, fs, traces = F.make_frames(pkg, ora, nframes, batches, seed=1)
d = torch.from_numpy(buf.reshape(-1)).cuda()
import os
for avg in ([tuple(int(v) for v in a.split(',')) for a in os.environ['DBG_AVGS'].split(';')] if os.environ.get('DBG_AVGS') else (None, (5, 100), (1000000, 1000000))):
    for calls in (1, 3):
        g = pkg.PsdCascadeBank(n, 4)
        if avg: g.set_avg(pkg.AvgOpts(*avg))
        step = nframes // calls
        pos = 0
        for i in range(calls):
            m = nframes - pos if i == calls - 1 else step
            assert g.process_adcdac_frames_device(d.data_ptr() + pos * fs, fs, m) == m
            pos += m
        ref = ora.PsdCascade(n, "f64")
        if avg: ref.set_avg(*avg)
        ref.process(traces[0])
        sg = np.asarray(g.stage_spectrum(0, 0), dtype=np.float64); sr = np.asarray(ref.stage_spectrum(0), dtype=np.float64)
        rel = np.abs(sg - sr) / np.max(sr)
        print("avg", avg, "calls", calls, "count", g.stage_info(0,0)["count"], ref.stage_info(0)["count"], "max rel err", rel.max(), "bins bad", int((rel > 1e-4).sum()))
        g.close()
import sys as _s
if os.environ.get("DBG_AVGS"): _s.exit(0)
print("---- detail")
for n in (8192, 4096):
    nframes = (220 * n) // (8 * batches) + 5
    buf, fs, traces = F.make_frames(pkg, ora, nframes, batches, seed=1)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4); g.set_avg(pkg.AvgOpts(5, 100))
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    # the same traces as f32 device spans on another handle (EWMA kernel without frames)
    h = pkg.PsdCascadeBank(n, 4); h.set_avg(pkg.AvgOpts(5, 100))
    dx = [torch.from_numpy(traces[c]).cuda() for c in range(4)]
    for c in range(4): h.process_device(c, dx[c].data_ptr(), traces[c].size)
    for c in (0, 3):
        ref = ora.PsdCascade(n, "f64"); ref.set_avg(5, 100); ref.process(traces[c])
        for k in range(ref.num_stages):
            sr = np.asarray(ref.stage_spectrum(k), dtype=np.float64)
            sg = np.asarray(g.stage_spectrum(c, k), dtype=np.float64)
            sh = np.asarray(h.stage_spectrum(c, k), dtype=np.float64)
            m = sr > 1e-3 * sr.max()
            print(f"N={n} trace {c} stage {k} count {ref.stage_info(k)['count']}: frames ratio min/max {np.min(sg[m]/sr[m]):.6f} {np.max(sg[m]/sr[m]):.6f}   f32-span ratio {np.min(sh[m]/sr[m]):.6f} {np.max(sh[m]/sr[m]):.6f}")
