import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import torch, numpy as np
pkg = entry.load_package()
n, T = int(os.environ.get("PSD_N", "1024")), 1 << 26
bank = pkg.PsdCascadeBank(n, 1)
d = torch.empty(T, dtype=torch.float32, device="cuda")
pkg.fill_noise_device(d.data_ptr(), T, seed=1)
for _ in range(63):
    bank.process_device(0, d.data_ptr(), T)
bank.sync()
def tm(label, f):
    t = time.perf_counter(); r = f(); print(f"{label}: {(time.perf_counter()-t)*1e3:.3f} ms"); return r
tm("import shard", lambda: __import__("stabilizer_stream_amd.shard"))
from stabilizer_stream_amd import shard
tm("num_stages", lambda: bank.num_stages(0))
tm("read_channel #1", lambda: bank.read_channel(0))
tm("read_channel #2", lambda: bank.read_channel(0))
tm("stage_spectrum", lambda: bank.stage_spectrum(0, 0))
tm("stage_info", lambda: bank.stage_info(0, 0))
spec, meta = tm("pack_readout", lambda: shard.pack_readout(bank, 1, n, torch))
tm("stitch", lambda: shard.stitch_gathered(pkg, n, [spec], [meta], [1]))
tm("psd", lambda: bank.psd(0))
tm("process+sync", lambda: (bank.process_device(0, d.data_ptr(), T), bank.sync()))
tm("read_channel #3", lambda: bank.read_channel(0))
