import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
import torch, numpy as np
pkg = entry.load_package()
from stabilizer_stream_amd import shard
n, T = int(os.environ.get("PSD_N", "512")), 1 << 26
bank = pkg.PsdCascadeBank(n, 1)
d = torch.empty(T, dtype=torch.float32, device="cuda")
pkg.fill_noise_device(d.data_ptr(), T, seed=1)
def tm(label, f):
    t = time.perf_counter(); r = f(); print(f"{label}: {(time.perf_counter()-t)*1e3:.3f} ms"); return r
for _ in range(3):
    bank.process_device(0, d.data_ptr(), T)
tm("warm readout", lambda: shard.pack_readout(bank, 1, n, torch))
bank.sync()
for _ in range(60):
    bank.process_device(0, d.data_ptr(), T)
tm("sync (incl drain)", lambda: bank.sync())
tm("num_stages", lambda: bank.num_stages(0))
tm("read_channel cap0", lambda: bank._L.psdc_read_channel(bank._h, 0, 0, None, None, None))
tm("read_channel", lambda: bank.read_channel(0))
tm("pack_readout", lambda: shard.pack_readout(bank, 1, n, torch))
spec, meta = shard.pack_readout(bank, 1, n, torch)
tm("stitch", lambda: shard.stitch_gathered(pkg, n, [spec], [meta], [1]))
print("--- second pass ---")
for _ in range(60):
    bank.process_device(0, d.data_ptr(), T)
tm("sync (incl drain)", lambda: bank.sync())
tm("stage_spectrum(0,0)", lambda: bank.stage_spectrum(0, 0))
tm("stage_spectrum(0,8)", lambda: bank.stage_spectrum(0, bank.num_stages(0) - 1))
tm("read_channel", lambda: bank.read_channel(0))
tm("read_channel", lambda: bank.read_channel(0))
print("stages", bank.num_stages(0))
