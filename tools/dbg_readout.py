"""Where does a slow read-out spend its time?  Repeats (feed K steps, read out) and times the pieces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
T = 1 << 26
x = torch.empty(T, dtype=torch.float32, device="cuda")
pkg.fill_noise_device(x.data_ptr(), T, seed=1)
torch.cuda.synchronize()
for rep in range(reps):
    bank = pkg.PsdCascadeBank(n, 1)
    for i in range(3):
        bank.process_device(0, x.data_ptr(), T)
    bank.read_channel(0)
    for i in range(30):
        bank.process_device(0, x.data_ptr(), T)
    t0 = time.perf_counter()
    bank.flush()              # enqueue the drain rounds (no wait)
    t1 = time.perf_counter()
    bank.sync()               # wait for them
    t2 = time.perf_counter()
    ns = bank.num_stages(0)
    t3 = time.perf_counter()
    infos, sp = bank.read_channel(0)
    t4 = time.perf_counter()
    print(f"rep {rep}: flush {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f}, num_stages {1e3*(t3-t2):.2f}, read_channel {1e3*(t4-t3):.2f}", flush=True)
    bank.close()
