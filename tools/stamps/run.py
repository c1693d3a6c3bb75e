"""Phase breakdown of the fused kernel (GPU box): runs the bench workload once on the
instrumented library and prints the s_memtime ticks wave 0 of workgroup 0 spent per phase."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.LIB_PATH = os.path.join(ROOT, "tools", "stamps", "libpsdcascade_stamps.so")
L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
log2 = int(sys.argv[2]) if len(sys.argv) > 2 else 26
total = 1 << log2
x = torch.empty(total, dtype=torch.float32, device="cuda:0")
pkg.fill_noise_device(x.data_ptr(), total, seed=0x7654321)
torch.cuda.synchronize()
bank = pkg.PsdCascadeBank(n, n_channels=1)
for _ in range(3):
    bank.process_device(0, x.data_ptr(), total)
bank.sync()
out = (C.c_ulonglong * 16)()
rc = L.psdc_debug_stamps(out)
assert rc == 0, rc
names = ["(between pairs)", "dec: state+x -> LDS | N=1024: wait for the look-ahead loads", "dec: stage A", "dec: stage B", "dec: stage C + state save",
         "detrend/window -> v", "pass0 + store0", "load1 + pass1 + store1", "load2 + pass2 + |Z|^2"]
run = out[12]
tot = sum(out[k] for k in range(9))
print(f"N={n} run={run} pairs/team; shader cycles per pair (s_memtime)")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {out[k] / max(run, 1):9.1f}  {100.0 * out[k] / max(tot, 1):5.1f}%")
print(f"  total {tot / max(run, 1):.1f} cycles/pair, {tot} in the loop")
print(f"  once per run: tables/window {out[9]}, first loads + warm-up {out[10]}, combine + store {out[11]} cycles")
