"""Phase breakdown of a workgroup-level fused kernel (GPU box): tools/stamps/build_big.sh N first.
usage: PSDC_LIB=tools/stamps/libpsdcascade_bstamps_<N>.so python tools/stamps/run_big.py N [log2 samples]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
os.environ.setdefault("PSDC_LIB", os.path.join(ROOT, "tools", "stamps", f"libpsdcascade_bstamps_{n}.so"))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
log2 = int(sys.argv[2]) if len(sys.argv) > 2 else 26
total = 1 << log2
x = torch.empty(total, dtype=torch.float32, device="cuda:0")
pkg.fill_noise_device(x.data_ptr(), total, seed=0x7654321)
torch.cuda.synchronize()
bank = pkg.PsdCascadeBank(n, n_channels=1)
for _ in range(40):
    bank.process_device(0, x.data_ptr(), total)
bank.sync()
out = (C.c_ulonglong * 16)()
rc = L.psdc_debug_stamps_big(out)
assert rc == 0, rc
names = ["(between pairs)", "state + samples -> LDS, barrier", "stage A, barrier", "stage B, barrier", "stage C + state save, barrier",
         "tables + window + pass 0, barrier", "pass A + look-ahead issue, barrier", "pass B, barrier", "pass C + |Z|^2, barrier"]
run = out[12]
tot = sum(out[k] for k in range(9))
print(f"N={n} run={run} pairs; shader cycles per pair (s_memtime), wave 0 of workgroup 0")
for k, nm in enumerate(names):
    print(f"  {nm:40s} {out[k] / max(run, 1):9.1f}  {100.0 * out[k] / max(tot, 1):5.1f}%")
print(f"  total {tot / max(run, 1):.1f} cycles/pair")
