import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as e
from conftest import test_signal as make_signal
import test_gpu_parity as T
pkg, ora = e.load_package(), e.load_oracle()
for n in (400, 304, 496, 272):
    x = make_signal(pkg, 200*n+8, seed=n, tone=0.3)
    g = pkg.PsdCascadeBank(n)
    g.process(0, x)
    try:
        T.check_against_oracle(pkg, ora, g, [x], n, what=f"N={n}", justify=False)
        print(n, "ok")
    except AssertionError as ex:
        print(n, "FAIL", str(ex)[:200])
    g.close()
