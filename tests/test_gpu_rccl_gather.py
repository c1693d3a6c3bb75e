"""The RCCL branch of the read-out gather on a real GPU (src/bin/psd.rs:174-182: one cascade per trace).

`shard.gather_readout(dist, rec, device=cuda)` is what `bench.py --gpus N` runs at read-out: one H2D copy of the packed
record, ONE `dist.gather` (backend "nccl" = RCCL on ROCm), one D2H copy on rank 0, then the host stitch per channel
(psdc_unpack_stitch).  The CPU suite drives the same code over gloo (tests/test_multi_gpu_gloo.py); here the collective
itself is RCCL.  The box has ONE GPU, so the group has one rank -- the communicator is created, the gather kernel runs
on the device and its output is what gets stitched; ranks > 1 change the peer count, not the code path.  The group lives
in a fresh child process (an `nccl` process group cannot be re-initialised inside the long-lived pytest process, and a
failed RCCL init must not take the suite down with it)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["PSDC_ROOT"])
import torch
import torch.distributed as dist
import __graft_entry__ as entry
pkg = entry.load_package()
from stabilizer_stream_amd import shard

n, n_channels, total = int(os.environ["T_N"]), int(os.environ["T_CH"]), int(os.environ["T_TOTAL"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
mine = shard.channel_shard(n_channels, dist.get_world_size(), dist.get_rank())
bank = pkg.PsdCascadeBank(n, len(mine))
bank.set_detrend(pkg.Detrend.MEAN)
x = torch.empty(total, dtype=torch.float32, device="cuda")
for c, g in enumerate(mine):                      # device-resident streams, uneven lengths, two calls each
    m = total - 4096 * c - 3
    pkg.fill_noise_device(x.data_ptr(), m, 0x7654321 + g)
    bank.process_device(c, x.data_ptr(), m // 2)
    bank.process_device(c, x.data_ptr() + 4 * (m // 2), m - m // 2)
    bank.sync()
pad = len(mine) + int(os.environ.get("T_PAD", "0"))   # as if another rank held a larger shard
rec = shard.pack_readout(bank, len(mine), n, pkg, pad_to=pad)
assert rec.size == pkg.readout_bytes(n, pad)
recs = shard.gather_readout(dist, rec, device=torch.device("cuda", 0))
torch.cuda.synchronize()
assert recs is not None and len(recs) == 1 and recs[0].dtype == np.uint8
identical = bool(np.array_equal(recs[0], rec))
ok, worst = identical, ""
for opts in (pkg.MergeOpts(), pkg.MergeOpts(True, 0, True), pkg.MergeOpts(False, 3, False)):
    res = shard.stitch_gathered(pkg, recs, [len(mine)], opts)
    for c, (p, br) in enumerate(res):
        q, bq = bank.psd(c, opts)                 # psdc_psd on the shard itself
        same = np.array_equal(p, q, equal_nan=True) and br == bq and len(br) >= 4
        if not same:
            worst = f"channel {c} {opts}"
        ok = ok and same
empty = pkg.unpack_info(recs[0], pad - 1)[2] if pad > len(mine) else 0
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"ok": bool(ok), "identical": identical, "worst": worst, "channels": len(mine),
                              "bytes": int(rec.size), "empty_pad_stages": int(empty),
                              "stages": [bank.num_stages(c) for c in range(len(mine))]}))
"""


@pytest.mark.timeout(600)
@pytest.mark.parametrize("pad", [0, 3])
def test_rccl_gather_of_a_real_bank_equals_psd(pkg, gpu_required, tmp_path, pad):
    """pack_readout -> gather_readout(device=cuda) over a 1-rank RCCL group -> stitch_gathered on an 8-channel bank fed on
    the GPU: every channel's merged PSD and breaks are bit-identical to psdc_psd on the shard (three MergeOpts), and the
    gathered bytes are the packed bytes.  pad = 3: the record padded as for an uneven shard (psdc_pack_pad)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PSDC_ROOT": ROOT, "T_N": "1024", "T_CH": "8", "T_TOTAL": str(1 << 21),
                "T_PAD": str(pad)})
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=560)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout + r.stderr)[-3000:]
    res = json.loads(lines[-1][7:])
    assert res["ok"] and res["identical"], res
    assert res["channels"] == 8 and min(res["stages"]) >= 4 and res["empty_pad_stages"] == 0
    assert res["bytes"] == pkg.readout_bytes(1024, 8 + pad)
