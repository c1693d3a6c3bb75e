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


# ---- world size 2: uneven shards across real peers ----------------------------------------------------------------------------
CHILD_WORLD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["PSDC_ROOT"])
import torch
import torch.distributed as dist
import __graft_entry__ as entry
pkg = entry.load_package()
from stabilizer_stream_amd import shard

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
backend, out_dir = os.environ["T_BACKEND"], os.environ["T_OUT"]
n, total = int(os.environ["T_N"]), int(os.environ["T_TOTAL"])
split = [int(v) for v in os.environ["T_SPLIT"].split(",")]          # channels per rank: uneven on purpose
assert len(split) == world
dev = local % torch.cuda.device_count()                             # one GPU per rank where the box has them
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
first = sum(split[:rank])
mine = list(range(first, first + split[rank]))                      # global channel ids of this rank
bank = pkg.PsdCascadeBank(n, len(mine), device=dev)
bank.set_detrend(pkg.Detrend.MEAN)
x = torch.empty(total, dtype=torch.float32, device=torch.device("cuda", dev))
for c, g in enumerate(mine):                                        # the stream of GLOBAL channel g, wherever it lives
    m = total - 4096 * g - 3
    pkg.fill_noise_device(x.data_ptr(), m, 0x7654321 + g, device=dev)
    bank.process_device(c, x.data_ptr(), m // 2)
    bank.process_device(c, x.data_ptr() + 4 * (m // 2), m - m // 2)
    bank.sync()
rec = shard.pack_readout(bank, len(mine), n, pkg, pad_to=max(split))    # every rank gathers max(split) rows (psdc_pack_pad)
assert rec.size == pkg.readout_bytes(n, max(split))
OPTS = (pkg.MergeOpts(), pkg.MergeOpts(True, 0, True), pkg.MergeOpts(False, 3, False))
own = {}
for c, g in enumerate(mine):                                        # what psdc_psd says on the shard itself
    for oi, opts in enumerate(OPTS):
        p, br = bank.psd(c, opts)
        own[f"p_{g}_{oi}"] = p
        own[f"b_{g}_{oi}"] = np.array([[b.start, b.include, b.count, b.avg, b.bins.start, b.bins.stop, b.pending, b.processed] for b in br],
                                      dtype=np.int64)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **own)
recs = shard.gather_readout(dist, rec, device=torch.device("cuda", dev) if backend == "nccl" else None)
dist.barrier()                                                      # (every rank's file is written before rank 0 reads it)
res = {"rank": rank, "device": dev, "channels": len(mine)}
if rank != 0:
    assert recs is None                                             # only the destination holds the records (shard.py)
    res["ok"] = True
else:
    assert recs is not None and len(recs) == world
    ok, worst = True, ""
    for oi, opts in enumerate(OPTS):
        stitched = shard.stitch_gathered(pkg, recs, split, opts)    # rank-major = global channel order
        assert len(stitched) == sum(split)
        for g, (p, br) in enumerate(stitched):
            r = next(i for i in range(world) if g < sum(split[:i + 1]))
            f = np.load(os.path.join(out_dir, f"rank{r}.npz"))
            b = np.array([[q.start, q.include, q.count, q.avg, q.bins.start, q.bins.stop, q.pending, q.processed] for q in br], dtype=np.int64)
            same = np.array_equal(p, f[f"p_{g}_{oi}"], equal_nan=True) and np.array_equal(b, f[f"b_{g}_{oi}"]) and len(br) >= 4
            if not same:
                worst = f"global channel {g} (rank {r}) {opts}"
            ok = ok and same
    res.update({"ok": bool(ok), "worst": worst, "pad_rows_empty": int(pkg.unpack_info(recs[1], max(split) - 1)[2]) == 0,
                "bytes": [int(r_.size) for r_ in recs]})
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""


def _run_world(tmp_path, backend, world, split, n=1024, total=1 << 21):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_world_child.py"
    script.write_text(CHILD_WORLD)
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                    "LOCAL_RANK": str(rank), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PSDC_ROOT": ROOT, "T_N": str(n), "T_TOTAL": str(total),
                    "T_SPLIT": ",".join(str(v) for v in split), "T_BACKEND": backend, "T_OUT": str(tmp_path)})
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=500)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o))
    results = []
    for rc, o in outs:
        lines = [ln for ln in o.splitlines() if ln.startswith("RESULT ")]
        assert rc == 0 and lines, o[-3000:]
        results.append(json.loads(lines[-1][7:]))
    return results


def _check_world(results, split, n):
    import __graft_entry__ as entry
    pkg = entry.load_package()
    results = sorted(results, key=lambda r: r["rank"])
    assert [r["channels"] for r in results] == list(split) and all(r["ok"] for r in results), results
    r0 = results[0]
    assert r0["pad_rows_empty"] and r0["bytes"] == [pkg.readout_bytes(n, max(split))] * len(split), r0
    return results


@pytest.mark.timeout(600)
def test_rccl_gather_two_ranks_uneven_shards(gpu_required, tmp_path):
    """Two ranks, one GPU each (cuda:LOCAL_RANK), eight channels split 5 + 3: each rank packs its shard padded to five rows
    (psdc_pack_pad), ONE RCCL gather to rank 0, rank 1 sees None, rank 0 stitches all eight channels and every one is bit-identical
    (three MergeOpts, every Break field) to what psdc_psd said on the rank that owns it.  Needs two visible GPUs: skipped on the
    one-GPU box, where test_gather_two_ranks_on_one_gpu_over_gloo runs the same child over gloo."""
    import torch
    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU in this, the parent, process)
    if ndev < 2:
        pytest.skip(f"two-rank RCCL gather needs 2 GPUs, torch.cuda.device_count() = {ndev}")
    res = _check_world(_run_world(tmp_path, "nccl", 2, (5, 3)), (5, 3), 1024)
    assert [r["device"] for r in res] == [0, 1]


@pytest.mark.timeout(600)
def test_gather_two_ranks_on_one_gpu_over_gloo(gpu_required, tmp_path):
    """The same two-rank child with the records gathered over gloo and both ranks' cascades on whatever GPUs are visible (one here):
    rank != 0 behaviour, uneven 5 + 3 shards padded with psdc_pack_pad across real peer processes, and the stitch on rank 0 against
    the per-channel psdc_psd records both ranks wrote -- everything of the two-rank path except the RCCL transport itself, which
    the one-rank test above and (where two GPUs exist) the test before this one cover."""
    _check_world(_run_world(tmp_path, "gloo", 2, (5, 3)), (5, 3), 1024)
