"""N>1 path on CPU: world_size-2 gloo run of the channel shard -> gather -> stitch read-out.

The HIP kernels cannot run here, so each rank fills its "bank" from the CPU oracle (test
infrastructure); what is under test is the product's sharding, the single-gather payload and
the host stitch (psdc_stitch) on rank 0, compared with each channel's own PsdCascade::psd."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBank:
    """Duck-types PsdCascadeBank's read-out methods over oracle cascades (f32 mirror)."""

    def __init__(self, ora, n, streams):
        self.cs = []
        for x in streams:
            c = ora.PsdCascade(n, "f32")
            c.process(x)
            self.cs.append(c)

    def num_stages(self, c):
        return self.cs[c].num_stages

    def stage_info(self, c, k):
        return self.cs[c].stage_info(k)

    def stage_spectrum(self, c, k):
        return self.cs[c].stage_spectrum(k)


def _worker(rank, world, port, n, n_channels, total, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg, ora = entry.load_package(), entry.load_oracle()
    from stabilizer_stream_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.channel_shard(n_channels, world, rank)
    streams = [pkg.noise_host(total + 1000 * g, 0x7654321 + g) for g in mine]
    bank = OracleBank(ora, n, streams)
    width = max(len(shard.channel_shard(n_channels, world, r)) for r in range(world))
    rec = shard.pack_readout(bank, len(mine), n, pkg, pad_to=width)  # equal blocks for the gather
    assert rec.size == pkg.readout_bytes(n, width)
    recs = shard.gather_readout(dist, rec)
    if rank == 0:
        per_rank = [len(shard.channel_shard(n_channels, world, r)) for r in range(world)]
        res = shard.stitch_gathered(pkg, recs, per_rank)
        ok = len(res) == n_channels
        for g, (p, br) in enumerate(res):
            ref = ora.PsdCascade(n, "f32")
            ref.process(pkg.noise_host(total + 1000 * g, 0x7654321 + g))
            pr, brr, _ = ref.psd()
            ok = ok and np.array_equal(p, pr) and [b.count for b in br] == [b["count"] for b in brr] \
                and [b.pending for b in br] == [b["pending"] for b in brr]
        open(out_path, "w").write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


def test_channel_shard_partition(pkg):
    from stabilizer_stream_amd import shard
    for nch, world in [(64, 8), (5, 2), (3, 4), (1, 1), (7, 3)]:
        parts = [shard.channel_shard(nch, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(nch))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


@pytest.mark.timeout(300)
def test_gather_and_stitch_world2_gloo(pkg, ora, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, 64, 5, 40000, out), nprocs=2, join=True)
    assert open(out).read() == "OK"
