"""Channel sharding across the GPUs of one node and the read-out gather.

Each trace is an independent cascade (src/bin/psd.rs:174-182), so channels shard with no
data-path collective.  The only exchange is at read-out: every rank packs its raw per-stage
accumulators and 64-bit counters into the C ABI's fixed-size byte record (psdc_pack_readout) and the
records go to rank 0 in ONE gather (RCCL over xGMI on GPUs, gloo on CPU); rank 0 runs the host stitch
of PsdCascade::psd (src/psd.rs:479-543) per channel straight from the records (psdc_unpack_stitch).
Gathering the un-normalised accumulators keeps the result bit-identical to a single-GPU run, past 2^32
segments too (the record carries the 64-bit counts psd() normalises by).  The same three C calls serve a
Rust / C++ host with any transport.
"""
import numpy as np


def channel_shard(n_channels, world, rank):
    """Contiguous block of global channel ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_channels, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def pack_readout(bank, n_local, n, pkg=None, pad_to=None):
    """The rank's read-out record as a uint8 array of psdc_readout_bytes(n, rows) bytes, rows =
    max(n_local, pad_to): ranks must gather equally sized blocks, so pad to the largest shard.  A real
    PsdCascadeBank packs itself in the library (one flush, one copy per channel); anything that only
    duck-types the read-out accessors (a stand-in bank in the CPU tests) goes through psdc_pack_channel."""
    rows = max(n_local, pad_to or 0)
    if pkg is None:
        import sys
        pkg = sys.modules["stabilizer_stream_amd"]
    if hasattr(bank, "pack_readout") and n_local == bank.n_channels:
        # (always the library's own record, padded as bytes: the per-channel accessors below report the u32-saturated
        # count, which parts company with psd()'s 64-bit normalisation past 2^32 segments)
        return pkg.pack_pad(bank.pack_readout(), rows)
    chans = []
    for c in range(n_local):
        if hasattr(bank, "read_channel"):
            infos, sp = bank.read_channel(c)
        else:
            infos = [bank.stage_info(c, k) for k in range(bank.num_stages(c))]
            sp = (np.stack([bank.stage_spectrum(c, k) for k in range(len(infos))]) if infos
                  else np.zeros((0, n // 2 + 1), np.float32))
        chans.append(([i["count"] for i in infos], [i["avg"] for i in infos], [i["pending"] for i in infos], sp))
    window = getattr(bank, "window", None)
    return pkg.pack_record(n, chans, window if window is not None else pkg.Window.HANN, rows=rows)


def gather_readout(dist, rec, device=None, dst=0):
    """One gather of the byte records to rank `dst` (equal sizes on every rank).  Returns the list of
    records (uint8 numpy arrays, rank order) on dst, None elsewhere.  With `device` the payload makes one
    H2D copy, one RCCL gather and one D2H copy on `dst`."""
    import torch
    payload = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.uint8))
    if device is not None:
        payload = payload.to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return None
    return [p.cpu().numpy() for p in out]


def stitch_gathered(pkg, recs, counts_per_rank, opts=None):
    """Merged PSD + breaks for every global channel, in rank-major channel order."""
    opts = opts if opts is not None else pkg.MergeOpts()
    results = []
    for r, rec in enumerate(recs):
        for c in range(counts_per_rank[r]):
            results.append(pkg.unpack_stitch(rec, c, opts))
    return results
