"""Channel sharding across the GPUs of one node and the read-out gather.

Each trace is an independent cascade (src/bin/psd.rs:174-182), so channels shard with no
data-path collective.  The only exchange is at read-out: every rank's raw per-stage
accumulators and counters go to rank 0 in ONE gather (RCCL over xGMI on GPUs, gloo on CPU),
and rank 0 runs the host stitch of PsdCascade::psd (src/psd.rs:479-543) per channel.
Gathering the un-normalised accumulators keeps the result bit-identical to a single-GPU run.
"""
import numpy as np

KMAX = 12  # stage slots per channel in the gather payload (8^12 * N samples: unreachable)


def channel_shard(n_channels, world, rank):
    """Contiguous block of global channel ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_channels, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def pack_readout(bank, n_local, n, torch=None, pad_to=None):
    """Raw spectra [rows, KMAX, n/2+1] f32 and (count, avg, pending, valid) [rows, KMAX, 4] i64 as numpy
    arrays, rows = max(n_local, pad_to) (ranks must gather equally shaped blocks: pad to the largest
    shard).  Plain numpy on purpose: the read-out sits inside timed loops, and small CPU tensor
    ops in torch were seen to stall for tens of ms now and then (thread-pool wake-ups) on the many-core
    GPU hosts.  `torch` is accepted for compatibility and not used."""
    bins = n // 2 + 1
    rows = max(n_local, pad_to or 0)
    spec = np.zeros((rows, KMAX, bins), dtype=np.float32)
    meta = np.zeros((rows, KMAX, 4), dtype=np.int64)
    for c in range(n_local):
        if hasattr(bank, "read_channel"):  # one flush + one copy per channel
            infos, sp = bank.read_channel(c)
        else:
            infos = [bank.stage_info(c, k) for k in range(bank.num_stages(c))]
            sp = np.stack([bank.stage_spectrum(c, k) for k in range(len(infos))]) if infos else np.zeros((0, bins))
        assert len(infos) <= KMAX
        if len(infos):
            spec[c, :len(infos)] = sp
        for k, info in enumerate(infos):
            meta[c, k] = (info["count"], info["avg"], info["pending"], 1)
    return spec, meta


def gather_readout(dist, spec, meta, device=None, dst=0):
    """One gather of (spec, meta) to rank `dst`.  All ranks must hold equally shaped arrays
    (numpy or CPU tensors).  Returns lists of numpy arrays on dst, (None, None) elsewhere.
    The payload is packed on the host (one H2D copy, one collective, one D2H copy on `dst`)."""
    import torch
    spec = np.ascontiguousarray(np.asarray(spec), dtype=np.float32)
    meta = np.ascontiguousarray(np.asarray(meta), dtype=np.int64)
    rows = spec.shape[0]
    nb = spec.shape[1] * spec.shape[2]
    nm = meta.shape[1] * meta.shape[2]
    # a single f32 payload: meta is carried as two exact halves (values < 2^48 split into 24-bit words)
    pay = np.empty((rows, nb + 2 * nm), dtype=np.float32)
    pay[:, :nb] = spec.reshape(rows, nb)
    m = meta.reshape(rows, nm)
    pay[:, nb:nb + nm] = (m & 0xFFFFFF).astype(np.float32)
    pay[:, nb + nm:] = (m >> 24).astype(np.float32)
    payload = torch.from_numpy(pay)
    if device is not None:
        payload = payload.to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return None, None
    got = torch.stack(out).cpu().numpy() if device is not None else [p.numpy() for p in out]
    specs, metas = [], []
    for p in got:
        specs.append(p[:, :nb].reshape(spec.shape))
        lo_ = p[:, nb:nb + nm].astype(np.int64)
        hi_ = p[:, nb + nm:nb + 2 * nm].astype(np.int64)
        metas.append(((hi_ << 24) | lo_).reshape(meta.shape))
    return specs, metas


def stitch_gathered(pkg, n, specs, metas, counts_per_rank, opts=None, window=None):
    """Merged PSD + breaks for every global channel, in rank-major channel order."""
    opts = opts if opts is not None else pkg.MergeOpts()
    window = window if window is not None else pkg.Window.HANN
    results = []
    for r, (spec, meta) in enumerate(zip(specs, metas)):
        spec, meta = np.asarray(spec), np.asarray(meta)
        for c in range(counts_per_rank[r]):
            ns = int(meta[c, :, 3].sum())
            counts = [int(v) for v in meta[c, :ns, 0]]
            avgs = [int(v) for v in meta[c, :ns, 1]]
            pend = [int(v) for v in meta[c, :ns, 2]]
            results.append(pkg.stitch(n, counts, avgs, pend, np.ascontiguousarray(spec[c, :ns]), opts, window))
    return results
