"""FFT sizes that are not powers of two.  rustfft plans any length (`FftPlanner::new().plan_fft_forward(N)`,
src/psd.rs:418) and the reference only asks for N >= 2 (:138) and (N - overlap) % 8 == 0 (:246-247), so `PsdCascade::<1200>`
is a valid reference type.  Here such sizes (16 < N <= 8192) run the generic two-pass kernels with the N-point DFT in chirp-z
form on the power-of-two passes; the oracle evaluates the DFT by its definition in f64."""
import numpy as np
import pytest

from conftest import assert_psd_close, test_signal as make_signal
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


# (one size at least per chirp-z transform length M = 128 ... 16384: 48 | 80, 112 | 240 | 272, 400, 496 | 1008 | 1200 | 3056 | 6000, 8176)
@pytest.mark.parametrize("n,detrend", [(48, "none"), (80, "mean"), (112, "midpoint"), (240, "span"), (272, "none"), (400, "mean"),
                                       (496, "none"), (1008, "none"), (1200, "none"), (1200, "mean"),
                                       (3056, "none"), (6000, "none"), (8176, "none")])
def test_cascade_any_n_hann(pkg, ora, gpu_required, n, detrend):
    """PsdCascade::<N>::default() (Hann, overlap N/2: N a multiple of 16) for sizes with factors 3, 5, 7, 191 ...: host-fed in odd
    chunks and device-fed, every stage and the stitched PSD against the f64 oracle (the DFT by definition)."""
    import torch
    total = (300 if n <= 1200 else 40) * n + 8 * 11  # (the oracle's DFT by definition is O(N^2) per segment)
    x = make_signal(pkg, total, seed=n, tone=0.3, dc=0.2, f0=0.0123)
    g = pkg.PsdCascadeBank(n)
    g.set_detrend(pkg.Detrend[detrend.upper()])
    cut = (total // 3) | 1
    g.process(0, x[:cut])
    d = torch.from_numpy(x[cut:]).cuda()
    g.process_device(0, d.data_ptr(), total - cut)
    check_against_oracle(pkg, ora, g, [x], n, detrend=detrend, what=f"N={n} {detrend}", justify=n <= 1200)  # (the f32 DFT by definition too, where it is cheap)
    g.close()


def test_white_noise_bound_any_n(pkg, ora, gpu_required):
    """The reference's own acceptance test (src/psd.rs:634-643) at N = 1200: unit white noise reads PSD = 2 in every included bin of
    every stage within 10 / sqrt(count) -- the normalisation (gain, N/2, nenbw * power) and the merged bin ranges (2N/5 = 480) hold for
    a size that is not a power of two."""
    n = 1200
    x = pkg.noise_host(1 << 21, seed=0x7654321)
    c = pkg.PsdCascade(n)
    c.process(x)
    p, br = c.psd()
    assert [b.bins.stop for b in br][:-1] == [2 * n // 5] * (len(br) - 1) and br[-1].bins.stop == n // 2 + 1
    for b in br:
        if b.include:
            seg = p[b.start:b.start + len(b.bins)]
            assert np.all(np.abs(seg * 0.5 - 1.0) < 10.0 / np.sqrt(b.count)), b
    assert pkg.Break.frequencies(br)[-1] == 0.5
    c.close()


def test_single_stage_and_caller_window_any_n(pkg, ora, gpu_required):
    """Psd::new(fft, win) with N = 1000 (not a multiple of 16: Hann's overlap N/2 = 500 leaves (N - overlap) % 8 = 4, which the
    reference rejects at the first decimation, src/psd.rs:246-247) and a caller-built window whose overlap keeps the hop a
    multiple of 8."""
    n = 1000
    with pytest.raises(pkg.PsdError) as e:
        pkg.Psd(n)
    assert "overlap" in str(e.value)
    i = np.arange(n, dtype=np.float64)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * i / n)).astype(np.float32)
    m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
    win = pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), 504)  # hop 496
    x = make_signal(pkg, 120 * n, seed=9, tone=0.2)
    s = pkg.Psd.new(n, win)
    ref = ora.Psd(n, "f64", window=win.as_tuple())
    y, yr = s.process(x), ref.process(x)
    assert y.size == yr.size and np.max(np.abs(y - yr)) <= 4e-6 * np.max(np.abs(yr))
    assert s.count() == ref.count() and s.buf().size == ref.pending()
    r32 = ora.Psd(n, "f32", window=win.as_tuple())  # the yardstick of the widened comparison: the reference's arithmetic in f32
    r32.process(x)
    assert_psd_close(s.spectrum(), ref.spectrum(), "Psd<1000> Hamming", ref_f32=r32.spectrum(), real_bins=(0, n // 2))
    s.close()


def test_size_limits(pkg, gpu_required):
    for bad in (8, 15, 8200, 16400, 20000):  # below 16, or not a power of two above 8192, or above 16384
        with pytest.raises(pkg.PsdError):
            pkg.PsdCascade(bad)


@pytest.mark.parametrize("n,detrend,avg", [(32768, "none", None), (32768, "mean", (5, 200)), (65536, "span", None),
                                           (65536, "midpoint", (3, 40)), (131072, "none", None)])
def test_fft_sizes_above_an_lds_frame(pkg, ora, gpu_required, n, detrend, avg):
    """Powers of two 32768 ... 131072 (`FftPlanner::plan_fft_forward(N)` takes any N, src/psd.rs:417-418; the reference's own stack
    frames bound what it can run): the generic path with a four-step FFT through global memory (csrc/bigfft.hip).  White noise,
    host-fed in odd chunks then device-fed, a read-out in between: counters exactly, every stage against the f64 oracle (pure 1e-5
    where a plain-sum stage has >= 4 averages and no detrend nulls a bin)."""
    import torch
    total = 21 * n + 8 * 13
    x = pkg.noise_host(total, seed=1300 + n // 1024)
    g = pkg.PsdCascadeBank(n)
    g.set_detrend(pkg.Detrend[detrend.upper()])
    av = pkg.AvgOpts(*avg) if avg else None
    if av:
        g.set_avg(av)
    cut = (total // 3) | 1
    g.process(0, x[:cut])
    assert g.stage_info(0, 0)["count"] == min((cut - n) // (n // 2) + 1, (min(av.count, av.limit) + 1) if av else 1 << 40)
    d = torch.from_numpy(x[cut:]).cuda()
    g.process_device(0, d.data_ptr(), total - cut)
    check_against_oracle(pkg, ora, g, [x], n, detrend=detrend, avg=av, what=f"N={n} {detrend} avg={avg}",
                         pure_min_count=4 if (av is None and detrend == "none") else None)
    g.close()
    with pytest.raises(pkg.PsdError):
        pkg.PsdCascadeBank(262144)  # beyond the largest size


def test_big_fft_batches_and_chunk_splits(pkg, ora, gpu_required, monkeypatch):
    """The four-step path shares its launches between every job of a round and splits a job that does not fit a chunk at a multiple
    of its group size (csrc/bigfft.hip `launch_welch_big`): two channels of ~50 segment pairs each, Mean detrend (the per-segment
    means ride in the chunk too), once with the chunk the scratch allows and once capped at 16 pairs (`PSDC_DBG_BIGFFT_CHUNK`: jobs
    split four ways, chunks holding pieces of two jobs).  Both against the f64 oracle, and bit-identical to each other: the order of
    the additions into a job's row does not depend on what else was in the batch."""
    n = 32768
    total = 101 * (n // 2) + 77
    xs = [pkg.noise_host(total, seed=4100 + c) for c in range(2)]
    outs = []
    for limit in (None, "16"):
        if limit:
            monkeypatch.setenv("PSDC_DBG_BIGFFT_CHUNK", limit)
        else:
            monkeypatch.delenv("PSDC_DBG_BIGFFT_CHUNK", raising=False)
        g = pkg.PsdCascadeBank(n, n_channels=2)
        g.set_detrend(pkg.Detrend.MEAN)
        for c in range(2):
            g.process(c, xs[c])
        for c in range(2):
            check_against_oracle(pkg, ora, g, [xs[c]], n, detrend="mean", channel=c, what=f"N={n} channel {c} of two, chunk limit {limit}")
        outs.append([g.stage_spectrum(c, s) for c in range(2) for s in range(g.num_stages(c))])
        g.close()
    assert len(outs[0]) == len(outs[1])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
