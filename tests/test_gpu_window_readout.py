"""GPU parity of the round-3 boundary additions:

* `Psd::new(fft, win)` / a cascade over a caller-built `Window<N>` (src/psd.rs:12-20 pub struct with pub fields,
  :137-152 constructor) through psdc_stage_create_window / psdc_create_window, against the oracle given the SAME table;
* the packed read-out (psdc_pack_readout -> psdc_unpack_stitch), which a multi-GPU host gathers with any transport;
* the device selector (PSDC_DEVICE_DEFAULT / $PSDC_DEVICE, one handle per device in one process).
"""
import os

import numpy as np
import pytest

from conftest import assert_psd_close, test_signal as make_signal
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def hamming(pkg, n, overlap):
    """A Window<N> the library has no kind for: Hamming weights in f32, constants as src/psd.rs:12-20 defines them
    (power = mean(w)^2, nenbw = mean(w^2) / mean(w)^2), a caller-chosen overlap."""
    i = np.arange(n, dtype=np.float64)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * i / n)).astype(np.float32)
    m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
    return pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), overlap)


@pytest.mark.parametrize("n,overlap_of", [(64, lambda n: n // 2), (512, lambda n: n // 2), (512, lambda n: 3 * n // 4),
                                          (1024, lambda n: n - 8), (2048, lambda n: 0), (4096, lambda n: n // 2 + 8)])
def test_stage_with_caller_window(pkg, ora, gpu_required, n, overlap_of):
    """The crate's own single-stage test body (src/psd.rs:614-632) with `Window` built by the caller: Psd::new(fft,
    Arc::new(win)), process(&x, &mut y), spectrum / gain / count / buf against the oracle with the same table."""
    win = hamming(pkg, n, overlap_of(n))
    x = make_signal(pkg, 70 * n + 8 * 13, seed=50 + n, tone=0.3, dc=0.1)
    s = pkg.Psd.new(n, win)  # fft.len() == N asserted like src/psd.rs:139
    ref = ora.Psd(n, "f64", window=win.as_tuple())
    r32 = ora.Psd(n, "f32", window=win.as_tuple())
    ys, yrs = [], []
    for c in (x[:5], x[5:3 * n + 1], x[3 * n + 1:]):
        y, yr = s.process(c), ref.process(c)
        r32.process(c)
        assert y.size == yr.size
        ys.append(y.copy())
        yrs.append(yr)
    y, yr = np.concatenate(ys), np.concatenate(yrs)
    assert np.max(np.abs(y - yr)) <= 4e-6 * np.max(np.abs(yr))
    assert s.count() == ref.count() and s.buf().size == ref.pending()
    assert s.gain() == pytest.approx(ref.gain(), rel=1e-6)
    assert_psd_close(s.spectrum(), ref.spectrum(), f"Psd<{n}> Hamming overlap {win.overlap}", ref_f32=r32.spectrum())
    c = s.clone()  # #[derive(Clone)] carries the caller's window
    assert np.array_equal(c.spectrum(), s.spectrum()) and c.count() == s.count()
    c.close()
    s.close()
    with pytest.raises(pkg.PsdError):  # assert_eq!(N, fft.len()) src/psd.rs:139
        pkg.Psd.new(n // 2, win)


@pytest.mark.parametrize("n,detrend", [(256, "none"), (1024, "mean"), (4096, "span")])
def test_cascade_with_caller_window(pkg, ora, gpu_required, n, detrend):
    """A whole cascade over a caller-built window (generic two-pass kernels, hop = n/4), host-fed in odd chunks and
    device-fed in place, every stage and the stitched PSD against the oracle with the same table."""
    import torch
    win = hamming(pkg, n, 3 * n // 4)
    total = 700 * n + 8 * 5
    x = make_signal(pkg, total, seed=60 + n, tone=0.2, dc=0.5)
    g = pkg.PsdCascadeBank(n, window=win)
    kind, got = g.window_get()
    assert kind == pkg.Window.CUSTOM and np.array_equal(got.win, win.win) and got.overlap == win.overlap
    g.set_detrend(pkg.Detrend[detrend.upper()])
    cut = (total // 5) | 1
    g.process(0, x[:cut])
    d = torch.from_numpy(x[cut:]).cuda()
    g.process_device(0, d.data_ptr(), total - cut)
    check_against_oracle(pkg, ora, g, [x], n, detrend=detrend, window=win.as_tuple(), what=f"Hamming cascade N={n} {detrend}")
    c = g.clone()
    assert np.array_equal(c.psd(0)[0], g.psd(0)[0])
    c.close()
    g.close()


def blackman(pkg, n):
    """Another caller-built Window<N> with the usual Welch overlap of N/2."""
    i = np.arange(n, dtype=np.float64)
    w = (0.42 - 0.5 * np.cos(2 * np.pi * i / n) + 0.08 * np.cos(4 * np.pi * i / n)).astype(np.float32)
    m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
    return pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), n // 2)


@pytest.mark.parametrize("n,detrend,make", [(256, "none", "hamming"), (512, "mean", "blackman"), (1024, "none", "hamming"),
                                            (1024, "span", "blackman"), (2048, "midpoint", "hamming"), (4096, "none", "blackman"),
                                            (4096, "mean", "hamming"), (8192, "none", "hamming"), (16384, "none", "blackman")])
def test_caller_window_with_half_overlap_runs_the_fused_kernels(pkg, ora, gpu_required, n, detrend, make):
    """A caller-built Window<N> whose overlap is N/2 (src/psd.rs:12-20: any table, any constants) takes the same single-pass
    kernels as Window::hann() -- they read the table and assume only the hop: host-fed in odd chunks, device-fed in place,
    finite averaging on a second handle, every stage and the stitched PSD against the oracle given the SAME table."""
    import torch
    win = hamming(pkg, n, n // 2) if make == "hamming" else blackman(pkg, n)
    total = 1400 * n + 8 * 3
    x = make_signal(pkg, total, seed=70 + n, tone=0.25, dc=0.3)
    d = torch.from_numpy(x).cuda()
    for avg in (None, (30, 2000)):
        g = pkg.PsdCascadeBank(n, window=win)
        assert g.window_get()[0] == pkg.Window.CUSTOM
        g.set_detrend(pkg.Detrend[detrend.upper()])
        a = pkg.AvgOpts(*avg) if avg else None
        if a:
            g.set_avg(a)
        cut = 4 * ((total // 28) | 1)  # (16-byte aligned: the device span is read in place)
        g.process(0, x[:5])
        g.process(0, x[5:cut])
        g.process_device(0, d.data_ptr() + 4 * cut, total - cut)
        check_against_oracle(pkg, ora, g, [x], n, detrend=detrend, avg=a, window=win.as_tuple(),
                             what=f"{make} overlap N/2, N={n} {detrend} avg={avg}")
        g.close()


def test_caller_window_frames_in_place(pkg, ora, gpu_required):
    """AdcDac frames read in place under a caller-built window (overlap N/2): same kernels, the caller's table."""
    import torch
    from test_gpu_frames_inplace import make_frames
    n, batches = 4096, 22
    win = hamming(pkg, n, n // 2)
    nframes = (200 * n) // (batches * 8) + 11
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=77)
    dd = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4, window=win)
    half = nframes // 2
    assert g.process_adcdac_frames_device(dd.data_ptr(), fs, half) == half
    assert g.process_adcdac_frames_device(dd.data_ptr() + half * fs, fs, nframes - half) == nframes - half
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, channel=c, window=win.as_tuple(),
                             what=f"frames in place, Hamming table, trace {c}")
    g.close()


@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_library_windows_passed_as_tables(pkg, ora, gpu_required, n):
    """Window::hann() / Window::rectangular() handed over as tables are recognised (Hann keeps the fused kernels)
    and give bit-identical results to the by-kind constructors."""
    x = pkg.noise_host(90 * n, seed=70 + n)
    for kind, table in ((pkg.Window.HANN, pkg.WindowTable.hann(n)), (pkg.Window.RECTANGULAR, pkg.WindowTable.rectangular(n))):
        w_ora, p, e, ov = ora.window(n, "hann" if kind == pkg.Window.HANN else "rect", "f32")
        assert np.array_equal(table.win, w_ora) and (table.power, table.nenbw, table.overlap) == (p, e, ov)
        a, b = pkg.PsdCascadeBank(n, window=kind), pkg.PsdCascadeBank(n, window=table)
        assert b.window_get()[0] == kind
        a.process(0, x)
        b.process(0, x)
        assert np.array_equal(a.psd(0)[0], b.psd(0)[0])
        a.close()
        b.close()
    # one weight off by an ulp: no longer Hann, generic kernels, same spectrum to rounding
    t = pkg.WindowTable.hann(n)
    w = t.win.copy()
    w[n // 3] = np.nextafter(w[n // 3], np.float32(2.0))
    c = pkg.PsdCascadeBank(n, window=pkg.WindowTable(w, t.power, t.nenbw, t.overlap))
    assert c.window_get()[0] == pkg.Window.CUSTOM
    c.process(0, x)
    check_against_oracle(pkg, ora, c, [x], n, window=(w, t.power, t.nenbw, t.overlap), what=f"almost-Hann N={n}")
    c.close()


def test_create_window_argument_errors(pkg, gpu_required):
    n = 256
    t = pkg.WindowTable.hann(n)
    for bad_overlap in (n, n + 8, n - 4, 3):  # `N - overlap` underflows / (N - overlap) % 8 != 0 (src/psd.rs:246-247)
        with pytest.raises(pkg.PsdError) as e:
            pkg.PsdCascadeBank(n, window=pkg.WindowTable(t.win, t.power, t.nenbw, bad_overlap))
        assert e.value.code in (pkg.ERR_ARG, pkg.ERR_DEVICE) and "overlap" in str(e.value)
    with pytest.raises(pkg.PsdError):
        pkg.PsdCascadeBank(n, window=pkg.WindowTable(t.win[:-1], t.power, t.nenbw, t.overlap))
    with pytest.raises(pkg.PsdError):
        pkg.Psd(n, pkg.WindowTable(t.win, float("nan"), t.nenbw, t.overlap))


def test_packed_readout_equals_psd(pkg, ora, gpu_required):
    """psdc_pack_readout -> (any transport) -> psdc_unpack_stitch == psdc_psd on the handle, channel by channel and for
    every MergeOpts; the record has the advertised fixed size, and records of two handles concatenate."""
    n, nch = 512, 3
    g = pkg.PsdCascadeBank(n, nch)
    g.set_avg(pkg.AvgOpts(1000, 100000))
    xs = [pkg.noise_host(300000 + 4096 * c, seed=80 + c) for c in range(nch)]
    for c in range(nch):
        g.process(c, xs[c])
    rec = g.pack_readout()
    assert rec.size == pkg.readout_bytes(n, nch)
    for c in range(nch):
        assert pkg.unpack_info(rec, c) == (n, nch, g.num_stages(c))
        for opts in (pkg.MergeOpts(), pkg.MergeOpts(True, 0, True), pkg.MergeOpts(False, 3, False)):
            p, br = g.psd(c, opts)
            q, bq = pkg.unpack_stitch(rec.tobytes(), c, opts)  # (a copy: the record is plain bytes)
            assert np.array_equal(p, q, equal_nan=True) and br == bq
    with pytest.raises(pkg.PsdError):
        pkg.unpack_stitch(rec[:-8], 0)
    with pytest.raises(pkg.PsdError):
        pkg.unpack_stitch(rec, nch)
    g.close()


def test_device_selector(pkg, gpu_required):
    """PSDC_DEVICE_DEFAULT (-1) takes the index from $PSDC_DEVICE: a shim whose constructor has no device argument
    (PsdCascade::<N>::default()) is placed from outside, one process per GPU."""
    import torch
    x = pkg.noise_host(100000, seed=5)
    old = os.environ.pop("PSDC_DEVICE", None)
    try:
        a = pkg.PsdCascade(1024, device=-1)  # unset -> 0
        a.process(x)
        os.environ["PSDC_DEVICE"] = "0"
        b = pkg.PsdCascade(1024, device=-1)
        b.process(x)
        assert np.array_equal(a.psd()[0], b.psd()[0])
        os.environ["PSDC_DEVICE"] = str(torch.cuda.device_count())  # one past the last device
        with pytest.raises(pkg.PsdError) as e:
            pkg.PsdCascade(1024, device=-1)
        assert "device index out of range" in str(e.value)
        os.environ["PSDC_DEVICE"] = "zero"
        with pytest.raises(pkg.PsdError):
            pkg.PsdCascade(1024, device=-1)
        a.close()
        b.close()
    finally:
        os.environ.pop("PSDC_DEVICE", None)
        if old is not None:
            os.environ["PSDC_DEVICE"] = old


@pytest.mark.parametrize("second", [0, 1])
def test_one_process_two_handles_gather_without_a_collective(pkg, ora, gpu_required, second):
    """The second multi-GPU mode of INTEGRATION.md: ONE process, one handle per device, read-outs packed and stitched
    with no collective at all.  `second` is the device of the second handle: 1 needs a second visible GPU (skipped on a
    one-GPU box), 0 shares the device -- the code path (per-handle device scope, records concatenated) is the same."""
    import torch
    if second >= torch.cuda.device_count():
        pytest.skip(f"device {second} not visible")
    n = 1024
    xs = [pkg.noise_host(250000, seed=90 + c) for c in range(4)]
    h0, h1 = pkg.PsdCascadeBank(n, 2, device=0), pkg.PsdCascadeBank(n, 2, device=second)
    for c in range(2):
        h0.process(c, xs[c])
        h1.process(c, xs[2 + c])
    recs = [h0.pack_readout(), h1.pack_readout()]
    cur = torch.cuda.current_device()
    for g, (r, c) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        p, br = pkg.unpack_stitch(recs[r], c)
        one, o32 = ora.PsdCascade(n, "f64"), ora.PsdCascade(n, "f32")
        one.process(xs[g])
        o32.process(xs[g])
        pr, brr, _ = one.psd()
        assert [b.count for b in br] == [b["count"] for b in brr]
        assert_psd_close(p, pr, f"global channel {g}", ref_f32=o32.psd()[0])
    assert torch.cuda.current_device() == cur  # every ABI call restores the caller's device
    h0.close()
    h1.close()
