"""Pin the CPU oracle against every known answer / constant the reference holds for this path
(tests/golden/reference_known_answers.json, each entry cites its reference line)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_exact_n4(ora, prec):
    g = GOLD["exact_n4_hann"]
    # window + FFT + gain directly: N=4 cannot go through process() (src/psd.rs:246-247)
    c = ora.detrend_apply(np.array(g["x"]), "none", "hann", prec)
    X = ora.fft_forward(c, prec)
    w, power, nenbw, ov = ora.window(g["n"], "hann", prec)
    gain = (g["n"] // 2 * 1) * nenbw * power  # src/psd.rs:282, count = 1
    p = np.abs(X[: g["n"] // 2 + 1]) ** 2 / gain
    assert np.all(np.abs(p - np.array(g["psd"])) < (g["tolerance"] if prec == "f64" else 1e-6))
    f = np.arange(g["n"] // 2 + 1) / g["n"]
    assert np.allclose(f, g["frequencies"], atol=g["tolerance"])


@pytest.mark.parametrize("n", [16, 512, 1024, 4096, 16384])
def test_window_constants(ora, n):
    w, power, nenbw, ov = ora.window(n, "hann", "f64")
    h = GOLD["hann_constants"]
    assert (power, nenbw, ov) == (h["power"], h["nenbw"], int(h["overlap_over_n"] * n))
    # power = mean(w)^2, nenbw*power = mean(w^2): exact for the periodic Hann of any N
    assert abs(np.mean(w) ** 2 - power) < 1e-12 and abs(np.mean(w * w) - nenbw * power) < 1e-12
    assert w[0] == 0.0 and w[0] != w[-1]  # period N, not N-1 (src/psd.rs:36-41)
    w32, *_ = ora.window(n, "hann", "f32")
    assert np.max(np.abs(w32 - w)) < 1e-6  # f32 argument rounding of df*i (src/psd.rs:44-47)
    r, rp, rn, ro = ora.window(n, "rect", "f64")
    rc = GOLD["rectangular_constants"]
    assert np.all(r == 1.0) and (rp, rn, ro) == (rc["power"], rc["nenbw"], rc["overlap"])


@pytest.mark.parametrize("n", [4, 8, 64, 1024, 16384])
def test_fft_is_unnormalised_forward_dft(ora, n):
    rng = np.random.default_rng(n)
    c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = np.fft.fft(c)
    assert np.max(np.abs(ora.fft_forward(c, "f64") - ref)) < 1e-9 * np.sqrt(n)
    c32 = c.astype(np.complex64)
    assert np.max(np.abs(ora.fft_forward(c32, "f32") - np.fft.fft(c32.astype(np.complex128)))) < 3e-6 * n ** 0.5 * np.log2(n)


def test_detrend_variants(ora):
    rng = np.random.default_rng(2)
    n = 64
    x = rng.standard_normal(n) + 5.0
    w, *_ = ora.window(n, "hann", "f64")
    assert np.allclose(ora.detrend_apply(x, "none", "hann", "f64").real, x * w, atol=1e-12)
    assert np.allclose(ora.detrend_apply(x, "midpoint", "hann", "f64").real, (x - x[n // 2]) * w, atol=1e-12)
    assert np.allclose(ora.detrend_apply(x, "mean", "hann", "f64").real, (x - x.mean()) * w, atol=1e-12)
    ramp = x[0] + np.arange(n) * (x[-1] - x[0]) / (n - 1)
    assert np.allclose(ora.detrend_apply(x, "span", "hann", "f64").real, (x - ramp) * w, atol=1e-10)
    assert np.all(ora.detrend_apply(x, "mean", "hann", "f64").imag == 0)


def test_hbf_pins(ora):
    """What the reference pins about the idsp decimator."""
    d = ora.hbf_response_length(3)
    assert d == 35
    # rate: exactly one output per 8 inputs; DC gain 8 (x2 per half-band stage)
    y = ora.hbf_dec8(np.ones(8 * 200), "f64")
    assert y.size == 200 and abs(y[-1] - 8.0) < 8 * 2e-4  # < 0.001 dB ripple
    # impulse response settles within d outputs of the last non-zero input
    last = []
    for phase in range(8):
        imp = np.zeros(8 * 100)
        imp[phase] = 1.0
        h = ora.hbf_dec8(imp, "f64")
        last.append(int(np.nonzero(h)[0][-1]))
        assert abs(h.sum() - 1.0) < 1e-4  # each polyphase branch has DC gain 1 (total 8)
    assert max(last) == d
    # pass band 0.4 of the output rate flat, alias band clean (~ -98 dB)
    n = 8 * 4096
    t = np.arange(n)
    for f_out, lo, hi in [(0.05, 7.99, 8.01), (0.2, 7.99, 8.01), (0.39, 7.99, 8.01)]:
        y = ora.hbf_dec8(np.sin(2 * np.pi * f_out / 8 * t), "f64")[200:]
        amp = np.sqrt(2 * np.mean(y * y))
        assert lo < amp < hi, (f_out, amp)
    for f_in in [0.6 / 8, 1.0 / 8 + 0.3 / 8, 0.45]:  # fold into the pass band after /8
        y = ora.hbf_dec8(np.sin(2 * np.pi * f_in * t), "f64")[200:]
        assert np.sqrt(2 * np.mean(y * y)) < 8 * 10 ** (-95 / 20), f_in


def test_statistical_reference_test(ora):
    """src/psd.rs:599-644 verbatim, on the oracle (f32 mirror)."""
    g = GOLD["statistical_test"]
    assert GOLD["hbf_passband"]["value"] == 0.4
    rng = np.random.default_rng()
    x = ((rng.random(g["samples"], dtype=np.float32) - np.float32(0.5)) * np.float32(np.sqrt(12))).astype(np.float32)
    n, k = g["n"], g["sigma_factor"]
    s = ora.Psd(n, "f32")
    y = s.process(x)
    assert y.size == (x.size >> 3) - ora.hbf_response_length(3)
    p = s.spectrum() / np.float32(s.gain())
    assert np.all(np.abs(p * 0.5 - 1.0) < k / np.sqrt(s.count()))
    d = ora.PsdCascade(n, "f32")
    d.process(x)
    p, br, _ = d.psd()
    for b in br:
        seg = p[b["start"]:b["start"] + b["bins_end"] - b["bins_start"]] if b["include"] else p[:0]
        assert np.all(np.abs(seg * 0.5 - 1.0) < k / np.sqrt(max(b["count"], 1)))


@pytest.mark.parametrize("n", [512, 1024, 4096, 16384])
def test_merged_bin_ranges(ora, n):
    """src/psd.rs:490-501: stitch ranges for lowest / middle / top stages."""
    g = GOLD["merged_bins"][str(n)]
    c = ora.PsdCascade(n, "f32")
    total = 80 * n  # reaches 3 stages with count >= 1
    c.process(np.random.default_rng(1).standard_normal(total).astype(np.float32))
    p, br, _ = c.psd()
    inc = [b for b in br if b["include"]]
    assert len(inc) >= 3
    assert [inc[0]["bins_start"], inc[0]["bins_end"]] == g["lowest"]
    assert [inc[1]["bins_start"], inc[1]["bins_end"]] == g["middle"]
    assert [inc[-1]["bins_start"], inc[-1]["bins_end"]] == g["top"]
    f = c.frequencies(_)
    assert f[0] == 0.0 and f[-1] == 0.5


def test_chunk_invariance_bit_exact(ora):
    """Any split of the stream across process() calls gives bit-identical state (src/psd.rs:196-208)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(50000).astype(np.float32)
    a = ora.PsdCascade(64, "f32")
    a.process(x)
    b = ora.PsdCascade(64, "f32")
    i = 0
    while i < x.size:
        m = int(rng.integers(0, 700))
        b.process(x[i:i + m])
        i += m
    assert a.num_stages == b.num_stages
    for k in range(a.num_stages):
        assert a.stage_info(k) == b.stage_info(k)
        assert np.array_equal(a.stage_spectrum(k), b.stage_spectrum(k))
        assert np.array_equal(a.stage_buf(k), b.stage_buf(k))


def test_f32_mirror_tracks_f64_truth(ora):
    rng = np.random.default_rng(4)
    x = rng.standard_normal(200000).astype(np.float32)
    a, b = ora.PsdCascade(1024, "f32"), ora.PsdCascade(1024, "f64")
    for d in ("none", "mean"):
        a.set_detrend(d)
        b.set_detrend(d)
        a.process(x)
        b.process(x)
    for k in range(b.num_stages):
        if b.stage_info(k)["count"]:
            r = b.stage_spectrum(k)
            assert np.max(np.abs(a.stage_spectrum(k) - r) / r) < 2e-5


def test_var_known_answer(ora):
    g = GOLD["var_basic"]
    assert abs(ora.var_eval(g["p"], g["f"], g["tau"]) - g["value"]) < g["tolerance"]


def test_adcdac_decode(ora):
    g = GOLD["adcdac"]
    b = 3
    raw = np.arange(4 * b * 8, dtype=np.int16).reshape(b, 4, 8) * 100 - 5000
    frame = bytes(g["magic"]) + bytes([g["format_id"], b]) + (77).to_bytes(4, "little") + raw.astype("<i2").tobytes()
    st, seq, nb, tr = ora.adcdac_decode(frame)
    assert (st, seq, nb) == (0, 77, b)
    # f32 constant arithmetic of src/de/data.rs:28-35 (DAC and ADC constants are asserted equal there)
    lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)
    assert lsb == np.float32(5.0) / np.float32(2.0) * np.float32(4.096) / np.float32(32768)
    assert abs(float(lsb) - g["volt_per_lsb"]) < 1e-10
    assert np.array_equal(tr[0], raw[:, 0, :].ravel().astype(np.float32) * lsb)
    assert np.array_equal(tr[1], raw[:, 1, :].ravel().astype(np.float32) * lsb)
    dac = (raw[:, 2, :].ravel().astype(np.int32) + 32768 + 32768) % 65536 - 32768  # wrapping_add(i16::MIN)
    assert np.array_equal(tr[2], dac.astype(np.float32) * lsb)
    assert ora.adcdac_decode(b"\x00\x05" + frame[2:])[0] == -1   # InvalidHeader
    assert ora.adcdac_decode(frame[:2] + b"\x09" + frame[3:])[0] == -2  # UnknownFormat
    assert ora.adcdac_decode(frame[:-3])[0] == -3  # PayloadSize
    assert ora.adcdac_decode(frame[:3] + b"\x04" + frame[4:])[0] == -4  # batches mismatch panics


def test_short_first_call_then_full_chunk(ora):
    """A feed the reference cannot take: its cascade hands every stage a [f32; N] output buffer
    (src/psd.rs:457-458) although PsdStage::process may emit x.len()/8 + N/8 items -- after a short
    first call (< N samples buffered, no segment yet) a full 8N chunk completes 16 segments INCLUDING
    the stream's first, N + N/16 - 35 items, and `&mut y[n..][..xb.len()]` (:253) panics.  The
    oracle sizes its buffers as the contract asks, flags the condition and keeps the stream
    semantics (same result as an aligned feed), which is what the GPU path is held to."""
    n = 1024
    rng = np.random.default_rng(5)
    x = rng.standard_normal(900 + 8 * n + 3000).astype(np.float32)
    a = ora.PsdCascade(n, "f64")
    a.process(x[:900])
    assert not a.ref_would_panic
    a.process(x[900:900 + 8 * n])
    assert a.ref_would_panic
    a.process(x[900 + 8 * n:])
    b = ora.PsdCascade(n, "f64")
    for i in range(0, x.size, 512):
        b.process(x[i:i + 512])
    assert not b.ref_would_panic
    assert a.num_stages == b.num_stages
    for k in range(a.num_stages):
        assert a.stage_info(k) == b.stage_info(k)
        np.testing.assert_array_equal(a.stage_spectrum(k), b.stage_spectrum(k))
        np.testing.assert_array_equal(a.stage_buf(k), b.stage_buf(k))


def test_config1_counts(pkg, ora):
    """BASELINE config 1: 1 channel, N = 1024, 2^20 samples on the CPU path -- four stages
    with spectra, counts 2047 / 254 / 30 / 2, and a fifth stage holding 157 pending samples with no segment yet
    (excluded by min_count = 1).  The closed-form planner of the library must say the same."""
    n, total = 1024, 1 << 20
    x = pkg.noise_host(total, 0x7654321)
    o = ora.PsdCascade(n, "f32")
    for i in range(0, total, 1 << 16):  # process() calls of 65536 samples like src/psd.rs:554-559
        o.process(x[i:i + (1 << 16)])
    assert o.num_stages == 5
    assert [o.stage_info(k)["count"] for k in range(5)] == [2047, 254, 30, 2, 0]
    assert o.stage_info(4)["pending"] == 157
    plan = pkg.plan_counts(n, total)
    assert [(segs, pend) for _, segs, pend in plan] == [(o.stage_info(k)["count"], o.stage_info(k)["pending"]) for k in range(5)]
    p, br, cbr = o.psd()
    assert [b["include"] for b in br] == [0, 1, 1, 1, 1]  # lowest rate first: the fifth stage is not included
    assert p.size == 409 + 357 * 2 + 461 == 1584          # merged length for K = 4
    f = o.frequencies(cbr)
    assert f[0] == 0.0 and f[-1] == 0.5 and np.all(np.diff(f) > 0)
    for b in br:  # the reference's own bound (src/psd.rs:634-643)
        if b["include"]:
            seg = p[b["start"]:b["start"] + b["bins_end"] - b["bins_start"]]
            assert np.all(np.abs(seg * 0.5 - 1.0) < 10.0 / np.sqrt(b["count"]))


def test_trace_plot_matches_the_restatement(pkg, ora):
    """Trace::plot / Trapezoidal (src/bin/psd.rs:98-157): the library's host function against the oracle's
    restatement (same f32 operations: bit-equal), and the closed forms a flat PSD gives."""
    n = 1024
    x = pkg.noise_host(1 << 18, 7)
    o = ora.PsdCascade(n, "f32")
    o.process(x)
    p, br, cbr = o.psd()
    f = o.frequencies(cbr)
    for fs, integ, lo, hi in [(1.0, False, 0.0, float("inf")), (1e6, True, 10.0, 1e5), (781250.0, True, 0.0, 100e3),
                              (2.0, False, 0.3, 0.31)]:
        r_lib, xy_lib = pkg.trace_plot(p, f, fs, integ, lo, hi)
        r_ora, xy_ora = ora.trace_plot(p, f, fs, integ, lo, hi)
        assert r_lib == r_ora and np.array_equal(xy_lib, xy_ora)
        assert xy_lib.shape == (f.size - 1, 2)  # f = 0 is not a normal float: no plot point (:141)
    # unit white noise: PSD = 2 over f in [0, 0.5] integrates to the variance 1
    rms, xy = pkg.trace_plot(p, f, 1.0, True)
    assert abs(rms - 1.0) < 0.02
    assert abs(xy[-1, 1] - rms) < 1e-6 and xy[-1, 0] == pytest.approx(np.log10(0.5))
    # flat PSD 3.0 on an irregular grid: the trapezoid from (0, 0) adds (3 + 0)/2 * f0 for the first interval
    ff = np.array([0.01, 0.02, 0.05, 0.1, 0.5], dtype=np.float32)
    rms, _ = pkg.trace_plot(np.full(5, 3.0, np.float32), ff, 1.0, False)
    assert rms == pytest.approx(np.sqrt(1.5 * 0.01 + 3.0 * 0.49), rel=1e-6)
    rms, _ = pkg.trace_plot(np.full(5, 3.0, np.float32), ff, 100.0, False, 2.0, 10.0)  # bins at 2, 5, 10 Hz
    assert rms == pytest.approx(np.sqrt(3.0 * (0.1 - 0.01)), rel=1e-6)
    assert pkg.trace_plot(np.zeros(0, np.float32), np.zeros(0, np.float32))[0] == 0.0


def test_stream_test_tail_on_the_oracle(pkg, ora):
    """The tail of src/bin/stream_test.rs (:57-71): psd(&MergeOpts::default()), then the FDEV sweep
    Var{dc_cut: 1, clip: 1.0}.eval(psd, f, tau).sqrt() for tau = 1, 2, 4, ... <= effective_fft_size/2 --
    the library's host helpers on the oracle's merged PSD must reproduce the restatement bit for bit."""
    n = 512  # the FFT size both reference binaries hard-code
    o = ora.PsdCascade(n, "f32")
    o.set_detrend("midpoint")  # stream_test.rs:41
    o.process(pkg.noise_host(1 << 18, 11))
    y, br, cbr = o.psd()
    f = o.frequencies(cbr)
    assert np.array_equal(f, pkg.Break.frequencies([pkg.Break(b["start"], bool(b["include"]), b["count"], b["avg"],
                                                               range(b["bins_start"], b["bins_end"]), b["fft_size"],
                                                               b["decimation"], b["pending"], b["processed"]) for b in br]))
    eff = br[0]["fft_size"] * br[0]["decimation"]
    tau, n_tau = 1.0, 0
    while tau <= eff // 2:
        a = pkg.var_eval(y, f, tau, dc_cut=1, clip=1.0)
        b = ora.var_eval(y, f, tau, dc_cut=1, clip=1.0)
        assert a == b and a >= 0.0
        tau *= 2.0
        n_tau += 1
    assert n_tau == int(np.log2(eff // 2)) + 1


@pytest.mark.parametrize("n", [48, 80, 1200])
def test_dft_by_definition_for_sizes_that_are_not_powers_of_two(ora, n):
    """rustfft plans any length (src/psd.rs:418); the oracle evaluates such a DFT by its definition."""
    rng = np.random.default_rng(n)
    c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    r = np.fft.fft(c)
    assert np.max(np.abs(ora.fft_forward(c, "f64") - r)) <= 1e-12 * np.max(np.abs(r))
    assert np.max(np.abs(ora.fft_forward(c, "f32") - r)) <= 5e-6 * np.max(np.abs(r))
    o = ora.PsdCascade(n, "f64")
    x = np.ones(n, dtype=np.float32)  # Hann of a constant: bins 0, 1 only; PSD[0] = n^2/4 / gain
    o.process(x)
    sp = o.stage_spectrum(0)
    assert abs(sp[0] - (n / 2.0) ** 2) < 1e-6 * sp[0] and abs(sp[1] - (n / 4.0) ** 2) < 1e-6 * sp[1] and np.all(sp[2:] < 1e-9 * sp[0])
