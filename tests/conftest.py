import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes over libpsdcascade.so). Built on demand, never falls back."""
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    p.lib()
    return p


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure only)."""
    o = entry.load_oracle()
    o.lib()
    return o


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not has_gpu():
        pytest.fail("test marked gpu but no HIP device is visible")


# Tolerance of the PSD parity checks (BASELINE.json: "PSD within 1e-5 relative of the CPU reference").
#
# PURE:   |gpu - ref| <= RTOL * ref on every bin.  Asserted wherever the spectrum is white noise with enough
#         averages (the headline path): nothing but the stated 1e-5.
# WIDENED (signals with a strong tone / DC level, nulled bins, stages with one or two segments):
#         |gpu - ref| <= RTOL*ref + ATOL_FRAC*mean(ref) + DYN*sqrt(ref*max(ref)).
#         DYN is the dynamic-range floor of ANY f32 FFT (the reference's rustfft included): a bin's amplitude
#         carries an error of ~1e-7 of the largest component of the frame, so a bin 40 dB below a strong tone
#         cannot be known to 1e-5 in power from f32 arithmetic; ATOL_FRAC only matters for bins that a detrend
#         nulls (e.g. DC under Detrend::Mean).
# The widening is not taken on trust.  When the f32 oracle's result (the reference's own arithmetic, oracle *_f32)
# is passed as `ref_f32`:
#   (a) over the bins whose tolerance the extra terms more than double (an a-priori set), the GPU's rms error must be
#       no larger than the f32 reference's own, or meet the pure 1e-5 there in the rms sense;
#   (b) EVERY bin whose error exceeds the pure 1e-5 bound -- wherever it sits -- may use no more of its widened
#       tolerance than EXCESS_K (= 6) x what the f32 reference's own arithmetic uses at ITS worst bin of the same spectrum
#       (err / tol against max_k e32_k / tol_k).  The comparison is with the reference's worst bin, not with the same
#       bin: on signals that need the widening (a step of 1e5 sigma inside a segment, a tone 60 dB above the noise) the
#       errors of an f32 FFT are outliers at bins its radix structure picks, and the GPU's (4, 16, ..., 16) passes pick
#       others than a radix-2 does -- tools/dbg_excess.py: equal rms, different bins.  So no bin passes on the widening
#       unless the reference's own arithmetic needs as much of it somewhere in the same spectrum.
# The terminal summary names the test and bin with the largest excess and the one closest to cap (b); assertions
# that are widened WITHOUT a ref_f32 are counted and their worst non-widened bin is named too.
RTOL = 1e-5
ATOL_FRAC = 1e-6
DYN = 5e-7
EXCESS_K = 6.0  # (two samples of heavy-tailed rounding errors: the fuzz campaigns' extremes over ~12 000 spectra are 4.1x and 5.1x, both the
#                 Nyquist bin of a stage with two or three averages under finite averaging, each below 0.2 of its widened tolerance;
#                 the suite's own closest is 3.06x)
WORST = {"pure": (0.0, ""), "widened": (0.0, ""), "excess_vs_f32": (0.0, ""), "unjustified": (0.0, "")}
COUNTS = {"pure": 0, "justified": 0, "unjustified": 0, "excess_bins": 0}


def _local_rms(v, half=8):
    k = np.ones(2 * half + 1)
    return np.sqrt(np.convolve(v * v, k, "same") / np.convolve(np.ones_like(v), k, "same"))


def _note(kind, value, text):
    if value > WORST[kind][0]:
        WORST[kind] = (value, text)


def assert_psd_close(got, ref, what="", rtol=RTOL, atol_frac=ATOL_FRAC, dyn=DYN, pure=False, ref_f32=None):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0.0
    # a stage with count 0 included by min_count = 0 reads 0 * (1/0) = NaN, in the reference too
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    keep = ~both_nan
    got, ref = got[keep], ref[keep]
    if ref.size == 0:
        return 0.0
    base = rtol * np.abs(ref)
    if pure:
        tol = base
    else:
        tol = base + atol_frac * np.mean(np.abs(ref)) + dyn * np.sqrt(np.abs(ref) * np.max(np.abs(ref)))
    err = np.abs(got - ref)
    worst = int(np.argmax(err / np.maximum(tol, 1e-300)))
    assert np.all(err <= tol), (f"{what}: bin {worst} got {got[worst]:.9g} ref {ref[worst]:.9g} "
                                f"err/tol {err[worst] / tol[worst]:.3g} ({'pure 1e-5' if pure else 'widened'})")
    relv = err / np.maximum(np.abs(ref), 1e-9 * np.max(np.abs(ref)) + 1e-300)  # (bins a detrend nulls: relative to the spectrum's scale)
    rel = float(np.max(relv))
    if pure:
        COUNTS["pure"] += 1
        _note("pure", rel, f"{what}, bin {int(np.argmax(relv))}")
        return rel
    wide = tol > 2.0 * base  # the bins that lean on the extra terms (a-priori set)
    excess = err > base      # the bins that actually exceed the pure bound
    if ref_f32 is not None:
        COUNTS["justified"] += 1
        e32 = np.abs(np.asarray(ref_f32, dtype=np.float64)[keep] - ref)
        rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
        if np.any(wide):
            g, r, p = rms(err[wide]), rms(e32[wide]), rms(base[wide])
            assert g <= max(r, p), (f"{what}: on the {int(wide.sum())} widened bins the GPU's rms error {g:.3g} exceeds both the "
                                    f"f32 reference arithmetic's {r:.3g} and the pure 1e-5 level {p:.3g}")
        if np.any(excess):
            COUNTS["excess_bins"] += int(excess.sum())
            used_g, used_f = err / np.maximum(tol, 1e-300), e32 / np.maximum(tol, 1e-300)
            f_worst = float(np.max(used_f))
            cap = np.maximum(EXCESS_K * f_worst, base / np.maximum(tol, 1e-300))
            k = int(np.argmax(np.where(excess, used_g, 0.0)))
            assert np.all(used_g[excess] <= cap[excess]), (
                f"{what}: bin {k} exceeds the pure 1e-5 bound (rel {relv[k]:.3g}) using {used_g[k]:.3g} of its widened tolerance, "
                f"more than {EXCESS_K}x the {f_worst:.3g} the f32 reference arithmetic uses at its worst bin "
                f"({int(np.argmax(used_f))}) of this spectrum")
            ratio = float(used_g[k] / max(f_worst, 1e-300))
            _note("excess_vs_f32", ratio, f"{what}, bin {k}: rel err {relv[k]:.3g}; the f32 reference's worst bin {int(np.argmax(used_f))} "
                                          f"has rel err {float(e32[int(np.argmax(used_f))] / max(abs(ref[int(np.argmax(used_f))]), 1e-300)):.3g}")
            ke = int(np.argmax(np.where(excess, relv, 0.0)))
            _note("widened", float(relv[ke]), f"{what}, bin {ke}")
        return float(np.max(relv[~excess])) if np.any(~excess) else 0.0
    COUNTS["unjustified"] += 1
    nw = relv[~wide]
    if nw.size:
        _note("unjustified", float(np.max(nw)), f"{what}, bin {int(np.flatnonzero(~wide)[int(np.argmax(nw))])}")
        rel = float(np.max(nw))
    else:
        rel = 0.0
    return rel


def pytest_terminal_summary(terminalreporter):
    if not (COUNTS["pure"] or COUNTS["justified"] or COUNTS["unjustified"]):
        return
    w = terminalreporter.write_line
    w(f"PSD parity: {COUNTS['pure']} pure-1e-5 assertions, worst relative error {WORST['pure'][0]:.3g} ({WORST['pure'][1]})")
    w(f"PSD parity: {COUNTS['justified']} widened assertions held to the f32 reference bin by bin; {COUNTS['excess_bins']} bins beyond the "
      f"pure 1e-5, the largest {WORST['widened'][0]:.3g} ({WORST['widened'][1]}); "
      f"most of the widened tolerance used relative to the f32 reference's own worst bin (cap {EXCESS_K}x): "
      f"{WORST['excess_vs_f32'][0]:.3g}x ({WORST['excess_vs_f32'][1]})")
    w(f"PSD parity: {COUNTS['unjustified']} widened assertions without an f32 comparison (GPU vs GPU, golden fixtures); worst relative "
      f"error on their non-widened bins {WORST['unjustified'][0]:.3g} ({WORST['unjustified'][1]})")


def assert_psd_close_anchored(got, ref, n, count, xmax, what=""):
    """assert_psd_close for spectra taken under Midpoint / Span detrend (src/psd.rs:87-102), whose
    offset is anchored on ONE sample of the segment.  A stage >= 1 stream is f32 (in the reference
    too): the anchor sample carries a rounding error of up to ~1 ulp(|x|) against the f64 oracle's
    stream, and that error is a coherent offset over the whole segment -- it lands in bins 0 and 1
    with the window's weight (Hann: N/2 and N/4).  Per segment the power moves by up to
    2 |X[k]| ulp W[k]; over `count` segments by 2 ulp W[k] sqrt(count * P[k]) at most.  All other
    bins keep the plain tolerance."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    tol = RTOL * np.abs(ref) + ATOL_FRAC * np.mean(np.abs(ref)) + DYN * np.sqrt(np.abs(ref) * np.max(np.abs(ref)))
    ulp = float(np.spacing(np.float32(xmax)))
    for k, wk in ((0, n / 2.0), (1, n / 4.0)):
        if k < ref.size:
            tol[k] += 2.0 * ulp * wk * np.sqrt(max(1, count) * abs(ref[k])) + (ulp * wk) ** 2 * max(1, count)
    err = np.abs(got - ref)
    worst = int(np.argmax(err / tol))
    assert np.all(err <= tol), (f"{what}: bin {worst} got {got[worst]:.9g} ref {ref[worst]:.9g} "
                                f"err/tol {err[worst] / tol[worst]:.3g}")


def test_signal(pkg, length, seed, tone=0.0, dc=0.0, f0=0.01234):
    """Unit-variance uniform noise (src/psd.rs:604-606) + optional tone and offset."""
    x = pkg.noise_host(length, seed)
    if tone:
        x = x + np.float32(tone) * np.sin(2 * np.pi * f0 * np.arange(length)).astype(np.float32)
    if dc:
        x = x + np.float32(dc)
    return x.astype(np.float32)


test_signal.__test__ = False
