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


# Tolerance of every PSD parity check (BASELINE.json: "PSD within 1e-5 relative of
# the CPU reference"):
#   |gpu - ref| <= RTOL*ref + ATOL_FRAC*mean(ref) + DYN*sqrt(ref*max(ref)).
# ATOL_FRAC only matters for bins that a detrend nulls (e.g. DC under Detrend::Mean).
# DYN is the dynamic-range floor of ANY f32 FFT (the reference's rustfft included): a
# bin's amplitude carries an error of ~1e-7 of the largest component of the frame, so a
# bin 40 dB below a strong tone cannot be known to 1e-5 in power from f32 arithmetic.
RTOL = 1e-5
ATOL_FRAC = 1e-6
DYN = 5e-7


def assert_psd_close(got, ref, what="", rtol=RTOL, atol_frac=ATOL_FRAC, dyn=DYN):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0.0
    # a stage with count 0 included by min_count = 0 reads 0 * (1/0) = NaN, in the reference too
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    got, ref = got[~both_nan], ref[~both_nan]
    if ref.size == 0:
        return 0.0
    tol = rtol * np.abs(ref) + atol_frac * np.mean(np.abs(ref)) + dyn * np.sqrt(np.abs(ref) * np.max(np.abs(ref)))
    err = np.abs(got - ref)
    worst = int(np.argmax(err / tol))
    assert np.all(err <= tol), (f"{what}: bin {worst} got {got[worst]:.9g} ref {ref[worst]:.9g} "
                                f"err/tol {err[worst] / tol[worst]:.3g}")
    return float(np.max(err / np.maximum(np.abs(ref), 1e-300)))


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
