"""The half-band tap tables are held to idsp's published design recipe.

tests/golden/hbf_taps_derived.json is produced by tests/golden/derive_hbf_taps.py (scipy.signal.remez with the
parameters idsp 0.20.0 documents for HBF_TAPS).  Both tables of this repo -- the oracle's
(oracle/hbf_taps_oracle.h) and the product's (stabilizer-stream_amd/csrc/hbf_taps.h) -- must match it, and each
other bit for bit as f32 (idsp's taps are f32 constants)."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "hbf_taps_derived.json")))

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def oracle_rows():
    text = _strip_comments(open(os.path.join(ROOT, "oracle", "hbf_taps_oracle.h")).read())
    rows = []
    for i in range(3):
        m = re.search(r"ORA_HBF_TAPS%d\[[^\]]*\]\s*=\s*\{([^}]*)\}" % i, text)
        rows.append([float(v) for v in re.findall(NUM, m.group(1))])
    return rows  # idsp order: lowest-rate stage first


def product_rows():
    text = _strip_comments(open(os.path.join(ROOT, "stabilizer-stream_amd", "csrc", "hbf_taps.h")).read())
    text = text.replace("\\\n", " ")
    rows = {}
    for name in "ABC":
        m = re.search(r"#define\s+PSDK_HBF_TAPS_%s\s+([^\n]*)" % name, text)
        rows[name] = [float(v) for v in re.findall(NUM + r"(?=f)", m.group(1))]
    return [rows["C"], rows["B"], rows["A"]]  # stage C is the lowest-rate one


def test_fixture_is_the_published_recipe():
    assert [(r["n"], r["df"]) for r in FIX["rows"]] == [(15, 0.2), (6, 0.47), (3, 0.754)]
    for r in FIX["rows"]:
        assert len(r["taps"]) == r["n"]
        assert abs(r["centre_tap"] - 1.0) < 1e-7          # centre tap 1: no multiply, stage DC gain 2
        assert r["max_abs_even_offcentre"] < 1e-7         # a true half-band design
        assert abs(r["sum_unique"] - 0.5) < 2e-5          # odd branch DC gain 1 (to the design's ripple)
    assert FIX["hbf_dec_response_length_3"] == 35


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_tables_match_the_derivation(which):
    rows = oracle_rows() if which == "oracle" else product_rows()
    for got, ref in zip(rows, FIX["rows"]):
        assert len(got) == ref["n"]
        d = np.max(np.abs(np.array(got) - np.array(ref["taps"])))
        assert d <= 1e-8, f"{which} row n={ref['n']}: max |tap - derived| = {d:.3g}"


def test_the_two_tables_agree_bitwise_as_f32():
    for a, b, ref in zip(oracle_rows(), product_rows(), FIX["rows"]):
        fa, fb = np.array(a, dtype=np.float32), np.array(b, dtype=np.float32)
        assert fa.tobytes() == fb.tobytes(), f"row n={ref['n']} differs between oracle and product"
        # and the f32 constants sit within 1e-8 of the f32-rounded derivation (the published rows carry 8-9 digits)
        fr = np.array(ref["taps_f32"], dtype=np.float32)
        assert np.max(np.abs(fa.astype(np.float64) - fr.astype(np.float64))) <= 1e-8, f"row n={ref['n']}"


def test_library_response_length_matches(pkg):
    L = pkg.lib()
    assert L.psdc_hbf_response_length(3) == FIX["hbf_dec_response_length_3"]


def test_fresh_derivation_matches_fixture():
    scipy_signal = pytest.importorskip("scipy.signal")
    for r in FIX["rows"]:
        n, df = r["n"], r["df"]
        h = 2.0 * scipy_signal.remez(4 * n - 1, (0, .5 - df / 2, .5 + df / 2, 1), (1, 0), fs=2, grid_density=512)
        d = np.max(np.abs(h[:2 * n:2] - np.array(r["taps"])))
        assert d < 1e-9, f"n={n}: scipy derivation moved by {d:.3g}"


def test_cascade_meets_the_reference_pins():
    """What src/psd.rs pins about the /8 cascade: pass band 0.4 of the output rate flat,
    DC amplitude gain 8, aliases from outside the pass band suppressed."""
    rows = FIX["rows"]

    def full(taps):  # full symmetric half-band impulse response, centre tap 1
        n = len(taps)
        h = np.zeros(4 * n - 1)
        h[0:2 * n:2] = taps
        h[2 * n - 1] = 1.0
        h[2 * n::2] = taps[::-1]
        return h

    ha, hb, hc = full(rows[2]["taps"]), full(rows[1]["taps"]), full(rows[0]["taps"])
    # composite at the input rate: A, then B upsampled by 2, then C upsampled by 4
    up = lambda h, k: np.kron(h, np.r_[1.0, np.zeros(k - 1)])[: (len(h) - 1) * k + 1]
    comp = np.convolve(np.convolve(ha, up(hb, 2)), up(hc, 4))
    assert len(comp) == 287  # 11 + 2*22 + 4*58 -> 286-sample halo
    # `hbf_dec_response_length(3)` (src/psd.rs:149): idsp's rule (per stage 2M-1 output samples, n = n/2 + len_i) gives 35,
    # which is also the length of the composite impulse response counted in OUTPUT samples, floor(287 / 8): the number of
    # outputs the zero initial state still shapes -- what the one-time drain of src/psd.rs:255-260 is there to discard
    assert FIX["hbf_dec_response_length_3"] == len(comp) // 8 == 35
    nfft = 1 << 16
    H = np.abs(np.fft.rfft(comp, nfft))
    f = np.arange(H.size) / nfft  # cycles per input sample; output Nyquist = 1/16
    assert abs(H[0] - 8.0) < 1e-3
    pass_edge = 0.4 / 8
    ripple_db = 20 * np.log10(H[f <= pass_edge] / 8.0)
    assert np.max(np.abs(ripple_db)) < 1e-3
    # everything that aliases INTO the pass band of the output: bands m/8 +- pass_edge, m = 1..4
    alias = np.zeros(H.size, dtype=bool)
    for m in range(1, 5):
        alias |= np.abs(f - m / 8.0) <= pass_edge
    assert 20 * np.log10(np.max(H[alias]) / 8.0) < -95.0
