"""Derive the half-band tap rows of idsp::hbf::HBF_TAPS from idsp's PUBLISHED recipe and commit them as a fixture.

The reference takes its /8 decimator from idsp 0.20.0 (Cargo.lock:1365-1366; call sites src/psd.rs:2,149,246-253),
which is not vendored under /root/reference.  idsp documents how its table was made (src/hbf.rs, doc comment of the
tap table): for n unique taps and transition width df,

    2 * scipy.signal.remez(4*n - 1, bands=(0, .5 - df/2, .5 + df/2, 1), desired=(1, 0), fs=2, grid_density=512)[:2*n:2]

with (n, df) = (15, 0.2), (6, 0.47), (3, 0.754) for the three /2 stages a depth-3 decimator uses (lowest rate first).
This script runs that recipe (build container only: needs scipy) and writes tests/golden/hbf_taps_derived.json;
tests/test_hbf_taps.py holds oracle/hbf_taps_oracle.h and stabilizer-stream_amd/csrc/hbf_taps.h to it.

    python tests/golden/derive_hbf_taps.py            # rewrite the fixture
    python tests/golden/derive_hbf_taps.py --check    # compare a fresh derivation with the committed fixture
"""
import json
import os
import sys

import numpy as np

RECIPE = [  # (unique taps n, transition width df), idsp row order: index 0 = lowest-rate (sharpest) stage
    (15, 0.2),
    (6, 0.47),
    (3, 0.754),
]
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "hbf_taps_derived.json")


def derive():
    from scipy.signal import remez
    rows = []
    for n, df in RECIPE:
        h = 2.0 * remez(4 * n - 1, (0, .5 - df / 2, .5 + df / 2, 1), (1, 0), fs=2, grid_density=512)
        # a half-band design: every second tap of the full filter is (numerically) zero except the centre
        full = np.asarray(h)
        centre = full[2 * n - 1]
        zeros = full[1:2 * n - 1:2]
        rows.append({
            "n": n, "df": df,
            "taps": [float(v) for v in full[:2 * n:2]],
            "taps_f32": [float(np.float32(v)) for v in full[:2 * n:2]],
            "centre_tap": float(centre),
            "max_abs_even_offcentre": float(np.max(np.abs(zeros))),
            "sum_unique": float(np.sum(full[:2 * n:2])),
        })
    return rows


def response_length(ms, depth):
    """idsp::hbf::hbf_dec_response_length: per-stage length 2M-1 output samples, combined n = n/2 + len_i
    from the input side down (src/psd.rs:149,622 use depth 3)."""
    n = 0
    for i in range(depth - 1, -1, -1):
        n = n // 2 + (2 * ms[i] - 1)
    return n


def main():
    rows = derive()
    doc = {
        "source": "idsp 0.20.0 src/hbf.rs published recipe, run with scipy.signal.remez (see derive_hbf_taps.py)",
        "recipe": "2*remez(4*n-1, (0,.5-df/2,.5+df/2,1), (1,0), fs=2, grid_density=512)[:2*n:2]",
        "rows": rows,
        "hbf_dec_response_length_3": response_length([r["n"] for r in rows], 3),
        "unverifiable_here": [
            "that HBF_DEC_CASCADE.inner.1.inner.1 (src/psd.rs:248-253) selects exactly these three rows in this order",
            "the summation order inside idsp's FIR kernel (affects the last f32 bit of each output only)",
            "the value hbf_dec_response_length(3) = 35 (src/psd.rs:149); the reference pins only the relation :622",
        ],
    }
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        for a, b in zip(old["rows"], rows):
            d = float(np.max(np.abs(np.array(a["taps"]) - np.array(b["taps"]))))
            print(f"n={a['n']}: fresh derivation vs fixture max |diff| = {d:.3g}")
            assert d < 1e-12
        return
    json.dump(doc, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
