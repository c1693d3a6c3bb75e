/*
 * TEST INFRASTRUCTURE ONLY (oracle).  Not part of the shipped library.
 *
 * Half-band decimator tap table used by the CPU oracle.
 *
 * The reference (src/psd.rs:2,246-253) takes its decimator from the crates.io
 * dependency idsp 0.20.0 (Cargo.lock:1365-1366), `idsp::hbf::HBF_DEC_CASCADE`,
 * whose source is NOT present under /root/reference (not vendored, no network).
 * The values below are the published idsp `HBF_TAPS` rows.  They are DERIVED, not
 * recalled: idsp documents the design recipe of its table (src/hbf.rs:
 * "2*signal.remez(4*n-1, bands=(0,.5-df/2,.5+df/2,1), desired=(1,0), fs=2,
 * grid_density=512)[:2*n:2]", df = 0.2 / 0.47 / 0.754 for n = 15 / 6 / 3);
 * tests/golden/derive_hbf_taps.py runs it with scipy, commits the result as
 * tests/golden/hbf_taps_derived.json, and tests/test_hbf_taps.py holds this table
 * and the product's (csrc/hbf_taps.h) to it (<= 1e-8 per tap; bit-identical to
 * each other as f32).  idsp's taps are f32 constants: both oracle instantiations
 * round the rows to f32 first and widen that (psd_oracle_impl.h, hbf2_init).
 *
 * What stays unverifiable without the crate (idsp 0.20.0 is not vendored and there
 * is no network): that `HBF_DEC_CASCADE.inner.1.inner.1` (src/psd.rs:248-253)
 * selects exactly these three rows in this order, the summation order inside
 * idsp's FIR kernel (last-bit differences per output), and the VALUE of
 * hbf_dec_response_length(3) = 35 (src/psd.rs:149; the reference pins only the
 * length relation at :622).  Exact decimator sample parity vs the crate is
 * therefore still unpinned; the TAPS are pinned to idsp's published algorithm.
 *
 * What the reference itself pins, and this table satisfies (tests/test_oracle_*):
 *   - each unique-tap row sums to 0.5 => odd branch DC gain 1, centre tap 1,
 *     per-stage DC gain 2, /8 cascade DC gain 8 (src/psd.rs:516 with :634-643)
 *   - pass band 0.4 of the output rate (src/psd.rs:601, :498)
 *   - hbf_dec_response_length(3) relation y.len() == (len>>3) - d (src/psd.rs:622)
 *   - all included bins of all stages within 10/sqrt(count) of PSD=2 on unit
 *     white noise (src/psd.rs:634-643)
 *
 * Row order follows idsp: index 0 is the LOWEST-rate (last applied, sharpest)
 * filter.  A depth-3 decimator applies row 2, then row 1, then row 0.
 *
 * The table is swappable: if idsp 0.20.0's src/hbf.rs ever becomes available,
 * replace the numbers here and in stabilizer-stream_amd/csrc/hbf_taps.h and
 * regenerate tests/golden (tests/golden/make_golden.py).  No code change.
 */
#ifndef ORACLE_HBF_TAPS_H
#define ORACLE_HBF_TAPS_H

#define ORA_HBF_M0 15
#define ORA_HBF_M1 6
#define ORA_HBF_M2 3

static const double ORA_HBF_TAPS0[ORA_HBF_M0] = {
    7.02144012e-05, -2.43279582e-04, 6.35026936e-04, -1.39782541e-03,
    2.74613582e-03, -4.96403839e-03, 8.41806912e-03, -1.35827601e-02,
    2.11004053e-02, -3.19267647e-02, 4.77024289e-02, -7.18014345e-02,
    1.12942004e-01, -2.03279594e-01, 6.33592923e-01,
};
static const double ORA_HBF_TAPS1[ORA_HBF_M1] = {
    -0.00086943, 0.00577837, -0.02201674, 0.06357869, -0.16627679, 0.61979312,
};
static const double ORA_HBF_TAPS2[ORA_HBF_M2] = {
    0.01414651, -0.10439639, 0.59026742,
};

/* idsp: per-stage response length (in output samples) is 2*M-1; stages
 * combine as n = n/2 + len_i from the highest-rate stage down. */
static inline int ora_hbf_dec_response_length(int depth)
{
    const int m[3] = {ORA_HBF_M0, ORA_HBF_M1, ORA_HBF_M2};
    int n = 0;
    for (int i = depth - 1; i >= 0; --i)
        n = n / 2 + (2 * m[i] - 1);
    return n;
}

#endif
