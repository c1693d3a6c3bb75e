// Half-band decimator taps of the /8 cascade (src/psd.rs:246-253 selects the
// last three /2 stages of idsp::hbf::HBF_DEC_CASCADE, idsp 0.20.0).
//
// idsp is a crates.io dependency of the reference and is not vendored under
// /root/reference; the numbers are the published idsp HBF_TAPS rows, derived from the
// design recipe idsp documents (remez, df = 0.754 / 0.47 / 0.2 for 3 / 6 / 15 unique
// taps): tests/golden/derive_hbf_taps.py regenerates them, tests/test_hbf_taps.py holds
// this table to the derivation (<= 1e-8 per tap) and to the oracle's table bit for bit.
// Each unique-tap row sums to 0.5 (centre tap 1, stage DC gain 2).
// ONE swappable table: replace these rows (and oracle/hbf_taps_oracle.h, then
// regenerate tests/golden) if the crate's table differs -- no code change.
//
// Stage order of application at depth 3: STAGE_A (3 unique taps, input rate),
// STAGE_B (6 taps, rate/2), STAGE_C (15 taps, rate/4).
// One stage:  y[j] = xe[j-(M-1)] + sum_{i<M} t[i] * (xo[j-(2M-1)+i] + xo[j-i]),
// xe[m] = x[2m], xo[m] = x[2m+1], zero initial state.
#pragma once

namespace psdk {

constexpr int HBF_MA = 3, HBF_MB = 6, HBF_MC = 15;

#define PSDK_HBF_TAPS_A 0.01414651f, -0.10439639f, 0.59026742f
#define PSDK_HBF_TAPS_B -0.00086943f, 0.00577837f, -0.02201674f, 0.06357869f, -0.16627679f, 0.61979312f
#define PSDK_HBF_TAPS_C                                                                          \
    7.02144012e-05f, -2.43279582e-04f, 6.35026936e-04f, -1.39782541e-03f, 2.74613582e-03f,       \
        -4.96403839e-03f, 8.41806912e-03f, -1.35827601e-02f, 2.11004053e-02f, -3.19267647e-02f,  \
        4.77024289e-02f, -7.18014345e-02f, 1.12942004e-01f, -2.03279594e-01f, 6.33592923e-01f

// idsp: stage response length 2M-1 (output samples); n = n/2 + len_i combined
// from the input side down.  depth 3 -> 35 (src/psd.rs:149 `drain`).
constexpr int hbf_response_length(int depth)
{
    const int m[3] = {HBF_MC, HBF_MB, HBF_MA};
    int n = 0;
    for (int i = depth - 1; i >= 0; --i)
        n = n / 2 + (2 * m[i] - 1);
    return n;
}
constexpr int HBF_DRAIN = hbf_response_length(3);

// Input history needed before the first new sample 8*m0 of a block so that a
// block can be evaluated independently of the previous one:
//   C output m needs B[2m-(4MC-3) .. 2m+1], B output b needs A[2b-(4MB-3) .. 2b+1],
//   A output a needs x[2a-(4MA-3) .. 2a+1].
constexpr int HBF_SPAN_A = 4 * HBF_MA - 3; // 9
constexpr int HBF_SPAN_B = 4 * HBF_MB - 3; // 21
constexpr int HBF_SPAN_C = 4 * HBF_MC - 3; // 57
// Block origins are kept even so that each stage's even/odd polyphase split is
// aligned: B0 = 2*m0 - HBF_PRE_B, A0 = 4*m0 - HBF_PRE_A, X0 = 8*m0 - HBF_HALO.
constexpr int HBF_PRE_B = (HBF_SPAN_C + 1) / 2 * 2;                 // 58
constexpr int HBF_PRE_A = (2 * HBF_PRE_B + HBF_SPAN_B + 1) / 2 * 2; // 138
constexpr int HBF_HALO = (2 * HBF_PRE_A + HBF_SPAN_A + 7) / 8 * 8;  // 288 inputs, multiple of 8

} // namespace psdk
