// fft_team.h -- N-point forward complex FFT by a TEAM of N/16 lanes of one
// wavefront (N = 256, 512, 1024 -> 16, 32, 64 lanes; 4, 2, 1 independent FFTs
// per wavefront), 16 elements per lane, radix (4, N/64, 16) decimation in
// frequency, two exchanges through a padded LDS frame private to the team.
//
// Why this plan: the pass-0 butterflies of team-lane tl work on elements
// n = 4 tl + c + (N/4) m (c, m = 0..3), exactly what four 16-byte loads
// x[(N/4) m + 4 tl .. +3] deliver -- the sample stream goes HBM -> VGPR in
// dwordx4 pieces with no staging copy.  After pass 2, register slot q of
// team-lane tl holds bin  k = q0 + 4 q1 + 4 R1 q  with q0 = tl / R1,
// q1 = tl % R1, R1 = N/64.  Only |X|^2 is consumed (src/psd.rs:228-233).
//
// Frame: physical = idx + idx/16 (N + N/16 elements).  Every 8-byte LDS access
// of every pass is bank-conflict free under the gfx950 rules, also across the
// teams of one wavefront (tests/host/fft_emul.cpp counts them), and every
// address is lane_base + constant, so the constants ride in the DS offset field.
#pragma once
#include "fft_core.h"

namespace psdk {

template <int N>
struct TeamFft {
    static_assert(N == 256 || N == 512 || N == 1024, "team FFT sizes");
    static constexpr int TEAM = N / 16;       // lanes per FFT
    static constexpr int TPW = 64 / TEAM;     // FFTs per wavefront
    static constexpr int R1 = N / 64;         // pass-1 radix (4, 8, 16)
    static constexpr int NB1 = 16 / R1;       // pass-1 butterflies per lane
    static constexpr int L1 = N / 4;          // pass-1 sub-transform length
    static constexpr int FRAME = N + N / 16;  // padded frame, complex elements
    static constexpr int TW0_SIZE = 4 * TEAM; // W_N^(4 tl + c), [c][tl]
    static constexpr int TW1_SIZE = (R1 - 1) * 16; // W_L1^(s q), [(q-1)][s]

    static PSDK_HD int swz(int idx) { return idx + (idx >> 4); }

    static PSDK_HD int freq_of(int tl, int q) { return tl / R1 + 4 * (tl % R1) + 4 * R1 * q; }

    // pass 0: v[4m + c] holds z[4 tl + c + (N/4) m]; afterwards v[4q + c] is output q of
    // butterfly s = 4 tl + c times W_N^(s q).  Only W^s is tabulated (tw0[c*TEAM + tl]);
    // W^2s and W^3s are formed by multiplication (LDS space goes to the decimator state).
    static PSDK_HD void pass0(int tl, cf *v, const cf *tw0)
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cf b[4] = {v[c], v[4 + c], v[8 + c], v[12 + c]};
            Dft<4>::run(b);
            const cf w1 = tw0[c * TEAM + tl]; // (plain loads: see pass1)
            const cf w2 = cmul(w1, w1);
            const cf w3 = cmul(w2, w1);
            v[c] = b[0];
            v[4 + c] = cmul(b[1], w1);
            v[8 + c] = cmul(b[2], w2);
            v[12 + c] = cmul(b[3], w3);
        }
    }

    static PSDK_HD void store0(int tl, const cf *v, cf *frame)
    {
        cf *base = frame + (4 * tl + (tl >> 2)); // swz(q L1 + 4 tl + c) = base + q (L1 + L1/16) + c
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                base[(L1 + L1 / 16) * q + c] = v[4 * q + c];
    }

    // pass 1: butterfly i of the lane works in block b = NB1 (tl / 16) + i with s = tl % 16:
    // elements b L1 + s + 16 m, m < R1; register slots v[R1 i + m].  (This block assignment keeps
    // the two 16-lane halves of an N = 512 team on different banks.)
    static constexpr int STEP1 = L1 + L1 / 16; // physical distance between blocks
    static PSDK_HD int base1(int tl)
    {
        return STEP1 * NB1 * (tl >> 4) + (tl & 15); // + STEP1 i + 17 m
    }
    static PSDK_HD void load1(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + base1(tl);
#pragma unroll
        for (int i = 0; i < NB1; ++i)
#pragma unroll
            for (int m = 0; m < R1; ++m)
                v[R1 * i + m] = lds_ld(base + STEP1 * i + 17 * m);
    }
    static PSDK_HD void pass1(int tl, cf *v, const cf *tw1)
    {
        const int s = tl & 15;
#if PSDK_TW_ROWS
        if constexpr (NB1 == 1) { // (measured per size: see twiddle_rows in fft_core.h)
            Dft<R1>::run(v);
            twiddle_rows<R1, 1>(v, tw1 + s);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NB1; ++i) {
            Dft<R1>::run(v + R1 * i);
#pragma unroll
            for (int q = 1; q < R1; ++q)
                // plain (not lds_ld) loads: kept single, the R1 - 1 twiddle reads ended up serialised
                // through one register pair -- a chain of LDS round trips; left to the compiler they pair
                // up as ds_read2_b64 (dearer per byte) but are all in flight together: +2.5 % at N = 1024,
                // +13 % at N = 256
                v[R1 * i + q] = cmul(v[R1 * i + q], tw1[(q - 1) * 16 + s]);
        }
    }
    static PSDK_HD void store1(int tl, const cf *v, cf *frame)
    {
        cf *base = frame + base1(tl);
#pragma unroll
        for (int i = 0; i < NB1; ++i)
#pragma unroll
            for (int q = 0; q < R1; ++q)
                base[STEP1 * i + 17 * q] = v[R1 * i + q];
    }

    // pass 2: 16 consecutive elements per lane
    static PSDK_HD void load2(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + 17 * tl; // swz(16 tl + m) = 17 tl + m
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = lds_ld(base + m);
    }
    static PSDK_HD void pass2(cf *v) { Dft<16>::run(v); }
};

} // namespace psdk
