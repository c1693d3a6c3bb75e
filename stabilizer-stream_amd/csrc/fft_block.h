// fft_block.h -- N-point forward complex FFT by a whole workgroup of N/16
// threads (N = 2048 ... 16384 -> 128 ... 1024 threads), 16 elements per thread,
// radix (4, RA, RB, 16) decimation in frequency with RA*RB*16 = N/4, three
// exchanges through a padded LDS frame (physical = idx + idx/16).  RA = 8 or 16.
//
// Same idea as fft_team.h one size class up: thread tl's pass-0 butterflies work
// on n = 4 tl + c + (N/4) m, i.e. on what four 16-byte loads deliver.  After the
// last pass register slot q of thread tl holds bin
//   k = q0 + 4 q1 + 4 RA q2 + 4 RA RB q,   q0 = tl / (RA RB), q1 = (tl % (RA RB)) / RB,
//   q2 = tl % RB.
#pragma once
#include "fft_core.h"

namespace psdk {

template <int N>
struct BlockFft {
    static_assert(N == 2048 || N == 4096 || N == 8192 || N == 16384, "block FFT sizes");
    static constexpr int TEAM = N / 16;
    static constexpr int L1 = N / 4;                         // sub-transform length after pass 0
    static constexpr int RA = N == 2048 ? 8 : 16;            // pass-1 radix
    static constexpr int SA = L1 / RA;                       // pass-1 stride = pass-2 length
    static constexpr int RB = SA / 16;                       // pass-2 radix (4, 4, 8, 16)
    static constexpr int NBA = 16 / RA, NBB = 16 / RB;       // butterflies per thread in passes 1, 2
    static constexpr int FRAME = N + N / 16;
    static constexpr int STEP0 = L1 + L1 / 16;               // physical distance between pass-0 output blocks
    static constexpr int STEPA = SA + SA / 16;               // ... between pass-1 elements / output blocks
    static constexpr int TW0_SIZE = 4 * TEAM;                // W_N^(4 tl + c), [c][tl]
    static constexpr int TWA_SIZE = (RA - 1) * SA;           // W_L1^(s q), [(q-1)][s]
    static constexpr int TWB_SIZE = (RB - 1) * 16;           // W_SA^(s q), [(q-1)][s]

    static PSDK_HD int swz(int idx) { return idx + (idx >> 4); }

    static PSDK_HD int freq_of(int tl, int q)
    {
        const int r = tl % (RA * RB);
        return tl / (RA * RB) + 4 * (r / RB) + 4 * RA * (r % RB) + 4 * RA * RB * q;
    }

    // The twiddles of passes 0 and 1 come from global tables (tw0 [c][tl] = W_N^(4 tl + c), twa [(q-1)][s] =
    // W_L1^(s q): too large for the LDS beside the frame).  A lane reads only three SEEDS of them per pair --
    // W_N^(4 tl), W_L1^s, W_L1^(4 s) -- plus three uniform entries W_N^c (scalar loads), all issued together at
    // the top of the pair so that they are in flight during the decimator, and forms the rest by
    // multiplication (at most three factors deep: ~2e-7 relative, below the rounding of the butterflies).
    // Reading all 4 + 15 entries instead cost the workgroup-level kernels most of their time: under the
    // 128-register budget the backend issued them one at a time, each an exposed L2 round trip (19 dependent
    // vmcnt(0) waits per pair in the N = 4096 listing; SQ_WAIT_ANY 42 % of wave time, profiles/r02b_n4096_sq.json).
    struct Seeds {
        cf w0;       // W_N^(4 tl)
        cf wc[3];    // W_N^1, W_N^2, W_N^3 (the same for every lane)
    };
    struct SeedsA {
        cf a1, a4;   // W_L1^s, W_L1^(4 s), s = tl % SA
    };
    static PSDK_HD Seeds load_seeds(int tl, const cf *tw0)
    {
        Seeds sd;
        sd.w0 = tw0[tl];
#pragma unroll
        for (int c = 1; c < 4; ++c)
            sd.wc[c - 1] = tw0[c * TEAM];
        return sd;
    }
    static PSDK_HD SeedsA load_seeds_a(int tl, const cf *twa)
    {
        const int s = tl % SA;
        return {twa[s], twa[3 * SA + s]};
    }

    // pass 0 (as in fft_team.h): v[4m + c] = z[4 tl + c + L1 m] -> v[4q + c] = output q of s = 4 tl + c
    static PSDK_HD void pass0(int tl, cf *v, const Seeds &sd)
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cf b[4] = {v[c], v[4 + c], v[8 + c], v[12 + c]};
            Dft<4>::run(b);
            const cf w1 = c == 0 ? sd.w0 : cmul(sd.w0, sd.wc[c > 0 ? c - 1 : 0]); // W_N^(4 tl + c)
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
        cf *base = frame + (4 * tl + (tl >> 2));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                base[STEP0 * q + c] = v[4 * q + c];
    }

    // pass 1: butterfly i of the thread is u = tl + TEAM i in [0, 4 SA): block b = u / SA, s = u % SA;
    // elements b L1 + s + SA m, m < RA; slots v[RA i + m].  TEAM is a multiple of SA.
    static PSDK_HD int sA(int tl) { return tl % SA; }
    static PSDK_HD int baseA(int tl)
    {
        const int s = tl % SA, b = tl / SA;
        return STEP0 * b + s + (s >> 4); // + STEP0 (TEAM / SA) i + STEPA m
    }
    static PSDK_HD void loadA(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + baseA(tl);
#pragma unroll
        for (int i = 0; i < NBA; ++i)
#pragma unroll
            for (int m = 0; m < RA; ++m)
                v[RA * i + m] = lds_ld(base + STEP0 * (TEAM / SA) * i + STEPA * m);
    }
    // twiddles W_L1^(s q), q = 1 ... RA - 1, from the seeds a1 = W^s and a4 = W^(4 s): W^(4k + j) = W^(4k) W^j
    static PSDK_HD void passA(int tl, cf *v, const SeedsA &sd)
    {
        const cf w1 = sd.a1, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
        for (int i = 0; i < NBA; ++i) {
            Dft<RA>::run(v + RA * i);
            cf wb = sd.a4; // W^(4 k s), k = 1, 2, 3
            v[RA * i + 1] = cmul(v[RA * i + 1], w1);
            v[RA * i + 2] = cmul(v[RA * i + 2], w2);
            v[RA * i + 3] = cmul(v[RA * i + 3], w3);
#pragma unroll
            for (int k = 1; 4 * k < RA; ++k) {
                v[RA * i + 4 * k] = cmul(v[RA * i + 4 * k], wb);
                v[RA * i + 4 * k + 1] = cmul(v[RA * i + 4 * k + 1], cmul(wb, w1));
                v[RA * i + 4 * k + 2] = cmul(v[RA * i + 4 * k + 2], cmul(wb, w2));
                v[RA * i + 4 * k + 3] = cmul(v[RA * i + 4 * k + 3], cmul(wb, w3));
                if (4 * (k + 1) < RA)
                    wb = cmul(wb, sd.a4);
            }
        }
    }
    static PSDK_HD void storeA(int tl, const cf *v, cf *frame)
    {
        cf *base = frame + baseA(tl);
#pragma unroll
        for (int i = 0; i < NBA; ++i)
#pragma unroll
            for (int q = 0; q < RA; ++q)
                base[STEP0 * (TEAM / SA) * i + STEPA * q] = v[RA * i + q];
    }

    // pass 2: sub-blocks of length SA (index sb = NBB (tl / 16) + i), stride 16, s = tl % 16;
    // elements sb SA + s + 16 m, m < RB; slots v[RB i + m]
    static PSDK_HD int baseB(int tl) { return STEPA * NBB * (tl >> 4) + (tl & 15); } // + STEPA i + 17 m
    static PSDK_HD void loadB(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + baseB(tl);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
#pragma unroll
            for (int m = 0; m < RB; ++m)
                v[RB * i + m] = lds_ld(base + STEPA * i + 17 * m);
    }
    // twb[(q-1) * 16 + s] = W_SA^(s q), in LDS
    static PSDK_HD void passB(int tl, cf *v, const cf *twb)
    {
        const int s = tl & 15;
#if PSDK_TW_ROWS
        if constexpr (PSDK_TW_ROWS & 2) {
#pragma unroll
            for (int i = 0; i < NBB; ++i)
                Dft<RB>::run(v + RB * i);
            twiddle_rows<RB, NBB>(v, twb + s);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NBB; ++i) {
            Dft<RB>::run(v + RB * i);
#pragma unroll
            for (int q = 1; q < RB; ++q)
                // plain loads as in fft_team.h pass1 (+3-5 %; +0.7 % at N = 16384)
                v[RB * i + q] = cmul(v[RB * i + q], twb[(q - 1) * 16 + s]);
        }
    }
    static PSDK_HD void storeB(int tl, const cf *v, cf *frame)
    {
        cf *base = frame + baseB(tl);
#pragma unroll
        for (int i = 0; i < NBB; ++i)
#pragma unroll
            for (int q = 0; q < RB; ++q)
                base[STEPA * i + 17 * q] = v[RB * i + q];
    }

    // pass 3: 16 consecutive elements per thread
    static PSDK_HD void loadC(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + 17 * tl;
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = lds_ld(base + m);
    }
    static PSDK_HD void passC(cf *v) { Dft<16>::run(v); }
};

} // namespace psdk
