// fft_block3.h -- N-point forward complex FFT by a workgroup of N/16 threads in THREE passes, radix (16, N/256, 16), for
// N = 2048 and 4096 (16 * 8 * 16 and 16 * 16 * 16): one LDS exchange and two workgroup barriers fewer per transform than the
// four passes of fft_block.h (4, RA, RB, 16).  Measured bound of what that buys (timing-only ablation of fft_block.h's pass B
// exchange): +8 %.
//
// What makes three passes possible is the layout of the INPUT: thread tl holds elements tl + (N/16) m, m = 0 ... 15 -- a
// radix-16 butterfly over m needs no neighbour -- which is what sixteen coalesced dword loads deliver (the four-pass plan's
// radix-4 first pass matches 16-byte loads instead).  After pass 0 (twiddles W_N^(tl q)) block q (length L1 = N/16) holds one
// sub-transform with element s = tl in thread tl; passes 1 and 2 are the last two passes of fft_team.h one size class up:
// radix R1 = L1/16 over stride 16 with the twiddles W_L1^(s q) from a small LDS table, then 16 consecutive elements per thread.
// After the last pass register slot q of thread tl holds bin  k = q0 + 16 q1 + 16 R1 q,  q0 = tl / R1, q1 = tl % R1.
// Frame: physical = idx + idx/16, as everywhere.
#pragma once
#include "fft_core.h"

namespace psdk {

template <int N>
struct BlockFft3 {
    static_assert(N == 2048 || N == 4096, "three-pass block FFT sizes");
    static constexpr int TEAM = N / 16;
    static constexpr int L1 = N / 16;              // sub-transform length after pass 0 (= TEAM)
    static constexpr int R1 = L1 / 16;             // pass-1 radix (8, 16)
    static constexpr int NB1 = 16 / R1;            // pass-1 butterflies per thread
    static constexpr int FRAME = N + N / 16;
    static constexpr int STEP1 = L1 + L1 / 16;     // physical distance between blocks
    static constexpr int TW0_SIZE = 2 * TEAM;      // seeds [2][tl]: W_N^tl, W_N^(4 tl)
    static constexpr int TW1_SIZE = (R1 - 1) * 16; // W_L1^(s q), [(q-1)][s]

    static PSDK_HD int swz(int idx) { return idx + (idx >> 4); }
    static PSDK_HD int freq_of(int tl, int q) { return tl / R1 + 16 * (tl % R1) + 16 * R1 * q; }

    struct Seeds {
        cf w1, w4; // W_N^tl, W_N^(4 tl)
    };
    static PSDK_HD Seeds load_seeds(int tl, const cf *tw0) { return {tw0[tl], tw0[TEAM + tl]}; }

    // pass 0: v[m] = z[tl + (N/16) m]  ->  v[q] = (output q of the butterfly) * W_N^(tl q); the fifteen twiddles from the two
    // seeds: W^(4k + j) = W^(4k) W^j, at most three factors deep
    static PSDK_HD void pass0(cf *v, const Seeds &sd)
    {
        Dft<16>::run(v);
        const cf w1 = sd.w1, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        cf wb = sd.w4; // W^(4 k tl), k = 1, 2, 3
        v[1] = cmul(v[1], w1);
        v[2] = cmul(v[2], w2);
        v[3] = cmul(v[3], w3);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            v[4 * k] = cmul(v[4 * k], wb);
            v[4 * k + 1] = cmul(v[4 * k + 1], cmul(wb, w1));
            v[4 * k + 2] = cmul(v[4 * k + 2], cmul(wb, w2));
            v[4 * k + 3] = cmul(v[4 * k + 3], cmul(wb, w3));
            if (k < 3)
                wb = cmul(wb, sd.w4);
        }
    }
    static PSDK_HD void store0(int tl, const cf *v, cf *frame)
    {
        cf *base = frame + (tl + (tl >> 4)); // swz(q L1 + tl) = q STEP1 + tl + tl/16
#pragma unroll
        for (int q = 0; q < 16; ++q)
            base[STEP1 * q] = v[q];
    }

    // pass 1 (fft_team.h's, with sixteen blocks): butterfly i of the thread works in block b = NB1 (tl / 16) + i with s = tl % 16:
    // elements b L1 + s + 16 m, m < R1; register slots v[R1 i + m]
    static PSDK_HD int base1(int tl) { return STEP1 * NB1 * (tl >> 4) + (tl & 15); } // + STEP1 i + 17 m
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
        if constexpr (PSDK_TW_ROWS & 4) {
#pragma unroll
            for (int i = 0; i < NB1; ++i)
                Dft<R1>::run(v + R1 * i);
            twiddle_rows<R1, NB1>(v, tw1 + s);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NB1; ++i) {
            Dft<R1>::run(v + R1 * i);
#pragma unroll
            for (int q = 1; q < R1; ++q)
                v[R1 * i + q] = cmul(v[R1 * i + q], tw1[(q - 1) * 16 + s]); // (plain loads: see fft_team.h pass1)
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

    // pass 2: 16 consecutive elements per thread
    static PSDK_HD void load2(int tl, cf *v, const cf *frame)
    {
        const cf *base = frame + 17 * tl;
#pragma unroll
        for (int m = 0; m < 16; ++m)
            v[m] = lds_ld(base + m);
    }
    static PSDK_HD void pass2(cf *v) { Dft<16>::run(v); }
};

} // namespace psdk
