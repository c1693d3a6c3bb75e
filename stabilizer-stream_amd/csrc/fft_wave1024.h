// fft_wave1024.h -- 1024-point forward complex FFT by ONE 64-lane wavefront,
// 16 elements per lane, radix (4, 16, 16) decimation in frequency, two
// exchanges through an 8 KB LDS frame private to the wave.
//
// Why this plan: the pass-0 butterflies of lane t work on elements
// n = 4t + c + 256m (c, m = 0..3), which is exactly what four 16-byte loads
// x[256m + 4t .. 4t+3] deliver -- the sample stream goes HBM -> VGPR in
// dwordx4 pieces with no staging copy.  Output bin of register slot q of lane t
// after pass 2:  k = (t >> 4) + 4 (t & 15) + 64 q.
// Only |X|^2 is consumed downstream (src/psd.rs:228-233).
//
// The frame is padded by one element per 16 (physical = idx + idx/16, 1088
// elements per wave).  That makes every 8-byte LDS access of every pass
// bank-conflict free under the gfx950 rules (tests/host/fft_emul.cpp counts
// them) AND keeps every address of a pass of the form lane_base + constant,
// so the constants ride in the DS instruction's offset field.
#pragma once
#include "fft_core.h"

namespace psdk {
namespace w1024 {

constexpr int N = 1024;
constexpr int TW0_SIZE = 3 * 256; // W_1024^(s q), q = 1..3, s < 256
constexpr int TW1_SIZE = 15 * 16; // W_256^(s q),  q = 1..15, s < 16

constexpr int FRAME = N + N / 16; // padded frame, complex elements

PSDK_HD int swz(int idx) { return idx + (idx >> 4); }

PSDK_HD int freq_of(int t, int q) { return (t >> 4) + 4 * (t & 15) + 64 * q; }

// pass 0: v[4m + c] holds z[4t + c + 256m]; afterwards v[4q + c] is output q of
// butterfly s = 4t + c, already multiplied by W_1024^(s q).
// tw0[(q-1)*256 + s] = W_1024^(s q)
PSDK_HD void pass0(int t, cf *v, const cf *tw0)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cf b[4] = {v[c], v[4 + c], v[8 + c], v[12 + c]};
        Dft<4>::run(b);
        const int s = 4 * t + c;
        v[c] = b[0];
        v[4 + c] = cmul(b[1], tw0[0 * 256 + s]);
        v[8 + c] = cmul(b[2], tw0[1 * 256 + s]);
        v[12 + c] = cmul(b[3], tw0[2 * 256 + s]);
    }
}

PSDK_HD void store0(int t, const cf *v, cf *frame)
{
    cf *base = frame + (4 * t + (t >> 2)); // swz(256 q + 4 t + c) = base + 272 q + c
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            base[272 * q + c] = v[4 * q + c];
}

// pass 1: sub-transforms of length 256 (b = t >> 4), stride 16 (s = t & 15)
PSDK_HD void load1(int t, cf *v, const cf *frame)
{
    const cf *base = frame + (272 * (t >> 4) + (t & 15)); // swz(256 b + s + 16 m) = base + 17 m
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = base[17 * m];
}

// tw1[(q-1)*16 + s] = W_256^(s q)
PSDK_HD void pass1(int t, cf *v, const cf *tw1)
{
    Dft<16>::run(v);
    const int s = t & 15;
#pragma unroll
    for (int q = 1; q < 16; ++q)
        v[q] = cmul(v[q], tw1[(q - 1) * 16 + s]);
}

PSDK_HD void store1(int t, const cf *v, cf *frame)
{
    cf *base = frame + (272 * (t >> 4) + (t & 15));
#pragma unroll
    for (int q = 0; q < 16; ++q)
        base[17 * q] = v[q];
}

// pass 2: 16 consecutive elements per lane
PSDK_HD void load2(int t, cf *v, const cf *frame)
{
    const cf *base = frame + 17 * t; // swz(16 t + m) = 17 t + m
#pragma unroll
    for (int m = 0; m < 16; ++m)
        v[m] = base[m];
}

PSDK_HD void pass2(cf *v) { Dft<16>::run(v); }

} // namespace w1024
} // namespace psdk
