// fft_core.h -- fixed-N forward complex FFT for a "team" of N/E threads, E
// elements per thread in registers, radix-<=16 decimation-in-frequency passes
// exchanged in place through an LDS frame.
//
// Replaces the rustfft call of the reference (src/psd.rs:213, plan :418):
// unnormalised forward DFT X[k] = sum_j c[j] exp(-2 pi i jk/N).  Only
// |X[k]|^2 is consumed (src/psd.rs:228-233), so the output is left in
// digit-reversed position and the last pass may rotate its inputs (unit phase).
//
// Everything here is __host__ __device__: tests/host/fft_emul.cpp runs the
// same code lane by lane on the CPU to check the index maps without a GPU.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PSDK_HD __host__ __device__ __forceinline__
#else
#define PSDK_HD inline
#endif

namespace psdk {

struct alignas(8) cf {
    float re, im;
};

// One 8-byte LDS read that the backend may not fuse with a neighbour: on gfx950 a
// ds_read2_b64 costs 8 LDS cycles (128 B/clk, 32-bank rules) where two ds_read_b64 cost 2 + 2
// (256 B/clk, 64 banks) -- MI355X_MICROARCH "LDS" table.  The volatile qualifier is what keeps
// the machine-level load/store optimizer from pairing the reads.
PSDK_HD cf lds_ld(const cf *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v r = *(const volatile __attribute__((address_space(3))) f2v *)p;
    return {r.x, r.y};
#else
    return *p;
#endif
}

// The inter-pass twiddles of a radix-R pass held in an LDS table tw[(q - 1) * 16 + s] (q = 1 .. R - 1): applied to the NB
// butterflies of a lane, v[R i + q] *= tw_q, reading the table as SINGLE 8-byte reads in batches of B that are in flight
// together.  Left to the compiler the reads pair up as ds_read2_b64 (8 LDS cycles for 16 bytes a lane where two ds_read_b64
// take 2 + 2); read one at a time through lds_ld right where they are used they serialise on one register pair.  Batches of
// five measured +1 % at N = 1024 (eight: +1.3 %, but -0.6 % with Mean, whose registers are tighter).
#ifndef PSDK_TW_BATCH
#define PSDK_TW_BATCH 5
#endif
#ifndef PSDK_TW_ROWS // bit 0: team FFT (N = 1024), bit 1: workgroup FFT pass B, bit 2: three-pass workgroup FFT pass 1
#define PSDK_TW_ROWS 1
#endif
template <int R, int NB, int B = PSDK_TW_BATCH>
PSDK_HD void twiddle_rows(cf *v, const cf *tw_s)
{
#pragma unroll
    for (int q0 = 1; q0 < R; q0 += B) {
        cf t[B];
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (q0 + k < R)
                t[k] = lds_ld(tw_s + (q0 + k - 1) * 16);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int k = 0; k < B; ++k)
                if (q0 + k < R)
                    v[R * i + q0 + k] = cmul(v[R * i + q0 + k], t[k]);
    }
}

// Span detrend (src/psd.rs:94-102): slope (x[N-1] - x[0]) / (N - 1) as an unevaluated sum hi + lo.
// A slope rounded to f32 leaves a ramp error of up to D 2^-24 at the end of the segment, coherent over
// the segment: in the lowest bins that is ~1e-5 of the power at N >= 8192 (D = the span, a few sigma).
// The two-term slope follows the exact ramp to ~1e-14; the reference's own f32 ramp (sequential
// offset += slope) does not, the f64 oracle does.
struct slope2 {
    float hi, lo;
};
PSDK_HD slope2 span_slope(float first, float last, int n)
{
    const float dh = last - first; // TwoSum of last + (-first): dh + dl is the exact difference
    const float bp = dh - last;
    const float dl = (last - (dh - bp)) + (-first - bp);
    const float nm1 = (float)(n - 1);
    const float r = 1.0f / nm1; // (n is a constant at the call sites.)  hi need not be the correctly rounded
    slope2 s;                   // quotient: whatever it misses is in the remainder, and so in lo
    s.hi = dh * r;
#if defined(__HIP_DEVICE_COMPILE__)
    const float rem = __fmaf_rn(-s.hi, nm1, dh) + dl;
#else
    const float rem = (float)((double)dh - (double)s.hi * (double)nm1) + dl;
#endif
    s.lo = rem * r;
    return s;
}

PSDK_HD cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
PSDK_HD cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
PSDK_HD cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PSDK_HD cf mul_mi(cf a) { return {a.im, -a.re}; } // a * (-i)

// a * exp(-2 pi i k16/16), k16 compile-time
template <int K16>
PSDK_HD cf mul_w16(cf a)
{
    constexpr int k = ((K16 % 16) + 16) % 16;
    constexpr float r = 0.70710678118654752440f;
    constexpr float c1 = 0.92387953251128675613f; // cos(pi/8)
    constexpr float s1 = 0.38268343236508977173f; // sin(pi/8)
    if constexpr (k == 0) return a;
    else if constexpr (k == 4) return {a.im, -a.re};
    else if constexpr (k == 8) return {-a.re, -a.im};
    else if constexpr (k == 12) return {-a.im, a.re};
    else if constexpr (k == 2) return {(a.re + a.im) * r, (a.im - a.re) * r};
    else if constexpr (k == 6) return {(a.im - a.re) * r, -(a.re + a.im) * r};
    else if constexpr (k == 10) return {-(a.re + a.im) * r, (a.re - a.im) * r};
    else if constexpr (k == 14) return {(a.re - a.im) * r, (a.re + a.im) * r};
    else {
        // w = cos(k pi/8) - i sin(k pi/8)
        constexpr float wr = (k == 1 || k == 15) ? c1 : (k == 3 || k == 13) ? s1
                           : (k == 5 || k == 11) ? -s1 : -c1;
        constexpr float wi = (k == 1 || k == 7) ? -s1 : (k == 3 || k == 5) ? -c1
                           : (k == 9 || k == 15) ? s1 : c1;
        return {a.re * wr - a.im * wi, a.re * wi + a.im * wr};
    }
}

// In-register DFT of R points, natural-order output.
template <int R>
struct Dft;

template <>
struct Dft<1> {
    static PSDK_HD void run(cf *) {}
};

template <>
struct Dft<2> {
    static PSDK_HD void run(cf *v)
    {
        cf a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <>
struct Dft<4> {
    static PSDK_HD void run(cf *v)
    {
        cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        cf t2 = cadd(v[1], v[3]), t3 = mul_mi(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3);
        v[3] = csub(t1, t3);
    }
};

// R = 4 * B (B = 2 or 4): radix-4 over the high index, constant twiddles
// W_R^(m0 q1), then radix-B over the low index; output q = q1 + 4 q0.
template <int R>
struct Dft {
    static_assert(R == 8 || R == 16, "radix");
    static constexpr int A = 4, B = R / 4;

    template <int M0, int Q1>
    static PSDK_HD cf tw(cf a)
    {
        return mul_w16<M0 * Q1 * (16 / R)>(a);
    }

    template <int M0>
    static PSDK_HD void col(const cf *v, cf *y)
    {
        cf t[A];
#pragma unroll
        for (int m1 = 0; m1 < A; ++m1)
            t[m1] = v[m1 * B + M0];
        Dft<A>::run(t);
        y[0 * B + M0] = t[0];
        y[1 * B + M0] = tw<M0, 1>(t[1]);
        y[2 * B + M0] = tw<M0, 2>(t[2]);
        y[3 * B + M0] = tw<M0, 3>(t[3]);
    }

    static PSDK_HD void run(cf *v)
    {
        cf y[R];
        col<0>(v, y);
        col<1>(v, y);
        if constexpr (B == 4) {
            col<2>(v, y);
            col<3>(v, y);
        }
#pragma unroll
        for (int q1 = 0; q1 < A; ++q1) {
            cf t[B];
#pragma unroll
            for (int m0 = 0; m0 < B; ++m0)
                t[m0] = y[q1 * B + m0];
            Dft<B>::run(t);
#pragma unroll
            for (int q0 = 0; q0 < B; ++q0)
                v[q1 + A * q0] = t[q0];
        }
    }
};

// Radix plan: E elements per thread, passes of radix min(E, remaining).
template <int N>
struct FftPlan {
    static_assert(N >= 16 && (N & (N - 1)) == 0, "N must be a power of two >= 16");
#ifndef PSDK_E16_MIN
#define PSDK_E16_MIN 64
#endif
    // sixteen elements a thread from N = 64 up: N = 128 two passes (16, 8) instead of four of radix 4 (176 -> 197 GS/s in round 3),
    // N = 64 two passes (16, 4) instead of three (234 -> 295 GS/s once the kernel was all-scalar; with packed ops it had read 5 % lower);
    // N = 32 / 16 with sixteen a thread are teams of 2 / 1 lanes whose loads no longer coalesce: 231 -> 157, 259 -> 178 GS/s
    static constexpr int E = N >= PSDK_E16_MIN ? 16 : 4;
    static constexpr int TEAM = N / E;

    static constexpr int len(int p) // sub-transform length entering pass p
    {
        int rem = N;
        for (int i = 0; i < p; ++i)
            rem /= (rem < E ? rem : E);
        return rem;
    }
    static constexpr int radix(int p) { return len(p) < E ? len(p) : E; }
    static constexpr int npass()
    {
        int p = 0;
        while (len(p) > 1)
            ++p;
        return p;
    }
    static constexpr int NPASS = npass();
};

template <int N, int P>
struct PassInfo {
    using Plan = FftPlan<N>;
    static constexpr int E = Plan::E;
    static constexpr int TEAM = Plan::TEAM;
    static constexpr int L = Plan::len(P);
    static constexpr int R = Plan::radix(P);
    static constexpr int S = L / R;  // stride between the R inputs of a butterfly
    static constexpr int NB = E / R; // butterflies per thread
    static constexpr bool LAST = (P == Plan::NPASS - 1);

    // natural element index held in register slot (i, m) of team-thread t
    static PSDK_HD int elem(int t, int i, int m)
    {
        const int u = t + i * TEAM;
        const int b = u / S, s = u % S;
        return b * L + s + m * S;
    }
};

// LDS frame swizzle (index in complex elements).  Chosen per N so that the
// b64 accesses of every pass are bank-conflict free (tests/host/fft_emul.cpp
// counts conflicts with the gfx950 banking rules).
template <int N>
PSDK_HD int lds_swz(int idx)
{
#ifndef PSDK_SWZ_MAX
#define PSDK_SWZ_MAX 4096
#endif
    if constexpr (N >= 256 && N <= PSDK_SWZ_MAX)
        return idx ^ ((idx >> 4) & 0x1F); // (512 ... 4096 since round 3: the unswizzled frame read at up to 9x the ideal LDS cycles; N <= 256: LdsFrame)
    else
        return idx;
}

// Where element e of a team's frame sits in LDS.  N > 256: the XOR swizzle above, frames of N elements.  N <= 256: a wavefront holds
// 64 / TEAM teams (4 ... 16) whose frames are 1 ... 2 KiB apart, i.e. on the SAME banks -- conflict-free per team is 4- to 8-fold
// conflicts per wavefront (N = 128 read 256 cycles for an ideal 32: `welch_kernel<128>` was LDS-bound, SQ_LDS_BANK_CONFLICT 0.62 of
// its LDS cycles).  One element of padding per 2^K (e + (e >> K)) and frames of N + (N >> K) + PAD elements put every read
// instruction of a wavefront at its ideal cycle count and every write at the ideal (N >= 128) or 1.5x of it (tests/host/fft_emul.cpp
// counts them per WAVEFRONT now; K and PAD found by exhaustive search over that model).
template <int N>
struct LdsFrame {
    static constexpr bool PADDED = N <= 256;
    static constexpr bool E16 = FftPlan<N>::E == 16;
    static constexpr int K = (N == 32 && E16) ? 1 : N <= 64 ? 2 : N == 128 ? 3 : 4;
    static constexpr int PAD = N == 128 ? 8 : (N == 64 && E16) ? 4 : (N == 32 && E16) ? 2 : 0;
    static constexpr int SIZE = PADDED ? N + (N >> K) + PAD : N; // elements of one team's frame
    static PSDK_HD int at(int e)
    {
        if constexpr (PADDED)
            return e + (e >> K);
        else
            return lds_swz<N>(e);
    }
};

// Frequency bin held at natural position `pos` after all DIF passes:
// pos = sum_p d_p * N/(R_0..R_p)  ->  k = sum_p d_p * (R_0..R_{p-1}).
template <int N>
PSDK_HD int freq_of_pos(int pos)
{
    using Plan = FftPlan<N>;
    int k = 0, mul = 1, rem = N;
#pragma unroll
    for (int p = 0; p < Plan::NPASS; ++p) {
        const int r = Plan::radix(p);
        rem /= r;
        const int d = (pos / rem) % r;
        k += d * mul;
        mul *= r;
    }
    return k;
}

// butterflies + inter-pass twiddles of pass P on the thread's registers.
// tw: table of W_N^j = exp(-2 pi i j/N), j < N.
template <int N, int P>
PSDK_HD void pass_compute(int t, cf *v, const cf *tw)
{
    using PI = PassInfo<N, P>;
#pragma unroll
    for (int i = 0; i < PI::NB; ++i) {
        Dft<PI::R>::run(v + i * PI::R);
        if constexpr (PI::S > 1) {
            const int u = t + i * PI::TEAM;
            const int s = u % PI::S;
#pragma unroll
            for (int q = 1; q < PI::R; ++q)
                v[i * PI::R + q] = cmul(v[i * PI::R + q], tw[(s * q) * (N / PI::L)]);
        }
    }
}

// write the outputs of pass P in place into the team's LDS frame
template <int N, int P>
PSDK_HD void pass_store(int t, const cf *v, cf *frame)
{
    using PI = PassInfo<N, P>;
#pragma unroll
    for (int i = 0; i < PI::NB; ++i)
#pragma unroll
        for (int q = 0; q < PI::R; ++q)
            frame[LdsFrame<N>::at(PI::elem(t, i, q))] = v[i * PI::R + q];
}

// read the inputs of pass P from the team's LDS frame.  In the LAST pass the
// inputs may be rotated by `rot` positions within the butterfly: the outputs
// then differ by a unit phase only, which |X|^2 does not see.
template <int N, int P>
PSDK_HD void pass_load(int t, cf *v, const cf *frame, int rot = 0)
{
    using PI = PassInfo<N, P>;
#pragma unroll
    for (int i = 0; i < PI::NB; ++i)
#pragma unroll
        for (int m = 0; m < PI::R; ++m) {
            const int mm = PI::LAST ? ((m + rot) % PI::R) : m;
            v[i * PI::R + m] = frame[LdsFrame<N>::at(PI::elem(t, i, mm))];
        }
}

// frequency bin of register slot `slot` of team-thread t after the last pass
template <int N>
PSDK_HD int freq_of_slot(int t, int slot)
{
    using PI = PassInfo<N, FftPlan<N>::NPASS - 1>;
    const int i = slot / PI::R, q = slot % PI::R;
    return freq_of_pos<N>(PI::elem(t, i, q));
}

} // namespace psdk
