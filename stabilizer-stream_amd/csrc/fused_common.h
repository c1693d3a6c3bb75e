// fused_common.h -- pieces shared by the team-level (fused.hip) and workgroup-level
// (bigfused_impl.h) fused kernels: the stateful /8 half-band decimator geometry and its
// two-outputs-per-lane evaluation, EWMA amplitude.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_core.h"
#include "hbf_taps.h"
#include "kernels.h"

#ifndef PSDK_HBF_WIDE
#define PSDK_HBF_WIDE 1 // stage B's shared inputs as 16-byte LDS reads (hbf_four)
#endif
#ifndef PSDK_XO_PAD
#define PSDK_XO_PAD 0
#endif
#ifndef PSDK_HBF_WIDE_A
#define PSDK_HBF_WIDE_A 0
#endif
#ifndef PSDK_DEC_PRIO
#define PSDK_DEC_PRIO 3 // wave priority during the decimator stages (0 during the FFT)
#endif

namespace psdk {

// Decimator arrays inside a team's LDS frame (floats): [history | new], even/odd polyphase,
// even sizes.  One pair = N new samples -> N/2 stage-A, N/4 stage-B, N/8 stage-C outputs.
// Carried state: the last HA stage-A and HB stage-B outputs (the 12 input samples before the
// new ones come from registers).  History sizes are chosen so that the polyphase offsets are
// the same as in a stateless block with a 288-sample halo, which is what the warm-up at the
// start of a run evaluates.
// NOX: no polyphase sample arrays (stage A reads the stream from registers): the A and B arrays start at 0, and at
// N = 16384 every decimator address then fits the 16-bit DS offset field -- with the sample arrays in front, the A and B
// arrays sat above 64 KiB, each access needed a base register of its own and the compiler spilled thirteen of them,
// reloaded (vmcnt(0)!) inside every pair.
template <int N, bool NOX = false>
struct FusedDec {
    static constexpr int HX = 12, HA = 22, HB = 58;
    // (PSDK_XO_PAD floats between the even and the odd sample array: with 10 the two arrays sit 16 banks apart mod 32 -- the
    // polyphase split writes lanes 2k / 2k + 1 to XE[k] / XO[k], which collide on 10 of 16 banks at the natural distance of 6 --
    // and XO becomes 16-byte aligned, so stage A can read its shared inputs as ds_read_b128 like stage B: PSDK_HBF_WIDE_A)
    static constexpr int XE = 0, XO = XE + HX / 2 + N / 2 + (N >= 2048 ? PSDK_XO_PAD : 0); // (the team frames have no room for it)
    static constexpr int AE = NOX ? 0 : XO + HX / 2 + N / 2, AO = AE + 12 + N / 4;
    static constexpr int BE = AO + 12 + N / 4, BO = BE + 30 + N / 8;
    static constexpr int END = BO + 30 + N / 8;
    static constexpr int HIST = HA + HB; // [0,11) AE, [11,22) AO, [22,51) BE, [51,80) BO
    // warm-up block: WX samples before the first new one
    static constexpr int WX = HBF_HALO, WA = HBF_PRE_A, WB = HBF_PRE_B; // 288, 138, 58
    static constexpr int WXE = 0, WXO = WX / 2, WAE = WX, WAO = WX + WA / 2 + 1;
    static constexpr int WEND = WAO + WA / 2 + 1;
    // polyphase offsets: out j = ev[j + CE] + sum_i t[i] (od[j + CO + i] + od[j + CO + 2M-1-i])
    static constexpr int A_CE = 4, A_CO = 1, B_CE = 6, B_CO = 0, C_CE = 15, C_CO = 0;
    static_assert(HX / 2 - 2 == A_CE && HX / 2 - 5 == A_CO, "x history vs stage A offsets");
    static_assert(HA / 2 - 5 == B_CE && HA / 2 - 11 == B_CO, "A history vs stage B offsets");
    static_assert(HB / 2 - 14 == C_CE && HB / 2 - 29 == C_CO, "B history vs stage C offsets");
    static_assert(WX / 2 - WA - HBF_MA + 1 == A_CE && WA / 2 - WB - HBF_MB + 1 == B_CE, "warm-up geometry");

    // frame offsets of carried element i < HIST: where it is read from after a pair (tail) and
    // where it must sit before the next one (front); packed tail << 16 | front
    static __device__ __forceinline__ unsigned hist_slot(int i)
    {
        int front = -1, shift = 0;
        if (i < 11) {
            front = AE + i;
            shift = N / 4;
        } else if (i < 22) {
            front = AO + (i - 11);
            shift = N / 4;
        } else if (i < 51) {
            front = BE + (i - 22);
            shift = N / 8;
        } else if (i < 80) {
            front = BO + (i - 51);
            shift = N / 8;
        }
        return front < 0 ? 0xFFFFu : ((unsigned)(front + shift) << 16) | (unsigned)front;
    }
};

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wavefront execute in order; this only stops the
    // compiler from moving LDS accesses across the hand-off between lanes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP wavefront shifts (gfx9 family): dpp_shr1(fill, v)[l] = v[l - 1] for l >= 1 and fill[0] for
// l = 0; dpp_ror1(v)[0] = v[63].  Together: a shift that continues into another register's top lane
// (semantics checked on the device by tools/probes/dpp_shift.cpp).
__device__ __forceinline__ float dpp_shr1(float fill, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_ror1(float v)
{
    // every lane is written: no "old" operand (update_dpp with a 0 costs a v_mov_b32 to materialise it)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}

// Sum over a row of 16 lanes with DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: plain
// VALU), every lane of the row receiving the row's sum; and over the wavefront (the four row sums
// through scalar reads).  A __shfl_xor butterfly would be six dependent ds_bpermute round trips.
__device__ __forceinline__ float row_sum16(float v)
{
#define PSDK_DPP(x, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))
    v += PSDK_DPP(v, 0xB1);  // quad_perm [1,0,3,2]
    v += PSDK_DPP(v, 0x4E);  // quad_perm [2,3,0,1]
    v += PSDK_DPP(v, 0x141); // row_half_mirror
    v += PSDK_DPP(v, 0x140); // row_mirror
#undef PSDK_DPP
    return v;
}
// Sum over aligned groups of W lanes (W = 2, 4, 8, 16), every lane of a group receiving the group's sum (the first
// log2 W steps of row_sum16).
template <int W>
__device__ __forceinline__ float group_sum(float v)
{
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "group of lanes within a row");
#define PSDK_DPP(x, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))
    if constexpr (W >= 2)
        v += PSDK_DPP(v, 0xB1); // quad_perm [1,0,3,2]
    if constexpr (W >= 4)
        v += PSDK_DPP(v, 0x4E); // quad_perm [2,3,0,1]
    if constexpr (W >= 8)
        v += PSDK_DPP(v, 0x141); // row_half_mirror
    if constexpr (W >= 16)
        v += PSDK_DPP(v, 0x140); // row_mirror
#undef PSDK_DPP
    return v;
}
__device__ __forceinline__ float wave_sum64(float v)
{
    const int b = __builtin_bit_cast(int, row_sum16(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

struct f2 {
    float x, y;
};

// A lane's register group of four consecutive stream samples.  In the kernels that read AdcDac frames in place a group holds,
// between its load and its conversion (volts()), the two raw 32-bit wire words of the four samples -- and it holds them as
// INTEGERS: the FRAMES kernels carry their groups in integer registers and view them as f32 where the samples are used (f()),
// so that no f32-typed value ever contains a wire word (f32 jobs of such a launch store their samples' bits).  Round 3 carried the
// words bit-cast into float4 components; the wrong spectra that round met turned out to be a register-allocation fault of the
// compiler, not this, but an integer payload in a floating-point-typed value is a standing invitation to
// any transform that reasons about f32 semantics.  The f32-only kernels keep plain float4 groups: same code as before.
template <bool FRAMES>
struct Grp4;
template <>
struct Grp4<false> {
    float4 v;
    __device__ __forceinline__ float4 f() const { return v; }
    __device__ __forceinline__ void set(const float4 &x) { v = x; }
};
template <>
struct Grp4<true> {
    unsigned x, y, z, w;
    __device__ __forceinline__ float4 f() const
    {
        return make_float4(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y), __builtin_bit_cast(float, z),
                           __builtin_bit_cast(float, w));
    }
    __device__ __forceinline__ void set(const float4 &v)
    {
        x = __builtin_bit_cast(unsigned, v.x), y = __builtin_bit_cast(unsigned, v.y);
        z = __builtin_bit_cast(unsigned, v.z), w = __builtin_bit_cast(unsigned, v.w);
    }
    __device__ __forceinline__ void set_raw(unsigned a, unsigned b) { x = a, y = b, z = 0u, w = 0u; } // four wire words
};
// wire words -> volts in place (src/de/data.rs:28-35, :64, :75): i16 (the DAC words offset binary: flip = 0x80008000) x LSB
__device__ __forceinline__ void grp_volts(Grp4<true> &g, unsigned flip, float lsb)
{
    const unsigned a = g.x ^ flip, b = g.y ^ flip;
    g.set(make_float4((float)(short)(unsigned short)(a & 0xffffu) * lsb, (float)(short)(unsigned short)(a >> 16) * lsb,
                      (float)(short)(unsigned short)(b & 0xffffu) * lsb, (float)(short)(unsigned short)(b >> 16) * lsb));
}
// aligned 8-byte LDS read, kept a single ds_read_b64 (see lds_ld in fft_core.h)
__device__ __forceinline__ f2 ld2(const float *p)
{
    const cf v = lds_ld(reinterpret_cast<const cf *>(p));
    return {v.re, v.im};
}

// two consecutive outputs (j, j+1), j even, of a half-band stage with M unique taps:
// out j = ev[j + CE] + sum_i taps[i] (od[j + CO + i] + od[j + CO + 2M-1-i]); odd array read
// as aligned pairs.
template <int M, int CE, int CO>
__device__ __forceinline__ void hbf_two(const float *__restrict__ ev, const float *__restrict__ od, int j,
                                        const float (&taps)[M], float &y0, float &y1)
{
    constexpr int LO = CO & ~1;                // aligned start
    constexpr int CNT = (CO - LO) + 2 * M + 1; // values needed from od[j + LO]
    constexpr int NP = (CNT + 1) / 2;
    float w[2 * NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const f2 v = ld2(od + j + LO + 2 * k);
        w[2 * k] = v.x;
        w[2 * k + 1] = v.y;
    }
    constexpr int O = CO - LO;
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        a0 += (w[O + i] + w[O + 2 * M - 1 - i]) * taps[i];
        a1 += (w[O + 1 + i] + w[O + 2 * M - i]) * taps[i];
    }
    float e0, e1;
    if constexpr ((CE & 1) == 0) {
        const f2 e = ld2(ev + j + CE);
        e0 = e.x;
        e1 = e.y;
    } else {
        e0 = ev[j + CE];
        e1 = ev[j + CE + 1];
    }
    y0 = e0 + a0;
    y1 = e1 + a1;
}

// four consecutive outputs (j .. j+3), j even: the same sums in the same order as two hbf_two calls, but the 2M + 3 inputs the
// four share are read once -- M + 2 (+1) eight-byte reads for four outputs where two calls make 2 (M + 1).  The decimator
// stages are the LDS-heavier half of a pair (more read instructions than the FFT's two exchanges).
// WIDE: od + j + (CO & ~3) is 16-byte aligned (j a multiple of 4, the array's offset in the frame too): the shared inputs
// are read as ds_read_b128 -- lanes 16 bytes apart are conflict-free for 16-byte reads (4 x 16 lanes), while 8-byte reads at
// that stride are 2-way bank conflicts (2 x 32 lanes over 128 dwords), which gives back what the shared reads save.
template <int M, int CE, int CO, bool WIDE = false>
__device__ __forceinline__ void hbf_four(const float *__restrict__ ev, const float *__restrict__ od, int j,
                                         const float (&taps)[M], float (&y)[4])
{
    constexpr int LO = WIDE ? (CO & ~3) : (CO & ~1);
    constexpr int CNT = (CO - LO) + 2 * M + 3;
    constexpr int NP = WIDE ? 2 * ((CNT + 3) / 4) : (CNT + 1) / 2;
    float w[2 * NP];
    if constexpr (WIDE) {
#pragma unroll
        for (int k = 0; k < NP / 2; ++k) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v v = *(const volatile __attribute__((address_space(3))) f4v *)(od + j + LO + 4 * k);
            w[4 * k] = v.x, w[4 * k + 1] = v.y, w[4 * k + 2] = v.z, w[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const f2 v = ld2(od + j + LO + 2 * k);
            w[2 * k] = v.x;
            w[2 * k + 1] = v.y;
        }
    }
    constexpr int O = CO - LO;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            a[t] += (w[O + t + i] + w[O + t + 2 * M - 1 - i]) * taps[i];
    float e[4];
    if constexpr ((CE & 1) == 0) {
        const f2 e01 = ld2(ev + j + CE), e23 = ld2(ev + j + CE + 2);
        e[0] = e01.x, e[1] = e01.y, e[2] = e23.x, e[3] = e23.y;
    } else {
        const f2 e12 = ld2(ev + j + CE + 1);
        e[0] = ev[j + CE], e[1] = e12.x, e[2] = e12.y, e[3] = ev[j + CE + 3];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        y[t] = e[t] + a[t];
}

__device__ __forceinline__ float fused_ewma_amp(const FusedJob &job, int step)
{
    // sqrt of W_step = gamma^max(0, nb - max(step, i_s - 1))  (plan.h)
    const int m = step > job.is_m1 ? step : job.is_m1;
    const int na = job.nb - m;
    if (na <= 0)
        return 1.0f;
    return (float)exp2(0.5 * (double)na * job.log2_gamma);
}

// The same amplitudes for consecutive steps of a run: one exp2 at the start, then a double multiplication per
// step (the exponent drops by one per step between i_s - 1 and nb) instead of a double exp2 per segment.
struct EwmaAmp {
    double a = 1.0, rho = 1.0;
    int step = 0;
    bool zero = false; // gamma == 0 (avg == 0): weights are 0 or 1
    __device__ __forceinline__ void init(const FusedJob &job, int s)
    {
        step = s;
        zero = !(job.log2_gamma > -1.0e300);
        rho = zero ? 0.0 : exp2(-0.5 * job.log2_gamma);
        const int m = s > job.is_m1 ? s : job.is_m1;
        const int na = job.nb - m;
        a = na <= 0 ? 1.0 : (zero ? 0.0 : exp2(0.5 * (double)na * job.log2_gamma));
    }
    __device__ __forceinline__ float next(const FusedJob &job)
    {
        const float r = (float)a;
        if (step >= job.is_m1 && step < job.nb) {
            if (step + 1 >= job.nb)
                a = 1.0;
            else if (!zero)
                a *= rho;
        }
        ++step;
        return r;
    }
};

// Detrend + window + EWMA amplitude of one segment pair into the 16 FFT inputs of a lane
// (src/psd.rs:75-113, :211): lane tl holds samples 4 tl + c + (N/4) m of segment a in (lo0, lo1, up0,
// up1)[m].c and of segment b in (up0, up1, nl0, nl1)[m].c; slot 4m + c gets
// {(xa - trend_a) w, (xb - trend_b) w}.  The trend is removed as (x - o) - (m + n s): see DESIGN.md.
struct DetrendParams {
    float oa = 0.0f, ob = 0.0f, ma = 0.0f, mb = 0.0f;
    slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
    float ea = 1.0f, eb = 1.0f; // EWMA amplitudes
};
// CENTRED (Mean only): lo and up arrive with the pivot d.ob already subtracted (they are dead afterwards and
// were rewritten in place), nl is raw.
// SINGLE (overlap 0): only segment b = (up, nl) is transformed -- exactly the N samples the step decimates, so a run's segments
// and its decimated samples are the same stretch of the stream; the imaginary part is an exact zero (no other segment's samples
// enter: an infinity next door must not reach this spectrum), |Z[k]|^2 = |Z[N-k]|^2 = |X_b[k]|^2 and the fold
// 1/2 (Q[k] + Q[N-k]) of post_kernel yields the segment's power unchanged.  (lo is read for the decimator's history only.)
template <int N, int DETREND, bool EWMA, bool CENTRED = false, bool SINGLE = false>
__device__ __forceinline__ void window_pair(cf (&v)[16], int tl, const float4 &lo0, const float4 &lo1, const float4 &up0,
                                            const float4 &up1, const float4 &nl0, const float4 &nl1, const float4 &w0,
                                            const float4 &w1, const float4 &w2, const float4 &w3, const DetrendParams &d)
{
    const float nf = (float)(4 * tl);
    auto put = [&](int slot, float xa, float xb, float w, int nofs) {
        if constexpr (DETREND == 1) {
            xa -= d.oa;
            xb -= d.ob;
        } else if constexpr (DETREND == 2) {
            const float n = nf + (float)nofs;
            xa = fmaf(-n, d.sa.lo, fmaf(-n, d.sa.hi, xa - d.oa));
            xb = fmaf(-n, d.sb.lo, fmaf(-n, d.sb.hi, xb - d.ob));
        } else if constexpr (DETREND == 3 && CENTRED) {
            xa -= d.ma;
            xb = slot < 8 ? xb - d.mb : (xb - d.ob) - d.mb;
        } else if constexpr (DETREND == 3) {
            xa = (xa - d.oa) - d.ma;
            xb = (xb - d.ob) - d.mb;
        }
        xa *= w;
        xb *= w;
        if constexpr (EWMA) {
            xa *= d.ea;
            xb *= d.eb;
        }
        if constexpr (SINGLE)
            v[slot] = {xb, 0.0f};
        else
            v[slot] = {xa, xb};
    };
    put(0, lo0.x, up0.x, w0.x, 0);
    put(1, lo0.y, up0.y, w0.y, 1);
    put(2, lo0.z, up0.z, w0.z, 2);
    put(3, lo0.w, up0.w, w0.w, 3);
    put(4, lo1.x, up1.x, w1.x, N / 4);
    put(5, lo1.y, up1.y, w1.y, N / 4 + 1);
    put(6, lo1.z, up1.z, w1.z, N / 4 + 2);
    put(7, lo1.w, up1.w, w1.w, N / 4 + 3);
    put(8, up0.x, nl0.x, w2.x, N / 2);
    put(9, up0.y, nl0.y, w2.y, N / 2 + 1);
    put(10, up0.z, nl0.z, w2.z, N / 2 + 2);
    put(11, up0.w, nl0.w, w2.w, N / 2 + 3);
    put(12, up1.x, nl1.x, w3.x, 3 * N / 4);
    put(13, up1.y, nl1.y, w3.y, 3 * N / 4 + 1);
    put(14, up1.z, nl1.z, w3.z, 3 * N / 4 + 2);
    put(15, up1.w, nl1.w, w3.w, 3 * N / 4 + 3);
}

} // namespace psdk
