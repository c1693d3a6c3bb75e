// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the cascaded PSD hot path.
//
//   welch_kernel<N>   detrend + window (src/psd.rs:75-113), forward FFT
//                     (src/psd.rs:213) and |X|^2 accumulation (src/psd.rs:228-233)
//                     for a tile of 50%-overlapped segments; two real segments
//                     ride one complex FFT (re = segment j, im = segment j+1).
//   hbf_dec8_kernel   the /8 half-band cascade (src/psd.rs:246-253) on blocks
//                     with a recomputed halo, drain applied on store (:255-260).
//   post_kernel       folds the per-workgroup power partials into the stage
//                     accumulators with the batch EWMA factor (src/psd.rs:218-233)
//                     and carries the stream tails, one launch per round.
//   fill/adcdac       synthetic noise, AdcDac payload decode (src/de/data.rs:11-82).
//
// No MFMA: the path is FP32-VALU / LDS bound.  64-wide
// wavefronts throughout; LDS frames are exchanged with 8-byte accesses.
#include "kernels.h"
#include "frames.h"
#include "hbf_taps.h"

namespace psdk {

// ---------------------------------------------------------------------------
// Welch kernel
// ---------------------------------------------------------------------------

template <int N>
struct WelchCfg {
    using Plan = FftPlan<N>;
    static constexpr int E = Plan::E;
    static constexpr int TEAM = Plan::TEAM;                 // threads per FFT
    static constexpr int BLOCK = TEAM > 256 ? TEAM : 256;   // threads per workgroup
    static constexpr int TEAMS = BLOCK / TEAM;              // concurrent FFTs per workgroup
    static constexpr int SPT = (2 * TEAMS * 4 > 32) ? 2 * TEAMS * 4 : 32; // segments per tile
    static constexpr int WAVES = BLOCK / 64;
};

__device__ __forceinline__ float ewma_amp(const SegJob &job, int step)
{
    // sqrt of W_step = gamma^max(0, nb - max(step, i_s - 1))  (plan.h)
    const int m = step > job.is_m1 ? step : job.is_m1;
    const int na = job.nb - m;
    if (na <= 0)
        return 1.0f;
    return (float)exp2(0.5 * (double)na * job.log2_gamma); // log2_gamma = -inf -> 0
}

// POWER_ONLY: only |X|^2 of the outputs is consumed, so the inputs of the last pass may be read rotated within their
// butterflies (pass_load: the outputs pick up unit phases) -- fewer LDS bank conflicts at the sizes listed, on top of the
// frame's XOR swizzle (fft_core.h lds_swz; tests/host/fft_emul.cpp, FFT_EMUL_ROT_ALL=1: read cycles 96 -> 64 at N = 1024,
// 192 -> 176 at 2048, 2048 -> 1536 at 16384; the other sizes are at or near their ideal without it).  The first transform of the
// chirp-z kernel needs the outputs themselves and must not rotate: it did at M = 1024 (sizes 256 < N <= 512 that are not
// powers of two read wrong spectra until round 3's last day; tests/test_gpu_any_n.py now covers every transform length).
// A team's frame is private to the team.  Teams of at most 64 lanes sit inside ONE wavefront, whose LDS operations execute in
// order: the hand-off between the lanes of a team then needs no workgroup barrier, only the compiler kept from moving LDS
// accesses across it (as in fused_common.h wave_sync).  Larger teams span wavefronts and keep __syncthreads().
template <int TEAM>
__device__ __forceinline__ void team_sync()
{
    if constexpr (TEAM <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int N, int P, bool POWER_ONLY>
__device__ __forceinline__ void fft_passes(int t, cf *v, cf *frame, const cf *__restrict__ tw)
{
    using PI = PassInfo<N, P>;
    if constexpr (P > 0) {
        int rot = 0;
        if constexpr (POWER_ONLY && PI::LAST && (N == 1024 || N == 2048 || N == 16384))
            rot = (t >> 3) & (PI::R - 1); // fewer bank conflicts in the last-pass reads
        pass_load<N, P>(t, v, frame, rot);
    }
    pass_compute<N, P>(t, v, tw);
    if constexpr (!PI::LAST) {
        if constexpr (P == 0)
            team_sync<PI::TEAM>(); // previous iteration's last-pass reads are done
        pass_store<N, P>(t, v, frame);
        team_sync<PI::TEAM>();
        fft_passes<N, P + 1, POWER_ONLY>(t, v, frame, tw);
    }
}

// No packed-f32 VALU ops in these kernels: on CDNA4 a v_pk_* is two passes through the SIMD and costs register pairing (moves);
// the backend forms them from the complex arithmetic anyway (welch_kernel<128>: 357 packed ops + 243 moves of 1117 VALU instructions).
// The attribute turns the target feature off per kernel -- the forced-inline helpers follow the kernel they are inlined into.
// Measured on the all-scalar build, same box, alternating: welch_kernel N = 16 ... 16384 +4 ... +48 % (128: 204 -> 301 GS/s, 256: 281 ->
// 380, 8192: 152 -> 193, 16384: 108 -> 134; 512 ... 4096: +6 %), the chirp-z kernel +44 % at M = 256, +3.5 / +8 % at M = 8192 / 16384 but
// -1.5 ... -9 % at M = 512 ... 4096, which therefore keep the packed form (launch_welch picks); the decimator and post kernels: no change.
// (__syncthreads() called DIRECTLY from a kernel with the attribute below stayed a real call: the header's function has the default
// target features.  Through this forced-inline wrapper it is inlined where it is written first.)
__device__ __forceinline__ void block_sync() { __syncthreads(); }

#if defined(__HIP_DEVICE_COMPILE__)
#define PSDK_SCALAR_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define PSDK_SCALAR_F32 // (the host pass of hipcc knows no such feature)
#endif

// FR: the launch holds jobs whose stream is read in place from AdcDac frames (a runtime branch per load otherwise sat in
// every launch: N = 128 lost 30 % to it)
template <int N, bool FR = false>
// (the kernel holds ~200 registers at N = 128: two wavefronts a SIMD.  Asking for three or four -- 168 / 128 registers, 148 / 288 bytes
// of scratch -- measured 403 -> 317 / 289 GS/s.)
__global__ __launch_bounds__(WelchCfg<N>::BLOCK) PSDK_SCALAR_F32 void welch_kernel(const WelchBatch batch,
                                                                  const float *__restrict__ win,
                                                                  const cf *__restrict__ tw)
{
    using Cfg = WelchCfg<N>;
    using P0 = PassInfo<N, 0>;
    constexpr int E = Cfg::E, TEAM = Cfg::TEAM, TEAMS = Cfg::TEAMS, SPT = Cfg::SPT;

    __shared__ cf frames[TEAMS * LdsFrame<N>::SIZE];
    __shared__ float red[Cfg::WAVES * 2];

    // workgroup -> job; this workgroup walks the job's tiles lt = wb, wb + nblocks, ...
    const int ji = job_of_unit(batch, (int)blockIdx.x, [](const SegJob &j) { return j.block_begin; });
    const SegJob &job = batch.jobs[ji];
    const int wb = blockIdx.x - job.block_begin;

    const int team = threadIdx.x / TEAM;
    const int t = threadIdx.x % TEAM;
    cf *frame = frames + team * LdsFrame<N>::SIZE;
    const int hop = batch.hop;
    const int detrend = batch.detrend;

    float q[E];
#pragma unroll
    for (int s = 0; s < E; ++s)
        q[s] = 0.0f;

    for (int lt = wb; lt < job.ntiles; lt += job.nblocks) {
    const int seg_lo = lt * SPT;
    const int seg_hi = min(job.nseg, seg_lo + SPT);
    const int npairs = (seg_hi - seg_lo + 1) >> 1;
    for (int p0 = 0; p0 < npairs; p0 += TEAMS) {
        const int p = p0 + team;
        const int la = seg_lo + 2 * p; // local segment index of the "real" lane
        const bool act_a = la < seg_hi, act_b = la + 1 < seg_hi;
        // sample j of segment a / b: from the f32 stream, or decoded from AdcDac frames (a stage-0 span read in place)
        const long long ofs_a = (job.seg0 + la) * (long long)hop - job.src_base;
        const bool fr = job.fspan >= 0;
        const FrameSpan &fsp = batch.fspans[fr ? job.fspan : 0];
        // (always_inline on the lambdas: a lambda does not carry the kernel's target attribute, and a callee with other target features
        // is not inlined unless it must be -- left alone they became 42 calls with scratch traffic inside the loop: N = 128 read 61 GS/s)
        // Lanes without a segment (the odd last one of a tile) load the job's FIRST segment instead and drop the values: every load of
        // the loop is then unconditional -- as `act ? x[..] : 0` each of the 32 loads of a pair sat in a basic block of its own behind an
        // exec-mask branch.
        const long long ofs_safe = job.seg0 * (long long)hop - job.src_base;
        const long long ofs_la = act_a ? ofs_a : ofs_safe, ofs_lb = act_b ? ofs_a + hop : ofs_safe;
        auto xat = [&](long long ofs, int j) __attribute__((always_inline)) {
            if constexpr (FR) {
                if (fr)
                    return frame_sample(fsp, job.fch, (unsigned long long)(ofs + job.s_off + j));
            }
            return job.src[ofs + j];
        };
        auto xa = [&](int j) __attribute__((always_inline)) { return xat(ofs_la, j); };
        auto xb = [&](int j) __attribute__((always_inline)) { return xat(ofs_lb, j); };

        float ra[E], rb[E];
#pragma unroll
        for (int i = 0; i < P0::NB; ++i)
#pragma unroll
            for (int m = 0; m < P0::R; ++m) {
                const int nidx = P0::elem(t, i, m);
                const float va = xa(nidx), vb = xb(nidx);
                ra[i * P0::R + m] = act_a ? va : 0.0f;
                rb[i * P0::R + m] = act_b ? vb : 0.0f;
            }

        // Detrend (src/psd.rs:75-113) as (x - o) - (m + n s): o is a sample of the segment, so the
        // first difference is exact and a DC level far above the noise costs no low bits
        float oa = 0.0f, ob = 0.0f, ma = 0.0f, mb = 0.0f;
        slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
        if (detrend == 1) { // Midpoint :87-93
            const float va = xa(N / 2), vb = xb(N / 2);
            oa = act_a ? va : 0.0f;
            ob = act_b ? vb : 0.0f;
        } else if (detrend == 2) { // Span :94-102 (ramp evaluated as o0 + n*slope)
            const float a0 = xa(0), a1 = xa(N - 1), b0 = xb(0), b1 = xb(N - 1);
            if (act_a) {
                oa = a0;
                sa = span_slope(oa, a1, N);
            }
            if (act_b) {
                ob = b0;
                sb = span_slope(ob, b1, N);
            }
        } else if (detrend == 3) { // Mean :103-109 in two steps: o = f32 mean of the samples, m = mean of x - o
            // (see fused.hip: neither a rounded offset nor a sample pivot leaves bins 0 and 1 alone)
            auto team_sum2 = [&](float &pa, float &pb) __attribute__((always_inline)) {
                constexpr int W = TEAM < 64 ? TEAM : 64;
#pragma unroll
                for (int o = W / 2; o > 0; o >>= 1) {
                    pa += __shfl_xor(pa, o);
                    pb += __shfl_xor(pb, o);
                }
                if constexpr (TEAM > 64) {
                    // a team spans TEAM / 64 wavefronts: combine exactly those through LDS
                    constexpr int WPT = TEAM / 64;
                    const int w = threadIdx.x >> 6;
                    if ((threadIdx.x & 63) == 0) {
                        red[2 * w] = pa;
                        red[2 * w + 1] = pb;
                    }
                    block_sync();
                    pa = 0.0f;
                    pb = 0.0f;
                    for (int i = 0; i < WPT; ++i) {
                        pa += red[2 * (team * WPT + i)];
                        pb += red[2 * (team * WPT + i) + 1];
                    }
                    block_sync();
                }
            };
            float pa = 0.0f, pb = 0.0f;
#pragma unroll
            for (int s = 0; s < E; ++s) {
                pa += ra[s];
                pb += rb[s];
            }
            team_sum2(pa, pb);
            oa = pa / (float)N;
            ob = pb / (float)N;
            pa = 0.0f;
            pb = 0.0f;
#pragma unroll
            for (int s = 0; s < E; ++s) {
                pa += ra[s] - oa;
                pb += rb[s] - ob;
            }
            team_sum2(pa, pb);
            ma = pa / (float)N;
            mb = pb / (float)N;
        }

        float ampa = 1.0f, ampb = 1.0f;
        if (job.ewma) {
            ampa = ewma_amp(job, job.step0 + la);
            ampb = ewma_amp(job, job.step0 + la + 1);
        }

        cf v[E];
#pragma unroll
        for (int i = 0; i < P0::NB; ++i)
#pragma unroll
            for (int m = 0; m < P0::R; ++m) {
                const int s = i * P0::R + m;
                const int nidx = P0::elem(t, i, m);
                const float w = win[nidx];
                float a = ra[s], b = rb[s];
                if (detrend != 0) {
                    a = fmaf(-(float)nidx, sa.lo, fmaf(-(float)nidx, sa.hi, a - oa)) - ma;
                    b = fmaf(-(float)nidx, sb.lo, fmaf(-(float)nidx, sb.hi, b - ob)) - mb;
                }
                a *= w;
                b *= w;
                if (job.ewma) {
                    a *= ampa;
                    b *= ampb;
                }
                v[s].re = a;
                v[s].im = b;
            }

        fft_passes<N, 0, true>(t, v, frame, tw);

#pragma unroll
        for (int s = 0; s < E; ++s)
            q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
    }
    }

    // combine the teams and write the workgroup's partial in natural bin order
    float *fq = reinterpret_cast<float *>(frames);
    block_sync();
#pragma unroll
    for (int s = 0; s < E; ++s)
        fq[team * N + freq_of_slot<N>(t, s)] = q[s];
    block_sync();
    float *out = job.partial + (size_t)wb * N;
    for (int k = threadIdx.x; k < N; k += Cfg::BLOCK) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < TEAMS; ++g)
            acc += fq[g * N + k];
        out[k] = acc;
    }
}

// ---------------------------------------------------------------------------
// Welch kernel for an FFT size that is not a power of two
// ---------------------------------------------------------------------------
// rustfft plans any length (src/psd.rs:418 `plan_fft_forward(N)`); the reference only asks for (N - overlap) % 8 == 0
// (:246-247).  Sizes other than 2^k run the same generic two-pass path with the N-point DFT evaluated as a chirp-z
// (Bluestein) convolution on the power-of-two passes above: with c[n] = exp(i pi n^2 / N),
//     X[k] = conj(c[k]) sum_n (z[n] conj(c[n])) c[k - n]
// i.e. one forward FFT of length M >= 2N - 1 of y = z conj(c) (zero padded), a pointwise product with B = FFT_M(c wrapped),
// and one inverse FFT (a forward one on the conjugate).  Only |X[k]|^2 is consumed (src/psd.rs:228-233): the final chirp
// and the conjugations of the inverse are unit phases and drop out; the 1/M of the inverse is applied to the partial.
// Two real segments ride one complex transform exactly as in welch_kernel (the DFT is linear).
template <int M>
struct BlueCfg {
    using Plan = FftPlan<M>;
    static constexpr int E = Plan::E;
    static constexpr int TEAM = Plan::TEAM;
    static constexpr int BLOCK = TEAM > 256 ? TEAM : 256;
    static constexpr int TEAMS = BLOCK / TEAM;
    static constexpr int SPT = (2 * TEAMS * 4 > 32) ? 2 * TEAMS * 4 : 32;
    static constexpr int WAVES = BLOCK / 64;
};

// UNCOND: every load of the pair loop unconditional, as in welch_kernel (lanes without a segment read the job's first one, slots of
// the zero padding read index n - 1, the values are dropped).  It removes a branch per load and costs registers: +8 ... +21 % at
// M = 512 ... 4096 (N = 240: 69 -> 79 GS/s, 2000: 63 -> 76), but M <= 256 goes from 240 to 360 registers (one wavefront a SIMD instead
// of two: N = 112 104 -> 82) and M = 16384 spills 300 more bytes (N = 8000 26 -> 21): those keep the conditional loads.
template <int M, bool UNCOND>
__device__ __forceinline__ void welch_bluestein_body(const WelchBatch &batch, int n, const float *__restrict__ win,
                                                     const cf *__restrict__ twm, const cf *__restrict__ chirp,
                                                     const cf *__restrict__ bhat)
{
    using Cfg = BlueCfg<M>;
    using P0 = PassInfo<M, 0>;
    constexpr int E = Cfg::E, TEAM = Cfg::TEAM, TEAMS = Cfg::TEAMS, SPT = Cfg::SPT;
    __shared__ cf frames[TEAMS * LdsFrame<M>::SIZE];
    __shared__ float red[Cfg::WAVES * 2];

    const int ji = job_of_unit(batch, (int)blockIdx.x, [](const SegJob &j) { return j.block_begin; });
    const SegJob &job = batch.jobs[ji];
    const int wb = blockIdx.x - job.block_begin;
    const int team = threadIdx.x / TEAM;
    const int t = threadIdx.x % TEAM;
    cf *frame = frames + team * LdsFrame<M>::SIZE;
    const int hop = batch.hop;
    const int detrend = batch.detrend;
    const float inv_n = 1.0f / (float)n;

    float q[E];
#pragma unroll
    for (int s = 0; s < E; ++s)
        q[s] = 0.0f;

    for (int lt = wb; lt < job.ntiles; lt += job.nblocks) {
    const int seg_lo = lt * SPT;
    const int seg_hi = min(job.nseg, seg_lo + SPT);
    const int npairs = (seg_hi - seg_lo + 1) >> 1;
    for (int p0 = 0; p0 < npairs; p0 += TEAMS) {
        const int p = p0 + team;
        const int la = seg_lo + 2 * p;
        const bool act_a = la < seg_hi, act_b = la + 1 < seg_hi;
        const long long ofs_a = (job.seg0 + la) * (long long)hop - job.src_base;
        // unconditional loads, as in welch_kernel: lanes without a segment read the job's first one, slots of the zero padding (index >= n)
        // read sample n - 1, and the values are dropped
        const long long ofs_safe = job.seg0 * (long long)hop - job.src_base;
        const long long ofs_la = (UNCOND && !act_a) ? ofs_safe : ofs_a, ofs_lb = (UNCOND && !act_b) ? ofs_safe - hop : ofs_a;
        auto xa = [&](int j) __attribute__((always_inline)) { return job.src[ofs_la + j]; };
        auto xb = [&](int j) __attribute__((always_inline)) { return job.src[ofs_lb + hop + j]; };

        float ra[E], rb[E];
#pragma unroll
        for (int i = 0; i < P0::NB; ++i)
#pragma unroll
            for (int m = 0; m < P0::R; ++m) {
                const int nidx = P0::elem(t, i, m);
                if constexpr (UNCOND) {
                    const int jc = nidx < n ? nidx : n - 1;
                    const float va = xa(jc), vb = xb(jc);
                    ra[i * P0::R + m] = (act_a && nidx < n) ? va : 0.0f;
                    rb[i * P0::R + m] = (act_b && nidx < n) ? vb : 0.0f;
                } else {
                    ra[i * P0::R + m] = (act_a && nidx < n) ? xa(nidx) : 0.0f;
                    rb[i * P0::R + m] = (act_b && nidx < n) ? xb(nidx) : 0.0f;
                }
            }

        // detrend parameters as in welch_kernel (src/psd.rs:75-113)
        float oa = 0.0f, ob = 0.0f, ma = 0.0f, mb = 0.0f;
        slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
        if (detrend == 1) {
            oa = act_a ? xa(n / 2) : 0.0f;
            ob = act_b ? xb(n / 2) : 0.0f;
        } else if (detrend == 2) {
            if (act_a) {
                oa = xa(0);
                sa = span_slope(oa, xa(n - 1), n);
            }
            if (act_b) {
                ob = xb(0);
                sb = span_slope(ob, xb(n - 1), n);
            }
        } else if (detrend == 3) {
            auto team_sum2 = [&](float &pa, float &pb) __attribute__((always_inline)) {
                constexpr int W = TEAM < 64 ? TEAM : 64;
#pragma unroll
                for (int o = W / 2; o > 0; o >>= 1) {
                    pa += __shfl_xor(pa, o);
                    pb += __shfl_xor(pb, o);
                }
                if constexpr (TEAM > 64) {
                    constexpr int WPT = TEAM / 64;
                    const int w = threadIdx.x >> 6;
                    if ((threadIdx.x & 63) == 0) {
                        red[2 * w] = pa;
                        red[2 * w + 1] = pb;
                    }
                    __syncthreads();
                    pa = 0.0f;
                    pb = 0.0f;
                    for (int i = 0; i < WPT; ++i) {
                        pa += red[2 * (team * WPT + i)];
                        pb += red[2 * (team * WPT + i) + 1];
                    }
                    __syncthreads();
                }
            };
            float pa = 0.0f, pb = 0.0f;
#pragma unroll
            for (int s = 0; s < E; ++s) { // (slots past the segment hold zeros)
                pa += ra[s];
                pb += rb[s];
            }
            team_sum2(pa, pb);
            oa = pa * inv_n;
            ob = pb * inv_n;
            pa = 0.0f;
            pb = 0.0f;
#pragma unroll
            for (int i = 0; i < P0::NB; ++i)
#pragma unroll
                for (int m = 0; m < P0::R; ++m)
                    if (P0::elem(t, i, m) < n) {
                        pa += ra[i * P0::R + m] - oa;
                        pb += rb[i * P0::R + m] - ob;
                    }
            team_sum2(pa, pb);
            ma = pa * inv_n;
            mb = pb * inv_n;
        }
        float ampa = 1.0f, ampb = 1.0f;
        if (job.ewma) {
            ampa = ewma_amp(job, job.step0 + la);
            ampb = ewma_amp(job, job.step0 + la + 1);
        }

        cf v[E];
#pragma unroll
        for (int i = 0; i < P0::NB; ++i)
#pragma unroll
            for (int m = 0; m < P0::R; ++m) {
                const int s = i * P0::R + m;
                const int nidx = P0::elem(t, i, m);
                cf z = {0.0f, 0.0f};
                if constexpr (UNCOND) { // (table reads unconditional too; padded slots are zeroed afterwards)
                    const int jc = nidx < n ? nidx : n - 1;
                    const float w = win[jc];
                    const cf c = chirp[jc]; // y = z conj(c)
                    float a = ra[s], b = rb[s];
                    if (detrend != 0) {
                        a = fmaf(-(float)nidx, sa.lo, fmaf(-(float)nidx, sa.hi, a - oa)) - ma;
                        b = fmaf(-(float)nidx, sb.lo, fmaf(-(float)nidx, sb.hi, b - ob)) - mb;
                    }
                    a *= w;
                    b *= w;
                    if (job.ewma) {
                        a *= ampa;
                        b *= ampb;
                    }
                    if (nidx < n)
                        z = {a * c.re + b * c.im, b * c.re - a * c.im};
                } else if (nidx < n) {
                    const float w = win[nidx];
                    float a = ra[s], b = rb[s];
                    if (detrend != 0) {
                        a = fmaf(-(float)nidx, sa.lo, fmaf(-(float)nidx, sa.hi, a - oa)) - ma;
                        b = fmaf(-(float)nidx, sb.lo, fmaf(-(float)nidx, sb.hi, b - ob)) - mb;
                    }
                    a *= w;
                    b *= w;
                    if (job.ewma) {
                        a *= ampa;
                        b *= ampb;
                    }
                    const cf c = chirp[nidx]; // y = z conj(c)
                    z = {a * c.re + b * c.im, b * c.re - a * c.im};
                }
                v[s] = z;
            }

        fft_passes<M, 0, false>(t, v, frame, twm); // Y = FFT_M(y): slot s holds bin freq_of_slot<M>(t, s) (the VALUES are used: no rotation)
        team_sync<TEAM>();                  // the team's last-pass reads of its frame are done
#pragma unroll
        for (int s = 0; s < E; ++s) { // conj(Y B), back in natural order for the second transform
            const int j = freq_of_slot<M>(t, s);
            const cf u = cmul(v[s], bhat[j]);
            frame[LdsFrame<M>::at(j)] = {u.re, -u.im};
        }
        team_sync<TEAM>();
#pragma unroll
        for (int i = 0; i < P0::NB; ++i)
#pragma unroll
            for (int m = 0; m < P0::R; ++m)
                v[i * P0::R + m] = frame[LdsFrame<M>::at(P0::elem(t, i, m))];
        fft_passes<M, 0, true>(t, v, frame, twm); // M conj(convolution): slot s holds output index freq_of_slot<M>(t, s); only |.|^2 is used
#pragma unroll
        for (int s = 0; s < E; ++s)
            q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
    }
    }

    // combine the teams; outputs 0 ... n - 1 of the convolution are the bins, in natural order; 1/M^2 of the inverse
    float *fq = reinterpret_cast<float *>(frames);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < E; ++s)
        fq[team * M + freq_of_slot<M>(t, s)] = q[s];
    __syncthreads();
    float *out = job.partial + (size_t)wb * n;
    const float scale = 1.0f / ((float)M * (float)M);
    for (int k = threadIdx.x; k < n; k += Cfg::BLOCK) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < TEAMS; ++g)
            acc += fq[g * M + k];
        out[k] = acc * scale;
    }
}

// the kernel in its two builds: packed f32 (PSDK_SCALAR_F32 above) and unconditional loads for M = 512 ... 4096, all-scalar with
// conditional loads elsewhere
template <int M>
__global__ __launch_bounds__(BlueCfg<M>::BLOCK) void welch_bluestein_kernel(const WelchBatch batch, int n, const float *__restrict__ win,
                                                                          const cf *__restrict__ twm, const cf *__restrict__ chirp,
                                                                          const cf *__restrict__ bhat)
{
    welch_bluestein_body<M, true>(batch, n, win, twm, chirp, bhat);
}
template <int M>
__global__ __launch_bounds__(BlueCfg<M>::BLOCK) PSDK_SCALAR_F32 void welch_bluestein_kernel_scalar(const WelchBatch batch, int n,
                                                                                              const float *__restrict__ win,
                                                                                              const cf *__restrict__ twm,
                                                                                              const cf *__restrict__ chirp,
                                                                                              const cf *__restrict__ bhat)
{
    welch_bluestein_body<M, false>(batch, n, win, twm, chirp, bhat);
}

// ---------------------------------------------------------------------------
// /8 half-band decimator
// ---------------------------------------------------------------------------

__constant__ float c_taps_a[HBF_MA] = {PSDK_HBF_TAPS_A};
__constant__ float c_taps_b[HBF_MB] = {PSDK_HBF_TAPS_B};
__constant__ float c_taps_c[HBF_MC] = {PSDK_HBF_TAPS_C};

// Block geometry for DEC_TILE outputs starting at output index mt0 (see
// hbf_taps.h for the per-stage spans); every block origin is even so that the
// even/odd polyphase split of each stage input is aligned.
namespace dec {
constexpr int B_PRE = HBF_PRE_B; // 58: B0 = 2*mt0 - B_PRE
constexpr int A_PRE = HBF_PRE_A; // 138: A0 = 4*mt0 - A_PRE
constexpr int X_PRE = HBF_HALO;  // 288: X0 = 8*mt0 - X_PRE
static_assert(2 * A_PRE + HBF_SPAN_A <= X_PRE, "halo too small");
constexpr int NB_OUT = 2 * DEC_TILE + B_PRE; // 570
constexpr int NA_OUT = 4 * DEC_TILE + A_PRE; // 1162
constexpr int NX = 8 * DEC_TILE + X_PRE;     // 2336
// polyphase index offsets: out j -> even in[j + CE], odd in[j + CO + i] and in[j + CO + 2M-1-i]
constexpr int A_D = X_PRE / 2 - A_PRE;       // A0 - X0/2 = 6
constexpr int A_CE = A_D - HBF_MA + 1, A_CO = A_D - 2 * HBF_MA + 1;
constexpr int B_D = A_PRE / 2 - B_PRE;       // 11
constexpr int B_CE = B_D - HBF_MB + 1, B_CO = B_D - 2 * HBF_MB + 1;
constexpr int C_D = B_PRE / 2;               // 29
constexpr int C_CE = C_D - HBF_MC + 1, C_CO = C_D - 2 * HBF_MC + 1;
static_assert(A_CO >= 0 && B_CO >= 0 && C_CO >= 0, "negative polyphase offset");
} // namespace dec

template <int M>
__device__ __forceinline__ float hbf_point(const float *__restrict__ ev, const float *__restrict__ od,
                                           const float *taps, int j, int ce, int co)
{
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i)
        acc += (od[j + co + i] + od[j + co + 2 * M - 1 - i]) * taps[i];
    return ev[j + ce] + acc;
}

// Four adjacent outputs j ... j + 3 (j a multiple of 4) of one half-band stage from ONE window of 2 M + 3 odd-phase samples held in
// registers: 2 M + 7 LDS reads for four outputs where hbf_point reads 2 M + 1 for each (stage A: 37 against 124) -- the kernel spent
// as long on its LDS reads as on its HBM loads (22 LDS words per input sample).  Every output's sum is formed in hbf_point's order.
template <int M>
__device__ __forceinline__ void hbf_points4(const float *__restrict__ ev, const float *__restrict__ od, const float *taps, int j, int ce,
                                            int co, float (&y)[4])
{
    float w[2 * M + 3];
#pragma unroll
    for (int k = 0; k < 2 * M + 3; ++k)
        w[k] = od[j + co + k];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i)
            acc += (w[u + i] + w[u + 2 * M - 1 - i]) * taps[i];
        y[u] = ev[j + u + ce] + acc;
    }
}

template <bool FR>
__global__ __launch_bounds__(256) void hbf_dec8_kernel(const DecBatch batch)
{
    using namespace dec;
    // (+ 4: the last group of four outputs of a stage may reach up to three outputs -- and their window -- past the stage's length;
    // those outputs are computed from whatever is there and never used)
    __shared__ float xe[NX / 2 + 4], xo[NX / 2 + 4];
    __shared__ __attribute__((aligned(8))) float ae[NA_OUT / 2 + 4], ao[NA_OUT / 2 + 4];
    __shared__ __attribute__((aligned(8))) float be[NB_OUT / 2 + 4], bo[NB_OUT / 2 + 4];

    const int tile = blockIdx.x;
    const int ji = job_of_unit(batch, tile, [](const DecJob &j) { return j.tile_begin; });
    const DecJob &job = batch.jobs[ji];
    const int lt = tile - job.tile_begin;
    const long long mt0 = job.m0 + (long long)lt * DEC_TILE;
    const int nvalid = (int)min((long long)DEC_TILE, job.m0 + job.nout - mt0);
    const long long x0 = 8 * mt0 - X_PRE;
    const long long x_end = 8 * (mt0 + nvalid); // inputs at or beyond are not needed
    const int tid = threadIdx.x;

    // stage input -> even/odd polyphase arrays (zero history before the stream start)
    // (all of a thread's loads first, then its LDS stores: five sample pairs in flight a thread -- the kernel waits on memory, SQ 0.65 of
    // its wave time; one 8-byte load a pair where the block's first sample is 8-byte aligned)
    constexpr int XIT = (NX / 2 + 255) / 256;
    float xev[XIT], xov[XIT];
    const bool al8 = !(FR && job.fspan >= 0) && (reinterpret_cast<uintptr_t>(job.src + (x0 - job.src_base)) & 7u) == 0;
#pragma unroll
    for (int u = 0; u < XIT; ++u) {
        const int r = tid + 256 * u;
        const long long i0 = x0 + 2 * r;
        float e = 0.0f, o = 0.0f;
        const bool want = r < NX / 2 && i0 >= 0 && i0 + 1 < x_end;
        if (FR && job.fspan >= 0) { // AdcDac frames read in place
            if (want) {
                const unsigned long long si = (unsigned long long)(i0 - job.src_base + job.s_off);
                e = frame_sample(batch.fspans[job.fspan], job.fch, si);
                o = frame_sample(batch.fspans[job.fspan], job.fch, si + 1);
            }
        } else {
            // unconditional (as in welch_kernel): the index clamped into [0, x_end - 2] -- always inside the source: a negative index is
            // the stream's start, where the source begins at sample 0 -- and the value dropped; (x0 and x_end are even: the clamp keeps
            // the pair's parity and the 8-byte alignment)
            const long long ic = i0 < 0 ? 0 : (i0 + 1 < x_end ? i0 : x_end - 2);
            const float *p = job.src + (ic - job.src_base);
            float ve, vo;
            if (al8) {
                const float2 v = *reinterpret_cast<const float2 *>(p);
                ve = v.x;
                vo = v.y;
            } else {
                ve = p[0];
                vo = p[1];
            }
            e = want ? ve : 0.0f;
            o = want ? vo : 0.0f;
        }
        xev[u] = e;
        xov[u] = o;
    }
#pragma unroll
    for (int u = 0; u < XIT; ++u) {
        const int r = tid + 256 * u;
        if (r < NX / 2) {
            xe[r] = xev[u];
            xo[r] = xov[u];
        }
    }
    __syncthreads();
    for (int j = 4 * tid; j < NA_OUT; j += 4 * 256) { // outputs j, j + 2 -> ae[j/2], ae[j/2 + 1]; j + 1, j + 3 -> ao[...]
        float y[4];
        hbf_points4<HBF_MA>(xe, xo, c_taps_a, j, A_CE, A_CO, y);
        *reinterpret_cast<float2 *>(ae + (j >> 1)) = make_float2(y[0], y[2]);
        *reinterpret_cast<float2 *>(ao + (j >> 1)) = make_float2(y[1], y[3]);
    }
    __syncthreads();
    for (int j = 4 * tid; j < NB_OUT; j += 4 * 256) {
        float y[4];
        hbf_points4<HBF_MB>(ae, ao, c_taps_b, j, B_CE, B_CO, y);
        *reinterpret_cast<float2 *>(be + (j >> 1)) = make_float2(y[0], y[2]);
        *reinterpret_cast<float2 *>(bo + (j >> 1)) = make_float2(y[1], y[3]);
    }
    __syncthreads();
    if (4 * tid < nvalid) { // (the first wavefront: 64 lanes x 4 outputs)
        float y[4];
        hbf_points4<HBF_MC>(be, bo, c_taps_c, 4 * tid, C_CE, C_CO, y);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long o = mt0 + 4 * tid + u - batch.drain; // drop the first `drain` outputs ever
            if (4 * tid + u < nvalid && o >= 0)
                job.dst[o - job.dst_base] = y[u];
        }
    }
}

// ---------------------------------------------------------------------------
// reduce / tail / fill / adcdac
// ---------------------------------------------------------------------------

constexpr int RED_BINS = 32, RED_SLICES = 32; // 1024 threads: 32 bins x 32 slices of the partial list

__device__ __forceinline__ void reduce_body(const RedJob &job, int n, int xblk)
{
    __shared__ double part[RED_SLICES][RED_BINS + 1];
    const int lane = threadIdx.x % RED_BINS, slice = threadIdx.x / RED_BINS;
    const int k = xblk * RED_BINS + lane;
    const bool live = k <= n / 2;
    double acc = 0.0; // f64 partial sums: the fold adds no rounding of its own
    if (live) {
        const int km = k ? n - k : 0; // (n - k) mod n: n need not be a power of two
        int t = slice;
        for (; t + 3 * RED_SLICES < job.nparts; t += 4 * RED_SLICES) { // (loaded four rows at a time, added one at a time: same sum)
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *p = job.partial + (size_t)(t + u * RED_SLICES) * n;
                v[u][0] = p[k];
                v[u][1] = p[km];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc += (double)v[u][0] + (double)v[u][1];
        }
        for (; t < job.nparts; t += RED_SLICES) {
            const float *p = job.partial + (size_t)t * n;
            acc += (double)p[k] + (double)p[km];
        }
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && live) {
        acc = 0.0;
#pragma unroll
        for (int i = 0; i < RED_SLICES; ++i)
            acc += part[i][lane];
        job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc);
    }
}

// One launch for the round's epilogue: workgroups [0, nred) fold the partials of the
// (channel, stage) jobs, the rest carry the stream tails.  The two roles touch disjoint data.
__global__ __launch_bounds__(RED_BINS *RED_SLICES) void post_kernel(const RedBatch red, const TailBatch tail,
                                                                  int red_xblocks)
{
    const int nred = red.njobs * red_xblocks;
    if ((int)blockIdx.x < nred) {
        reduce_body(red.jobs[blockIdx.x / red_xblocks], red.n, blockIdx.x % red_xblocks);
        return;
    }
    const TailJob &job = tail.jobs[blockIdx.x - nred];
    if (job.fspan >= 0) { // a seam or a carried tail whose samples still sit in AdcDac frames: decoded on the way
        const FrameSpan &fs = tail.fspans[job.fspan];
        for (int i = threadIdx.x; i < job.count; i += RED_BINS * RED_SLICES)
            job.dst[i] = frame_sample(fs, job.fch, (unsigned long long)job.s_off + (unsigned long long)i);
        return;
    }
    for (int i = threadIdx.x; i < job.count; i += RED_BINS * RED_SLICES)
        job.dst[i] = job.src[i];
}

// SplitMix64: output i of the generator seeded with `key` is mix64(key + (i + 1) * GAMMA).  The stride matters:
// hashing CONSECUTIVE integers (an earlier version) leaves structure at 2^24-sample scales that a ten-stage
// cascade resolves as spectral lines in its deepest stages.
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void fill_noise_kernel(float *x, size_t len, uint64_t seed,
                                                         uint64_t first)
{
    const float scale = 3.4641016151377544f; // sqrt(12), src/psd.rs:605
    const uint64_t gamma = 0x9E3779B97F4A7C15ull;
    const uint64_t key = mix64(seed + gamma); // seeds 1 apart (one per channel) give unrelated streams
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) {
        const uint64_t r = mix64(key + (first + i + 1) * gamma);
        const float u = (float)(r >> 40) * 5.9604644775390625e-08f; // 2^-24
        x[i] = (u - 0.5f) * scale;
    }
}

// AdcDac payload (src/de/data.rs:13): per batch [[[u8;2];8];4], channel-major.
// One thread per (frame, batch, channel): 16 payload bytes -> 8 samples.
__global__ __launch_bounds__(256) void adcdac_kernel(const uint8_t *__restrict__ frames,
                                                     size_t frame_size, size_t n_frames, int batches,
                                                     float *d0, float *d1, float *d2, float *d3)
{
    const float lsb = 4.096f * 2.5f / 32768.0f; // src/de/data.rs:28-35
    const size_t total = n_frames * (size_t)batches * 4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int ch = (int)(g & 3);
        const size_t fb = g >> 2;
        const size_t f = fb / (size_t)batches, b = fb % (size_t)batches;
        const uint8_t *p = frames + f * frame_size + 8 + (b * 4 + (size_t)ch) * 16;
        float *dst = (ch == 0 ? d0 : ch == 1 ? d1 : ch == 2 ? d2 : d3) + fb * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint16_t raw = (uint16_t)p[2 * i] | ((uint16_t)p[2 * i + 1] << 8); // i16::from_le_bytes
            if (ch >= 2)
                raw = (uint16_t)(raw + 0x8000u); // wrapping_add(i16::MIN) :64,:75
            dst[i] = (float)(int16_t)raw * lsb;
        }
    }
}

// The other payload formats (src/de/data.rs:84-212) -- Fls (format id 2), ThermostatEem (3), Mpll (4): ONE sample per batch and
// trace.  One thread per (frame, batch).  The arithmetic is the reference's, operation by operation in f32 (`as f32` conversions,
// separate products and sum -- rustc never fuses them --, a correctly rounded square root, the scale constants evaluated in f32
// in the reference's order): the traces are bit-identical to Payload::traces.
struct PayloadFmt {
    int batch_bytes, ntraces;
};
__host__ __device__ constexpr PayloadFmt payload_fmt(int id)
{
    return id == 2 ? PayloadFmt{56, 4}   // [[[u8;4];7];2]  data.rs:86
         : id == 3 ? PayloadFmt{80, 4}   // [[u8;4];16+4]   data.rs:144
                   : PayloadFmt{24, 3};  // [[u8;4];6]      data.rs:168
}
template <int FMT>
__global__ __launch_bounds__(256) void payload_kernel(const uint8_t *__restrict__ frames, size_t frame_size, size_t n_frames, int batches,
                                                      float *d0, float *d1, float *d2, float *d3)
{
    constexpr int BB = payload_fmt(FMT).batch_bytes;
    const bool aligned = ((reinterpret_cast<uintptr_t>(frames) | frame_size) & 3u) == 0;
    const size_t total = n_frames * (size_t)batches;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t f = g / (size_t)batches, b = g % (size_t)batches;
        const uint8_t *p = frames + f * frame_size + 8 + b * BB;
        auto word = [&](int i) { // u32::from_le_bytes of word i of the batch: one load when the frames are 4-byte aligned (every valid
            const uint8_t *q = p + 4 * i; // frame_size is a multiple of 8; the base is the caller's), bytes otherwise
            if (aligned)
                return *reinterpret_cast<const uint32_t *>(q);
            return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        };
        auto i32f = [&](int i) { return (float)(int32_t)word(i); }; // i32::from_le_bytes(..) as f32
        // a.powi(2) + c.powi(2), .sqrt(): two products, one sum, a correctly rounded root.  (HIP's __fmul_rn / __fadd_rn are plain
        // operators the backend may fuse, and __fsqrt_rn is the approximate native root: the pragma and sqrtf -- IEEE under hipcc's
        // default -fhip-fp32-correctly-rounded-divide-sqrt -- are what pins the arithmetic.)
        auto hyp = [](float a, float c) {
#pragma clang fp contract(off)
            const float aa = a * a, cc = c * c;
            const float sum = aa + cc;
            return sqrtf(sum);
        };
        if constexpr (FMT == 2) { // Fls::traces, data.rs:97-139
            constexpr float inv_max = 1.0f / 2147483648.0f;                // 1.0 / (i32::MAX as f32)
            constexpr float ap = 6.28318530717958647692f / 65536.0f;       // TAU / (1i64 << 16) as f32
            d0[g] = hyp(i32f(0), i32f(1)) * inv_max;                       // "AR" :100-110
            const long long ph = (long long)((unsigned long long)word(2) | ((unsigned long long)word(3) << 32));
            d1[g] = (float)ph * ap;                                        // "AP" :111-123
            d2[g] = i32f(7) / 2147483648.0f;                               // "BI" b[1][0] :124-130 (a power of two: exact)
            d3[g] = i32f(8) / 2147483648.0f;                               // "BQ" b[1][1] :131-137
        } else if constexpr (FMT == 3) { // ThermostatEem::traces, data.rs:154-163: words 0, 8, 13, 16 as f32
            d0[g] = __uint_as_float(word(0));
            d1[g] = __uint_as_float(word(8));
            d2[g] = __uint_as_float(word(13));
            d3[g] = __uint_as_float(word(16));
        } else { // Mpll::traces, data.rs:178-211
            constexpr float two32 = 4294967296.0f;
            constexpr float c_phase = 6.28318530717958647692f / two32;     // TAU / (1u64 << 32) as f32
            constexpr float c_freq = 1.0f / 1.28e-3f / two32;              // 1.0 / 1.28e-3 / (1u64 << 32) as f32
            constexpr float c_amp = 10.24f / 10.0f * 2.0f * 2.0f / two32;  // 10.24 / 10.0 * 2.0 * 2.0 / (1u64 << 32) as f32
            d0[g] = i32f(4) * c_phase;                                     // "phase (rad)"
            d1[g] = i32f(5) * c_freq;                                      // "frequency (kHz)"
            d2[g] = hyp(i32f(0), i32f(1)) * c_amp;                         // "amplitude (V/G10)"
        }
    }
}

// Device-resident frames, one pass: Header::parse (src/de/frame.rs:25-37) + the AdcDac size checks (src/de/data.rs:22-25) of
// every frame, in the reference's order -- acc[0] <- max over the bad frames of ~(index << 2 | code) (code 1 InvalidHeader,
// 2 UnknownFormat, 3 PayloadSize / batches mismatch), i.e. the FIRST bad frame; 0 if all are good -- and Loss::update
// (src/loss.rs:11-26) over frames [0, n_loss): acc[1] += batches, acc[2] += the u32 sequence gaps between consecutive frames
// (wrapping_sub), acc[3] <- seq of frame 0 (low half) | seq + batches of the last (high half).  The workgroup that
// finishes last (acc[4]: arrival ticket) copies the four words to `host_out` (pinned host memory) and zeroes acc for the
// next call: one launch, no memset, no copy kernel -- the call sits on a side stream and the host waits for it alone.
// One WAVEFRONT per workgroup: the launch runs beside a fused kernel whose workgroups hold every register of their SIMDs; the
// slots the fused launch leaves free are whole fused workgroups (two wavefronts at N = 2048), and a single small wavefront fits
// any of them, which a four-wavefront workgroup does not (N = 2048 frames did not overlap: 416 GS/s, one launch per call).
constexpr int VERDICT_THREADS = 64;
__global__ __launch_bounds__(VERDICT_THREADS) void adcdac_verdict_kernel(const uint8_t *__restrict__ frames, size_t frame_size, size_t n_frames,
                                                             int batches, int payload_ok, int check, size_t n_loss,
                                                             unsigned long long *acc, unsigned long long *host_out)
{
    // the 8 header bytes of frame f as two words {magic | id << 16 | batches << 24, seq}: one 8-byte load when the frames are
    // 8-byte aligned (frame_size = 8 + 64 B always is; the base is the caller's), bytes otherwise.  Every header is a cache
    // line of its own, so the scan is latency-bound: four frames (eight loads) in flight per thread.
    const bool aligned = ((reinterpret_cast<uintptr_t>(frames) | frame_size) & 7u) == 0;
    auto header = [&](size_t f) -> uint2 {
        const uint8_t *p = frames + f * frame_size;
        if (aligned)
            return *reinterpret_cast<const uint2 *>(p);
        return make_uint2((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24),
                          (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24));
    };
    unsigned long long rec = 0, drop = 0, bad = 0;
    const size_t stride = (size_t)gridDim.x * VERDICT_THREADS;
    for (size_t f0 = (size_t)blockIdx.x * VERDICT_THREADS + threadIdx.x; f0 < n_frames; f0 += 4 * stride) {
        uint2 h[4], hp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t f = f0 + (size_t)u * stride;
            const bool in = f < n_frames;
            h[u] = in ? header(f) : make_uint2(0, 0);
            hp[u] = (in && f > 0 && f < n_loss) ? header(f - 1) : make_uint2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t f = f0 + (size_t)u * stride;
            if (f >= n_frames)
                continue;
            const uint32_t w = h[u].x, b = w >> 24;
            if (check) {
                int code = 0;
                if ((w & 0xffffu) != 0x057bu) // magic [0x7b, 0x05]
                    code = 1;
                else if (((w >> 16) & 0xffu) != 1u)
                    code = 2;
                else if (!payload_ok || (int)b != batches)
                    code = 3;
                if (code) {
                    const unsigned long long key = ~(((unsigned long long)f << 2) | (unsigned long long)code);
                    bad = key > bad ? key : bad;
                }
            }
            if (f < n_loss) {
                rec += b;
                if (f > 0)
                    drop += (uint32_t)(h[u].y - (hp[u].y + (hp[u].x >> 24)));
                if (f == 0)
                    atomicOr(acc + 3, (unsigned long long)h[u].y);
                if (f == n_loss - 1)
                    atomicOr(acc + 3, (unsigned long long)(uint32_t)(h[u].y + b) << 32);
            }
        }
    }
    // the wavefront's sums (64-bit shuffles)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rec += __shfl_down(rec, o);
        drop += __shfl_down(drop, o);
        const unsigned long long ob = __shfl_down(bad, o);
        bad = ob > bad ? ob : bad;
    }
    if (threadIdx.x == 0) {
        if (bad)
            atomicMax(acc, bad);
        atomicAdd(acc + 1, rec);
        atomicAdd(acc + 2, drop);
        __threadfence(); // this workgroup's sums are visible device-wide before its ticket is
        const bool last = atomicAdd(acc + 4, 1ull) + 1 == (unsigned long long)gridDim.x;
        if (last) { // every other workgroup's atomics came before its ticket: read them back with atomics too
            for (int i = 0; i < 4; ++i)
                host_out[i] = atomicExch(acc + i, 0ull);
            atomicExch(acc + 4, 0ull);
            __threadfence_system();
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

hipError_t launch_adcdac_verdict(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, int payload_ok, int check,
                                 size_t n_loss, unsigned long long *acc, unsigned long long *host_out, hipStream_t s)
{
    if (n_frames == 0)
        return hipSuccess;
    // a handful of small workgroups: the call runs beside a fused launch that leaves FRAME_RESERVE_BLOCKS workgroup slots free
    // (a slot is two or more wavefronts of 128 registers: several of these one-wavefront, < 32-register workgroups fit one)
    const unsigned blocks = (unsigned)std::min<size_t>(16 * FRAME_RESERVE_BLOCKS, (n_frames + VERDICT_THREADS - 1) / VERDICT_THREADS);
    hipLaunchKernelGGL(adcdac_verdict_kernel, dim3(blocks), dim3(VERDICT_THREADS), 0, s, frames, frame_size, n_frames, batches, payload_ok, check,
                       n_loss, acc, host_out);
    return hipGetLastError();
}

int welch_segments_per_tile(int n)
{
    if (bigfft_size(n)) // one tile, one workgroup row per job: the segments go pair by pair through global memory (bigfft.hip)
        return 1 << 20;
    switch (bluestein_size(n)) { // not a power of two: the tile of the chirp-z kernel of its transform size
#define PSDK_CASE(MM) \
    case MM:          \
        return BlueCfg<MM>::SPT;
        PSDK_CASE(32)
        PSDK_CASE(64)
        PSDK_CASE(128)
        PSDK_CASE(256)
        PSDK_CASE(512)
        PSDK_CASE(1024)
        PSDK_CASE(2048)
        PSDK_CASE(4096)
        PSDK_CASE(8192)
        PSDK_CASE(16384)
#undef PSDK_CASE
    default:
        break;
    }
    switch (n) {
#define PSDK_CASE(NN) \
    case NN:          \
        return WelchCfg<NN>::SPT;
        PSDK_CASE(16)
        PSDK_CASE(32)
        PSDK_CASE(64)
        PSDK_CASE(128)
        PSDK_CASE(256)
        PSDK_CASE(512)
        PSDK_CASE(1024)
        PSDK_CASE(2048)
        PSDK_CASE(4096)
        PSDK_CASE(8192)
        PSDK_CASE(16384)
#undef PSDK_CASE
    default:
        return 0;
    }
}

// smallest power of two M >= 2n - 1 for the chirp-z form of an n-point DFT (0: n is a power of two or out of range)
int bluestein_size(int n)
{
    if (n < 16 || n > 8192 || (n & (n - 1)) == 0)
        return 0;
    int m = 32;
    while (m < 2 * n - 1)
        m <<= 1;
    return m;
}

bool welch_supported(int n) { return welch_segments_per_tile(n) != 0; }

hipError_t launch_welch(int n, const WelchBatch &b, const float *win, const cf *tw, const cf *chirp, const cf *bhat, hipStream_t s)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    switch (bluestein_size(n)) { // tw: W_M^j, chirp: exp(i pi j^2 / n), bhat: FFT_M of the wrapped chirp
#define PSDK_CASE(MM)                                                                                          \
    case MM:                                                                                                   \
        if (MM >= 512 && MM <= 4096)                                                                           \
            hipLaunchKernelGGL(welch_bluestein_kernel<MM>, dim3(b.nblocks), dim3(BlueCfg<MM>::BLOCK), 0, s, b, n, win, tw, \
                               chirp, bhat);                                                                   \
        else                                                                                                   \
            hipLaunchKernelGGL(welch_bluestein_kernel_scalar<MM>, dim3(b.nblocks), dim3(BlueCfg<MM>::BLOCK), 0, s, b, n, win, tw, \
                               chirp, bhat);                                                                   \
        return hipGetLastError();
        PSDK_CASE(32)
        PSDK_CASE(64)
        PSDK_CASE(128)
        PSDK_CASE(256)
        PSDK_CASE(512)
        PSDK_CASE(1024)
        PSDK_CASE(2048)
        PSDK_CASE(4096)
        PSDK_CASE(8192)
        PSDK_CASE(16384)
#undef PSDK_CASE
    default:
        break;
    }
    bool any_fr = false;
    for (int j = 0; j < b.njobs; ++j)
        any_fr = any_fr || b.jobs[j].fspan >= 0;
    switch (n) {
#define PSDK_CASE(NN)                                                                            \
    case NN:                                                                                     \
        if (any_fr)                                                                              \
            hipLaunchKernelGGL((welch_kernel<NN, true>), dim3(b.nblocks), dim3(WelchCfg<NN>::BLOCK), 0, s, b, win, tw); \
        else                                                                                     \
            hipLaunchKernelGGL((welch_kernel<NN, false>), dim3(b.nblocks), dim3(WelchCfg<NN>::BLOCK), 0, s, b, win, tw); \
        break;
        PSDK_CASE(16)
        PSDK_CASE(32)
        PSDK_CASE(64)
        PSDK_CASE(128)
        PSDK_CASE(256)
        PSDK_CASE(512)
        PSDK_CASE(1024)
        PSDK_CASE(2048)
        PSDK_CASE(4096)
        PSDK_CASE(8192)
        PSDK_CASE(16384)
#undef PSDK_CASE
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_dec(const DecBatch &b, hipStream_t s)
{
    if (b.ntiles <= 0)
        return hipSuccess;
    bool any_fr = false;
    for (int j = 0; j < b.njobs; ++j)
        any_fr = any_fr || b.jobs[j].fspan >= 0;
    if (any_fr)
        hipLaunchKernelGGL(hbf_dec8_kernel<true>, dim3(b.ntiles), dim3(256), 0, s, b);
    else
        hipLaunchKernelGGL(hbf_dec8_kernel<false>, dim3(b.ntiles), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_post(const RedBatch &red, const TailBatch &tail, hipStream_t s)
{
    const int xb = (red.n / 2 + 1 + RED_BINS - 1) / RED_BINS;
    const int grid = red.njobs * xb + tail.njobs;
    if (grid <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(post_kernel, dim3(grid), dim3(RED_BINS * RED_SLICES), 0, s, red, tail, xb);
    return hipGetLastError();
}

// Read-out copy device -> pinned host memory by the shader cores (the host pointer is device
// visible): the copy engines the runtime would use for a D2H hipMemcpyAsync sometimes take
// milliseconds to wake after a compute-only stretch, which a read-out in a timed loop cannot afford.
__global__ __launch_bounds__(256) void copy_out_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t count)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = i; k < count; k += stride)
        dst[k] = src[k];
}

hipError_t launch_copy_out(float *h_dst, const float *d_src, size_t count, hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (count + 255) / 256);
    hipLaunchKernelGGL(copy_out_kernel, dim3(blocks), dim3(256), 0, s, h_dst, d_src, count);
    return hipGetLastError();
}

hipError_t launch_fill_noise(float *d_x, size_t len, uint64_t seed, uint64_t first, hipStream_t s)
{
    if (len == 0)
        return hipSuccess;
    size_t blocks = (len + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(fill_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_x, len, seed, first);
    return hipGetLastError();
}

hipError_t launch_adcdac(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches,
                         float *dst0, float *dst1, float *dst2, float *dst3, hipStream_t s)
{
    const size_t total = n_frames * (size_t)batches * 4;
    if (total == 0)
        return hipSuccess;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(adcdac_kernel, dim3((unsigned)blocks), dim3(256), 0, s, frames, frame_size,
                       n_frames, batches, dst0, dst1, dst2, dst3);
    return hipGetLastError();
}

// the 8 header bytes of every frame, gathered into pinned host memory (out[f] = {magic | id << 16 | batches << 24, seq}): the general
// device ingest validates them on the host (psdc_process_frames_device); one 8-byte load per frame when the frames are 8-byte aligned
__global__ __launch_bounds__(256) void header_gather_kernel(const uint8_t *__restrict__ frames, size_t frame_size, size_t n_frames, uint2 *out)
{
    const bool aligned = ((reinterpret_cast<uintptr_t>(frames) | frame_size) & 7u) == 0;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < n_frames; f += (size_t)gridDim.x * 256) {
        const uint8_t *p = frames + f * frame_size;
        uint2 h;
        if (aligned)
            h = *reinterpret_cast<const uint2 *>(p);
        else
            h = make_uint2((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24),
                           (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24));
        out[f] = h;
    }
}
hipError_t launch_header_gather(const uint8_t *frames, size_t frame_size, size_t n_frames, void *out_pinned, hipStream_t s)
{
    if (n_frames == 0)
        return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (n_frames + 255) / 256);
    hipLaunchKernelGGL(header_gather_kernel, dim3(blocks), dim3(256), 0, s, frames, frame_size, n_frames, static_cast<uint2 *>(out_pinned));
    return hipGetLastError();
}

hipError_t launch_payload(int fmt, const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, float *dst0, float *dst1,
                          float *dst2, float *dst3, hipStream_t s)
{
    const size_t total = n_frames * (size_t)batches;
    if (total == 0)
        return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(4096, (total + 255) / 256);
    if (fmt == 2)
        hipLaunchKernelGGL(payload_kernel<2>, dim3(blocks), dim3(256), 0, s, frames, frame_size, n_frames, batches, dst0, dst1, dst2, dst3);
    else if (fmt == 3)
        hipLaunchKernelGGL(payload_kernel<3>, dim3(blocks), dim3(256), 0, s, frames, frame_size, n_frames, batches, dst0, dst1, dst2, dst3);
    else if (fmt == 4)
        hipLaunchKernelGGL(payload_kernel<4>, dim3(blocks), dim3(256), 0, s, frames, frame_size, n_frames, batches, dst0, dst1, dst2, dst3);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace psdk
