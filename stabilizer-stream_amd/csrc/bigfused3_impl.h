// bigfused3_impl.h -- the fused hot kernel for N = 2048 and 4096 (Hann) on the THREE-pass FFT of fft_block3.h: the same single
// pass over the stream as bigfused_impl.h (detrend + window + two-for-one FFT + |Z|^2, and the /8 half-band decimation of the
// same samples), with the stream arriving as sixteen coalesced dword loads per lane and pair instead of four 16-byte ones:
// lane tl holds samples tl + (N/16) j of each half chunk, which is what a radix-16 first pass needs in registers.  One LDS
// exchange, two workgroup barriers and ~32 LDS instructions fewer per pair than the four-pass kernel (+8 % at both sizes).
// FRAMES kernels: jobs whose stream is a trace of AdcDac frames read it in place, one 2-byte buffer load per sample here (a
// lane's samples are N/16 apart), raw in the register group until the loads have landed, then xor (DAC) + convert + scale.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "fft_block3.h"
#include "frames.h"
#include "fused_common.h"

#ifdef PSDK_ABL
#define PSDK_ABL3 PSDK_ABL
#else
#define PSDK_ABL3 0
#endif

namespace psdk {

template <int N>
struct Big3Geo : FusedDec<N> {
    using T = BlockFft3<N>;
    static constexpr int TEAM = T::TEAM;
    static constexpr int THREADS = TEAM;
    static constexpr int WAVES = THREADS / 64;
    static constexpr int WPS = BIG_WAVES_PER_SIMD;
    static constexpr int H = TEAM;       // lane stride within a half chunk: element j of a group is sample tl + H j
    static constexpr int SCR = 2 * T::FRAME;
    static_assert(FusedDec<N>::END <= SCR && FusedDec<N>::WEND <= SCR, "decimator arrays exceed the frame");
    static_assert(THREADS >= FusedDec<N>::HIST, "one carried filter-state element per thread at most");
};

// a register group: the lane's eight samples of one half chunk
struct g8 {
    float v[8];
};
// ... as the kernels carry it from pair to pair: f32 samples, or -- in the kernels that read AdcDac frames, between a load and its
// conversion -- eight raw 16-bit wire words, held as INTEGERS (fused_common.h Grp4 says why)
template <bool FRAMES>
struct Grp8;
template <>
struct Grp8<false> {
    float v[8];
    __device__ __forceinline__ float f(int j) const { return v[j]; }
    __device__ __forceinline__ void set(int j, float x) { v[j] = x; }
};
template <>
struct Grp8<true> {
    unsigned u[8];
    __device__ __forceinline__ float f(int j) const { return __builtin_bit_cast(float, u[j]); }
    __device__ __forceinline__ void set(int j, float x) { u[j] = __builtin_bit_cast(unsigned, x); }
    __device__ __forceinline__ void set_raw(int j, unsigned w) { u[j] = w; }
};

// Detrend + window + EWMA amplitude of one segment pair into the 16 FFT inputs of a lane (src/psd.rs:75-113, :211): slot m gets
// sample n = tl + (N/16) m of segment a = (lo, up) in .re and of segment b = (up, nl) in .im.  Mean: lo and up arrive with the
// pivot d.ob already subtracted (CENTRED, as in fused_common.h's window_pair), nl is raw.
template <int N, int DETREND, bool EWMA, bool SINGLE = false>
__device__ __forceinline__ void window_pair3(cf (&v)[16], int tl, const g8 &lo, const g8 &up, const g8 &nl, const float (&w)[16],
                                             const DetrendParams &d)
{
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        float xa = m < 8 ? lo.v[m & 7] : up.v[m & 7];
        float xb = m < 8 ? up.v[m & 7] : nl.v[m & 7];
        if constexpr (DETREND == 1) {
            xa -= d.oa;
            xb -= d.ob;
        } else if constexpr (DETREND == 2) {
            const float n = (float)(tl + (N / 16) * m);
            xa = fmaf(-n, d.sa.lo, fmaf(-n, d.sa.hi, xa - d.oa));
            xb = fmaf(-n, d.sb.lo, fmaf(-n, d.sb.hi, xb - d.ob));
        } else if constexpr (DETREND == 3) {
            xa -= d.ma;
            xb = m < 8 ? xb - d.mb : (xb - d.ob) - d.mb;
        }
        xa *= w[m];
        xb *= w[m];
        if constexpr (EWMA) {
            xa *= d.ea;
            xb *= d.eb;
        }
        if constexpr (SINGLE) // (overlap 0: segment b = the samples this step decimates -- fused_common.h window_pair)
            v[m] = {xb, 0.0f};
        else
            v[m] = {xa, xb};
    }
}

// SINGLE (overlap 0): 1 = one segment per transform, 2 = two disjoint segments per transform (fused.hip says how)
template <int N, int DETREND, bool EWMA, bool FRAMES = false, int SINGLE = 0>
__global__ __launch_bounds__(Big3Geo<N>::THREADS, Big3Geo<N>::WPS) void bigfused3_kernel(const FusedBatch batch,
                                                                           const float *__restrict__ win,
                                                                           const cf *__restrict__ tw0g)
{
    using G = Big3Geo<N>;
    using T = BlockFft3<N>;
    constexpr int TEAM = G::TEAM, THREADS = G::THREADS, H = G::H;
    __shared__ __attribute__((aligned(16))) cf s_frame[T::FRAME];
    __shared__ cf s_tw1[T::TW1_SIZE];
    __shared__ float s_hist[G::HIST];
    __shared__ __attribute__((aligned(8))) float s_red[2 * G::WAVES + 4];

    const int tp = threadIdx.x; // = the lane tl of the team
    for (int i = tp; i < T::TW1_SIZE; i += THREADS) { // [(q-1)][s]: W_L1^(s q)
        const int q = i / 16 + 1, s = i % 16;
        float sn, cs;
        sincospif(-2.0f * (float)(s * q) / (float)T::L1, &sn, &cs);
        s_tw1[i] = {cs, sn};
    }

    int bid = blockIdx.x;
    if constexpr (FRAMES) { // the four traces of a frame span on one XCD (see bigfused_impl.h)
        for (int g = 0; g < batch.n_fgroups; ++g) {
            const int b0 = batch.fg_begin[g], nb = batch.fg_nb[g];
            if (bid >= b0 && bid < b0 + 4 * nb) {
                const int p = bid - b0, full = (nb >> 3) * 32;
                int c, w;
                if (p < full) {
                    c = (p >> 3) & 3;
                    w = (p >> 5) * 8 + (p & 7);
                } else {
                    const int rem = nb & 7, q_ = p - full;
                    c = q_ / rem;
                    w = (nb & ~7) + q_ % rem;
                }
                bid = b0 + c * nb + w;
                break;
            }
        }
    }
    const int ji = job_of_unit(batch, bid, [](const FusedJob &j) { return j.block_begin; });
    const FusedJob &job = batch.jobs[ji];
    const int wb = bid - job.block_begin;
    const int npairs = job.npairs, run = job.run;

    cf *frame = s_frame;
    float *sf = reinterpret_cast<float *>(s_frame);
    float *hs = s_hist;

    const float ta[HBF_MA] = {PSDK_HBF_TAPS_A};
    const float tb[HBF_MB] = {PSDK_HBF_TAPS_B};
    const float tc[HBF_MC] = {PSDK_HBF_TAPS_C};
    const unsigned h_pack = G::hist_slot(tp);

    float q[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        q[s] = 0.0f;

    // this workgroup's run: pairs [p0, p1) of the job
    const int p0 = wb * run;
    const int p1 = min(npairs, p0 + run);
    const float *cp = job.src + (size_t)p0 * N + tp; // this lane's first sample of chunk p0
    const float *safe = cp;                           // look-ahead target once nothing is left to look ahead to
    // ... or, in a FRAMES launch, trace job.fch of a frame span (sp: the index of the same sample within the span)
    const bool fr = FRAMES && job.fspan >= 0;
    const FrameSpan &fsp = batch.fspans[fr ? job.fspan : 0];
    const unsigned ch_off = fr ? (unsigned)job.fch * 16u : 0u;
    const unsigned dac_flip = (fr && job.fch >= 2) ? 0xFFFF8000u : 0u; // on the sign-extended word
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(fsp.frames), 0, fr ? (int)min(fsp.bytes, 0x7FFFFFFFull) : 0, 0x00020000);
    unsigned sp = job.s_off + (unsigned)p0 * N + (unsigned)tp;
    unsigned safe_s = sp;
    // The eight samples of a group sit (H/8) j cells apart: cell c0 + (H/8) j = frame f0 + dq_j (+1), batch b0 + dr_j (- batches)
    // with (dq_j, dr_j) = divmod((H/8) j, batches) the same for every lane, and a wrap of the batch index into the next frame
    // costs exactly that frame's 8 header bytes (frame_size = 8 + 64 batches): ONE division per group and lane, then
    // offset_j = offset_0 + (dq_j frame_size + 64 dr_j) + (b0 + dr_j >= batches ? 8 : 0) -- four plain VALU operations a load.
    // (round 4: the wrap test as ONE compare of the lane's batch index with a uniform threshold, batches - dr_j, selecting between
    // offset_0 and offset_0 + 8, and the uniform part dq_j frame_size + 64 dr_j in the load's SCALAR offset field: two VALU
    // operations a load where there were four -- 64 of ~980 per pair)
    unsigned fthr[8], fdc[8];
    if constexpr (FRAMES) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned d = (unsigned)((H / 8) * j);
            const unsigned dq = fsp.batches == 1 ? d : __umulhi(d, fsp.magic);
            const unsigned dr = d - dq * fsp.batches;
            fthr[j] = fsp.batches - dr; // b0 + dr_j >= batches  <=>  b0 >= batches - dr_j   (dr_j < batches)
            fdc[j] = dq * fsp.frame_size + 64u * dr;
        }
    }
    using G8 = Grp8<FRAMES>;
    auto load8 = [&](G8 &g, const float *c, unsigned s_) { // a half chunk: samples c[H j] / samples s_ + H j of the trace (raw)
        if constexpr (FRAMES && (PSDK_ABL3 & 4096) != 0) { // timing only: the f32 kernel's coalesced dword loads on the frame bytes
            if (fr) {                                       // (garbage samples; the upper bound of what perfect wire loads buy)
                const unsigned *wsrc = reinterpret_cast<const unsigned *>(fsp.frames) + (s_ & 0xFFFFFFu);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    g.set_raw(j, wsrc[H * j] & 0x3F7FFFFFu); // (finite floats)
                return;
            }
        }
        if constexpr (FRAMES) {
            if (fr) {
                const unsigned c0 = s_ >> 3;
                const unsigned f0 = fsp.batches == 1 ? c0 : __umulhi(c0, fsp.magic);
                const unsigned b0 = c0 - __umul24(f0, fsp.batches);
                // (H is a multiple of 8: the same place in the cell for every j)
                const unsigned off0 = __umul24(f0, fsp.frame_size) + 8u + b0 * 64u + ch_off + (s_ & 7u) * 2u;
                const unsigned off8 = off0 + 8u; // (a wrap of the batch index into the next frame: that frame's 8 header bytes)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned off = b0 >= fthr[j] ? off8 : off0;
                    g.set_raw(j, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, off, fdc[j], 0));
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g.set(j, c[H * j]);
    };
    auto volts8 = [&](G8 &g) { // raw wire words -> volts, in place (a no-op for f32 jobs)
        if constexpr (FRAMES && (PSDK_ABL3 & 4096) == 0) {
            if (fr) {
                const float lsb = adcdac_lsb();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned w = (unsigned)(int)(short)(unsigned short)g.u[j] ^ dac_flip;
                    g.set(j, (float)(short)(unsigned short)w * lsb);
                }
            }
        }
    };
    auto samples = [](const G8 &g) { // the group's samples as f32
        g8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r.v[j] = g.f(j);
        return r;
    };
    G8 ga, gb, gc; // (lo, up) = chunk p, nl = lower half of chunk p + 1 (see pair_step)
    load8(ga, cp, sp);
    load8(gb, cp + N / 2, sp + N / 2);
    load8(gc, cp + N, sp + N);
    volts8(ga);
    volts8(gb);
    volts8(gc);

    // ---- warm-up: filter state at the first new sample of the run (hop >= 1024 > 288: inside the run's own first chunk) ----
    {
        const float *xn = job.src + (size_t)p0 * N + N / 2;
        for (int r = tp; r < G::WX / 2; r += THREADS) {
            const int i0 = 2 * r - G::WX;
            float e, o;
            if (fr) {
                const unsigned long long si = (unsigned long long)job.s_off + (unsigned long long)p0 * N + N / 2 + i0;
                e = frame_sample(fsp, job.fch, si);
                o = frame_sample(fsp, job.fch, si + 1);
            } else {
                e = xn[i0];
                o = xn[i0 + 1];
            }
            sf[G::WXE + r] = e;
            sf[G::WXO + r] = o;
        }
        __syncthreads();
        for (int u = tp; u < G::WA / 2; u += THREADS) {
            float y0, y1;
            hbf_two<HBF_MA, G::A_CE, G::A_CO>(sf + G::WXE, sf + G::WXO, 2 * u, ta, y0, y1);
            sf[G::WAE + u] = y0;
            sf[G::WAO + u] = y1;
            if (u >= G::WA / 2 - 11) {
                hs[u - (G::WA / 2 - 11)] = y0;
                hs[11 + u - (G::WA / 2 - 11)] = y1;
            }
        }
        __syncthreads();
        for (int u = tp; u < G::WB / 2; u += THREADS) {
            float y0, y1;
            hbf_two<HBF_MB, G::B_CE, G::B_CO>(sf + G::WAE, sf + G::WAO, 2 * u, tb, y0, y1);
            hs[22 + u] = y0;
            hs[51 + u] = y1;
        }
        __syncthreads();
    }

    auto sum8 = [](const g8 &g) { return ((g.v[0] + g.v[1]) + (g.v[2] + g.v[3])) + ((g.v[4] + g.v[5]) + (g.v[6] + g.v[7])); };
    auto sum8c = [](const g8 &g, float pv) {
        return (((g.v[0] - pv) + (g.v[1] - pv)) + ((g.v[2] - pv) + (g.v[3] - pv))) +
               (((g.v[4] - pv) + (g.v[5] - pv)) + ((g.v[6] - pv) + (g.v[7] - pv)));
    };
    // Mean (src/psd.rs:103-109) with a carried pivot, as in the other fused kernels
    float piv = 0.0f, s0c = 0.0f;
    if constexpr (DETREND == 3) {
        auto block_sum = [&](float v) {
            v = wave_sum64(v);
            __syncthreads();
            if ((tp & 63) == 0)
                s_red[4 + (tp >> 6)] = v;
            __syncthreads();
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < G::WAVES; ++w)
                t += s_red[4 + w];
            return t;
        };
        piv = block_sum(sum8(samples(ga)) + sum8(samples(gb))) * (1.0f / (float)N);
        s0c = block_sum(sum8c(samples(ga), piv));
        __syncthreads();
    }

    EwmaAmp eamp;
    if constexpr (EWMA) {
        if (job.ewma)
            eamp.init(job, job.step0 + (SINGLE != 0 ? 1 : 2) * p0);
    }
    // the lane's twiddle seeds, held across the run (opaque per pair: see bigfused_impl.h)
    const typename T::Seeds sd_run = T::load_seeds(tp, tw0g);

    float keep[16]; // SINGLE == 2: the windowed segment of the even step, until the odd step's transform
#pragma unroll
    for (int s = 0; s < 16; ++s)
        keep[s] = 0.0f;
    auto pair_step = [&](G8 &glo, G8 &gup, G8 &gnl, const float *cnext, unsigned snext, bool more, float *o, auto odd_step) {
        constexpr bool DOUBLE = SINGLE == 2, ODD = decltype(odd_step)::value;
        // the samples of this pair (converted at the end of the pair before): lo / up are dead once windowed -- Mean centres
        // these copies in place -- and their groups are reloaded further down
        g8 lo = samples(glo), up = samples(gup);
        const g8 nl = samples(gnl);
        const float *winp = win;
        {
            size_t zofs = 0; // the window loads stay inside the pair (bigfused_impl.h: hoisted, they pin 16 registers for the run)
            asm volatile("" : "+s"(zofs));
            winp += zofs;
        }
        // ---- decimator ---------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(PSDK_DEC_PRIO);
        // (an opaque asm on the packed slot per pair, so that the two unpacked LDS addresses are not kept -- and spilled -- as
        // loop invariants, measured 3-6 % SLOWER at N = 8192: the packed word itself is then what gets reloaded, behind a
        // vmcnt(0) at the top of the pair)
        const unsigned hpk = h_pack;
        if ((hpk & 0xFFFFu) != 0xFFFFu)
            sf[hpk & 0xFFFFu] = hs[tp];
        {
            // samples -> polyphase arrays: sample tl + H j of a half chunk is even / odd with tl (H is even); new sample i sits
            // at [HX/2 + i/2] of its array, the 12 samples before the new ones (the end of chunk p's lower half) in front
            float *arr = sf + ((tp & 1) ? G::XO : G::XE) + G::HX / 2 + (tp >> 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                arr[(H / 2) * j] = up.v[j];
                arr[N / 4 + (H / 2) * j] = nl.v[j];
            }
            if (tp >= TEAM - 12)
                sf[((tp & 1) ? G::XO : G::XE) + ((tp - (TEAM - 12)) >> 1)] = lo.v[7];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) { // stage A: N/2 outputs, four per step
            const int u = tp + THREADS * r;
            float y[4];
            hbf_four<HBF_MA, G::A_CE, G::A_CO, PSDK_HBF_WIDE_A != 0 && (G::XO % 4 == 0)>(sf + G::XE, sf + G::XO, 4 * u, ta, y);
            sf[G::AE + 11 + 2 * u] = y[0];
            sf[G::AO + 11 + 2 * u] = y[1];
            sf[G::AE + 12 + 2 * u] = y[2];
            sf[G::AO + 12 + 2 * u] = y[3];
        }
        __syncthreads();
        { // stage B: N/4 outputs, four per lane
            float y[4];
            hbf_four<HBF_MB, G::B_CE, G::B_CO, PSDK_HBF_WIDE != 0 && (G::AO % 4 == 0)>(sf + G::AE, sf + G::AO, 4 * tp, tb, y);
            sf[G::BE + 29 + 2 * tp] = y[0];
            sf[G::BO + 29 + 2 * tp] = y[1];
            sf[G::BE + 30 + 2 * tp] = y[2];
            sf[G::BO + 30 + 2 * tp] = y[3];
        }
        __syncthreads();
        f2 yc; // stage C: N/8 outputs, two per lane; stored further down
        hbf_two<HBF_MC, G::C_CE, G::C_CO>(sf + G::BE, sf + G::BO, 2 * tp, tc, yc.x, yc.y);
        if ((hpk & 0xFFFFu) != 0xFFFFu)
            hs[tp] = sf[hpk >> 16];
        __builtin_amdgcn_s_setprio(0);

        // ---- detrend parameters ------------------------------------------------------------
        DetrendParams dp;
        float &oa = dp.oa, &ob = dp.ob, &ma = dp.ma, &mb = dp.mb;
        slope2 &sa = dp.sa, &sb = dp.sb;
        if constexpr (DETREND == 1) { // the segments' midpoint samples x[N/2]: lane 0, element 8 (= element 0 of the upper half)
            if (tp == 0) {
                s_red[0] = up.v[0];
                s_red[1] = nl.v[0];
            }
        } else if constexpr (DETREND == 2) { // first (lane 0, element 0) and last (lane TEAM - 1, element 15) samples
            if (tp == 0) {
                s_red[0] = lo.v[0];
                s_red[1] = up.v[0];
            }
            if (tp == THREADS - 1) {
                s_red[2] = up.v[7];
                s_red[3] = nl.v[7];
            }
        }
        if constexpr (DETREND == 3) { // centre lo and up in place (dead after the window); partial sums of up, nl
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo.v[j] -= piv;
                up.v[j] -= piv;
            }
            float t1 = wave_sum64(sum8(up));
            float t2 = wave_sum64(sum8c(nl, piv)); // nl stays raw: it is the next pair's lo
            if ((tp & 63) == 0) {
                s_red[4 + 2 * (tp >> 6)] = t1;
                s_red[5 + 2 * (tp >> 6)] = t2;
            }
        }
        __syncthreads(); // the frame is reused by the FFT; s_red published
        if constexpr (DETREND == 1) {
            oa = s_red[0];
            ob = s_red[1];
        } else if constexpr (DETREND == 2) {
            oa = s_red[0];
            ob = s_red[1];
            sa = span_slope(oa, s_red[2], N);
            sb = span_slope(ob, s_red[3], N);
        }
        if constexpr (DETREND == 3) {
            // the wavefronts' partial sums: lane l reads those of wavefront l mod WAVES (one 8-byte read) and the group of
            // WAVES lanes adds them up with DPP -- a loop over s_red is 2 WAVES LDS reads a lane (32 at N = 16384)
            const f2 tw_ = ld2(s_red + 4 + 2 * (tp & (G::WAVES - 1)));
            const float s1 = group_sum<G::WAVES>(tw_.x), s2 = group_sum<G::WAVES>(tw_.y);
            ob = piv;
            ma = (s0c + s1) * (1.0f / (float)N);
            mb = (s1 + s2) * (1.0f / (float)N);
            const float pnext = piv + mb;
            s0c = fmaf(-(float)(N / 2), pnext - piv, s2);
            piv = pnext;
        }

        // ---- FFT of the pair ---------------------------------------------------------------
        cf vv[16];
        if constexpr (EWMA) {
            if (job.ewma) {
                if constexpr (SINGLE == 0)
                    dp.ea = eamp.next(job);
                dp.eb = eamp.next(job);
            }
        }
        {
            float w[16]; // the window of this lane: one batch of coalesced loads (L2 resident)
#pragma unroll
            for (int m = 0; m < 16; ++m)
                w[m] = (PSDK_ABL3 & 2048) ? 0.5f + dp.ea * (float)m : winp[tp + (N / 16) * m]; // (2048: timing only, no window loads)
            window_pair3<N, DETREND, EWMA, SINGLE != 0>(vv, tp, lo, up, nl, w, dp);
        }
        // chunk p + 1 upper -> up, chunk p + 2 lower -> lo, in flight during passes 1 and 2 and the next decimator's first
        // stage; issued unconditionally (after the last pair: re-reads of pieces read before, unused).  The decimator's
        // outputs leave here too (behind the window loads in the in-order counter).
        auto lookahead = [&] {
            *reinterpret_cast<f2 *>(o + 2 * tp) = yc;
            const float *src = more ? cnext : safe;
            const unsigned ssrc = more ? snext : safe_s;
            safe = src;
            safe_s = ssrc;
            load8(gup, src + N / 2, ssrc + N / 2);
            load8(glo, src + N, ssrc + N);
        };
        if constexpr (DOUBLE && !ODD) { // the even step keeps its windowed segment: no transform (and no frame use: the barrier
                                        // behind stage C already separates this pair's decimator from the next one's)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                keep[s] = vv[s].re;
            lookahead();
            return;
        }
        if constexpr (DOUBLE) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                vv[s] = {keep[s], vv[s].re};
        }
        {
            typename T::Seeds sd = sd_run;
            asm volatile("" : "+v"(sd.w1.re), "+v"(sd.w1.im), "+v"(sd.w4.re), "+v"(sd.w4.im));
            T::pass0(vv, sd);
        }
        T::store0(tp, vv, frame);
        lookahead();
        __syncthreads();
        T::load1(tp, vv, frame);
        T::pass1(tp, vv, s_tw1);
        T::store1(tp, vv, frame);
        __syncthreads();
        T::load2(tp, vv, frame);
        T::pass2(vv);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            q[s] = fmaf(vv[s].re, vv[s].re, fmaf(vv[s].im, vv[s].im, q[s]));
        if constexpr (FRAMES) { // the look-ahead groups hold raw wire words: to volts before the next pair reads them
            volts8(gup);
            volts8(glo);
        }
        __syncthreads(); // next pair's decimator writes the frame
    };

    {
        float *o = job.dst + (size_t)p0 * (N / 8);
        for (int p = p0; p < p1; p += 2) {
            pair_step(ga, gb, gc, cp + N, sp + N, p + 1 < p1, o, std::false_type{});
            cp += N;
            sp += N;
            o += N / 8;
            if (p + 1 < p1) {
                pair_step(gc, gb, ga, cp + N, sp + N, p + 2 < p1, o, std::true_type{});
                cp += N;
                sp += N;
                o += N / 8;
            }
        }
    }

    // one team per workgroup: its accumulators are the partial
    float *out = job.partial + (size_t)wb * N;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        out[T::freq_of(tp, s)] = q[s];
}

template <int N>
hipError_t launch_bigfused3_n(const FusedBatch &b, const float *win, const cf *tw0g, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    const dim3 grid(b.nblocks), block(Big3Geo<N>::THREADS);
    const bool ew_ = b.any_ewma || (dbg_variant() & 1), frm_ = b.any_frames || (dbg_variant() & 2);
#define PSDK_BIG3_CASE(D)                                                                         \
    case D:                                                                                       \
        if (frm_ && ew_)                                                                          \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, true, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g);  \
        else if (frm_)                                                                            \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, false, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g); \
        else if (ew_)                                                                             \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g);  \
        else                                                                                      \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, false>), grid, block, 0, s, ea, eb, 0, b, win, tw0g); \
        break;
#define PSDK_BIG3_SINGLE(D)                                                                                   \
    case D:                                                                                                   \
        if (b.single == 2 && ew_)                                                                             \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, true, false, 2>), grid, block, 0, s, ea, eb, 0, b, win, tw0g);  \
        else if (b.single == 2)                                                                               \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, false, false, 2>), grid, block, 0, s, ea, eb, 0, b, win, tw0g); \
        else if (ew_)                                                                                         \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, true, false, 1>), grid, block, 0, s, ea, eb, 0, b, win, tw0g);  \
        else                                                                                                  \
            hipExtLaunchKernelGGL((bigfused3_kernel<N, D, false, false, 1>), grid, block, 0, s, ea, eb, 0, b, win, tw0g); \
        break;
    if (b.single) {
        if (b.any_frames)
            return hipErrorInvalidValue;
        switch (b.detrend) {
            PSDK_BIG3_SINGLE(0)
            PSDK_BIG3_SINGLE(1)
            PSDK_BIG3_SINGLE(2)
            PSDK_BIG3_SINGLE(3)
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#undef PSDK_BIG3_SINGLE
    switch (b.detrend) {
        PSDK_BIG3_CASE(0)
        PSDK_BIG3_CASE(1)
        PSDK_BIG3_CASE(2)
        PSDK_BIG3_CASE(3)
    default:
        return hipErrorInvalidValue;
    }
#undef PSDK_BIG3_CASE
    return hipGetLastError();
}

} // namespace psdk
