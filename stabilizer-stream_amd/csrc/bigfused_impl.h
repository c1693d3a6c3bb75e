// bigfused_impl.h -- the fused hot kernel for N = 2048 ... 16384 (Hann): the same single
// pass over the stream as fused.hip (detrend + window + two-for-one FFT + |Z|^2, and the /8
// half-band decimation of the same samples), with a whole workgroup of N/16 threads as the
// team: the (4, RA, RB, 16) FFT of fft_block.h, hand-offs by workgroup barriers, window and
// the two large twiddle tables read from global memory (L2 resident), the small one in LDS.
// Each bigfused_<N>.hip instantiates one size.
#pragma once
#include "fft_block.h"
#include "fused_common.h"

namespace psdk {

template <int N>
struct BigGeo : FusedDec<N> {
    using T = BlockFft<N>;
    static constexpr int TEAM = T::TEAM;
    static constexpr int WAVES = TEAM / 64;
    static constexpr int SCR = 2 * T::FRAME;
    static_assert(FusedDec<N>::END <= SCR && FusedDec<N>::WEND <= SCR, "decimator arrays exceed the frame");
};

// DETREND / EWMA as in fused.hip
// Built for two wavefronts per SIMD (the global window / twiddle loads are hoisted early and
// need the registers); N = 16384 has a 1024-thread workgroup and therefore 128 VGPRs at most.
template <int N, int DETREND, bool EWMA>
__global__ __launch_bounds__(N / 16, N < 16384 ? 2 : 4) void bigfused_kernel(const FusedBatch batch, const float *__restrict__ win,
                                                         const cf *__restrict__ tw0g, const cf *__restrict__ twag)
{
    using G = BigGeo<N>;
    using T = BlockFft<N>;
    constexpr int TEAM = G::TEAM;
    __shared__ cf s_frame[T::FRAME];
    __shared__ cf s_twb[T::TWB_SIZE];
    __shared__ float s_hist[G::HIST];
    __shared__ float s_red[2 * G::WAVES + 4];

    const int tl = threadIdx.x;
    for (int i = tl; i < T::TWB_SIZE; i += TEAM) { // [(q-1)][s]: W_SA^(s q)
        const int q = i / 16 + 1, s = i % 16;
        float sn, cs;
        sincospif(-2.0f * (float)(s * q) / (float)T::SA, &sn, &cs);
        s_twb[i] = {cs, sn};
    }

    int ji = 0;
    while (ji + 1 < batch.njobs && (int)blockIdx.x >= batch.jobs[ji + 1].block_begin)
        ++ji;
    const FusedJob &job = batch.jobs[ji];
    const int wb = blockIdx.x - job.block_begin;
    const int npairs = job.npairs, run = job.run;

    cf *frame = s_frame;
    float *sf = reinterpret_cast<float *>(s_frame);
    float *hs = s_hist;

    const float ta[HBF_MA] = {PSDK_HBF_TAPS_A};
    const float tb[HBF_MB] = {PSDK_HBF_TAPS_B};
    const float tc[HBF_MC] = {PSDK_HBF_TAPS_C};
    const unsigned h_pack = G::hist_slot(tl); // TEAM >= 128 > 80: one carried element per thread at most

    float q[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        q[s] = 0.0f;

    // this workgroup's run: pairs [p0, p0 + run) of the job (cut by npairs)
    const int p0 = wb * run;
    const int p1 = min(npairs, p0 + run);
    const float4 *cp = reinterpret_cast<const float4 *>(job.src) + (size_t)p0 * (N / 4) + tl;
    const float4 *safe = cp; // look-ahead target once nothing is left to look ahead to
    float4 ga[2], gb[2], gc[2];
    ga[0] = cp[0];
    ga[1] = cp[TEAM];
    gb[0] = cp[2 * TEAM];
    gb[1] = cp[3 * TEAM];
    gc[0] = cp[N / 4];
    gc[1] = cp[N / 4 + TEAM];

    // ---- warm-up: filter state at the first new sample of the run (hop >= 1024 > 288, so the
    // history is always inside the run's own first chunk) ------------------------------------
    {
        const float *xn = job.src + (size_t)p0 * N + N / 2;
        for (int r = tl; r < G::WX / 2; r += TEAM) {
            const int i0 = 2 * r - G::WX;
            sf[G::WXE + r] = xn[i0];
            sf[G::WXO + r] = xn[i0 + 1];
        }
        __syncthreads();
        for (int u = tl; u < G::WA / 2; u += TEAM) {
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
        for (int u = tl; u < G::WB / 2; u += TEAM) {
            float y0, y1;
            hbf_two<HBF_MB, G::B_CE, G::B_CO>(sf + G::WAE, sf + G::WAO, 2 * u, tb, y0, y1);
            hs[22 + u] = y0;
            hs[51 + u] = y1;
        }
        __syncthreads();
    }

    // one pair; register groups as in fused.hip: (lo, up) = chunk p, nl = lower half of chunk p + 1
    auto pair_step = [&](float4(&lo)[2], float4(&up)[2], float4(&nl)[2], const float4 *cnext, bool more,
                         float *o, int p) {
        // ---- decimator ------------------------------------------------------------------
        if ((h_pack & 0xFFFFu) != 0xFFFFu)
            sf[h_pack & 0xFFFFu] = hs[tl];
        // samples -> polyphase arrays as single floats (ds_write2_b32 from the registers the loads
        // filled; an 8-byte store of {x, z} would cost moves right behind the loads)
        auto split = [&](int h, const float4 &x) {
            sf[G::XE + h] = x.x;
            sf[G::XE + h + 1] = x.z;
            sf[G::XO + h] = x.y;
            sf[G::XO + h + 1] = x.w;
        };
        if (tl >= TEAM - 3)
            split(2 * (tl - (TEAM - 3)), lo[1]);
        {
            const int h = G::HX / 2 + 2 * tl;
            split(h, up[0]);
            split(h + N / 8, up[1]);
            split(h + N / 4, nl[0]);
            split(h + 3 * N / 8, nl[1]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) { // stage A
            const int u = tl + TEAM * r;
            float y0, y1;
            hbf_two<HBF_MA, G::A_CE, G::A_CO>(sf + G::XE, sf + G::XO, 2 * u, ta, y0, y1);
            sf[G::AE + 11 + u] = y0;
            sf[G::AO + 11 + u] = y1;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) { // stage B
            const int u = tl + TEAM * r;
            float y0, y1;
            hbf_two<HBF_MB, G::B_CE, G::B_CO>(sf + G::AE, sf + G::AO, 2 * u, tb, y0, y1);
            sf[G::BE + 29 + u] = y0;
            sf[G::BO + 29 + u] = y1;
        }
        __syncthreads();
        { // stage C: N/8 outputs, two per thread
            float y0, y1;
            hbf_two<HBF_MC, G::C_CE, G::C_CO>(sf + G::BE, sf + G::BO, 2 * tl, tc, y0, y1);
            o[2 * tl] = y0;
            o[2 * tl + 1] = y1;
        }
        if ((h_pack & 0xFFFFu) != 0xFFFFu)
            hs[tl] = sf[h_pack >> 16];

        // ---- detrend parameters (block-wide broadcast / reduction through LDS) -------------
        float oa = 0.0f, ob = 0.0f, ma = 0.0f, mb = 0.0f;
        slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
        if constexpr (DETREND == 1) { // the segments' midpoint samples
            if (tl == 0) {
                s_red[0] = up[0].x;
                s_red[1] = nl[0].x;
            }
        } else if constexpr (DETREND == 2) {
            if (tl == 0) {
                s_red[0] = lo[0].x;
                s_red[1] = up[0].x;
            }
            if (tl == TEAM - 1) {
                s_red[2] = up[1].w;
                s_red[3] = nl[1].w;
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
        if constexpr (DETREND == 3) { // Mean in two steps: o = f32 mean of the samples, m = mean of x - o
            // (see fused.hip: neither a rounded offset nor a sample pivot leaves bins 0 and 1 alone)
            auto block_sum2 = [&](float &pa, float &pb) { // both sums over the workgroup, same value in every thread
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) {
                    pa += __shfl_xor(pa, o2, 64);
                    pb += __shfl_xor(pb, o2, 64);
                }
                if ((tl & 63) == 0) {
                    s_red[4 + 2 * (tl >> 6)] = pa;
                    s_red[5 + 2 * (tl >> 6)] = pb;
                }
                __syncthreads();
                pa = 0.0f;
                pb = 0.0f;
#pragma unroll
                for (int w = 0; w < G::WAVES; ++w) {
                    pa += s_red[4 + 2 * w];
                    pb += s_red[5 + 2 * w];
                }
                __syncthreads(); // s_red is reused by the second sum
            };
            auto r4 = [](const float4 &x) { return (x.x + x.y) + (x.z + x.w); };
            const float rl = r4(lo[0]) + r4(lo[1]), ru = r4(up[0]) + r4(up[1]), rn = r4(nl[0]) + r4(nl[1]);
            float ra = rl + ru, rb = ru + rn;
            block_sum2(ra, rb);
            oa = ra * (1.0f / (float)N);
            ob = rb * (1.0f / (float)N);
            auto s4 = [](const float4 &x, float pv) { return ((x.x - pv) + (x.y - pv)) + ((x.z - pv) + (x.w - pv)); };
            float pa = s4(lo[0], oa) + s4(lo[1], oa) + s4(up[0], oa) + s4(up[1], oa);
            float pb = s4(up[0], ob) + s4(up[1], ob) + s4(nl[0], ob) + s4(nl[1], ob);
            block_sum2(pa, pb);
            ma = pa * (1.0f / (float)N);
            mb = pb * (1.0f / (float)N);
        }

        // ---- FFT of the pair ---------------------------------------------------------------
        cf v[16];
        {
            float ea = 1.0f, eb = 1.0f;
            if constexpr (EWMA) {
                if (job.ewma) {
                    ea = fused_ewma_amp(job, job.step0 + 2 * p);
                    eb = fused_ewma_amp(job, job.step0 + 2 * p + 1);
                }
            }
            const float nf = (float)(4 * tl);
            auto put = [&](int slot, float xa, float xb, float w, int nofs) {
                if constexpr (DETREND == 1) {
                    xa -= oa;
                    xb -= ob;
                } else if constexpr (DETREND == 2) {
                    const float n = nf + (float)nofs;
                    xa = fmaf(-n, sa.lo, fmaf(-n, sa.hi, xa - oa));
                    xb = fmaf(-n, sb.lo, fmaf(-n, sb.hi, xb - ob));
                } else if constexpr (DETREND == 3) {
                    xa = (xa - oa) - ma;
                    xb = (xb - ob) - mb;
                }
                xa *= w;
                xb *= w;
                if constexpr (EWMA) {
                    xa *= ea;
                    xb *= eb;
                }
                v[slot] = {xa, xb};
            };
            const float4 *wp = reinterpret_cast<const float4 *>(win) + tl;
            const float4 w0 = wp[0], w1 = wp[TEAM], w2 = wp[2 * TEAM], w3 = wp[3 * TEAM];
            const float4 a0 = lo[0], a1 = lo[1], a2 = up[0], a3 = up[1], b2 = nl[0], b3 = nl[1];
            put(0, a0.x, a2.x, w0.x, 0);
            put(1, a0.y, a2.y, w0.y, 1);
            put(2, a0.z, a2.z, w0.z, 2);
            put(3, a0.w, a2.w, w0.w, 3);
            put(4, a1.x, a3.x, w1.x, N / 4);
            put(5, a1.y, a3.y, w1.y, N / 4 + 1);
            put(6, a1.z, a3.z, w1.z, N / 4 + 2);
            put(7, a1.w, a3.w, w1.w, N / 4 + 3);
            put(8, a2.x, b2.x, w2.x, N / 2);
            put(9, a2.y, b2.y, w2.y, N / 2 + 1);
            put(10, a2.z, b2.z, w2.z, N / 2 + 2);
            put(11, a2.w, b2.w, w2.w, N / 2 + 3);
            put(12, a3.x, b3.x, w3.x, 3 * N / 4);
            put(13, a3.y, b3.y, w3.y, 3 * N / 4 + 1);
            put(14, a3.z, b3.z, w3.z, 3 * N / 4 + 2);
            put(15, a3.w, b3.w, w3.w, 3 * N / 4 + 3);
        }
        { // chunk p + 1 upper -> up, chunk p + 2 lower -> lo, in flight during the FFT; issued
          // unconditionally (after the last pair: re-reads of pieces read before, unused) so that
          // the compiler does not wait for them at the end of a branch
            const float4 *src = more ? cnext : safe;
            safe = src;
            up[0] = src[2 * TEAM];
            up[1] = src[3 * TEAM];
            lo[0] = src[N / 4];
            lo[1] = src[N / 4 + TEAM];
        }
        T::pass0(tl, v, tw0g);
        T::store0(tl, v, frame);
        __syncthreads();
        T::loadA(tl, v, frame);
        T::passA(tl, v, twag);
        T::storeA(tl, v, frame); // in place: each thread rewrites exactly what it read
        __syncthreads();
        T::loadB(tl, v, frame);
        T::passB(tl, v, s_twb);
        T::storeB(tl, v, frame);
        __syncthreads();
        T::loadC(tl, v, frame);
        T::passC(v);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
        __syncthreads(); // next pair's decimator writes the frame
    };

    {
        float *o = job.dst + (size_t)p0 * (N / 8);
        for (int p = p0; p < p1; p += 2) {
            pair_step(ga, gb, gc, cp + N / 4, p + 1 < p1, o, p);
            cp += N / 4;
            o += N / 8;
            if (p + 1 < p1) {
                pair_step(gc, gb, ga, cp + N / 4, p + 2 < p1, o, p + 1);
                cp += N / 4;
                o += N / 8;
            }
        }
    }

    // one team per workgroup: its accumulators are the partial
    float *out = job.partial + (size_t)wb * N;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        out[T::freq_of(tl, s)] = q[s];
}

template <int N>
hipError_t launch_bigfused_n(const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag, hipStream_t s)
{
    const dim3 grid(b.nblocks), block(N / 16);
#define PSDK_BIG_CASE(D)                                                                          \
    case D:                                                                                       \
        if (b.any_ewma)                                                                           \
            hipLaunchKernelGGL((bigfused_kernel<N, D, true>), grid, block, 0, s, b, win, tw0g, twag);  \
        else                                                                                      \
            hipLaunchKernelGGL((bigfused_kernel<N, D, false>), grid, block, 0, s, b, win, tw0g, twag); \
        break;
    switch (b.detrend) {
        PSDK_BIG_CASE(0)
        PSDK_BIG_CASE(1)
        PSDK_BIG_CASE(2)
        PSDK_BIG_CASE(3)
    default:
        return hipErrorInvalidValue;
    }
#undef PSDK_BIG_CASE
    return hipGetLastError();
}

} // namespace psdk
