// fused1024.hip -- the N = 1024 hot kernel: Hann window + two-for-one FFT +
// |Z|^2 accumulate (src/psd.rs:211-233) AND the /8 half-band decimation of the
// same samples (src/psd.rs:246-253) in one pass over the stream.
//
// One wavefront owns a run of consecutive segment pairs.  Per pair it
//   - receives 1024 new samples HBM -> VGPR as four dwordx4 per lane (the next
//     chunk is prefetched in registers while this one is processed),
//   - decimates the pair's 1024 new samples (+288 of history it already holds in
//     registers) through its private 8 KB LDS scratch: polyphase even/odd arrays,
//     two outputs per lane per step so every LDS access is an aligned 8-byte one,
//   - runs the (4,16,16) wave FFT of fft_wave1024.h through the same scratch,
//   - adds |Z|^2 into 16 registers per lane.
// Nothing but wave-level ordering is needed inside a run (LDS operations of one
// wavefront execute in order), so there is no workgroup barrier in the loop.
#include "fft_wave1024.h"
#include "hbf_taps.h"
#include "kernels.h"

namespace psdk {

namespace f1024 {
constexpr int NX = 1024 + HBF_HALO;          // 1312 stage inputs per pair
constexpr int NA = 512 + HBF_PRE_A;          // 650 A outputs
constexpr int NB = 256 + HBF_PRE_B;          // 314 B outputs
constexpr int X_SKIP = 512 - HBF_HALO;       // 224: first needed sample of the older chunk
// scratch layout (floats) inside the wave's 2176-float frame
constexpr int XE = 0, XO = NX / 2;           // 656 each
constexpr int AE = NX, AO = NX + 328;        // 325 each (NA / 2)
constexpr int BE = 0, BO = 160;              // 157 each (NB / 2), overlays XE once A is done
static_assert(AO + NA / 2 <= 2048 && BO + NB / 2 <= XO, "scratch layout");
// polyphase offsets (see kernels.hip namespace dec): out j -> even in[j+CE], odd in[j+CO+i], in[j+CO+2M-1-i]
constexpr int A_D = HBF_HALO / 2 - HBF_PRE_A; // 6
constexpr int A_CE = A_D - HBF_MA + 1, A_CO = A_D - 2 * HBF_MA + 1; // 4, 1
constexpr int B_D = HBF_PRE_A / 2 - HBF_PRE_B;                     // 11
constexpr int B_CE = B_D - HBF_MB + 1, B_CO = B_D - 2 * HBF_MB + 1; // 6, 0
constexpr int C_D = HBF_PRE_B / 2;                                 // 29
constexpr int C_CE = C_D - HBF_MC + 1, C_CO = C_D - 2 * HBF_MC + 1; // 15, 0
static_assert(A_CO == 1 && B_CO == 0 && C_CO == 0 && (A_CE % 2) == 0 && (B_CE % 2) == 0 && (C_CE % 2) == 1,
              "the aligned 8-byte read pattern below assumes these offsets");
} // namespace f1024

// pass 0 with the twiddles laid out [q][c][lane] so that a wavefront reads consecutive
// 8-byte words (conflict free): tw0t[((q-1)*4 + c)*64 + t] = W_1024^((4t + c) q)
__device__ __forceinline__ void pass0_t(int t, cf *v, const cf *tw0t)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cf b[4] = {v[c], v[4 + c], v[8 + c], v[12 + c]};
        Dft<4>::run(b);
        v[c] = b[0];
        v[4 + c] = cmul(b[1], tw0t[(0 * 4 + c) * 64 + t]);
        v[8 + c] = cmul(b[2], tw0t[(1 * 4 + c) * 64 + t]);
        v[12 + c] = cmul(b[3], tw0t[(2 * 4 + c) * 64 + t]);
    }
}

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wavefront execute in order; this only stops the
    // compiler from moving LDS accesses across the hand-off between lanes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct f2 {
    float x, y;
};
__device__ __forceinline__ f2 ld2(const float *p) { return *reinterpret_cast<const f2 *>(p); }

// two consecutive outputs (j, j+1), j even, of a half-band stage with M unique
// taps; odd array read as aligned pairs od[j + CO' ...], CO' even.
template <int M, int CE, int CO>
__device__ __forceinline__ void hbf_two(const float *__restrict__ ev, const float *__restrict__ od, int j,
                                        const float (&taps)[M], float &y0, float &y1)
{
    // output j needs od[j+CO .. j+CO+2M-1], output j+1 needs od[j+CO+1 .. j+CO+2M]
    constexpr int LO = CO & ~1;              // aligned start
    constexpr int CNT = (CO - LO) + 2 * M + 1; // values needed from od[j+LO]
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

__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
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

// DETREND: 0 None, 1 Midpoint, 2 Span, 3 Mean (src/psd.rs:75-113).  EWMA: per-segment
// amplitude sqrt(W) so that the two-for-one identity still yields the weighted sum.
// The Span / Mean / EWMA variants need a few more registers than the 128 that four
// wavefronts per SIMD allow; they are built for two per SIMD rather than spilling.
template <int DETREND, bool EWMA>
__global__ __launch_bounds__(FUSED_WAVES * 64, (DETREND >= 2 || EWMA) ? 2 : FUSED_WAVES_PER_SIMD) void fused1024_kernel(const FusedBatch batch,
                                                                       const float *__restrict__ win)
{
    using namespace w1024;
    using namespace f1024;
    __shared__ cf s_frames[FUSED_WAVES * FRAME];
    __shared__ cf s_tw0[TW0_SIZE];
    __shared__ cf s_tw1[TW1_SIZE];
    __shared__ float4 s_win[256]; // window, float4 piece m of lane t at [64 m + t]

    const int tid = threadIdx.x;
    const int t = tid & 63, wv = tid >> 6;

    // twiddle tables: W_1024^(s q) and W_256^(s q) (sincospi keeps them exact to f32 rounding)
    for (int i = tid; i < TW0_SIZE; i += FUSED_WAVES * 64) {
        const int q = i / 256 + 1, c = (i / 64) & 3, l = i & 63; // [q][c][lane]
        const int s = 4 * l + c;
        float sn, cs;
        sincospif(-2.0f * (float)((s * q) & 1023) / 1024.0f, &sn, &cs);
        s_tw0[i] = {cs, sn};
    }
    for (int i = tid; i < 256; i += FUSED_WAVES * 64)
        s_win[i] = *reinterpret_cast<const float4 *>(win + 4 * i); // (src/psd.rs:44-48 table)
    for (int i = tid; i < TW1_SIZE; i += FUSED_WAVES * 64) {
        const int q = i / 16 + 1, s = i % 16;
        float sn, cs;
        sincospif(-2.0f * (float)((s * q) & 255) / 256.0f, &sn, &cs);
        s_tw1[i] = {cs, sn};
    }

    int ji = 0;
    while (ji + 1 < batch.njobs && (int)blockIdx.x >= batch.jobs[ji + 1].block_begin)
        ++ji;
    const FusedJob &job = batch.jobs[ji];
    const int wb = blockIdx.x - job.block_begin;
    const int npairs = job.npairs, run = job.run;
    const int tile_pairs = FUSED_WAVES * run;
    const int ntiles = (npairs + tile_pairs - 1) / tile_pairs;

    cf *frame = s_frames + wv * FRAME;
    float *sf = reinterpret_cast<float *>(frame);

    const float ta[HBF_MA] = {PSDK_HBF_TAPS_A};
    const float tb[HBF_MB] = {PSDK_HBF_TAPS_B};
    const float tc[HBF_MC] = {PSDK_HBF_TAPS_C};

    float q[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        q[s] = 0.0f;

    __syncthreads(); // twiddle tables ready

    // One pair p: decimate its 1024 new samples, FFT it, accumulate.  Register groups of two
    // float4 each: lo/up = lower/upper half of chunk p, nl = lower half of chunk p + 1 (the
    // upper half of chunk p + 1 is not needed by pair p).  Once lo/up have been windowed into
    // the FFT registers they are dead: the upper half of chunk p + 1 is loaded into `up` and
    // the lower half of chunk p + 2 into `lo`, in flight during the FFT passes.  For pair
    // p + 1 the roles are (lo, up, nl) <- (nl, up, lo).
    auto pair_step = [&](float4(&lo)[2], float4(&up)[2], float4(&nl)[2], const float4 *cnext, bool more,
                         float *o, int p) {
        // ---- decimator: inputs rel r = 0..1311 <-> src[1024 p + 224 + r] ----
        if (t >= X_SKIP / 4) {
            const int h = 2 * (t - X_SKIP / 4);
            *reinterpret_cast<f2 *>(sf + XE + h) = {lo[0].x, lo[0].z};
            *reinterpret_cast<f2 *>(sf + XO + h) = {lo[0].y, lo[0].w};
        }
        {
            const float4 c1 = lo[1], c2 = up[0], c3 = up[1], n0 = nl[0], n1 = nl[1];
            const int h = 2 * t - X_SKIP / 2;
            *reinterpret_cast<f2 *>(sf + XE + h + 128) = {c1.x, c1.z};
            *reinterpret_cast<f2 *>(sf + XO + h + 128) = {c1.y, c1.w};
            *reinterpret_cast<f2 *>(sf + XE + h + 256) = {c2.x, c2.z};
            *reinterpret_cast<f2 *>(sf + XO + h + 256) = {c2.y, c2.w};
            *reinterpret_cast<f2 *>(sf + XE + h + 384) = {c3.x, c3.z};
            *reinterpret_cast<f2 *>(sf + XO + h + 384) = {c3.y, c3.w};
            *reinterpret_cast<f2 *>(sf + XE + h + 512) = {n0.x, n0.z};
            *reinterpret_cast<f2 *>(sf + XO + h + 512) = {n0.y, n0.w};
            *reinterpret_cast<f2 *>(sf + XE + h + 640) = {n1.x, n1.z};
            *reinterpret_cast<f2 *>(sf + XO + h + 640) = {n1.y, n1.w};
        }
        wave_sync();
        for (int u = t; u < NA / 2; u += 64) { // stage A: outputs 2u, 2u+1 -> AE[u], AO[u]
            float y0, y1;
            hbf_two<HBF_MA, A_CE, A_CO>(sf + XE, sf + XO, 2 * u, ta, y0, y1);
            sf[AE + u] = y0;
            sf[AO + u] = y1;
        }
        wave_sync();
        for (int u = t; u < NB / 2; u += 64) { // stage B
            float y0, y1;
            hbf_two<HBF_MB, B_CE, B_CO>(sf + AE, sf + AO, 2 * u, tb, y0, y1);
            sf[BE + u] = y0;
            sf[BO + u] = y1;
        }
        wave_sync();
        { // stage C: 128 outputs, two per lane
            float y0, y1;
            hbf_two<HBF_MC, C_CE, C_CO>(sf + BE, sf + BO, 2 * t, tc, y0, y1);
            o[2 * t] = y0;
            o[2 * t + 1] = y1;
        }
        wave_sync(); // scratch is reused by the FFT

        // ---- FFT of the pair: re = segment 2p, im = segment 2p + 1 ----
        cf v[16];
        {
            // segment a = samples [0, 1024) of (lo, up), segment b = (up, nl).  The trend is removed as
            // (x - o) - (m + n s): o is a sample of the segment (exact difference), the remainder is small,
            // so a DC level far above the noise does not cost the result its low bits.
            float oa = 0.0f, ob = 0.0f, sa = 0.0f, sb = 0.0f, ma = 0.0f, mb = 0.0f;
            if constexpr (DETREND == 1) { // Midpoint: x[N/2] (src/psd.rs:87-93)
                oa = lane_bcast(up[0].x, 0);
                ob = lane_bcast(nl[0].x, 0);
            } else if constexpr (DETREND == 2) { // Span (src/psd.rs:94-102), ramp as o + n s
                oa = lane_bcast(lo[0].x, 0);
                sa = (lane_bcast(up[1].w, 63) - oa) / 1023.0f;
                ob = lane_bcast(up[0].x, 0);
                sb = (lane_bcast(nl[1].w, 63) - ob) / 1023.0f;
            } else if constexpr (DETREND == 3) { // Mean (src/psd.rs:103-109)
                // summed about a pivot (the segment's midpoint sample) so that a large DC level
                // does not cost the f32 sum its low bits: mean = pivot + sum(x - pivot) / N
                const float pa = lane_bcast(up[0].x, 0), pb = lane_bcast(nl[0].x, 0);
                auto s4 = [](const float4 &v, float pv) { return ((v.x - pv) + (v.y - pv)) + ((v.z - pv) + (v.w - pv)); };
                const float sl = s4(lo[0], pa) + s4(lo[1], pa);
                const float sua = s4(up[0], pa) + s4(up[1], pa);
                const float sub = s4(up[0], pb) + s4(up[1], pb);
                const float sn = s4(nl[0], pb) + s4(nl[1], pb);
                oa = pa;
                ob = pb;
                ma = wave_sum(sl + sua) * (1.0f / 1024.0f);
                mb = wave_sum(sub + sn) * (1.0f / 1024.0f);
            }
            float ga = 1.0f, gb = 1.0f;
            if constexpr (EWMA) {
                if (job.ewma) {
                    ga = fused_ewma_amp(job, job.step0 + 2 * p);
                    gb = fused_ewma_amp(job, job.step0 + 2 * p + 1);
                }
            }
            const float nf = (float)(4 * t);
            auto put = [&](int slot, float xa, float xb, float w, int nofs) {
                if constexpr (DETREND == 1) {
                    xa -= oa;
                    xb -= ob;
                } else if constexpr (DETREND == 2) {
                    const float n = nf + (float)nofs;
                    xa = fmaf(-n, sa, xa - oa);
                    xb = fmaf(-n, sb, xb - ob);
                } else if constexpr (DETREND == 3) {
                    xa = (xa - oa) - ma;
                    xb = (xb - ob) - mb;
                }
                xa *= w;
                xb *= w;
                if constexpr (EWMA) {
                    xa *= ga;
                    xb *= gb;
                }
                v[slot] = {xa, xb};
            };
            const float4 w0 = s_win[t], w1 = s_win[64 + t], w2 = s_win[128 + t], w3 = s_win[192 + t];
            const float4 a0 = lo[0], a1 = lo[1], a2 = up[0], a3 = up[1], b2 = nl[0], b3 = nl[1];
            put(0, a0.x, a2.x, w0.x, 0);
            put(1, a0.y, a2.y, w0.y, 1);
            put(2, a0.z, a2.z, w0.z, 2);
            put(3, a0.w, a2.w, w0.w, 3);
            put(4, a1.x, a3.x, w1.x, 256);
            put(5, a1.y, a3.y, w1.y, 257);
            put(6, a1.z, a3.z, w1.z, 258);
            put(7, a1.w, a3.w, w1.w, 259);
            put(8, a2.x, b2.x, w2.x, 512);
            put(9, a2.y, b2.y, w2.y, 513);
            put(10, a2.z, b2.z, w2.z, 514);
            put(11, a2.w, b2.w, w2.w, 515);
            put(12, a3.x, b3.x, w3.x, 768);
            put(13, a3.y, b3.y, w3.y, 769);
            put(14, a3.z, b3.z, w3.z, 770);
            put(15, a3.w, b3.w, w3.w, 771);
        }
        if (more) { // pair p + 1 exists: chunk p + 1 upper -> up, chunk p + 2 lower -> lo
            up[0] = cnext[128];
            up[1] = cnext[192];
            lo[0] = cnext[256];
            lo[1] = cnext[256 + 64];
        }
        pass0_t(t, v, s_tw0);
        store0(t, v, frame);
        wave_sync();
        load1(t, v, frame);
        pass1(t, v, s_tw1);
        wave_sync();
        store1(t, v, frame);
        wave_sync();
        load2(t, v, frame);
        pass2(v);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
        wave_sync(); // next pair's decimator writes the scratch
    };

    for (int tile = wb; tile < ntiles; tile += job.nblocks) {
        const int p0 = tile * tile_pairs + wv * run;
        const int p1 = min(npairs, p0 + run);
        if (p0 >= p1)
            continue; // wave-uniform
        // chunk c = src[1024 c ..): lane holds float4 pieces m = 0..3 at 256 m + 4 t; pair p needs
        // chunk p and the lower half (m = 0, 1) of chunk p + 1, which is all a job guarantees.
        const float4 *cp = reinterpret_cast<const float4 *>(job.src) + (size_t)p0 * 256 + t;
        float4 ga[2], gb[2], gc[2];
        ga[0] = cp[0];
        ga[1] = cp[64];
        gb[0] = cp[128];
        gb[1] = cp[192];
        gc[0] = cp[256];
        gc[1] = cp[256 + 64];
        float *o = job.dst + (size_t)p0 * 128;
        for (int p = p0; p < p1; p += 2) {
            pair_step(ga, gb, gc, cp + 256, p + 1 < p1, o, p);
            cp += 256;
            o += 128;
            if (p + 1 < p1) {
                pair_step(gc, gb, ga, cp + 256, p + 2 < p1, o, p + 1);
                cp += 256;
                o += 128;
            }
        }
    }

    // combine the wavefronts; partial in natural bin order
#pragma unroll
    for (int s = 0; s < 16; ++s)
        sf[freq_of(t, s)] = q[s];
    __syncthreads();
    const float *all = reinterpret_cast<const float *>(s_frames);
    float *out = job.partial + (size_t)wb * N;
    for (int k = tid; k < N; k += FUSED_WAVES * 64) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < FUSED_WAVES; ++g)
            acc += all[g * 2 * FRAME + k];
        out[k] = acc;
    }
}

hipError_t launch_fused1024(const FusedBatch &b, const float *win, hipStream_t s)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    const dim3 grid(b.nblocks), block(FUSED_WAVES * 64);
#define PSDK_FUSED_CASE(D)                                                              \
    case D:                                                                             \
        if (b.any_ewma)                                                                 \
            hipLaunchKernelGGL((fused1024_kernel<D, true>), grid, block, 0, s, b, win);  \
        else                                                                            \
            hipLaunchKernelGGL((fused1024_kernel<D, false>), grid, block, 0, s, b, win); \
        break;
    switch (b.detrend) {
        PSDK_FUSED_CASE(0)
        PSDK_FUSED_CASE(1)
        PSDK_FUSED_CASE(2)
        PSDK_FUSED_CASE(3)
    default:
        return hipErrorInvalidValue;
    }
#undef PSDK_FUSED_CASE
    return hipGetLastError();
}

} // namespace psdk
