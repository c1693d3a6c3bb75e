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
// scratch layout (floats), 2048 per wave
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

__global__ __launch_bounds__(FUSED_WAVES * 64, 2) void fused1024_kernel(const FusedBatch batch,
                                                                       const float *__restrict__ win)
{
    using namespace w1024;
    using namespace f1024;
    __shared__ cf s_frames[FUSED_WAVES * N];
    __shared__ cf s_tw0[TW0_SIZE];
    __shared__ cf s_tw1[TW1_SIZE];

    const int tid = threadIdx.x;
    const int t = tid & 63, wv = tid >> 6;

    // twiddle tables: W_1024^(s q) and W_256^(s q) (sincospi keeps them exact to f32 rounding)
    for (int i = tid; i < TW0_SIZE; i += FUSED_WAVES * 64) {
        const int q = i / 256 + 1, s = i % 256;
        float sn, cs;
        sincospif(-2.0f * (float)((s * q) & 1023) / 1024.0f, &sn, &cs);
        s_tw0[i] = {cs, sn};
    }
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

    cf *frame = s_frames + wv * N;
    float *sf = reinterpret_cast<float *>(frame);

    // window of this lane's 16 FFT inputs n = 4t + c + 256m (src/psd.rs:44-48 table)
    float wn[16];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 w4 = *reinterpret_cast<const float4 *>(win + 256 * m + 4 * t);
        wn[4 * m + 0] = w4.x;
        wn[4 * m + 1] = w4.y;
        wn[4 * m + 2] = w4.z;
        wn[4 * m + 3] = w4.w;
    }
    const float ta[HBF_MA] = {PSDK_HBF_TAPS_A};
    const float tb[HBF_MB] = {PSDK_HBF_TAPS_B};
    const float tc[HBF_MC] = {PSDK_HBF_TAPS_C};

    float q[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        q[s] = 0.0f;

    __syncthreads(); // twiddle tables ready

    for (int tile = wb; tile < ntiles; tile += job.nblocks) {
        const int p0 = tile * tile_pairs + wv * run;
        const int p1 = min(npairs, p0 + run);
        if (p0 >= p1)
            continue; // wave-uniform
        // chunk q = src[1024 q ..): lane holds float4 pieces m = 0..3 at 256 m + 4 t.
        // The upper half (m = 2, 3) of chunk q exists only if pair q exists.
        const float4 *cp = reinterpret_cast<const float4 *>(job.src) + (size_t)p0 * 256 + t;
        float4 cur[4], nxt[4], pre[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            cur[m] = cp[64 * m];
        nxt[0] = cp[256 + 0];
        nxt[1] = cp[256 + 64];
        if (p0 + 1 < npairs) {
            nxt[2] = cp[256 + 128];
            nxt[3] = cp[256 + 192];
        } else {
            nxt[2] = nxt[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int p = p0; p < p1; ++p) {
            // prefetch chunk p + 2 for the next pair of this run
            if (p + 1 < p1) {
                const float4 *np = cp + (size_t)(p - p0 + 2) * 256;
                pre[0] = np[0];
                pre[1] = np[64];
                if (p + 2 < npairs) {
                    pre[2] = np[128];
                    pre[3] = np[192];
                } else {
                    pre[2] = pre[3] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }

            // ---- decimator: inputs rel r = 0..1311 <-> src[1024 p + 224 + r] ----
            // r of a float4 piece: cur m: 256 m + 4 t - 224; nxt m: 1024 + 256 m + 4 t - 224
            if (t >= X_SKIP / 4) {
                const int h = 2 * (t - X_SKIP / 4);
                *reinterpret_cast<f2 *>(sf + XE + h) = {cur[0].x, cur[0].z};
                *reinterpret_cast<f2 *>(sf + XO + h) = {cur[0].y, cur[0].w};
            }
#pragma unroll
            for (int m = 1; m < 4; ++m) {
                const int h = (256 * m - X_SKIP) / 2 + 2 * t;
                *reinterpret_cast<f2 *>(sf + XE + h) = {cur[m].x, cur[m].z};
                *reinterpret_cast<f2 *>(sf + XO + h) = {cur[m].y, cur[m].w};
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int h = (1024 + 256 * m - X_SKIP) / 2 + 2 * t;
                *reinterpret_cast<f2 *>(sf + XE + h) = {nxt[m].x, nxt[m].z};
                *reinterpret_cast<f2 *>(sf + XO + h) = {nxt[m].y, nxt[m].w};
            }
            wave_sync();
            // stage A: outputs j = 2u, 2u+1 -> AE[u], AO[u]
            for (int u = t; u < NA / 2; u += 64) {
                float y0, y1;
                hbf_two<HBF_MA, A_CE, A_CO>(sf + XE, sf + XO, 2 * u, ta, y0, y1);
                sf[AE + u] = y0;
                sf[AO + u] = y1;
            }
            wave_sync();
            // stage B
            for (int u = t; u < NB / 2; u += 64) {
                float y0, y1;
                hbf_two<HBF_MB, B_CE, B_CO>(sf + AE, sf + AO, 2 * u, tb, y0, y1);
                sf[BE + u] = y0;
                sf[BO + u] = y1;
            }
            wave_sync();
            // stage C: 128 outputs, two per lane
            {
                float y0, y1;
                hbf_two<HBF_MC, C_CE, C_CO>(sf + BE, sf + BO, 2 * t, tc, y0, y1);
                float *o = job.dst + (size_t)p * 128 + 2 * t;
                o[0] = y0;
                o[1] = y1;
            }
            wave_sync(); // scratch is reused by the FFT

            // ---- FFT of the pair: re = segment 2p, im = segment 2p + 1 ----
            cf v[16];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 a = cur[m];
                const float4 b = (m < 2) ? cur[m + 2] : nxt[m - 2];
                v[4 * m + 0] = {a.x * wn[4 * m + 0], b.x * wn[4 * m + 0]};
                v[4 * m + 1] = {a.y * wn[4 * m + 1], b.y * wn[4 * m + 1]};
                v[4 * m + 2] = {a.z * wn[4 * m + 2], b.z * wn[4 * m + 2]};
                v[4 * m + 3] = {a.w * wn[4 * m + 3], b.w * wn[4 * m + 3]};
            }
            pass0(t, v, s_tw0);
            store0(t, v, frame);
            wave_sync();
            load1(t, v, frame);
            pass1(t, v, s_tw1);
            wave_sync(); // all lanes have read before anyone overwrites in place
            store1(t, v, frame);
            wave_sync();
            load2(t, v, frame);
            pass2(v);
#pragma unroll
            for (int s = 0; s < 16; ++s)
                q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
            wave_sync(); // next pair's decimator writes the scratch

#pragma unroll
            for (int m = 0; m < 4; ++m) {
                cur[m] = nxt[m];
                nxt[m] = pre[m];
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
            acc += all[g * 2 * N + k];
        out[k] = acc;
    }
}

hipError_t launch_fused1024(const FusedBatch &b, const float *win, hipStream_t s)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(fused1024_kernel, dim3(b.nblocks), dim3(FUSED_WAVES * 64), 0, s, b, win);
    return hipGetLastError();
}

} // namespace psdk
