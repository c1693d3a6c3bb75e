// bigfft.hip -- the generic Welch path for FFT sizes an LDS frame cannot hold: powers of two 32768 ... 131072.
//
// The reference plans ANY length (`FftPlanner::plan_fft_forward(N)`, src/psd.rs:417-418); what it can actually run is bounded
// by its own `[Complex<f32>; N]` / `[f32; N]` stack frames (src/psd.rs:75-80, 457-458: a few hundred KiB per call at N = 65536
// on a 2 MiB thread stack).  No BASELINE config uses these sizes; this is the slow, simple path that makes them RUN with the
// same semantics and the same 1e-5 parity (a chunk of up to a few hundred pairs per pass launch, so that the passes are HBM-bound
// and not launch-bound): the segments of a job are processed pair by pair (two-for-one, as everywhere:
// z = x_a + i x_b, sum of the two segments' power = 1/2 (|Z[k]|^2 + |Z[N-k]|^2), folded by post_kernel) through a Stockham
// autosort FFT whose passes go through global memory -- radix-16 passes and one of radix 8 or 2 for what is left --, natural-order
// output, |Z|^2 accumulated into the job's ONE partial row.  The decimator (hbf_dec8_kernel) and everything else are size-
// independent already.  ~7 launches per chunk and ~(8 + 16 log16 N) bytes of traffic per sample: 17-23 GS/s measured (N = 131072 ...
// 32768), not hundreds.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fft_core.h"
#include "kernels.h"

namespace psdk {

constexpr int BIGFFT_THREADS = 256;

bool bigfft_size(int n) { return n > 16384 && n <= BIGFFT_MAX_N && (n & (n - 1)) == 0; }

__device__ __forceinline__ float big_ewma_amp(const SegJob &job, int step)
{
    // sqrt of W_step = gamma^max(0, nb - max(step, i_s - 1))  (plan.h), as in welch_kernel
    const int m = step > job.is_m1 ? step : job.is_m1;
    const int na = job.nb - m;
    if (na <= 0)
        return 1.0f;
    return (float)exp2(0.5 * (double)na * job.log2_gamma);
}

// pair q of the chunk (one workgroup): detrend (src/psd.rs:75-113) + window + EWMA amplitude of segments (a, b) into z[q][0..N)
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_load_kernel(const SegJob job, int hop, int detrend, int n,
                                                                     const float *__restrict__ win, cf *__restrict__ z, int pair0)
{
    __shared__ double red[2 * BIGFFT_THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int la = 2 * (pair0 + q); // local index of segment a
    const bool act_b = la + 1 < job.nseg;
    const float *xa = job.src + ((job.seg0 + la) * (long long)hop - job.src_base);
    const float *xb = xa + hop;
    float oa = 0.0f, ob = 0.0f;
    double ma = 0.0, mb = 0.0;
    slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
    if (detrend == 1) { // Midpoint :87-93
        oa = xa[n / 2];
        ob = act_b ? xb[n / 2] : 0.0f;
    } else if (detrend == 2) { // Span :94-102, the ramp as o + j (s_hi + s_lo) (fft_core.h span_slope)
        oa = xa[0];
        sa = span_slope(oa, xa[n - 1], n);
        if (act_b) {
            ob = xb[0];
            sb = span_slope(ob, xb[n - 1], n);
        }
    } else if (detrend == 3) { // Mean :103-109: the sums in f64 (the reference's sequential f32 sum is what the f32 oracle keeps)
        double pa = 0.0, pb = 0.0;
        for (int j = tid; j < n; j += BIGFFT_THREADS) {
            pa += (double)xa[j];
            if (act_b)
                pb += (double)xb[j];
        }
        for (int o = 32; o > 0; o >>= 1) {
            pa += __shfl_xor(pa, o);
            pb += __shfl_xor(pb, o);
        }
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = pa;
            red[2 * (tid >> 6) + 1] = pb;
        }
        __syncthreads();
        pa = pb = 0.0;
        for (int w = 0; w < BIGFFT_THREADS / 64; ++w) {
            pa += red[2 * w];
            pb += red[2 * w + 1];
        }
        ma = pa / (double)n;
        mb = pb / (double)n;
    }
    float ampa = 1.0f, ampb = 1.0f;
    if (job.ewma) {
        ampa = big_ewma_amp(job, job.step0 + la);
        ampb = big_ewma_amp(job, job.step0 + la + 1);
    }
    cf *out = z + (size_t)q * n;
    for (int j = tid; j < n; j += BIGFFT_THREADS) {
        float a = xa[j], b = act_b ? xb[j] : 0.0f;
        if (detrend == 1) {
            a -= oa;
            b -= ob;
        } else if (detrend == 2) {
            const float f = (float)j;
            a = fmaf(-f, sa.lo, fmaf(-f, sa.hi, a - oa));
            b = fmaf(-f, sb.lo, fmaf(-f, sb.hi, b - ob));
        } else if (detrend == 3) {
            a = (float)((double)a - ma);
            b = (float)((double)b - mb);
        }
        const float w = win[j];
        out[j] = {a * w * ampa, act_b ? b * w * ampb : 0.0f};
    }
}

// One Stockham pass of radix R over every pair of the chunk: sub-transform length ns, stride s (ns * s = n).  Butterfly t of a pair:
// p = t / s, q = t % s; inputs x[q + s (p + r m)], m = ns / R; outputs y[q + s (R p + r)] = W_ns^(r p) DFT_R(inputs)[r].
// tw[j] = W_n^j, so W_ns^(r p) = tw[r p s mod n].
template <int R>
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_pass_kernel(const cf *__restrict__ x, cf *__restrict__ y,
                                                                     const cf *__restrict__ tw, int n, int ns, int s, int npairs)
{
    const long long g = (long long)blockIdx.x * BIGFFT_THREADS + threadIdx.x;
    const int per = n / R;
    if (g >= (long long)per * npairs)
        return;
    const int pair = (int)(g / per), t = (int)(g % per);
    const int p = t / s, q = t % s, m = ns / R;
    const cf *xi = x + (size_t)pair * n + q + (size_t)s * p;
    cf *yo = y + (size_t)pair * n + q + (size_t)s * R * p;
    cf v[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        v[r] = xi[(size_t)s * m * r];
    Dft<R>::run(v); // natural-order outputs (fft_core.h)
    const int k = (int)(((long long)p * s) % n); // W_ns^p = tw[p s]
    yo[0] = v[0];
#pragma unroll
    for (int r = 1; r < R; ++r)
        yo[(size_t)s * r] = cmul(tw[(int)(((long long)k * r) % n)], v[r]);
}

// |Z|^2 of the chunk's pairs into the job's partial row (natural bin order; post_kernel folds k with N - k), in two steps with a
// fixed order of additions (bit-reproducible, like the fused paths): groups of 16 pairs are summed per bin into the frame the last
// pass left free, then the groups' sums are added to the row in group order.
constexpr int BIGFFT_PGROUP = 16;
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_power_kernel(const cf *__restrict__ z, float *__restrict__ gsum, int n, int npairs)
{
    const int k = blockIdx.x * BIGFFT_THREADS + threadIdx.x;
    const int grp = blockIdx.y;
    if (k >= n)
        return;
    const int q0 = grp * BIGFFT_PGROUP, q1 = min(npairs, q0 + BIGFFT_PGROUP);
    float acc = 0.0f;
    for (int q = q0; q < q1; ++q) {
        const cf v = z[(size_t)q * n + k];
        acc = fmaf(v.re, v.re, fmaf(v.im, v.im, acc));
    }
    gsum[(size_t)grp * n + k] = acc;
}
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_fold_kernel(const float *__restrict__ gsum, float *__restrict__ partial, int n,
                                                                     int groups, int accumulate)
{
    const int k = blockIdx.x * BIGFFT_THREADS + threadIdx.x;
    if (k >= n)
        return;
    float acc = accumulate ? partial[k] : 0.0f;
    for (int g = 0; g < groups; ++g)
        acc += gsum[(size_t)g * n + k];
    partial[k] = acc;
}

hipError_t launch_welch_big(int n, const WelchBatch &b, const float *win, const cf *tw, cf *scratch, size_t scratch_elems, hipStream_t s)
{
    if (!bigfft_size(n) || !scratch || scratch_elems < 2 * (size_t)n)
        return hipErrorInvalidValue;
    const int chunk_max = (int)std::min<size_t>(1024, scratch_elems / (2 * (size_t)n));
    cf *buf[2] = {scratch, scratch + (size_t)chunk_max * n};
    for (int ji = 0; ji < b.njobs; ++ji) {
        const SegJob &job = b.jobs[ji];
        if (job.fspan >= 0 || job.nblocks != 1)
            return hipErrorInvalidValue; // (frames are decoded into f32 streams at these sizes; one partial row per job)
        const int pairs = (job.nseg + 1) / 2;
        for (int p0 = 0; p0 < pairs; p0 += chunk_max) {
            const int np = std::min(chunk_max, pairs - p0);
            hipLaunchKernelGGL(bigfft_load_kernel, dim3(np), dim3(BIGFFT_THREADS), 0, s, job, b.hop, b.detrend, n, win, buf[0], p0);
            int cur = 0, ns = n, st = 1;
            auto pass = [&](auto radix) {
                constexpr int R = decltype(radix)::value;
                const long long work = (long long)(n / R) * np;
                hipLaunchKernelGGL(bigfft_pass_kernel<R>, dim3((unsigned)((work + BIGFFT_THREADS - 1) / BIGFFT_THREADS)), dim3(BIGFFT_THREADS), 0,
                                   s, buf[cur], buf[cur ^ 1], tw, n, ns, st, np);
                ns /= R;
                st *= R;
                cur ^= 1;
            };
            while (ns > 1) { // radix-16 passes, then whatever is left (32768 = 16^3 x 8, 65536 = 16^4, 131072 = 16^4 x 2)
                if (ns % 16 == 0)
                    pass(std::integral_constant<int, 16>{});
                else if (ns % 8 == 0)
                    pass(std::integral_constant<int, 8>{});
                else if (ns % 4 == 0)
                    pass(std::integral_constant<int, 4>{});
                else
                    pass(std::integral_constant<int, 2>{});
            }
            const unsigned kb = (unsigned)((n + BIGFFT_THREADS - 1) / BIGFFT_THREADS);
            const int groups = (np + BIGFFT_PGROUP - 1) / BIGFFT_PGROUP;
            float *gsum = reinterpret_cast<float *>(buf[cur ^ 1]); // (groups * n floats <= np * n complex elements: it fits)
            hipLaunchKernelGGL(bigfft_power_kernel, dim3(kb, (unsigned)groups), dim3(BIGFFT_THREADS), 0, s, buf[cur], gsum, n, np);
            hipLaunchKernelGGL(bigfft_fold_kernel, dim3(kb), dim3(BIGFFT_THREADS), 0, s, gsum, job.partial, n, groups, p0 > 0 ? 1 : 0);
        }
    }
    return hipGetLastError();
}

} // namespace psdk
