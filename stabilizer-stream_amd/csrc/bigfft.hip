// bigfft.hip -- the generic Welch path for FFT sizes an LDS frame cannot hold: powers of two 32768 ... 131072.
//
// The reference plans ANY length (`FftPlanner::plan_fft_forward(N)`, src/psd.rs:417-418); what it can actually run is bounded
// by its own `[Complex<f32>; N]` / `[f32; N]` stack frames (src/psd.rs:75-80, 457-458: a few hundred KiB per call at N = 65536
// on a 2 MiB thread stack).  No BASELINE config uses these sizes; this path makes them RUN with the same semantics and the same
// 1e-5 parity: the segments of a job are processed pair by pair (two-for-one, as everywhere: z = x_a + i x_b, sum of the two
// segments' power = 1/2 (|Z[k]|^2 + |Z[N-k]|^2), folded by post_kernel) through a four-step transform N = N1 x 256 with ONE
// intermediate frame in global memory (a chunk of up to a thousand pairs per launch), natural-order output, |Z|^2 accumulated
// into the job's ONE partial row.  The decimator (hbf_dec8_kernel) and everything else are size-independent already.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
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

// ---- the four-step path: N = N1 x 256, two kernels a pair, the transposition done by which index each kernel walks ----
//
// Sample n = 256 n1 + n2, bin k = k1 + N1 k2:  Z[k] = sum_n2 W_256^(n2 k2) [ W_N^(n2 k1) sum_n1 z[256 n1 + n2] W_N1^(n1 k1) ].
// bigfft_col_kernel: a workgroup takes C adjacent columns n2 of one pair -- detrend, window and EWMA amplitude at the loads (C
// consecutive samples a row: 64 or 128 bytes), the first radix-16 pass over n1 straight from those registers, the rest of the
// length-N1 transform through one LDS frame [n1][c] -- and writes its part of T[n2 / 16][k1][n2 % 16], one contiguous run.
// bigfft_row_kernel: a workgroup takes sixteen rows k1 for a group of sixteen pairs: each lane keeps the sixteen W_N^(n2 k1) of
// its elements in registers for the whole group, reads its row elements (128-byte runs), does the 256-point transform over n2 as
// radix 16 x 16 with the one exchange inside its own 16-lane team, and accumulates |Z|^2 over the group in registers.
// ~20 bytes of traffic per sample (4 + 4 read, 8 written, 8 read; the transposed frame T never leaves the Infinity Cache at the
// smaller chunks) where the pass-by-pass path above moved ~(8 + 16 log16 N).
constexpr int BIG_N2 = 256;
constexpr int BIGFFT_PGROUP = 16;  // pairs whose |Z|^2 one row-kernel workgroup sums in registers
constexpr int BIG_MAX_PIECES = 32; // (job, pair range) pieces a chunk may hold: every job of a round shares the chunk's launches
struct BigPiece {
    SegJob job;
    int pair0, npairs; // pairs [pair0, pair0 + npairs) of the job ...
    int tpair0;        // ... sit at pairs [tpair0, ...) of the chunk's frame T,
    int group0;        // their groups of BIGFFT_PGROUP at gsum[group0 ...)
    int accumulate;    // 0: the job's first piece (the partial row is written, not added to)
    int pad;
};
struct BigChunk {
    int npieces, npairs, ngroups, pad;
    BigPiece pieces[BIG_MAX_PIECES];
};
__device__ __forceinline__ int big_piece_of_pair(const BigChunk &ch, int tq)
{
    int pi = 0;
    while (pi + 1 < ch.npieces && tq >= ch.pieces[pi + 1].tpair0)
        ++pi;
    return pi;
}
constexpr int BIG_RSTRIDE = 272; // LDS row stride of the row kernel: element 16 p + r of a row sits at 17 p + r

#ifndef PSDK_BIG_COLPAD
#define PSDK_BIG_COLPAD 1
#endif
template <int N1, int C>
struct ColFrame {
    // sixteen elements of padding per sixteen rows: the first pass's stores of a wavefront (rows 16 apart) spread over the banks.
    // (Without it the frame is 32 KiB and five workgroups fit a CU instead of four: measured -2 ... -4 %.)
    static constexpr int ELEMS = N1 * C + (PSDK_BIG_COLPAD ? (N1 / 16) * 16 : 0);
    static __device__ __forceinline__ int at(int n1, int c) { return n1 * C + c + (PSDK_BIG_COLPAD ? (n1 >> 4) * 16 : 0); }
};

// One in-LDS Stockham pass of radix R over the N1 index of every column of the tile (ns, s as in bigfft_pass_kernel); the LAST pass
// hands its outputs (k1 natural) to `out(k1, c, value)` instead of storing them back.
template <int N1, int C, int TH, int R, int NS, int S, bool LAST, typename Out>
__device__ __forceinline__ void col_lds_pass(cf *frame, const cf *__restrict__ tw, int n, int tid, Out out)
{
    using F = ColFrame<N1, C>;
    constexpr int NB = (N1 / R) * C, PER = NB / TH, M = NS / R;
    static_assert(NB % TH == 0, "butterflies per thread");
    cf v[PER][R];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = tid + i * TH, c = b % C, t = b / C;
        const int p = t / S, q = t % S;
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[i][r] = frame[F::at(q + S * (p + r * M), c)];
    }
    if (!LAST)
        __syncthreads(); // (in place: every read before any store)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = tid + i * TH, c = b % C, t = b / C;
        const int p = t / S, q = t % S;
        Dft<R>::run(v[i]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cf y = v[i][r];
            if (NS > R && r > 0)
                y = cmul(tw[(n / NS) * r * p], y); // W_NS^(r p)
            const int o = q + S * (R * p + r);
            if constexpr (LAST)
                out(o, c, y);
            else
                frame[F::at(o, c)] = y;
        }
    }
    if (!LAST)
        __syncthreads();
}

template <int N1, int C, int TH>
__global__ __launch_bounds__(TH) void bigfft_col_kernel(const BigChunk ch, int hop, int detrend, const float *__restrict__ win,
                                                                    const cf *__restrict__ tw, const double *__restrict__ means,
                                                                    cf *__restrict__ T)
{
    using F = ColFrame<N1, C>;
    constexpr int n = N1 * BIG_N2, M = N1 / 16, NB = M * C, PER = NB / TH;
    static_assert(NB % TH == 0, "butterflies per thread");
    __shared__ cf frame[F::ELEMS];
    // The workgroups of one pair (and of its neighbours, which share half their samples) on ONE XCD, whose L2 then holds the lines
    // they share: the dispatcher deals consecutive workgroup ids round the eight XCDs.
    const int tid = threadIdx.x, tiles = BIG_N2 / C;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
    const unsigned item = total % 8 == 0 ? (lin % 8) * (total / 8) + lin / 8 : lin;
    const int c0 = (int)(item % tiles) * C, pq = (int)(item / tiles); // pq: the pair's place in the chunk
    const BigPiece &pc = ch.pieces[big_piece_of_pair(ch, pq)];
    const SegJob &job = pc.job;
    const int la = 2 * (pc.pair0 + pq - pc.tpair0); // local index of segment a
    const bool act_b = la + 1 < job.nseg;
    const float *xa = job.src + ((job.seg0 + la) * (long long)hop - job.src_base);
    const float *xb = act_b ? xa + hop : xa; // (an odd last segment: b reads a's samples and is zeroed below)
    float oa = 0.0f, ob = 0.0f;
    double ma = 0.0, mb = 0.0;
    slope2 sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
    if (detrend == 1) { // Midpoint :87-93
        oa = xa[n / 2];
        ob = xb[n / 2];
    } else if (detrend == 2) { // Span :94-102
        oa = xa[0];
        sa = span_slope(oa, xa[n - 1], n);
        ob = xb[0];
        sb = span_slope(ob, xb[n - 1], n);
    } else if (detrend == 3) { // Mean :103-109 (bigfft_mean_kernel)
        ma = means[2 * pq];
        mb = means[2 * pq + 1];
    }
    float ampa = 1.0f, ampb = 1.0f;
    if (job.ewma) {
        ampa = big_ewma_amp(job, job.step0 + la);
        ampb = big_ewma_amp(job, job.step0 + la + 1);
    }
    if (!act_b)
        ampb = 0.0f;
    // pass 1 (radix 16, sub-transform length N1, stride 1): butterfly p of column c takes rows p + M r
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = tid + i * TH, c = b % C, p = b / C;
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = BIG_N2 * (p + M * r) + c0 + c;
            float a = xa[j], bb = xb[j];
            if (detrend == 1) {
                a -= oa;
                bb -= ob;
            } else if (detrend == 2) {
                const float f = (float)j;
                a = fmaf(-f, sa.lo, fmaf(-f, sa.hi, a - oa));
                bb = fmaf(-f, sb.lo, fmaf(-f, sb.hi, bb - ob));
            } else if (detrend == 3) {
                a = (float)((double)a - ma);
                bb = (float)((double)bb - mb);
            }
            const float w = win[j];
            v[r] = {a * w * ampa, act_b ? bb * w * ampb : 0.0f};
        }
        Dft<16>::run(v);
        frame[F::at(16 * p, c)] = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r)
            frame[F::at(16 * p + r, c)] = cmul(tw[BIG_N2 * r * p], v[r]); // W_N1^(r p)
    }
    __syncthreads();
    // T[pair][n2 / 16][k1][n2 % 16]: this workgroup's outputs are ONE contiguous run of N1 x 128 bytes per sixteen columns
    cf *Tp = T + (size_t)pq * n + (size_t)(c0 >> 4) * (N1 * 16);
    auto store = [&](int k1, int c, cf y) { Tp[(c >> 4) * (N1 * 16) + k1 * 16 + (c & 15)] = y; };
    if constexpr (N1 == 128)
        col_lds_pass<N1, C, TH, 8, 8, 16, true>(frame, tw, n, tid, store);
    else if constexpr (N1 == 256)
        col_lds_pass<N1, C, TH, 16, 16, 16, true>(frame, tw, n, tid, store);
    else {
        static_assert(N1 == 512, "N1");
        col_lds_pass<N1, C, TH, 16, 32, 16, false>(frame, tw, n, tid, store);
        col_lds_pass<N1, C, TH, 2, 2, 256, true>(frame, tw, n, tid, store);
    }
}

// the f64 means of the chunk's segments (detrend Mean): one workgroup a segment, means[2 pq + (0 | 1)]
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_mean_kernel(const BigChunk ch, int hop, int n, double *__restrict__ means)
{
    __shared__ double red[BIGFFT_THREADS / 64];
    const int tid = threadIdx.x, pq = blockIdx.x >> 1;
    const BigPiece &pc = ch.pieces[big_piece_of_pair(ch, pq)];
    const SegJob &job = pc.job;
    const int l = 2 * (pc.pair0 + pq - pc.tpair0) + (blockIdx.x & 1);
    if (l >= job.nseg) {
        if (tid == 0)
            means[blockIdx.x] = 0.0;
        return;
    }
    const float *x = job.src + ((job.seg0 + l) * (long long)hop - job.src_base);
    // sixteen independent loads a lane in flight (n is a multiple of 4096 here): the dependent chain of a plain loop is what this
    // kernel's time was (Mean at N = 131072 read 58 GS/s against 104 without detrend)
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j0 = 0; j0 < n; j0 += 16 * BIGFFT_THREADS) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            v[u] = x[j0 + u * BIGFFT_THREADS + tid];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            s4[u & 3] += (double)v[u];
    }
    double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
    if ((tid & 63) == 0)
        red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < BIGFFT_THREADS / 64; ++w)
            t += red[w];
        means[blockIdx.x] = t / (double)n;
    }
}

// rows k1 = 16 blockIdx.x ... + 15 of the pairs of group blockIdx.y (up to BIGFFT_PGROUP pairs of ONE piece): gsum[group][256 k1 + k2]
// = sum over the group of |Z[k1 + N1 k2]|^2, the additions in pair order
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_row_kernel(const BigChunk ch, const cf *__restrict__ T, const cf *__restrict__ tw,
                                                                    float *__restrict__ gsum, int n, int n1)
{
    __shared__ cf buf[16 * BIG_RSTRIDE];
    __shared__ cf w256[256];
    const int tid = threadIdx.x, p = tid & 15, rho = tid >> 4;
    const int k1 = 16 * blockIdx.x + rho, grp = blockIdx.y;
    w256[tid] = tw[n1 * (tid >> 4) * (tid & 15)]; // [r][p] = W_256^(r p)
    cf twd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        twd[r] = tw[k1 * (p + 16 * r)]; // W_N^(n2 k1), n2 = p + 16 r  (k1 n2 < N1 256 = N)
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = 0.0f;
    __syncthreads();
    int pi = 0;
    while (pi + 1 < ch.npieces && grp >= ch.pieces[pi + 1].group0)
        ++pi;
    const BigPiece &pc = ch.pieces[pi];
    const int q0 = pc.tpair0 + (grp - pc.group0) * BIGFFT_PGROUP, q1 = min(pc.tpair0 + pc.npairs, q0 + BIGFFT_PGROUP);
    cf *mine = buf + rho * BIG_RSTRIDE;
    const cf *row = T + (size_t)q0 * n + (size_t)k1 * 16 + p; // element n2 = p + 16 r: T[pair][r][k1][p], the workgroup's 16 rows 2 KiB runs
    const size_t rs = (size_t)n1 * 16;
    cf nx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        nx[r] = row[rs * r];
    for (int q = q0; q < q1; ++q) {
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = cmul(nx[r], twd[r]);
        if (q + 1 < q1) {
            row += n;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                nx[r] = row[rs * r];
        }
        Dft<16>::run(v); // pass 1 (sub-transform length 256, stride 1): outputs element 16 p + r, times W_256^(r p)
        mine[17 * p] = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r)
            mine[17 * p + r] = cmul(w256[16 * r + p], v[r]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = mine[17 * r + p]; // pass 2 (length 16, stride 16): butterfly p takes elements p + 16 r
        __syncthreads();
        Dft<16>::run(v);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[r] = fmaf(v[r].re, v[r].re, fmaf(v[r].im, v[r].im, acc[r])); // k2 = p + 16 r
    }
    float *g = gsum + (size_t)grp * n + (size_t)k1 * BIG_N2 + p;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        g[16 * r] = acc[r];
}

// the groups' sums of piece blockIdx.z into its job's partial row, natural bin order k = k1 + N1 k2 (post_kernel folds k with N - k):
// 16 x 16 tiles transposed through LDS, the groups added in their order (bit-reproducible, like the fused paths)
__global__ __launch_bounds__(BIGFFT_THREADS) void bigfft_fold_kernel(const BigChunk ch, const float *__restrict__ gsum, int n, int n1)
{
    __shared__ float tile[16][17];
    const BigPiece &pc = ch.pieces[blockIdx.z];
    const int groups = (pc.npairs + BIGFFT_PGROUP - 1) / BIGFFT_PGROUP;
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    const int k2_0 = 16 * blockIdx.x, k1_0 = 16 * blockIdx.y;
    const float *g = gsum + (size_t)pc.group0 * n + (size_t)(k1_0 + y) * BIG_N2 + k2_0 + x;
    float acc = 0.0f;
#pragma unroll 8
    for (int i = 0; i < groups; ++i)
        acc += g[(size_t)i * n];
    tile[y][x] = acc;
    __syncthreads();
    float *dst = pc.job.partial + (size_t)(k1_0 + x) + (size_t)n1 * (k2_0 + y);
    *dst = (pc.accumulate ? *dst : 0.0f) + tile[x][y];
}

hipError_t launch_welch_big(int n, const WelchBatch &b, const float *win, const cf *tw, cf *scratch, size_t scratch_elems, int chunk_limit,
                            hipStream_t s)
{
    // scratch (in complex elements): T [chunk][n] | gsum [chunk / 16 + BIG_MAX_PIECES][n] f32 | means [2 chunk] f64
    const int n1 = n / BIG_N2;
    const size_t fixed = (size_t)BIG_MAX_PIECES * n / 2 + 64;
    if (!bigfft_size(n) || !scratch || scratch_elems < fixed + 2 * ((size_t)n + n / 32 + 2))
        return hipErrorInvalidValue;
    int chunk_max = (int)std::min<size_t>(1024, (scratch_elems - fixed) / ((size_t)n + n / 32 + 2));
    if (chunk_limit > 0)
        chunk_max = std::min(chunk_max, chunk_limit);
    chunk_max = chunk_max >= BIGFFT_PGROUP ? chunk_max / BIGFFT_PGROUP * BIGFFT_PGROUP : chunk_max;
    cf *T = scratch;
    float *gsum = reinterpret_cast<float *>(scratch + (size_t)chunk_max * n);
    double *means = reinterpret_cast<double *>(gsum + (size_t)(chunk_max / BIGFFT_PGROUP + BIG_MAX_PIECES) * n);
    BigChunk ch{};
    auto flush = [&] {
        if (ch.npieces == 0)
            return;
        if (b.detrend == 3)
            hipLaunchKernelGGL(bigfft_mean_kernel, dim3(2 * ch.npairs), dim3(BIGFFT_THREADS), 0, s, ch, b.hop, n, means);
        auto col = [&](auto n1c, auto cc, auto thc) {
            constexpr int N1 = decltype(n1c)::value, C = decltype(cc)::value, TH = decltype(thc)::value;
            hipLaunchKernelGGL((bigfft_col_kernel<N1, C, TH>), dim3(BIG_N2 / C, ch.npairs), dim3(TH), 0, s, ch, b.hop, b.detrend, win, tw, means, T);
        };
        using std::integral_constant;
        // columns a workgroup x threads: ONE first-pass butterfly a thread (two at N1 = 512 measured -21 %); 32 columns with 512 threads
        // at N1 = 256 (128-byte instead of 64-byte row pieces) measured the same as 16 with 256
        if (n1 == 128)
            col(integral_constant<int, 128>{}, integral_constant<int, 32>{}, integral_constant<int, 256>{});
        else if (n1 == 256)
            col(integral_constant<int, 256>{}, integral_constant<int, 16>{}, integral_constant<int, 256>{});
        else
            col(integral_constant<int, 512>{}, integral_constant<int, 16>{}, integral_constant<int, 512>{});
        hipLaunchKernelGGL(bigfft_row_kernel, dim3(n1 / 16, ch.ngroups), dim3(BIGFFT_THREADS), 0, s, ch, T, tw, gsum, n, n1);
        hipLaunchKernelGGL(bigfft_fold_kernel, dim3(BIG_N2 / 16, n1 / 16, ch.npieces), dim3(BIGFFT_THREADS), 0, s, ch, gsum, n, n1);
        ch.npieces = ch.npairs = ch.ngroups = 0;
    };
    for (int ji = 0; ji < b.njobs; ++ji) {
        const SegJob &job = b.jobs[ji];
        if (job.fspan >= 0 || job.nblocks != 1)
            return hipErrorInvalidValue; // (frames are decoded into f32 streams at these sizes; one partial row per job)
        const int pairs = (job.nseg + 1) / 2;
        for (int done = 0; done < pairs;) {
            // a job split over chunks is split at a multiple of the group size: its additions keep their order whatever else is in the batch
            int take = std::min(pairs - done, chunk_max - ch.npairs);
            if (take < pairs - done && chunk_max >= BIGFFT_PGROUP)
                take = take / BIGFFT_PGROUP * BIGFFT_PGROUP;
            if (take <= 0 || ch.npieces == BIG_MAX_PIECES) {
                flush();
                continue;
            }
            BigPiece &pc = ch.pieces[ch.npieces++];
            pc.job = job;
            pc.pair0 = done;
            pc.npairs = take;
            pc.tpair0 = ch.npairs;
            pc.group0 = ch.ngroups;
            pc.accumulate = done > 0;
            ch.npairs += take;
            ch.ngroups += (take + BIGFFT_PGROUP - 1) / BIGFFT_PGROUP;
            done += take;
        }
    }
    flush();
    return hipGetLastError();
}

} // namespace psdk
