// fused.hip -- the hot kernel for N = 256, 512, 1024 (Hann): detrend + window +
// two-for-one FFT + |Z|^2 accumulate (src/psd.rs:211-233) AND the /8 half-band
// decimation of the same samples (src/psd.rs:246-253) in one pass over the stream.
//
// A TEAM of N/16 lanes (1, 2 or 4 teams per wavefront) owns a run of consecutive
// segment pairs.  Pair p = segments (2p, 2p+1) of the job = samples
// src[N p .. N p + 3N/2); its N new samples src[N p + N/2 ..) are decimated to N/8
// outputs.  Per pair the team
//   - holds chunk p and the lower half of chunk p + 1 in registers (four dwordx4 per
//     lane per chunk, straight from HBM; the next ones are in flight during the FFT),
//   - runs the three half-band stages out of its private LDS frame: polyphase even/odd
//     arrays, two outputs per lane per step so that LDS accesses are aligned 8-byte
//     ones; the filter state (the last 22 stage-A and 58 stage-B outputs) is carried
//     from pair to pair in a small LDS side buffer, so nothing is recomputed,
//   - runs the (4, N/64, 16) team FFT of fft_team.h through the same frame,
//   - adds |Z|^2 into 16 registers per lane.
// LDS operations of one wavefront execute in order, so inside a run only compiler-level
// ordering is needed: there is no workgroup barrier in the loop.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <hip/hip_ext.h>

#include "fft_block.h"
#include "fft_team.h"
#include "frames.h"
#include "fused_common.h"

#ifndef PSDK_HOIST_LOOKAHEAD
#define PSDK_HOIST_LOOKAHEAD 1
#endif

namespace psdk {

// -DPSDK_STAMPS (tools/stamps): the first wavefront of workgroup 0 sums the s_memtime ticks it
// spends in each phase of pair_step into g_stamps; never defined in the shipped build.
#ifdef PSDK_STAMPS
__device__ unsigned long long g_stamps[16];
#define PSDK_STAMP(k)                                                \
    do {                                                             \
        if (stamp_on) {                                              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
            st[k] += t_ - tprev;                                     \
            tprev = t_;                                              \
        }                                                            \
    } while (0)
#else
#define PSDK_STAMP(k) \
    do {              \
    } while (0)
#endif

// -DPSDK_ABL=<bits>: TIMING-ONLY ablations (wrong results; tools/build_variants.sh), never in the shipped build:
//   1 no pass-1 twiddles   2 no carried-state copy in/out   4 no decimator   8 no FFT passes 1 and 2
//   16 no stage A   32 no stage C   64 no look-ahead loads   128 stage C computed but not stored
#ifndef PSDK_ABL
#define PSDK_ABL 0
#endif

template <int N>
struct FusedGeo : FusedDec<N> {
    using T = TeamFft<N>;
    static constexpr int TEAM = T::TEAM, TPW = T::TPW;
    static constexpr int TEAMS = FUSED_WAVES * TPW; // teams per workgroup
    static constexpr int SCR = 2 * T::FRAME;        // floats of a team's frame
    static_assert(FusedDec<N>::END <= SCR && FusedDec<N>::WEND <= SCR, "decimator arrays exceed the frame");
    static constexpr size_t LDS_BYTES = sizeof(cf) * (TEAMS * T::FRAME + T::TW0_SIZE + T::TW1_SIZE) +
                                        sizeof(float) * (N + TEAMS * FusedDec<N>::HIST);
};

// value of team-lane tl (a compile-time 0 or TEAM - 1 at the call sites) in every lane of the team
template <int TEAM>
__device__ __forceinline__ float team_bcast(float v, int team_lane0, int tl)
{
    if constexpr (TEAM == 64) // the team is the wavefront: a scalar read, no LDS-crossbar shuffle
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), tl));
    else
        return __shfl(v, team_lane0 + tl, 64);
}

// sum over the lanes of a team, the same value in every lane (row_sum16 / wave_sum64 of
// fused_common.h; only the 32-lane team needs one shuffle to join its two rows)
template <int TEAM>
__device__ __forceinline__ float team_sum(float v)
{
    if constexpr (TEAM == 64)
        return wave_sum64(v);
    v = row_sum16(v);
    if constexpr (TEAM == 32)
        v += __shfl_xor(v, 16, 64);
    return v;
}

// DETREND: 0 None, 1 Midpoint, 2 Span, 3 Mean (src/psd.rs:75-113).  EWMA: per-segment
// amplitude sqrt(W) so that the two-for-one identity still yields the weighted sum.
// The Span / Mean / EWMA variants need a few more registers than the 128 that four wavefronts
// per SIMD allow: they spill a few dwords outside the loop rather than halve the occupancy.
#ifndef PSDK_EWMA_WPS
#define PSDK_EWMA_WPS FUSED_WAVES_PER_SIMD // the EWMA variants too (6 dwords spilled outside the loop; 2 waves/SIMD read 10 % lower)
#endif
// FRAMES: as in bigfused_impl.h -- jobs with fspan >= 0 read their stream in place from AdcDac frames (8-byte buffer loads of
// four wire words, converted in the register group once the loads have landed); separate kernels, so the f32-only launches keep
// their instruction stream and registers.
// SINGLE: overlap 0 -- one segment per "pair" (FusedBatch::single; fused_common.h window_pair).  1: every segment is transformed
// by itself with a zero imaginary part.  2 (DOUBLE): the proper two-for-one of two DISJOINT segments -- step 2i windows segment
// 2i into sixteen kept registers, step 2i + 1 windows segment 2i + 1 into the imaginary parts and runs the ONE transform of
// both (|X_a[k]|^2 + |X_b[k]|^2 = 1/2 (|Z[k]|^2 + |Z[N-k]|^2), folded by post_kernel as ever): half the FFT work per sample of
// the Hann path; the decimator runs every step as before.  Jobs and runs hold an even number of segments (the planner sees to it).
#ifndef PSDK_FOLD
#define PSDK_FOLD 1 // one launch per round for the team kernels (planner.cpp run_launches); 0: the hooks compiled out (A/B)
#endif
#if PSDK_FOLD
// The aux workgroups of a one-launch round (kernels.h FusedAux): post_kernel's two roles on this kernel's workgroup size, with the
// team frames' LDS as the reduce's scratch.  The SAME sum as post_kernel's reduce_body, addition by addition -- 32 slices of the
// partial list (slice s: rows s, s + 32, ... in order), then the slices in order; half the bins a workgroup -- so a round gives the
// same bits through either (tests/test_gpu_parity.py test_one_launch_rounds_give_the_same_bits).  The rows of a slice are loaded
// four at a time and added one at a time: the loop is latency-bound and these workgroups hold a compute workgroup's slot.
template <int THREADS>
__device__ __forceinline__ void fused_aux_role(const FusedAux &aux, int b, double *scratch)
{
    constexpr int SLICES = 32, BINS = THREADS / SLICES; // (kernels.hip RED_SLICES)
    static_assert(BINS == AUX_RED_BINS, "the host sizes the aux grid by AUX_RED_BINS");
    const int n = aux.n, lane = threadIdx.x % BINS, slice = threadIdx.x / BINS;
    const int tall_blocks = aux.nred_tall * aux.red_xb, mid_blocks = aux.nred_mid * aux.red_mb;
    if (b < tall_blocks) { // a job of many rows (stage 0 of a long round): 16 bins a workgroup
        const RedJob &job = aux.red[b / aux.red_xb];
        const int k = (b % aux.red_xb) * BINS + lane;
        const bool live = k <= n / 2;
        double acc = 0.0;
        if (live) {
            const int km = k ? n - k : 0;
            int t = slice;
            for (; t + 3 * SLICES < job.nparts; t += 4 * SLICES) {
                float v[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float *p = job.partial + (size_t)(t + u * SLICES) * n;
                    v[u][0] = p[k];
                    v[u][1] = p[km];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc += (double)v[u][0] + (double)v[u][1];
            }
            for (; t < job.nparts; t += SLICES) {
                const float *p = job.partial + (size_t)t * n;
                acc += (double)p[k] + (double)p[km];
            }
        }
        scratch[slice * (BINS + 1) + lane] = acc;
        __syncthreads();
        if (slice == 0 && live) {
            acc = 0.0;
#pragma unroll
            for (int i = 0; i < SLICES; ++i)
                acc += scratch[i * (BINS + 1) + lane];
            job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc);
        }
        return;
    }
    b -= tall_blocks;
    if (b < mid_blocks) { // at most two rows a slice: AUX_MID_GROUPS groups of 16 bins a workgroup, every load in flight at once,
        const RedJob &job = aux.red[aux.nred_tall + b / aux.red_mb]; // then group g's slices are summed by the threads of slice g
        const int g0 = (b % aux.red_mb) * AUX_MID_GROUPS;
        float v[AUX_MID_GROUPS][2][2];
        const bool r0 = slice < job.nparts, r1 = slice + SLICES < job.nparts;
#pragma unroll
        for (int g = 0; g < AUX_MID_GROUPS; ++g) {
            const int k = (g0 + g) * BINS + lane;
            if (k <= n / 2) {
                const int km = k ? n - k : 0;
                if (r0) {
                    v[g][0][0] = job.partial[(size_t)slice * n + k];
                    v[g][0][1] = job.partial[(size_t)slice * n + km];
                }
                if (r1) {
                    v[g][1][0] = job.partial[(size_t)(slice + SLICES) * n + k];
                    v[g][1][1] = job.partial[(size_t)(slice + SLICES) * n + km];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < AUX_MID_GROUPS; ++g) {
            double acc = 0.0;
            if ((g0 + g) * BINS + lane <= n / 2) {
                if (r0)
                    acc += (double)v[g][0][0] + (double)v[g][0][1];
                if (r1)
                    acc += (double)v[g][1][0] + (double)v[g][1][1];
            }
            scratch[(g * SLICES + slice) * (BINS + 1) + lane] = acc;
        }
        __syncthreads();
        const int k = (g0 + slice) * BINS + lane;
        if (slice < AUX_MID_GROUPS && k <= n / 2) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < SLICES; ++i)
                acc += scratch[(slice * SLICES + i) * (BINS + 1) + lane];
            job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc);
        }
        return;
    }
    b -= mid_blocks;
    if (b < aux.nred - aux.nred_tall - aux.nred_mid) { // a job of few rows (a deep stage): one workgroup, a thread per bin; each row
        const RedJob &job = aux.red[aux.nred_tall + aux.nred_mid + b]; // is a slice of its own, so the sum over slices is the sum over rows
        for (int k = threadIdx.x; k <= n / 2; k += THREADS) {
            const int km = k ? n - k : 0;
            float v[AUX_SHORT_ROWS][2];
#pragma unroll
            for (int t = 0; t < AUX_SHORT_ROWS; ++t)
                if (t < job.nparts) {
                    v[t][0] = job.partial[(size_t)t * n + k];
                    v[t][1] = job.partial[(size_t)t * n + km];
                }
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < AUX_SHORT_ROWS; ++t)
                if (t < job.nparts)
                    acc += (double)v[t][0] + (double)v[t][1];
            job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc);
        }
        return;
    }
    const TailJob &job = aux.tail[b - (aux.nred - aux.nred_tall - aux.nred_mid)];
    for (int i = threadIdx.x; i < job.count; i += THREADS)
        job.dst[i] = job.src[i];
}
#define PSDK_FOLD_PARAM , const FusedAux aux
#ifndef PSDK_AUX_BACK
#define PSDK_AUX_BACK 0 // 1: the aux workgroups behind the compute workgroups in the grid instead of in front (A/B)
#endif
#ifndef PSDK_PRE_SCOPE
#define PSDK_PRE_SCOPE "workgroup"
#endif
#else
#define PSDK_FOLD_PARAM
#endif

template <int N, int DETREND, bool EWMA, bool FRAMES = false, int SINGLE = 0>
__global__ __launch_bounds__(FUSED_WAVES * 64, EWMA ? PSDK_EWMA_WPS : FUSED_WAVES_PER_SIMD) void fused_kernel(
    const FusedBatch batch, const float *__restrict__ win PSDK_FOLD_PARAM)
{
    using G = FusedGeo<N>;
    using T = TeamFft<N>;
    constexpr int TEAM = G::TEAM, TPW = G::TPW, TEAMS = G::TEAMS;
    __shared__ __attribute__((aligned(16))) cf s_frames[TEAMS * T::FRAME];
    __shared__ cf s_tw0[T::TW0_SIZE];
    __shared__ cf s_tw1[T::TW1_SIZE > 0 ? T::TW1_SIZE : 1];
    __shared__ float4 s_win[N / 4]; // window: float4 piece m of team-lane tl at [TEAM m + tl]
    __shared__ float s_hist[TEAMS * G::HIST];

#if PSDK_FOLD
#if PSDK_AUX_BACK
    if ((int)blockIdx.x >= batch.nblocks) { // (uniform: the whole workgroup takes the aux role and leaves)
        fused_aux_role<FUSED_WAVES * 64>(aux, (int)blockIdx.x - batch.nblocks, reinterpret_cast<double *>(s_frames));
        return;
    }
#else
    if ((int)blockIdx.x < aux.nblocks) { // (uniform: the whole workgroup takes the aux role and leaves)
        fused_aux_role<FUSED_WAVES * 64>(aux, (int)blockIdx.x, reinterpret_cast<double *>(s_frames));
        return;
    }
#endif
#endif
#ifdef PSDK_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tm = lane / TEAM, tl = lane % TEAM; // team within the wavefront, lane within the team
    const int team = wv * TPW + tm;               // team within the workgroup

    for (int i = tid; i < T::TW0_SIZE; i += FUSED_WAVES * 64) { // [c][tl]: W_N^(4 tl + c)
        const int c = i / TEAM, l = i % TEAM;
        float sn, cs;
        sincospif(-2.0f * (float)(4 * l + c) / (float)N, &sn, &cs);
        s_tw0[i] = {cs, sn};
    }
    for (int i = tid; i < T::TW1_SIZE; i += FUSED_WAVES * 64) { // [(q-1)][s]: W_L1^(s q)
        const int q = i / 16 + 1, s = i % 16;
        float sn, cs;
        sincospif(-2.0f * (float)(s * q) / (float)T::L1, &sn, &cs);
        s_tw1[i] = {cs, sn};
    }
    for (int i = tid; i < N / 4; i += FUSED_WAVES * 64)
        s_win[i] = *reinterpret_cast<const float4 *>(win + 4 * i); // (src/psd.rs:44-48 table)

#if PSDK_FOLD && !PSDK_AUX_BACK
    int bid = (int)blockIdx.x - aux.nblocks;
#else
    int bid = blockIdx.x;
#endif
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
#if PSDK_FOLD
    // copy prologue of a single-workgroup job (the seam: the head of a new span behind the carried tail): nobody else reads these
    // samples in this launch, so the hand-over is inside the workgroup -- stores, workgroup-scope release, the barrier below, acquire.
    // (Workgroup scope is enough because the wavefronts of a workgroup share one CU and its write-through vector L1 -- no
    // threadgroup-split mode here -- and it compiles to the waits alone; at agent scope each of these jobs wrote the XCD's L2 back
    // and invalidated its L1, and a round of 64 such jobs took 38 us instead of 17.)
    const bool has_pre = job.pre_count != 0;
    if (has_pre) {
        for (int q = 0; q < job.pre_count; ++q) {
            const TailJob &t = aux.tail[job.pre_first + q];
            for (int i = tid; i < t.count; i += FUSED_WAVES * 64)
                t.dst[i] = t.src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, PSDK_PRE_SCOPE);
    }
#endif

    cf *frame = s_frames + team * T::FRAME;
    float *sf = reinterpret_cast<float *>(frame);
    float *hs = s_hist + team * G::HIST; // [0,11) AE, [11,22) AO, [22,51) BE, [51,80) BO

    const float ta[HBF_MA] = {PSDK_HBF_TAPS_A};
    const float tb[HBF_MB] = {PSDK_HBF_TAPS_B};
    const float tc[HBF_MC] = {PSDK_HBF_TAPS_C};

    // where the carried filter state sits in the frame: tail (after a pair) and front (before one)
    // (packed: tail << 16 | front, 0xFFFF = none, to keep the register count at 128)
    constexpr int HR = (G::HIST + TEAM - 1) / TEAM;
    unsigned h_pack[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r)
        h_pack[r] = G::hist_slot(tl + TEAM * r);

    float q[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        q[s] = 0.0f;

    __syncthreads(); // tables ready
#if PSDK_FOLD
    if (has_pre)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, PSDK_PRE_SCOPE);
#endif
#ifdef PSDK_STAMPS
    const bool stamp_on = wb == 0 && run >= 8 && tid < 64; // first workgroup of a long-run job
    unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    st[9] = tprev - t_entry; // tables + window + job lookup
#endif

    // this team's run: pairs [p0, p0 + nrun) of the job (the job's last teams get fewer or none)
    const int p0 = (wb * TEAMS + team) * run;
    const int nrun = min(run, npairs - p0);
    const bool act0 = nrun > 0;
    const float4 *cp = reinterpret_cast<const float4 *>(job.src) + (size_t)p0 * (N / 4) + tl;
    // ... or, in a FRAMES launch, trace job.fch of a frame span (sp: the sample index of the same piece within the span)
    const bool fr = FRAMES && job.fspan >= 0;
    const FrameSpan &fsp = batch.fspans[fr ? job.fspan : 0];
    const unsigned ch_off = fr ? (unsigned)job.fch * 16u : 0u;
    const unsigned dac_flip = (fr && job.fch >= 2) ? 0x80008000u : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(fsp.frames), 0, fr ? (int)min(fsp.bytes, 0x7FFFFFFFull) : 0, 0x00020000);
    unsigned sp = job.s_off + (unsigned)p0 * N + 4u * tl;
    // piece k (float4 units from c / 4 k samples from s): f32 data, or the 8 raw bytes of the four wire words (integers:
    // fused_common.h Grp4)
    using G4 = Grp4<FRAMES>;
    auto piece = [&](const float4 *c, unsigned s_, int k) -> G4 {
        G4 g;
        if constexpr (FRAMES) {
            if (fr) {
                const unsigned si = s_ + 4u * (unsigned)k;
                const unsigned off = frame_cell_offset(fsp, si >> 3) + ch_off + (si & 4u) * 2u;
                const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0); // (a GCC-style vector of two u32: index it)
                g.set_raw((unsigned)r[0], (unsigned)r[1]);
                return g;
            }
        }
        g.set(c[k]);
        return g;
    };
    auto volts = [&](G4 &g) { // raw wire words -> volts, in place (a no-op for f32 jobs)
        if constexpr (FRAMES) {
            if (fr)
                grp_volts(g, dac_flip, adcdac_lsb());
        }
    };
    // where the look-ahead loads go once there is nothing left to look ahead to: pieces that were
    // read before (every job holds at least one pair = 3N/2 samples)
    const float4 *safe = act0 ? cp : reinterpret_cast<const float4 *>(job.src) + tl;
    unsigned safe_s = act0 ? sp : job.s_off + 4u * tl;
    G4 ga[2], gb[2], gc[2];
    {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        ga[0].set(z4), ga[1].set(z4), gb[0].set(z4), gb[1].set(z4), gc[0].set(z4), gc[1].set(z4);
    }
    if (act0) { // chunk p0 (all of it exists) and the lower half of chunk p0 + 1
        ga[0] = piece(cp, sp, 0);
        ga[1] = piece(cp, sp, TEAM);
        gb[0] = piece(cp, sp, 2 * TEAM);
        gb[1] = piece(cp, sp, 3 * TEAM);
        gc[0] = piece(cp, sp, N / 4);
        gc[1] = piece(cp, sp, N / 4 + TEAM);
        volts(ga[0]);
        volts(ga[1]);
        volts(gb[0]);
        volts(gb[1]);
        volts(gc[0]);
        volts(gc[1]);
    }

    // ---- warm-up: filter state at the first new sample of the run ------------------------
    // Evaluate stages A and B on the 288 samples before it (zeros before the start of the
    // stream, which is the reference's zero initial state, src/psd.rs:141) and keep their tails.
    {
        const float *xn = job.src + (size_t)p0 * N + N / 2; // first new sample of the run
        const long long lim = -(long long)job.pre - ((long long)p0 * N + N / 2); // xn[i] valid for i >= lim
        for (int r = tl; r < G::WX / 2; r += TEAM) {
            const int i0 = 2 * r - G::WX;
            float e = 0.0f, o = 0.0f;
            if (act0 && i0 >= lim) {
                if (fr) {
                    const unsigned long long si = (unsigned long long)((long long)job.s_off + (long long)p0 * N + N / 2 + i0);
                    e = frame_sample(fsp, job.fch, si);
                    o = frame_sample(fsp, job.fch, si + 1);
                } else {
                    e = xn[i0];
                    o = xn[i0 + 1];
                }
            }
            sf[G::WXE + r] = e;
            sf[G::WXO + r] = o;
        }
        wave_sync();
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
        wave_sync();
        for (int u = tl; u < G::WB / 2; u += TEAM) {
            float y0, y1;
            hbf_two<HBF_MB, G::B_CE, G::B_CO>(sf + G::WAE, sf + G::WAO, 2 * u, tb, y0, y1);
            hs[22 + u] = y0;
            hs[51 + u] = y1;
        }
        wave_sync();
    }

    // between pairs the carried state (element tl + TEAM r of the 80) lives in registers when that
    // takes few of them (N >= 512), else in the LDS side buffer
    constexpr bool HREG = HR <= 3;
    float hreg[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r)
        hreg[r] = (HREG && tl + TEAM * r < G::HIST) ? hs[tl + TEAM * r] : 0.0f;
    // Mean (src/psd.rs:103-109): the offset of a segment is pivot + m, the pivot a value close to the mean
    // (x - pivot is exact when a DC level dwarfs the noise, small otherwise) and m the mean of the residuals
    // (tiny, so its rounding does not matter; a single rounded offset would be a coherent error of half an
    // ulp of the mean over the segment, i.e. in bins 0 and 1).  The pivot is carried from pair to pair --
    // the mean found for the segment before -- so every sample is centred once and every half-segment
    // summed once: piv, and s0c = the sum of (lo - piv) over the team.
    float piv = 0.0f, s0c = 0.0f;
    if constexpr (DETREND == 3) {
        auto r4 = [](const float4 &x) { return (x.x + x.y) + (x.z + x.w); };
        piv = team_sum<TEAM>((r4(ga[0].f()) + r4(ga[1].f())) + (r4(gb[0].f()) + r4(gb[1].f()))) * (1.0f / (float)N);
        auto s4 = [](const float4 &x, float pv) { return ((x.x - pv) + (x.y - pv)) + ((x.z - pv) + (x.w - pv)); };
        s0c = team_sum<TEAM>(s4(ga[0].f(), piv) + s4(ga[1].f(), piv));
    }
    EwmaAmp eamp;
    if constexpr (EWMA) {
        if (job.ewma)
            eamp.init(job, job.step0 + (SINGLE != 0 ? 1 : 2) * p0);
    }
    PSDK_STAMP(10); // first loads issued + warm-up
    // One pair p.  Register groups of two float4 each: lo/up = lower/upper half of chunk p,
    // nl = lower half of chunk p + 1.  Once lo/up have been windowed into the FFT registers
    // they are dead: the upper half of chunk p + 1 is loaded into `up` and the lower half of
    // chunk p + 2 into `lo`, in flight during the FFT passes.  For pair p + 1 the roles are
    // (lo, up, nl) <- (nl, up, lo).
    float keep[16]; // DOUBLE: the windowed segment of the even step, until the odd step's transform
#pragma unroll
    for (int s = 0; s < 16; ++s)
        keep[s] = 0.0f;
    auto pair_step = [&](G4(&glo)[2], G4(&gup)[2], G4(&gnl)[2], const float4 *cnext, unsigned snext, bool more,
                         float *o, int p, auto odd_step) {
        constexpr bool DOUBLE = SINGLE == 2, ODD = decltype(odd_step)::value;
        // the samples of this pair (converted at the end of the pair before): lo / up are dead once windowed -- Mean centres
        // these copies in place -- and their groups are reloaded further down
        float4 lo[2] = {glo[0].f(), glo[1].f()}, up[2] = {gup[0].f(), gup[1].f()};
        const float4 nl[2] = {gnl[0].f(), gnl[1].f()};
        // ---- decimator ------------------------------------------------------------------
        // (a handful of VALU instructions between LDS round trips: at raised priority the wavefront
        // gets its few issue slots at once instead of queueing behind the butterflies of the other
        // wavefronts of the SIMD: +3-4 %)
        PSDK_STAMP(0);
#ifdef PSDK_STAMPS
        if constexpr (TEAM == 64) { // diagnostic: how long the wave waits for the look-ahead loads of this pair
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PSDK_STAMP(1);
        }
#endif
        __builtin_amdgcn_s_setprio(PSDK_DEC_PRIO);
        if constexpr (!(PSDK_ABL & 4)) {
        if constexpr (!(PSDK_ABL & 2)) {
#pragma unroll
        for (int r = 0; r < HR; ++r) // carried state -> fronts of the A and B arrays
            if ((h_pack[r] & 0xFFFFu) != 0xFFFFu)
                sf[h_pack[r] & 0xFFFFu] = HREG ? hreg[r] : hs[tl + TEAM * r];
        }
        if constexpr (PSDK_ABL & 16) {
        } else if constexpr (TEAM == 64) {
            // Stage A straight from the registers the loads filled: lane tl holds samples
            // 4tl..4tl+3 of each 256-sample piece = (xe[2tl], xo[2tl], xe[2tl+1], xo[2tl+1]); the
            // outputs (2tl, 2tl+1) of a piece need the three lanes below, fetched with DPP
            // wavefront shifts (lane 0.. of a piece continue into the top lanes of the piece before).
            // No LDS traffic for the input side of the stage.
            const float4 *pc[5] = {&lo[1], &up[0], &up[1], &nl[0], &nl[1]};
            float p1y = dpp_shr1(0.0f, pc[0]->y), p1w = dpp_shr1(0.0f, pc[0]->w); // only lane 63 is used
            float p2w = dpp_shr1(0.0f, p1w);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 &pv = *pc[r], &cu = *pc[r + 1];
                const float s1x = dpp_shr1(dpp_ror1(pv.x), cu.x), s1y = dpp_shr1(dpp_ror1(pv.y), cu.y);
                const float s1z = dpp_shr1(dpp_ror1(pv.z), cu.z), s1w = dpp_shr1(dpp_ror1(pv.w), cu.w);
                const float s2y = dpp_shr1(dpp_ror1(p1y), s1y), s2w = dpp_shr1(dpp_ror1(p1w), s1w);
                const float s3w = dpp_shr1(dpp_ror1(p2w), s2w);
                // out j = xe[j-2] + t0 (xo[j-5] + xo[j]) + t1 (xo[j-4] + xo[j-1]) + t2 (xo[j-3] + xo[j-2])
                float a0 = 0.0f, a1 = 0.0f;
                a0 += (s3w + cu.y) * ta[0];
                a1 += (s2y + cu.w) * ta[0];
                a0 += (s2y + s1w) * ta[1];
                a1 += (s2w + cu.y) * ta[1];
                a0 += (s2w + s1y) * ta[2];
                a1 += (s1y + s1w) * ta[2];
                const int u = tl + TEAM * r;
                sf[G::AE + 11 + u] = s1x + a0;
                sf[G::AO + 11 + u] = s1z + a1;
                p1y = s1y;
                p1w = s1w;
                p2w = s2w;
            }
        } else {
            // samples -> polyphase arrays as single floats: pairs of them leave as ds_write2_b32 from
            // whatever registers the loads delivered them to (an 8-byte store of {x, z} would need the
            // two in adjacent registers, i.e. moves right behind the loads)
            auto split = [&](int h, const float4 &x) {
                sf[G::XE + h] = x.x;
                sf[G::XE + h + 1] = x.z;
                sf[G::XO + h] = x.y;
                sf[G::XO + h + 1] = x.w;
            };
            if (tl >= TEAM - 3) // the 12 samples before the new ones (end of chunk p's lower half)
                split(2 * (tl - (TEAM - 3)), lo[1]);
            {
                const int h = G::HX / 2 + 2 * tl;
                split(h, up[0]);
                split(h + N / 8, up[1]);
                split(h + N / 4, nl[0]);
                split(h + 3 * N / 8, nl[1]);
            }
            wave_sync();
            PSDK_STAMP(1);
#pragma unroll
            for (int r = 0; r < 2; ++r) { // stage A: N/2 outputs, (4u .. 4u+3) -> AE/AO[11 + 2u, + 1]
                const int u = tl + TEAM * r;
                float y[4];
                hbf_four<HBF_MA, G::A_CE, G::A_CO, PSDK_HBF_WIDE_A != 0 && (G::XO % 4 == 0)>(sf + G::XE, sf + G::XO, 4 * u, ta, y);
                sf[G::AE + 11 + 2 * u] = y[0];
                sf[G::AO + 11 + 2 * u] = y[1];
                sf[G::AE + 12 + 2 * u] = y[2];
                sf[G::AO + 12 + 2 * u] = y[3];
            }
        }
        wave_sync();
        PSDK_STAMP(2);
        { // stage B: N/4 outputs, four per lane (hbf_four) -> BE/BO[29 + 2 tl, + 1]
            float y[4];
            hbf_four<HBF_MB, G::B_CE, G::B_CO, PSDK_HBF_WIDE != 0 && (G::AO % 4 == 0)>(sf + G::AE, sf + G::AO, 4 * tl, tb, y);
            sf[G::BE + 29 + 2 * tl] = y[0];
            sf[G::BO + 29 + 2 * tl] = y[1];
            sf[G::BE + 30 + 2 * tl] = y[2];
            sf[G::BO + 30 + 2 * tl] = y[3];
        }
        wave_sync();
        PSDK_STAMP(3);
        if constexpr (!(PSDK_ABL & 32)) { // stage C: N/8 outputs, two per lane, straight to the next stage's stream
            float y0, y1;
            hbf_two<HBF_MC, G::C_CE, G::C_CO>(sf + G::BE, sf + G::BO, 2 * tl, tc, y0, y1);
            if ((PSDK_ABL & 1024) ? job.npairs >= 0 : (!(PSDK_ABL & 128) || job.npairs < 0)) { // (1024: the store under an always-true branch the compiler cannot see through) // (128: the stage is computed, its outputs are not stored -- what the inter-stage WRITE costs)
                if constexpr (PSDK_ABL & 512) { // (512: a nontemporal store)
                    __builtin_nontemporal_store(y0, o + 2 * tl);
                    __builtin_nontemporal_store(y1, o + 2 * tl + 1);
                } else {
                    *reinterpret_cast<f2 *>(o + 2 * tl) = {y0, y1};
                }
            }
        }
        if constexpr (!(PSDK_ABL & 2)) {
#pragma unroll
        for (int r = 0; r < HR; ++r) // tails of A and B -> carried state
            if ((h_pack[r] & 0xFFFFu) != 0xFFFFu)
                (HREG ? hreg[r] : hs[tl + TEAM * r]) = sf[h_pack[r] >> 16];
        }
        } // PSDK_ABL & 4
        wave_sync(); // the frame is reused by the FFT
        PSDK_STAMP(4);
        __builtin_amdgcn_s_setprio(0);

        // ---- FFT of the pair: re = segment 2p, im = segment 2p + 1 -------------------------
        cf v[16];
        {
            // segment a = (lo, up), segment b = (up, nl).  The trend is removed as
            // (x - o) - (m + n s): o is a sample of the segment (exact difference), the remainder is
            // small, so a DC level far above the noise does not cost the result its low bits.
            DetrendParams dp;
            float &oa = dp.oa, &ob = dp.ob, &ma = dp.ma, &mb = dp.mb;
            slope2 &sa = dp.sa, &sb = dp.sb;
            const int l0 = lane - tl; // first lane of this team
            if constexpr (DETREND == 1) { // Midpoint: x[N/2] (src/psd.rs:87-93)
                oa = team_bcast<TEAM>(up[0].x, l0, 0);
                ob = team_bcast<TEAM>(nl[0].x, l0, 0);
            } else if constexpr (DETREND == 2) { // Span (src/psd.rs:94-102), ramp as o + n s
                oa = team_bcast<TEAM>(lo[0].x, l0, 0);
                sa = span_slope(oa, team_bcast<TEAM>(up[1].w, l0, TEAM - 1), N);
                ob = team_bcast<TEAM>(up[0].x, l0, 0);
                sb = span_slope(ob, team_bcast<TEAM>(nl[1].w, l0, TEAM - 1), N);
            } else if constexpr (DETREND == 3) { // Mean, pivot carried (see above)
                auto sub4 = [](float4 &x, float pv) {
                    x.x -= pv;
                    x.y -= pv;
                    x.z -= pv;
                    x.w -= pv;
                };
                sub4(lo[0], piv); // lo and up are dead after the window: centred in place
                sub4(lo[1], piv);
                sub4(up[0], piv);
                sub4(up[1], piv);
                auto r4 = [](const float4 &x) { return (x.x + x.y) + (x.z + x.w); };
                auto s4 = [](const float4 &x, float pv) { return ((x.x - pv) + (x.y - pv)) + ((x.z - pv) + (x.w - pv)); };
                const float s1 = team_sum<TEAM>(r4(up[0]) + r4(up[1]));
                const float s2 = team_sum<TEAM>(s4(nl[0], piv) + s4(nl[1], piv)); // nl stays raw: the next decimator reads it
                ob = piv;
                ma = (s0c + s1) * (1.0f / (float)N);
                mb = (s1 + s2) * (1.0f / (float)N);
                const float pnext = piv + mb; // the mean of segment b
                s0c = fmaf(-(float)(N / 2), pnext - piv, s2); // nl becomes lo: its sum about the new pivot
                piv = pnext;
            }
            if constexpr (EWMA) {
                if (job.ewma) {
                    if constexpr (SINGLE == 0)
                        dp.ea = eamp.next(job); // steps job.step0 + 2 p and + 1: the pairs of a run are consecutive
                    dp.eb = eamp.next(job);    // (SINGLE: one step per pair, segment b)
                }
            }
            window_pair<N, DETREND, EWMA, true, SINGLE != 0>(v, tl, lo[0], lo[1], up[0], up[1], nl[0], nl[1], s_win[tl], s_win[TEAM + tl],
                                          s_win[2 * TEAM + tl], s_win[3 * TEAM + tl], dp);
        }
        { // chunk p + 1 upper -> up, chunk p + 2 lower -> lo, in flight during the FFT.  Issued
          // unconditionally (after the last pair: re-reads, unused) -- a load under a branch is
          // waited for at the branch's end, which would expose the HBM latency once per pair.
            const float4 *src = more ? cnext : safe;
            const unsigned ssrc = more ? snext : safe_s;
            safe = src;
            safe_s = ssrc;
            if constexpr (!(PSDK_ABL & 64)) {
#if PSDK_HOIST_LOOKAHEAD
            // (the frame / f32 decision once for the four loads, as in bigfused_impl.h: a branch per load put each load behind
            // an s_waitcnt vmcnt(0) at its join)
            if constexpr (FRAMES) {
                if (fr) {
                    auto fp = [&](G4 &g, int k) {
                        const unsigned si = ssrc + 4u * (unsigned)k;
                        const unsigned off = frame_cell_offset(fsp, si >> 3) + ch_off + (si & 4u) * 2u;
                        const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
                        g.set_raw((unsigned)r[0], (unsigned)r[1]);
                    };
                    fp(gup[0], 2 * TEAM);
                    fp(gup[1], 3 * TEAM);
                    fp(glo[0], N / 4);
                    fp(glo[1], N / 4 + TEAM);
                } else {
                    gup[0].set(src[2 * TEAM]);
                    gup[1].set(src[3 * TEAM]);
                    glo[0].set(src[N / 4]);
                    glo[1].set(src[N / 4 + TEAM]);
                }
            } else {
                gup[0].set(src[2 * TEAM]);
                gup[1].set(src[3 * TEAM]);
                glo[0].set(src[N / 4]);
                glo[1].set(src[N / 4 + TEAM]);
            }
#else
            gup[0] = piece(src, ssrc, 2 * TEAM);
            gup[1] = piece(src, ssrc, 3 * TEAM);
            glo[0] = piece(src, ssrc, N / 4);
            glo[1] = piece(src, ssrc, N / 4 + TEAM);
#endif
            }
        }
        PSDK_STAMP(5);
#ifdef PSDK_EXTRA_VALU // sensitivity probe: dummy VALU work per pair (never defined in the shipped build)
        {
            float d[8] = {v[0].re, v[0].im, v[1].re, v[1].im, v[2].re, v[2].im, v[3].re, v[3].im};
#pragma unroll
            for (int e = 0; e < PSDK_EXTRA_VALU; ++e) // eight independent chains
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d[e & 7]) : "v"(v[4].re));
#pragma unroll
            for (int e = 0; e < 8; ++e)
                asm volatile("" ::"v"(d[e]));
        }
#endif
#ifdef PSDK_EXTRA_LDS // sensitivity probe: dummy 8-byte LDS reads per pair
        {
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < PSDK_EXTRA_LDS; ++e)
                d += lds_ld(frame + 17 * tl + (e & 15)).re;
            asm volatile("" ::"v"(d));
        }
#endif
        if constexpr (DOUBLE && !ODD) { // the even step keeps its windowed segment; no transform
#pragma unroll
            for (int s = 0; s < 16; ++s)
                keep[s] = v[s].re;
        } else {
        if constexpr (DOUBLE) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                v[s] = {keep[s], v[s].re};
        }
        T::pass0(tl, v, s_tw0);
        T::store0(tl, v, frame);
        wave_sync();
        PSDK_STAMP(6);
        if constexpr (!(PSDK_ABL & 8)) {
        T::load1(tl, v, frame);
        if constexpr (PSDK_ABL & 1)
            Dft<T::R1>::run(v);
        else
            T::pass1(tl, v, s_tw1);
        T::store1(tl, v, frame); // in place: each lane rewrites exactly what it read
        wave_sync();
        PSDK_STAMP(7);
        T::load2(tl, v, frame);
        T::pass2(v);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            q[s] = fmaf(v[s].re, v[s].re, fmaf(v[s].im, v[s].im, q[s]));
        } // (DOUBLE: odd step)
        if constexpr (FRAMES) { // the look-ahead groups hold raw wire words: to volts before the next pair reads them
            volts(gup[0]);
            volts(gup[1]);
            volts(glo[0]);
            volts(glo[1]);
        }
        wave_sync(); // next pair's decimator writes the frame
        PSDK_STAMP(8);
    };

    {
        float *o = job.dst + (size_t)p0 * (N / 8);
        if constexpr (PSDK_ABL & 2048) // (2048: the output stream rounded down to a 128-byte boundary -- what its alignment costs)
            o = reinterpret_cast<float *>(reinterpret_cast<uintptr_t>(o) & ~(uintptr_t)127);
        for (int i = 0; i < nrun; i += 2) {
            pair_step(ga, gb, gc, cp + N / 4, sp + N, i + 1 < nrun, o, p0 + i, std::false_type{});
            cp += N / 4;
            sp += N;
            if (!(PSDK_ABL & 256)) // (256: every pair of a run stores to the run's first N/8 outputs -- the store instructions without their traffic)
                o += N / 8;
            if (i + 1 < nrun) {
                pair_step(gc, gb, ga, cp + N / 4, sp + N, i + 2 < nrun, o, p0 + i + 1, std::true_type{});
                cp += N / 4;
                sp += N;
                if (!(PSDK_ABL & 256))
                    o += N / 8;
            }
        }
    }

    PSDK_STAMP(0);
    // combine the teams; partial in natural bin order
#pragma unroll
    for (int s = 0; s < 16; ++s)
        sf[T::freq_of(tl, s)] = q[s];
    __syncthreads();
    const float *all = reinterpret_cast<const float *>(s_frames);
    float *out = job.partial + (size_t)wb * N;
    for (int k = tid; k < N; k += FUSED_WAVES * 64) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < TEAMS; ++g)
            acc += all[g * G::SCR + k];
        out[k] = acc;
    }
#ifdef PSDK_STAMPS
    PSDK_STAMP(11); // team combine + partial store
    if (stamp_on && tid == 0) {
        for (int k = 0; k < 12; ++k)
            g_stamps[k] = st[k];
        g_stamps[12] = (unsigned long long)nrun;
    }
#endif
}

bool fused_supported(int n)
{
    return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384;
}

bool fused_frames_supported(int n) { return fused_supported(n); }

bool fused_fold_supported(int n) { return PSDK_FOLD != 0 && (n == 256 || n == 512 || n == 1024); }

// overlap 0: the sizes whose kernels have the DOUBLE form (FusedBatch::single == 2)
bool fused_double_supported(int n) { return fused_supported(n); }

int fused_pairs_per_block(int n, int run)
{
    switch (n) {
    case 256:
        return FusedGeo<256>::TEAMS * run;
    case 512:
        return FusedGeo<512>::TEAMS * run;
    case 1024:
        return FusedGeo<1024>::TEAMS * run;
    default:
        return fused_supported(n) ? run : 0; // one team (the whole workgroup) per workgroup
    }
}

int fused_block_threads(int n) { return n >= 2048 ? n / 16 : FUSED_WAVES * 64; }

// resident workgroups the launch is sized for (per CU: LDS and thread limits of each size)
int fused_max_blocks(int n)
{
    // workgroups per CU of the workgroup-level kernels: 4 SIMDs x wavefronts per SIMD / wavefronts per workgroup
    auto big = [](int threads, int wps) { return 256 * std::max(1, 4 * wps / (threads / 64)); };
    switch (n) {
    case 2048:
        return big(128, BIG_WAVES_PER_SIMD); // 17 KB of LDS each
    case 4096:
        return big(256, BIG_WAVES_PER_SIMD); // 35 KB
    case 8192:
        return big(512, BIG_WAVES_PER_SIMD); // 70 KB
    case 16384:
        return 256; // 1024 threads, 139 KB of LDS: one per CU
    default:
        return 256 * (4 * FUSED_WAVES_PER_SIMD / FUSED_WAVES); // 8 wavefronts, <= 80 KB: two per CU
    }
}

// global twiddle tables of the workgroup-level kernels: tw0[c * TEAM + tl] = W_N^(4 tl + c),
// twa[(q - 1) * SA + s] = W_(N/4)^(s q)
template <int N>
static void big_tables_n(std::vector<cf> &tw0, std::vector<cf> &twa)
{
    using T = BlockFft<N>;
    tw0.resize(T::TW0_SIZE);
    twa.resize(T::TWA_SIZE);
    for (int c = 0; c < 4; ++c)
        for (int tl = 0; tl < T::TEAM; ++tl) {
            const double a = -2.0 * M_PI * (double)(4 * tl + c) / (double)N;
            tw0[c * T::TEAM + tl] = {(float)cos(a), (float)sin(a)};
        }
    for (int q = 1; q < T::RA; ++q)
        for (int s2 = 0; s2 < T::SA; ++s2) {
            const double a = -2.0 * M_PI * (double)(s2 * q) / (double)T::L1;
            twa[(q - 1) * T::SA + s2] = {(float)cos(a), (float)sin(a)};
        }
}

void fused_big_tables(int n, std::vector<cf> &tw0, std::vector<cf> &twa)
{
    tw0.clear();
    twa.clear();
    switch (n) {
    case 2048:
        big_tables_n<2048>(tw0, twa);
        break;
    case 4096:
        big_tables_n<4096>(tw0, twa);
        break;
    case 8192:
        big_tables_n<8192>(tw0, twa);
        break;
    case 16384:
        big_tables_n<16384>(tw0, twa);
        break;
    default:
        break;
    }
}

void fused_big3_table(int n, std::vector<cf> &tw3)
{
    tw3.clear();
    if (n != 2048 && n != 4096)
        return;
    const int team = n / 16;
    tw3.resize(2 * (size_t)team);
    for (int tl = 0; tl < team; ++tl) {
        const double a = -2.0 * M_PI * (double)tl / (double)n;
        tw3[tl] = {(float)cos(a), (float)sin(a)};
        tw3[team + tl] = {(float)cos(4.0 * a), (float)sin(4.0 * a)};
    }
}

template <int N>
static hipError_t launch_fused_n(const FusedBatch &b, const float *win, hipStream_t s, hipEvent_t ea, hipEvent_t eb, const FusedAux *auxp)
{
#if PSDK_FOLD
    static const FusedAux no_aux{};
    const FusedAux &aux = auxp ? *auxp : no_aux;
#define PSDK_FOLD_ARG , aux
#else
    (void)auxp;
#define PSDK_FOLD_ARG
#endif

    static_assert(FusedGeo<N>::LDS_BYTES * (4 * FUSED_WAVES_PER_SIMD / FUSED_WAVES) <= 163840,
                  "the workgroups of a CU (two of eight waves by default) share its 160 KiB of LDS");
#if PSDK_FOLD
    const dim3 grid(b.nblocks + aux.nblocks), block(FUSED_WAVES * 64);
#else
    const dim3 grid(b.nblocks), block(FUSED_WAVES * 64);
#endif
    const bool ew_ = b.any_ewma || (dbg_variant() & 1), frm_ = b.any_frames || (dbg_variant() & 2);
#define PSDK_FUSED_CASE(D)                                                                \
    case D:                                                                               \
        if (frm_ && ew_)                                                                  \
            hipExtLaunchKernelGGL((fused_kernel<N, D, true, true>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG);  \
        else if (frm_)                                                                    \
            hipExtLaunchKernelGGL((fused_kernel<N, D, false, true>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG); \
        else if (ew_)                                                                     \
            hipExtLaunchKernelGGL((fused_kernel<N, D, true>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG);  \
        else                                                                              \
            hipExtLaunchKernelGGL((fused_kernel<N, D, false>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG); \
        break;
#define PSDK_FUSED_SINGLE(D)                                                                              \
    case D:                                                                                               \
        if (b.single == 2 && ew_)                                                                         \
            hipExtLaunchKernelGGL((fused_kernel<N, D, true, false, 2>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG);  \
        else if (b.single == 2)                                                                           \
            hipExtLaunchKernelGGL((fused_kernel<N, D, false, false, 2>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG); \
        else if (ew_)                                                                                     \
            hipExtLaunchKernelGGL((fused_kernel<N, D, true, false, 1>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG);  \
        else                                                                                              \
            hipExtLaunchKernelGGL((fused_kernel<N, D, false, false, 1>), grid, block, 0, s, ea, eb, 0, b, win PSDK_FOLD_ARG); \
        break;
    if (b.single) {
        if (b.any_frames)
            return hipErrorInvalidValue; // (frames are read in place by the half-overlap kernels only)
        switch (b.detrend) {
            PSDK_FUSED_SINGLE(0)
            PSDK_FUSED_SINGLE(1)
            PSDK_FUSED_SINGLE(2)
            PSDK_FUSED_SINGLE(3)
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#undef PSDK_FUSED_SINGLE
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

hipError_t launch_bigfused_2048(const FusedBatch &, const float *, const cf *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_bigfused_4096(const FusedBatch &, const float *, const cf *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_bigfused_8192(const FusedBatch &, const float *, const cf *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_bigfused_16384(const FusedBatch &, const float *, const cf *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_bigfused3_2048(const FusedBatch &, const float *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);
hipError_t launch_bigfused3_4096(const FusedBatch &, const float *, const cf *, hipStream_t, hipEvent_t, hipEvent_t);

hipError_t launch_fused(int n, const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag, const cf *tw3g, hipStream_t s,
                        hipEvent_t ea, hipEvent_t eb, const FusedAux *aux)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    // N = 2048 / 4096: the three-pass kernels (bigfused3_impl.h).  PSDC_FFT3=0 selects the four-pass kernels (A/B aid).
    static const bool fft3 = !(getenv("PSDC_FFT3") && getenv("PSDC_FFT3")[0] == '0');
    if (fft3 && tw3g) {
        if (n == 2048)
            return launch_bigfused3_2048(b, win, tw3g, s, ea, eb);
        if (n == 4096)
            return launch_bigfused3_4096(b, win, tw3g, s, ea, eb);
    }
    switch (n) {
    case 256:
        return launch_fused_n<256>(b, win, s, ea, eb, aux);
    case 512:
        return launch_fused_n<512>(b, win, s, ea, eb, aux);
    case 1024:
        return launch_fused_n<1024>(b, win, s, ea, eb, aux);
    case 2048:
        return launch_bigfused_2048(b, win, tw0g, twag, s, ea, eb);
    case 4096:
        return launch_bigfused_4096(b, win, tw0g, twag, s, ea, eb);
    case 8192:
        return launch_bigfused_8192(b, win, tw0g, twag, s, ea, eb);
    case 16384:
        return launch_bigfused_16384(b, win, tw0g, twag, s, ea, eb);
    default:
        return hipErrorInvalidValue;
    }
}

} // namespace psdk

#ifdef PSDK_STAMPS
extern "C" int psdc_debug_stamps(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(psdk::g_stamps), 16 * sizeof(unsigned long long));
}
#endif
