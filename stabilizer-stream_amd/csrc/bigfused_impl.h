// bigfused_impl.h -- the fused hot kernel for N = 2048 ... 16384 (Hann): the same single
// pass over the stream as fused.hip (detrend + window + two-for-one FFT + |Z|^2, and the /8
// half-band decimation of the same samples), with a whole workgroup as the team: the
// (4, RA, RB, 16) FFT of fft_block.h over N/16 "lanes" of 16 elements, hand-offs by workgroup
// barriers, window and the two large twiddle tables read from global memory (L2 resident), the
// small one in LDS.  A thread plays VT lanes (tl = thread + THREADS v).  VT = 1 everywhere now; VT = 2
// (512 threads with 32 elements each) was how N = 16384 fitted while the table loads were hoisted out of the
// pair loop -- 1024 threads are capped at 128 VGPRs (16 wavefronts on one CU) and spilled ~450 bytes per lane
// then -- and stays available (-DPSDK_BIG16K_VT=2).  Each bigfused_<N>.hip instantiates one size.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "fft_block.h"
#include "frames.h"
#include "fused_common.h"

namespace psdk {

// -DPSDK_ABL=256 (tools/build_variants.sh): TIMING-ONLY, wrong results -- the two FFT-internal workgroup barriers as
// wave-level syncs.  N = 2048 / 4096: +-0 %, N = 16384: +3.7 %: the barriers are not what these kernels wait for.
#if defined(PSDK_ABL) && (PSDK_ABL & 256)
#define PSDK_FFT_BARRIER() wave_sync()
#else
#define PSDK_FFT_BARRIER() __syncthreads()
#endif
// more TIMING-ONLY ablations (wrong results): 512 = pass B without its exchange (its butterflies on the registers pass A left:
// the cost of a three-pass (16, 16, 16) plan); 2048 = no window loads.  (1024, stage A without LDS on its input side, became
// the real thing: REGA below.)
#ifndef PSDK_ABL
#define PSDK_ABL 0
#endif

// -DPSDK_STAMPS (tools/stamps, ONE bigfused_<N>.hip at a time): wave 0 of workgroup 0 sums the s_memtime ticks between the
// phase boundaries of pair_step into g_bstamps; never defined in the shipped build.
#ifdef PSDK_STAMPS
__device__ unsigned long long g_bstamps[16];
#define PSDK_BSTAMP(k)                                                    \
    do {                                                                  \
        if (stamp_on) {                                                   \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
            bst[k] += t_ - tprev;                                         \
            tprev = t_;                                                   \
        }                                                                 \
    } while (0)
#else
#define PSDK_BSTAMP(k) \
    do {               \
    } while (0)
#endif

template <int N, bool NOX = false>
struct BigGeo : FusedDec<N, NOX> {
    using T = BlockFft<N>;
    static constexpr int TEAM = T::TEAM;          // lanes of the team
// N = 16384: one lane per thread (1024 threads, four wavefronts per SIMD) now that the tables are not held in
// registers; 512 threads x 2 lanes (the only way it fitted while they were) reads 14 % lower.
#ifndef PSDK_BIG16K_VT
#define PSDK_BIG16K_VT 1
#endif
    static constexpr int VT = N >= 16384 ? PSDK_BIG16K_VT : 1; // lanes per thread
    static constexpr int THREADS = TEAM / VT;
    static constexpr int WAVES = THREADS / 64;
    // wavefronts per SIMD the kernel is built for (register budget 512 / WPS) and workgroups per CU
    static constexpr int WPS = N >= 16384 ? (PSDK_BIG16K_VT == 2 ? 2 : 4) : BIG_WAVES_PER_SIMD;
    static constexpr int BLOCKS_PER_CU = (4 * WPS / WAVES) > 0 ? (4 * WPS / WAVES) : 1;
    static constexpr int SCR = 2 * T::FRAME;
    static_assert(FusedDec<N, NOX>::END <= SCR && FusedDec<N, NOX>::WEND <= SCR, "decimator arrays exceed the frame");
    static_assert(THREADS >= FusedDec<N, NOX>::HIST, "one carried filter-state element per thread at most");
};
#ifndef PSDK_REGA
#define PSDK_REGA 1
#endif

// DETREND / EWMA as in fused.hip.  Built for four wavefronts per SIMD (two at N = 16384, whose one
// workgroup per CU is all the LDS holds).
// FRAMES: the launch may hold jobs whose stream is read in place from AdcDac frames (job.fspan >= 0; psdc_process_adcdac_
// frames_device): a lane's four consecutive samples are one 8-byte buffer load of wire words (src/de/data.rs:13), kept raw in
// the register group until the loads have landed and converted there (i16 -> f32 x LSB, DAC words offset binary, :28-35,
// :64,:75) -- the four f32 streams of the traces never exist in memory.  Built as separate kernels: the f32-only launches
// keep their instruction stream and registers.
// SINGLE: overlap 0 -- one segment per "pair" (FusedBatch::single; fused_common.h window_pair); 2: two disjoint segments per
// transform (fused.hip says how).
template <int N, int DETREND, bool EWMA, bool FRAMES = false, int SINGLE = 0>
__global__ __launch_bounds__(BigGeo<N>::THREADS, BigGeo<N>::WPS) void bigfused_kernel(const FusedBatch batch,
                                                                         const float *__restrict__ win,
                                                                         const cf *__restrict__ tw0g,
                                                                         const cf *__restrict__ twag)
{
    // stage A of the decimator from registers + scalar-loaded boundary samples (see pair_step); f32 streams only
    constexpr bool REGA = PSDK_REGA != 0 && !FRAMES && BigGeo<N>::VT == 1;
#ifndef PSDK_HOIST_LOOKAHEAD
#define PSDK_HOIST_LOOKAHEAD 1
#endif
#ifndef PSDK_EARLY_LOOKAHEAD
#define PSDK_EARLY_LOOKAHEAD 0
#endif
#ifndef PSDK_HOIST_FIRST
#define PSDK_HOIST_FIRST 0 // the six loads at the start of a run as ONE frame / f32 decision: see there
#endif
    using G = BigGeo<N, REGA>;
    using T = BlockFft<N>;
    constexpr int TEAM = G::TEAM, VT = G::VT, THREADS = G::THREADS;
    __shared__ __attribute__((aligned(16))) cf s_frame[T::FRAME];
    __shared__ cf s_twb[T::TWB_SIZE];
    __shared__ float s_hist[G::HIST];
    __shared__ __attribute__((aligned(8))) float s_red[2 * G::WAVES + 4];

    const int tp = threadIdx.x; // lane v of this thread is tl = tp + THREADS v
    for (int i = tp; i < T::TWB_SIZE; i += THREADS) { // [(q-1)][s]: W_SA^(s q)
        const int q = i / 16 + 1, s = i % 16;
        float sn, cs;
        sincospif(-2.0f * (float)(s * q) / (float)T::SA, &sn, &cs);
        s_twb[i] = {cs, sn};
    }

    int bid = blockIdx.x;
    if constexpr (FRAMES) {
        // the four traces of a frame span read the SAME bytes: deal the workgroups of a group so that the four with one
        // run index w get physical ids 8 apart -- one XCD under the observed round-robin placement (speed only): its L2
        // serves three of the four reads and the frames cross the fabric once
        for (int g = 0; g < batch.n_fgroups; ++g) {
            const int b0 = batch.fg_begin[g], nb = batch.fg_nb[g];
            if (bid >= b0 && bid < b0 + 4 * nb) {
                const int p = bid - b0, full = (nb >> 3) * 32;
                int c, w;
                if (p < full) {
                    c = (p >> 3) & 3;
                    w = (p >> 5) * 8 + (p & 7);
                } else {
                    const int rem = nb & 7, q = p - full;
                    c = q / rem;
                    w = (nb & ~7) + q % rem;
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
    const unsigned h_pack = G::hist_slot(tp); // THREADS >= 128 > 80: one carried element per thread at most

    float q[VT][16];
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            q[v][s] = 0.0f;

    // this workgroup's run: pairs [p0, p0 + run) of the job (cut by npairs)
    const int p0 = wb * run;
    const int p1 = min(npairs, p0 + run);
    // the stream of this job: an f32 array (cp: this thread's piece of chunk p0, in float4 units) or, in a FRAMES launch,
    // trace job.fch of a frame span (sp: the sample index of the same piece within the span)
    const bool fr = FRAMES && job.fspan >= 0;
    const FrameSpan &fsp = batch.fspans[fr ? job.fspan : 0];
    const unsigned ch_off = fr ? (unsigned)job.fch * 16u : 0u;
    const unsigned dac_flip = (fr && job.fch >= 2) ? 0x80008000u : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(fsp.frames), 0, fr ? (int)min(fsp.bytes, 0x7FFFFFFFull) : 0, 0x00020000);
    const float4 *cp = reinterpret_cast<const float4 *>(job.src) + (size_t)p0 * (N / 4) + tp;
    unsigned sp = job.s_off + (unsigned)p0 * N + 4u * tp;
    // piece k (float4 units from c / 4 k samples from s): f32 data, or the 8 raw bytes of the four wire words (integers:
    // fused_common.h Grp4)
    using G4 = Grp4<FRAMES>;
    auto piece = [&](const float4 *c, unsigned s, int k) -> G4 {
        G4 g;
        if constexpr (FRAMES) {
            if (fr) {
                const unsigned si = s + 4u * (unsigned)k;
                const unsigned off = frame_cell_offset(fsp, si >> 3) + ch_off + (si & 4u) * 2u;
                const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0); // (a GCC-style vector of two u32: index it)
                g.set_raw((unsigned)r[0], (unsigned)r[1]);
                return g;
            }
        }
        g.set(c[k]);
        return g;
    };
    // raw wire words -> volts, in place (a no-op for f32 jobs)
    auto volts = [&](G4 &g) {
        if constexpr (FRAMES) {
            if (fr)
                grp_volts(g, dac_flip, adcdac_lsb());
        }
    };
    const float4 *safe = cp; // look-ahead target once nothing is left to look ahead to
    unsigned safe_s = sp;
    // register groups of a lane: two float4 each (see pair_step)
    G4 ga[VT][2], gb[VT][2], gc[VT][2];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const float4 *c = cp + THREADS * v;
        const unsigned s = sp + 4u * THREADS * v;
#if PSDK_HOIST_FIRST
        // (the form that tripped si-form-memory-clauses in round 3 -- DESIGN.md section 4; `make verify` catches it)
        bool done = false;
        if constexpr (FRAMES) {
            if (fr) {
                auto fp = [&](G4 &g, int k) {
                    const unsigned si = s + 4u * (unsigned)k;
                    const unsigned off = frame_cell_offset(fsp, si >> 3) + ch_off + (si & 4u) * 2u;
                    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
                    g.set_raw((unsigned)r[0], (unsigned)r[1]);
                };
                fp(ga[v][0], 0);
                fp(ga[v][1], TEAM);
                fp(gb[v][0], 2 * TEAM);
                fp(gb[v][1], 3 * TEAM);
                fp(gc[v][0], N / 4);
                fp(gc[v][1], N / 4 + TEAM);
                done = true;
            }
        }
        if (!done) {
            ga[v][0].set(c[0]);
            ga[v][1].set(c[TEAM]);
            gb[v][0].set(c[2 * TEAM]);
            gb[v][1].set(c[3 * TEAM]);
            gc[v][0].set(c[N / 4]);
            gc[v][1].set(c[N / 4 + TEAM]);
        }
#else
        ga[v][0] = piece(c, s, 0);
        ga[v][1] = piece(c, s, TEAM);
        gb[v][0] = piece(c, s, 2 * TEAM);
        gb[v][1] = piece(c, s, 3 * TEAM);
        gc[v][0] = piece(c, s, N / 4);
        gc[v][1] = piece(c, s, N / 4 + TEAM);
#endif
    }
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        volts(ga[v][0]);
        volts(ga[v][1]);
        volts(gb[v][0]);
        volts(gb[v][1]);
        volts(gc[v][0]);
        volts(gc[v][1]);
    }

    // ---- warm-up: filter state at the first new sample of the run (hop >= 1024 > 288, so the
    // history is always inside the run's own first chunk) ------------------------------------
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

    // Mean (src/psd.rs:103-109) with a carried pivot, as in fused.hip: the offset of a segment is piv + m,
    // piv the mean found for the segment before (for the first segment of the run: its own f32 mean), m
    // the mean of the residuals.  s0c = sum of (lo - piv) over the workgroup.
    float piv = 0.0f, s0c = 0.0f;
    if constexpr (DETREND == 3) {
        auto block_sum = [&](float v) { // sum over the workgroup, the same value in every thread
            v = wave_sum64(v);
            __syncthreads(); // s_red free
            if ((tp & 63) == 0)
                s_red[4 + (tp >> 6)] = v;
            __syncthreads();
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < G::WAVES; ++w)
                t += s_red[4 + w];
            return t;
        };
        auto r4 = [](const float4 &x) { return (x.x + x.y) + (x.z + x.w); };
        auto s4 = [](const float4 &x, float pv) { return ((x.x - pv) + (x.y - pv)) + ((x.z - pv) + (x.w - pv)); };
        float r = 0.0f;
#pragma unroll
        for (int v = 0; v < VT; ++v)
            r += (r4(ga[v][0].f()) + r4(ga[v][1].f())) + (r4(gb[v][0].f()) + r4(gb[v][1].f()));
        piv = block_sum(r) * (1.0f / (float)N);
        r = 0.0f;
#pragma unroll
        for (int v = 0; v < VT; ++v)
            r += s4(ga[v][0].f(), piv) + s4(ga[v][1].f(), piv);
        s0c = block_sum(r);
        __syncthreads(); // s_red is written again in the first pair
    }

    EwmaAmp eamp;
    if constexpr (EWMA) {
        if (job.ewma)
            eamp.init(job, job.step0 + (SINGLE != 0 ? 1 : 2) * p0);
    }

    // The lane's own twiddle seeds (W_N^(4 tl); W_L1^s, W_L1^(4 s): fft_block.h) do not change from pair to pair: six registers
    // held across the run instead of three L2 loads per pair -- pass A's two had nothing in front of them to hide their
    // latency behind (stamps: 22 % of a pair at N = 4096).  The window (16 values a lane) stays a per-pair batch of loads.
    // -DPSDK_HOIST_SEEDS=0: the round-2 form (all table loads inside the pair).
#ifndef PSDK_HOIST_SEEDS
#define PSDK_HOIST_SEEDS 1
#endif
    typename T::Seeds sd_run[VT];
    typename T::SeedsA sda_run[VT];
    if constexpr (PSDK_HOIST_SEEDS != 0) {
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            sd_run[v] = T::load_seeds(tp + THREADS * v, tw0g);
            sda_run[v] = T::load_seeds_a(tp + THREADS * v, twag);
        }
    }

    // With several lanes per thread the scheduler must not interleave their sections (it would keep
    // every lane's butterflies and twiddles live at once): a scheduling barrier between lanes.
    auto lane_fence = [] {
        if constexpr (VT > 1)
            __builtin_amdgcn_sched_barrier(0);
    };

#ifdef PSDK_STAMPS
    const bool stamp_on = bid == 0 && run >= 4 && tp < 64;
    unsigned long long bst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    // REGA: the boundary samples of the four pieces stage A consumes in pair p (see there), scalar loads into SGPRs; issued
    // a pass ahead (the start of pass C of the pair before) so that nothing waits for them
#ifndef PSDK_REGA_PREFETCH
#define PSDK_REGA_PREFETCH 1
#endif
    float halo[4][7];
    auto load_halo = [&](int p) {
        typedef const float __attribute__((address_space(4))) *kptr;
        const int wv = __builtin_amdgcn_readfirstlane(tp >> 6);
        const float *hb = job.src + (size_t)p * N + N / 2 + 256 * wv; // this wavefront's first sample of piece 0
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            kptr h = (kptr)(hb + r * (N / 4));
            halo[r][0] = h[-4]; // the lane below lane 0: x, y, z, w
            halo[r][1] = h[-3];
            halo[r][2] = h[-2];
            halo[r][3] = h[-1];
            halo[r][4] = h[-7]; // two below: y, w
            halo[r][5] = h[-5];
            halo[r][6] = h[-9]; // three below: w
        }
    };
    if constexpr (REGA && PSDK_REGA_PREFETCH != 0)
        load_halo(p0);

    // one pair; register groups as in fused.hip: (lo, up) = chunk p, nl = lower half of chunk p + 1
    float keep[VT][16]; // SINGLE == 2: the windowed segment of the even step, until the odd step's transform
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            keep[v][s] = 0.0f;
    auto pair_step = [&](G4(&glo)[VT][2], G4(&gup)[VT][2], G4(&gnl)[VT][2], const float4 *cnext, unsigned snext,
                         bool more, float *o, int p, auto odd_step) {
        constexpr bool DOUBLE = SINGLE == 2, ODD = decltype(odd_step)::value;
        // the samples of this pair (converted at the end of the pair before): lo / up are dead once windowed -- Mean centres
        // these copies in place -- and their groups are reloaded further down
        float4 lo[VT][2], up[VT][2], nl[VT][2];
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                lo[v][e] = glo[v][e].f(), up[v][e] = gup[v][e].f(), nl[v][e] = gnl[v][e].f();
        // The window and twiddle tables are the same for every pair, and left alone the compiler hoists the
        // loads and keeps all of a lane's entries (8 + 30 + 16 registers) live across the whole run -- which pins
        // the kernels to two wavefronts per SIMD (or spills, with two lanes per thread).  The table pointers are
        // re-derived per pair instead, so the loads stay inside the pair (L1 / L2 hits) and the kernels fit the
        // 128 registers of four wavefronts per SIMD with a few dwords spilled: +10 % (N = 4096) to +22 % (N = 8192)
        // at steady state.  (A first, cold measurement had this 2-5 % slower)
        // Two ways to do that, measured per size: an opaque zero OFFSET keeps the pointers global (N <= 4096:
        // +3 % at 4096); pointers that went through the asm themselves are generic to the backend, their flat
        // loads wait on the LDS counter too and end up later in the schedule, which suits the sizes with more
        // table registers per lane (N = 8192: +3 %, N = 16384: +6 % over the global form).
        const cf *tw0p = tw0g, *twap = twag;
        const float *winp = win;
        if constexpr (N <= 4096) {
            size_t zofs = 0;
            asm volatile("" : "+s"(zofs));
            tw0p += zofs;
            twap += zofs;
            winp += zofs;
        } else {
            asm volatile("" : "+s"(tw0p), "+s"(twap), "+s"(winp));
        }
        PSDK_BSTAMP(0); // between pairs
        f2 yc[VT]; // stage C: N/8 outputs, two per lane; stored further down, see there
        // -DPSDK_ABL=4 (timing only, WRONG results): no decimator at all -- state, stages A / B / C and their two barriers gone, the
        // output store kept (round 5: what a decimator taken OUT of the sixteen-wavefront lockstep could give back at N = 16384)
        if constexpr ((PSDK_ABL & 4) != 0) {
#pragma unroll
            for (int r = 0; r < VT; ++r)
                yc[r] = {up[r][0].x, nl[r][0].y};
        } else {
        // ---- decimator (at raised priority, as in fused.hip: +3 % at N = 2048 / 4096) ----------
        if constexpr (VT == 1)
            __builtin_amdgcn_s_setprio(PSDK_DEC_PRIO);
        // (an opaque asm on the packed slot per pair, so that the two unpacked LDS addresses are not kept -- and spilled -- as
        // loop invariants, measured 3-6 % SLOWER at N = 8192: the packed word itself is then what gets reloaded, behind a
        // vmcnt(0) at the top of the pair)
        const unsigned hpk = h_pack;
        if ((hpk & 0xFFFFu) != 0xFFFFu)
            sf[hpk & 0xFFFFu] = hs[tp];
        // samples -> polyphase arrays as single floats (ds_write2_b32 from the registers the loads
        // filled; an 8-byte store of {x, z} would cost moves right behind the loads)
        auto split = [&](int h, const float4 &x) {
            sf[G::XE + h] = x.x;
            sf[G::XE + h + 1] = x.z;
            sf[G::XO + h] = x.y;
            sf[G::XO + h + 1] = x.w;
        };
        if constexpr (REGA) {
            // Stage A straight from the registers the loads filled, as fused.hip does for its one-wavefront team: lane tl
            // holds samples 4 tl .. 4 tl + 3 of each N/4-sample piece and takes the three lanes below with DPP wavefront
            // shifts.  Here the lanes below lane 0 belong to ANOTHER wavefront (or, for the first wavefront, to the piece
            // before): the seven samples of theirs that lanes 0..2 need -- x[-4..-1], x[-7], x[-5], x[-9] counted from the
            // wavefront's first sample -- are wavefront-uniform, so they come as SCALAR loads from the stream (constant
            // address space: lines the neighbouring wavefront's vector loads brought into L2 a pair ago) and enter the
            // shifts as the fill of lane 0.  No polyphase sample arrays, no LDS reads on the input side of the stage and
            // one workgroup barrier fewer per pair.
            const float4 *pc[4] = {&up[0][0], &up[0][1], &nl[0][0], &nl[0][1]};
            if constexpr (PSDK_REGA_PREFETCH == 0)
                load_halo(p);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 &cu = *pc[r];
                const float s1x = dpp_shr1(halo[r][0], cu.x), s1y = dpp_shr1(halo[r][1], cu.y);
                const float s1z = dpp_shr1(halo[r][2], cu.z), s1w = dpp_shr1(halo[r][3], cu.w);
                const float s2y = dpp_shr1(halo[r][4], s1y), s2w = dpp_shr1(halo[r][5], s1w);
                const float s3w = dpp_shr1(halo[r][6], s2w);
                // out j = xe[j-2] + t0 (xo[j-5] + xo[j]) + t1 (xo[j-4] + xo[j-1]) + t2 (xo[j-3] + xo[j-2])   (fused.hip)
                float a0 = 0.0f, a1 = 0.0f;
                a0 += (s3w + cu.y) * ta[0];
                a1 += (s2y + cu.w) * ta[0];
                a0 += (s2y + s1w) * ta[1];
                a1 += (s2w + cu.y) * ta[1];
                a0 += (s2w + s1y) * ta[2];
                a1 += (s1y + s1w) * ta[2];
                const int u = tp + THREADS * r;
                sf[G::AE + 11 + u] = s1x + a0;
                sf[G::AO + 11 + u] = s1z + a1;
            }
        } else {
        {
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int tl = tp + THREADS * v;
            if (tl >= TEAM - 3)
                split(2 * (tl - (TEAM - 3)), lo[v][1]);
            const int h = G::HX / 2 + 2 * tl;
            split(h, up[v][0]);
            split(h + N / 8, up[v][1]);
            split(h + N / 4, nl[v][0]);
            split(h + 3 * N / 8, nl[v][1]);
        }
        __syncthreads();
        }
        PSDK_BSTAMP(1); // state + samples -> LDS (waits for the look-ahead loads) + barrier
#pragma unroll
        for (int r = 0; r < 2 * VT; ++r) { // stage A: N/2 outputs, four per step
            const int u = tp + THREADS * r;
            float y[4];
            hbf_four<HBF_MA, G::A_CE, G::A_CO, PSDK_HBF_WIDE_A != 0 && (G::XO % 4 == 0)>(sf + G::XE, sf + G::XO, 4 * u, ta, y);
            sf[G::AE + 11 + 2 * u] = y[0];
            sf[G::AO + 11 + 2 * u] = y[1];
            sf[G::AE + 12 + 2 * u] = y[2];
            sf[G::AO + 12 + 2 * u] = y[3];
        }
        } // !REGA
        __syncthreads();
        PSDK_BSTAMP(2); // stage A + barrier
#pragma unroll
        for (int r = 0; r < VT; ++r) { // stage B: N/4 outputs, four per lane
            const int u = tp + THREADS * r;
            float y[4];
            hbf_four<HBF_MB, G::B_CE, G::B_CO, PSDK_HBF_WIDE != 0 && (G::AO % 4 == 0)>(sf + G::AE, sf + G::AO, 4 * u, tb, y);
            sf[G::BE + 29 + 2 * u] = y[0];
            sf[G::BO + 29 + 2 * u] = y[1];
            sf[G::BE + 30 + 2 * u] = y[2];
            sf[G::BO + 30 + 2 * u] = y[3];
        }
        __syncthreads();
        PSDK_BSTAMP(3); // stage B + barrier
#pragma unroll
        for (int r = 0; r < VT; ++r) {
            const int u = tp + THREADS * r;
            hbf_two<HBF_MC, G::C_CE, G::C_CO>(sf + G::BE, sf + G::BO, 2 * u, tc, yc[r].x, yc[r].y);
        }
        if ((hpk & 0xFFFFu) != 0xFFFFu)
            hs[tp] = sf[hpk >> 16];

        if constexpr (VT == 1)
            __builtin_amdgcn_s_setprio(0);
        } // !(PSDK_ABL & 4)

        // ---- detrend parameters (block-wide broadcast / reduction through LDS) -------------
        DetrendParams dp;
        float &oa = dp.oa, &ob = dp.ob, &ma = dp.ma, &mb = dp.mb;
        slope2 &sa = dp.sa, &sb = dp.sb;
        if constexpr (DETREND == 1) { // the segments' midpoint samples (lane 0)
            if (tp == 0) {
                s_red[0] = up[0][0].x;
                s_red[1] = nl[0][0].x;
            }
        } else if constexpr (DETREND == 2) { // first (lane 0) and last (lane TEAM - 1) samples
            if (tp == 0) {
                s_red[0] = lo[0][0].x;
                s_red[1] = up[0][0].x;
            }
            if (tp == THREADS - 1) {
                s_red[2] = up[VT - 1][1].w;
                s_red[3] = nl[VT - 1][1].w;
            }
        }
        if constexpr (DETREND == 3) { // centre lo and up in place (dead after the window); partial sums of up, nl
            auto sub4 = [](float4 &x, float pv) {
                x.x -= pv;
                x.y -= pv;
                x.z -= pv;
                x.w -= pv;
            };
            auto r4 = [](const float4 &x) { return (x.x + x.y) + (x.z + x.w); };
            auto s4 = [](const float4 &x, float pv) { return ((x.x - pv) + (x.y - pv)) + ((x.z - pv) + (x.w - pv)); };
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                sub4(lo[v][0], piv);
                sub4(lo[v][1], piv);
                sub4(up[v][0], piv);
                sub4(up[v][1], piv);
                t1 += r4(up[v][0]) + r4(up[v][1]);
                t2 += s4(nl[v][0], piv) + s4(nl[v][1], piv); // nl stays raw: it is the next pair's lo
            }
            t1 = wave_sum64(t1);
            t2 = wave_sum64(t2);
            if ((tp & 63) == 0) {
                s_red[4 + 2 * (tp >> 6)] = t1;
                s_red[5 + 2 * (tp >> 6)] = t2;
            }
        }
        __syncthreads(); // the frame is reused by the FFT; s_red published
        PSDK_BSTAMP(4); // stage C + state save + detrend prep + barrier
        if constexpr (DETREND == 1) {
            oa = s_red[0];
            ob = s_red[1];
        } else if constexpr (DETREND == 2) {
            oa = s_red[0];
            ob = s_red[1];
            sa = span_slope(oa, s_red[2], N);
            sb = span_slope(ob, s_red[3], N);
        }
        if constexpr (DETREND == 3) { // (s_red is next written a pair later, several barriers on)
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
        cf vv[VT][16];
        auto lookahead = [&] {
            const float4 *src = more ? cnext : safe;
            const unsigned ssrc = more ? snext : safe_s;
            safe = src;
            safe_s = ssrc;
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const float4 *c = src + THREADS * v;
                const unsigned s = ssrc + 4u * THREADS * v;
#if PSDK_HOIST_LOOKAHEAD
                // (the frame / f32 decision once for the four loads: with it inside piece(), every load sat in a branch of its own whose
                // result reached its register group through a copy at the join, behind an s_waitcnt vmcnt(0) -- the look-ahead loads
                // went out one at a time; N = 8192 frames +8 %, 16384 +5 %.  The SIX loads at the start of a run stay as they are:
                // hoisted the same way they read components .x / .y of the first group wrong in the EWMA x FRAMES variant alone --
                // DESIGN.md section 4)
                bool done = false;
                if constexpr (FRAMES) {
                    if (fr) {
                        auto fp = [&](G4 &g, int k) {
                            const unsigned si = s + 4u * (unsigned)k;
                            const unsigned off = frame_cell_offset(fsp, si >> 3) + ch_off + (si & 4u) * 2u;
                            const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
                            g.set_raw((unsigned)r[0], (unsigned)r[1]);
                        };
                        fp(gup[v][0], 2 * TEAM);
                        fp(gup[v][1], 3 * TEAM);
                        fp(glo[v][0], N / 4);
                        fp(glo[v][1], N / 4 + TEAM);
                        done = true;
                    }
                }
                if (!done) {
                    gup[v][0].set(c[2 * TEAM]);
                    gup[v][1].set(c[3 * TEAM]);
                    glo[v][0].set(c[N / 4]);
                    glo[v][1].set(c[N / 4 + TEAM]);
                }
#else
                gup[v][0] = piece(c, s, 2 * TEAM);
                gup[v][1] = piece(c, s, 3 * TEAM);
                glo[v][0] = piece(c, s, N / 4);
                glo[v][1] = piece(c, s, N / 4 + TEAM);
#endif
            }
        };
        if constexpr (EWMA) {
            if (job.ewma) {
                if constexpr (SINGLE == 0)
                    dp.ea = eamp.next(job); // steps job.step0 + 2 p and + 1: the pairs of a run are consecutive
                dp.eb = eamp.next(job);    // (SINGLE: one step per pair, segment b)
            }
        }
        // the tables of this pair in ONE batch of loads: the window and the twiddle seeds (fft_block.h) -- one
        // exposed L2 round trip per pair where reading all table entries at their uses was nineteen.  (Issuing the
        // batch before the decimator instead hides that one as well but keeps 26 more registers live across it:
        // measured slower at every size, and much slower at N = 16384, whose 1024 threads spill.  The window alone in front of
        // stage C, round 3: N = 8192 -2.7 %, Mean -10 %; N = 16384 +-0, Mean -12 % -- spills again.)
        typename T::Seeds sd[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int tl = tp + THREADS * v;
            const float4 *wp = reinterpret_cast<const float4 *>(winp) + tl;
            float4 wq0, wq1, wq2, wq3;
            if constexpr (PSDK_ABL & 2048) { // timing only: no window loads
                wq0 = wq1 = wq2 = wq3 = make_float4(0.5f, 0.25f + dp.ea, 0.125f, 0.75f);
            } else {
                wq0 = wp[0], wq1 = wp[TEAM], wq2 = wp[2 * TEAM], wq3 = wp[3 * TEAM];
            }
            if constexpr (PSDK_HOIST_SEEDS != 0) {
                sd[v] = sd_run[v];
                // (opaque per pair: the products formed from the seeds must not be hoisted out of the loop with them --
                // left alone the compiler keeps every derived twiddle of the run live and spills 200 bytes a lane)
                asm volatile("" : "+v"(sd[v].w0.re), "+v"(sd[v].w0.im));
            } else {
                sd[v] = T::load_seeds(tl, tw0p);
            }
            window_pair<N, DETREND, EWMA, true, SINGLE != 0>(vv[v], tl, lo[v][0], lo[v][1], up[v][0], up[v][1], nl[v][0], nl[v][1], wq0, wq1,
                                          wq2, wq3, dp);
            if constexpr (DOUBLE && !ODD) { // the even step keeps its windowed segment: no transform
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    keep[v][s] = vv[v][s].re;
            } else {
                if constexpr (DOUBLE) {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        vv[v][s] = {keep[v][s], vv[v][s].re};
                }
                // (every lane rewrites exactly the frame positions it read, so the lanes of a thread
                // go one after the other between two barriers)
                T::pass0(tl, vv[v], sd[v]);
                T::store0(tl, vv[v], frame);
            }
            lane_fence();
        }
        if constexpr (DOUBLE && !ODD) { // (no frame use: the barrier behind stage C already separates this pair's decimator from
                                        // the next one's; the decimator's outputs and the look-ahead loads leave as in an odd step)
#pragma unroll
            for (int r = 0; r < VT; ++r)
                *reinterpret_cast<f2 *>(o + 2 * (tp + THREADS * r)) = yc[r];
            lookahead();
            if constexpr (REGA && PSDK_REGA_PREFETCH != 0)
                load_halo(more ? p + 1 : p);
            return;
        }
        // -DPSDK_EARLY_LOOKAHEAD=1 (experiment): the look-ahead loads here, a pass earlier -- the twiddle seeds are held across the
        // run since round 3, so nothing behind these loads waits on vmcnt any more; their 16 destination registers are then live
        // through passes 0 and A
        if constexpr (PSDK_EARLY_LOOKAHEAD != 0)
            lookahead();
        __syncthreads();
        PSDK_BSTAMP(5); // table loads + window + pass 0 + store + barrier
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            T::loadA(tp + THREADS * v, vv[v], frame);
            typename T::SeedsA sa_ = PSDK_HOIST_SEEDS != 0 ? sda_run[v] : T::load_seeds_a(tp + THREADS * v, twap);
            if constexpr (PSDK_HOIST_SEEDS != 0)
                asm volatile("" : "+v"(sa_.a1.re), "+v"(sa_.a1.im), "+v"(sa_.a4.re), "+v"(sa_.a4.im));
            T::passA(tp + THREADS * v, vv[v], sa_);
            T::storeA(tp + THREADS * v, vv[v], frame);
            lane_fence();
        }
        { // chunk p + 1 upper -> up, chunk p + 2 lower -> lo, in flight during passes B and C and the
          // next decimator's first stage -- AFTER pass A's twiddle seeds (vmcnt retires loads in order: a wait for the
          // seeds behind these would wait for HBM); issued
          // unconditionally (after the last pair: re-reads of pieces read before, unused) so that
          // the compiler does not wait for them at the end of a branch
            // (the decimator's outputs leave here too: a store issued in stage C would sit in front of the table
            // loads in the same in-order counter, and the wait for the window would wait for its write as well)
#pragma unroll
            for (int r = 0; r < VT; ++r)
                *reinterpret_cast<f2 *>(o + 2 * (tp + THREADS * r)) = yc[r];
            if constexpr (PSDK_EARLY_LOOKAHEAD == 0)
                lookahead();
        }
        PSDK_FFT_BARRIER();
        PSDK_BSTAMP(6); // pass A (seeds from L2) + output store + look-ahead issue + barrier
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            if constexpr (PSDK_ABL & 512) {
                T::loadC(tp + THREADS * v, vv[v], frame); // (what pass A stored: same traffic as a real exchange)
                T::passB(tp + THREADS * v, vv[v], s_twb);
            } else {
            T::loadB(tp + THREADS * v, vv[v], frame);
            T::passB(tp + THREADS * v, vv[v], s_twb);
            T::storeB(tp + THREADS * v, vv[v], frame);
            }
            lane_fence();
        }
        if constexpr (!(PSDK_ABL & 512))
        PSDK_FFT_BARRIER();
        PSDK_BSTAMP(7); // pass B + barrier
        if constexpr (REGA && PSDK_REGA_PREFETCH != 0)
            load_halo(more ? p + 1 : p);
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            if constexpr (!(PSDK_ABL & 512))
            T::loadC(tp + THREADS * v, vv[v], frame);
            T::passC(vv[v]);
#pragma unroll
            for (int s = 0; s < 16; ++s)
                q[v][s] = fmaf(vv[v][s].re, vv[v][s].re, fmaf(vv[v][s].im, vv[v][s].im, q[v][s]));
            lane_fence();
        }
        if constexpr (FRAMES) { // the look-ahead groups hold raw wire words: to volts before the next pair reads them
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                volts(gup[v][0]);
                volts(gup[v][1]);
                volts(glo[v][0]);
                volts(glo[v][1]);
            }
        }
        __syncthreads(); // next pair's decimator writes the frame
        PSDK_BSTAMP(8); // pass C + |Z|^2 + barrier
    };

    {
        float *o = job.dst + (size_t)p0 * (N / 8);
        for (int p = p0; p < p1; p += 2) {
            pair_step(ga, gb, gc, cp + N / 4, sp + N, p + 1 < p1, o, p, std::false_type{});
            cp += N / 4;
            sp += N;
            o += N / 8;
            if (p + 1 < p1) {
                pair_step(gc, gb, ga, cp + N / 4, sp + N, p + 2 < p1, o, p + 1, std::true_type{});
                cp += N / 4;
                sp += N;
                o += N / 8;
            }
        }
    }

#ifdef PSDK_STAMPS
    if (stamp_on && tp == 0) {
        for (int k = 0; k < 12; ++k)
            g_bstamps[k] = bst[k];
        g_bstamps[12] = (unsigned long long)(p1 - p0);
    }
#endif
    // one team per workgroup: its accumulators are the partial
    float *out = job.partial + (size_t)wb * N;
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            out[T::freq_of(tp + THREADS * v, s)] = q[v][s];
}

template <int N>
hipError_t launch_bigfused_n(const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag, hipStream_t s,
                             hipEvent_t ea, hipEvent_t eb)
{
    const dim3 grid(b.nblocks), block(BigGeo<N>::THREADS);
    const bool ew_ = b.any_ewma || (dbg_variant() & 1), frm_ = b.any_frames || (dbg_variant() & 2);
#define PSDK_BIG_CASE(D)                                                                          \
    case D:                                                                                       \
        if (frm_ && ew_)                                                                          \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, true, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag);  \
        else if (frm_)                                                                            \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, false, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag); \
        else if (ew_)                                                                             \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, true>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag);  \
        else                                                                                      \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, false>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag); \
        break;
#define PSDK_BIG_SINGLE(D)                                                                                          \
    case D:                                                                                                         \
        if (b.single == 2 && ew_)                                                                                   \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, true, false, 2>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag);  \
        else if (b.single == 2)                                                                                     \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, false, false, 2>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag); \
        else if (ew_)                                                                                               \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, true, false, 1>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag);  \
        else                                                                                                        \
            hipExtLaunchKernelGGL((bigfused_kernel<N, D, false, false, 1>), grid, block, 0, s, ea, eb, 0, b, win, tw0g, twag); \
        break;
    if (b.single) {
        if (b.any_frames)
            return hipErrorInvalidValue;
        switch (b.detrend) {
            PSDK_BIG_SINGLE(0)
            PSDK_BIG_SINGLE(1)
            PSDK_BIG_SINGLE(2)
            PSDK_BIG_SINGLE(3)
        default:
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#undef PSDK_BIG_SINGLE
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

#ifdef PSDK_STAMPS
extern "C" int psdc_debug_stamps_big(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(psdk::g_bstamps), 16 * sizeof(unsigned long long));
}
#endif
