// tests/host/sim/sim_kernels.cpp -- TEST INFRASTRUCTURE: host models of the kernel launchers of csrc/kernels.h, linked with
// the library's host runtime (csrc/{runtime,planner,frames_ingest,readout}.cpp, unchanged) in tests/host/round_plan_check.  See sim_device.h for what the
// models track.  Each model names the real kernel whose memory accesses it reproduces.
#include "kernels.h"

#include <algorithm>

#include "hbf_taps.h"
#include "sim_device.h"

namespace psdk {

using sim::bits_of;
using sim::error;
using sim::ident;
using sim::mark;
using sim::put_bits;
using sim::world;

// ---- geometry the planner asks for (fused.hip / kernels.hip) -----------------------------------------------------
bool fused_supported(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384; }
bool fused_frames_supported(int n) { return fused_supported(n); }
bool fused_double_supported(int n) { return fused_supported(n); }
bool fused_fold_supported(int n) { return n == 256 || n == 512 || n == 1024; }
static int fused_teams(int n) { return n >= 2048 ? 1 : FUSED_WAVES * (64 / (n / 16)); } // FusedGeo<N>::TEAMS
int fused_pairs_per_block(int n, int run) { return fused_supported(n) ? fused_teams(n) * run : 0; }
int fused_block_threads(int n) { return n >= 2048 ? n / 16 : FUSED_WAVES * 64; }
int fused_max_blocks(int n)
{
    if (world().cap_blocks > 0)
        return world().cap_blocks;
    auto big = [](int threads, int wps) { return 256 * std::max(1, 4 * wps / (threads / 64)); };
    switch (n) {
    case 2048:
        return big(128, BIG_WAVES_PER_SIMD);
    case 4096:
        return big(256, BIG_WAVES_PER_SIMD);
    case 8192:
        return big(512, BIG_WAVES_PER_SIMD);
    case 16384:
        return 256;
    default:
        return 256 * (4 * FUSED_WAVES_PER_SIMD / FUSED_WAVES);
    }
}
void fused_big_tables(int n, std::vector<cf> &tw0, std::vector<cf> &twa)
{
    tw0.clear();
    twa.clear();
    if (n >= 2048 && fused_supported(n)) {
        tw0.assign(4, cf{1.0f, 0.0f});
        twa.assign(4, cf{1.0f, 0.0f});
    }
}
void fused_big3_table(int n, std::vector<cf> &tw3)
{
    tw3.clear();
    if (n == 2048 || n == 4096)
        tw3.assign(4, cf{1.0f, 0.0f});
}
int bluestein_size(int n)
{
    if (n < 16 || n > 8192 || (n & (n - 1)) == 0)
        return 0;
    int m = 32;
    while (m < 2 * n - 1)
        m <<= 1;
    return m;
}
bool bigfft_size(int n) { return n > 16384 && n <= BIGFFT_MAX_N && (n & (n - 1)) == 0; }
int welch_segments_per_tile(int n)
{
    if (bigfft_size(n))
        return 1 << 20;
    if (bluestein_size(n))
        return 32;
    return (n >= 16 && n <= 16384 && (n & (n - 1)) == 0) ? 32 : 0;
}
bool welch_supported(int n) { return welch_segments_per_tile(n) != 0; }

// ---- AdcDac frames as a sample source (frames.h frame_cell_offset / frame_sample) ----------------------------------
// The check program builds frames whose header `seq` is the absolute batch index of the frame's first batch, so sample
// (seq + batch) * 8 + i of trace ch is identified from the bytes themselves.  `width` = bytes the real load touches.
static uint32_t frame_ident(const FrameSpan &fs, int ch, unsigned long long si, int width, const char *who)
{
    const unsigned long long cell = si >> 3, f = cell / fs.batches, b = cell % fs.batches;
    const unsigned long long off = f * fs.frame_size + 8 + b * 64 + (unsigned)ch * 16 + ((si & 7) & ~(unsigned long long)(width / 2 - 1)) * 2;
    if (off + (unsigned)width > fs.bytes) {
        error("%s: frame load at byte %llu + %d beyond the span's %llu bytes", who, off, width, fs.bytes);
        return 0;
    }
    volatile uint8_t sink = 0;
    for (int k = 0; k < width; ++k)
        sink = sink + fs.frames[off + k]; // (ASan: the bytes exist)
    (void)sink;
    const uint8_t *h = fs.frames + f * fs.frame_size;
    const uint32_t seq = (uint32_t)h[4] | ((uint32_t)h[5] << 8) | ((uint32_t)h[6] << 16) | ((uint32_t)h[7] << 24);
    return ident(ch * 16, ((unsigned long long)seq + b) * 8 + (si & 7));
}

// `count` consecutive samples at p must be samples idx0, idx0 + 1, ... of ONE stream; returns its tag (-1 on a mismatch)
static int check_run(const float *p, size_t count, uint64_t idx0, bool idx_known, uint64_t *first_idx, const char *who)
{
    if (!count)
        return -1;
    const uint32_t b0 = bits_of(p);
    const int tag = (int)(b0 >> 24);
    const uint64_t a = b0 & 0xFFFFFFu;
    if (first_idx)
        *first_idx = a;
    if (idx_known && a != (idx0 & 0xFFFFFFu)) {
        error("%s: read sample %llu of stream tag %d where sample %llu was meant", who, (unsigned long long)a, tag, (unsigned long long)idx0);
        return -1;
    }
    for (size_t i = 1; i < count; ++i) {
        const uint32_t b = bits_of(p + i);
        if (b != ident(tag, a + i)) {
            error("%s: stream tag %d breaks at offset %zu of a run of %zu from sample %llu (found tag %u sample %u)", who, tag, i, count,
                  (unsigned long long)a, b >> 24, b & 0xFFFFFFu);
            return -1;
        }
    }
    return tag;
}

// a partial row as a modelled kernel leaves it: [0] segments folded into it, [1] a marker the fold consumes, zeros elsewhere
static void write_partial_row(float *row, int n, int segments)
{
    for (int k = 0; k < n; ++k)
        row[k] = 0.0f;
    row[0] = (float)segments;
    row[1] = 1.0f;
}

// ---- welch_kernel / welch_bluestein_kernel ---------------------------------------------------------------------------
hipError_t launch_welch(int n, const WelchBatch &b, const float *win, const cf *tw, const cf *chirp, const cf *bhat, hipStream_t s)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    const int spt = welch_segments_per_tile(n);
    if (!win || !tw || (bluestein_size(n) && (!chirp || !bhat)))
        error("launch_welch: null table");
    sim::enqueue(s, [b, n, spt] {
        int blocks = 0;
        for (int ji = 0; ji < b.njobs; ++ji) {
            const SegJob &job = b.jobs[ji];
            world().seg_jobs += 1;
            if (job.block_begin != blocks)
                error("welch job %d: block_begin %d, expected %d", ji, job.block_begin, blocks);
            blocks += job.nblocks;
            if (job.nblocks < 1 || job.nblocks > job.ntiles || job.ntiles != (job.nseg + spt - 1) / spt)
                error("welch job %d: %d workgroups for %d tiles of %d segments", ji, job.nblocks, job.ntiles, job.nseg);
            for (int wb = 0; wb < job.nblocks; ++wb) {
                int cnt = 0;
                for (int lt = wb; lt < job.ntiles; lt += job.nblocks)
                    for (int la = lt * spt; la < std::min(job.nseg, (lt + 1) * spt); ++la) {
                        const long long sidx = job.seg0 + la;
                        const long long ofs = sidx * (long long)b.hop - job.src_base;
                        int tag = -1;
                        if (job.fspan >= 0) {
                            const FrameSpan &fs = b.fspans[job.fspan];
                            tag = job.fch * 16;
                            for (int j = 0; j < n; ++j)
                                if (frame_ident(fs, job.fch, (unsigned long long)(ofs + job.s_off + j), 2, "welch (frames)") !=
                                    ident(tag, (uint64_t)sidx * b.hop + j)) {
                                    error("welch (frames): segment %lld sample %d is not sample %lld of trace %d", sidx, j,
                                          sidx * (long long)b.hop + j, job.fch);
                                    break;
                                }
                        } else {
                            tag = check_run(job.src + ofs, (size_t)n, (uint64_t)sidx * b.hop, true, nullptr, "welch");
                        }
                        if (tag >= 0)
                            mark(world().seg, tag, (uint64_t)sidx);
                        world().seg_segments += 1;
                        ++cnt;
                    }
                write_partial_row(job.partial + (size_t)wb * n, n, cnt);
            }
        }
        if (blocks != b.nblocks)
            error("welch launch: %d workgroups named by the jobs, grid of %d", blocks, b.nblocks);
    });
    return hipSuccess;
}

// ---- bigfft.hip (n > 16384): the same jobs, ONE partial row each, scratch for at least one pair -----------------------
hipError_t launch_welch_big(int n, const WelchBatch &b, const float *win, const cf *tw, cf *scratch, size_t scratch_elems, int chunk_limit, hipStream_t s)
{
    if (!bigfft_size(n) || !scratch || scratch_elems < 2 * (size_t)n)
        return hipErrorInvalidValue;
    for (int ji = 0; ji < b.njobs; ++ji)
        if (b.jobs[ji].nblocks != 1 || b.jobs[ji].fspan >= 0)
            error("big-FFT welch job %d: %d partial rows / frame source %d", ji, b.jobs[ji].nblocks, b.jobs[ji].fspan);
    volatile cf *sc = scratch; // (the frames exist: first and last element)
    sc[0].re = 0.0f;
    sc[scratch_elems - 1].im = 0.0f;
    return launch_welch(n, b, win, tw, nullptr, nullptr, s);
}

// ---- fused_kernel / bigfused_kernel / bigfused3_kernel -------------------------------------------------------------
hipError_t launch_fused(int n, const FusedBatch &b, const float *win, const cf *, const cf *, const cf *, hipStream_t s, hipEvent_t,
                        hipEvent_t, const FusedAux *aux)
{
    if (b.nblocks <= 0)
        return hipSuccess;
    if (!win)
        error("launch_fused: null window");
    const FusedAux ax = aux ? *aux : FusedAux{};
    if (aux && !fused_fold_supported(n))
        error("launch_fused: aux workgroups at n = %d, whose kernel has none", n);
    if (ax.nblocks != ax.red_blocks + ax.ntail || ax.nred_tall < 0 || ax.nred_mid < 0 || ax.nred_tall + ax.nred_mid > ax.nred ||
        ax.red_blocks != ax.nred_tall * ax.red_xb + ax.nred_mid * ax.red_mb + (ax.nred - ax.nred_tall - ax.nred_mid) ||
        (ax.nred && ax.red_mb != (ax.red_xb + AUX_MID_GROUPS - 1) / AUX_MID_GROUPS) || ax.nred > AUX_MAX_RED ||
        ax.ntail + ax.npre > AUX_MAX_TAIL || (ax.nred && (ax.n != n || ax.red_xb != (n / 2 + 1 + AUX_RED_BINS - 1) / AUX_RED_BINS)))
        error("launch_fused: inconsistent aux table (%d workgroups: %d fold jobs x %d + %d copies, %d prologues)", ax.nblocks, ax.nred,
              ax.red_xb, ax.ntail, ax.npre);
    world().fused_enqueued += 1;
    sim::enqueue(s, [b, n, ax] {
        // The aux workgroups run BESIDE the compute workgroups: the real order is any, so the model runs both of their roles behind the
        // jobs in one launch and in front of them in the next.  Behind: breaks if a job of this launch read what the tail carries write
        // (it would have seen the stale buffer front), and if the fold read the partial slab this launch writes (the jobs have by
        // then overwritten its rows, this fold takes the new round's, and the next fold of the same rows finds them consumed:
        // tools/planner_mutations.sh, mutation 6).  In front: breaks if a job wrote what the tail carries read.
        auto aux_roles = [&] {
            for (int ji = 0; ji < ax.ntail; ++ji) { // the aux workgroups' tail carries (see above)
                const TailJob &job = ax.tail[ji];
                world().tail_jobs += 1;
                if (job.fspan >= 0)
                    error("fused launch: aux copy %d decodes frames", ji);
                else
                    for (int i = 0; i < job.count; ++i)
                        job.dst[i] = job.src[i];
            }
            for (int ji = 0; ji < ax.nred; ++ji) {
                const RedJob &job = ax.red[ji];
                world().red_jobs += 1;
                const int shape = ji < ax.nred_tall ? 0 : ji < ax.nred_tall + ax.nred_mid ? 1 : 2;
                if (shape != (job.nparts > AUX_MID_ROWS ? 0 : job.nparts > AUX_SHORT_ROWS ? 1 : 2))
                    error("fold (aux) job %d of %d rows sits among the jobs of shape %d", ji, job.nparts, shape);
                std::vector<double> acc((size_t)n / 2 + 1, 0.0);
                for (int t = 0; t < job.nparts; ++t) {
                    float *row = const_cast<float *>(job.partial) + (size_t)t * n;
                    if (row[1] != 1.0f)
                        error("fold (aux) job %d: partial row %d of %d was not written by the round before (or is folded twice)", ji, t, job.nparts);
                    row[1] = 0.0f; // consumed
                    for (int k = 0; k <= n / 2; ++k)
                        acc[(size_t)k] += (double)row[k] + (double)row[k ? n - k : 0];
                }
                for (int k = 0; k <= n / 2; ++k)
                    job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc[(size_t)k]);
            }
        };
        static unsigned long aux_launch = 0;
        const bool aux_first = ax.nblocks && (aux_launch++ & 1); // (every other launch that has aux workgroups)
        if (aux_first)
            aux_roles();
        std::vector<char> pre_used((size_t)ax.npre, 0);
        if (ax.nblocks)
            world().aux_launches += 1;
        if (ax.npre || ax.nred)
            world().one_launch_rounds += 1; // (the post launch in front of this round is gone)
        world().prologue_copies += ax.npre;
        // (single: overlap 0 -- pair p is segment seg_a + p = samples [N p, N p + N); it still decimates [N p + N/2, N p + 3N/2))
        const int teams = fused_teams(n), hop = b.single ? n : n / 2;
        world().fused_launches += 1;
        int blocks = 0;
        long big_blocks = 0;
        for (int ji = 0; ji < b.njobs; ++ji) {
            const FusedJob &job = b.jobs[ji];
            world().fused_jobs += 1;
            world().fused_pairs += job.npairs;
            world().max_run = std::max<long>(world().max_run, job.run);
            if (job.nblocks > 1)
                world().multi_block_jobs += 1;
            if (job.block_begin != blocks)
                error("fused job %d: block_begin %d, expected %d", ji, job.block_begin, blocks);
            blocks += job.nblocks;
            if (job.pre_count) { // the job's copy prologue (its seam): by its ONE workgroup, in front of its first read
                if (job.nblocks != 1 || job.pre_first < ax.ntail || job.pre_first + job.pre_count > ax.ntail + ax.npre)
                    error("fused job %d: prologue [%d, +%d) with %d workgroups (aux: %d copies, %d prologues)", ji, job.pre_first,
                          job.pre_count, job.nblocks, ax.ntail, ax.npre);
                else
                    for (int q = 0; q < job.pre_count; ++q) {
                        const TailJob &t = ax.tail[job.pre_first + q];
                        world().tail_jobs += 1;
                        if (t.fspan >= 0)
                            error("fused job %d: a prologue that decodes frames", ji);
                        else
                            for (int i = 0; i < t.count; ++i)
                                t.dst[i] = t.src[i];
                        pre_used[(size_t)(job.pre_first + q - ax.ntail)] += 1;
                    }
            }
            if (job.npairs < 1 || job.run < 1 || (long long)job.nblocks * teams * job.run < job.npairs ||
                (long long)(job.nblocks - 1) * teams * job.run >= job.npairs) {
                error("fused job %d: %d pairs in %d workgroups of %d teams x run %d", ji, job.npairs, job.nblocks, teams, job.run);
                continue;
            }
            if (4LL * job.npairs > (long long)job.run * teams || job.nblocks > 1)
                big_blocks += job.nblocks;
            const bool fr = job.fspan >= 0;
            if (fr)
                world().fused_frame_jobs += 1;
            if (fr && b.single)
                error("fused job %d reads frames in a single-segment launch", ji);
            if (b.single == 2 && ((job.npairs & 1) || (job.run & 1) || !fused_double_supported(n)))
                error("fused job %d: %d segments in runs of %d for the two-segments-per-transform kernels (n = %d)", ji, job.npairs, job.run, n);
            if (fr && !b.any_frames)
                error("fused job %d reads frames in a launch without any_frames", ji);
            if (job.ewma && !b.any_ewma)
                error("fused job %d has finite averaging in a launch without any_ewma", ji);
            if (!fr && (reinterpret_cast<uintptr_t>(job.src) & 15u))
                error("fused job %d: source not 16-byte aligned", ji);
            if (fr && (job.s_off & 3u))
                error("fused job %d: frame sample offset %u not a multiple of 4", ji, job.s_off);
            const FrameSpan *fs = fr ? &b.fspans[job.fspan] : nullptr;
            const int tag = fr ? job.fch * 16 : (int)(bits_of(job.src) >> 24);
            // absolute index of the job's first sample (segment seg_a starts there)
            const uint64_t a0 = fr ? (frame_ident(*fs, job.fch, job.s_off, 8, "fused (frames)") & 0xFFFFFFu) : (bits_of(job.src) & 0xFFFFFFu);
            if ((a0 + (b.single ? n / 2 : 0)) % (unsigned)hop)
                error("fused job %d: starts at sample %llu, not on a segment boundary", ji, (unsigned long long)a0);
            auto sample_ok = [&](long long j, int width) { // sample j of the job (j may be negative: the warm-up) is sample a0 + j
                if (fr)
                    return frame_ident(*fs, job.fch, (unsigned long long)((long long)job.s_off + j), width, "fused (frames)") ==
                           ident(tag, a0 + j);
                return bits_of(job.src + j) == ident(tag, a0 + j);
            };
            for (int wb = 0; wb < job.nblocks; ++wb) {
                int segs = 0;
                for (int team = 0; team < teams; ++team) {
                    const long long p0 = ((long long)wb * teams + team) * job.run;
                    const long long nrun = std::min<long long>(job.run, job.npairs - p0);
                    if (nrun <= 0)
                        continue;
                    // warm-up: the 288 samples before the run's first new sample; those in front of src - pre read as zeros,
                    // which is right only in front of the stream's first sample
                    const long long xn = p0 * n + n / 2;
                    for (long long i0 = -(long long)HBF_HALO; i0 < 0; ++i0) {
                        const long long j = xn + i0;
                        if (j >= -(long long)job.pre) {
                            if (!sample_ok(j, 2)) {
                                error("fused job %d: warm-up of the run at pair %lld reads a wrong sample at offset %lld", ji, p0, j);
                                break;
                            }
                        } else if ((long long)a0 + j >= 0) {
                            error("fused job %d: warm-up at pair %lld takes sample %lld of the stream for history before its start "
                                  "(pre = %d)", ji, p0, (long long)a0 + j, job.pre);
                            break;
                        }
                    }
                    for (long long p = p0; p < p0 + nrun; ++p) {
                        bool ok = true;
                        for (long long j = (long long)n * p; j < (long long)n * p + 3 * n / 2 && ok; j += 4) {
                            for (int e = 0; e < 4 && ok; ++e)
                                ok = sample_ok(j + e, fr ? 8 : 4);
                            if (!ok)
                                error("fused job %d pair %lld: sample at offset %lld is not sample %lld of stream tag %d", ji, p, j,
                                      (long long)a0 + j, tag);
                        }
                        // (single: the segment is the pair's new samples, half a chunk behind the source pointer)
                        const uint64_t s0 = (a0 + (uint64_t)n * p + (b.single ? n / 2 : 0)) / (unsigned)hop;
                        mark(world().seg, tag, s0);
                        if (!b.single)
                            mark(world().seg, tag, s0 + 1);
                        segs += b.single ? 1 : 2;
                        // the pair's N new samples -> N/8 outputs of the next stage's stream
                        const uint64_t m0 = (a0 + (uint64_t)n * p + n / 2) / 8;
                        float *o = job.dst + (size_t)p * (n / 8);
                        for (int u = 0; u < n / 8; ++u) {
                            mark(world().dec, tag, m0 + u);
                            if (m0 + u < (uint64_t)HBF_DRAIN) {
                                error("fused job %d pair %lld: output %llu lies inside the drain", ji, p, (unsigned long long)(m0 + u));
                                break;
                            }
                            put_bits(o + u, ident(tag + 1, m0 + u - HBF_DRAIN));
                        }
                    }
                }
                write_partial_row(job.partial + (size_t)wb * n, n, segs);
            }
        }
        for (int q = 0; q < ax.npre; ++q)
            if (pre_used[(size_t)q] != 1)
                error("fused launch: prologue copy %d is carried by %d jobs", q, (int)pre_used[(size_t)q]);
        if (!aux_first)
            aux_roles();
        if (blocks != b.nblocks)
            error("fused launch: %d workgroups named by the jobs, grid of %d", blocks, b.nblocks);
        if (big_blocks > fused_max_blocks(n))
            world().launches_over_cap += 1;
        // frame groups: four jobs, traces 0..3 of one span at one offset, equal workgroup counts, inside the grid
        for (int g = 0; g < b.n_fgroups; ++g) {
            int ji = -1;
            for (int k = 0; k < b.njobs; ++k)
                if (b.jobs[k].block_begin == b.fg_begin[g])
                    ji = k;
            if (ji < 0 || ji + 3 >= b.njobs) {
                error("fused launch: frame group %d does not start at a job", g);
                continue;
            }
            for (int c = 0; c < 4; ++c)
                if (b.jobs[ji + c].fch != c || b.jobs[ji + c].fspan != b.jobs[ji].fspan || b.jobs[ji + c].nblocks != b.fg_nb[g] ||
                    b.jobs[ji + c].s_off != b.jobs[ji].s_off)
                    error("fused launch: frame group %d is not four traces of one span with %d workgroups each", g, b.fg_nb[g]);
        }
    });
    return hipSuccess;
}

// ---- hbf_dec8_kernel --------------------------------------------------------------------------------------------------
hipError_t launch_dec(const DecBatch &b, hipStream_t s)
{
    if (b.ntiles <= 0)
        return hipSuccess;
    sim::enqueue(s, [b] {
        int tiles = 0;
        for (int ji = 0; ji < b.njobs; ++ji) {
            const DecJob &job = b.jobs[ji];
            world().dec_jobs += 1;
            if (job.tile_begin != tiles)
                error("decimator job %d: tile_begin %d, expected %d", ji, job.tile_begin, tiles);
            const int nt = (job.nout + DEC_TILE - 1) / DEC_TILE;
            tiles += nt;
            for (int lt = 0; lt < nt; ++lt) {
                const long long mt0 = job.m0 + (long long)lt * DEC_TILE;
                const int nvalid = (int)std::min<long long>(DEC_TILE, job.m0 + job.nout - mt0);
                const long long x0 = 8 * mt0 - HBF_HALO, x_end = 8 * (mt0 + nvalid);
                int tag = -1;
                bool ok = true;
                for (long long i0 = std::max<long long>(x0, 0); i0 + 1 < x_end && ok; i0 += 2) {
                    for (int e = 0; e < 2 && ok; ++e) {
                        uint32_t got;
                        if (job.fspan >= 0) {
                            got = frame_ident(b.fspans[job.fspan], job.fch, (unsigned long long)(i0 + e - job.src_base + job.s_off), 2,
                                              "decimator (frames)");
                        } else {
                            got = bits_of(job.src + (i0 + e - job.src_base));
                        }
                        if (tag < 0)
                            tag = (int)(got >> 24);
                        ok = got == ident(tag, (uint64_t)(i0 + e));
                    }
                    if (!ok)
                        error("decimator job %d: input %lld of stream tag %d is not there (outputs %lld..)", ji, i0, tag, mt0);
                }
                if (tag < 0)
                    continue;
                for (int t = 0; t < nvalid; ++t) {
                    mark(world().dec, tag, (uint64_t)(mt0 + t));
                    const long long o = mt0 + t - b.drain;
                    if (o >= 0)
                        put_bits(job.dst + (o - job.dst_base), ident(tag + 1, (uint64_t)o));
                }
            }
        }
        if (tiles != b.ntiles)
            error("decimator launch: %d tiles named by the jobs, grid of %d", tiles, b.ntiles);
    });
    return hipSuccess;
}

// ---- post_kernel: the fold of a round's partials and its copy jobs (seams, carried tails) ---------------------------
hipError_t launch_post(const RedBatch &red, const TailBatch &tail, hipStream_t s)
{
    if (red.njobs + tail.njobs <= 0)
        return hipSuccess;
    sim::enqueue(s, [red, tail] {
        const int n = red.n;
        for (int ji = 0; ji < red.njobs; ++ji) {
            const RedJob &job = red.jobs[ji];
            world().red_jobs += 1;
            std::vector<double> acc((size_t)n / 2 + 1, 0.0);
            for (int t = 0; t < job.nparts; ++t) {
                float *row = const_cast<float *>(job.partial) + (size_t)t * n;
                if (row[1] != 1.0f)
                    error("fold job %d: partial row %d of %d was not written by this round's kernels (or is folded twice)", ji, t, job.nparts);
                row[1] = 0.0f; // consumed
                for (int k = 0; k <= n / 2; ++k)
                    acc[(size_t)k] += (double)row[k] + (double)row[k ? n - k : 0];
            }
            for (int k = 0; k <= n / 2; ++k)
                job.spectrum[k] = job.g_total * job.spectrum[k] + (float)(0.5 * acc[(size_t)k]);
        }
        for (int ji = 0; ji < tail.njobs; ++ji) {
            const TailJob &job = tail.jobs[ji];
            world().tail_jobs += 1;
            if (job.fspan >= 0) {
                world().tail_frame_jobs += 1;
                for (int i = 0; i < job.count; ++i)
                    put_bits(job.dst + i, frame_ident(tail.fspans[job.fspan], job.fch, (unsigned long long)job.s_off + (unsigned)i, 2, "copy (frames)"));
            } else {
                for (int i = 0; i < job.count; ++i) // (forwards, element by element, as the kernel's threads do)
                    job.dst[i] = job.src[i];
            }
        }
    });
    return hipSuccess;
}

hipError_t launch_copy_out(float *h_dst, const float *d_src, size_t count, hipStream_t s)
{
    sim::enqueue(s, [=] { memcpy(h_dst, d_src, sizeof(float) * count); });
    return hipSuccess;
}

hipError_t launch_fill_noise(float *d_x, size_t len, uint64_t, uint64_t, hipStream_t s)
{
    sim::enqueue(s, [=] {
        for (size_t i = 0; i < len; ++i)
            d_x[i] = 0.0f;
    });
    return hipSuccess;
}

// ---- adcdac_kernel: the decode path (frames -> the four stage-0 stream buffers) ----------------------------------------
hipError_t launch_adcdac(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, float *d0, float *d1, float *d2,
                         float *d3, hipStream_t s)
{
    sim::enqueue(s, [=] {
        FrameSpan fs{};
        fs.frames = frames;
        fs.bytes = (unsigned long long)n_frames * frame_size;
        fs.frame_size = (unsigned)frame_size;
        fs.batches = (unsigned)batches;
        float *dst[4] = {d0, d1, d2, d3};
        for (int ch = 0; ch < 4; ++ch)
            for (size_t i = 0; i < n_frames * (size_t)batches * 8; ++i)
                put_bits(dst[ch] + i, frame_ident(fs, ch, i, 2, "adcdac decode"));
    });
    return hipSuccess;
}

// ---- payload_kernel (Fls / ThermostatEem / Mpll): every payload byte is read, one sample per batch and trace written.  No scenario
// of round_plan_check feeds these formats (their ingest shares ingest_frames_host with AdcDac, which the frames scenario drives):
// the model keeps the link complete and the accesses honest.
hipError_t launch_payload(int fmt, const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, float *d0, float *d1, float *d2,
                          float *d3, hipStream_t s)
{
    sim::enqueue(s, [=] {
        const size_t bb = fmt == 2 ? 56 : fmt == 3 ? 80 : 24;
        float *dst[4] = {d0, d1, d2, d3};
        for (size_t f = 0; f < n_frames; ++f)
            for (int b = 0; b < batches; ++b) {
                const uint8_t *p = frames + f * frame_size + 8 + (size_t)b * bb;
                unsigned acc = 0;
                for (size_t i = 0; i < bb; ++i)
                    acc += p[i];
                for (int ch = 0; ch < (fmt == 4 ? 3 : 4); ++ch)
                    dst[ch][f * (size_t)batches + (size_t)b] = (float)(acc & 1);
            }
    });
    return hipSuccess;
}

// ---- header_gather_kernel: the 8 header bytes of every frame into pinned host memory ----------------------------------------
hipError_t launch_header_gather(const uint8_t *frames, size_t frame_size, size_t n_frames, void *out_pinned, hipStream_t s)
{
    sim::enqueue(s, [=] {
        for (size_t f = 0; f < n_frames; ++f)
            memcpy(static_cast<uint8_t *>(out_pinned) + 8 * f, frames + f * frame_size, 8);
    });
    return hipSuccess;
}

// ---- adcdac_verdict_kernel: Header::parse + the AdcDac size checks + Loss::update, as the kernel defines its four words --
hipError_t launch_adcdac_verdict(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, int payload_ok, int check,
                                 size_t n_loss, unsigned long long *acc, unsigned long long *host_out, hipStream_t s)
{
    if (n_frames == 0)
        return hipSuccess;
    if (!acc || !host_out || !s)
        error("frame verdict launched with a half-built scan state");
    sim::enqueue(s, [=] {
        for (int i = 0; i < 5; ++i)
            if (acc[i] != 0)
                error("frame verdict: accumulator word %d not zero before the launch", i);
        unsigned long long bad = 0, rec = 0, drop = 0, w3 = 0;
        auto word = [&](size_t f, int k) {
            const uint8_t *p = frames + f * frame_size + 4 * k;
            return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        };
        for (size_t f = 0; f < n_frames; ++f) {
            const uint32_t w = word(f, 0), sq = word(f, 1), bt = w >> 24;
            if (check) {
                int code = 0;
                if ((w & 0xffffu) != 0x057bu)
                    code = 1;
                else if (((w >> 16) & 0xffu) != 1u)
                    code = 2;
                else if (!payload_ok || (int)bt != batches)
                    code = 3;
                if (code)
                    bad = std::max(bad, ~(((unsigned long long)f << 2) | (unsigned long long)code));
            }
            if (f < n_loss) {
                rec += bt;
                if (f > 0)
                    drop += (uint32_t)(sq - (word(f - 1, 1) + (word(f - 1, 0) >> 24)));
                if (f == 0)
                    w3 |= sq;
                if (f == n_loss - 1)
                    w3 |= (unsigned long long)(uint32_t)(sq + bt) << 32;
            }
        }
        host_out[0] = bad;
        host_out[1] = rec;
        host_out[2] = drop;
        host_out[3] = w3;
    });
    return hipSuccess;
}

} // namespace psdk
