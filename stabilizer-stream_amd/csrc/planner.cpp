// planner.cpp -- one round of the cascade pipeline (PsdCascade::process, src/psd.rs:456-468, for every channel at once): the
// segments and decimator outputs each (channel, stage) owes are turned into kernel jobs and launched.
#include "host_runtime.h"

#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <thread>

#include <algorithm>
#include <chrono>

using namespace psdrt;

namespace {

struct Span { // one contiguous source of a (channel, stage) batch
    const float *src;
    uint64_t src_base;
    uint64_t seg_a, seg_b; // segments [seg_a, seg_b)
    uint64_t m_a, m_b;     // decimator outputs [m_a, m_b)
    bool fixed = false;    // src is not the start of the stage's stream buffer (caller memory or a
                           // seam region inside the buffer): leave it alone when buffers are re-based
    int fpool = -1;        // >= 0: the source is trace fch of frame span fs_pool[fpool] read in place (src == nullptr)
    int fch = 0;
    int pre_first = -1, pre_count = 0; // the seam copies this (buffer-side) source needs in front of its first read: Round::seams[...]
};

struct Work {
    uint32_t c, k;
    uint64_t j_old, j_new, p_old, p_new;
    EwmaPlan ew;
    Span spans[2 * MAX_COALESCE];
    int nspans = 0;
};

} // namespace

namespace {
// $PSDC_DBG_HOST_TIMING (debugging aid): where the HOST time of a round goes -- totals printed at exit
struct HostTiming {
    bool on = getenv("PSDC_DBG_HOST_TIMING") != nullptr;
    double round = 0, post = 0, fused = 0, other_launch = 0;
    unsigned long rounds = 0, spans = 0, span_hist[66] = {};
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~HostTiming()
    {
        if (on && rounds)
            fprintf(stderr, "host timing: %lu rounds, %.2f us a round: post launch %.2f, fused launch %.2f, other launches %.2f, planning %.2f\n",
                    rounds, 1e6 * round / rounds, 1e6 * post / rounds, 1e6 * fused / rounds, 1e6 * other_launch / rounds,
                    1e6 * (round - post - fused - other_launch) / rounds);
        if (on && rounds) {
            fprintf(stderr, "host timing: %lu spans held when their rounds went out; rounds by spans:", spans);
            for (int i = 0; i < 66; ++i)
                if (span_hist[i])
                    fprintf(stderr, " %d: %lu", i, span_hist[i]);
            fprintf(stderr, "\n");
        }
    }
} g_ht;
struct HtScope {
    double &acc;
    double t0;
    explicit HtScope(double &a) : acc(a), t0(g_ht.on ? HostTiming::now() : 0.0) {}
    ~HtScope()
    {
        if (g_ht.on)
            acc += HostTiming::now() - t0;
    }
};
} // namespace

namespace psdrt {

// The fused single-pass kernels read the window from its table and assume nothing about it but a hop of N/2: Window::hann()
// and every caller-built Window<N> with overlap N/2 (Hamming, Blackman, ... -- src/psd.rs:12-20 has pub fields) run on them;
// overlap 0 (Window::rectangular(), src/psd.rs:24-32, or a caller's table) runs the same kernels in their SINGLE form -- one
// segment per "pair", transformed with a zero imaginary part; the decimator consumes the stream exactly as before (round 4:
// rectangular windows ran the generic two-pass kernels at a third of the rate).  Other overlaps take the generic kernels.
// 0: no fused kernel for this window, 1: half-overlapped pairs, 2: single segments.
int fused_window(const psdc_handle *h)
{
    if (h->window_kind == PSDC_WINDOW_HANN || (h->window_kind == PSDC_WINDOW_CUSTOM && 2 * (uint64_t)h->geo.overlap == h->n))
        return 1;
    static const bool no_single = getenv("PSDC_NO_SINGLE") != nullptr; // (A/B aid: rectangular windows on the generic kernels)
    if (h->geo.overlap == 0 && !no_single)
        return 2;
    return 0;
}


} // namespace psdrt

namespace {

struct Region { // seam region of span i >= 1 of a channel
    size_t off;    // floats from the start of the stream buffer
    uint64_t base; // absolute index of its first sample (the keep_from point after span i-1)
};
struct PlanFused {
    FusedJob j;
    size_t work;
};
struct PlanSeg {
    SegJob j;
    size_t work;
};

// One round of the cascade pipeline, phase by phase (advance_round runs them in order).  Every (channel, stage) that has complete
// segments in its stream buffer is issued, all stages in the SAME launches; the decimator output of this round becomes visible to
// the next stage in the next round (stage k+1 lags one round behind stage k), so a steady-state round costs two launches whatever
// the depth: a post launch (the seam copies of this round with the deferred epilogue of the last one) and the fused launch (plus
// the generic welch / decimator kernels when something does not fit a pair).
struct Round {
    psdc_handle *h;
    const Geometry &g;
    const bool all; // issue odd segments of decimated stages too (read-outs); the ingest path leaves them for their partner
    const int spt;
    const int fmode; // 0: no fused kernel for this window, 1: half-overlapped pairs, 2: single segments (fused_window)
    const bool fast_ok, single;
    bool dbl;        // overlap 0 at the team-kernel sizes: two disjoint segments per transform (FusedBatch::single == 2) -- jobs and runs
                     // then hold an even number of single-segment "pairs".  $PSDC_NO_DOUBLE: one segment per transform everywhere (A/B aid)
    const unsigned fstep; // segments per fused "pair"
    const uint64_t half;
    // fused runs rebuild their decimator state from the 288 samples before their first new
    // sample (which sits N/2 after the run's first segment start): samples needed in front of it
    const uint64_t need_pre;
    // the seam must complete every segment that starts in the carried tail; on the fast path it
    // is long enough for the tail side to end on a whole segment pair with need_pre samples of
    // the new span in front of the in-place side
    const uint64_t seam;
    const uint64_t teams; // teams (= pairs in flight) per workgroup of the fused kernel
    size_t fused_jpl = MAX_JOBS; // fused jobs per launch (share_workgroups)

    std::vector<std::vector<Region>> regions; // per channel: the seam regions of its spans
    std::vector<TailJob> seams;               // this round's seam copies (the head of every span behind the tail it continues)
    struct SeamRef {
        int first = -1, count = 0;
    };
    std::vector<std::vector<SeamRef>> seam_ref; // per (channel, span): its copies in `seams`
    std::vector<TailJob> tjobs;                 // this round's tail carries (book)
    std::vector<Work> works;                  // what every (channel, stage) owes this round
    std::vector<PlanFused> fjobs;
    std::vector<PlanSeg> sjobs;
    std::vector<DecJob> djobs;
    std::vector<RedJob> rjobs;                // one per work: the fold of its partials (the deferred epilogue)
    uint64_t prof_samples = 0, prof_samples0 = 0;
    size_t blocks_total = 0;

    Round(psdc_handle *h_, bool all_)
        : h(h_), g(h_->geo), all(all_), spt(welch_segments_per_tile((int)h_->n)), fmode(fused_supported((int)h_->n) ? fused_window(h_) : 0),
          fast_ok(fmode != 0), single(fmode == 2), fstep(fmode == 2 ? 1 : 2), half((uint64_t)h_->n / 2),
          need_pre(HBF_HALO > (uint64_t)h_->n / 2 ? HBF_HALO - (uint64_t)h_->n / 2 : 0),
          seam(std::max<uint64_t>((uint64_t)h_->n + HBF_HALO, fmode != 0 ? need_pre + 3 * half : 0)),
          teams((uint64_t)std::max(1, fused_pairs_per_block((int)h_->n, 1))), regions(h_->n_channels), seam_ref(h_->n_channels)
    {
        static const bool no_double = getenv("PSDC_NO_DOUBLE") != nullptr;
        dbl = single && !no_double && fused_double_supported((int)h->n);
    }
    // a fused run starting at segment j decimates from sample j hop + N/2 on (half-overlapped pairs: the pair's new samples) -- or,
    // overlap 0, from j N on: a single-segment step transforms exactly the samples it decimates, and its source pointer sits half
    // a segment in front of the segment (run_src0)
    uint64_t run_new0(uint64_t seg) const { return seg * (uint64_t)g.hop + (single ? 0 : half); }
    uint64_t run_src0(uint64_t seg) const { return seg * (uint64_t)g.hop - (single ? half : 0); } // (seg >= 1 in single mode)
    // where a stream will start after this round (its tail is carried to the
    // front of its other buffer; the decimator appends the new samples behind it)
    uint64_t kf_after(uint32_t ci, uint32_t k) const
    {
        StageState t = h->ch[ci].st[k];
        for (auto &w : works)
            if (w.c == ci && w.k == k) {
                t.segs = w.j_new;
                t.dec = w.p_new;
            }
        return keep_from(g, t);
    }

    int place_seams();
    int collect_work();
    int size_next_stages();
    void make_jobs();
    void share_workgroups();
    int place_partials();
    int launch(const FusedAux *aux);
    int book();
    bool fold_tails() const;
    bool fold_seams() const;
    int run_launches();
};

int Round::place_seams()
{
    // zero-copy spans: copy the seam (the part that completes segments begun in
    // the carried tail) behind the tail; the bulk is read in place.  One copy
    // launch for all channels.  A channel may hold several spans (PSDC_OPT_COALESCE): each
    // further span gets a seam REGION of its own in the stream buffer, behind the contiguous
    // part -- the tail the span before it would have carried (read from that span's end) followed
    // by the head of the span -- so that the segments straddling two spans see contiguous memory.
    // (The copies are only COLLECTED here: run_launches decides whether they ride as prologues of the jobs that read them -- a round
    // of one launch -- or go out in a post launch in front of the fused one.)
    for (uint32_t ci = 0; ci < h->n_channels; ++ci) {
        Channel &c = h->ch[ci];
        if (!c.has_span())
            continue;
        StageState &s0 = c.st[0];
        const size_t ns = c.spans.size();
        // contiguous part: the carried tail + the seam of the first span
        const uint64_t cp0 = std::min<uint64_t>(seam, c.spans[0].len);
        size_t need = (size_t)(c.spans[0].first + cp0 - s0.buf.base);
        regions[ci].resize(ns);
        seam_ref[ci].resize(ns);
        for (size_t i = 1; i < ns; ++i) {
            StageState t; // the stage as it stands once span i-1 is consumed
            t.segs = segments_for(g, c.spans[i].first);
            t.dec = decimated_prefix(g, t.segs);
            const uint64_t kf = keep_from(g, t);
            if (kf < c.spans[i - 1].first)
                return fail(h, PSDC_ERR_DEVICE, "internal: coalesced span shorter than the carried tail");
            regions[ci][i] = {need, kf};
            need += (size_t)(c.spans[i].first - kf) + (size_t)std::min<uint64_t>(seam, c.spans[i].len);
        }
        int rc = ensure_room(h, s0, s0.buf.base + need);
        if (rc)
            return rc;
        float *buf = s0.buf.p[s0.buf.cur];
        seam_ref[ci][0] = {(int)seams.size(), 1};
        seams.push_back(span_copy(h, c.spans[0], c.spans[0].first, buf + (c.spans[0].first - s0.buf.base), (size_t)cp0));
        s0.buf.end = c.spans[0].first + cp0;
        for (size_t i = 1; i < ns; ++i) {
            const Region &r = regions[ci][i];
            const DeviceSpan &pv = c.spans[i - 1], &sp = c.spans[i];
            const size_t back = (size_t)(sp.first - r.base);
            seam_ref[ci][i] = {(int)seams.size(), 2};
            seams.push_back(span_copy(h, pv, r.base, buf + r.off, back));
            seams.push_back(span_copy(h, sp, sp.first, buf + r.off + back, (size_t)std::min<uint64_t>(seam, sp.len)));
        }
    }
    return PSDC_OK;
}

// the work of this round from the totals as they stand now
int Round::collect_work()
{
    for (uint32_t ci = 0; ci < h->n_channels; ++ci) {
        Channel &c = h->ch[ci];
        for (uint32_t k = 0; k < c.st.size(); ++k) {
            StageState &s = c.st[k];
            if (s.sink)
                continue; // handed to the caller as it is (psdc_stage_process)
            uint64_t j_new = segments_for(g, s.total);
            // ingest path (all == false): a decimated stage issues whole segment pairs only, the odd
            // segment waits for its partner -- it would cost a launch of the generic kernels every
            // other round; read-outs (all == true) issue everything
            if (!all && fmode == 1 && k >= 1 && ((j_new - s.segs) & 1))
                j_new -= 1;
            // ... and a decimated stage waits until it has a worthwhile batch: every job of a launch occupies at
            // least one resident workgroup for the whole launch, and the deep stages of many channels (a handful
            // of pairs per round each) otherwise hold ~10 % of the GPU's workgroup slots nearly idle (8 channels:
            // 578 -> 6xx GS/s).  The pending samples simply stay in the stage's stream buffer (<= 1 MiB).
            if (!all && fast_ok && k >= 1 && j_new - s.segs < 2 * (uint64_t)h->min_pairs)
                j_new = s.segs;
            if (j_new == s.segs)
                continue;
            Work w;
            w.c = ci;
            w.k = k;
            w.j_old = s.segs;
            w.j_new = j_new;
            w.p_old = s.dec;
            w.p_new = decimated_prefix(g, j_new);
            w.ew = plan_ewma(s.count, cur_stage_avg(h, k), j_new - s.segs);
            const uint64_t m_old = w.p_old >> 3, m_new = w.p_new >> 3;
            if (k == 0 && c.has_span()) {
                // span by span, each exactly as a round of its own would split it: the buffer side
                // (carried tail + seam: contiguous part for the first span, its seam region for the
                // others) and the in-place side
                uint64_t j_lo = w.j_old, m_lo = m_old;
                const size_t ns = c.spans.size();
                for (size_t i = 0; i < ns; ++i) {
                    const DeviceSpan &sp = c.spans[i];
                    const uint64_t first = sp.first;
                    const uint64_t j_hi = i + 1 < ns ? segments_for(g, c.spans[i + 1].first) : j_new;
                    const uint64_t m_hi = i + 1 < ns ? decimated_prefix(g, j_hi) >> 3 : m_new;
                    uint64_t j_split =
                        std::min<uint64_t>(j_hi, std::max<uint64_t>(j_lo, (first + g.hop - 1) / g.hop));
                    uint64_t m_split =
                        std::min<uint64_t>(m_hi, std::max<uint64_t>(m_lo, (first + HBF_HALO + 7) / 8));
                    if (fast_ok && j_lo > 0) {
                        // fast path: the tail side gets a whole number of segment pairs and exactly their
                        // decimator outputs; the in-place side starts >= need_pre samples into the span
                        uint64_t js = std::max<uint64_t>(j_lo, (first + need_pre + (single ? half : 0) + g.hop - 1) / g.hop);
                        if (!single && ((js - j_lo) & 1))
                            js += 1;
                        if (js < j_hi && run_new0(js) <= first + seam && sp.len >= seam) {
                            j_split = js;
                            m_split = std::min<uint64_t>(m_hi, run_new0(js) / 8);
                        }
                    }
                    if (j_split > j_lo || m_split > m_lo) {
                        if (i == 0)
                            w.spans[w.nspans++] = {s.buf.p[s.buf.cur], s.buf.base, j_lo, j_split, m_lo, m_split, false};
                        else
                            w.spans[w.nspans++] = {s.buf.p[s.buf.cur] + regions[ci][i].off, regions[ci][i].base,
                                                   j_lo, j_split, m_lo, m_split, true};
                        w.spans[w.nspans - 1].pre_first = seam_ref[ci][i].first;
                        w.spans[w.nspans - 1].pre_count = seam_ref[ci][i].count;
                    }
                    if (j_hi > j_split || m_hi > m_split) {
                        Span ip{sp.d_x, first, j_split, j_hi, m_split, m_hi, true};
                        if (sp.framed()) {
                            ip.fpool = pool_fspan(h, sp.fs);
                            ip.fch = sp.fch;
                        }
                        w.spans[w.nspans++] = ip;
                    }
                    j_lo = j_hi;
                    m_lo = m_hi;
                }
            } else {
                w.spans[w.nspans++] = {s.buf.p[s.buf.cur], s.buf.base, w.j_old, j_new, m_old, m_new, false};
            }
            works.push_back(w);
        }
    }
    return PSDC_OK;
}

// room for what the decimators of this round will append to the next stages' streams
int Round::size_next_stages()
{
    for (auto &w : works) {
        Channel &c = h->ch[w.c];
        const uint64_t t_next = emitted_for(g, w.p_new);
        if (t_next > 0) {
            if (c.st.size() <= (size_t)w.k + 1) {
                int rc = add_stage(h, c);
                if (rc)
                    return rc;
            }
            StageState &nx = c.st[w.k + 1];
            // Rounds hold 1 ... PSDC_OPT_COALESCE in-place spans, and stage j sees a round's samples j rounds
            // later: when a stream has to grow it grows at once to what the LARGEST round will bring it
            // (longest span seen x the coalescing depth once a round has coalesced, / 8^j), not round size by
            // round size -- every growth is two allocations and a copy in the middle of a live stream.
            const size_t need = (size_t)(t_next - kf_after(w.c, w.k + 1));
            size_t grow_to = 0;
            if (c.span_max) {
                const uint64_t round_max = std::min<uint64_t>((uint64_t)c.span_max * (c.coalesced_seen ? coalesce_limit(h, c) : 1),
                                                              std::max<uint64_t>(c.span_max, hold_max(h))); // (a channel never holds more)
                const unsigned sh = 3u * (w.k + 1);
                grow_to = (size_t)(sh < 64 ? round_max >> sh : 0) + (size_t)4 * (h->n + HBF_HALO) + 64;
            }
            int rc = ensure_cap(h, nx, need, grow_to);
            if (rc)
                return rc;
        }
    }
    // stream buffers may have been reallocated by ensure_room: refresh span pointers
    for (auto &w : works) {
        StageState &s = h->ch[w.c].st[w.k];
        for (int i = 0; i < w.nspans; ++i)
            if (!w.spans[i].fixed) {
                w.spans[i].src = s.buf.p[s.buf.cur];
                w.spans[i].src_base = s.buf.base;
            }
    }
    return PSDC_OK;
}

void Round::make_jobs()
{
    // ---- turn the work into kernel jobs ---------------------------------
    // Fast path (fused_kernel / bigfused_kernel): whole segment pairs of a Hann stream with
    // N = 256 ... 16384, any implemented detrend, plain-sum or EWMA averaging, 16-byte aligned.
    // Everything else -- other N, the rectangular window, an odd last segment, the decimator
    // ranges a pair does not cover (the first segment of a stream is decimated whole,
    // src/psd.rs:235-238; outputs still inside the drain; unaligned zero-copy spans) -- goes
    // through the generic welch / hbf_dec8 kernels.  Both write the same partial slab and
    // next-stage stream, so the reduce and the bookkeeping do not care which ran.
    for (size_t wi = 0; wi < works.size(); ++wi) {
        Work &w = works[wi];
        Channel &c = h->ch[w.c];
        const uint64_t t_next = emitted_for(g, w.p_new);
        StageState *nx = t_next > 0 ? &c.st[w.k + 1] : nullptr;
        const uint64_t nx_base = nx ? kf_after(w.c, w.k + 1) : 0;
        auto add_dec = [&](const Span &sp, uint64_t ma, uint64_t mb) {
            if (mb <= ma || mb <= g.drain || !nx)
                return;
            DecJob dj{};
            dj.src = sp.src;
            dj.fspan = sp.fpool;
            dj.fch = sp.fch;
            dj.src_base = (long long)sp.src_base;
            dj.m0 = (long long)ma;
            dj.dst = nx->buf.p[nx->buf.cur ^ 1];
            dj.dst_base = (long long)nx_base;
            dj.nout = (int)(mb - ma);
            djobs.push_back(dj);
        };
        auto add_seg = [&](const Span &sp, uint64_t sa, uint64_t sb) {
            if (sb <= sa)
                return;
            SegJob sj{};
            sj.src = sp.src;
            sj.fspan = sp.fpool;
            sj.fch = sp.fch;
            sj.src_base = (long long)sp.src_base;
            sj.seg0 = (long long)sa;
            sj.log2_gamma = w.ew.gamma > 0.0f ? std::log2((double)w.ew.gamma)
                                              : -std::numeric_limits<double>::infinity();
            sj.nseg = (int)(sb - sa);
            sj.ntiles = (int)((sb - sa + spt - 1) / spt);
            sj.step0 = (int)(sa - w.j_old) + 1;
            sj.nb = (int)w.ew.nb;
            sj.is_m1 = (int)std::min<int64_t>(w.ew.i_s - 1, std::numeric_limits<int>::max());
            sj.ewma = w.ew.ewma ? 1 : 0;
            sjobs.push_back({sj, wi});
        };
        for (int i = 0; i < w.nspans; ++i) {
            const Span &sp = w.spans[i];
            // first fused segment: far enough into the stream that every output survives the drain
            uint64_t fs = sp.seg_a;
            while (run_new0(fs) / 8 < g.drain)
                fs += fstep;
            const bool framed = sp.fpool >= 0;
            uint64_t np = (fast_ok && nx && fs + fstep <= sp.seg_b && (!framed || (!single && fused_frames_supported((int)h->n))))
                              ? (sp.seg_b - fs) / fstep : 0;
            if (single && np) {
                if (run_src0(fs) < sp.src_base) // (the half chunk in front of the first segment is not in this source)
                    np = 0;
                if (dbl)
                    np &= ~(uint64_t)1; // (an odd last segment goes to the generic kernels)
            }
            const uint64_t fofs = np ? run_src0(fs) - sp.src_base : 0; // samples of this span in front of the pairs' source
            const float *fsrc = framed ? nullptr : sp.src + fofs;
            const uint64_t mf0 = run_new0(fs) / 8, mf1 = mf0 + (h->n / 8) * np;
            const bool aligned = framed ? (fofs & 3u) == 0 : (reinterpret_cast<uintptr_t>(fsrc) & 15u) == 0;
            if (np && (!aligned || mf0 < sp.m_a || mf1 > sp.m_b || (fofs < need_pre && sp.src_base != 0)))
                np = 0;
            if (np) {
                FusedJob fj{};
                // (a frame job never reads through src, but the kernels form per-lane pointers from it before they know: keep that
                // arithmetic off a null pointer -- any valid device address will do)
                fj.src = framed ? h->d_win : fsrc;
                fj.fspan = sp.fpool;
                fj.fch = sp.fch;
                fj.s_off = framed ? (unsigned)fofs : 0u;
                fj.dst = nx->buf.p[nx->buf.cur ^ 1] + (mf0 - g.drain - nx_base);
                fj.npairs = (int)np;
                fj.pre = (int)std::min<uint64_t>(fofs, HBF_HALO);
                fj.log2_gamma = w.ew.gamma > 0.0f ? std::log2((double)w.ew.gamma)
                                                  : -std::numeric_limits<double>::infinity();
                fj.step0 = (int)(fs - w.j_old) + 1;
                fj.nb = (int)w.ew.nb;
                fj.is_m1 = (int)std::min<int64_t>(w.ew.i_s - 1, std::numeric_limits<int>::max());
                fj.ewma = w.ew.ewma ? 1 : 0;
                fj.pre_first = sp.pre_first; // (indices into Round::seams until run_launches maps them into the launch's table)
                fj.pre_count = sp.pre_count;
                fjobs.push_back({fj, wi});
                add_seg(sp, sp.seg_a, fs);
                add_seg(sp, fs + fstep * np, sp.seg_b);
                add_dec(sp, sp.m_a, mf0);
                add_dec(sp, mf1, sp.m_b);
            } else {
                add_seg(sp, sp.seg_a, sp.seg_b);
                add_dec(sp, sp.m_a, sp.m_b);
            }
        }
        prof_samples += w.p_new - w.p_old;
        if (w.k == 0)
            prof_samples0 += w.p_new - w.p_old;
    }
}

void Round::share_workgroups()
{
    // share the persistent workgroups so that every workgroup walks about the same amount
    // shares are computed per launch batch (MAX_JOBS jobs): every launch fills the GPU by itself
    for (size_t b0 = 0; b0 < sjobs.size(); b0 += MAX_JOBS) {
        const size_t b1 = std::min(sjobs.size(), b0 + (size_t)MAX_JOBS);
        size_t tiles = 0;
        for (size_t i = b0; i < b1; ++i)
            tiles += (size_t)sjobs[i].j.ntiles;
        const size_t per = std::max<size_t>(1, (tiles + WELCH_MAX_BLOCKS - 1) / WELCH_MAX_BLOCKS);
        for (size_t i = b0; i < b1; ++i) {
            sjobs[i].j.nblocks = (int)(((size_t)sjobs[i].j.ntiles + per - 1) / per);
            blocks_total += (size_t)sjobs[i].j.nblocks;
        }
    }
    // Jobs that read frames in place go first in their launch, the four traces of one span side by side (same span, same
    // offset: the same geometry, hence equal workgroup counts): the kernel deals such a group's workgroups over the XCDs so
    // that the four readers of the same bytes share an L2 (FusedBatch::fg_*).
    bool any_frames = false;
    for (const PlanFused &pf : fjobs)
        any_frames = any_frames || pf.j.fspan >= 0;
    if (any_frames)
        std::stable_sort(fjobs.begin(), fjobs.end(), [](const PlanFused &a, const PlanFused &b) {
            const bool fa = a.j.fspan >= 0, fb = b.j.fspan >= 0;
            if (fa != fb)
                return fa;
            if (!fa)
                return false;
            if (a.j.fspan != b.j.fspan)
                return a.j.fspan < b.j.fspan;
            if (a.j.s_off != b.j.s_off)
                return a.j.s_off < b.j.s_off;
            return a.j.fch < b.j.fch;
        });
    // Fused jobs per launch: all MAX_JOBS for a single channel (a round of 128 scattered spans is then ONE launch) and for several
    // channels fed in spans of at most 2^23 samples (eight channels x 2^22: 651 against 614 GS/s, four x sixteen 2^20-sample spans 587
    // against 491); 128 for several channels fed in longer spans -- eight channels x eight 2^24-sample spans (208 jobs) measured 2.6 %
    // SLOWER as one launch than as 128 + 80 (656 against 674 GS/s, kernel-only 0.331 against 0.345: each launch gets the one run length
    // that fills the GPU for ITS jobs, and in the bench the fewer channels a launch covers the more of their re-fed buffers stay in
    // the Infinity Cache).  The longest span a channel has SEEN decides: a function of the calls alone.
#ifndef PSDK_MULTI_JPL
#define PSDK_MULTI_JPL 128
#endif
    size_t longest = 0;
    for (const Channel &c : h->ch)
        longest = std::max(longest, c.span_max);
    fused_jpl = h->n_channels == 1 || longest <= ((size_t)1 << 23) ? (size_t)MAX_JOBS : std::min<size_t>(MAX_JOBS, PSDK_MULTI_JPL);
    for (size_t b0 = 0; b0 < fjobs.size(); b0 += fused_jpl) {
        const size_t b1 = std::min(fjobs.size(), b0 + fused_jpl);
        // One run length R for the whole launch: the smallest R for which the jobs' workgroups
        // (ceil(pairs / (R teams)) each) fit the resident capacity.  A launch that asks for more
        // workgroups than are resident at once runs the surplus as a second wave behind the first.
        // (A launch that reads frames leaves a few workgroup slots free: the header scan of the NEXT call runs on a
        // side stream while this launch is resident, and would otherwise wait for it to drain.)
        bool batch_frames = false;
        for (size_t i = b0; i < b1; ++i)
            batch_frames = batch_frames || fjobs[i].j.fspan >= 0;
        // (FRAME_RESERVE_BLOCKS slots of at least 256 threads' worth of registers each)
        const uint64_t reserve = (uint64_t)FRAME_RESERVE_BLOCKS * (uint64_t)std::max(1, 256 / std::max(1, fused_block_threads((int)h->n)));
        const uint64_t max_blocks = (uint64_t)fused_max_blocks((int)h->n);
        const uint64_t cap_slots = max_blocks > reserve + 1 && batch_frames ? max_blocks - reserve : max_blocks;
        // Small jobs ride on top: a job whose one workgroup has at most a quarter of a full workgroup's work (the deep stages
        // of every channel: a handful of pairs per round) does not count against the capacity.  Its workgroup goes FIRST in
        // the grid, is resident for a few microseconds and leaves its slot to one of the surplus workgroups of the large
        // jobs, so the launch asks for cap + (small jobs) workgroups and ends about one small job later than a launch of
        // the large jobs alone -- where counting them against the capacity left their slots empty for nearly the whole launch
        // (8 channels x 8 deep stages: 64 of 512 slots).
        static const bool no_oversub = getenv("PSDC_NO_OVERSUB") != nullptr; // (A/B aid)
        auto small_at = [&](uint64_t np, uint64_t r) { return !no_oversub && 4 * np <= r * teams; };
        auto plan_r = [&](uint64_t r_small) { // the smallest R whose LARGE jobs fit the capacity, given which jobs count as small
            uint64_t pairs = 0, nlarge = 0;
            for (size_t i = b0; i < b1; ++i)
                if (!small_at((uint64_t)fjobs[i].j.npairs, r_small)) {
                    pairs += (uint64_t)fjobs[i].j.npairs;
                    ++nlarge;
                }
            // (every job holds at least one workgroup: with more large jobs than slots -- MAX_JOBS = 160 against >= 248 slots, so
            // only under the CPU model's artificially small capacities -- the search below would never end)
            const uint64_t cap = std::max<uint64_t>(cap_slots, nlarge);
            auto blocks_at = [&](uint64_t r) {
                uint64_t nb = 0;
                for (size_t i = b0; i < b1; ++i)
                    if (!small_at((uint64_t)fjobs[i].j.npairs, r_small))
                        nb += ((uint64_t)fjobs[i].j.npairs + r * teams - 1) / (r * teams);
                return nb;
            };
            uint64_t r = std::max<uint64_t>(1, (pairs + cap * teams - 1) / (cap * teams));
            while (blocks_at(r) > cap)
                ++r;
            return r;
        };
        uint64_t R = plan_r(0); // every job counted
        for (int it = 0; it < 3; ++it) {
            const uint64_t rn = plan_r(R);
            if (rn == R)
                break;
            R = rn;
        }
        // $PSDC_DBG_FIXED_RUN=<pairs> (measurement aid): ONE run length whatever the launch holds -- what run boundaries that are a
        // function of the absolute pair index (chunk-invariant grouping of the partial sums, include/psdcascade.h Conventions)
        // would cost: launches then ask for more or fewer workgroups than are resident at once.  Results stay correct.
        static const uint64_t fixed_run = getenv("PSDC_DBG_FIXED_RUN") ? strtoull(getenv("PSDC_DBG_FIXED_RUN"), nullptr, 10) : 0;
        if (fixed_run)
            R = fixed_run;
        for (size_t i = b0; i < b1; ++i) {
            FusedJob &j = fjobs[i].j;
            const uint64_t np = (uint64_t)j.npairs;
            const uint64_t nb = small_at(np, R) ? 1 : (np + R * teams - 1) / (R * teams);
            uint64_t run = (np + nb * teams - 1) / (nb * teams); // evened out within the job (<= R)
            if (dbl)
                run += run & 1; // every team starts on an even segment and holds whole segment pairs
            j.run = (int)run;
            j.nblocks = (int)((np + run * teams - 1) / (run * teams));
            blocks_total += (size_t)j.nblocks;
        }
        // small jobs first in the grid (stable: the frame groups stay together behind them)
        std::stable_partition(fjobs.begin() + (std::ptrdiff_t)b0, fjobs.begin() + (std::ptrdiff_t)b1,
                              [&](const PlanFused &pf) { return small_at((uint64_t)pf.j.npairs, R); });
    }
}

int Round::place_partials()
{
    int rc = ensure_partial(h, blocks_total * h->n);
    if (rc)
        return rc;
    // slab: the partials of one work are contiguous (fused first, then generic); this round writes slab partial_cur, the other one may
    // still hold the round before's partials until their fold has run (in this round's fused launch when the round is one launch)
    float *const slab_base = h->d_partial + (size_t)h->partial_cur * h->partial_cap;
    rjobs.assign(works.size(), RedJob{});
    {
        // (the jobs of a work need not be adjacent in fjobs: frame jobs were moved to the front)
        std::vector<size_t> nblk(works.size(), 0), base(works.size(), 0);
        for (const PlanFused &pf : fjobs)
            nblk[pf.work] += (size_t)pf.j.nblocks;
        for (const PlanSeg &ps : sjobs)
            nblk[ps.work] += (size_t)ps.j.nblocks;
        size_t slab = 0;
        for (size_t wi = 0; wi < works.size(); ++wi) {
            RedJob &rj = rjobs[wi];
            base[wi] = slab;
            rj.partial = slab_base + slab;
            rj.spectrum = h->ch[works[wi].c].st[works[wi].k].spectrum;
            rj.g_total = (float)works[wi].ew.g_total;
            rj.nparts = (int)nblk[wi];
            slab += nblk[wi] * h->n;
        }
        if (slab > h->partial_cap)
            return fail(h, PSDC_ERR_DEVICE, "internal: partial slab overflow");
        for (PlanFused &pf : fjobs) {
            pf.j.partial = slab_base + base[pf.work];
            base[pf.work] += (size_t)pf.j.nblocks * h->n;
        }
        for (PlanSeg &ps : sjobs) {
            ps.j.partial = slab_base + base[ps.work];
            base[ps.work] += (size_t)ps.j.nblocks * h->n;
        }
    }
    return PSDC_OK;
}

int Round::launch(const FusedAux *aux)
{
    // ---- launches: fused, generic welch, reduce, generic decimator -------
    // HIP events time the dominant kernel of the round (fused when present).  The fused launches
    // hand their events to hipExtLaunchKernelGGL, which stamps the kernel's own start and stop (what
    // rocprofv3 --kernel-trace reports); events recorded around a launch would include the ~6 us
    // dependent-dispatch gap in front of it.  The generic welch kernel keeps the bracket.
    int rc = PSDC_OK;
    const bool prof_fused = fast_ok; // the handle's dominant kernel kind, not the round's
    auto prof_begin = [&](ProfEvents &pe, bool record) -> int {
        if (!h->profile)
            return PSDC_OK;
        HIPCHK(h, hipEventCreate(&pe.a));
        HIPCHK(h, hipEventCreate(&pe.b));
        if (record)
            HIPCHK(h, hipEventRecord(pe.a, h->stream));
        return PSDC_OK;
    };
    auto prof_end = [&](ProfEvents &pe, bool first, bool record) -> int {
        if (!h->profile)
            return PSDC_OK;
        if (record)
            HIPCHK(h, hipEventRecord(pe.b, h->stream));
        h->prof_pending.push_back(pe);
        h->prof.launches += 1;
        if (first) {
            h->prof.samples += prof_samples;
            h->prof.stage0_samples += prof_samples0;
        }
        return PSDC_OK;
    };
    for (size_t i = 0; i < fjobs.size();) {
        FusedBatch fb{};
        fb.detrend = h->detrend;
        fb.single = dbl ? 2 : single ? 1 : 0;
        FspanMap fm(fb.fspans);
        for (; i < fjobs.size() && fb.njobs < (int)fused_jpl; ++i) {
            FusedJob j = fjobs[i].j;
            j.block_begin = fb.nblocks;
            fb.nblocks += j.nblocks;
            fb.any_ewma |= j.ewma;
            if (j.fspan >= 0) {
                fb.any_frames = 1;
                if ((j.fspan = fm.map(h, j.fspan)) < 0)
                    return fail(h, PSDC_ERR_DEVICE, "internal: frame span table");
            }
            if (aux && aux->npre && j.pre_count > 0)
                j.pre_first += aux->ntail; // its seam copies sit behind the aux workgroups' tail jobs in FusedAux::tail
            else
                j.pre_first = j.pre_count = 0;
            fb.jobs[fb.njobs++] = j;
        }
        static const bool no_groups = getenv("PSDC_DBG_NOGROUPS") != nullptr; // (debugging aid)
        for (int a = 0; !no_groups && a + 3 < fb.njobs && fb.n_fgroups < MAX_FSPANS; ) { // the four traces of one span, side by side
            const FusedJob *q = fb.jobs + a;
            const bool group = q[0].fspan >= 0 && q[0].fch == 0 && q[1].fch == 1 && q[2].fch == 2 && q[3].fch == 3 &&
                               q[1].fspan == q[0].fspan && q[2].fspan == q[0].fspan && q[3].fspan == q[0].fspan &&
                               q[1].s_off == q[0].s_off && q[2].s_off == q[0].s_off && q[3].s_off == q[0].s_off &&
                               q[1].nblocks == q[0].nblocks && q[2].nblocks == q[0].nblocks && q[3].nblocks == q[0].nblocks;
            if (!group) {
                ++a;
                continue;
            }
            fb.fg_begin[fb.n_fgroups] = q[0].block_begin;
            fb.fg_nb[fb.n_fgroups] = q[0].nblocks;
            ++fb.n_fgroups;
            a += 4;
        }
        static const bool dbg_plan = getenv("PSDC_DBG_PLAN") != nullptr; // (debugging aid: what each fused launch holds)
        if (dbg_plan) {
            long long np = 0;
            for (int q = 0; q < fb.njobs; ++q)
                np += fb.jobs[q].npairs;
            fprintf(stderr, "fused launch: %d jobs, %d workgroups (+%d aux), %lld pairs:", fb.njobs, fb.nblocks, aux ? aux->nblocks : 0, np);
            for (int q = 0; q < fb.njobs; ++q)
                fprintf(stderr, " %dp/%dwg/r%d%s", fb.jobs[q].npairs, fb.jobs[q].nblocks, fb.jobs[q].run, fb.jobs[q].pre_count ? "s" : "");
            fprintf(stderr, "\n");
        }
        ProfEvents pe{};
        const bool first = (i <= fused_jpl);
        if ((rc = prof_begin(pe, false)))
            return rc;
        {
            HtScope ht_f(g_ht.fused);
            HIPCHK(h, launch_fused((int)h->n, fb, h->d_win, h->d_tw0g, h->d_twag, h->d_tw3g, h->stream, pe.a, pe.b, aux));
        }
        if ((rc = prof_end(pe, first, false)))
            return rc;
    }
    for (size_t i = 0; i < sjobs.size();) {
        WelchBatch wb{};
        wb.hop = (int)g.hop;
        wb.detrend = h->detrend;
        FspanMap fm(wb.fspans);
        for (; i < sjobs.size() && wb.njobs < MAX_JOBS; ++i) {
            SegJob j = sjobs[i].j;
            if (j.fspan >= 0 && (j.fspan = fm.map(h, j.fspan)) < 0)
                return fail(h, PSDC_ERR_DEVICE, "internal: frame span table");
            j.block_begin = wb.nblocks;
            wb.nblocks += j.nblocks;
            wb.jobs[wb.njobs++] = j;
        }
        ProfEvents pe{};
        const bool first = (i <= (size_t)MAX_JOBS);
        if (!prof_fused && (rc = prof_begin(pe, true)))
            return rc;
        if (bigfft_size((int)h->n))
            HIPCHK(h, launch_welch_big((int)h->n, wb, h->d_win, h->d_tw, h->d_bigfft, h->bigfft_elems, h->bigfft_chunk_limit, h->stream));
        else
            HIPCHK(h, launch_welch((int)h->n, wb, h->d_win, h->d_tw, h->d_chirp, h->d_bhat, h->stream));
        if (!prof_fused && (rc = prof_end(pe, first, true)))
            return rc;
    }
    for (size_t i = 0; i < djobs.size();) {
        DecBatch db{};
        db.drain = (int)g.drain;
        FspanMap fm(db.fspans);
        for (; i < djobs.size() && db.njobs < MAX_JOBS; ++i) {
            DecJob j = djobs[i];
            if (j.fspan >= 0 && (j.fspan = fm.map(h, j.fspan)) < 0)
                return fail(h, PSDC_ERR_DEVICE, "internal: frame span table");
            j.tile_begin = db.ntiles;
            db.ntiles += (j.nout + DEC_TILE - 1) / DEC_TILE;
            db.jobs[db.njobs++] = j;
        }
        HIPCHK(h, launch_dec(db, h->stream));
    }
    return PSDC_OK;
}

int Round::book()
{
    // bookkeeping: counts and stream positions
    std::vector<std::vector<uint64_t>> old_total(h->n_channels);
    for (uint32_t ci = 0; ci < h->n_channels; ++ci)
        for (auto &s : h->ch[ci].st)
            old_total[ci].push_back(s.total);
    for (auto &w : works) {
        StageState &s = h->ch[w.c].st[w.k];
        s.count64 = count_after64(s.count64, cur_stage_avg(h, w.k), w.j_new - w.j_old);
        s.count = count_report(s.count64);
        s.segs = w.j_new;
        s.dec = w.p_new;
    }
    for (auto &w : works) {
        const uint64_t t_next = emitted_for(g, w.p_new);
        if (t_next > 0)
            h->ch[w.c].st[w.k + 1].total = t_next; // visible to the next stage from the next round on
    }

    // carry the small tail [keep_from, old total) of every stream that consumed or
    // received samples to the front of its other buffer, then swap
    tjobs.clear();
    for (uint32_t ci = 0; ci < h->n_channels; ++ci) {
        Channel &c = h->ch[ci];
        for (uint32_t k = 0; k < c.st.size(); ++k) {
            StageState &s = c.st[k];
            const uint64_t kf = keep_from(g, s);
            const uint64_t told = old_total[ci][k];
            const bool received = s.total != told;
            if (kf == s.buf.base && !received && s.buf.end == s.total)
                continue;
            const bool span0 = (k == 0 && c.has_span());
            const DeviceSpan last = span0 ? c.spans.back() : DeviceSpan{};
            const uint64_t cnt = told > kf ? told - kf : 0;
            if (span0 && kf < last.first)
                return fail(h, PSDC_ERR_DEVICE, "internal: zero-copy span tail not in the span");
            if (s.total - kf > s.buf.cap)
                return fail(h, PSDC_ERR_DEVICE, "internal: tail exceeds stream buffer");
            const int other = s.buf.cur ^ 1;
            if (cnt && span0) {
                tjobs.push_back(span_copy(h, last, kf, s.buf.p[other], (size_t)cnt));
            } else if (cnt) {
                TailJob t{};
                t.src = s.buf.p[s.buf.cur] + (kf - s.buf.base);
                t.dst = s.buf.p[other];
                t.count = (int)cnt;
                tjobs.push_back(t);
            }
            s.buf.cur = other;
            s.buf.base = kf;
            s.buf.end = s.total;
        }
    }
    for (auto &c : h->ch) {
        c.spans.clear();
        c.submitted = false;
    }
    h->partial_cur ^= 1;
    return PSDC_OK;
}

// ---- one launch per round ------------------------------------------------------------------------------------------------------
// A steady-state round used to be two launches: post_kernel (the fold of the round before, its tail carries, this round's seam copies)
// and the fused kernel.  Where the fused kernel can carry them (fused_fold_supported: N <= 1024) the three ride in the fused launch:
//   * the TAIL CARRIES of this round as aux workgroups of this round's launch: their sources -- the unconsumed end of each stream's
//     current buffer, or of the caller's last span -- are not written by the launch, their destinations (the front of the other
//     buffers) are read by nobody before the next round;
//   * the FOLD of the round before as aux workgroups too: it reads the OTHER partial slab;
//   * the SEAM copies as prologues of the single-workgroup jobs that read them (the buffer side of a span: a handful of segments).
// fold_tails: this round's launch can carry aux workgroups at all; fold_seams: and the post launch in front of it can go.
bool Round::fold_tails() const
{
    if (!h->fold || !fused_fold_supported((int)h->n) || fjobs.empty() || !sjobs.empty() || !djobs.empty() || fjobs.size() > fused_jpl)
        return false; // (a round of several fused launches: measured for eight channels x eight spans with the aux workgroups in the first of them -- -0.6 %)
    size_t pieces = 0;
    for (const TailJob &t : tjobs) {
        if (t.fspan >= 0)
            return false; // (a tail decoded from frames: post_kernel's copy role knows how)
        pieces += (size_t)(t.count + 16383) / 16384;
    }
    return pieces <= (size_t)AUX_MAX_TAIL;
}

bool Round::fold_seams() const
{
    if (!fold_tails() || !h->pend_tail.empty() || h->pend_red.size() > (size_t)AUX_MAX_RED)
        return false;
    size_t pieces = 0;
    for (const TailJob &t : tjobs)
        pieces += (size_t)(t.count + 16383) / 16384;
    if (pieces + seams.size() > (size_t)AUX_MAX_TAIL)
        return false;
    std::vector<char> used(seams.size(), 0);
    for (const TailJob &t : seams)
        if (t.fspan >= 0)
            return false;
    for (const PlanFused &pf : fjobs)
        if (pf.j.pre_count > 0) {
            if (pf.j.nblocks != 1)
                return false; // (the hand-over of a prologue is inside ONE workgroup)
            for (int q = 0; q < pf.j.pre_count; ++q)
                used[(size_t)(pf.j.pre_first + q)] = 1;
        }
    for (char u : used)
        if (!u)
            return false; // (a seam nobody's prologue carries: its reader went to the generic kernels)
    return true;
}

int Round::run_launches()
{
    const bool ft = fold_tails(), fs = ft && fold_seams();
    int rc;
    if (!fs) { // the post launch in front: the round before's epilogue with this round's seams
        HtScope ht_post(g_ht.post);
        if ((rc = launch_deferred(h, seams)))
            return rc;
    }
    FusedAux aux{};
    if (ft) {
        aux.n = (int)h->n;
        std::vector<TailJob> pieces;
        split_copy_jobs(tjobs, pieces);
        aux.ntail = (int)pieces.size();
        for (size_t i = 0; i < pieces.size(); ++i)
            aux.tail[i] = pieces[i];
        if (fs) {
            aux.npre = (int)seams.size();
            for (size_t i = 0; i < seams.size(); ++i)
                aux.tail[(size_t)aux.ntail + i] = seams[i];
            aux.nred = (int)h->pend_red.size(); // by shape: many rows (16 bins a workgroup), up to 64 (128 bins), up to 4 (every bin)
            auto shape_of = [](const RedJob &j) { return j.nparts > AUX_MID_ROWS ? 0 : j.nparts > AUX_SHORT_ROWS ? 1 : 2; };
            int at = 0;
            for (int shape = 0; shape < 3; ++shape) {
                for (const RedJob &j : h->pend_red)
                    if (shape_of(j) == shape)
                        aux.red[at++] = j;
                if (shape == 0)
                    aux.nred_tall = at;
                else if (shape == 1)
                    aux.nred_mid = at - aux.nred_tall;
            }
            aux.red_xb = ((int)h->n / 2 + 1 + AUX_RED_BINS - 1) / AUX_RED_BINS;
            aux.red_mb = (aux.red_xb + AUX_MID_GROUPS - 1) / AUX_MID_GROUPS;
            aux.red_blocks = aux.nred_tall * aux.red_xb + aux.nred_mid * aux.red_mb + (aux.nred - aux.nred_tall - aux.nred_mid);
            h->pend_red.clear();
        }
        aux.nblocks = aux.red_blocks + aux.ntail;
    }
    if ((rc = launch(ft ? &aux : nullptr)))
        return rc;
    // what is left of the epilogue waits for the next round (or a read-out): the fold of THIS round's partials always, its tail
    // carries unless they rode in the launch
    h->pend_red = std::move(rjobs);
    if (ft)
        h->pend_tail.clear();
    else
        h->pend_tail = std::move(tjobs);
    h->post_dirty = true; // the compute stream has work an upload must wait for: order_upload records the event then
    return PSDC_OK;
}

} // namespace

namespace psdrt {

// The pool of frame spans that jobs name by index lives from round to round only through the deferred tail carries: at the start of
// a round it is cut down to the spans those still name (their indices re-mapped), so that it cannot grow with the stream.
static void compact_fs_pool(psdc_handle *h)
{
    std::vector<FrameSpan> keep;
    for (TailJob &t : h->pend_tail)
        if (t.fspan >= 0) {
            const FrameSpan &fs = h->fs_pool[(size_t)t.fspan];
            int idx = -1;
            for (size_t i = 0; i < keep.size(); ++i)
                if (keep[i].frames == fs.frames && keep[i].bytes == fs.bytes && keep[i].frame_size == fs.frame_size && keep[i].batches == fs.batches)
                    idx = (int)i;
            if (idx < 0) {
                keep.push_back(fs);
                idx = (int)keep.size() - 1;
            }
            t.fspan = idx;
        }
    h->fs_pool.swap(keep);
}

// `all`: issue odd segments of decimated stages too (read-outs); the ingest path
// leaves them for their partner.  *did_work tells whether anything was issued;
// read-outs call rounds until idle.
int advance_round(psdc_handle *h, bool *did_work, bool all)
{
    *did_work = false;
    HtScope ht_round(g_ht.round);
    if (g_ht.on) {
        ++g_ht.rounds;
        size_t ns = 0;
        for (const Channel &c : h->ch)
            ns += c.spans.size();
        g_ht.spans += ns;
        g_ht.span_hist[std::min<size_t>(ns, 65)] += 1;
    }
    for (Channel &c : h->ch) { // a held span that never grew long enough to be read in place becomes a copy (runtime.cpp)
        int rc = settle_short_span(h, c);
        if (rc)
            return rc;
    }
    int rc = wait_uploads(h);
    if (rc)
        return rc;
    compact_fs_pool(h);
    Round r(h, all);
    if ((rc = r.place_seams()) || (rc = r.collect_work()))
        return rc;
    if (r.works.empty()) {
        for (auto &c : h->ch) {
            if (c.has_span())
                return fail(h, PSDC_ERR_DEVICE, "internal: zero-copy span left unconsumed");
            c.submitted = false;
        }
        return launch_deferred(h, r.seams); // (no work: whatever epilogue is pending goes out now -- read-outs end here)
    }
    *did_work = true;
    if ((rc = r.size_next_stages()))
        return rc;
    r.make_jobs();
    r.share_workgroups();
    // (the bookkeeping comes BEFORE the launches since round 5: this round's tail carries may ride in its fused launch; nothing the
    // launches use -- job tables with absolute pointers, made above -- is touched by it)
    if ((rc = r.place_partials()) || (rc = r.book()))
        return rc;
    return r.run_launches();
}

// one pipeline round (ingest path)
int advance(psdc_handle *h)
{
    bool did = false;
    return advance_round(h, &did, false);
}

// rounds until the pipeline is idle (read-out path)
int drain(psdc_handle *h)
{
    if (h->idle)
        return PSDC_OK;
    for (int guard = 0; guard < 64; ++guard) {
        bool did = false;
        int rc = advance_round(h, &did, true);
        if (rc)
            return rc;
        if (!did) {
            h->idle = true;
            return PSDC_OK;
        }
    }
    return fail(h, PSDC_ERR_DEVICE, "internal: pipeline did not drain");
}

} // namespace psdrt

