// runtime.cpp -- the host runtime behind include/psdcascade.h: handle lifecycle, per-stream device buffers, pinned staging and
// uploads, the feed calls (psdc_process, psdc_process_device) and the flush / sync points.  See host_runtime.h for the parts.
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

using namespace psdrt;

namespace psdrt {

thread_local std::string g_last_error;

int fail(psdc_handle *h, int code, const std::string &msg)
{
    if (h)
        h->err = msg;
    g_last_error = msg;
    return code;
}

} // namespace psdrt

namespace {

// Host samples reach the device through pinned staging buffers.  One core copies ~34 GB/s into
// pinned memory while the link takes ~55 GB/s, so large copies are split over a few threads
// (PSDC_COPY_THREADS, default 4, 1 = caller only).  The workers are created on first use and
// shared by all handles; a copy that finds them busy is done by its caller alone.
class CopyPool {
public:
    static CopyPool &get()
    {
        static CopyPool p;
        return p;
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        constexpr size_t kMin = (size_t)2 << 20;
        if (bytes < kMin || nthreads_ <= 1 || !busy_.try_lock()) {
            memcpy(dst, src, bytes);
            return;
        }
        start_workers();
        const size_t nw = th_.size();
        if (nw == 0) {
            busy_.unlock();
            memcpy(dst, src, bytes);
            return;
        }
        const size_t part = ((bytes / (nw + 1)) + 4095) & ~(size_t)4095;
        char *d = static_cast<char *>(dst);
        const char *sp = static_cast<const char *>(src);
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t i = 0; i < nw; ++i) {
                const size_t o = std::min(bytes, part * (i + 1));
                parts_[i] = {d + o, sp + o, std::min(part, bytes - o)};
            }
            pending_ = (int)nw;
            ++gen_;
        }
        cv_.notify_all();
        memcpy(d, sp, std::min(part, bytes));
        {
            std::unique_lock<std::mutex> lk(m_);
            done_cv_.wait(lk, [&] { return pending_ == 0; });
        }
        busy_.unlock();
    }

private:
    struct Part {
        char *d;
        const char *s;
        size_t n;
    };
    CopyPool()
    {
        int n = 4;
        if (const char *e = getenv("PSDC_COPY_THREADS"))
            n = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0)
            n = std::min(n, hw);
        nthreads_ = std::max(1, std::min(n, 16));
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_)
            t.join();
    }
    void start_workers()
    {
        if (started_)
            return;
        started_ = true;
        parts_.resize((size_t)nthreads_ - 1);
        try {
            for (int i = 0; i + 1 < nthreads_; ++i)
                th_.emplace_back([this, i] { worker((size_t)i); });
        } catch (...) { // no more threads: the ones that started (possibly none) do the work
        }
        parts_.resize(th_.size());
    }
    void worker(size_t id)
    {
        uint64_t seen = 0;
        for (;;) {
            Part p;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_)
                    return;
                seen = gen_;
                p = id < parts_.size() ? parts_[id] : Part{nullptr, nullptr, 0};
            }
            if (p.n)
                memcpy(p.d, p.s, p.n);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0)
                    done_cv_.notify_one();
            }
        }
    }
    int nthreads_ = 1;
    bool started_ = false, stop_ = false;
    std::vector<std::thread> th_;
    std::vector<Part> parts_;
    std::mutex m_, busy_;
    std::condition_variable cv_, done_cv_;
    uint64_t gen_ = 0;
    int pending_ = 0;
};

} // namespace

namespace psdrt {

bool window_consts(uint32_t n, int kind, WindowConsts *w)
{
    if (kind == PSDC_WINDOW_RECTANGULAR) { // src/psd.rs:24-32
        *w = {1.0f, 1.0f, 0};
        return true;
    }
    if (kind == PSDC_WINDOW_HANN) { // src/psd.rs:49-54
        *w = {1.5f, 0.25f, n / 2};
        return true;
    }
    return false;
}

// the weights of Window::hann() / Window::rectangular() exactly as the reference builds them (src/psd.rs:24-32, :42-55)
void window_weights(uint32_t n, int kind, float *win)
{
    if (kind == PSDC_WINDOW_HANN) {
        const float df = 3.14159265358979323846f / (float)n; // core::f32::consts::PI / N as f32  :44
        for (uint32_t i = 0; i < n; ++i) {
            const float s = sinf(df * (float)i); // (df * i as f32).sin().powi(2)  :47
            win[i] = s * s;
        }
    } else {
        for (uint32_t i = 0; i < n; ++i)
            win[i] = 1.0f;
    }
}

// a caller-built Window<N>: which of the library's kinds is it?
int classify_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap)
{
    for (int kind : {PSDC_WINDOW_HANN, PSDC_WINDOW_RECTANGULAR}) {
        WindowConsts wc{};
        window_consts(n, kind, &wc);
        if (wc.power != power || wc.nenbw != nenbw || (size_t)wc.overlap != overlap)
            continue;
        std::vector<float> ref(n);
        window_weights(n, kind, ref.data());
        if (memcmp(ref.data(), win, sizeof(float) * n) == 0)
            return kind;
    }
    return PSDC_WINDOW_CUSTOM;
}

// PSDC_DEVICE_DEFAULT -> the index in $PSDC_DEVICE (0 when unset or unparsable)
int resolve_device(int device)
{
    if (device != PSDC_DEVICE_DEFAULT)
        return device;
    const char *e = getenv("PSDC_DEVICE");
    if (!e || !*e)
        return 0;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end && *end == 0 && v >= 0 && v < 1024) ? (int)v : -2; // -2: rejected as out of range below
}

// powers of two 16 ... 16384 (every kernel), or any other size 16 < n <= 8192 (rustfft plans any length, src/psd.rs:418): those
// run the generic kernels with the DFT in chirp-z form (kernels.hip welch_bluestein_kernel)
// ... and the powers of two 32768 ... 131072 through a global-memory FFT (bigfft.hip: slow, but every size the reference's own
// stack frames let it run)
bool valid_n(uint32_t n) { return n >= 16 && n <= (uint32_t)BIGFFT_MAX_N && ((n & (n - 1)) == 0 || bluestein_size((int)n) != 0); }

// PsdStage::gain (src/psd.rs:279-283): (N/2 * count) as f32, then two f32 multiplies.  The
// reference forms the product in u32, which overflows (panic in debug builds, wrap-around in
// release) once count > 2^32 / (N/2): 8.4 M segments at N = 1024 -- hours for the CPU path,
// about ten seconds of continuous plain-sum ingest here.  The product is widened; below the
// overflow the value is bit-identical to the reference's.
float stage_gain(uint32_t n, uint64_t count, float nenbw, float power)
{
    const uint64_t m = (uint64_t)(n / 2u) * count;
    return (float)m * nenbw * power;
}

uint32_t cur_stage_avg(const psdc_handle *h, size_t i) { return stage_avg(h->avg_limit, h->avg_count, (unsigned)i); }

// lowest absolute index a stage must keep for its next batch: the start of the
// next segment and the decimator history
uint64_t keep_from(const Geometry &g, const StageState &s)
{
    if (s.sink)
        return s.sink_pos;
    if (s.segs == 0)
        return 0;
    // (overlap 0: the fused single-segment runs read half a segment in front of their first segment -- for the decimator's
    // history registers --, so that much is carried too; it only matters for n / 2 > 288)
    const uint64_t back = std::max<uint64_t>(std::max<uint64_t>(g.overlap, HBF_HALO), g.overlap == 0 ? g.n / 2 : 0);
    return s.dec > back ? s.dec - back : 0;
}

int free_stage(psdc_handle *h, StageState &s)
{
    for (int i = 0; i < 2; ++i) {
        if (s.buf.p[i] && !s.buf.pooled)
            HIPCHK(h, hipFree(s.buf.p[i]));
        s.buf.p[i] = nullptr;
    }
    s.buf.pooled = false;
    s.buf.cap = 0;
    s.spectrum = nullptr; // a slot of the handle's slab
    return PSDC_OK;
}

int add_stage(psdc_handle *h, Channel &c)
{
    if (c.st.size() >= MAX_STAGES)
        return fail(h, PSDC_ERR_ARG, "more than 16 cascade stages");
    StageState s;
    const size_t ci = (size_t)(&c - h->ch.data());
    s.spectrum = h->d_spectra + (ci * MAX_STAGES + c.st.size()) * h->n;
    HIPCHK(h, hipMemsetAsync(s.spectrum, 0, sizeof(float) * h->n, h->stream));
    // every stream starts in the pre-allocated pool (enough for a deep stage's trickle); a stream
    // that needs more moves to its own allocation in ensure_room
    float *slot = h->d_pool + ((ci * MAX_STAGES + c.st.size()) * 2) * h->pool_cap;
    s.buf.p[0] = slot;
    s.buf.p[1] = slot + h->pool_cap;
    s.buf.cap = h->pool_cap;
    s.buf.pooled = true;
    s.sink = c.st.size() >= h->stage_limit;
    c.st.push_back(s);
    return PSDC_OK;
}

// index of a frame span in the round's pool (jobs name it by that index until launch_* maps it into the launch's table)
int pool_fspan(psdc_handle *h, const FrameSpan &fs)
{
    for (size_t i = 0; i < h->fs_pool.size(); ++i)
        if (h->fs_pool[i].frames == fs.frames && h->fs_pool[i].bytes == fs.bytes && h->fs_pool[i].frame_size == fs.frame_size &&
            h->fs_pool[i].batches == fs.batches)
            return (int)i;
    h->fs_pool.push_back(fs);
    return (int)h->fs_pool.size() - 1;
}

// copy job: `count` samples of a zero-copy span from absolute stream index `from` to dst (decoded on the way when the
// span is a run of frames)
TailJob span_copy(psdc_handle *h, const DeviceSpan &sp, uint64_t from, float *dst, size_t count)
{
    TailJob t{};
    t.dst = dst;
    t.count = (int)count;
    if (sp.framed()) {
        t.src = nullptr;
        t.fspan = pool_fspan(h, sp.fs);
        t.fch = sp.fch;
        t.s_off = (unsigned)(from - sp.first);
    } else {
        t.src = sp.d_x + (from - sp.first);
    }
    return t;
}


// The epilogue of a round -- fold its partials (RedJob), carry its stream tails (TailJob) -- is
// not launched when the round ends: the next round starts with a copy launch of its own (the
// zero-copy seams), and one launch does both.  Nothing on the device reads what the epilogue
// writes before that point; host-visible state never waits for it (read-outs drain first).
void split_copy_jobs(const std::vector<TailJob> &in, std::vector<TailJob> &out)
{
    constexpr int kPiece = 16384;
    for (const TailJob &t : in)
        for (int o = 0; o < t.count; o += kPiece) {
            TailJob q = t;
            q.src = t.src ? t.src + o : nullptr;
            q.dst = t.dst + o;
            q.count = std::min(kPiece, t.count - o);
            q.s_off = t.s_off + (unsigned)o;
            out.push_back(q);
        }
}

int launch_deferred(psdc_handle *h, const std::vector<TailJob> &extra)
{
    // a copy job is one workgroup of post_kernel: long tails (a stage that collects a batch keeps up to
    // PSDC_OPT_MIN_PAIRS pairs pending) are cut into pieces so that the launch does not wait on one workgroup
    std::vector<TailJob> tails;
    tails.reserve(h->pend_tail.size() + extra.size());
    split_copy_jobs(h->pend_tail, tails);
    split_copy_jobs(extra, tails);
    // $PSDC_DBG_SKIP_POST (timing only, WRONG results): the epilogue and the seam copies are dropped, not launched -- the upper bound of
    // what folding post_kernel's work into the fused launch can buy (one launch per round)
    static const bool skip_post = getenv("PSDC_DBG_SKIP_POST") != nullptr;
    if (skip_post) {
        h->pend_red.clear();
        h->pend_tail.clear();
        return PSDC_OK;
    }
    const size_t nr = h->pend_red.size(), nt = tails.size();
    for (size_t ri = 0, ti = 0; ri < nr || ti < nt;) {
        RedBatch rb{};
        rb.n = (int)h->n;
        for (; ri < nr && rb.njobs < MAX_JOBS; ++ri)
            rb.jobs[rb.njobs++] = h->pend_red[ri];
        TailBatch tb{};
        FspanMap fm(tb.fspans);
        for (; ti < nt && tb.njobs < MAX_JOBS; ++ti) {
            TailJob q = tails[ti];
            if (q.fspan >= 0 && (q.fspan = fm.map(h, q.fspan)) < 0) {
                if (tb.njobs == 0)
                    return fail(h, PSDC_ERR_DEVICE, "internal: frame span table");
                break; // this launch's table is full: the job opens the next one
            }
            tb.jobs[tb.njobs++] = q;
        }
        HIPCHK(h, launch_post(rb, tb, h->stream));
    }
    h->pend_red.clear();
    h->pend_tail.clear();
    return PSDC_OK;
}

// compute-stream work enqueued from here on sees every upload enqueued so far
int wait_uploads(psdc_handle *h)
{
    if (h->upload_pending) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_upload, 0));
        h->upload_pending = false;
    }
    return PSDC_OK;
}
// before an upload is enqueued: the copy stream waits for the latest round's post launch
int order_upload(psdc_handle *h)
{
    if (h->post_dirty) { // rounds were enqueued since the last upload: the event goes behind everything the compute stream holds
        HIPCHK(h, hipEventRecord(h->ev_post, h->stream));
        h->post_marked = true;
        h->post_dirty = false;
    }
    if (h->post_marked)
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_post, 0));
    return PSDC_OK;
}
int mark_upload(psdc_handle *h)
{
    HIPCHK(h, hipEventRecord(h->ev_upload, h->copy_stream));
    h->upload_pending = true;
    return PSDC_OK;
}

// make room for absolute indices [base, new_end) in the current buffer
int ensure_room(psdc_handle *h, StageState &s, uint64_t new_end)
{
    const size_t need = (size_t)(new_end - s.buf.base);
    if (need <= s.buf.cap)
        return PSDC_OK;
    {
        int rc = launch_deferred(h, {}); // a carried tail may still be on its way into this buffer
        if (rc)
            return rc;
        rc = wait_uploads(h); // ... or an upload
        if (rc)
            return rc;
    }
    const size_t min_cap = (size_t)4 * (h->n + HBF_HALO) + 64;
    size_t cap = std::max(need + need / 2, min_cap);
    float *np[2] = {nullptr, nullptr};
    HIPCHK(h, hipMalloc(&np[0], sizeof(float) * cap));
    HIPCHK(h, hipMalloc(&np[1], sizeof(float) * cap));
    const size_t have = (size_t)(s.buf.end - s.buf.base);
    if (have && s.buf.p[s.buf.cur])
        HIPCHK(h, hipMemcpyAsync(np[0], s.buf.p[s.buf.cur], sizeof(float) * have,
                                 hipMemcpyDeviceToDevice, h->stream));
    for (int i = 0; i < 2; ++i)
        if (s.buf.p[i] && !s.buf.pooled)
            h->retired.push_back(s.buf.p[i]); // freed once the stream is idle (release_retired)
    s.buf.p[0] = np[0];
    s.buf.p[1] = np[1];
    s.buf.pooled = false;
    s.buf.cur = 0;
    s.buf.cap = cap;
    return PSDC_OK;
}

// both ping-pong buffers can hold `need` floats (content of the current one is kept); when they have
// to grow they grow to `grow_to` (>= need) at once -- growing never waits for the device: the buffers it replaces
// are retired and released at the next sync or read-out (release_retired)
int ensure_cap(psdc_handle *h, StageState &s, size_t need, size_t grow_to)
{
    if (need <= s.buf.cap)
        return PSDC_OK;
    return ensure_room(h, s, s.buf.base + std::max(need, grow_to));
}

int ensure_partial(psdc_handle *h, size_t floats)
{
    if (floats <= h->partial_cap)
        return PSDC_OK;
    if (h->d_partial) {
        int rc = launch_deferred(h, {}); // the last round's partials are still to be folded
        if (rc)
            return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(h->d_partial));
        h->d_partial = nullptr;
    }
    const size_t cap = floats + floats / 2;
    HIPCHK(h, hipMalloc(&h->d_partial, sizeof(float) * 2 * cap)); // two slabs (host_runtime.h)
    h->partial_cap = cap;
    return PSDC_OK;
}

int collect_profile(psdc_handle *h)
{
    for (auto &e : h->prof_pending) {
        HIPCHK(h, hipEventSynchronize(e.b));
        float ms = 0.0f;
        HIPCHK(h, hipEventElapsedTime(&ms, e.a, e.b));
        h->prof.kernel_ms += (double)ms;
        HIPCHK(h, hipEventDestroy(e.a));
        HIPCHK(h, hipEventDestroy(e.b));
    }
    h->prof_pending.clear();
    return PSDC_OK;
}

// In-place spans of a channel that may share a round.  A round costs ~20 us of launch boundaries whatever it holds (the post
// launch and two dependent dispatches): 8 spans of 2^26 samples are 0.7 ms of kernel, 8 of 2^22 are 50 us -- so a single channel
// holds sixteen (2^26 a call: +1.0-1.7 %, 2^24: +3 %, 2^22: +21 %; eight channels x 2^24 measured -2 % with sixteen and stay at eight),
// and more of shorter spans.  An explicit PSDC_OPT_COALESCE is taken as given.
//
// WHICH spans share a round is a function of the call sequence alone (round 5): a joinable span is held until its channel holds
// hold_max samples or a call arrives that cannot join (one span more than `coalesce_limit`, a host-fed or short span, a settings change, a read-out, psdc_flush /
// psdc_sync / psdc_record_consumed).  Through round 4 a held span also went out as soon as hipStreamQuery saw the device idle: lower
// latency on a trickle feed, but the grouping of the partial sums -- run lengths, the one f32 add per round -- then followed HOST
// TIMING, and the same calls could give spectra that differ in the last bits from run to run (the driver's bench coalesced 6.8 spans
// a launch where the builder's box made 7.9, same code, same calls).  The reference is deterministic to the bit (src/psd.rs:228-233);
// so is the default here.  PSDC_OPT_EAGER = 1 brings the timing rule back for callers who want the first span of a burst on the
// device at once.
uint32_t coalesce_limit(const psdc_handle *h, const Channel &c, size_t len)
{
    if (!h->coalesce_auto)
        return h->coalesce;
    const size_t m = std::max<size_t>(std::max(c.span_max, len), 1);
    if (h->n_channels == 1) {
        // sixteen spans a round (2^30 samples of 2^26-sample spans: hold_max), up to 128 spans shorter than 2^24 samples -- rounds
        // of about 2^28 (round 5: spans that do NOT continue each other in memory -- those merge -- at 2^22 / 2^20 / 2^18 / 2^16 samples a
        // call: tests/host/devcall_probe "scattered")
        return (uint32_t)std::min<size_t>(MAX_COALESCE, std::max<size_t>(16, ((size_t)1 << 28) / m));
    }
    // several channels: eight a channel, more of short spans -- rounds of about 2^28 samples over all channels, as many spans as keep
    // the round ONE launch (two jobs a span and about twelve deep stages a channel within MAX_JOBS: 34 a channel for four channels, 14
    // for eight; planner.cpp share_workgroups).  Four channels fed in 2^20-sample spans: 502 GS/s at eight, 587 at sixteen
    const size_t fit = ((size_t)MAX_JOBS - std::min<size_t>(MAX_JOBS / 2, (size_t)12 * h->n_channels)) / ((size_t)2 * h->n_channels);
    const size_t most = std::min<size_t>(MAX_COALESCE, std::max<size_t>(h->coalesce, fit));
    return (uint32_t)std::min<size_t>(most, std::max<size_t>(h->coalesce, ((size_t)1 << 28) / (m * h->n_channels)));
}

bool holds_short_span(const psdc_handle *h, const Channel &c)
{
    return c.spans.size() == 1 && !c.spans[0].framed() && c.spans[0].len < (size_t)4 * (h->n + HBF_HALO);
}

// A channel's ONLY held span is still shorter than what the planner reads in place (4 (n + 288) samples): it becomes a copy behind
// the stream buffer's content, like host-fed samples.  (Only a first span can be that short: psdc_process_device holds a short span
// only when nothing is held in front of it.)
int settle_short_span(psdc_handle *h, Channel &c)
{
    if (!holds_short_span(h, c))
        return PSDC_OK;
    const DeviceSpan sp = c.spans[0];
    StageState &s0 = c.st[0];
    c.spans.clear();
    int rc = ensure_room(h, s0, s0.total);
    if (rc)
        return rc;
    HIPCHK(h, hipMemcpyAsync(s0.buf.p[s0.buf.cur] + (sp.first - s0.buf.base), sp.d_x, sizeof(float) * sp.len, hipMemcpyDeviceToDevice,
                             h->stream));
    s0.buf.end = s0.total;
    c.submitted = true;
    return PSDC_OK;
}

size_t held_samples(const Channel &c)
{
    size_t t = 0;
    for (const DeviceSpan &sp : c.spans)
        t += sp.len;
    return t;
}

// eager handles only: nothing of this handle is executing or queued on the device (~0.1 us, tools/probes/stream_query.cpp)
bool device_idle(psdc_handle *h) { return h->eager && hipStreamQuery(h->stream) == hipSuccess; }

int submit_host(psdc_handle *h, Channel &c)
{
    if (c.fill == 0)
        return PSDC_OK;
    h->idle = false;
    if (c.st.empty()) {
        int rc = add_stage(h, c);
        if (rc)
            return rc;
    }
    StageState &s0 = c.st[0];
    int rc = ensure_room(h, s0, s0.total + c.fill);
    if (rc)
        return rc;
    const int b = c.cur_stage;
    rc = order_upload(h);
    if (rc)
        return rc;
    HIPCHK(h, hipMemcpyAsync(s0.buf.p[s0.buf.cur] + (s0.total - s0.buf.base), c.stage_host[b],
                             sizeof(float) * c.fill, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(c.stage_ev[b], h->copy_stream));
    {
        int rc2 = mark_upload(h);
        if (rc2)
            return rc2;
    }
    c.ev_pending[b] = true;
    s0.total += c.fill;
    s0.buf.end = s0.total;
    c.fill = 0;
    c.submitted = true;
    c.cur_stage = b ^ 1;
    if (c.ev_pending[c.cur_stage]) {
        HIPCHK(h, hipEventSynchronize(c.stage_ev[c.cur_stage]));
        c.ev_pending[c.cur_stage] = false;
    }
    return PSDC_OK;
}

int ensure_staging(psdc_handle *h, Channel &c)
{
    if (c.stage_host[0])
        return PSDC_OK;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&c.stage_host[i]), sizeof(float) * h->quantum,
                                hipHostMallocDefault));
        HIPCHK(h, hipEventCreateWithFlags(&c.stage_ev[i], hipEventDisableTiming));
    }
    return PSDC_OK;
}

int free_staging(psdc_handle *h, Channel &c)
{
    for (int i = 0; i < 2; ++i) {
        if (c.stage_host[i]) {
            HIPCHK(h, hipHostFree(c.stage_host[i]));
            c.stage_host[i] = nullptr;
        }
        if (c.stage_ev[i]) {
            HIPCHK(h, hipEventDestroy(c.stage_ev[i]));
            c.stage_ev[i] = nullptr;
        }
        c.ev_pending[i] = false;
    }
    return PSDC_OK;
}

int flush_all(psdc_handle *h)
{
    for (auto &c : h->ch) {
        int rc = submit_host(h, c);
        if (rc)
            return rc;
    }
    return drain(h);
}

// the stream is idle: nothing can still read the buffers that growth replaced
int release_retired(psdc_handle *h)
{
    while (!h->retired.empty()) { // popped before it is freed: a failing hipFree never leaves a freed pointer listed
        float *p = h->retired.back();
        h->retired.pop_back();
        HIPCHK(h, hipFree(p));
    }
    return PSDC_OK;
}

int flush_sync(psdc_handle *h)
{
    int rc = flush_all(h);
    if (rc)
        return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return release_retired(h);
}

// device -> caller memory through the handle's pinned buffer (pageable D2H copies take a slow,
// lazily initialised staging path in the runtime)
int read_back(psdc_handle *h, float *dst, const float *d_src, size_t count)
{
    const size_t chunk = (size_t)MAX_STAGES * h->n;
    while (count > 0) {
        const size_t m = std::min(count, chunk);
        HIPCHK(h, launch_copy_out(h->h_read, d_src, m, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        memcpy(dst, h->h_read, sizeof(float) * m);
        dst += m;
        d_src += m;
        count -= m;
    }
    return PSDC_OK;
}

int check_channel(psdc_handle *h, uint32_t channel)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    if (channel >= h->n_channels)
        return fail(h, PSDC_ERR_ARG, "channel out of range");
    return PSDC_OK;
}

} // namespace psdrt

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" {

int psdc_abi_version(void) { return PSDC_ABI_VERSION; }

const char *psdc_last_error(const psdc_handle *h) { return h ? h->err.c_str() : g_last_error.c_str(); }

#ifdef PSDK_SEGV_TRACE // debugging aid (tools/build_variants.sh ... "-g -DPSDK_SEGV_TRACE"), never in the shipped build
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void psdk_segv(int sig)
{
    void *bt[64];
    const int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}
struct PsdkSegvInstall {
    PsdkSegvInstall() { signal(SIGSEGV, psdk_segv); }
} g_psdk_segv_install;
#endif

} // extern "C"

namespace psdrt {

void pinned_copy(void *dst, const void *src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }

// every constructor ends here: window_kind HANN / RECTANGULAR (win == nullptr: the library's table) or CUSTOM
// (win = the caller's n weights, wc = its constants)
psdc_handle *create_impl(uint32_t n, int window_kind, const float *win_in, WindowConsts wc, uint32_t n_channels,
                         int device)
{
    device = resolve_device(device);
    if (!valid_n(n) || !welch_supported((int)n)) {
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: n must be a power of two in [16, 131072] or any size in [16, 8192]");
        return nullptr;
    }
    if (window_kind != PSDC_WINDOW_CUSTOM && !window_consts(n, window_kind, &wc)) {
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: unknown window kind");
        return nullptr;
    }
    if (wc.overlap >= n || (n - wc.overlap) % 8 != 0) { // src/psd.rs:246-247 (overlap >= n: `N - overlap` underflows / no progress)
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: overlap must be below n and (n - overlap) a multiple of 8 (src/psd.rs:247)");
        return nullptr;
    }
    if (n_channels == 0 || n_channels > 4096) {
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: n_channels out of range");
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fail(nullptr, PSDC_ERR_DEVICE,
             std::string("psdc_create: no HIP device (there is no CPU fallback): ") +
                 (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: device index out of range");
        return nullptr;
    }
    psdc_handle *h = new (std::nothrow) psdc_handle();
    if (!h) {
        fail(nullptr, PSDC_ERR_NOMEM, "psdc_create: out of memory");
        return nullptr;
    }
    h->n = n;
    h->window_kind = window_kind;
    h->geo.n = n;
    h->geo.overlap = wc.overlap;
    h->geo.hop = n - wc.overlap;
    h->geo.drain = (uint32_t)HBF_DRAIN;
    h->nenbw = wc.nenbw;
    h->power = wc.power;
    h->n_channels = n_channels;
    h->device = device;
    h->ch.resize(n_channels);
    h->min_pairs = fused_supported((int)n) ? 32u * (uint32_t)std::max(1, fused_pairs_per_block((int)n, 1)) : 0u;

    // window table exactly as the reference builds it (src/psd.rs:44-48) or as the caller did, twiddles in f64
    std::vector<float> win(n);
    if (window_kind == PSDC_WINDOW_CUSTOM)
        memcpy(win.data(), win_in, sizeof(float) * n);
    else
        window_weights(n, window_kind, win.data());
    h->win_host = win;
    // twiddles of the generic kernels' transform: length n, or -- n not a power of two -- the chirp-z length M with its tables
    const uint32_t m_fft = bluestein_size((int)n) ? (uint32_t)bluestein_size((int)n) : n;
    std::vector<cf> tw(m_fft);
    for (uint32_t i = 0; i < m_fft; ++i) {
        const double a = -2.0 * M_PI * (double)i / (double)m_fft;
        tw[i] = {(float)cos(a), (float)sin(a)};
    }
    std::vector<cf> chirp, bhat;
    if (m_fft != n) {
        // c[j] = exp(i pi j^2 / n): j^2 mod 2n in integers keeps the phase exact; B = FFT_M(c wrapped) by an f64 radix-2 FFT
        std::vector<std::complex<double>> c(n), b(m_fft, 0.0);
        for (uint32_t j = 0; j < n; ++j) {
            const uint64_t r = ((uint64_t)j * j) % (2ull * n);
            c[j] = std::polar(1.0, M_PI * (double)r / (double)n);
        }
        b[0] = c[0];
        for (uint32_t j = 1; j < n; ++j)
            b[j] = b[m_fft - j] = c[j];
        int bits = 0;
        while ((1u << bits) < m_fft)
            ++bits;
        std::vector<std::complex<double>> y(m_fft);
        for (uint32_t i = 0; i < m_fft; ++i) {
            uint32_t r = 0;
            for (int k = 0; k < bits; ++k)
                if (i & (1u << k))
                    r |= 1u << (bits - 1 - k);
            y[r] = b[i];
        }
        for (uint32_t len = 2; len <= m_fft; len <<= 1)
            for (uint32_t b0 = 0; b0 < m_fft; b0 += len)
                for (uint32_t k = 0; k < len / 2; ++k) {
                    const std::complex<double> w = std::polar(1.0, -2.0 * M_PI * (double)k / (double)len);
                    const std::complex<double> u = y[b0 + k], t = w * y[b0 + k + len / 2];
                    y[b0 + k] = u + t;
                    y[b0 + k + len / 2] = u - t;
                }
        chirp.resize(n);
        bhat.resize(m_fft);
        for (uint32_t j = 0; j < n; ++j)
            chirp[j] = {(float)c[j].real(), (float)c[j].imag()};
        for (uint32_t j = 0; j < m_fft; ++j)
            bhat[j] = {(float)y[j].real(), (float)y[j].imag()};
    }
    auto dev_fail = [&](hipError_t err, const char *what) -> psdc_handle * {
        fail(nullptr, PSDC_ERR_DEVICE, std::string("psdc_create: ") + what + ": " + hipGetErrorString(err));
        psdc_destroy(h);
        return nullptr;
    };
    DevScope dev_scope_(device);
    if ((e = dev_scope_.err) != hipSuccess)
        return dev_fail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_post, hipEventDisableTiming)) != hipSuccess)
        return dev_fail(e, "hipStreamCreate");
    if ((e = hipMalloc(&h->d_win, sizeof(float) * n)) != hipSuccess)
        return dev_fail(e, "hipMalloc(win)");
    if ((e = hipMalloc(&h->d_tw, sizeof(cf) * m_fft)) != hipSuccess)
        return dev_fail(e, "hipMalloc(tw)");
    if (!chirp.empty() && ((e = hipMalloc(&h->d_chirp, sizeof(cf) * chirp.size())) != hipSuccess ||
                           (e = hipMalloc(&h->d_bhat, sizeof(cf) * bhat.size())) != hipSuccess ||
                           (e = hipMemcpy(h->d_chirp, chirp.data(), sizeof(cf) * chirp.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                           (e = hipMemcpy(h->d_bhat, bhat.data(), sizeof(cf) * bhat.size(), hipMemcpyHostToDevice)) != hipSuccess))
        return dev_fail(e, "chirp-z tables");
    if ((e = hipMalloc(&h->d_spectra, sizeof(float) * (size_t)n_channels * MAX_STAGES * n)) != hipSuccess)
        return dev_fail(e, "hipMalloc(spectra)");
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->h_read), sizeof(float) * (size_t)MAX_STAGES * n,
                           hipHostMallocDefault)) != hipSuccess)
        return dev_fail(e, "hipHostMalloc(read-out)");
    // one full-size read through the pinned buffer now (read-outs copy with a kernel that writes
    // the pinned host buffer directly, see launch_copy_out): first-use costs do not belong in the
    // first psd() of a live stream
    if ((e = hipMemsetAsync(h->d_spectra, 0, sizeof(float) * (size_t)MAX_STAGES * n, h->stream)) != hipSuccess ||
        (e = launch_copy_out(h->h_read, h->d_spectra, (size_t)MAX_STAGES * n, h->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(h->stream)) != hipSuccess)
        return dev_fail(e, "read-out warm-up");
    h->pool_cap = (size_t)4 * (n + HBF_HALO) + 64;
    if ((e = hipMalloc(&h->d_pool, sizeof(float) * (size_t)n_channels * MAX_STAGES * 2 * h->pool_cap)) != hipSuccess)
        return dev_fail(e, "hipMalloc(stream pool)");
    // partial slab for a full round (grows only if many channels need more)
    h->partial_cap = bigfft_size((int)n) ? (size_t)MAX_JOBS * n // (one row per job at these sizes)
                                         : (size_t)(fused_max_blocks((int)n) + WELCH_MAX_BLOCKS + 4 * MAX_JOBS) * n;
    if (bigfft_size((int)n)) {
        h->bigfft_elems = BIGFFT_SCRATCH_ELEMS;
        if (const char *e = getenv("PSDC_DBG_BIGFFT_CHUNK"))
            h->bigfft_chunk_limit = atoi(e);
        if ((e = hipMalloc(&h->d_bigfft, sizeof(cf) * h->bigfft_elems)) != hipSuccess)
            return dev_fail(e, "hipMalloc(big FFT frames)");
    }
    if ((e = hipMalloc(&h->d_partial, sizeof(float) * 2 * h->partial_cap)) != hipSuccess)
        return dev_fail(e, "hipMalloc(partials)");
    {
        std::vector<cf> t0, ta;
        fused_big_tables((int)n, t0, ta);
        if (!t0.empty()) {
            if ((e = hipMalloc(&h->d_tw0g, sizeof(cf) * t0.size())) != hipSuccess ||
                (e = hipMalloc(&h->d_twag, sizeof(cf) * ta.size())) != hipSuccess ||
                (e = hipMemcpy(h->d_tw0g, t0.data(), sizeof(cf) * t0.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_twag, ta.data(), sizeof(cf) * ta.size(), hipMemcpyHostToDevice)) != hipSuccess)
                return dev_fail(e, "twiddle tables");
        }
    }
    {
        std::vector<cf> t3;
        fused_big3_table((int)n, t3);
        if (!t3.empty() && ((e = hipMalloc(&h->d_tw3g, sizeof(cf) * t3.size())) != hipSuccess ||
                            (e = hipMemcpy(h->d_tw3g, t3.data(), sizeof(cf) * t3.size(), hipMemcpyHostToDevice)) != hipSuccess))
            return dev_fail(e, "twiddle seeds");
    }
    if ((e = hipMemcpy(h->d_win, win.data(), sizeof(float) * n, hipMemcpyHostToDevice)) != hipSuccess)
        return dev_fail(e, "hipMemcpy(win)");
    if ((e = hipMemcpy(h->d_tw, tw.data(), sizeof(cf) * m_fft, hipMemcpyHostToDevice)) != hipSuccess)
        return dev_fail(e, "hipMemcpy(tw)");
    return h;
}

// a caller's Window<N> -> (kind, constants) or an error message
const char *check_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap, int *kind, WindowConsts *wc)
{
    if (!valid_n(n))
        return "n must be a power of two in [16, 131072] or any size in [16, 8192]";
    if (!win)
        return "null window";
    if (overlap >= n || (n - overlap) % 8 != 0)
        return "overlap must be below n and (n - overlap) a multiple of 8 (src/psd.rs:247)";
    if (!std::isfinite(power) || !std::isfinite(nenbw))
        return "window power / nenbw not finite";
    *kind = classify_window(n, win, power, nenbw, overlap);
    *wc = {nenbw, power, (uint32_t)overlap};
    return nullptr;
}

} // namespace psdrt

extern "C" {

psdc_handle *psdc_create(uint32_t n, int window_kind, uint32_t n_channels, int device)
{
    if (window_kind != PSDC_WINDOW_HANN && window_kind != PSDC_WINDOW_RECTANGULAR) {
        fail(nullptr, PSDC_ERR_ARG, "psdc_create: unknown window kind (a caller-built Window goes through psdc_create_window)");
        return nullptr;
    }
    return create_impl(n, window_kind, nullptr, WindowConsts{}, n_channels, device);
}

psdc_handle *psdc_create_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap,
                                uint32_t n_channels, int device)
{
    int kind = 0;
    WindowConsts wc{};
    if (const char *msg = check_window(n, win, power, nenbw, overlap, &kind, &wc)) {
        fail(nullptr, PSDC_ERR_ARG, std::string("psdc_create_window: ") + msg);
        return nullptr;
    }
    return create_impl(n, kind, kind == PSDC_WINDOW_CUSTOM ? win : nullptr, wc, n_channels, device);
}

int psdc_window_get(const psdc_handle *h, int *kind, float *power, float *nenbw, size_t *overlap, float *win)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    if (kind)
        *kind = h->window_kind;
    if (power)
        *power = h->power;
    if (nenbw)
        *nenbw = h->nenbw;
    if (overlap)
        *overlap = h->geo.overlap;
    if (win)
        memcpy(win, h->win_host.data(), sizeof(float) * h->n);
    return PSDC_OK;
}

int psdc_window_table(uint32_t n, int window_kind, float *win, float *power, float *nenbw, size_t *overlap)
{
    WindowConsts wc{};
    if (n < 2 || !window_consts(n, window_kind, &wc))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_window_table: bad arguments");
    if (win)
        window_weights(n, window_kind, win);
    if (power)
        *power = wc.power;
    if (nenbw)
        *nenbw = wc.nenbw;
    if (overlap)
        *overlap = wc.overlap;
    return PSDC_OK;
}

void psdc_destroy(psdc_handle *h)
{
    if (!h)
        return;
    DevScope dev_scope_(h->device);
    if (h->copy_stream)
        (void)hipStreamSynchronize(h->copy_stream);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    (void)collect_profile(h);
    for (auto &c : h->ch) {
        for (auto &s : c.st)
            (void)free_stage(h, s);
        (void)free_staging(h, c);
    }
    (void)release_retired(h);
    if (h->d_partial)
        (void)hipFree(h->d_partial);
    if (h->d_spectra)
        (void)hipFree(h->d_spectra);
    if (h->d_pool)
        (void)hipFree(h->d_pool);
    if (h->h_read)
        (void)hipHostFree(h->h_read);
    if (h->scan_stream) {
        (void)hipStreamSynchronize(h->scan_stream);
        (void)hipStreamDestroy(h->scan_stream);
    }
    if (h->d_scan)
        (void)hipFree(h->d_scan);
    if (h->h_scan)
        (void)hipHostFree(h->h_scan);
    if (h->hdr_stream) {
        (void)hipStreamSynchronize(h->hdr_stream);
        (void)hipStreamDestroy(h->hdr_stream);
    }
    if (h->h_hdr)
        (void)hipHostFree(h->h_hdr);
    for (int i = 0; i < 2; ++i) {
        if (h->d_frames[i])
            (void)hipFree(h->d_frames[i]);
        if (h->h_frames[i])
            (void)hipHostFree(h->h_frames[i]);
        if (h->frames_ev[i])
            (void)hipEventDestroy(h->frames_ev[i]);
        if (h->frames_dec_ev[i])
            (void)hipEventDestroy(h->frames_dec_ev[i]);
    }
    if (h->d_win)
        (void)hipFree(h->d_win);
    if (h->d_tw)
        (void)hipFree(h->d_tw);
    if (h->d_tw0g)
        (void)hipFree(h->d_tw0g);
    if (h->d_twag)
        (void)hipFree(h->d_twag);
    if (h->d_tw3g)
        (void)hipFree(h->d_tw3g);
    if (h->d_chirp)
        (void)hipFree(h->d_chirp);
    if (h->d_bhat)
        (void)hipFree(h->d_bhat);
    if (h->d_bigfft)
        (void)hipFree(h->d_bigfft);
    if (h->ev_upload)
        (void)hipEventDestroy(h->ev_upload);
    if (h->ev_post)
        (void)hipEventDestroy(h->ev_post);
    if (h->copy_stream)
        (void)hipStreamDestroy(h->copy_stream);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
}

int psdc_reset(psdc_handle *h)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    ON_DEVICE(h, h->device);
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->upload_pending = false;
    {
        int rc = release_retired(h);
        if (rc)
            return rc;
    }
    h->pend_red.clear(); // the state they would update is discarded
    h->pend_tail.clear();
    h->idle = true;
    for (auto &c : h->ch) {
        for (auto &s : c.st) {
            int rc = free_stage(h, s);
            if (rc)
                return rc;
        }
        c.st.clear();
        c.fill = 0;
        c.submitted = false;
        c.spans.clear();
    }
    return PSDC_OK;
}

int psdc_configure(psdc_handle *h, int option, int64_t value)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    ON_DEVICE(h, h->device);
    switch (option) {
    case PSDC_OPT_QUANTUM: {
        if (value < 1 || value > ((int64_t)1 << 30))
            return fail(h, PSDC_ERR_ARG, "quantum out of range");
        int rc = flush_sync(h);
        if (rc)
            return rc;
        for (auto &c : h->ch) {
            rc = free_staging(h, c);
            if (rc)
                return rc;
        }
        h->quantum = (size_t)value;
        return PSDC_OK;
    }
    case PSDC_OPT_PROFILE:
        h->profile = value != 0;
        return PSDC_OK;
    case PSDC_OPT_COALESCE: {
        const int64_t k = value < 0 ? -value : value; // (-k: what "hold even on an idle device" was spelled through round 4; the default now)
        if (k < 1 || k > MAX_COALESCE_OPT)
            return fail(h, PSDC_ERR_ARG, "coalesce out of range (1..16)");
        int rc = flush_all(h);
        if (rc)
            return rc;
        h->coalesce = (uint32_t)k;
        h->coalesce_auto = false;
        return PSDC_OK;
    }
    case PSDC_OPT_MERGE: {
        int rc = flush_all(h);
        if (rc)
            return rc;
        h->merge = value != 0;
        return PSDC_OK;
    }
    case PSDC_OPT_EAGER: {
        int rc = flush_all(h);
        if (rc)
            return rc;
        h->eager = value != 0;
        return PSDC_OK;
    }
    case PSDC_OPT_MIN_PAIRS: {
        if (value < 0 || value > (1 << 20))
            return fail(h, PSDC_ERR_ARG, "min_pairs out of range");
        h->min_pairs = (uint32_t)value;
        return PSDC_OK;
    }
    default:
        return fail(h, PSDC_ERR_ARG, "unknown option");
    }
}

int psdc_set_detrend(psdc_handle *h, int kind)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    if (kind == PSDC_DETREND_LINEAR)
        return fail(h, PSDC_ERR_UNIMPLEMENTED, "Detrend::Linear is unimplemented (src/psd.rs:110)");
    if (kind < 0 || kind > PSDC_DETREND_LINEAR)
        return fail(h, PSDC_ERR_ARG, "unknown detrend kind");
    ON_DEVICE(h, h->device);
    int rc = flush_all(h); // segments completed so far keep the old setting
    if (rc)
        return rc;
    h->detrend = kind;
    return PSDC_OK;
}

int psdc_set_avg(psdc_handle *h, uint32_t limit, uint32_t count)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    ON_DEVICE(h, h->device);
    int rc = flush_all(h);
    if (rc)
        return rc;
    h->avg_limit = limit;
    h->avg_count = count;
    return PSDC_OK;
}

int psdc_process(psdc_handle *h, uint32_t channel, const float *x, size_t len)
{
    // The reference's callers hand over 512 samples (Source::get for Data::Raw, src/source.rs:150-157) or one frame's worth per call
    // (src/bin/psd.rs:172-181): a call that only adds to a staging buffer which stays below its quantum is a bounds check and a
    // memcpy -- no HIP call at all, not even the device query of ON_DEVICE (tests/host/smallcall_probe.cpp measures the boundary
    // at these sizes).  Everything else takes the general path below.
    static const bool no_fast = getenv("PSDC_NO_FASTPATH") != nullptr; // (A/B aid: every call through the general path)
    if (h && channel < h->n_channels && x && len && !no_fast) {
        Channel &cf_ = h->ch[channel];
        if (cf_.stage_host[0] && !cf_.st.empty() && cf_.spans.empty() && len < h->quantum - cf_.fill && len < ((size_t)1 << 19)) { // (>= 2 MiB: the copy threads)
            memcpy(cf_.stage_host[cf_.cur_stage] + cf_.fill, x, sizeof(float) * len);
            cf_.fill += len;
            h->idle = false;
            return PSDC_OK;
        }
    }
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (len == 0)
        return PSDC_OK; // x.chunks() yields nothing: no stage is created (src/psd.rs:459)
    if (!x)
        return fail(h, PSDC_ERR_ARG, "null input");
    Channel &c = h->ch[channel];
    if (c.has_span()) { // keep the stream in order behind a pending zero-copy span
        rc = advance(h);
        if (rc)
            return rc;
    }
    rc = ensure_staging(h, c);
    if (rc)
        return rc;
    if (c.st.empty()) {
        rc = add_stage(h, c);
        if (rc)
            return rc;
    }
    h->idle = false;
    while (len > 0) {
        const size_t take = std::min(len, h->quantum - c.fill);
        CopyPool::get().copy(c.stage_host[c.cur_stage] + c.fill, x, sizeof(float) * take);
        c.fill += take;
        x += take;
        len -= take;
        if (c.fill == h->quantum) {
            if (c.submitted) { // a full round of channels is on the device: run it as one batch
                rc = advance(h);
                if (rc)
                    return rc;
            }
            rc = submit_host(h, c);
            if (rc)
                return rc;
            if (h->n_channels == 1) {
                rc = advance(h);
                if (rc)
                    return rc;
            }
        }
    }
    return PSDC_OK;
}

int psdc_process_device(psdc_handle *h, uint32_t channel, const float *d_x, size_t len)
{
    return psdc_process_device_after(h, channel, d_x, len, nullptr);
}

int psdc_process_device_after(psdc_handle *h, uint32_t channel, const float *d_x, size_t len, void *producer_event)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (len == 0)
        return PSDC_OK;
    if (!d_x)
        return fail(h, PSDC_ERR_ARG, "null input");
    // everything enqueued on the handle's stream from here on runs behind the producer's event; the span is
    // read only by work enqueued later (this call's round or a later, coalesced one)
    if (producer_event)
        HIPCHK(h, hipStreamWaitEvent(h->stream, static_cast<hipEvent_t>(producer_event), 0));
    Channel &c = h->ch[channel];
    // A span that CONTINUES the last held one in memory (a ring buffer being filled, a capture buffer handed over piece by piece)
    // simply extends it: no seam, no job of its own, whatever its length -- a stream fed in 2^16-sample calls from one buffer runs
    // like one fed in 2^26-sample calls (tests/host/devcall_probe.cpp).  Contiguity is a property of the call sequence, so the
    // grouping stays deterministic.  PSDC_OPT_MERGE = 0 turns it off (tests of the multi-span planner).
    if (h->merge && c.has_span() && c.fill == 0 && !c.submitted) {
        DeviceSpan &last = c.spans.back();
        if (!last.framed() && last.d_x + last.len == d_x && last.len + len <= h->span_cap && held_samples(c) + len <= hold_max(h)) {
            last.len += len;
            c.st[0].total += len;
            c.span_max = std::max(c.span_max, last.len);
            h->idle = false;
            if (h->n_channels == 1 && (held_samples(c) >= hold_max(h) || device_idle(h)))
                return advance(h);
            return PSDC_OK;
        }
    }
    const bool in_place = len >= (size_t)4 * (h->n + HBF_HALO);
    // Earlier spans of this channel must go out first -- unless this one can join them: an in-place
    // span behind in-place spans, fewer than PSDC_OPT_COALESCE of them and at most hold_max samples in all (the rule
    // is a function of the calls alone).  An EAGER handle also sends them out when it sees the device idle (the stream is asked
    // at most ONCE per call; a "busy" answer stands for the rest of the call).
    bool flush = c.submitted, known_busy = false;
    if (c.has_span()) {
        if (!in_place || c.fill > 0 || c.spans.size() >= coalesce_limit(h, c, len) || held_samples(c) + len > hold_max(h) ||
            holds_short_span(h, c)) // (a short span is held only while it can still grow: this call does not continue it)
            flush = true;
        else if (device_idle(h))
            flush = true;
        else
            known_busy = true;
    }
    if (flush) {
        rc = advance(h);
        if (rc)
            return rc;
    }
    rc = submit_host(h, c); // host-fed samples staged earlier come first
    if (rc)
        return rc;
    if (c.st.empty()) {
        rc = add_stage(h, c);
        if (rc)
            return rc;
    }
    StageState &s0 = c.st[0];
    h->idle = false;
    if (!in_place && h->merge && !c.has_span() && !c.submitted && c.fill == 0) {
        // A short span with nothing held in front of it is held all the same: the calls that continue it in memory extend it
        // (above), and it is read in place once it is long enough; if its round comes first, advance_round copies it
        // (settle_short_span) -- so a buffer handed over in pieces of a few hundred samples costs a round per ROUND, not per call.
        c.spans.push_back({d_x, s0.total, len});
        s0.total += len;
        if (h->n_channels == 1 && device_idle(h))
            return advance(h);
        return PSDC_OK;
    }
    if (!in_place) {
        // short span: append a copy, like host-fed samples
        rc = ensure_room(h, s0, s0.total + len);
        if (rc)
            return rc;
        HIPCHK(h, hipMemcpyAsync(s0.buf.p[s0.buf.cur] + (s0.total - s0.buf.base), d_x,
                                 sizeof(float) * len, hipMemcpyDeviceToDevice, h->stream));
        s0.total += len;
        s0.buf.end = s0.total;
        c.submitted = true;
    } else {
        c.spans.push_back({d_x, s0.total, len});
        s0.total += len;
        c.span_max = std::max(c.span_max, len);
        if (c.spans.size() > 1)
            c.coalesced_seen = true;
    }
    if (h->n_channels == 1) {
        // held: the next span may share this one's round, or continue this span in memory -- so a round that holds its LAST span goes
        // out when the call arrives that cannot join it (above), not when that span starts: a buffer handed over in pieces would
        // otherwise leave every round with the first piece of its last span (667 against 707 GS/s for 2^24-sample pieces)
        if (c.has_span() && !c.submitted && held_samples(c) < hold_max(h) && coalesce_limit(h, c) > 1 && (known_busy || !device_idle(h)))
            return PSDC_OK;
        return advance(h);
    }
    return PSDC_OK;
}

int psdc_record_consumed(psdc_handle *h, void *consumed_event)
{
    if (!h || !consumed_event)
        return fail(h, PSDC_ERR_ARG, "null argument");
    ON_DEVICE(h, h->device);
    bool pend = false;
    for (auto &c : h->ch)
        pend = pend || c.has_span() || c.submitted;
    if (pend) { // spans held back for coalescing go out now
        int rc = advance(h);
        if (rc)
            return rc;
    }
    // the tail carry of a round (deferred into the next launch) still reads the end of the caller's span
    int rc = launch_deferred(h, {});
    if (rc)
        return rc;
    HIPCHK(h, hipEventRecord(static_cast<hipEvent_t>(consumed_event), h->stream));
    return PSDC_OK;
}

int psdc_loss_read(psdc_handle *h, psdc_loss *out, int reset)
{
    if (!h || !out)
        return fail(h, PSDC_ERR_ARG, "null argument");
    *out = h->loss;
    if (reset)
        h->loss = psdc_loss{};
    return PSDC_OK;
}

int psdc_flush(psdc_handle *h)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    ON_DEVICE(h, h->device);
    return flush_all(h);
}

int psdc_sync(psdc_handle *h)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    ON_DEVICE(h, h->device);
    return flush_sync(h);
}

} // extern "C"
