// psdcascade.cpp -- host runtime behind include/psdcascade.h.
//
// Mirrors PsdCascade<N> (src/psd.rs:399-544) for `n_channels` independent
// traces on one MI355X: per (channel, stage) it keeps the stream position, the
// count and a device stream buffer, turns each batch of newly completed
// segments into kernel jobs (plan.h gives the closed forms of the reference's
// per-sample loop), and stitches the read-out on the host.
//
// There is no CPU compute path: every spectrum comes from the HIP kernels.
#include "../../include/psdcascade.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hbf_taps.h"
#include "kernels.h"
#include "plan.h"

using namespace psdk;

namespace {

constexpr uint32_t MAX_STAGES = 16; // 8^16 N samples: unreachable; slots of the spectra slab

thread_local std::string g_last_error;

struct DevBuf {
    bool pooled = false;              // p[] are slots of the handle's deep-stage pool (never freed singly)
    float *p[2] = {nullptr, nullptr}; // ping-pong: the tail is carried to the other buffer
    int cur = 0;
    size_t cap = 0;    // floats per buffer
    uint64_t base = 0; // absolute stream index of p[cur][0]
    uint64_t end = 0;  // the buffer holds [base, end); == total unless a zero-copy span is pending
};

struct StageState {
    uint64_t total = 0; // samples received by this stage (absolute end of its stream)
    uint64_t segs = 0;  // segments issued (J)
    uint64_t dec = 0;   // samples handed to the decimator (P)
    uint32_t count = 0; // PsdStage::count (src/psd.rs:128) as reported: count_report(count64)
    uint64_t count64 = 0; // the count in 64 bits (plan.h count_after64): gain() past 2^32 segments
    bool sink = false;    // single-stage handles (psdc_stage_*): this stream is handed to the caller, never analysed
    uint64_t sink_pos = 0; // ... and everything before this absolute index has been handed over
    DevBuf buf;
    float *spectrum = nullptr; // device, n floats (first n/2+1 used, src/psd.rs:127)
};

struct DeviceSpan {
    const float *d_x = nullptr;
    uint64_t first = 0; // absolute index of d_x[0] in the stage-0 stream
    size_t len = 0;
    // ... or trace fch of a run of AdcDac frames in device memory (d_x == nullptr): the samples are read in place as wire
    // words (psdc_process_adcdac_frames_device), sample first + i of the stream = sample i of the trace
    FrameSpan fs{};
    int fch = 0;
    bool framed() const { return fs.frames != nullptr; }
};

struct Channel {
    std::vector<StageState> st;
    float *stage_host[2] = {nullptr, nullptr}; // pinned staging (host-fed samples)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool ev_pending[2] = {false, false};
    int cur_stage = 0;
    size_t fill = 0;
    bool submitted = false; // device holds samples that advance() has not looked at yet
    // zero-copy spans registered but not enqueued yet, in stream order.  More than one is held
    // while the device is still busy with earlier rounds (PSDC_OPT_COALESCE): they go out as ONE
    // round, which halves / quarters the per-round launch overhead per sample.
    std::vector<DeviceSpan> spans;
    bool has_span() const { return !spans.empty(); }
    size_t span_max = 0;         // longest in-place span seen
    bool coalesced_seen = false; // some round of this channel carried more than one span
};

struct ProfEvents {
    hipEvent_t a, b;
};

} // namespace

struct psdc_handle {
    uint32_t n = 0;
    int window_kind = PSDC_WINDOW_HANN;
    Geometry geo;
    float nenbw = 1.5f, power = 0.25f;
    std::vector<float> win_host; // the Window's weights as uploaded (psdc_window_get, psdc_clone, pack_readout)
    uint32_t n_channels = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    // Uploads (host-fed samples, frame blobs) run on a stream of their own so that the link works
    // while the kernels of the previous piece run; `ev_upload` marks the last upload enqueued, and
    // the compute stream waits for it before it touches anything (wait_uploads()).
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_upload = nullptr;
    bool upload_pending = false;
    // An upload lands in the stage-0 buffer that the round BEFORE the latest one read (ping-pong),
    // whose end region is also the source of that round's tail carry -- which is deferred into the
    // latest round's post launch.  So an upload may overlap the latest round's fused kernel but must
    // wait for its post launch: ev_post is recorded right behind it.
    hipEvent_t ev_post = nullptr;
    bool post_marked = false;
    float *d_win = nullptr;
    cf *d_tw = nullptr;
    cf *d_tw0g = nullptr, *d_twag = nullptr; // twiddle tables of the N >= 2048 fused kernels
    cf *d_tw3g = nullptr;                    // twiddle seeds of the three-pass kernels (N = 2048, 4096)
    cf *d_chirp = nullptr, *d_bhat = nullptr; // chirp-z tables of a size that is not a power of two (launch_welch)
    cf *d_bigfft = nullptr;                   // n > 16384: the ping-pong frames of the global-memory FFT (launch_welch_big)
    size_t bigfft_elems = 0;
    int bigfft_chunk_limit = 0;               // PSDC_DBG_BIGFFT_CHUNK at create: pairs per chunk (tests: a job split over chunks)
    int detrend = PSDC_DETREND_NONE;
    uint32_t avg_limit = 0xFFFFFFFFu, avg_count = 0xFFFFFFFFu;
    std::vector<Channel> ch;
    float *d_spectra = nullptr; // [n_channels][MAX_STAGES][n] accumulators, one slab
    float *h_read = nullptr;    // pinned bounce buffer for read-outs (MAX_STAGES * n floats)
    unsigned long long *d_scan = nullptr; // 5 words: accumulators of the device-side frame header scan + Loss sums (kept zero)
    uint8_t *h_hdr = nullptr;             // pinned: the frame headers of one psdc_process_frames_device call (launch_header_gather)
    size_t h_hdr_cap = 0;                 // bytes
    hipStream_t hdr_stream = nullptr;     // the gather runs beside the compute stream's work (the host waits for it alone)
    unsigned long long *h_scan = nullptr; // pinned: its four result words
    hipStream_t scan_stream = nullptr;    // the scan runs beside the compute stream's work (the host waits for it alone)
    std::vector<FrameSpan> fs_pool;       // frame spans named by this round's jobs (FusedJob::fspan ... index this until a launch maps them)
    float *d_pool = nullptr;    // [n_channels][MAX_STAGES][2][pool_cap] small stream buffers (deep stages)
    size_t pool_cap = 0;        // floats per pooled buffer
    bool idle = true;           // nothing ingested since the pipeline was last drained
    float *d_partial = nullptr;
    size_t partial_cap = 0; // floats
    // stream buffers replaced by larger ones: work already enqueued may still read them, so they are freed
    // at the next point where the stream is known to be idle (release_retired) -- growing never waits
    std::vector<float *> retired;
    // epilogue of the last round (fold the partials into the spectra, carry the stream tails),
    // not launched yet: it rides in the first launch of the next round or of a read-out
    std::vector<RedJob> pend_red;
    std::vector<TailJob> pend_tail;
    // frame ingest: two pinned bounce buffers and their device images, used alternately
    uint8_t *d_frames[2] = {nullptr, nullptr};
    uint8_t *h_frames[2] = {nullptr, nullptr};
    hipEvent_t frames_ev[2] = {nullptr, nullptr}; // H2D of the buffer finished
    hipEvent_t frames_dec_ev[2] = {nullptr, nullptr}; // the decode kernel has read the device image
    bool frames_dec_pending[2] = {false, false};
    bool frames_ev_pending[2] = {false, false};
    size_t frames_cap = 0; // bytes per buffer
    int frames_cur = 0;
    size_t quantum = (size_t)1 << 22;
    uint32_t coalesce = 8; // zero-copy spans per channel held back while the device is busy (1 = none)
    bool coalesce_auto = true; // PSDC_OPT_COALESCE not set: `coalesce`, or MAX_COALESCE for one channel fed in short spans (coalesce_limit)
    uint32_t stage_limit = MAX_STAGES; // stages that analyse their stream; 1 for a single Psd<N> (psdc_stage_*)
    uint32_t min_pairs = 0; // PSDC_OPT_MIN_PAIRS: segment pairs a decimated stage collects before it issues on the ingest path
    bool coalesce_always = false; // hold them back even when the device is idle (tests)
    bool profile = false;
    std::vector<ProfEvents> prof_pending;
    psdc_profile prof{};
    psdc_loss loss{};
    std::string err;
};

namespace {

int fail(psdc_handle *h, int code, const std::string &msg)
{
    if (h)
        h->err = msg;
    g_last_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, PSDC_ERR_DEVICE,                                                      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// Every ABI entry runs on the handle's device and leaves the caller's current device as it found it
// (a caller with several GPUs -- one handle per device, or torch's current device -- must not see it move).
struct DevScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DevScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != dev) {
            err = hipSetDevice(dev);
            changed = (err == hipSuccess && prev >= 0);
        }
    }
    ~DevScope()
    {
        if (changed)
            (void)hipSetDevice(prev);
    }
    DevScope(const DevScope &) = delete;
    DevScope &operator=(const DevScope &) = delete;
};
#define ON_DEVICE(h, dev)                                                                        \
    DevScope dev_scope_(dev);                                                                    \
    if (dev_scope_.err != hipSuccess)                                                            \
        return fail(h, PSDC_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(dev_scope_.err))

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

struct WindowConsts {
    float nenbw, power;
    uint32_t overlap;
};

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

// pool index -> index in a launch's own table (at most MAX_FSPANS distinct spans per launch: the planner holds a channel
// to MAX_COALESCE = MAX_FSPANS spans per round, and the four traces of a span share one entry)
struct FspanMap {
    FrameSpan *table;
    int used = 0;
    int pool_of[MAX_FSPANS];
    explicit FspanMap(FrameSpan *t) : table(t) {}
    int map(const psdc_handle *h, int pool_idx)
    {
        if (pool_idx < 0)
            return -1;
        for (int i = 0; i < used; ++i)
            if (pool_of[i] == pool_idx)
                return i;
        if (used >= MAX_FSPANS)
            return -2;
        pool_of[used] = pool_idx;
        table[used] = h->fs_pool[(size_t)pool_idx];
        return used++;
    }
};

// The epilogue of a round -- fold its partials (RedJob), carry its stream tails (TailJob) -- is
// not launched when the round ends: the next round starts with a copy launch of its own (the
// zero-copy seams), and one launch does both.  Nothing on the device reads what the epilogue
// writes before that point; host-visible state never waits for it (read-outs drain first).
int launch_deferred(psdc_handle *h, const std::vector<TailJob> &extra)
{
    // a copy job is one workgroup of post_kernel: long tails (a stage that collects a batch keeps up to
    // PSDC_OPT_MIN_PAIRS pairs pending) are cut into pieces so that the launch does not wait on one workgroup
    constexpr int kPiece = 16384;
    std::vector<TailJob> tails;
    tails.reserve(h->pend_tail.size() + extra.size());
    auto add_tail = [&](const TailJob &t) {
        for (int o = 0; o < t.count; o += kPiece) {
            TailJob q = t;
            q.src = t.src ? t.src + o : nullptr;
            q.dst = t.dst + o;
            q.count = std::min(kPiece, t.count - o);
            q.s_off = t.s_off + (unsigned)o;
            tails.push_back(q);
        }
    };
    for (const TailJob &t : h->pend_tail)
        add_tail(t);
    for (const TailJob &t : extra)
        add_tail(t);
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
int ensure_cap(psdc_handle *h, StageState &s, size_t need, size_t grow_to = 0)
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
    HIPCHK(h, hipMalloc(&h->d_partial, sizeof(float) * cap));
    h->partial_cap = cap;
    return PSDC_OK;
}

constexpr int MAX_COALESCE = 16; // zero-copy spans of one channel in one round
static_assert(MAX_COALESCE <= MAX_FSPANS, "a launch's frame-span table holds every span of a round");

struct Span { // one contiguous source of a (channel, stage) batch
    const float *src;
    uint64_t src_base;
    uint64_t seg_a, seg_b; // segments [seg_a, seg_b)
    uint64_t m_a, m_b;     // decimator outputs [m_a, m_b)
    bool fixed = false;    // src is not the start of the stage's stream buffer (caller memory or a
                           // seam region inside the buffer): leave it alone when buffers are re-based
    int fpool = -1;        // >= 0: the source is trace fch of frame span fs_pool[fpool] read in place (src == nullptr)
    int fch = 0;
};

struct Work {
    uint32_t c, k;
    uint64_t j_old, j_new, p_old, p_new;
    EwmaPlan ew;
    Span spans[2 * MAX_COALESCE];
    int nspans = 0;
};

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

// nothing of this handle is executing or queued on the device
// In-place spans of a channel that may share a round.  A round costs ~20 us of launch boundaries whatever it holds (the post
// launch and two dependent dispatches): 8 spans of 2^26 samples are 0.7 ms of kernel, 8 of 2^22 are 50 us -- so a single channel fed in
// spans of at most 2^25 samples may hold sixteen (2^24 a call: +3 %, 2^22: +21 %; eight channels x 2^24 measured -2 % with sixteen and stay
// at eight).  An explicit PSDC_OPT_COALESCE is taken as given.
uint32_t coalesce_limit(const psdc_handle *h, const Channel &c, size_t len = 0)
{
    if (h->coalesce_auto && h->n_channels == 1 && std::max(c.span_max, len) <= ((size_t)1 << 25))
        return MAX_COALESCE;
    return h->coalesce;
}

bool device_idle(psdc_handle *h) { return !h->coalesce_always && hipStreamQuery(h->stream) == hipSuccess; }

// One round of the cascade pipeline: every (channel, stage) that has complete
// segments in its stream buffer is issued, all stages in the SAME launches.
// The decimator output of this round becomes visible to the next stage in the
// next round (stage k+1 lags one round behind stage k), so a steady-state round
// costs two launches whatever the depth: a post launch (the seam copy of this round with the
// deferred epilogue of the last one) and the fused launch (plus the generic welch / decimator
// kernels when something does not fit a pair).
// The fused single-pass kernels read the window from its table and assume nothing about it but a hop of N/2: Window::hann()
// and every caller-built Window<N> with overlap N/2 (Hamming, Blackman, ... -- src/psd.rs:12-20 has pub fields) run on them;
// overlap 0 (Window::rectangular(), src/psd.rs:24-32, or a caller's table) runs the same kernels in their SINGLE form -- one
// segment per "pair", transformed with a zero imaginary part; the decimator consumes the stream exactly as before (round 4:
// rectangular windows ran the generic two-pass kernels at a third of the rate).  Other overlaps take the generic kernels.
// 0: no fused kernel for this window, 1: half-overlapped pairs, 2: single segments.
static int fused_window(const psdc_handle *h)
{
    if (h->window_kind == PSDC_WINDOW_HANN || (h->window_kind == PSDC_WINDOW_CUSTOM && 2 * (uint64_t)h->geo.overlap == h->n))
        return 1;
    static const bool no_single = getenv("PSDC_NO_SINGLE") != nullptr; // (A/B aid: rectangular windows on the generic kernels)
    if (h->geo.overlap == 0 && !no_single)
        return 2;
    return 0;
}

// `all`: issue odd segments of decimated stages too (read-outs); the ingest path
// leaves them for their partner.  *did_work tells whether anything was issued;
// read-outs call rounds until idle.
int advance_round(psdc_handle *h, bool *did_work, bool all)
{
    const Geometry &g = h->geo;
    const int spt = welch_segments_per_tile((int)h->n);
    const int fmode = fused_supported((int)h->n) ? fused_window(h) : 0;
    const bool fast_ok = fmode != 0, single = fmode == 2;
    // overlap 0 at the team-kernel sizes: two disjoint segments per transform (FusedBatch::single == 2) -- jobs and runs then hold
    // an even number of single-segment "pairs".  $PSDC_NO_DOUBLE: one segment per transform everywhere (A/B aid)
    static const bool no_double = getenv("PSDC_NO_DOUBLE") != nullptr;
    const bool dbl = single && !no_double && fused_double_supported((int)h->n);
    const unsigned fstep = single ? 1 : 2;   // segments per fused "pair"
    // a fused run starting at segment j decimates from sample j hop + N/2 on (half-overlapped pairs: the pair's new samples) -- or,
    // overlap 0, from j N on: a single-segment step transforms exactly the samples it decimates, and its source pointer sits half
    // a segment in front of the segment (run_src0)
    const uint64_t half = (uint64_t)h->n / 2;
    auto run_new0 = [&](uint64_t seg) { return seg * (uint64_t)g.hop + (single ? 0 : half); };
    auto run_src0 = [&](uint64_t seg) { return seg * (uint64_t)g.hop - (single ? half : 0); }; // (seg >= 1 in single mode)
    // fused runs rebuild their decimator state from the 288 samples before their first new
    // sample (which sits N/2 after the run's first segment start): samples needed in front of it
    const uint64_t need_pre = HBF_HALO > half ? HBF_HALO - half : 0;
    // the seam must complete every segment that starts in the carried tail; on the fast path it
    // is long enough for the tail side to end on a whole segment pair with need_pre samples of
    // the new span in front of the in-place side
    const uint64_t seam = std::max<uint64_t>((uint64_t)h->n + HBF_HALO, fast_ok ? need_pre + 3 * half : 0);
    *did_work = false;
    {
        int rc = wait_uploads(h);
        if (rc)
            return rc;
    }

    // zero-copy spans: copy the seam (the part that completes segments begun in
    // the carried tail) behind the tail; the bulk is read in place.  One copy
    // launch for all channels.  A channel may hold several spans (PSDC_OPT_COALESCE): each
    // further span gets a seam REGION of its own in the stream buffer, behind the contiguous
    // part -- the tail the span before it would have carried (read from that span's end) followed
    // by the head of the span -- so that the segments straddling two spans see contiguous memory.
    struct Region { // seam region of span i >= 1 of a channel
        size_t off;      // floats from the start of the stream buffer
        uint64_t base;   // absolute index of its first sample (the keep_from point after span i-1)
    };
    std::vector<std::vector<Region>> regions(h->n_channels);
    {
        std::vector<TailJob> seams;
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
            seams.push_back(span_copy(h, c.spans[0], c.spans[0].first, buf + (c.spans[0].first - s0.buf.base), (size_t)cp0));
            s0.buf.end = c.spans[0].first + cp0;
            for (size_t i = 1; i < ns; ++i) {
                const Region &r = regions[ci][i];
                const DeviceSpan &pv = c.spans[i - 1], &sp = c.spans[i];
                const size_t back = (size_t)(sp.first - r.base);
                seams.push_back(span_copy(h, pv, r.base, buf + r.off, back));
                seams.push_back(span_copy(h, sp, sp.first, buf + r.off + back, (size_t)std::min<uint64_t>(seam, sp.len)));
            }
        }
        int rc = launch_deferred(h, seams); // with the last round's epilogue
        if (rc)
            return rc;
        h->fs_pool.clear(); // every job that named a pooled span has been launched; this round's jobs pool theirs afresh
        HIPCHK(h, hipEventRecord(h->ev_post, h->stream)); // see order_upload
        h->post_marked = true;
    }

    // collect the work of this round from the totals as they stand now
    std::vector<Work> works;
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
    if (works.empty()) {
        for (auto &c : h->ch) {
            if (c.has_span())
                return fail(h, PSDC_ERR_DEVICE, "internal: zero-copy span left unconsumed");
            c.submitted = false;
        }
        return PSDC_OK;
    }
    *did_work = true;

    // where every stream will start after this round (its tail is carried to the
    // front of its other buffer; the decimator appends the new samples behind it)
    auto kf_after = [&](uint32_t ci, uint32_t k) -> uint64_t {
        StageState t = h->ch[ci].st[k];
        for (auto &w : works)
            if (w.c == ci && w.k == k) {
                t.segs = w.j_new;
                t.dec = w.p_new;
            }
        return keep_from(g, t);
    };
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
                const uint64_t round_max = (uint64_t)c.span_max * (c.coalesced_seen ? coalesce_limit(h, c) : 1);
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

    // ---- turn the work into kernel jobs ---------------------------------
    // Fast path (fused_kernel / bigfused_kernel): whole segment pairs of a Hann stream with
    // N = 256 ... 16384, any implemented detrend, plain-sum or EWMA averaging, 16-byte aligned.
    // Everything else -- other N, the rectangular window, an odd last segment, the decimator
    // ranges a pair does not cover (the first segment of a stream is decimated whole,
    // src/psd.rs:235-238; outputs still inside the drain; unaligned zero-copy spans) -- goes
    // through the generic welch / hbf_dec8 kernels.  Both write the same partial slab and
    // next-stage stream, so the reduce and the bookkeeping do not care which ran.
    struct PlanFused { FusedJob j; size_t work; };
    struct PlanSeg { SegJob j; size_t work; };
    std::vector<PlanFused> fjobs;
    std::vector<PlanSeg> sjobs;
    std::vector<DecJob> djobs;
    uint64_t prof_samples = 0, prof_samples0 = 0;
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

    // share the persistent workgroups so that every workgroup walks about the same amount
    // shares are computed per launch batch (MAX_JOBS jobs): every launch fills the GPU by itself
    size_t blocks_total = 0;
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
    const uint64_t teams = (uint64_t)std::max(1, fused_pairs_per_block((int)h->n, 1));
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
    for (size_t b0 = 0; b0 < fjobs.size(); b0 += MAX_JOBS) {
        const size_t b1 = std::min(fjobs.size(), b0 + (size_t)MAX_JOBS);
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
            // (every job holds at least one workgroup: with more large jobs than slots -- MAX_JOBS = 128 against >= 248 slots, so
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
    int rc = ensure_partial(h, blocks_total * h->n);
    if (rc)
        return rc;
    // slab: the partials of one work are contiguous (fused first, then generic)
    std::vector<RedJob> rjobs(works.size());
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
            rj.partial = h->d_partial + slab;
            rj.spectrum = h->ch[works[wi].c].st[works[wi].k].spectrum;
            rj.g_total = (float)works[wi].ew.g_total;
            rj.nparts = (int)nblk[wi];
            slab += nblk[wi] * h->n;
        }
        if (slab > h->partial_cap)
            return fail(h, PSDC_ERR_DEVICE, "internal: partial slab overflow");
        for (PlanFused &pf : fjobs) {
            pf.j.partial = h->d_partial + base[pf.work];
            base[pf.work] += (size_t)pf.j.nblocks * h->n;
        }
        for (PlanSeg &ps : sjobs) {
            ps.j.partial = h->d_partial + base[ps.work];
            base[ps.work] += (size_t)ps.j.nblocks * h->n;
        }
    }

    // ---- launches: fused, generic welch, reduce, generic decimator -------
    // HIP events time the dominant kernel of the round (fused when present).  The fused launches
    // hand their events to hipExtLaunchKernelGGL, which stamps the kernel's own start and stop (what
    // rocprofv3 --kernel-trace reports); events recorded around a launch would include the ~6 us
    // dependent-dispatch gap in front of it.  The generic welch kernel keeps the bracket.
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
        for (; i < fjobs.size() && fb.njobs < MAX_JOBS; ++i) {
            FusedJob j = fjobs[i].j;
            j.block_begin = fb.nblocks;
            fb.nblocks += j.nblocks;
            fb.any_ewma |= j.ewma;
            if (j.fspan >= 0) {
                fb.any_frames = 1;
                if ((j.fspan = fm.map(h, j.fspan)) < 0)
                    return fail(h, PSDC_ERR_DEVICE, "internal: frame span table");
            }
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
        ProfEvents pe{};
        const bool first = (i <= (size_t)MAX_JOBS);
        if ((rc = prof_begin(pe, false)))
            return rc;
        HIPCHK(h, launch_fused((int)h->n, fb, h->d_win, h->d_tw0g, h->d_twag, h->d_tw3g, h->stream, pe.a, pe.b));
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
    std::vector<TailJob> tjobs;
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
    // epilogue (fold the partials, carry the tails): deferred to the next launch_deferred()
    h->pend_red = std::move(rjobs);
    h->pend_tail = std::move(tjobs);
    for (auto &c : h->ch) {
        c.spans.clear();
        c.submitted = false;
    }
    return PSDC_OK;
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

int stitch_impl(uint32_t n, float nenbw, float power, uint32_t overlap, uint32_t n_stages,
                const uint32_t *counts, const uint32_t *avgs, const uint64_t *pendings,
                const float *spectra, int keep_overlap, uint32_t min_count, int keep_transition_band,
                float *psd_out, size_t psd_cap, size_t *psd_len, psdc_break *breaks,
                size_t breaks_cap, size_t *n_breaks, const uint64_t *counts64 = nullptr)
{
    // PsdCascade::psd (src/psd.rs:479-543); counts64: the counts in 64 bits for gain() where the u32 saturated
    const size_t bins = n / 2 + 1;
    size_t plen = 0, nb = 0;
    uint64_t decimation = 1ull << (3 * n_stages); // :482
    size_t end = 0;
    bool overflow = false;
    for (int si = (int)n_stages - 1; si >= 0; --si) { // .rev() :484
        decimation >>= 3;
        const size_t start = keep_overlap ? 0 : ((end + 7) >> 3);                     // :490-495
        end = (decimation > 1 && !keep_transition_band) ? (size_t)(2 * n / 5) : bins; // :496-501
        const bool include = counts[si] >= min_count;                                 // :502
        if (breaks) {
            if (nb < breaks_cap) {
                psdc_break &b = breaks[nb];
                b.start = plen;
                b.include = include ? 1u : 0u;
                b.count = counts[si];
                b.avg = avgs[si];
                b._pad = 0;
                b.bins_start = start;
                b.bins_end = end;
                b.fft_size = n;
                b.decimation = decimation;
                const uint32_t cm1 = counts[si] ? counts[si] - 1 : 0; // saturating_sub(1)
                b.processed = (uint64_t)n * counts[si] - (uint64_t)overlap * cm1; // :511-512
                b.pending = pendings[si];
            } else {
                overflow = true;
            }
        }
        ++nb;
        if (include) { // :515-517
            const float gsc = 1.0f / (stage_gain(n, counts64 ? counts64[si] : counts[si], nenbw, power) * (float)decimation);
            for (size_t k = start; k < end; ++k) {
                if (psd_out) {
                    if (plen < psd_cap)
                        psd_out[plen] = spectra[(size_t)si * bins + k] * gsc;
                    else
                        overflow = true;
                }
                ++plen;
            }
        } else {
            end = start; // :518-520
        }
    }
    if (psd_len)
        *psd_len = plen;
    if (n_breaks)
        *n_breaks = nb;
    return overflow ? PSDC_ERR_CAPACITY : PSDC_OK;
}

} // namespace

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

namespace {

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
    if ((e = hipMalloc(&h->d_partial, sizeof(float) * h->partial_cap)) != hipSuccess)
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

} // namespace

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
        const int64_t k = value < 0 ? -value : value; // negative: hold spans back even on an idle device
        if (k < 1 || k > MAX_COALESCE)
            return fail(h, PSDC_ERR_ARG, "coalesce out of range (1..16)");
        int rc = flush_all(h);
        if (rc)
            return rc;
        h->coalesce = (uint32_t)k;
        h->coalesce_auto = false;
        h->coalesce_always = value < 0;
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
    const bool in_place = len >= (size_t)4 * (h->n + HBF_HALO);
    // Earlier spans of this channel must go out first -- unless this one can join them: an in-place
    // span behind in-place spans, fewer than PSDC_OPT_COALESCE of them, and a device that is still
    // busy with earlier rounds (when it is idle nothing is ever held back).
    // (the stream is asked at most ONCE per call -- ~0.1 us on an idle or a busy stream, tools/probes/stream_query.cpp; a "busy" answer
    // stands for the rest of the call)
    bool flush = c.submitted, known_busy = false;
    if (c.has_span()) {
        if (!in_place || c.fill > 0 || c.spans.size() >= coalesce_limit(h, c, len))
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
        if (c.has_span() && !c.submitted && c.spans.size() < coalesce_limit(h, c) && (known_busy || !device_idle(h)))
            return PSDC_OK; // the device is busy: the next span may share this one's round
        return advance(h);
    }
    return PSDC_OK;
}

namespace {
// Payload layouts by Format id (src/de/mod.rs:12-17; src/de/data.rs:13, 86, 144, 168): bytes per batch, samples per batch and
// trace, traces (Payload::traces).
struct WireFmt {
    int id;
    size_t batch_bytes;
    int spb, ntr;
    const char *what;
};
const WireFmt *wire_fmt(int id)
{
    static const WireFmt t[4] = {{1, 64, 8, 4, "AdcDac"}, {2, 56, 1, 4, "Fls"}, {3, 80, 1, 4, "ThermostatEem"}, {4, 24, 1, 3, "Mpll"}};
    return id >= 1 && id <= 4 ? &t[id - 1] : nullptr;
}

// Frames in host memory, frame by frame as Source::get does for Data::File / Data::Udp (src/source.rs:135-142, 158-165):
// Frame::from_bytes (src/de/frame.rs:49-60), Loss::update (src/loss.rs:11-26), Payload::traces into channels 0 .. ntraces - 1
// (src/bin/psd.rs:174-182: trace i goes to cascade i whatever the frame's format).  The frames are taken in RUNS of one format:
// within a run the headers are validated on the host, the payloads uploaded in pieces and decoded on the device.
// adcdac_only: any other valid format id is de::Error::UnknownFormat's code, as psdc_process_adcdac_frames documents.
int ingest_frames_host(psdc_handle *h, bool adcdac_only, const uint8_t *frames, size_t frame_size, size_t n_frames, size_t *n_ok)
{
    if (n_ok)
        *n_ok = 0;
    int rc = check_channel(h, 0);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (adcdac_only && h->n_channels < 4)
        return fail(h, PSDC_ERR_ARG, "AdcDac frames carry four traces: need n_channels >= 4");
    if (n_frames == 0)
        return PSDC_OK;
    if (!frames)
        return fail(h, PSDC_ERR_ARG, "null input");
    if (frame_size < 8) // &input[..HEADER_SIZE] panics (src/de/frame.rs:50)
        return fail(h, PSDC_ERR_FRAME_SIZE, "frame shorter than its header");
    size_t good = 0;
    int bad = PSDC_OK;
    const size_t payload = frame_size - 8;
    size_t f0 = 0;
    while (f0 < n_frames && bad == PSDC_OK) {
        // the run's format: its first frame's (Header::parse, src/de/frame.rs:25-37)
        const uint8_t *first = frames + f0 * frame_size;
        if (first[0] != 0x7b || first[1] != 0x05) {
            bad = PSDC_ERR_FRAME_HEADER;
            break;
        }
        const WireFmt *wf = wire_fmt(first[2]);
        if (!wf || (adcdac_only && wf->id != 1)) { // unknown id -- or Fls / ThermostatEem / Mpll where only AdcDac is asked for
            bad = PSDC_ERR_FRAME_FORMAT;
            break;
        }
        if ((int)h->n_channels < wf->ntr) {
            if (n_ok)
                *n_ok = good;
            return fail(h, PSDC_ERR_ARG, "the frames carry more traces than the handle has channels");
        }
        const int ntr = wf->ntr;
        const int batches = (int)(payload / wf->batch_bytes);
        bool run_end = false; // a frame of another (valid) format: the next run starts there
        // host: validate headers (src/de/frame.rs:25-37, src/de/data.rs:22-25, 91-93, 149-150, 173-174) and keep the loss
        // counters (Loss::update, src/loss.rs:11-26), piece by piece inside the upload loop below so
        // that the scan of one piece runs while the piece before it is on the link
        auto scan = [&](size_t fa, size_t cnt) -> size_t { // frames accepted from fa on; sets `bad` at the first bad one
            for (size_t i = 0; i < cnt; ++i) {
                const uint8_t *f = frames + (fa + i) * frame_size;
                if (f[0] != 0x7b || f[1] != 0x05) {
                    bad = PSDC_ERR_FRAME_HEADER;
                    return i;
                }
                if (f[2] != wf->id) {
                    if (!adcdac_only && wire_fmt(f[2])) {
                        run_end = true;
                        return i;
                    }
                    bad = PSDC_ERR_FRAME_FORMAT; // unknown id (or, for psdc_process_adcdac_frames, not AdcDac)
                    return i;
                }
                if (payload % wf->batch_bytes != 0 || (int)f[3] != batches) {
                    bad = PSDC_ERR_FRAME_SIZE;
                    return i;
                }
                const uint32_t seq = (uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24);
                h->loss.received += f[3];
                if (h->loss.have_seq)
                    h->loss.dropped += (uint32_t)(seq - h->loss.next_seq); // wrapping_sub
                h->loss.next_seq = seq + f[3];                              // wrapping_add
                h->loss.have_seq = 1;
            }
            return cnt;
        };
        if (batches == 0) {
            const size_t cnt = scan(f0, n_frames - f0); // header-only frames carry no samples
            good += cnt;
            f0 += cnt;
            continue;
        }
        // order behind anything pending on these channels
        bool pend = false;
        for (int ci = 0; ci < ntr; ++ci)
            pend = pend || h->ch[ci].has_span() || h->ch[ci].submitted || h->ch[ci].fill;
        if (pend) {
            rc = flush_all(h);
            if (rc)
                return rc;
        }
        // The frames go up in pieces of ~16 MiB through two pinned buffers: while one piece is on
        // the link the host copies the next (several threads), and every piece is decoded and
        // cascaded as soon as it has landed.
        const size_t piece_frames = std::max<size_t>(1, ((size_t)16 << 20) / frame_size);
        const size_t piece_bytes = piece_frames * frame_size;
        if (piece_bytes > h->frames_cap) {
            HIPCHK(h, hipStreamSynchronize(h->copy_stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            for (int i = 0; i < 2; ++i) {
                if (h->d_frames[i])
                    HIPCHK(h, hipFree(h->d_frames[i]));
                if (h->h_frames[i])
                    HIPCHK(h, hipHostFree(h->h_frames[i]));
                h->d_frames[i] = nullptr;
                h->h_frames[i] = nullptr;
                HIPCHK(h, hipMalloc(&h->d_frames[i], piece_bytes));
                HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_frames[i]), piece_bytes,
                                        hipHostMallocDefault));
                if (!h->frames_ev[i])
                    HIPCHK(h, hipEventCreateWithFlags(&h->frames_ev[i], hipEventDisableTiming));
                h->frames_ev_pending[i] = false;
                h->frames_dec_pending[i] = false;
            }
            h->frames_cap = piece_bytes;
        }
        h->idle = false;
        while (f0 < n_frames && bad == PSDC_OK && !run_end) {
            const size_t cnt = scan(f0, std::min(piece_frames, n_frames - f0));
            good += cnt;
            if (cnt == 0)
                break;
            const size_t bytes = cnt * frame_size;
            const int b = h->frames_cur;
            if (h->frames_ev_pending[b]) { // the bounce buffer's last upload must have left it
                HIPCHK(h, hipEventSynchronize(h->frames_ev[b]));
                h->frames_ev_pending[b] = false;
            }
            CopyPool::get().copy(h->h_frames[b], frames + f0 * frame_size, bytes);
            if (h->frames_dec_pending[b]) { // the decode kernel of two pieces ago has read this device image
                HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->frames_dec_ev[b], 0));
                h->frames_dec_pending[b] = false;
            }
            HIPCHK(h, hipMemcpyAsync(h->d_frames[b], h->h_frames[b], bytes, hipMemcpyHostToDevice, h->copy_stream));
            HIPCHK(h, hipEventRecord(h->frames_ev[b], h->copy_stream));
            rc = mark_upload(h);
            if (rc)
                return rc;
            h->frames_ev_pending[b] = true;
            h->frames_cur = b ^ 1;
            const size_t per_ch = cnt * (size_t)batches * (size_t)wf->spb;
            float *dst[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int ci = 0; ci < ntr; ++ci) {
                Channel &c = h->ch[ci];
                if (c.st.empty()) {
                    rc = add_stage(h, c);
                    if (rc)
                        return rc;
                }
                StageState &s0 = c.st[0];
                rc = ensure_room(h, s0, s0.total + per_ch);
                if (rc)
                    return rc;
                dst[ci] = s0.buf.p[s0.buf.cur] + (s0.total - s0.buf.base);
            }
            rc = wait_uploads(h); // the decode kernel reads what the copy stream is bringing
            if (rc)
                return rc;
            if (wf->id == 1)
                HIPCHK(h, launch_adcdac(h->d_frames[b], frame_size, cnt, batches, dst[0], dst[1], dst[2], dst[3], h->stream));
            else
                HIPCHK(h, launch_payload(wf->id, h->d_frames[b], frame_size, cnt, batches, dst[0], dst[1], dst[2], dst[3], h->stream));
            // the device image d_frames[b] is written again two pieces later: that upload waits for this
            if (!h->frames_dec_ev[b])
                HIPCHK(h, hipEventCreateWithFlags(&h->frames_dec_ev[b], hipEventDisableTiming));
            HIPCHK(h, hipEventRecord(h->frames_dec_ev[b], h->stream));
            h->frames_dec_pending[b] = true;
            for (int ci = 0; ci < ntr; ++ci) {
                h->ch[ci].st[0].total += per_ch;
                h->ch[ci].st[0].buf.end = h->ch[ci].st[0].total;
                h->ch[ci].submitted = true;
            }
            rc = advance(h);
            if (rc)
                return rc;
            f0 += cnt;
        }
    }
    if (n_ok)
        *n_ok = good;
    if (bad != PSDC_OK)
        return fail(h, bad,
                    bad == PSDC_ERR_FRAME_HEADER   ? "Invalid frame header"
                    : bad == PSDC_ERR_FRAME_FORMAT ? (adcdac_only ? "Unknown or non-AdcDac format ID" : "Unknown format ID")
                                                   : "Payload size");
    return PSDC_OK;
}
} // namespace

int psdc_process_adcdac_frames(psdc_handle *h, const uint8_t *frames, size_t frame_size, size_t n_frames, size_t *n_ok)
{
    return ingest_frames_host(h, true, frames, frame_size, n_frames, n_ok);
}

int psdc_process_frames(psdc_handle *h, const uint8_t *frames, size_t frame_size, size_t n_frames, size_t *n_ok)
{
    return ingest_frames_host(h, false, frames, frame_size, n_frames, n_ok);
}

int psdc_process_adcdac_frames_device(psdc_handle *h, const uint8_t *d_frames, size_t frame_size, size_t n_frames,
                                      size_t *n_ok)
{
    if (n_ok)
        *n_ok = 0;
    int rc = check_channel(h, 0);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (h->n_channels < 4)
        return fail(h, PSDC_ERR_ARG, "AdcDac frames carry four traces: need n_channels >= 4");
    if (n_frames == 0)
        return PSDC_OK;
    if (!d_frames)
        return fail(h, PSDC_ERR_ARG, "null input");
    if (frame_size < 8) // &input[..HEADER_SIZE] panics (src/de/frame.rs:50)
        return fail(h, PSDC_ERR_FRAME_SIZE, "frame shorter than its header");
    const size_t payload = frame_size - 8;
    const int batches = (int)(payload / 64);
    // host-fed samples staged on these channels come first in their streams
    for (int ci = 0; ci < 4; ++ci)
        if (h->ch[ci].fill || h->ch[ci].submitted) {
            rc = flush_all(h);
            if (rc)
                return rc;
            break;
        }
    // Headers are checked on the device (the frames are there) and the Loss counters summed there too, over all frames at
    // first -- the common case has no bad frame -- and again over the accepted ones if there was one.  The scan is ONE small
    // launch on a stream of its own and the host waits for that launch alone: the compute stream keeps working on the rounds
    // of earlier calls meanwhile (their fused launches leave FRAME_RESERVE_BLOCKS workgroup slots free for it), so the verdict
    // -- which frames are ingested is known, and reported, when the call returns, as the reference's per-frame `?` does
    // (src/source.rs:139) -- costs the device no idle time.  Four words come back through pinned memory: {~(first bad frame
    // << 2 | error) or 0, batches received, sequence gaps, first seq | next seq << 32}.
    if (!h->d_scan || !h->h_scan || !h->scan_stream) {
        // built into locals and committed to the handle only when every step has succeeded: a half-built state (accumulators
        // not zeroed, no pinned result words, the null stream) must never reach the verdict launch
        unsigned long long *d_scan = nullptr, *h_scan = nullptr;
        hipStream_t scan_stream = nullptr;
        int lo = 0, hi = 0;
        hipError_t e = hipMalloc(&d_scan, 5 * sizeof(unsigned long long));
        if (e == hipSuccess)
            e = hipMemset(d_scan, 0, 5 * sizeof(unsigned long long));
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void **>(&h_scan), 4 * sizeof(unsigned long long), hipHostMallocDefault);
        if (e == hipSuccess)
            e = hipDeviceGetStreamPriorityRange(&lo, &hi); // (hi = the numerically lowest = greatest priority)
        if (e == hipSuccess)
            e = hipStreamCreateWithPriority(&scan_stream, hipStreamNonBlocking, hi);
        if (e != hipSuccess) {
            if (scan_stream)
                (void)hipStreamDestroy(scan_stream);
            if (h_scan)
                (void)hipHostFree(h_scan);
            if (d_scan)
                (void)hipFree(d_scan);
            HIPCHK(h, e);
        }
        h->d_scan = d_scan;
        h->h_scan = h_scan;
        h->scan_stream = scan_stream;
    }
    const unsigned long long *res = h->h_scan;
    auto scan = [&](size_t n_loss, bool check) -> int {
        HIPCHK(h, launch_adcdac_verdict(d_frames, frame_size, n_frames, batches, payload % 64 == 0, check ? 1 : 0, n_loss, h->d_scan,
                                        h->h_scan, h->scan_stream));
        HIPCHK(h, hipStreamSynchronize(h->scan_stream));
        return PSDC_OK;
    };
    rc = scan(n_frames, true);
    if (rc)
        return rc;
    size_t good = n_frames;
    int bad = PSDC_OK;
    if (res[0] != 0) {
        const unsigned long long key = ~res[0];
        good = (size_t)(key >> 2);
        const int code = (int)(key & 3);
        bad = code == 1 ? PSDC_ERR_FRAME_HEADER : code == 2 ? PSDC_ERR_FRAME_FORMAT : PSDC_ERR_FRAME_SIZE;
        if (good) {
            rc = scan(good, false);
            if (rc)
                return rc;
        }
    }
    if (good) { // Loss::update over the accepted frames (src/loss.rs:11-26)
        h->loss.received += res[1];
        const uint32_t seq0 = (uint32_t)res[3], next = (uint32_t)(res[3] >> 32);
        if (h->loss.have_seq)
            h->loss.dropped += (uint32_t)(seq0 - h->loss.next_seq); // wrapping_sub
        h->loss.dropped += res[2];
        h->loss.next_seq = next;
        h->loss.have_seq = 1;
    }
    if (good && batches > 0) {
        h->idle = false;
        const size_t per_frame = (size_t)batches * 8; // samples per trace and frame
        // (the in-place kernels read wire words with 8-, 4- and 2-byte loads at offsets that are aligned relative to the
        // base only: a base that is not a multiple of 8 takes the byte-wise decode kernel, as the verdict scan does)
        const bool in_place_ok = fused_frames_supported((int)h->n) && fused_window(h) == 1 && (reinterpret_cast<uintptr_t>(d_frames) & 7) == 0 &&
                                 frame_size % 8 == 0;
        // pieces of <= FSPAN_MAX_SAMPLES samples per trace (the kernels' cell arithmetic) / 2^24 on the decode path
        const size_t piece_frames = std::max<size_t>(1, (in_place_ok ? (size_t)FSPAN_MAX_SAMPLES : ((size_t)1 << 24)) / per_frame);
        for (size_t f0 = 0; f0 < good; f0 += piece_frames) {
            const size_t cnt = std::min(piece_frames, good - f0);
            const size_t per_ch = cnt * per_frame;
            const uint8_t *piece = d_frames + f0 * frame_size;
            if (in_place_ok && per_ch >= (size_t)4 * (h->n + HBF_HALO)) {
                // The four traces are read IN PLACE, as wire words, by the stage-0 loads of the fused kernel: a zero-copy span
                // per trace, exactly like psdc_process_device's -- held back while the device is busy so that calls share rounds.
                bool flush = false;
                for (int ci = 0; ci < 4; ++ci) {
                    Channel &c = h->ch[ci];
                    flush = flush || c.submitted || c.fill > 0 || c.spans.size() >= h->coalesce;
                }
                bool any_span = false;
                for (int ci = 0; ci < 4; ++ci)
                    any_span = any_span || h->ch[ci].has_span();
                if (flush || (any_span && device_idle(h))) {
                    rc = flush_all(h);
                    if (rc)
                        return rc;
                }
                FrameSpan fs{};
                fs.frames = piece;
                fs.bytes = (unsigned long long)cnt * frame_size;
                fs.frame_size = (unsigned)frame_size;
                fs.batches = (unsigned)batches;
                fs.magic = batches >= 2 ? (unsigned)((0x100000000ull + (unsigned)batches - 1) / (unsigned)batches) : 0u;
                for (int ci = 0; ci < 4; ++ci) {
                    Channel &c = h->ch[ci];
                    if (c.st.empty()) {
                        rc = add_stage(h, c);
                        if (rc)
                            return rc;
                    }
                    StageState &s0 = c.st[0];
                    DeviceSpan sp;
                    sp.first = s0.total;
                    sp.len = per_ch;
                    sp.fs = fs;
                    sp.fch = ci;
                    c.spans.push_back(sp);
                    s0.total += per_ch;
                    h->idle = false; // (a flush above may have drained the pipeline and marked it idle)
                    c.span_max = std::max(c.span_max, per_ch);
                    if (c.spans.size() > 1)
                        c.coalesced_seen = true;
                }
                // on an idle device nothing is held back; on a busy one the next call may share this one's round
                if (h->ch[0].spans.size() >= h->coalesce || device_idle(h)) {
                    rc = advance(h);
                    if (rc)
                        return rc;
                }
                continue;
            }
            // sizes / windows without an in-place kernel, and pieces too short to split: decoded into the stage-0 stream
            // buffers by a kernel of their own, each piece a round
            bool pend = false;
            for (int ci = 0; ci < 4; ++ci)
                pend = pend || h->ch[ci].has_span();
            if (pend) {
                rc = advance(h);
                if (rc)
                    return rc;
            }
            float *dst[4];
            for (int ci = 0; ci < 4; ++ci) {
                Channel &c = h->ch[ci];
                if (c.st.empty()) {
                    rc = add_stage(h, c);
                    if (rc)
                        return rc;
                }
                StageState &s0 = c.st[0];
                rc = ensure_room(h, s0, s0.total + per_ch);
                if (rc)
                    return rc;
                dst[ci] = s0.buf.p[s0.buf.cur] + (s0.total - s0.buf.base);
            }
            HIPCHK(h, launch_adcdac(piece, frame_size, cnt, batches, dst[0], dst[1], dst[2], dst[3], h->stream));
            h->idle = false;
            for (int ci = 0; ci < 4; ++ci) {
                h->ch[ci].st[0].total += per_ch;
                h->ch[ci].st[0].buf.end = h->ch[ci].st[0].total;
                h->ch[ci].submitted = true;
            }
            rc = advance(h);
            if (rc)
                return rc;
        }
    }
    if (n_ok)
        *n_ok = good;
    if (bad != PSDC_OK)
        return fail(h, bad,
                    bad == PSDC_ERR_FRAME_HEADER   ? "Invalid frame header"
                    : bad == PSDC_ERR_FRAME_FORMAT ? "Unknown or non-AdcDac format ID"
                                                   : "Payload size");
    return PSDC_OK;
}

// psdc_process_frames for frames that already sit in device memory.  The headers (8 of every frame_size bytes) come to the host through
// one small gather kernel and are validated there exactly as ingest_frames_host does; the payloads never leave the device: runs of Fls /
// ThermostatEem / Mpll frames are decoded by payload_kernel straight from the caller's buffer into the stage-0 streams, runs of AdcDac
// frames go through psdc_process_adcdac_frames_device (read in place where a fused kernel exists).
int psdc_process_frames_device(psdc_handle *h, const uint8_t *d_frames, size_t frame_size, size_t n_frames, size_t *n_ok)
{
    if (n_ok)
        *n_ok = 0;
    int rc = check_channel(h, 0);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (n_frames == 0)
        return PSDC_OK;
    if (!d_frames)
        return fail(h, PSDC_ERR_ARG, "null input");
    if (frame_size < 8) // &input[..HEADER_SIZE] panics (src/de/frame.rs:50)
        return fail(h, PSDC_ERR_FRAME_SIZE, "frame shorter than its header");
    // The headers come to the host through ONE small kernel that writes them into pinned memory, on a stream of its own: the host
    // waits for that launch alone while the compute stream keeps working on earlier calls (a strided hipMemcpy2D of 70 000 headers
    // took ~0.25 ms of a 0.39 ms call: Mpll frames 32 -> 84 GS/s, tools/bench_frames.py).
    if (!h->hdr_stream) {
        hipStream_t st = nullptr;
        HIPCHK(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->hdr_stream = st;
    }
    if (h->h_hdr_cap < 8 * n_frames) {
        const size_t cap = std::max<size_t>(8 * n_frames + (8 * n_frames) / 2, (size_t)1 << 16);
        uint8_t *nb = nullptr;
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&nb), cap, hipHostMallocDefault));
        if (h->h_hdr)
            (void)hipHostFree(h->h_hdr);
        h->h_hdr = nb;
        h->h_hdr_cap = cap;
    }
    HIPCHK(h, launch_header_gather(d_frames, frame_size, n_frames, h->h_hdr, h->hdr_stream));
    HIPCHK(h, hipStreamSynchronize(h->hdr_stream));
    struct HdrView { // (hdr.data() / hdr[i] as the vector this replaced)
        const uint8_t *p;
        const uint8_t *data() const { return p; }
        uint8_t operator[](size_t i) const { return p[i]; }
    } hdr{h->h_hdr};
    const size_t payload = frame_size - 8;
    size_t good = 0, f0 = 0;
    int bad = PSDC_OK;
    auto done = [&](int code) {
        if (n_ok)
            *n_ok = good;
        return code == PSDC_OK ? PSDC_OK
                               : fail(h, code,
                                      code == PSDC_ERR_FRAME_HEADER   ? "Invalid frame header"
                                      : code == PSDC_ERR_FRAME_FORMAT ? "Unknown format ID"
                                                                      : "Payload size");
    };
    while (f0 < n_frames) {
        const uint8_t *first = hdr.data() + 8 * f0;
        if (first[0] != 0x7b || first[1] != 0x05)
            return done(PSDC_ERR_FRAME_HEADER);
        const WireFmt *wf = wire_fmt(first[2]);
        if (!wf)
            return done(PSDC_ERR_FRAME_FORMAT);
        if ((int)h->n_channels < wf->ntr) {
            if (n_ok)
                *n_ok = good;
            return fail(h, PSDC_ERR_ARG, "the frames carry more traces than the handle has channels");
        }
        if (wf->id == 1) { // a run of AdcDac frames: its own entry point checks them (and counts their Loss) on the device
            size_t run = 1;
            while (f0 + run < n_frames && hdr[8 * (f0 + run)] == 0x7b && hdr[8 * (f0 + run) + 1] == 0x05 && hdr[8 * (f0 + run) + 2] == 1)
                ++run;
            size_t ok = 0;
            rc = psdc_process_adcdac_frames_device(h, d_frames + f0 * frame_size, frame_size, run, &ok);
            good += ok;
            if (rc) {
                if (n_ok)
                    *n_ok = good;
                return rc;
            }
            f0 += run;
            continue;
        }
        const int ntr = wf->ntr;
        const int batches = (int)(payload / wf->batch_bytes);
        // order behind anything pending on these channels (held spans, host-fed samples)
        bool pend = false;
        for (int ci = 0; ci < ntr; ++ci)
            pend = pend || h->ch[ci].has_span() || h->ch[ci].submitted || h->ch[ci].fill;
        if (pend) {
            rc = flush_all(h);
            if (rc)
                return rc;
        }
        // pieces of ~2^22 samples per trace: a stage-0 stream buffer never grows by more than that at once
        const size_t piece_frames = std::max<size_t>(1, ((size_t)1 << 22) / (size_t)std::max(1, batches));
        bool run_end = false;
        while (f0 < n_frames && bad == PSDC_OK && !run_end) {
            size_t cnt = 0;
            const size_t lim = std::min(piece_frames, n_frames - f0);
            for (; cnt < lim; ++cnt) { // Header::parse + the payload's size checks + Loss::update, as ingest_frames_host's scan
                const uint8_t *f = hdr.data() + 8 * (f0 + cnt);
                if (f[0] != 0x7b || f[1] != 0x05) {
                    bad = PSDC_ERR_FRAME_HEADER;
                    break;
                }
                if (f[2] != wf->id) {
                    if (wire_fmt(f[2]))
                        run_end = true;
                    else
                        bad = PSDC_ERR_FRAME_FORMAT;
                    break;
                }
                if (payload % wf->batch_bytes != 0 || (int)f[3] != batches) {
                    bad = PSDC_ERR_FRAME_SIZE;
                    break;
                }
                const uint32_t seq = (uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24);
                h->loss.received += f[3];
                if (h->loss.have_seq)
                    h->loss.dropped += (uint32_t)(seq - h->loss.next_seq); // wrapping_sub
                h->loss.next_seq = seq + f[3];                              // wrapping_add
                h->loss.have_seq = 1;
            }
            good += cnt;
            if (cnt == 0)
                break;
            if (batches > 0) {
                const size_t per_ch = cnt * (size_t)batches;
                float *dst[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int ci = 0; ci < ntr; ++ci) {
                    Channel &c = h->ch[ci];
                    if (c.st.empty()) {
                        rc = add_stage(h, c);
                        if (rc)
                            return rc;
                    }
                    StageState &s0 = c.st[0];
                    rc = ensure_room(h, s0, s0.total + per_ch);
                    if (rc)
                        return rc;
                    dst[ci] = s0.buf.p[s0.buf.cur] + (s0.total - s0.buf.base);
                }
                HIPCHK(h, launch_payload(wf->id, d_frames + f0 * frame_size, frame_size, cnt, batches, dst[0], dst[1], dst[2], dst[3], h->stream));
                h->idle = false;
                for (int ci = 0; ci < ntr; ++ci) {
                    h->ch[ci].st[0].total += per_ch;
                    h->ch[ci].st[0].buf.end = h->ch[ci].st[0].total;
                    h->ch[ci].submitted = true;
                }
                rc = advance(h);
                if (rc)
                    return rc;
            }
            f0 += cnt;
        }
        if (bad != PSDC_OK)
            return done(bad);
    }
    return done(PSDC_OK);
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

int psdc_num_stages(psdc_handle *h, uint32_t channel)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    rc = flush_all(h);
    if (rc)
        return rc;
    return (int)h->ch[channel].st.size();
}

int psdc_stage_info(psdc_handle *h, uint32_t channel, uint32_t stage, psdc_stage_stat *out)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (!out)
        return fail(h, PSDC_ERR_ARG, "null output");
    rc = flush_all(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    if (stage >= c.st.size())
        return fail(h, PSDC_ERR_ARG, "stage out of range");
    const StageState &s = c.st[stage];
    out->count = s.count;
    out->avg = cur_stage_avg(h, stage);
    out->pending = pending_for(h->geo, s.total);
    const uint32_t cm1 = s.count ? s.count - 1 : 0;
    out->processed = (uint64_t)h->n * s.count - (uint64_t)h->geo.overlap * cm1;
    return PSDC_OK;
}

int psdc_stage_spectrum(psdc_handle *h, uint32_t channel, uint32_t stage, float *out)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (!out)
        return fail(h, PSDC_ERR_ARG, "null output");
    rc = flush_sync(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    if (stage >= c.st.size())
        return fail(h, PSDC_ERR_ARG, "stage out of range");
    return read_back(h, out, c.st[stage].spectrum, h->n / 2 + 1);
}

int psdc_stage_gain(psdc_handle *h, uint32_t channel, uint32_t stage, float *out)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    if (!out)
        return fail(h, PSDC_ERR_ARG, "null output");
    rc = flush_all(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    if (stage >= c.st.size())
        return fail(h, PSDC_ERR_ARG, "stage out of range");
    *out = stage_gain(h->n, c.st[stage].count64, h->nenbw, h->power);
    return PSDC_OK;
}

int psdc_stage_buf(psdc_handle *h, uint32_t channel, uint32_t stage, float *out, size_t cap, size_t *len)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    rc = flush_sync(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    if (stage >= c.st.size())
        return fail(h, PSDC_ERR_ARG, "stage out of range");
    const StageState &s = c.st[stage];
    const uint64_t pend = pending_for(h->geo, s.total);
    if (len)
        *len = (size_t)pend;
    if (!out)
        return PSDC_OK;
    if (cap < pend)
        return fail(h, PSDC_ERR_CAPACITY, "output too small");
    if (pend) {
        const uint64_t from = s.total - pend;
        rc = read_back(h, out, s.buf.p[s.buf.cur] + (from - s.buf.base), (size_t)pend);
        if (rc)
            return rc;
    }
    return PSDC_OK;
}

int psdc_read_channel(psdc_handle *h, uint32_t channel, uint32_t cap, uint32_t *n_stages,
                      psdc_stage_stat *stats, float *spectra)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    rc = spectra ? flush_sync(h) : flush_all(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    const uint32_t ns = (uint32_t)c.st.size();
    if (n_stages)
        *n_stages = ns;
    if ((stats || spectra) && cap < ns)
        return fail(h, PSDC_ERR_CAPACITY, "psdc_read_channel: output too small");
    if (stats)
        for (uint32_t k = 0; k < ns; ++k) {
            const StageState &s = c.st[k];
            stats[k].count = s.count;
            stats[k].avg = cur_stage_avg(h, k);
            stats[k].pending = pending_for(h->geo, s.total);
            const uint32_t cm1 = s.count ? s.count - 1 : 0;
            stats[k].processed = (uint64_t)h->n * s.count - (uint64_t)h->geo.overlap * cm1;
        }
    if (spectra && ns) { // the channel's accumulators are consecutive rows of one slab: one copy
        HIPCHK(h, launch_copy_out(h->h_read, c.st[0].spectrum, (size_t)ns * h->n, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const size_t bins = h->n / 2 + 1;
        for (uint32_t k = 0; k < ns; ++k)
            memcpy(spectra + k * bins, h->h_read + (size_t)k * h->n, sizeof(float) * bins);
    }
    return PSDC_OK;
}

int psdc_psd(psdc_handle *h, uint32_t channel, int keep_overlap, uint32_t min_count,
             int keep_transition_band, float *psd_out, size_t psd_cap, size_t *psd_len,
             psdc_break *breaks, size_t breaks_cap, size_t *n_breaks)
{
    int rc = check_channel(h, channel);
    if (rc)
        return rc;
    ON_DEVICE(h, h->device);
    rc = flush_sync(h);
    if (rc)
        return rc;
    Channel &c = h->ch[channel];
    const size_t ns = c.st.size();
    const size_t bins = h->n / 2 + 1;
    std::vector<uint32_t> counts(ns), avgs(ns);
    std::vector<uint64_t> pend(ns), counts64(ns);
    std::vector<float> spectra(psd_out ? ns * bins : 0);
    for (size_t i = 0; i < ns; ++i) {
        counts[i] = c.st[i].count;
        counts64[i] = c.st[i].count64;
        avgs[i] = cur_stage_avg(h, i);
        pend[i] = pending_for(h->geo, c.st[i].total);
    }
    if (psd_out && ns) {
        HIPCHK(h, launch_copy_out(h->h_read, c.st[0].spectrum, ns * h->n, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < ns; ++i)
            memcpy(spectra.data() + i * bins, h->h_read + i * h->n, sizeof(float) * bins);
    }
    rc = stitch_impl(h->n, h->nenbw, h->power, h->geo.overlap, (uint32_t)ns, counts.data(), avgs.data(),
                     pend.data(), spectra.data(), keep_overlap, min_count, keep_transition_band, psd_out,
                     psd_cap, psd_len, breaks, breaks_cap, n_breaks, counts64.data());
    if (rc)
        return fail(h, rc, "psdc_psd: output too small");
    return PSDC_OK;
}

float psdc_rbw(const psdc_handle *h)
{
    // (1 << DEPTH) as f32 / (N as f32 * HBF_PASSBAND) (src/psd.rs:427-429)
    return h ? 8.0f / ((float)h->n * 0.4f) : 0.0f;
}

psdc_handle *psdc_clone(psdc_handle *h)
{
    if (!h) {
        fail(nullptr, PSDC_ERR_ARG, "null handle");
        return nullptr;
    }
    DevScope dev_scope_(h->device);
    if (dev_scope_.err != hipSuccess || flush_sync(h) != PSDC_OK)
        return nullptr;
    psdc_handle *o = create_impl(h->n, h->window_kind, h->win_host.data(),
                                 WindowConsts{h->nenbw, h->power, h->geo.overlap}, h->n_channels, h->device);
    if (!o)
        return nullptr;
    o->detrend = h->detrend;
    o->avg_limit = h->avg_limit;
    o->avg_count = h->avg_count;
    o->quantum = h->quantum;
    o->profile = h->profile;
    o->coalesce = h->coalesce;
    o->coalesce_auto = h->coalesce_auto;
    o->coalesce_always = h->coalesce_always;
    o->stage_limit = h->stage_limit;
    o->min_pairs = h->min_pairs;
    auto bad = [&](const char *what) -> psdc_handle * {
        fail(nullptr, PSDC_ERR_DEVICE, std::string("psdc_clone: ") + what);
        psdc_destroy(o);
        return nullptr;
    };
    for (uint32_t ci = 0; ci < h->n_channels; ++ci) {
        for (const StageState &s : h->ch[ci].st) {
            if (add_stage(o, o->ch[ci]) != PSDC_OK)
                return bad("alloc");
            StageState &d = o->ch[ci].st.back();
            d.total = s.total;
            d.segs = s.segs;
            d.dec = s.dec;
            d.count = s.count;
            d.count64 = s.count64;
            d.sink_pos = s.sink_pos;
            d.buf.base = s.buf.base;
            d.buf.end = s.buf.base; // nothing resident yet
            if (ensure_room(o, d, s.total) != PSDC_OK)
                return bad("alloc");
            if (hipMemcpyAsync(d.spectrum, s.spectrum, sizeof(float) * h->n, hipMemcpyDeviceToDevice,
                               o->stream) != hipSuccess)
                return bad("copy");
            const size_t have = (size_t)(s.total - s.buf.base);
            if (have && hipMemcpyAsync(d.buf.p[d.buf.cur], s.buf.p[s.buf.cur], sizeof(float) * have,
                                       hipMemcpyDeviceToDevice, o->stream) != hipSuccess)
                return bad("copy");
            d.buf.end = s.total;
        }
    }
    if (hipStreamSynchronize(o->stream) != hipSuccess)
        return bad("sync");
    return o;
}

size_t psdc_frequencies(const psdc_break *b, size_t n, float *out, size_t cap)
{
    // Break::frequencies (src/psd.rs:315-327)
    size_t len = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!b[i].include)
            continue;
        const float rbw = 1.0f / (float)(b[i].fft_size * b[i].decimation); // :334-336
        for (uint64_t f = b[i].bins_start; f < b[i].bins_end; ++f) {
            if (out && len < cap)
                out[len] = (float)f * rbw;
            ++len;
        }
    }
    return len;
}

int psdc_hbf_response_length(int depth)
{
    if (depth < 0 || depth > 3)
        return PSDC_ERR_ARG;
    return hbf_response_length(depth);
}

int psdc_stitch(uint32_t n, int window_kind, uint32_t n_stages, const uint32_t *counts,
                const uint32_t *avgs, const uint64_t *pendings, const float *spectra, int keep_overlap,
                uint32_t min_count, int keep_transition_band, float *psd_out, size_t psd_cap,
                size_t *psd_len, psdc_break *breaks, size_t breaks_cap, size_t *n_breaks)
{
    WindowConsts wc{};
    if (n < 2 || !window_consts(n, window_kind, &wc) || n_stages > 20)
        return fail(nullptr, PSDC_ERR_ARG, "psdc_stitch: bad arguments");
    if (n_stages && (!counts || !avgs || !pendings || (psd_out && !spectra)))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_stitch: null input");
    int rc = stitch_impl(n, wc.nenbw, wc.power, wc.overlap, n_stages, counts, avgs, pendings, spectra,
                         keep_overlap, min_count, keep_transition_band, psd_out, psd_cap, psd_len, breaks,
                         breaks_cap, n_breaks);
    if (rc)
        return fail(nullptr, rc, "psdc_stitch: output too small");
    return PSDC_OK;
}

int psdc_stitch_window(uint32_t n, float power, float nenbw, size_t overlap, uint32_t n_stages,
                       const uint64_t *counts64, const uint32_t *avgs, const uint64_t *pendings, const float *spectra,
                       int keep_overlap, uint32_t min_count, int keep_transition_band, float *psd_out, size_t psd_cap,
                       size_t *psd_len, psdc_break *breaks, size_t breaks_cap, size_t *n_breaks)
{
    if (n < 2 || overlap >= n || n_stages > 20)
        return fail(nullptr, PSDC_ERR_ARG, "psdc_stitch_window: bad arguments");
    if (n_stages && (!counts64 || !avgs || !pendings || (psd_out && !spectra)))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_stitch_window: null input");
    uint32_t counts[20];
    for (uint32_t i = 0; i < n_stages; ++i)
        counts[i] = count_report(counts64[i]);
    int rc = stitch_impl(n, nenbw, power, (uint32_t)overlap, n_stages, counts, avgs, pendings, spectra, keep_overlap,
                         min_count, keep_transition_band, psd_out, psd_cap, psd_len, breaks, breaks_cap, n_breaks, counts64);
    if (rc)
        return fail(nullptr, rc, "psdc_stitch_window: output too small");
    return PSDC_OK;
}

// ---- packed read-out (include/psdcascade.h) ------------------------------------------------------
// Layout (native endian, 8-byte aligned throughout):
//   header   { u32 magic 'PSDR', u32 version, u32 n, u32 n_channels, f32 power, f32 nenbw, u32 overlap, u32 window_kind }
//   channel  { u32 n_stages, u32 pad, stage[MAX_STAGES] { u64 count64, u64 pending, u32 avg, u32 pad },
//              f32 spectra[MAX_STAGES][n/2 + 1 (+1 if even, to keep 8-byte alignment)] }   x n_channels
} // extern "C"

namespace {

constexpr uint32_t PACK_MAGIC = 0x52445350u, PACK_VERSION = 1;
struct PackHeader {
    uint32_t magic, version, n, n_channels;
    float power, nenbw;
    uint32_t overlap, window_kind;
};
struct PackStage {
    uint64_t count64, pending;
    uint32_t avg, pad;
};
size_t pack_row_floats(uint32_t n) { return ((size_t)n / 2 + 1 + 1) & ~(size_t)1; }
size_t pack_channel_bytes(uint32_t n) { return 8 + sizeof(PackStage) * MAX_STAGES + sizeof(float) * MAX_STAGES * pack_row_floats(n); }

// header + bounds of a record; nullptr (and the error recorded) if it is not one.  A record is documented to arrive over ANY
// transport, so nothing in it is trusted: every field is held to the range the library itself can produce BEFORE it enters a size
// computation (n a supported FFT size -- psdc_pack_init also admits the small powers of two the host-only tests use --,
// n_channels <= 4096 as in psdc_create, overlap < n), and the length test is a division, which cannot wrap.
constexpr uint32_t PACK_MAX_N = (uint32_t)BIGFFT_MAX_N, PACK_MAX_CHANNELS = 4096;
bool pack_dims_ok(uint32_t n, uint32_t n_channels, uint64_t overlap)
{
    return n >= 2 && n <= PACK_MAX_N && n_channels <= PACK_MAX_CHANNELS && overlap < n;
}
const PackHeader *pack_check(const void *buf, size_t len, uint32_t channel)
{
    const PackHeader *hd = static_cast<const PackHeader *>(buf);
    if (!buf || len < sizeof(PackHeader) || hd->magic != PACK_MAGIC || hd->version != PACK_VERSION ||
        !pack_dims_ok(hd->n, hd->n_channels, hd->overlap) || !(hd->power > 0.0f) || !(hd->nenbw > 0.0f) ||
        (hd->n_channels && (len - sizeof(PackHeader)) / pack_channel_bytes(hd->n) < hd->n_channels)) {
        fail(nullptr, PSDC_ERR_ARG, "not a packed read-out (psdc_pack_readout), truncated, or fields out of range");
        return nullptr;
    }
    if (channel >= hd->n_channels) {
        fail(nullptr, PSDC_ERR_ARG, "channel out of range of the packed read-out");
        return nullptr;
    }
    return hd;
}

} // namespace

extern "C" {

size_t psdc_readout_bytes(uint32_t n, uint32_t n_channels)
{
    if (!pack_dims_ok(n, n_channels, 0)) // (0: no record of such dimensions exists)
        return 0;
    return sizeof(PackHeader) + (size_t)n_channels * pack_channel_bytes(n);
}

int psdc_pack_init(void *buf, size_t cap, uint32_t n, float power, float nenbw, size_t overlap, uint32_t n_channels)
{
    if (!buf || !pack_dims_ok(n, n_channels, overlap) || !(power > 0.0f) || !(nenbw > 0.0f) || cap < psdc_readout_bytes(n, n_channels))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_pack_init: bad arguments or buffer too small (psdc_readout_bytes)");
    memset(buf, 0, psdc_readout_bytes(n, n_channels));
    const PackHeader hd{PACK_MAGIC, PACK_VERSION, n, n_channels, power, nenbw, (uint32_t)overlap, 0};
    memcpy(buf, &hd, sizeof(hd));
    return PSDC_OK;
}

int psdc_pack_channel(void *buf, size_t len, uint32_t channel, uint32_t n_stages, const uint64_t *counts64,
                      const uint32_t *avgs, const uint64_t *pendings, const float *spectra)
{
    const PackHeader *hd = pack_check(buf, len, channel);
    if (!hd)
        return PSDC_ERR_ARG;
    if (n_stages > MAX_STAGES || (n_stages && (!counts64 || !avgs || !pendings || !spectra)))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_pack_channel: bad arguments");
    char *p = static_cast<char *>(buf) + sizeof(PackHeader) + (size_t)channel * pack_channel_bytes(hd->n);
    memset(p, 0, pack_channel_bytes(hd->n));
    memcpy(p, &n_stages, sizeof(n_stages));
    PackStage *ps = reinterpret_cast<PackStage *>(p + 8);
    float *sp = reinterpret_cast<float *>(p + 8 + sizeof(PackStage) * MAX_STAGES);
    const size_t bins = hd->n / 2 + 1, row = pack_row_floats(hd->n);
    for (uint32_t k = 0; k < n_stages; ++k) {
        ps[k] = {counts64[k], pendings[k], avgs[k], 0};
        memcpy(sp + k * row, spectra + k * bins, sizeof(float) * bins);
    }
    return PSDC_OK;
}

int psdc_pack_readout(psdc_handle *h, void *buf, size_t cap, size_t *len)
{
    if (!h)
        return fail(nullptr, PSDC_ERR_ARG, "null handle");
    const size_t need = psdc_readout_bytes(h->n, h->n_channels);
    if (len)
        *len = need;
    if (!buf)
        return PSDC_OK; // size query
    if (cap < need)
        return fail(h, PSDC_ERR_CAPACITY, "psdc_pack_readout: buffer too small (psdc_readout_bytes)");
    ON_DEVICE(h, h->device);
    int rc = flush_sync(h);
    if (rc)
        return rc;
    memset(buf, 0, need);
    char *p = static_cast<char *>(buf);
    PackHeader hd{PACK_MAGIC, PACK_VERSION, h->n, h->n_channels, h->power, h->nenbw, h->geo.overlap, (uint32_t)h->window_kind};
    memcpy(p, &hd, sizeof(hd));
    p += sizeof(hd);
    const size_t bins = h->n / 2 + 1, row = pack_row_floats(h->n);
    for (uint32_t ci = 0; ci < h->n_channels; ++ci, p += pack_channel_bytes(h->n)) {
        Channel &c = h->ch[ci];
        const uint32_t ns = (uint32_t)c.st.size();
        memcpy(p, &ns, sizeof(ns));
        PackStage *ps = reinterpret_cast<PackStage *>(p + 8);
        float *sp = reinterpret_cast<float *>(p + 8 + sizeof(PackStage) * MAX_STAGES);
        for (uint32_t k = 0; k < ns; ++k)
            ps[k] = {c.st[k].count64, pending_for(h->geo, c.st[k].total), cur_stage_avg(h, k), 0};
        if (ns) { // the channel's accumulators are consecutive rows of one slab: one copy
            HIPCHK(h, launch_copy_out(h->h_read, c.st[0].spectrum, (size_t)ns * h->n, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            for (uint32_t k = 0; k < ns; ++k)
                memcpy(sp + k * row, h->h_read + (size_t)k * h->n, sizeof(float) * bins);
        }
    }
    return PSDC_OK;
}

int psdc_pack_pad(const void *rec, size_t len, void *out, size_t cap, uint32_t n_channels)
{
    const PackHeader *hd = static_cast<const PackHeader *>(rec);
    if (hd && len >= sizeof(PackHeader) && hd->n_channels == 0 && hd->magic == PACK_MAGIC && hd->version == PACK_VERSION &&
        pack_dims_ok(hd->n, 0, hd->overlap))
        ; // (a record of no channels -- a rank that owns none -- pads like any other)
    else if (!pack_check(rec, len, 0))
        return PSDC_ERR_ARG;
    const size_t own = psdc_readout_bytes(hd->n, hd->n_channels), need = psdc_readout_bytes(hd->n, n_channels);
    if (!out || out == rec || n_channels < hd->n_channels || need == 0 || cap < need)
        return fail(nullptr, PSDC_ERR_ARG, "psdc_pack_pad: fewer channels than the record holds, or buffer too small (psdc_readout_bytes)");
    memcpy(out, rec, own);
    memset(static_cast<char *>(out) + own, 0, need - own); // empty channels: no stages
    PackHeader nh = *hd;
    nh.n_channels = n_channels;
    memcpy(out, &nh, sizeof(nh));
    return PSDC_OK;
}

int psdc_unpack_info(const void *buf, size_t len, uint32_t channel, uint32_t *n, uint32_t *n_channels, uint32_t *n_stages)
{
    const PackHeader *hd = pack_check(buf, len, channel);
    if (!hd)
        return PSDC_ERR_ARG;
    if (n)
        *n = hd->n;
    if (n_channels)
        *n_channels = hd->n_channels;
    if (n_stages)
        memcpy(n_stages, static_cast<const char *>(buf) + sizeof(PackHeader) + (size_t)channel * pack_channel_bytes(hd->n), 4);
    return PSDC_OK;
}

int psdc_unpack_stitch(const void *buf, size_t len, uint32_t channel, int keep_overlap, uint32_t min_count,
                       int keep_transition_band, float *psd_out, size_t psd_cap, size_t *psd_len, psdc_break *breaks,
                       size_t breaks_cap, size_t *n_breaks)
{
    const PackHeader *hd = pack_check(buf, len, channel);
    if (!hd)
        return PSDC_ERR_ARG;
    const char *p = static_cast<const char *>(buf) + sizeof(PackHeader) + (size_t)channel * pack_channel_bytes(hd->n);
    uint32_t ns = 0;
    memcpy(&ns, p, 4);
    if (ns > MAX_STAGES)
        return fail(nullptr, PSDC_ERR_ARG, "packed read-out: stage count out of range");
    const PackStage *ps = reinterpret_cast<const PackStage *>(p + 8);
    const float *sp = reinterpret_cast<const float *>(p + 8 + sizeof(PackStage) * MAX_STAGES);
    const size_t bins = hd->n / 2 + 1, row = pack_row_floats(hd->n);
    uint32_t counts[MAX_STAGES], avgs[MAX_STAGES];
    uint64_t counts64[MAX_STAGES], pend[MAX_STAGES];
    std::vector<float> spectra;
    try { // (nothing unwinds across the ABI; ns <= 16 and bins <= 8193 here, so this is 512 KiB at most)
        spectra.resize((size_t)ns * bins);
    } catch (const std::bad_alloc &) {
        return fail(nullptr, PSDC_ERR_NOMEM, "psdc_unpack_stitch: out of memory");
    }
    for (uint32_t k = 0; k < ns; ++k) {
        counts64[k] = ps[k].count64;
        counts[k] = count_report(ps[k].count64);
        avgs[k] = ps[k].avg;
        pend[k] = ps[k].pending;
        memcpy(spectra.data() + k * bins, sp + k * row, sizeof(float) * bins);
    }
    int rc = stitch_impl(hd->n, hd->nenbw, hd->power, hd->overlap, ns, counts, avgs, pend, spectra.data(), keep_overlap,
                         min_count, keep_transition_band, psd_out, psd_cap, psd_len, breaks, breaks_cap, n_breaks, counts64);
    if (rc)
        return fail(nullptr, rc, "psdc_unpack_stitch: output too small");
    return PSDC_OK;
}

int psdc_plan_counts(uint32_t n, int window_kind, uint64_t total, uint32_t cap, uint64_t *received,
                     uint64_t *segments, uint64_t *pending)
{
    WindowConsts wc{};
    if (n < 2 || !window_consts(n, window_kind, &wc) || (n - wc.overlap) % 8 != 0)
        return fail(nullptr, PSDC_ERR_ARG, "psdc_plan_counts: bad arguments");
    Geometry g;
    g.n = n;
    g.overlap = wc.overlap;
    g.hop = n - wc.overlap;
    g.drain = (uint32_t)HBF_DRAIN;
    int k = 0;
    uint64_t t = total;
    while (t > 0 && k < 64) {
        const uint64_t j = segments_for(g, t);
        if ((uint32_t)k < cap) {
            if (received)
                received[k] = t;
            if (segments)
                segments[k] = j;
            if (pending)
                pending[k] = pending_for(g, t);
        }
        ++k;
        t = emitted_for(g, decimated_prefix(g, j));
    }
    return k;
}

float psdc_var_eval(int x_exp, int sinx_exp, float clip, size_t dc_cut, const float *phase_psd,
                    const float *frequencies, size_t n, float tau)
{
    // Var::eval (src/var.rs:26-45); powi = repeated multiplication
    auto powi = [](float x, int e) {
        const bool neg = e < 0;
        unsigned u = (unsigned)(neg ? -e : e);
        float r = 1.0f, b = x;
        while (u) {
            if (u & 1u)
                r *= b;
            b *= b;
            u >>= 1;
        }
        return neg ? 1.0f / r : r;
    };
    const float pi = 3.14159265358979323846f;
    float accu = 0.0f, a0 = 0.0f, f0 = 0.0f;
    for (size_t i = dc_cut; i < n; ++i) {
        const float f = frequencies[i], sp = phase_psd[i];
        if (!(f <= clip / tau))
            break;
        const float sy = sp * f * f;
        const float pft = pi * (f * tau);
        const float hahd = powi(sinf(pft), sinx_exp) * powi(pft, x_exp);
        const float a = sy * hahd;
        accu = accu + (a + a0) * (f - f0);
        a0 = a;
        f0 = f;
    }
    return accu;
}

int psdc_trace_plot(const float *psd, const float *frequencies, size_t n, float fs, int integrate,
                    float integral_start, float integral_end, float *rms, double *plot_xy, size_t plot_cap,
                    size_t *n_points)
{
    // Trace::plot (src/bin/psd.rs:125-157) with Trapezoidal (src/bin/psd.rs:98-116), all in f32
    if (n && (!psd || !frequencies))
        return fail(nullptr, PSDC_ERR_ARG, "psdc_trace_plot: null input");
    const float logfs = log10f(fs);                 // :127
    float tx = 0.0f, ty = 0.0f, ti = 0.0f;          // Trapezoidal::default()
    float pi = 0.0f;                                // :129
    size_t np = 0;
    bool overflow = false;
    for (size_t k = 0; k < n; ++k) {
        const float p = psd[k], f = frequencies[k];
        const float di = (p + ty) * 0.5f * (f - tx); // Trapezoidal::push :105-110
        tx = f;
        ty = p;
        ti += di;
        const float hz = fs * f;
        if (hz >= integral_start && hz <= integral_end) // RangeInclusive::contains :137
            pi += di;
        if (std::fpclassify(f) == FP_NORMAL) { // f32::is_normal :141
            if (plot_xy) {
                if (np < plot_cap) {
                    plot_xy[2 * np] = (double)(log10f(f) + logfs);
                    plot_xy[2 * np + 1] = (double)(integrate ? sqrtf(ti) : 10.0f * (log10f(p) - logfs));
                } else {
                    overflow = true;
                }
            }
            ++np;
        }
    }
    if (rms)
        *rms = sqrtf(pi); // :156
    if (n_points)
        *n_points = np;
    return overflow ? fail(nullptr, PSDC_ERR_CAPACITY, "psdc_trace_plot: plot output too small") : PSDC_OK;
}

int psdc_hbf_dec8(int device, const float *x, size_t len, float *y)
{
    const size_t nout = len / 8;
    if (nout == 0)
        return PSDC_OK;
    if (!x || !y)
        return fail(nullptr, PSDC_ERR_ARG, "null argument");
    psdc_handle *h = nullptr;
    ON_DEVICE(h, device);
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(h, hipMalloc(&dx, sizeof(float) * nout * 8));
    HIPCHK(h, hipMalloc(&dy, sizeof(float) * nout));
    HIPCHK(h, hipMemcpy(dx, x, sizeof(float) * nout * 8, hipMemcpyHostToDevice));
    size_t done = 0;
    while (done < nout) {
        DecBatch db{};
        db.drain = 0;
        const size_t chunk = std::min<size_t>(nout - done, (size_t)1 << 24);
        DecJob &dj = db.jobs[db.njobs++];
        dj.src = dx;
        dj.src_base = 0;
        dj.m0 = (long long)done;
        dj.dst = dy;
        dj.dst_base = 0;
        dj.nout = (int)chunk;
        dj.tile_begin = 0;
        db.ntiles = (int)((chunk + DEC_TILE - 1) / DEC_TILE);
        HIPCHK(h, launch_dec(db, nullptr));
        done += chunk;
    }
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(y, dy, sizeof(float) * nout, hipMemcpyDeviceToHost));
    HIPCHK(h, hipFree(dx));
    HIPCHK(h, hipFree(dy));
    return PSDC_OK;
}

int psdc_fill_noise_device(int device, float *d_x, size_t len, uint64_t seed, uint64_t first_index)
{
    psdc_handle *h = nullptr;
    ON_DEVICE(h, device);
    HIPCHK(h, launch_fill_noise(d_x, len, seed, first_index, nullptr));
    HIPCHK(h, hipStreamSynchronize(nullptr));
    return PSDC_OK;
}

// ---- Psd<N>: one stage (src/psd.rs:122-288) --------------------------------------------------
// A one-channel handle whose stage 0 is the Psd and whose stage 1 is a SINK: the decimated stream that
// PsdStage::process returns in `y` (src/psd.rs:246-268) lands in its stream buffer and is handed to the
// caller instead of being analysed.  Same kernels, same bookkeeping as the cascade.
} // extern "C"

struct psdc_stage {
    psdc_handle *h = nullptr;
};

extern "C" {

psdc_stage *psdc_stage_create(uint32_t n, int window_kind, int device)
{
    psdc_handle *h = psdc_create(n, window_kind, 1, device);
    if (!h)
        return nullptr;
    h->stage_limit = 1;
    psdc_stage *st = new (std::nothrow) psdc_stage();
    if (!st) {
        psdc_destroy(h);
        fail(nullptr, PSDC_ERR_NOMEM, "psdc_stage_create: out of memory");
        return nullptr;
    }
    st->h = h;
    return st;
}

psdc_stage *psdc_stage_create_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap, int device)
{
    psdc_handle *h = psdc_create_window(n, win, power, nenbw, overlap, 1, device);
    if (!h)
        return nullptr;
    h->stage_limit = 1;
    psdc_stage *st = new (std::nothrow) psdc_stage();
    if (!st) {
        psdc_destroy(h);
        fail(nullptr, PSDC_ERR_NOMEM, "psdc_stage_create_window: out of memory");
        return nullptr;
    }
    st->h = h;
    return st;
}

void psdc_stage_destroy(psdc_stage *st)
{
    if (!st)
        return;
    psdc_destroy(st->h);
    delete st;
}

psdc_stage *psdc_stage_clone(psdc_stage *st)
{
    if (!st) {
        fail(nullptr, PSDC_ERR_ARG, "null stage");
        return nullptr;
    }
    psdc_handle *o = psdc_clone(st->h);
    if (!o)
        return nullptr;
    psdc_stage *c = new (std::nothrow) psdc_stage();
    if (!c) {
        psdc_destroy(o);
        return nullptr;
    }
    c->h = o;
    return c;
}

const char *psdc_stage_last_error(const psdc_stage *st) { return psdc_last_error(st ? st->h : nullptr); }

int psdc_stage_set_avg(psdc_stage *st, uint32_t avg)
{
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    return psdc_set_avg(st->h, avg, avg); // stage 0 uses min(count >> 0, limit) = avg (src/psd.rs:154-156)
}

int psdc_stage_set_detrend(psdc_stage *st, int detrend_kind)
{
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    return psdc_set_detrend(st->h, detrend_kind);
}

int psdc_stage_process(psdc_stage *st, const float *x, size_t len, float *y, size_t cap, size_t *n_out)
{
    if (n_out)
        *n_out = 0;
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    psdc_handle *h = st->h;
    ON_DEVICE(h, h->device);
    int rc = psdc_process(h, 0, x, len);
    if (rc)
        return rc;
    rc = flush_sync(h); // every segment these samples complete is issued and decimated now (src/psd.rs:199-267)
    if (rc)
        return rc;
    Channel &c = h->ch[0];
    if (c.st.size() < 2)
        return PSDC_OK; // nothing emitted yet (still inside the first segment or the drain)
    StageState &sk = c.st[1];
    const uint64_t avail = sk.total - sk.sink_pos;
    if (avail == 0)
        return PSDC_OK;
    if (!y || cap < avail) // the reference indexes y[n..][..xb.len()] and panics (src/psd.rs:253)
        return fail(h, PSDC_ERR_CAPACITY, "psdc_stage_process: y too small (needs x.len()/8 + n/8 items, src/psd.rs:187-190)");
    rc = read_back(h, y, sk.buf.p[sk.buf.cur] + (sk.sink_pos - sk.buf.base), (size_t)avail);
    if (rc)
        return rc;
    sk.sink_pos = sk.total; // handed over: the next round drops it from the stream buffer
    if (n_out)
        *n_out = (size_t)avail;
    return PSDC_OK;
}

int psdc_stage_process_device(psdc_stage *st, const float *d_x, size_t len, float *d_y, size_t cap, size_t *n_out)
{
    if (n_out)
        *n_out = 0;
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    psdc_handle *h = st->h;
    ON_DEVICE(h, h->device);
    int rc = psdc_process_device(h, 0, d_x, len);
    if (rc)
        return rc;
    rc = flush_all(h);
    if (rc)
        return rc;
    Channel &c = h->ch[0];
    uint64_t avail = 0;
    if (c.st.size() >= 2) {
        StageState &sk = c.st[1];
        avail = sk.total - sk.sink_pos;
        if (avail) {
            if (!d_y || cap < avail) {
                // x has been consumed (as in the reference, which panics AFTER buffering, src/psd.rs:201-253) and
                // the enqueued kernels may still be reading d_x: wait for them, so that d_x is free on return as
                // documented; the outputs stay pending and a later call with room returns them
                HIPCHK(h, hipStreamSynchronize(h->stream));
                (void)release_retired(h);
                return fail(h, PSDC_ERR_CAPACITY, "psdc_stage_process_device: y too small (x was consumed; the outputs stay pending)");
            }
            HIPCHK(h, hipMemcpyAsync(d_y, sk.buf.p[sk.buf.cur] + (sk.sink_pos - sk.buf.base), sizeof(float) * avail,
                                     hipMemcpyDeviceToDevice, h->stream));
            sk.sink_pos = sk.total;
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream)); // d_x has been read, d_y is complete
    rc = release_retired(h);
    if (rc)
        return rc;
    if (n_out)
        *n_out = (size_t)avail;
    return PSDC_OK;
}

int psdc_stage_get_spectrum(psdc_stage *st, float *out)
{
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    if (!out)
        return fail(st->h, PSDC_ERR_ARG, "null output");
    if (st->h->ch[0].st.empty()) { // a fresh Psd: spectrum is all zeros (src/psd.rs:146)
        memset(out, 0, sizeof(float) * (st->h->n / 2 + 1));
        return PSDC_OK;
    }
    return psdc_stage_spectrum(st->h, 0, 0, out);
}

int psdc_stage_get_count(psdc_stage *st, uint32_t *count)
{
    if (!st || !count)
        return fail(st ? st->h : nullptr, PSDC_ERR_ARG, "null argument");
    *count = st->h->ch[0].st.empty() ? 0u : st->h->ch[0].st[0].count;
    return PSDC_OK;
}

int psdc_stage_get_gain(psdc_stage *st, float *gain)
{
    if (!st || !gain)
        return fail(st ? st->h : nullptr, PSDC_ERR_ARG, "null argument");
    psdc_handle *h = st->h;
    *gain = stage_gain(h->n, h->ch[0].st.empty() ? 0 : h->ch[0].st[0].count64, h->nenbw, h->power);
    return PSDC_OK;
}

int psdc_stage_get_buf(psdc_stage *st, float *out, size_t cap, size_t *len)
{
    if (!st)
        return fail(nullptr, PSDC_ERR_ARG, "null stage");
    if (st->h->ch[0].st.empty()) {
        if (len)
            *len = 0;
        return PSDC_OK;
    }
    return psdc_stage_buf(st->h, 0, 0, out, cap, len);
}

int psdc_profile_read(psdc_handle *h, psdc_profile *out, int reset)
{
    if (!h || !out)
        return fail(h, PSDC_ERR_ARG, "null argument");
    ON_DEVICE(h, h->device);
    int rc = collect_profile(h);
    if (rc)
        return rc;
    *out = h->prof;
    if (reset)
        h->prof = psdc_profile{};
    return PSDC_OK;
}

} // extern "C"
