// host_runtime.h -- what the translation units of the HOST runtime behind include/psdcascade.h share: the per-stream state
// (StageState, Channel, psdc_handle), error plumbing and the internal entry points of each part.
//
//   runtime.cpp        handle lifecycle, stream buffers, staging / uploads, psdc_process / psdc_process_device, flush / sync
//   planner.cpp        advance_round: one round of the cascade pipeline turned into kernel jobs (plan.h gives the closed forms)
//   frames_ingest.cpp  frames in host or device memory (Frame::from_bytes, Loss::update, Payload::traces)
//   readout.cpp        PsdStage accessors, PsdCascade::psd stitch, Break, packed read-out, Var / Trace::plot, the single-stage Psd<N>
//
// Mirrors PsdCascade<N> (src/psd.rs:399-544) for `n_channels` independent traces on one MI355X: per (channel, stage) the
// stream position, the count and a device stream buffer.  There is no CPU compute path: every spectrum comes from the HIP kernels.
#ifndef PSDC_HOST_RUNTIME_H
#define PSDC_HOST_RUNTIME_H

#include "../../include/psdcascade.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "hbf_taps.h"
#include "kernels.h"
#include "plan.h"

namespace psdrt {

using namespace psdk;

constexpr uint32_t MAX_STAGES = 16; // 8^16 N samples: unreachable; slots of the spectra slab
#ifndef PSDC_MAX_COALESCE
#define PSDC_MAX_COALESCE 128
#endif
constexpr int MAX_COALESCE = PSDC_MAX_COALESCE; // zero-copy spans of one channel in one round (the automatic depth for one channel fed in short f32 spans)
constexpr int MAX_COALESCE_OPT = 16; // ... as an explicit PSDC_OPT_COALESCE, and for runs of AdcDac frames: a launch's frame-span table
// ... and samples a channel holds back at most (hold_max below): what eight 2^26-sample spans make -- sixteen for a handle of ONE channel,
// whose round is all its own (a 2^30-sample round of sixteen 2^26-sample spans reads 1.0-1.7 % above two of eight: half the launch
// gaps and run starts a sample).  A span merged from contiguous calls stops growing at HOLD_MAX_SAMPLES either way: 2^31 bytes, so
// every byte offset into a span fits 32 bits.
constexpr size_t HOLD_MAX_SAMPLES = (size_t)1 << 29;
inline size_t hold_cap_from_env()
{
    const char *e = getenv("PSDC_DBG_HOLD_LOG2"); // (testing aid: tests/host/round_plan_check.cpp reaches the caps with streams of a few million samples)
    const long v = e ? strtol(e, nullptr, 10) : 0;
    return v >= 12 && v <= 29 ? (size_t)1 << v : HOLD_MAX_SAMPLES;
}
static_assert(MAX_COALESCE_OPT <= MAX_FSPANS, "a launch's frame-span table holds every FRAMED span of a round (frame calls hold at most PSDC_OPT_COALESCE <= 16)");

extern thread_local std::string g_last_error;

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

struct WindowConsts {
    float nenbw, power;
    uint32_t overlap;
};

} // namespace psdrt

struct psdc_handle {
    uint32_t n = 0;
    int window_kind = PSDC_WINDOW_HANN;
    psdk::Geometry geo;
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
    // wait for its post launch: ev_post is recorded behind it -- lazily, when an upload is about to be enqueued (order_upload), so
    // that device-fed streams record no event at all (a recorded event is a system-scope release behind the launch).
    hipEvent_t ev_post = nullptr;
    bool post_marked = false;
    bool post_dirty = false; // a post launch has been enqueued since ev_post was last recorded (recorded lazily, by order_upload)
    float *d_win = nullptr;
    psdk::cf *d_tw = nullptr;
    psdk::cf *d_tw0g = nullptr, *d_twag = nullptr; // twiddle tables of the N >= 2048 fused kernels
    psdk::cf *d_tw3g = nullptr;                    // twiddle seeds of the three-pass kernels (N = 2048, 4096)
    psdk::cf *d_chirp = nullptr, *d_bhat = nullptr; // chirp-z tables of a size that is not a power of two (launch_welch)
    psdk::cf *d_bigfft = nullptr;                   // n > 16384: the ping-pong frames of the global-memory FFT (launch_welch_big)
    size_t bigfft_elems = 0;
    int bigfft_chunk_limit = 0;               // PSDC_DBG_BIGFFT_CHUNK at create: pairs per chunk (tests: a job split over chunks)
    int detrend = PSDC_DETREND_NONE;
    uint32_t avg_limit = 0xFFFFFFFFu, avg_count = 0xFFFFFFFFu;
    std::vector<psdrt::Channel> ch;
    float *d_spectra = nullptr; // [n_channels][MAX_STAGES][n] accumulators, one slab
    float *h_read = nullptr;    // pinned bounce buffer for read-outs (MAX_STAGES * n floats)
    unsigned long long *d_scan = nullptr; // 5 words: accumulators of the device-side frame header scan + Loss sums (kept zero)
    uint8_t *h_hdr = nullptr;             // pinned: the frame headers of one psdc_process_frames_device call (launch_header_gather)
    size_t h_hdr_cap = 0;                 // bytes
    hipStream_t hdr_stream = nullptr;     // the gather runs beside the compute stream's work (the host waits for it alone)
    unsigned long long *h_scan = nullptr; // pinned: its four result words
    hipStream_t scan_stream = nullptr;    // the scan runs beside the compute stream's work (the host waits for it alone)
    std::vector<psdk::FrameSpan> fs_pool;       // frame spans named by this round's jobs (FusedJob::fspan ... index this until a launch maps them)
    float *d_pool = nullptr;    // [n_channels][MAX_STAGES][2][pool_cap] small stream buffers (deep stages)
    size_t pool_cap = 0;        // floats per pooled buffer
    bool idle = true;           // nothing ingested since the pipeline was last drained
    float *d_partial = nullptr; // TWO slabs of partial_cap floats: a round writes slab partial_cur while the fold of the round before may
    size_t partial_cap = 0;     // still be reading the other one (it rides in this round's fused launch when the round is one launch)
    int partial_cur = 0;
    // stream buffers replaced by larger ones: work already enqueued may still read them, so they are freed
    // at the next point where the stream is known to be idle (release_retired) -- growing never waits
    std::vector<float *> retired;
    // epilogue of the last round (fold the partials into the spectra, carry the stream tails),
    // not launched yet: it rides in the first launch of the next round or of a read-out
    std::vector<psdk::RedJob> pend_red;
    std::vector<psdk::TailJob> pend_tail;
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
    bool coalesce_auto = true; // PSDC_OPT_COALESCE not set: `coalesce`, or 16 ... MAX_COALESCE for one channel fed in short spans (coalesce_limit)
    uint32_t stage_limit = psdrt::MAX_STAGES; // stages that analyse their stream; 1 for a single Psd<N> (psdc_stage_*)
    uint32_t min_pairs = 0; // PSDC_OPT_MIN_PAIRS: segment pairs a decimated stage collects before it issues on the ingest path
    size_t span_cap = psdrt::hold_cap_from_env(); // HOLD_MAX_SAMPLES ($PSDC_DBG_HOLD_LOG2, read when the handle is made: the CPU model's streams are short)
    bool merge = true;  // PSDC_OPT_MERGE: a device span that continues the last held one in memory extends it
    bool fold = getenv("PSDC_NO_FOLD") == nullptr; // (A/B aid, read when the handle is made: unset = one launch per round where the kernel allows)
    bool eager = false; // PSDC_OPT_EAGER: a held span goes out as soon as the device is seen idle (round composition then follows host timing)
    bool profile = false;
    std::vector<psdrt::ProfEvents> prof_pending;
    psdc_profile prof{};
    psdc_loss loss{};
    std::string err;
};

namespace psdrt {

int fail(psdc_handle *h, int code, const std::string &msg);

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

// ---- runtime.cpp ------------------------------------------------------------------------------------------------------------
bool window_consts(uint32_t n, int kind, WindowConsts *w);
void window_weights(uint32_t n, int kind, float *win);
int classify_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap);
bool valid_n(uint32_t n);
float stage_gain(uint32_t n, uint64_t count, float nenbw, float power);
uint32_t cur_stage_avg(const psdc_handle *h, size_t i);
uint64_t keep_from(const Geometry &g, const StageState &s);
int free_stage(psdc_handle *h, StageState &s);
int add_stage(psdc_handle *h, Channel &c);
int pool_fspan(psdc_handle *h, const FrameSpan &fs);
TailJob span_copy(psdc_handle *h, const DeviceSpan &sp, uint64_t from, float *dst, size_t count);
int launch_deferred(psdc_handle *h, const std::vector<TailJob> &extra);
void split_copy_jobs(const std::vector<TailJob> &in, std::vector<TailJob> &out); // long copies cut into pieces of 16 Ki samples
int wait_uploads(psdc_handle *h);
int order_upload(psdc_handle *h);
int mark_upload(psdc_handle *h);
int ensure_room(psdc_handle *h, StageState &s, uint64_t new_end);
int ensure_cap(psdc_handle *h, StageState &s, size_t need, size_t grow_to = 0);
int ensure_partial(psdc_handle *h, size_t floats);
int collect_profile(psdc_handle *h);
uint32_t coalesce_limit(const psdc_handle *h, const Channel &c, size_t len = 0);
bool device_idle(psdc_handle *h);
size_t held_samples(const Channel &c);
inline size_t hold_max(const psdc_handle *h) { return h->n_channels == 1 ? 2 * h->span_cap : h->span_cap; }
int settle_short_span(psdc_handle *h, Channel &c);
bool holds_short_span(const psdc_handle *h, const Channel &c);
int submit_host(psdc_handle *h, Channel &c);
int ensure_staging(psdc_handle *h, Channel &c);
int free_staging(psdc_handle *h, Channel &c);
int flush_all(psdc_handle *h);
int release_retired(psdc_handle *h);
int flush_sync(psdc_handle *h);
int read_back(psdc_handle *h, float *dst, const float *d_src, size_t count);
int check_channel(psdc_handle *h, uint32_t channel);
void pinned_copy(void *dst, const void *src, size_t bytes); // into pinned staging memory, split over a few threads when large
psdc_handle *create_impl(uint32_t n, int window_kind, const float *win_in, WindowConsts wc, uint32_t n_channels, int device);
const char *check_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap, int *kind, WindowConsts *wc);

// pool index -> index in a launch's own table (at most MAX_FSPANS distinct spans per launch: the planner holds a channel
// to MAX_COALESCE_OPT = MAX_FSPANS FRAMED spans per round, and the four traces of a span share one entry)
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

// ---- planner.cpp ------------------------------------------------------------------------------------------------------------
int fused_window(const psdc_handle *h); // 0: no fused kernel for this window, 1: half-overlapped pairs, 2: single segments
int advance_round(psdc_handle *h, bool *did_work, bool all);
int advance(psdc_handle *h); // one pipeline round (ingest path)
int drain(psdc_handle *h);   // rounds until the pipeline is idle (read-out path)

// ---- readout.cpp ------------------------------------------------------------------------------------------------------------
int stitch_impl(uint32_t n, float nenbw, float power, uint32_t overlap, uint32_t n_stages, const uint32_t *counts,
                const uint32_t *avgs, const uint64_t *pendings, const float *spectra, int keep_overlap, uint32_t min_count,
                int keep_transition_band, float *psd_out, size_t psd_cap, size_t *psd_len, psdc_break *breaks, size_t breaks_cap,
                size_t *n_breaks, const uint64_t *counts64 = nullptr);

} // namespace psdrt

#endif
