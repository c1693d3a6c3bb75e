// tests/host/round_plan_check.cpp -- the library's host runtime (stabilizer-stream_amd/csrc/{runtime,planner,frames_ingest,readout}.cpp, UNCHANGED: the
// round planner advance_round, the staging / upload pipeline, frame ingest, read-outs) on the CPU, under AddressSanitizer and
// UBSan, against a host model of the HIP runtime and of the kernels (tests/host/sim/).  TEST INFRASTRUCTURE.
//
// The feeds are those of the GPU fuzz campaigns (tools/stress_campaign.py, span_campaign.py, frames_device_campaign.py): host
// and in-place device spans of odd lengths and alignments, several channels fed unevenly, coalescing depths +-1..16 on a
// "device" that is busy until the host synchronises, mid-stream read-outs, detrend / averaging changes, AdcDac frames in host
// and in device memory mixed with f32 feeds, single-stage handles.  Instead of spectra the model tracks IDENTITY
// (sim_device.h): every kernel job must read exactly the samples it is meant to read, from memory that exists, and at the end
//   - every segment index of every (channel, stage) stream was transformed exactly once,
//   - every decimator output index was produced exactly once (the invariants of src/psd.rs:196-269),
//   - the counters, the pending samples (psdc_stage_buf) and, for plain sums, the accumulator's segment count are the closed
//     forms of csrc/plan.h.
// usage: round_plan_check [first_seed [count]]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/psdcascade.h"
#include "hbf_taps.h"
#include "plan.h"
#include <hip/hip_runtime.h> // tests/host/sim/hip/hip_runtime.h: the host model of the runtime (-Itests/host/sim)

#include "sim_device.h"

using namespace psdk;
using sim::ident;
using sim::world;

namespace {

int g_failures = 0;
std::string g_ctx;

void fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g_failures < 30)
        fprintf(stderr, "FAIL [%s] %s\n", g_ctx.c_str(), buf);
    ++g_failures;
}
#define CK(call)                                                                 \
    do {                                                                         \
        const int rc_ = (call);                                                  \
        if (rc_ < 0)                                                             \
            fail("%s -> %d (%s)", #call, rc_, psdc_last_error(h));              \
    } while (0)

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s * 0x9E3779B97F4A7C15ull + 12345) {}
    uint64_t u(uint64_t lo, uint64_t hi) { return lo + g() % (hi - lo); } // [lo, hi)
    double f() { return (double)(g() >> 11) / 9007199254740992.0; }
};

std::vector<float> ident_stream(int ch, size_t total)
{
    std::vector<float> x(total);
    for (size_t i = 0; i < total; ++i)
        sim::put_bits(&x[i], ident(ch * 16, i));
    return x;
}

// frames whose header seq is the absolute batch index; payload bytes arbitrary (the model identifies samples from the header)
std::vector<uint8_t> make_frames(size_t n_frames, int batches, uint32_t seq0)
{
    const size_t fs = 8 + 64 * (size_t)batches;
    std::vector<uint8_t> b(n_frames * fs, 0x5a);
    for (size_t f = 0; f < n_frames; ++f) {
        uint8_t *p = &b[f * fs];
        p[0] = 0x7b, p[1] = 0x05, p[2] = 1, p[3] = (uint8_t)batches;
        const uint32_t s = seq0 + (uint32_t)(f * (size_t)batches);
        p[4] = (uint8_t)s, p[5] = (uint8_t)(s >> 8), p[6] = (uint8_t)(s >> 16), p[7] = (uint8_t)(s >> 24);
    }
    return b;
}

// after a final read-out: closed forms against what the model saw and what the ABI reports
void verify(psdc_handle *h, uint32_t n, uint32_t overlap, int nch, const std::vector<uint64_t> &fed, bool plain_sum, uint32_t stage_limit = 16)
{
    Geometry g;
    g.n = n, g.overlap = overlap, g.hop = n - overlap, g.drain = (uint32_t)HBF_DRAIN;
    sim::drain();
    for (const std::string &e : world().errors)
        fail("model: %s", e.c_str());
    for (int c = 0; c < nch; ++c) {
        uint64_t total = fed[(size_t)c];
        const int ns = psdc_num_stages(h, (uint32_t)c);
        int k = 0;
        for (; total > 0 && k < 16; ++k) {
            const int tag = c * 16 + k;
            const bool sink = (uint32_t)k >= stage_limit;
            const uint64_t J = sink ? 0 : segments_for(g, total), P = decimated_prefix(g, J);
            if (k >= ns) {
                fail("channel %d: stage %d holds %llu samples but the handle has %d stages", c, k, (unsigned long long)total, ns);
                break;
            }
            const std::vector<uint8_t> &sv = world().seg[tag], &dv = world().dec[tag];
            for (uint64_t s = 0; s < std::max<uint64_t>(J, sv.size()); ++s) {
                const int got = s < sv.size() ? sv[s] : 0, want = s < J ? 1 : 0;
                if (got != want) {
                    fail("channel %d stage %d: segment %llu transformed %d times (of %llu segments)", c, k, (unsigned long long)s, got,
                         (unsigned long long)J);
                    break;
                }
            }
            for (uint64_t m = 0; m < std::max<uint64_t>(P / 8, dv.size()); ++m) {
                const int got = m < dv.size() ? dv[m] : 0, want = m < P / 8 ? 1 : 0;
                // (the first `drain` outputs of a stream are discarded, src/psd.rs:255-260: a job that would produce nothing else is
                // not issued at all, so those may be computed once or never)
                if (got != want && !(m < (uint64_t)HBF_DRAIN && got == 0)) {
                    fail("channel %d stage %d: decimator output %llu produced %d times (of %llu)", c, k, (unsigned long long)m, got,
                         (unsigned long long)(P / 8));
                    break;
                }
            }
            psdc_stage_stat st{};
            CK(psdc_stage_info(h, (uint32_t)c, (uint32_t)k, &st));
            if (!sink) {
                // (processed = N count - overlap (count - 1), src/psd.rs:511-512: follows the count, which finite averaging saturates)
                if (st.pending != pending_for(g, total) ||
                    (plain_sum && st.processed != (J ? (uint64_t)n * J - (uint64_t)overlap * (J - 1) : 0)))
                    fail("channel %d stage %d: pending %llu processed %llu, closed form %llu / %llu", c, k, (unsigned long long)st.pending,
                         (unsigned long long)st.processed, (unsigned long long)pending_for(g, total),
                         (unsigned long long)(J ? (uint64_t)n * J - (uint64_t)overlap * (J - 1) : 0));
                if (plain_sum && st.count != J)
                    fail("channel %d stage %d: count %u, %llu segments", c, k, st.count, (unsigned long long)J);
                std::vector<float> sp(n / 2 + 1);
                CK(psdc_stage_spectrum(h, (uint32_t)c, (uint32_t)k, sp.data()));
                if (plain_sum && sp[0] != (float)J)
                    fail("channel %d stage %d: the accumulator holds %g segments, %llu were due", c, k, (double)sp[0], (unsigned long long)J);
                // the pending samples are the stream's last ones, in order (tail carries, seams, staging)
                std::vector<float> pb(2 * (size_t)n + 64);
                size_t len = 0;
                CK(psdc_stage_buf(h, (uint32_t)c, (uint32_t)k, pb.data(), pb.size(), &len));
                if (len != st.pending)
                    fail("channel %d stage %d: buf() holds %zu samples, pending %llu", c, k, len, (unsigned long long)st.pending);
                for (size_t i = 0; i < len; ++i)
                    if (sim::bits_of(&pb[i]) != ident(tag, total - len + i)) {
                        fail("channel %d stage %d: pending sample %zu is tag %u sample %u, not sample %llu", c, k, i, sim::bits_of(&pb[i]) >> 24,
                             sim::bits_of(&pb[i]) & 0xFFFFFFu, (unsigned long long)(total - len + i));
                        break;
                    }
            }
            total = sink ? 0 : emitted_for(g, P);
        }
        if (k != ns && !(k < ns && total == 0))
            fail("channel %d: %d stages expected, the handle has %d", c, k, ns);
    }
}

void new_scenario(const std::string &ctx, int cap_blocks)
{
    g_ctx = ctx;
    if (getenv("ROUND_PLAN_VERBOSE"))
        fprintf(stderr, "%s\n", ctx.c_str());
    sim::drain();
    world() = sim::World{};
    world().cap_blocks = cap_blocks;
}

// ---- the randomized feed stress (tests/test_gpu_parity.py::test_randomized_feed_stress) ---------------------------------
void scenario_stress(uint64_t seed)
{
    Rng r(seed);
    static const uint32_t sizes[] = {64, 256, 512, 1024, 1024, 2048, 4096, 8192, 16384, 80, 1200, 32768, 512};
    const uint32_t n = sizes[seed % 13];
    const bool rect = r.f() < 0.15; // overlap 0: the generic kernels only
    const uint32_t overlap = rect ? 0 : n / 2;
    const int nch = (int)r.u(1, 4);
    const int caps[] = {0, 0, 3, 8, 40};
    new_scenario("stress seed " + std::to_string(seed) + " n=" + std::to_string(n) + (rect ? " rectangular" : ""), caps[r.u(0, 5)]);
    psdc_handle *h = psdc_create(n, rect ? PSDC_WINDOW_RECTANGULAR : PSDC_WINDOW_HANN, (uint32_t)nch, 0);
    if (!h) {
        fail("psdc_create: %s", psdc_last_error(nullptr));
        return;
    }
    const size_t total = std::min<size_t>((size_t)r.u(60, 140) * n * 8, (size_t)12 << 20);
    std::vector<std::vector<float>> xs;
    for (int c = 0; c < nch; ++c)
        xs.push_back(ident_stream(c, total));
    CK(psdc_configure(h, PSDC_OPT_QUANTUM, (int64_t)r.u(2, 20) * n));
    const int co[] = {1, 4, 8, 16, -2, -4, -8, -16};
    const int cov = co[r.u(0, 8)];
    CK(psdc_configure(h, PSDC_OPT_COALESCE, cov));
    CK(psdc_configure(h, PSDC_OPT_EAGER, cov > 0 ? 1 : 0)); // (positive: held spans also go out when the modelled stream is idle)
    CK(psdc_configure(h, PSDC_OPT_MERGE, r.f() < 0.35 ? 1 : 0)); // (the feeds are slices of one array: merged they are ONE span)
    if (r.f() < 0.5)
        CK(psdc_configure(h, PSDC_OPT_MIN_PAIRS, (int64_t)r.u(0, 40)));
    std::vector<uint64_t> pos((size_t)nch, 0);
    bool plain = true;
    while (*std::min_element(pos.begin(), pos.end()) < total) {
        size_t c = r.u(0, (uint64_t)nch);
        if (pos[c] >= total)
            c = (size_t)(std::min_element(pos.begin(), pos.end()) - pos.begin());
        const double kind = r.f();
        if (kind < 0.06) {
            CK(psdc_set_detrend(h, (int)r.u(0, 4)));
            continue;
        }
        if (kind < 0.09) {
            CK(psdc_set_avg(h, (uint32_t)r.u(1, 50), (uint32_t)r.u(1, 400)));
            plain = false;
            continue;
        }
        if (kind < 0.16) { // mid-stream read-out of one channel
            uint32_t ns = 0;
            std::vector<psdc_stage_stat> st(16);
            std::vector<float> sp((size_t)16 * (n / 2 + 1));
            CK(psdc_read_channel(h, (uint32_t)r.u(0, (uint64_t)nch), 16, &ns, st.data(), sp.data()));
            continue;
        }
        if (kind < 0.18) {
            CK(psdc_sync(h));
            continue;
        }
        if (kind < 0.195) { // Clone (src/psd.rs:399): the copy carries on where the original stood, the original goes
            psdc_handle *c = psdc_clone(h);
            if (!c) {
                fail("psdc_clone: %s", psdc_last_error(nullptr));
                continue;
            }
            psdc_destroy(h);
            h = c;
            continue;
        }
        if (kind < 0.2 && *std::max_element(pos.begin(), pos.end()) < total / 3) { // Cmd::Reset (src/bin/psd.rs:190): start over
            CK(psdc_reset(h));
            sim::drain();
            const int cap = world().cap_blocks;
            world() = sim::World{};
            world().cap_blocks = cap;
            std::fill(pos.begin(), pos.end(), 0);
            continue;
        }
        const uint64_t pick = r.u(0, 3);
        uint64_t m = pick == 0 ? r.u(1, 50) : pick == 1 ? r.u(1, 6 * (uint64_t)n) : r.u(6 * (uint64_t)n, 40 * (uint64_t)n);
        m = std::min<uint64_t>(m, total - pos[c]);
        if (r.f() < 0.5)
            CK(psdc_process(h, (uint32_t)c, xs[c].data() + pos[c], (size_t)m));
        else
            CK(psdc_process_device(h, (uint32_t)c, xs[c].data() + pos[c], (size_t)m)); // ("device" memory is host memory here)
        pos[c] += m;
    }
    float psd[16 * 8200];
    psdc_break br[16];
    size_t pl = 0, nb = 0;
    for (int c = 0; c < nch; ++c)
        CK(psdc_psd(h, (uint32_t)c, 0, 1, 0, psd, sizeof psd / sizeof psd[0], &pl, br, 16, &nb));
    verify(h, n, overlap, nch, pos, plain);
    psdc_destroy(h);
}

// ---- zero-copy spans: many per round, pair-aligned and not, on a busy device (tools/span_campaign.py) ----------------
void scenario_spans(uint64_t seed)
{
    Rng r(seed);
    static const uint32_t sizes[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 1024};
    const uint32_t n = sizes[seed % 8];
    // (one seed in five: ONE channel at the library's own coalescing depth -- up to 128 short spans in a round, ~270 fused jobs)
    const bool deep = seed % 5 == 2;
    const int nch = deep ? 1 : (int)r.u(1, 9);
    const int caps[] = {0, 2, 5, 16, 64};
    const bool rect = r.f() < 0.3; // overlap 0: the single-segment form of the fused kernels
    new_scenario("spans seed " + std::to_string(seed) + " n=" + std::to_string(n) + (rect ? " rectangular" : ""), caps[r.u(0, 5)]);
    // (two seeds in five: the caps on what a channel holds and on how far a merged span grows, HOLD_MAX_SAMPLES = 2^29 in the library,
    // brought down to 2^16 ... 2^19 samples so that streams of a few million samples reach them)
    const bool low_caps = seed % 5 == 2 || seed % 5 == 4;
    if (low_caps)
        setenv("PSDC_DBG_HOLD_LOG2", std::to_string(16 + seed / 5 % 4).c_str(), 1);
    else
        unsetenv("PSDC_DBG_HOLD_LOG2");
    psdc_handle *h = psdc_create(n, rect ? PSDC_WINDOW_RECTANGULAR : PSDC_WINDOW_HANN, (uint32_t)nch, 0);
    unsetenv("PSDC_DBG_HOLD_LOG2");
    if (!h) {
        fail("psdc_create: %s", psdc_last_error(nullptr));
        return;
    }
    const size_t total = std::min<size_t>((size_t)r.u(100, 400) * n, (size_t)6 << 20);
    std::vector<std::vector<float>> xs;
    for (int c = 0; c < nch; ++c)
        xs.push_back(ident_stream(c, total));
    if (!deep)
        CK(psdc_configure(h, PSDC_OPT_COALESCE, (int64_t)r.u(1, 17)));
    CK(psdc_configure(h, PSDC_OPT_EAGER, !deep && r.f() < 0.4 ? 1 : 0)); // (eager: held spans go out when the modelled stream is idle)
    CK(psdc_configure(h, PSDC_OPT_MERGE, r.f() < (deep ? 0.5 : 0.3) ? 1 : 0)); // (the feeds are slices of one array: merged they grow ONE span, up to the cap)
    if (r.f() < 0.5)
        CK(psdc_configure(h, PSDC_OPT_MIN_PAIRS, (int64_t)r.u(0, 300)));
    std::vector<uint64_t> pos((size_t)nch, 0);
    while (*std::min_element(pos.begin(), pos.end()) < total) {
        for (int c = 0; c < nch; ++c) { // the channels advance together, like the bench's rounds
            if (pos[(size_t)c] >= total)
                continue;
            uint64_t m = r.f() < 0.3 ? r.u(1, 3 * (uint64_t)n) : r.u(4 * ((uint64_t)n + 288), 30 * (uint64_t)n);
            if (r.f() < 0.5)
                m &= ~(uint64_t)3; // 16-byte aligned continuations (the in-place fast path) half of the time
            m = std::min<uint64_t>(std::max<uint64_t>(m, 1), total - pos[(size_t)c]);
            CK(psdc_process_device(h, (uint32_t)c, xs[(size_t)c].data() + pos[(size_t)c], (size_t)m));
            pos[(size_t)c] += m;
        }
        if (r.f() < 0.1)
            CK(psdc_flush(h));
        if (r.f() < 0.05)
            (void)psdc_num_stages(h, 0);
    }
    CK(psdc_sync(h));
    verify(h, n, rect ? 0 : n / 2, nch, pos, true);
    psdc_destroy(h);
}

// ---- AdcDac frames in host and in device memory, mixed with f32 feeds (tools/frames_device_campaign.py) ---------------
void scenario_frames(uint64_t seed)
{
    Rng r(seed);
    static const uint32_t sizes[] = {2048, 4096, 256, 8192, 16384, 512, 1024, 64, 1200};
    const uint32_t n = sizes[seed % 9];
    const int batches = (int)r.u(1, 32);
    const size_t per_frame = (size_t)batches * 8, fs = 8 + 64 * (size_t)batches;
    const size_t n_frames = std::min<size_t>((size_t)r.u(30 * (uint64_t)n, 200 * (uint64_t)n), (size_t)5 << 20) / per_frame;
    const int caps[] = {0, 4, 12, 64};
    new_scenario("frames seed " + std::to_string(seed) + " n=" + std::to_string(n) + " batches=" + std::to_string(batches), caps[r.u(0, 4)]);
    psdc_handle *h = psdc_create(n, PSDC_WINDOW_HANN, 4, 0);
    if (!h) {
        fail("psdc_create: %s", psdc_last_error(nullptr));
        return;
    }
    // the stream: f32 prefix of pre8 * 8 samples (possibly none), then the frames, whose seq continue from there
    const size_t pre = r.f() < 0.3 ? (size_t)r.u(1, 3 * (uint64_t)n / 8) * 8 : 0;
    std::vector<uint8_t> frames = make_frames(n_frames, batches, (uint32_t)(pre / 8));
    const size_t shift = r.f() < 0.2 ? (size_t)r.u(1, 8) : 0; // a base that is not 8-byte aligned: the decode path
    std::vector<uint8_t> moved(frames.size() + 16);
    uint8_t *base = moved.data() + ((8 - (reinterpret_cast<uintptr_t>(moved.data()) & 7)) & 7) + shift;
    memcpy(base, frames.data(), frames.size());
    CK(psdc_configure(h, PSDC_OPT_COALESCE, (int64_t)r.u(1, 9)));
    CK(psdc_configure(h, PSDC_OPT_EAGER, r.f() < 0.5 ? 1 : 0));
    std::vector<std::vector<float>> xs;
    for (int c = 0; c < 4; ++c)
        xs.push_back(ident_stream(c, pre + 8));
    for (int c = 0; c < 4 && pre; ++c) {
        if (r.f() < 0.5)
            CK(psdc_process(h, (uint32_t)c, xs[(size_t)c].data(), pre));
        else
            CK(psdc_process_device(h, (uint32_t)c, xs[(size_t)c].data(), pre));
    }
    size_t pos = 0;
    while (pos < n_frames) {
        const bool big = r.f() < 0.7;
        const size_t lo = 4 * ((size_t)n + 288) / per_frame + 1, hi = std::max<size_t>(lo + 1, 90 * (size_t)n / per_frame);
        const size_t m = std::min<size_t>(n_frames - pos, big ? (size_t)r.u(lo, hi) : (size_t)r.u(1, 60));
        size_t ok = 0;
        if (r.f() < 0.75)
            CK(psdc_process_adcdac_frames_device(h, base + pos * fs, fs, m, &ok));
        else
            CK(psdc_process_adcdac_frames(h, base + pos * fs, fs, m, &ok));
        if (ok != m)
            fail("frames call accepted %zu of %zu frames", ok, m);
        pos += m;
        if (r.f() < 0.15)
            (void)psdc_num_stages(h, (uint32_t)r.u(0, 4));
    }
    psdc_loss loss{};
    CK(psdc_loss_read(h, &loss, 0));
    if (loss.received != (uint64_t)n_frames * (uint64_t)batches || loss.dropped != 0)
        fail("Loss: received %llu dropped %llu, %zu frames of %d batches without a gap", (unsigned long long)loss.received,
             (unsigned long long)loss.dropped, n_frames, batches);
    CK(psdc_sync(h));
    std::vector<uint64_t> fed(4, pre + n_frames * per_frame);
    verify(h, n, n / 2, 4, fed, true);
    // a bad frame in the middle of a call: the frames before it are ingested, the rest is not
    if (n_frames > 8) {
        std::vector<uint8_t> more = make_frames(8, batches, (uint32_t)((pre + n_frames * per_frame) / 8));
        more[5 * fs] = 0; // magic of frame 5
        size_t ok = 99;
        const int rc = psdc_process_adcdac_frames_device(h, more.data(), fs, 8, &ok);
        if (rc != PSDC_ERR_FRAME_HEADER || ok != 5)
            fail("bad frame 5 of 8: rc %d, %zu frames accepted", rc, ok);
        CK(psdc_sync(h));
        for (auto &f : fed)
            f += 5 * per_frame;
        verify(h, n, n / 2, 4, fed, true);
    }
    psdc_destroy(h);
}

// ---- the single stage Psd<N> (psdc_stage_*): the decimated stream is handed back, in order -----------------------------
void scenario_single(uint64_t seed)
{
    Rng r(seed);
    static const uint32_t sizes[] = {512, 1024, 256, 4096, 64};
    const uint32_t n = sizes[seed % 5];
    new_scenario("single-stage seed " + std::to_string(seed) + " n=" + std::to_string(n), 0);
    psdc_stage *st = psdc_stage_create(n, PSDC_WINDOW_HANN, 0);
    if (!st) {
        fail("psdc_stage_create: %s", psdc_last_error(nullptr));
        return;
    }
    const size_t total = (size_t)r.u(20, 90) * n;
    std::vector<float> x = ident_stream(0, total), y(total / 8 + n);
    size_t pos = 0, out = 0;
    Geometry g;
    g.n = n, g.overlap = n / 2, g.hop = n / 2, g.drain = (uint32_t)HBF_DRAIN;
    while (pos < total) {
        const size_t m = std::min<size_t>(total - pos, (size_t)r.u(1, 8 * (uint64_t)n));
        size_t got = 0;
        const int rc = r.f() < 0.5 ? psdc_stage_process(st, x.data() + pos, m, y.data() + out, y.size() - out, &got)
                                   : psdc_stage_process_device(st, x.data() + pos, m, y.data() + out, y.size() - out, &got);
        if (rc < 0)
            fail("psdc_stage_process -> %d (%s)", rc, psdc_stage_last_error(st));
        sim::drain();
        pos += m;
        out += got;
        const uint64_t want = emitted_for(g, decimated_prefix(g, segments_for(g, pos)));
        if (out != want)
            fail("after %zu samples the stage has handed back %zu items, %llu are due (src/psd.rs:622)", pos, out, (unsigned long long)want);
    }
    for (size_t i = 0; i < out; ++i)
        if (sim::bits_of(&y[i]) != ident(1, i)) {
            fail("item %zu of the decimated stream is tag %u sample %u", i, sim::bits_of(&y[i]) >> 24, sim::bits_of(&y[i]) & 0xFFFFFFu);
            break;
        }
    for (const std::string &e : world().errors)
        fail("model: %s", e.c_str());
    psdc_stage_destroy(st);
}

// ---- when a held round goes out (csrc/runtime.cpp psdc_process_device_after): a function of the calls alone -------------
// One channel; the caps on a merged span / on what the channel holds (2^29 / 2^30 samples in the library) brought down to 2^16 / 2^17.
void scenario_hold_rules()
{
    const uint32_t n = 256;
    const size_t piece = 4096, span = 8192; // (a span is read in place from 4 (n + 288) = 2176 samples)
    struct Case {
        const char *what;
        int hold_log2;       // PSDC_DBG_HOLD_LOG2 for the handle (0: the library's caps, far out of reach here)
        int merge, coalesce; // coalesce 0: the library's own depth
        bool contiguous;     // pieces of one array in order (they continue each other in memory), or every span an array of its own
        size_t len;
        int calls;
        bool (*goes_out)(int i); // does call i send a round out?
    };
    static const Case cases[] = {
        // pieces 0 ... 15 grow one span to its cap, 16 ... 31 a second one; the channel then holds 2^17 samples: out with piece 31, 63
        {"hold rules: contiguous pieces merge", 16, 1, 0, true, piece, 70, [](int i) { return i % 32 == 31; }},
        // the library's depth for 8192-sample spans is 64; sixteen of them are 2^17 samples: out with the 16th, 32nd ... call
        {"hold rules: scattered spans, the library's depth", 16, 1, 0, false, span, 40, [](int i) { return i % 16 == 15; }},
        // an explicit depth of four: the round goes out with the call that cannot join it (the fifth, ninth ...), which is then held
        {"hold rules: four spans a round", 16, 0, 4, false, span, 14, [](int i) { return i > 0 && i % 4 == 0; }},
        // ... also when the spans continue each other in memory but merging is off
        {"hold rules: four spans a round, contiguous, no merging", 16, 0, 4, true, span, 14, [](int i) { return i > 0 && i % 4 == 0; }},
        // the library's caps: spans just long enough to be read in place are held 128 to a round (MAX_COALESCE; ~270 fused jobs in ONE
        // launch), which goes out with the 129th, 257th ... call
        {"hold rules: 128 short spans a round", 0, 1, 0, false, 2304, 300, [](int i) { return i > 0 && i % 128 == 0; }},
        // depth one: nothing is ever held
        {"hold rules: every span its own round", 16, 1, 1, false, span, 6, [](int) { return true; }},
    };
    for (const Case &cs : cases) {
        new_scenario(cs.what, 0);
        if (cs.hold_log2)
            setenv("PSDC_DBG_HOLD_LOG2", std::to_string(cs.hold_log2).c_str(), 1);
        psdc_handle *h = psdc_create(n, PSDC_WINDOW_HANN, 1, 0);
        unsetenv("PSDC_DBG_HOLD_LOG2");
        if (!h) {
            fail("psdc_create: %s", psdc_last_error(nullptr));
            continue;
        }
        CK(psdc_configure(h, PSDC_OPT_MERGE, cs.merge));
        if (cs.coalesce)
            CK(psdc_configure(h, PSDC_OPT_COALESCE, cs.coalesce));
        const std::vector<float> x = ident_stream(0, cs.len * (size_t)cs.calls);
        std::vector<std::vector<float>> own; // (kept alive until the sync: held spans are caller memory)
        for (int i = 0; i < cs.calls; ++i) {
            const float *p = x.data() + cs.len * (size_t)i;
            if (!cs.contiguous) {
                own.emplace_back(p, p + cs.len);
                p = own.back().data();
            }
            const long before = world().fused_enqueued;
            CK(psdc_process_device(h, 0, p, cs.len));
            const bool went = world().fused_enqueued != before;
            if (went != cs.goes_out(i))
                fail("call %d: %s", i, went ? "a round went out" : "no round went out");
        }
        CK(psdc_sync(h));
        verify(h, n, n / 2, 1, {(uint64_t)cs.len * (uint64_t)cs.calls}, true);
        psdc_destroy(h);
    }
}

// ---- failures on the way: an allocation that fails must leave a handle that still destroys cleanly ---------------------
void scenario_alloc_failure()
{
    new_scenario("allocation failure", 0);
    // the first device allocation of psdc_create fails: no handle, nothing leaked (checked at the end of main)
    sim::rt().fail_next_malloc = true;
    psdc_handle *h = psdc_create(1024, PSDC_WINDOW_HANN, 2, 0);
    if (h) {
        fail("psdc_create succeeded although its first allocation failed");
        psdc_destroy(h);
    }
    // the lazily built frame-scan state: a failed allocation must not leave a half-built state behind (round-3 advisor: the next
    // call then launched the verdict on a null stream with a null result pointer)
    h = psdc_create(1024, PSDC_WINDOW_HANN, 4, 0);
    std::vector<uint8_t> fr = make_frames(4, 3, 0);
    size_t ok = 0;
    sim::rt().fail_next_malloc = true;
    const int rc = psdc_process_adcdac_frames_device(h, fr.data(), 8 + 64 * 3, 4, &ok);
    if (rc != PSDC_ERR_DEVICE)
        fail("frames call with a failing scan-state allocation -> %d", rc);
    const int rc2 = psdc_process_adcdac_frames_device(h, fr.data(), 8 + 64 * 3, 4, &ok); // builds it now
    if (rc2 != PSDC_OK || ok != 4)
        fail("frames call after the failed one -> %d, %zu frames (%s)", rc2, ok, psdc_last_error(h));
    psdc_destroy(h);
    for (const std::string &e : world().errors)
        fail("model: %s", e.c_str());
}

} // namespace

int main(int argc, char **argv)
{
    const uint64_t first = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const uint64_t count = argc > 2 ? strtoull(argv[2], nullptr, 10) : 40;
    long fused = 0, pairs = 0, segs = 0, decs = 0, tails = 0, ftails = 0, fjobs = 0, launches = 0, multi = 0, maxrun = 0, aux = 0, one = 0, pro = 0;
    auto tally = [&] {
        fused += world().fused_jobs, pairs += world().fused_pairs, segs += world().seg_segments, decs += world().dec_jobs;
        tails += world().tail_jobs, ftails += world().tail_frame_jobs, fjobs += world().fused_frame_jobs, launches += world().fused_launches;
        multi += world().multi_block_jobs, maxrun = std::max(maxrun, world().max_run);
        aux += world().aux_launches, one += world().one_launch_rounds, pro += world().prologue_copies;
    };
    for (uint64_t s = first; s < first + count; ++s) {
        scenario_stress(s);
        tally();
        scenario_spans(s);
        tally();
        scenario_frames(s);
        tally();
        if (s % 4 == 0) {
            scenario_single(s);
            tally();
        }
    }
    scenario_hold_rules();
    scenario_alloc_failure();
    new_scenario("end", 0);
    if (sim::rt().live_blocks != 0) {
        fprintf(stderr, "FAIL %zu device / pinned allocations were never freed\n", sim::rt().live_blocks);
        ++g_failures;
    }
    printf("round_plan_check: seeds %llu..%llu: %ld fused launches, %ld fused jobs (%ld in several workgroups, longest run %ld) of %ld "
           "pairs, %ld generic segments, %ld decimator jobs, %ld copy jobs (%ld from frames); %ld launches with aux workgroups, %ld rounds of ONE "
           "launch, %ld seam copies as job prologues; %s\n",
           (unsigned long long)first, (unsigned long long)(first + count - 1), launches, fused, multi, maxrun, pairs, segs, decs, tails, ftails, aux, one, pro,
           g_failures ? "FAILED" : "every segment and every decimator output exactly once, every read inside its source");
    return g_failures ? 1 : 0;
}
