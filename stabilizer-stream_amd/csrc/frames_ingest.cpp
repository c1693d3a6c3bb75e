// frames_ingest.cpp -- frames in host or device memory (src/de/frame.rs, src/de/data.rs, src/loss.rs) into the cascades.
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

using namespace psdrt;

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
    // *n_ok is the number of frames ingested at EVERY exit, device errors in mid-call included; a piece's frames enter `good` and
    // Loss only once the piece is enqueued (its samples are in the streams), never before
    struct StoreOk {
        size_t *p;
        const size_t &v;
        ~StoreOk()
        {
            if (p)
                *p = v;
        }
    } store_ok{n_ok, good};
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
        if ((int)h->n_channels < wf->ntr)
            return fail(h, PSDC_ERR_ARG, "the frames carry more traces than the handle has channels");
        const int ntr = wf->ntr;
        const int batches = (int)(payload / wf->batch_bytes);
        bool run_end = false; // a frame of another (valid) format: the next run starts there
        psdc_loss trial = h->loss; // Loss::update over the piece being scanned: committed to the handle with the piece
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
                trial.received += f[3];
                if (trial.have_seq)
                    trial.dropped += (uint32_t)(seq - trial.next_seq); // wrapping_sub
                trial.next_seq = seq + f[3];                            // wrapping_add
                trial.have_seq = 1;
            }
            return cnt;
        };
        if (batches == 0) {
            const size_t cnt = scan(f0, n_frames - f0); // header-only frames carry no samples
            h->loss = trial;
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
            trial = h->loss;
            const size_t cnt = scan(f0, std::min(piece_frames, n_frames - f0));
            if (cnt == 0)
                break;
            const size_t bytes = cnt * frame_size;
            const int b = h->frames_cur;
            if (h->frames_ev_pending[b]) { // the bounce buffer's last upload must have left it
                HIPCHK(h, hipEventSynchronize(h->frames_ev[b]));
                h->frames_ev_pending[b] = false;
            }
            pinned_copy(h->h_frames[b], frames + f0 * frame_size, bytes);
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
            h->loss = trial; // the piece is in the streams: its frames count from here on
            good += cnt;
            rc = advance(h);
            if (rc)
                return rc;
            f0 += cnt;
        }
    }
    if (bad != PSDC_OK)
        return fail(h, bad,
                    bad == PSDC_ERR_FRAME_HEADER   ? "Invalid frame header"
                    : bad == PSDC_ERR_FRAME_FORMAT ? (adcdac_only ? "Unknown or non-AdcDac format ID" : "Unknown format ID")
                                                   : "Payload size");
    return PSDC_OK;
}
} // namespace

extern "C" {

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
    // Loss::update over the accepted frames (src/loss.rs:11-26): summed on the device above, committed to the handle below once
    // every piece is enqueued -- a device error in mid-call leaves Loss as it was and reports the frames enqueued so far in *n_ok
    psdc_loss trial = h->loss;
    if (good) {
        trial.received += res[1];
        const uint32_t seq0 = (uint32_t)res[3], next = (uint32_t)(res[3] >> 32);
        if (trial.have_seq)
            trial.dropped += (uint32_t)(seq0 - trial.next_seq); // wrapping_sub
        trial.dropped += res[2];
        trial.next_seq = next;
        trial.have_seq = 1;
    }
    size_t enq = 0; // frames whose samples are in the streams
    struct StoreOk {
        size_t *p;
        const size_t &v;
        ~StoreOk()
        {
            if (p)
                *p = v;
        }
    } store_ok{n_ok, enq};
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
            // frames that CONTINUE the run the four traces hold back (a capture ring handed over piece by piece) extend those spans:
            // no seam, whatever the call size (PSDC_OPT_MERGE, as psdc_process_device)
            if (in_place_ok && h->merge) {
                bool can = true;
                for (int ci = 0; ci < 4 && can; ++ci) {
                    const Channel &c = h->ch[ci];
                    can = c.has_span() && c.fill == 0 && !c.submitted && c.spans.back().framed() && c.spans.back().fch == ci &&
                          c.spans.back().fs.frames + c.spans.back().fs.bytes == piece && c.spans.back().fs.frame_size == (unsigned)frame_size &&
                          c.spans.back().fs.batches == (unsigned)batches && c.spans.back().fs.frames == h->ch[0].spans.back().fs.frames &&
                          c.spans.back().len + per_ch <= (size_t)FSPAN_MAX_SAMPLES && held_samples(c) + per_ch <= hold_max(h);
                }
                if (can) {
                    for (int ci = 0; ci < 4; ++ci) {
                        Channel &c = h->ch[ci];
                        DeviceSpan &last = c.spans.back();
                        last.fs.bytes += (unsigned long long)cnt * frame_size;
                        last.len += per_ch;
                        c.st[0].total += per_ch;
                        c.span_max = std::max(c.span_max, last.len);
                    }
                    h->idle = false;
                    enq += cnt;
                    if (device_idle(h)) {
                        rc = advance(h);
                        if (rc)
                            return rc;
                    }
                    continue;
                }
            }
            if (in_place_ok && per_ch >= (size_t)4 * (h->n + HBF_HALO)) {
                // The four traces are read IN PLACE, as wire words, by the stage-0 loads of the fused kernel: a zero-copy span
                // per trace, exactly like psdc_process_device's -- held back so that calls share rounds (the same rule: runtime.cpp coalesce_limit).
                bool flush = false;
                for (int ci = 0; ci < 4; ++ci) {
                    Channel &c = h->ch[ci];
                    flush = flush || c.submitted || c.fill > 0 || c.spans.size() >= h->coalesce || holds_short_span(h, c);
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
                enq += cnt; // (registered: the spans are part of the streams now)
                // held until PSDC_OPT_COALESCE calls share the round (an eager handle: until the device is seen idle)
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
            enq += cnt;
            rc = advance(h);
            if (rc)
                return rc;
        }
    }
    enq = good; // (batches == 0: header-only frames carry no samples and count all the same)
    h->loss = trial;
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
    struct StoreOk { // *n_ok = the frames ingested, at every exit (device errors in mid-call included)
        size_t *p;
        const size_t &v;
        ~StoreOk()
        {
            if (p)
                *p = v;
        }
    } store_ok{n_ok, good};
    auto done = [&](int code) {
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
        if ((int)h->n_channels < wf->ntr)
            return fail(h, PSDC_ERR_ARG, "the frames carry more traces than the handle has channels");
        if (wf->id == 1) { // a run of AdcDac frames: its own entry point checks them (and counts their Loss) on the device
            size_t run = 1;
            while (f0 + run < n_frames && hdr[8 * (f0 + run)] == 0x7b && hdr[8 * (f0 + run) + 1] == 0x05 && hdr[8 * (f0 + run) + 2] == 1)
                ++run;
            size_t ok = 0;
            rc = psdc_process_adcdac_frames_device(h, d_frames + f0 * frame_size, frame_size, run, &ok);
            good += ok;
            if (rc)
                return rc;
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
            psdc_loss trial = h->loss; // committed with the piece, once it is enqueued
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
                trial.received += f[3];
                if (trial.have_seq)
                    trial.dropped += (uint32_t)(seq - trial.next_seq); // wrapping_sub
                trial.next_seq = seq + f[3];                            // wrapping_add
                trial.have_seq = 1;
            }
            if (cnt == 0)
                break;
            if (batches == 0) { // header-only frames carry no samples
                h->loss = trial;
                good += cnt;
            } else {
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
                h->loss = trial;
                good += cnt;
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

} // extern "C"
