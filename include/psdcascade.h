/*
 * psdcascade.h -- C ABI of the MI355X-native cascaded PSD estimator.
 *
 * Drop-in boundary for the hot path of quartiq/stabilizer-stream src/psd.rs
 * (`Psd<N>` / `PsdCascade<N>`).  The reference has no FFI of its own; the
 * boundary is the Rust type surface its binaries use.
 * Each entry point below names the reference item it replaces (file:line under
 * the reference checkout).  INTEGRATION.md shows the Rust `extern "C"` shim and
 * the `PsdCascade<N>` wrapper a maintainer would add on the reference side.
 *
 * Conventions
 *   - opaque handle, plain pointers and sizes, int status: 0 ok, <0 error
 *     (PSDC_ERR_*).  Nothing unwinds across the ABI.  Where the reference
 *     panics on misuse (assert!/unimplemented!, src/psd.rs:110,138,139,247) the
 *     call returns an error and `psdc_last_error` describes it; the Rust shim
 *     turns that back into panic!.
 *   - a handle is used from one thread at a time (like `&mut self`); it may be
 *     moved between threads (Send, not Sync).  Distinct handles are independent.
 *   - one handle holds `n_channels` independent cascades (one `PsdCascade` per
 *     trace, src/bin/psd.rs:174-182) that are batched onto one GPU.
 *   - all sample data is IEEE f32, native endian (src/bin/stream_to_raw.rs:24-25).
 *   - there is NO CPU fallback: without a usable HIP device `psdc_create` fails.
 *   - call chunking: like the reference (src/psd.rs:196-208) every result is a function of the CONCATENATED stream of a
 *     channel.  Counters (stage count, count, pending, processed, Break fields, frequencies, Loss) are exactly independent of
 *     how the stream was cut into process() calls.  Spectra are independent of it only to rounding: the reference adds
 *     segment after segment into one f32 accumulator and is bit-identical under any chunking; here the segments a call brings
 *     are summed in groups (per workgroup run in f32, across runs in f64, one f32 add into the accumulator per round) whose
 *     boundaries follow the calls, so two chunkings of one stream agree to <= 2e-6 relative per bin (asserted by
 *     tests/test_gpu_parity.py::test_chunking_invariance).  The same stream fed by the same CALLS is bit-reproducible, run to
 *     run and whatever the host's timing (tests/test_gpu_parity.py::test_same_calls_same_bits: the same spans with sleeps
 *     injected between the calls): which device spans share a round -- hence the grouping of the sums -- is decided by the call
 *     sequence alone (PSDC_OPT_COALESCE), never by how busy the device happens to be.  Only a handle switched to
 *     PSDC_OPT_EAGER gives that up: its held spans go out as soon as the device is seen idle, so its round composition
 *     follows host timing and repeated runs agree to the same <= 2e-6, not to the bit (host-fed samples and one-span feeds
 *     stay bit-reproducible there too).
 *     Either grouping is closer to the exact sum than the reference's sequential f32 accumulation.
 */
#ifndef PSDCASCADE_H
#define PSDCASCADE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSDC_ABI_VERSION 3

/* status codes */
#define PSDC_OK 0
#define PSDC_ERR_ARG (-1)           /* bad argument (reference: assert!/index panic) */
#define PSDC_ERR_DEVICE (-2)        /* HIP error / no device */
#define PSDC_ERR_NOMEM (-3)
#define PSDC_ERR_UNIMPLEMENTED (-4) /* Detrend::Linear: unimplemented!() src/psd.rs:110 */
#define PSDC_ERR_FRAME_HEADER (-5)  /* de::Error::InvalidHeader  src/de/frame.rs:27-29 */
#define PSDC_ERR_FRAME_FORMAT (-6)  /* de::Error::UnknownFormat  src/de/frame.rs:30 */
#define PSDC_ERR_FRAME_SIZE (-7)    /* de::Error::PayloadSize / batches mismatch src/de/data.rs:23-24 */
#define PSDC_ERR_CAPACITY (-8)      /* caller-provided output too small */

/* Window<N> constructors (src/psd.rs:24-32, :42-55) */
#define PSDC_WINDOW_RECTANGULAR 0
#define PSDC_WINDOW_HANN 1
#define PSDC_WINDOW_CUSTOM 2 /* a caller-built Window<N> (pub fields, src/psd.rs:12-20): psdc_create_window */

/* `device` argument of the constructors: a HIP device index, or PSDC_DEVICE_DEFAULT = the index in the
 * environment variable PSDC_DEVICE (0 when unset) -- what a shim whose constructor has no device argument
 * (PsdCascade::<N>::default(), src/psd.rs:408) passes, so that one process per GPU selects its device from outside. */
#define PSDC_DEVICE_DEFAULT (-1)

/* Detrend (src/psd.rs:59-72) */
#define PSDC_DETREND_NONE 0
#define PSDC_DETREND_MIDPOINT 1
#define PSDC_DETREND_SPAN 2
#define PSDC_DETREND_MEAN 3
#define PSDC_DETREND_LINEAR 4 /* accepted by the enum, rejected like the reference */

/* DEPTH (src/psd.rs:117): each stage decimates by 1 << 3 */
#define PSDC_DEPTH 3

/* options for psdc_configure */
#define PSDC_OPT_QUANTUM 1 /* host-fed samples buffered per channel before a launch (default 1<<22) */
#define PSDC_OPT_COALESCE 3 /* in-place device spans of a channel that share one round (1..16; default: 8 -- more of short spans,
                             * so that a round of all channels holds ~2^28 samples and stays one launch -- and at most 2^29 samples a channel; a handle of ONE channel: 16 and at most 2^30 samples, and of f32 spans shorter than 2^24 samples
                             * as many as make a round of ~2^28 samples, up to 128 -- a round costs 5 ... 20 us whatever it holds;
                             * 1 = every span its own round).  A span is HELD until its channel holds that many samples or a call arrives
                             * that cannot join them (one span more than that many, host-fed or short spans, settings changes, every
                             * read-out, psdc_flush, psdc_sync, psdc_record_consumed): which spans share a round depends on the calls alone, so results are
                             * bit-reproducible.  Held spans are caller memory the library has not read yet: the rule of
                             * psdc_process_device (unmodified until sync / read-out / consumed event) covers them.
                             * (-k is accepted and means k: through ABI 3's first builds it asked for exactly this hold.) */
#define PSDC_OPT_MERGE 6 /* 1 (default): a device span that starts where the last HELD span of its channel ends (d_x == previous d_x +
                          * previous len: a ring or capture buffer handed over piece by piece) extends that span instead of becoming
                          * one of its own -- no seam between them, whatever the call size; a span stops growing at 2^29 samples.
                          * 0: every call is a span of its own (tests of the multi-span planner). */
#define PSDC_OPT_EAGER 5 /* 1: a held span also goes out as soon as the device is seen idle (hipStreamQuery) -- the first span of a
                          * burst starts at once instead of waiting for its round to fill, at the price of a round composition
                          * that follows host timing: repeated runs then agree to rounding (<= 2e-6), not to the bit.  Default 0. */
#define PSDC_OPT_PROFILE 2 /* 1: time the dominant kernel with HIP events (psdc_profile_read) */
#define PSDC_OPT_MIN_PAIRS 4 /* segment pairs a decimated stage (k >= 1) collects before it issues work on the ingest
                              * path (default 32 x teams per workgroup: 256 at n = 1024; 0 = issue at once).  Read-outs,
                              * psdc_flush and psdc_sync always issue everything: results do not depend on it. */

typedef struct psdc_handle psdc_handle;

/* Break (src/psd.rs:290-311); `bins: Range<usize>` is flattened. */
typedef struct psdc_break {
    uint64_t start;      /* start index in PSD and frequencies */
    uint32_t include;    /* was included in output */
    uint32_t count;      /* number of averages */
    uint32_t avg;        /* averaging limit */
    uint32_t _pad;
    uint64_t bins_start; /* FFT bins [bins_start, bins_end) */
    uint64_t bins_end;
    uint64_t fft_size;
    uint64_t decimation;
    uint64_t pending;    /* unprocessed input samples (includes overlap) */
    uint64_t processed;  /* total samples processed (excluding overlap) */
} psdc_break;

/* PsdStage accessors (src/psd.rs:271-287) + Break bookkeeping in one record */
typedef struct psdc_stage_stat {
    uint32_t count; /* PsdStage::count  src/psd.rs:275-277; saturates at u32::MAX where the reference's
                     * u32 wraps after 2^32 segments (the library counts in 64 bits; gain() follows that) */
    uint32_t avg;   /* Psd::avg         src/psd.rs:133 */
    uint64_t pending;   /* PsdStage::buf().len()  src/psd.rs:285-287 */
    uint64_t processed; /* src/psd.rs:511-512 */
} psdc_stage_stat;

typedef struct psdc_profile {
    uint64_t launches;      /* dominant-kernel launches bracketed so far */
    double kernel_ms;       /* sum of their HIP-event durations */
    uint64_t samples;       /* input samples those launches consumed (all stages) */
    uint64_t stage0_samples;/* of which stage-0 (raw stream) samples */
} psdc_profile;

/* ---- lifecycle ----------------------------------------------------------- */

/* PsdCascade::<N>::default() (src/psd.rs:408-423) for `n_channels` traces on HIP
 * device `device`.  n: a power of two 16 ... 131072, or ANY size 16 < n <= 8192 (the reference takes any
 * N >= 2 with (N - overlap) % 8 == 0, src/psd.rs:138,247 -- rustfft plans any length, :418; with the Hann
 * window's overlap N/2 that is every multiple of 16).  Returns NULL on failure; psdc_last_error(NULL)
 * explains.
 * Which kernels run: the single-pass fused kernels (stream read once: detrend +
 * window + FFT + |X|^2 + /8 decimator in one launch) exist for n = 256 ... 16384 (powers of two) and every window
 * whose overlap is n / 2 -- the HANN window (what the reference's binaries and BASELINE configs use) and caller-built
 * tables with that overlap; they read the table and assume only the hop -- or 0: Window::rectangular() and caller-built
 * tables without overlap, where two disjoint segments share one transform (faster than the Hann path: half the FFT work).
 * Caller-built windows of another overlap, n < 256 and sizes that are not powers of two take the generic
 * two-pass kernels (welch + hbf_dec8: same results, the stream is read twice, about half the rate;
 * sizes that are not powers of two evaluate the DFT in chirp-z form on a power-of-two transform of at
 * least twice the length: two such transforms per segment pair); the powers of two 32768 ... 131072 a four-step FFT with
 * one intermediate frame in device memory (100 ... 120 GS/s, +256 MiB per handle). */
psdc_handle *psdc_create(uint32_t n, int window_kind, uint32_t n_channels, int device);

/* The same with a caller-supplied `Window<N>` -- the struct is public with public fields `win`, `power`,
 * `nenbw`, `overlap` (src/psd.rs:12-20) and `Psd::new(fft, win)` takes any (src/psd.rs:137) --: `win` holds n
 * f32 weights (host memory, copied), `power` / `nenbw` feed gain() (src/psd.rs:279-283), `overlap` the segment
 * hop.  The reference asserts (n - overlap) % 8 == 0 when the first segment is decimated (src/psd.rs:246-247);
 * here that -- and overlap < n -- is checked at construction (PSDC_ERR_ARG through psdc_last_error(NULL)).
 * A table that compares equal, bit for bit and in its three constants, to Window::hann() or
 * Window::rectangular() is recognised as such; any table with overlap == n / 2 (Hann, a caller's Hamming,
 * Blackman, ...) runs the single-pass fused kernels, any other overlap the generic two-pass kernels (same
 * results, about half the rate). */
psdc_handle *psdc_create_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap,
                                uint32_t n_channels, int device);

/* The Window of a handle as the library holds it: kind (PSDC_WINDOW_*), constants, and the n weights
 * (win may be NULL).  What a gather of raw spectra needs beside them to run the stitch elsewhere. */
int psdc_window_get(const psdc_handle *h, int *kind, float *power, float *nenbw, size_t *overlap, float *win);

/* Window::hann() / Window::rectangular() (src/psd.rs:24-55) as the library builds them: n weights + constants
 * (pure host; any n >= 2).  For callers that want to derive a table from them or compare. */
int psdc_window_table(uint32_t n, int window_kind, float *win, float *power, float *nenbw, size_t *overlap);

/* Drop (src/bin/psd.rs:190 `dec.clear()`). */
void psdc_destroy(psdc_handle *h);

/* #[derive(Clone)] (src/psd.rs:399): deep copy of every channel's state. */
psdc_handle *psdc_clone(psdc_handle *h);

/* Cmd::Reset (src/bin/psd.rs:190): forget all stages of all channels, keep settings. */
int psdc_reset(psdc_handle *h);

int psdc_configure(psdc_handle *h, int option, int64_t value);

/* ---- settings ------------------------------------------------------------ */

/* PsdCascade::set_detrend (src/psd.rs:438-443); applies to segments completed
 * after the call (pending complete segments are flushed first). */
int psdc_set_detrend(psdc_handle *h, int detrend_kind);

/* PsdCascade::set_avg(AvgOpts{limit, count}) (src/psd.rs:431-436); stage i uses
 * min(count >> (3 i), limit). */
int psdc_set_avg(psdc_handle *h, uint32_t limit, uint32_t count);

/* ---- ingest -------------------------------------------------------------- */

/* PsdCascade::process(&[f32]) (src/psd.rs:456-468) for one channel.  `x` is host
 * memory and is copied before returning.  GPU work may be deferred until
 * PSDC_OPT_QUANTUM samples are buffered or a read-out/flush happens; results
 * depend only on the concatenated stream, not on call chunking (counters exactly, spectra to rounding: Conventions). */
int psdc_process(psdc_handle *h, uint32_t channel, const float *x, size_t len);

/* Same, but `d_x` is device memory on the handle's device and is read in place
 * (no staging copy).  The work is enqueued asynchronously on the handle's own stream, possibly
 * after this call returns (PSDC_OPT_COALESCE): whatever produced `d_x` must have COMPLETED before
 * the call (host-synchronised; or use psdc_process_device_after), and `d_x` must stay valid and
 * unmodified until psdc_sync()/any read-out returns or a psdc_record_consumed event completes. */
int psdc_process_device(psdc_handle *h, uint32_t channel, const float *d_x, size_t len);

/* psdc_process_device for a producer that runs on ANOTHER HIP stream (a decode kernel, torch's
 * current stream, ...).  The handle works on a private non-blocking stream and, with
 * PSDC_OPT_COALESCE, may not even have enqueued the kernels that read `d_x` when the call
 * returns, so plain psdc_process_device requires that the producer has FINISHED (host-synchronised)
 * before the call.  Here `producer_event` -- a hipEvent_t the caller recorded behind the work that
 * writes d_x, passed as void*; NULL = none -- is waited for on the device by the handle's stream
 * before anything reads the span: no host synchronisation. */
int psdc_process_device_after(psdc_handle *h, uint32_t channel, const float *d_x, size_t len,
                              void *producer_event);

/* The other direction: enqueue everything held so far and record `consumed_event` (hipEvent_t as
 * void*) on the handle's stream.  When it has completed, every span handed to psdc_process_device*
 * before this call has been read for the last time and its memory may be overwritten (the producer
 * waits for it with hipStreamWaitEvent).  Does not wait on the host. */
int psdc_record_consumed(psdc_handle *h, void *consumed_event);

/* Frame::from_bytes + AdcDac::traces (src/de/frame.rs:49-60, src/de/data.rs:11-82)
 * + process() of the four traces ADC0, ADC1, DAC0, DAC1 into channels 0..3
 * (src/bin/psd.rs:174-182), for `n_frames` frames of `frame_size` bytes each,
 * as read by Source::get for Data::File (src/source.rs:135-142).  Host memory.
 * Headers are validated on the host; payloads are de-interleaved on the device.
 * On a bad frame: frames before it are ingested, *n_ok says how many, and the
 * frame's de::Error is returned.  Needs n_channels >= 4. */
int psdc_process_adcdac_frames(psdc_handle *h, const uint8_t *frames, size_t frame_size,
                               size_t n_frames, size_t *n_ok);

/* Frame::from_bytes + Payload::traces for ANY of the reference's four payload formats (src/de/mod.rs:12-17,
 * src/de/frame.rs:49-60, src/de/data.rs): AdcDac (id 1: traces ADC0, ADC1, DAC0, DAC1, eight samples per batch), Fls (2: AR, AP,
 * BI, BQ), ThermostatEem (3: T00, T20, I0, I1), Mpll (4: "phase (rad)", "frequency (kHz)", "amplitude (V/G10)") -- one sample
 * per batch for the last three --, + Loss::update + process() of trace i into channel i (src/bin/psd.rs:174-182: the trace's
 * index picks the cascade, whatever the frame's format), for `n_frames` frames of `frame_size` bytes each in host memory.
 * The format is each frame's own (header byte 2); the frames of a call are taken in runs of one format, headers validated on
 * the host, payloads decoded on the device -- the decoded traces are bit-identical to Payload::traces (same f32 operations in
 * the same order).  On a bad frame: frames before it are ingested, *n_ok says how many, and the frame's de::Error is returned
 * (PSDC_ERR_FRAME_HEADER / _FORMAT / _SIZE).  Needs n_channels >= the traces of every format met (4, 4, 4, 3), else
 * PSDC_ERR_ARG at the first frame that carries more.  (The reference CLI's default --frame-size 1448 is 60 Mpll batches,
 * src/source.rs:31.)  psdc_process_adcdac_frames is this call restricted to AdcDac. */
int psdc_process_frames(psdc_handle *h, const uint8_t *frames, size_t frame_size, size_t n_frames, size_t *n_ok);

/* The same for frames that already sit in device memory (a capture buffer filled by a NIC / another kernel).
 * Headers: Header::parse and the AdcDac size checks of every frame plus the Loss sums are ONE small launch on a side stream
 *   of the handle; the call waits for that launch alone, so `*n_ok` and the returned de::Error are final when it returns.
 * Payloads: at every size with a fused kernel (powers of two 256 ... 16384, windows with overlap n/2) the four traces are
 *   read IN PLACE, as wire words, by the stage-0 loads of the fused kernels -- the f32 traces never exist in memory --, under
 *   the same rules as psdc_process_device: spans may be held back and share a round with later calls (PSDC_OPT_COALESCE), and
 *   the seam / tail of a span is read by the FIRST launch of the next round, i.e. possibly after this call AND the next one
 *   have returned.  Elsewhere (other sizes and windows, pieces shorter than 4 (n + 288) samples per trace) a decode kernel
 *   writes the four traces into the stage-0 stream buffers.
 * Lifetime: d_frames must stay valid and its PAYLOAD bytes unmodified until psdc_sync(), any read-out, or an event from
 *   psdc_record_consumed has completed.  The 8 header bytes of each frame are read by this call's verdict launch only, which
 *   has completed when the call returns: a ring that re-stamps `seq` words (or a replay that moves them on, bench.py
 *   FrameReplay) may rewrite headers as soon as the call is back
 *   (tests/test_gpu_frames_inplace.py::test_headers_may_change_once_the_call_has_returned).  Ordering: the handle works on its own non-blocking streams and there is no implicit null-stream order:
 *   the producer of d_frames must have COMPLETED before the call (or use psdc_process_device_after's event for f32 spans).
 * Alignment: any base address is accepted.  The in-place path needs d_frames to be a multiple of 8 bytes (frame_size =
 *   8 + 64 batches keeps every later frame aligned); other bases take the byte-wise decode kernel -- same results, slower. */
int psdc_process_adcdac_frames_device(psdc_handle *h, const uint8_t *d_frames, size_t frame_size,
                                      size_t n_frames, size_t *n_ok);

/* psdc_process_frames for frames that already sit in device memory: any of the four formats, in runs.  The headers (8 of every
 * frame_size bytes) are gathered into pinned memory by one small kernel on a side stream and validated on the host; the payloads stay on the device -- Fls /
 * ThermostatEem / Mpll runs are decoded straight from `d_frames` into the stage-0 streams, AdcDac runs take
 * psdc_process_adcdac_frames_device (whose lifetime, ordering and alignment rules apply to the whole call: d_frames valid and
 * unmodified until psdc_sync() / a read-out / a psdc_record_consumed event, its producer COMPLETED before the call). */
int psdc_process_frames_device(psdc_handle *h, const uint8_t *d_frames, size_t frame_size, size_t n_frames, size_t *n_ok);

/* Loss (src/loss.rs:3-26): sequence-gap accounting over the frames ingested by
 * psdc_process_frames / psdc_process_adcdac_frames[_device].  received counts batches; dropped the batches missing
 * between consecutive frames (u32 wrapping_sub); gaps are counted, never zero-filled. */
typedef struct psdc_loss {
    uint64_t received;
    uint64_t dropped;
    uint32_t next_seq;  /* seq expected from the next frame */
    uint32_t have_seq;  /* 0 until the first frame was seen */
} psdc_loss;

int psdc_loss_read(psdc_handle *h, psdc_loss *out, int reset);

/* Enqueue every complete segment of every stage now (does not wait). */
int psdc_flush(psdc_handle *h);

/* Wait until all enqueued device work of this handle has finished. */
int psdc_sync(psdc_handle *h);

/* ---- read-out (each implies flush + sync) -------------------------------- */

/* PsdCascade.stages.len() (src/psd.rs:401,445-453): stages are created when
 * the first sample reaches them. */
int psdc_num_stages(psdc_handle *h, uint32_t channel);

int psdc_stage_info(psdc_handle *h, uint32_t channel, uint32_t stage, psdc_stage_stat *out);

/* PsdStage::spectrum (src/psd.rs:271-273): n/2+1 un-normalised accumulators. */
int psdc_stage_spectrum(psdc_handle *h, uint32_t channel, uint32_t stage, float *out);

/* PsdStage::gain (src/psd.rs:279-283). */
int psdc_stage_gain(psdc_handle *h, uint32_t channel, uint32_t stage, float *out);

/* PsdStage::buf (src/psd.rs:285-287): the pending input samples of a stage. */
int psdc_stage_buf(psdc_handle *h, uint32_t channel, uint32_t stage, float *out, size_t cap,
                   size_t *len);

/* Bulk read-out of one channel: stage count, per-stage records and the raw accumulators of
 * every stage (stage 0 first, n/2+1 floats each) with one flush, one sync and one copy.
 * stats / spectra may be NULL; cap = stages the caller has room for.  What a multi-GPU
 * gather or a GUI refresh (src/bin/psd.rs:201) needs per trace. */
int psdc_read_channel(psdc_handle *h, uint32_t channel, uint32_t cap, uint32_t *n_stages,
                      psdc_stage_stat *stats, float *spectra);

/* PsdCascade::psd(&MergeOpts) (src/psd.rs:479-543).  psd_out needs room for
 * num_stages*(n/2+1) floats, breaks for num_stages records (lowest rate first).
 * Either output pointer may be NULL to query sizes only. */
int psdc_psd(psdc_handle *h, uint32_t channel, int keep_overlap, uint32_t min_count,
             int keep_transition_band, float *psd_out, size_t psd_cap, size_t *psd_len,
             psdc_break *breaks, size_t breaks_cap, size_t *n_breaks);

/* PsdCascade::rbw (src/psd.rs:427-429). */
float psdc_rbw(const psdc_handle *h);

/* ---- Psd<N>: one stage (src/psd.rs:122-288) ------------------------------ */

/* `Psd<N>` with its `PsdStage` trait (src/psd.rs:163-193), the type the reference's own test
 * drives directly (src/psd.rs:615-632).  Same kernels as the cascade: one stage analyses the
 * stream, and the /8-decimated stream it emits is handed back instead of feeding a next stage. */
typedef struct psdc_stage psdc_stage;

/* Psd::new(fft, win) (src/psd.rs:137-152): the FFT plan is the library's own (n as in psdc_create),
 * window_kind one of PSDC_WINDOW_*; detrend None, avg = u32::MAX, drain = hbf_dec_response_length(3). */
psdc_stage *psdc_stage_create(uint32_t n, int window_kind, int device);
/* Psd::new(fft, win) (src/psd.rs:137-152) with the caller's Window<N> (see psdc_create_window); the FFT plan
 * argument of the reference has no counterpart: the library's own FFT of length n is used, and the shim keeps the
 * reference's assert_eq!(N, fft.len()) (src/psd.rs:139) on its side. */
psdc_stage *psdc_stage_create_window(uint32_t n, const float *win, float power, float nenbw, size_t overlap,
                                     int device);
void psdc_stage_destroy(psdc_stage *s);
psdc_stage *psdc_stage_clone(psdc_stage *s); /* #[derive(Clone)] src/psd.rs:122 */
int psdc_stage_set_avg(psdc_stage *s, uint32_t avg);             /* Psd::set_avg      src/psd.rs:154-156 */
int psdc_stage_set_detrend(psdc_stage *s, int detrend_kind);      /* Psd::set_detrend  src/psd.rs:158-160 */

/* PsdStage::process(x, y) -> &mut y[..n] (src/psd.rs:196-269): buffers x, completes every full
 * segment (detrend, window, FFT, accumulate), decimates the samples new to each segment by 8 and
 * writes the outputs -- minus the one-time drain of hbf_dec_response_length(3) -- to y; *n_out is
 * the length of the returned slice.  It depends only on the samples fed so far: after T samples in
 * total, (N + (J-1)(N-overlap))/8 - 35 outputs have been returned, J the segments completed.  cap <
 * that many new outputs fails with PSDC_ERR_CAPACITY where the reference panics on the slice index
 * (src/psd.rs:253).  x and y are host memory. */
int psdc_stage_process(psdc_stage *s, const float *x, size_t len, float *y, size_t cap, size_t *n_out);

/* The same with x and y in device memory (y complete and x free to reuse on return). */
int psdc_stage_process_device(psdc_stage *s, const float *d_x, size_t len, float *d_y, size_t cap,
                              size_t *n_out);

int psdc_stage_get_spectrum(psdc_stage *s, float *out /* n/2+1 */); /* PsdStage::spectrum src/psd.rs:271-273 */
int psdc_stage_get_count(psdc_stage *s, uint32_t *count);           /* PsdStage::count    src/psd.rs:275-277 */
int psdc_stage_get_gain(psdc_stage *s, float *gain);                /* PsdStage::gain     src/psd.rs:279-283 */
int psdc_stage_get_buf(psdc_stage *s, float *out, size_t cap, size_t *len); /* PsdStage::buf src/psd.rs:285-287 */
const char *psdc_stage_last_error(const psdc_stage *s);

/* ---- pure host helpers (no device needed) -------------------------------- */

/* Break::frequencies (src/psd.rs:315-327).  Returns the number written, or the
 * number required if out is NULL / cap too small. */
size_t psdc_frequencies(const psdc_break *breaks, size_t n_breaks, float *out, size_t cap);

/* idsp::hbf::hbf_dec_response_length(depth) (src/psd.rs:149,622). */
int psdc_hbf_response_length(int depth);

/* The stitch of PsdCascade::psd (src/psd.rs:479-543) on caller-provided stage
 * data: spectra is n_stages rows of (n/2+1) floats, stage 0 (highest rate)
 * first; window_kind selects nenbw/power/overlap.  Used by psdc_psd and by
 * multi-GPU read-out after a gather of raw spectra. */
int psdc_stitch(uint32_t n, int window_kind, uint32_t n_stages, const uint32_t *counts,
                const uint32_t *avgs, const uint64_t *pendings, const float *spectra,
                int keep_overlap, uint32_t min_count, int keep_transition_band, float *psd_out,
                size_t psd_cap, size_t *psd_len, psdc_break *breaks, size_t breaks_cap,
                size_t *n_breaks);

/* The same stitch with the window given by its constants (any Window<N>) and the counts in 64 bits: a handle
 * counts segments in 64 bits where the reference's u32 wraps (psdc_stage_stat.count saturates), and psd() divides
 * by gain() of the 64-bit count; a gathered read-out must do the same to equal single-GPU psd() past 2^32
 * segments.  `counts` (u32, as reported) fills Break.count / Break.processed. */
int psdc_stitch_window(uint32_t n, float power, float nenbw, size_t overlap, uint32_t n_stages,
                       const uint64_t *counts64, const uint32_t *avgs, const uint64_t *pendings,
                       const float *spectra, int keep_overlap, uint32_t min_count, int keep_transition_band,
                       float *psd_out, size_t psd_cap, size_t *psd_len, psdc_break *breaks, size_t breaks_cap,
                       size_t *n_breaks);

/* ---- read-out for a gather (multi-GPU, src/bin/psd.rs:174-182 one cascade per trace) ------------------
 * One trace per cascade shards by channel: every GPU (one process per GPU, or one process with one handle per
 * device) runs whole cascades and nothing is exchanged during ingest.  At read-out every shard packs its raw
 * accumulators and counters into a flat, fixed-size byte record -- psdc_readout_bytes(n, n_channels) bytes,
 * the same on every shard with the same n and channel count, so one all-gather / gather of equal blocks over
 * ANY transport (RCCL ncclAllGather on device copies, MPI, a socket, or plain memcpy between the handles of one
 * process) collects them -- and the receiver stitches any channel of any record with psdc_unpack_stitch:
 * bit-identical to psdc_psd on the shard itself (raw accumulators travel, normalisation happens after).
 * A record is UNTRUSTED input to the psdc_unpack_* calls: every header field is held to the range the library can produce
 * (2 <= n <= 131072, n_channels <= 4096, overlap < n, power and nenbw > 0, stage counts <= 16) and the length to what those
 * fields imply, before anything is indexed; a record that fails is PSDC_ERR_ARG, never an out-of-bounds read.
 * psdc_readout_bytes returns 0 for dimensions outside those ranges (no such record exists). */
size_t psdc_readout_bytes(uint32_t n, uint32_t n_channels);
/* flush + sync + copy: fills `buf` (cap >= psdc_readout_bytes(n, n_channels) of this handle) */
int psdc_pack_readout(psdc_handle *h, void *buf, size_t cap, size_t *len);
/* The same record built from stage data the caller holds (pure host): psdc_pack_init writes the header of an empty
 * record of n_channels channels (no stages), psdc_pack_channel fills one channel (stage 0 first; spectra =
 * n_stages rows of n/2+1 floats).  psdc_pack_readout is these two over a handle's own state. */
int psdc_pack_init(void *buf, size_t cap, uint32_t n, float power, float nenbw, size_t overlap, uint32_t n_channels);
int psdc_pack_channel(void *buf, size_t len, uint32_t channel, uint32_t n_stages, const uint64_t *counts64,
                      const uint32_t *avgs, const uint64_t *pendings, const float *spectra);
/* A record copied into a larger one of `n_channels` channels, the added ones empty (no stages): a gather moves equal blocks,
 * so with channels % world != 0 every shard pads its record to the largest shard's channel count.  Pure host; the counters stay
 * the 64-bit ones.  cap >= psdc_readout_bytes(n, n_channels); out must not alias rec. */
int psdc_pack_pad(const void *rec, size_t len, void *out, size_t cap, uint32_t n_channels);
/* record header: FFT size, channels in the record, stages of `channel` (pure host) */
int psdc_unpack_info(const void *buf, size_t len, uint32_t channel, uint32_t *n, uint32_t *n_channels,
                     uint32_t *n_stages);
/* PsdCascade::psd (src/psd.rs:479-543) of one channel of a packed record (pure host; outputs as psdc_psd) */
int psdc_unpack_stitch(const void *buf, size_t len, uint32_t channel, int keep_overlap, uint32_t min_count,
                       int keep_transition_band, float *psd_out, size_t psd_cap, size_t *psd_len,
                       psdc_break *breaks, size_t breaks_cap, size_t *n_breaks);

/* Stream bookkeeping of src/psd.rs:196-269 in closed form: after `total`
 * samples have entered stage 0, how many stages exist and, per stage, samples
 * received, segments completed (= count when averaging is unbounded) and
 * pending samples.  Arrays hold up to `cap` stages. Returns the stage count. */
int psdc_plan_counts(uint32_t n, int window_kind, uint64_t total, uint32_t cap,
                     uint64_t *received, uint64_t *segments, uint64_t *pending);

/* Var::eval (src/var.rs:26-45) on a merged PSD (host, f32). */
float psdc_var_eval(int x_exp, int sinx_exp, float clip, size_t dc_cut, const float *phase_psd,
                    const float *frequencies, size_t n, float tau);

/* Trace::plot (src/bin/psd.rs:125-157) on a merged PSD: the Trapezoidal integrator for irregular
 * sampling (src/bin/psd.rs:98-116) runs over (frequencies[i], psd[i]); *rms = sqrt of the integral over
 * the bins with integral_start <= fs * f <= integral_end (no interpolation at the limits, like the
 * reference); the plot points -- for every bin whose f is a normal float -- are
 * x = log10(f) + log10(fs), y = integrate ? sqrt(running integral) : 10 (log10(p) - log10(fs)), written
 * as pairs of doubles to plot_xy (room for plot_cap points; NULL to skip).  Host, f32 like the reference. */
int psdc_trace_plot(const float *psd, const float *frequencies, size_t n, float fs, int integrate,
                    float integral_start, float integral_end, float *rms, double *plot_xy,
                    size_t plot_cap, size_t *n_points);

/* ---- device utilities ---------------------------------------------------- */

/* HbfDec8 block processing (src/psd.rs:246-253) of a whole host array from zero
 * state on the device: y[m], m < len/8.  Standalone check of the decimator. */
int psdc_hbf_dec8(int device, const float *x, size_t len, float *y);

/* Fill device memory with the bench/test stream: x_i = (u_i - 0.5) * sqrt(12),
 * u_i = (r_i >> 40) * 2^-24 with r_i output first_index + i of SplitMix64 seeded
 * with mix64(seed + GAMMA), i.e. r_i = mix64(key + (first_index + i + 1) * GAMMA)
 * (unit-variance uniform noise, the reference's own test signal src/psd.rs:604-606). */
int psdc_fill_noise_device(int device, float *d_x, size_t len, uint64_t seed,
                           uint64_t first_index);

int psdc_profile_read(psdc_handle *h, psdc_profile *out, int reset);

/* Last error text of a handle; with h == NULL, of the calling thread's last
 * failed psdc_create / handle-less call. */
const char *psdc_last_error(const psdc_handle *h);

int psdc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PSDCASCADE_H */
