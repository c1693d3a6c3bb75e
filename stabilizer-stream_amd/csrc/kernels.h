// kernels.h -- launch interface between the host runtime (runtime.cpp, planner.cpp, frames_ingest.cpp, readout.cpp) and
// the gfx950 kernels (kernels.hip).  Job descriptors travel by value in the
// kernel argument segment (no descriptor copies, graph-capturable).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

#include "fft_core.h"

namespace psdk {

#ifndef PSDK_MAX_JOBS
// 320: a round of 128 scattered f32 spans of one channel (two jobs a span + its deep stages: ~270) is ONE launch.  (128 until round 5,
// then 160 for sixty-four spans a round; every kernel found its job by a LINEAR scan of this table then -- ~0.1 us a job in front of
// every workgroup -- so longer tables cost what they saved.  With the bisection: 64 -> 128 spans a round reads scattered 2^16 / 2^18 /
// 2^20-sample calls 210 / 433 / 570 -> 235 / 498 / 605 GS/s, 2^22 and the headline unchanged.  There is a cliff further out: 640 jobs --
// about 90 KB of kernel arguments with the aux table -- read 71 / 95 / 104 GS/s, five times slower.)
#define PSDK_MAX_JOBS 320
#endif
constexpr int MAX_JOBS = PSDK_MAX_JOBS; // jobs per launch (they travel in the kernel-argument segment)

#ifdef __HIPCC__
// Launch unit (workgroup or tile) u -> its job: the last job whose first unit is <= u.  The tables are in launch order, so this is
// a bisection -- eight dependent scalar loads for 160 jobs.  (It was a linear scan until round 5: ~0.1 us a job, paid by every
// workgroup before its first load; at 128 jobs a launch that was 12 us.)
template <class Batch, class First>
__device__ __forceinline__ int job_of_unit(const Batch &b, int u, First first)
{
    int lo = 0, hi = b.njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (u >= first(b.jobs[mid]))
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}
#endif

// AdcDac frames resident in device memory as a stage-0 sample source (src/de/data.rs:11-82): the four traces of a run
// of whole frames.  Sample i of trace ch: cell = i >> 3 (one 16-byte (batch, channel) cell = 8 i16 samples, src/de/data.rs:13),
// frame = cell / batches, batch = cell % batches, at frames + frame * frame_size + 8 + batch * 64 + ch * 16 + (i & 7) * 2;
// ADC words are i16, DAC words offset binary (i16.wrapping_add(i16::MIN), :64,:75), both times 4.096 * 2.5 / 32768 (:28-35).
// Jobs name a span by its index in their batch's table (fspan >= 0) and say where in the span they start (s_off, samples).
struct FrameSpan {
    const uint8_t *frames;
    unsigned long long bytes; // n_frames * frame_size (the bound of the buffer descriptor the fused kernels read through)
    unsigned frame_size;
    unsigned batches;         // 1 ... 255
    unsigned magic;           // ceil(2^32 / batches): cell / batches == umulhi(cell, magic) for cell < 2^24 (batches >= 2)
    unsigned pad;
};
constexpr int MAX_FSPANS = 16;                      // frame spans per launch (= spans of one channel in one round)
constexpr unsigned long long FSPAN_MAX_SAMPLES = 1ull << 26; // per trace and span: cells stay below 2^23

// One span of consecutive segments of one (channel, stage) stream.
struct SegJob {
    const float *src;    // sample with absolute stream index i is src[i - src_base]
    long long src_base;
    long long seg0;      // absolute index of the first segment (segment j starts at j*hop)
    float *partial;      // [nblocks][n] power partials, natural bin order
    double log2_gamma;   // EWMA: log2(avg/(avg+1)); -inf when avg == 0
    int nseg;            // segments in this span
    int block_begin;     // first workgroup of this job within the launch
    int nblocks;         // workgroups of this job; workgroup b walks tiles b, b+nblocks, ...
    int ntiles;          // tiles (of welch_segments_per_tile segments) in this span
    int step0;           // EWMA: 1-based batch step of seg0
    int nb;              // EWMA: steps in the whole (channel, stage) batch
    int is_m1;           // EWMA: i_s - 1
    int ewma;            // 0: plain sum (all weights 1)
    int fspan = -1;      // >= 0: src is trace fch of frame span fspans[fspan]; sample i of the stream is sample
    int fch = 0;         //       (i - src_base) + s_off of that trace (src unused)
    unsigned s_off = 0;
};

struct WelchBatch {
    int njobs;
    int nblocks; // grid size
    int hop;
    int detrend;
    FrameSpan fspans[MAX_FSPANS];
    SegJob jobs[MAX_JOBS];
};

// One span of /8 decimator outputs of one (channel, stage) stream.
struct DecJob {
    const float *src;
    long long src_base;
    long long m0;        // first decimator output index (output m consumes inputs 8m..8m+7)
    float *dst;          // next stage's stream: output m lands at dst[(m - drain) - dst_base]
    long long dst_base;
    int nout;
    int tile_begin;
    int fspan = -1, fch = 0; // as in SegJob
    unsigned s_off = 0;
};

struct DecBatch {
    int njobs;
    int ntiles;
    int drain;
    FrameSpan fspans[MAX_FSPANS];
    DecJob jobs[MAX_JOBS];
};

// Fold the partials of one (channel, stage) into its spectrum accumulator:
// spectrum[k] = g_total*spectrum[k] + 0.5*sum_b (P[b][k] + P[b][(n-k)%n]).
struct RedJob {
    const float *partial;
    float *spectrum;
    float g_total;
    int nparts;
};

struct RedBatch {
    int njobs;
    int n;
    RedJob jobs[MAX_JOBS];
};

// Fused fast path (N = 256, 512, 1024, Hann; any implemented detrend; sum or EWMA): a run
// of whole segment PAIRS.  Pair i = segments (seg_a + 2i, seg_a + 2i + 1) = samples
// src[N i .. N i + 3N/2); its N new samples src[N i + N/2 ..) are decimated to N/8 outputs
// dst[(N/8) i ..).  src must be 16-byte aligned.  The decimator state at the start of a run
// is rebuilt from the 288 samples before its first new sample: `pre` says how many samples in
// front of src are real memory; anything earlier is the zero history of a fresh stream.
struct FusedJob {
    const float *src;   // first sample of segment seg_a
    float *dst;         // where decimator output 64 (seg_a + 1) lands in the next stage's stream
    float *partial;     // [nblocks][N]
    double log2_gamma;  // EWMA, as in SegJob
    int npairs;
    int run;            // consecutive pairs per team (workgroup b owns pairs [b, b+1) * teams * run)
    int pre;            // readable samples in front of src
    int block_begin;
    int nblocks;
    int step0;          // EWMA: 1-based batch step of segment seg_a
    int nb;
    int is_m1;
    int ewma;
    int fspan = -1, fch = 0; // as in SegJob: src = sample s_off of trace fch of fspans[fspan] (s_off a multiple of 4)
    unsigned s_off = 0;
    int pre_first = 0, pre_count = 0; // copy prologue of this (single-workgroup) job: FusedAux::tail[pre_first .. pre_first + pre_count)
};

struct FusedBatch {
    int njobs;
    int nblocks;
    int detrend;  // Detrend kind 0..3 for every job of the launch
    int any_ewma; // some job has finite averaging weights
    int any_frames; // some job reads AdcDac frames (the kernels built with the frame loads run this launch)
    // overlap 0 (Window::rectangular(), src/psd.rs:24-32, or a caller's table with overlap 0): a "pair" is ONE segment -- pair i =
    // segment seg_a + i = samples src[N i + N/2 .. N i + 3N/2), transformed with a zero imaginary part: exactly the N samples the
    // pair decimates, as ever; src points HALF A SEGMENT IN FRONT of segment seg_a (that half chunk is read for the decimator's
    // history only), so the stream is consumed as with half-overlapped pairs (the SINGLE kernels).
    // 2 (N <= 1024): as 1, but segments 2i and 2i + 1 share ONE transform (the two-for-one of two disjoint segments); every job
    // then holds an even number of pairs and `run` is even
    int single;
    // frame jobs come first in the launch, four by four (the traces of one span, equal workgroup counts): group g
    // = workgroups [fg_begin[g], fg_begin[g] + 4 fg_nb[g]); the kernel deals a group's workgroups so that the four
    // that read the same frames sit on one XCD (they share its L2: the frame bytes cross the fabric once, not four times)
    int n_fgroups;
    int fg_begin[MAX_FSPANS], fg_nb[MAX_FSPANS];
    FrameSpan fspans[MAX_FSPANS];
    FusedJob jobs[MAX_JOBS];
};

#ifndef PSDK_FUSED_WAVES
#define PSDK_FUSED_WAVES 8
#endif
#ifndef PSDK_FUSED_WPS
#define PSDK_FUSED_WPS 4
#endif
#ifndef PSDK_BIG_WPS
#define PSDK_BIG_WPS 4
#endif
constexpr int BIG_WAVES_PER_SIMD = PSDK_BIG_WPS;     // N = 2048 ... 8192 kernels: wavefronts per SIMD they are built for
constexpr int FUSED_WAVES = PSDK_FUSED_WAVES;        // wavefronts per workgroup
constexpr int FUSED_WAVES_PER_SIMD = PSDK_FUSED_WPS; // launch bound: wavefronts per SIMD

// Carry the unconsumed tail of a stream to the front of its other buffer.
struct TailJob {
    const float *src;
    float *dst;
    int count;
    int fspan = -1, fch = 0; // >= 0: decode samples s_off ... of trace fch of fspans[fspan] instead of copying src
    unsigned s_off = 0;
};

struct TailBatch {
    int njobs;
    FrameSpan fspans[MAX_FSPANS];
    TailJob jobs[MAX_JOBS];
};

// What rides in a fused launch besides its jobs when a round is ONE launch (PSDK_FOLD, fused.hip): workgroups [0, nblocks) of the grid
// fold the partials of the round BEFORE (the other partial slab) and carry this round's stream tails -- neither depends on anything
// this launch writes --, and jobs [ntail, ntail + npre) of `tail` are copy PROLOGUES of single-workgroup fused jobs (the seams: the
// head of a new span behind the carried tail), named by FusedJob::pre_first / pre_count.
#ifndef PSDK_AUX_MAX_TAIL
#define PSDK_AUX_MAX_TAIL 320
#endif
constexpr int AUX_MAX_RED = 96, AUX_MAX_TAIL = PSDK_AUX_MAX_TAIL, AUX_RED_BINS = 16;
constexpr int AUX_SHORT_ROWS = 4, AUX_MID_ROWS = 64, AUX_MID_GROUPS = 8; // the three shapes of a fold job, by its partial rows
struct FusedAux {
    int nblocks;    // aux workgroups in front of the compute workgroups: red_blocks + ntail
    int red_blocks; // nred_tall * red_xb + nred_mid * red_mb + (nred - nred_tall - nred_mid)
    int red_xb;     // workgroups per tall job (AUX_RED_BINS bins each)
    int red_mb;     // workgroups per mid job (AUX_MID_GROUPS groups of AUX_RED_BINS bins each)
    int nred, ntail, npre;
    int n;          // FFT size
    int nred_tall;  // red[0, nred_tall): more than AUX_MID_ROWS partial rows
    int nred_mid;   // red[nred_tall, + nred_mid): AUX_SHORT_ROWS < rows <= AUX_MID_ROWS; the rest take ONE workgroup each
    RedJob red[AUX_MAX_RED];
    TailJob tail[AUX_MAX_TAIL];
};

// tile geometry (host needs it to size partial slabs and grids)
int welch_segments_per_tile(int n);
constexpr int WELCH_MAX_BLOCKS = 1024; // persistent workgroups per launch (4 per CU)
constexpr int DEC_TILE = 256; // decimator outputs per workgroup

bool welch_supported(int n);
// sizes that are not powers of two (16 < n <= 8192): the transform length M (a power of two >= 2n - 1) of the chirp-z form
// the generic kernel evaluates the n-point DFT in; 0 for powers of two.  tw is then W_M^j (M entries), chirp[j] =
// exp(i pi j^2 / n) (n entries), bhat = FFT_M of the chirp wrapped around M (M entries).
int bluestein_size(int n);
hipError_t launch_welch(int n, const WelchBatch &b, const float *win, const cf *tw, const cf *chirp, const cf *bhat, hipStream_t s);
// powers of two above what an LDS frame holds (bigfft.hip: 32768 ... BIGFFT_MAX_N): the same jobs through a four-step transform
// n = n/256 x 256 whose one intermediate frame lives in `scratch` -- every job of the batch shares the launches of a chunk of as many
// segment pairs as fit (BIGFFT_SCRATCH_ELEMS: 976 pairs at n = 32768, 480 at 65536, 224 at 131072; chunk_limit > 0 caps it: a test hook) --,
// tw[j] = W_n^j; ONE partial row per job (welch_segments_per_tile is "all of them" at these sizes)
constexpr int BIGFFT_MAX_N = 131072;
constexpr size_t BIGFFT_SCRATCH_ELEMS = (size_t)2 << 24; // 256 MiB of complex f32
bool bigfft_size(int n);
hipError_t launch_welch_big(int n, const WelchBatch &b, const float *win, const cf *tw, cf *scratch, size_t scratch_elems, int chunk_limit,
                            hipStream_t s);
bool fused_supported(int n);                 // N = 256 ... 16384
bool fused_frames_supported(int n);          // sizes whose fused kernel can read AdcDac frames in place
bool fused_double_supported(int n);          // overlap 0: sizes whose kernel transforms two disjoint segments at once
bool fused_fold_supported(int n);            // sizes whose fused kernel carries aux workgroups and job prologues (FusedAux): one launch per round
int fused_pairs_per_block(int n, int run);   // teams per workgroup x run
int fused_max_blocks(int n);                 // resident workgroups a launch is sized for
int fused_block_threads(int n);              // threads of one such workgroup
// twiddle tables in global memory for the workgroup-level kernels (N >= 2048); empty otherwise
void fused_big_tables(int n, std::vector<cf> &tw0, std::vector<cf> &twa);
// twiddle seeds of the three-pass kernels (N = 2048, 4096: fft_block3.h) [2][N/16]: W_N^tl, W_N^(4 tl); empty otherwise
void fused_big3_table(int n, std::vector<cf> &tw3);
// ev_a / ev_b (both or neither): events that receive the kernel's own start and stop times
// (hipExtLaunchKernelGGL), for PSDC_OPT_PROFILE
hipError_t launch_fused(int n, const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag, const cf *tw3g,
                        hipStream_t s, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr, const FusedAux *aux = nullptr);
hipError_t launch_dec(const DecBatch &b, hipStream_t s);
hipError_t launch_post(const RedBatch &red, const TailBatch &tail, hipStream_t s);
hipError_t launch_fill_noise(float *d_x, size_t len, uint64_t seed, uint64_t first, hipStream_t s);
// device -> pinned host copy by a kernel (read-outs; see kernels.hip)
hipError_t launch_copy_out(float *h_dst, const float *d_src, size_t count, hipStream_t s);
// frames: device copy of n_frames frames of frame_size bytes (AdcDac, `batches`
// batches each); dst[c] receives 8*batches*n_frames samples of trace c.
hipError_t launch_adcdac(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches,
                         float *dst0, float *dst1, float *dst2, float *dst3, hipStream_t s);

// the other payload formats (src/de/data.rs:84-212; fmt 2 Fls, 3 ThermostatEem, 4 Mpll: 56 / 80 / 24 bytes a batch, one sample per
// batch and trace, four / four / three traces -- dst3 unused for Mpll): dst[c] receives batches * n_frames samples of trace c
hipError_t launch_payload(int fmt, const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, float *dst0, float *dst1,
                          float *dst2, float *dst3, hipStream_t s);

// the 8 header bytes of each of n_frames device-resident frames into out_pinned (pinned host memory, 8 n_frames bytes, little-endian
// as on the wire)
hipError_t launch_header_gather(const uint8_t *frames, size_t frame_size, size_t n_frames, void *out_pinned, hipStream_t s);

// device-resident frames: header checks of every frame (check != 0) and the Loss counters over the first n_loss, in ONE
// launch; the four result words land in host_out (pinned host memory) when the stream has run the kernel: host_out[0] =
// max ~(index << 2 | code) over the bad frames (0: none), [1] batches received, [2] sequence gaps, [3] first seq | next seq
// << 32.  acc: 5 device words, zero before the first call (the kernel leaves them zero).
constexpr int FRAME_RESERVE_BLOCKS = 8; // workgroup slots a frame round leaves free for it (it runs beside the fused launch)
// $PSDC_DBG_VARIANT (test aid, read once): bit 0 runs every fused launch on the EWMA kernel variants (weights of 1 for plain
// sums), bit 1 on the FRAMES variants (no framed job: the f32 path of those kernels) -- the whole suite then exercises the
// variants that only finite averaging / AdcDac frames reach otherwise.
inline int dbg_variant()
{
    static const int v = [] {
        const char *e = getenv("PSDC_DBG_VARIANT");
        return e ? atoi(e) : 0;
    }();
    return v;
}

hipError_t launch_adcdac_verdict(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, int payload_ok, int check,
                                 size_t n_loss, unsigned long long *acc, unsigned long long *host_out, hipStream_t s);

} // namespace psdk
