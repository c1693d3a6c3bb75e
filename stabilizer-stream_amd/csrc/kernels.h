// kernels.h -- launch interface between the host runtime (psdcascade.cpp) and
// the gfx950 kernels (kernels.hip).  Job descriptors travel by value in the
// kernel argument segment (no descriptor copies, graph-capturable).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

#include "fft_core.h"

namespace psdk {

#ifndef PSDK_MAX_JOBS
#define PSDK_MAX_JOBS 128
#endif
constexpr int MAX_JOBS = PSDK_MAX_JOBS; // jobs per launch (they travel in the kernel-argument segment)

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
};

struct WelchBatch {
    int njobs;
    int nblocks; // grid size
    int hop;
    int detrend;
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
};

struct DecBatch {
    int njobs;
    int ntiles;
    int drain;
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
};

struct FusedBatch {
    int njobs;
    int nblocks;
    int detrend;  // Detrend kind 0..3 for every job of the launch
    int any_ewma; // some job has finite averaging weights
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
};

struct TailBatch {
    int njobs;
    TailJob jobs[MAX_JOBS];
};

// tile geometry (host needs it to size partial slabs and grids)
int welch_segments_per_tile(int n);
constexpr int WELCH_MAX_BLOCKS = 1024; // persistent workgroups per launch (4 per CU)
constexpr int DEC_TILE = 256; // decimator outputs per workgroup

bool welch_supported(int n);
hipError_t launch_welch(int n, const WelchBatch &b, const float *win, const cf *tw, hipStream_t s);
bool fused_supported(int n);                 // N = 256 ... 16384
int fused_pairs_per_block(int n, int run);   // teams per workgroup x run
int fused_max_blocks(int n);                 // resident workgroups a launch is sized for
// twiddle tables in global memory for the workgroup-level kernels (N >= 2048); empty otherwise
void fused_big_tables(int n, std::vector<cf> &tw0, std::vector<cf> &twa);
// ev_a / ev_b (both or neither): events that receive the kernel's own start and stop times
// (hipExtLaunchKernelGGL), for PSDC_OPT_PROFILE
hipError_t launch_fused(int n, const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag,
                        hipStream_t s, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr);
hipError_t launch_dec(const DecBatch &b, hipStream_t s);
hipError_t launch_post(const RedBatch &red, const TailBatch &tail, hipStream_t s);
hipError_t launch_fill_noise(float *d_x, size_t len, uint64_t seed, uint64_t first, hipStream_t s);
// device -> pinned host copy by a kernel (read-outs; see kernels.hip)
hipError_t launch_copy_out(float *h_dst, const float *d_src, size_t count, hipStream_t s);
// frames: device copy of n_frames frames of frame_size bytes (AdcDac, `batches`
// batches each); dst[c] receives 8*batches*n_frames samples of trace c.
hipError_t launch_adcdac(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches,
                         float *dst0, float *dst1, float *dst2, float *dst3, hipStream_t s);

// device-resident frames (out: device memory, zeroed): header checks of every frame (out[0] <- max ~(index << 2 | code)
// over the bad ones) and the
// Loss counters over the first n (out[1] += batches, out[2] += sequence gaps, out[3] = first seq | next seq << 32)
hipError_t launch_adcdac_scan(const uint8_t *frames, size_t frame_size, size_t n_frames, int batches, int payload_ok,
                              unsigned long long *out, hipStream_t s);
hipError_t launch_adcdac_loss(const uint8_t *frames, size_t frame_size, size_t n, unsigned long long *out, hipStream_t s);

} // namespace psdk
