// plan.h -- closed-form stream bookkeeping of the reference's segment loop
// (src/psd.rs:196-269) so that a whole batch of segments can be issued to the
// GPU at once.  Pure host code, no HIP: tests/host/plan_check.cpp checks every
// function against a literal simulation of the reference loop.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace psdk {

struct Geometry {
    uint32_t n = 0;       // FFT size (const generic N)
    uint32_t overlap = 0; // win.overlap (src/psd.rs:19,53)
    uint32_t hop = 0;     // n - overlap
    uint32_t drain = 35;  // hbf_dec_response_length(DEPTH) (src/psd.rs:149)
};

// Segments completed once `total` samples have entered a stage:
// the first needs n samples, each further one hop more (src/psd.rs:201-208,266).
inline uint64_t segments_for(const Geometry &g, uint64_t total)
{
    return total < g.n ? 0 : 1 + (total - g.n) / g.hop;
}

// Samples handed to the decimator after J segments: the whole first segment,
// then only the new part of each (src/psd.rs:235-253).
inline uint64_t decimated_prefix(const Geometry &g, uint64_t segs)
{
    return segs == 0 ? 0 : segs * g.hop + g.overlap;
}

// Samples emitted to the next stage: one per 8, minus the one-time drain
// (src/psd.rs:255-260).
inline uint64_t emitted_for(const Geometry &g, uint64_t dec_prefix)
{
    const uint64_t m = dec_prefix >> 3;
    return m > g.drain ? m - g.drain : 0;
}

// PsdStage::buf().len() (src/psd.rs:285-287): samples not yet part of a
// completed segment, including the kept overlap.
inline uint64_t pending_for(const Geometry &g, uint64_t total)
{
    const uint64_t j = segments_for(g, total);
    return j == 0 ? total : total - j * g.hop;
}

// (self.avg.count >> (DEPTH * i)).min(self.avg.limit) (src/psd.rs:434,449).
// A shift of 32 or more is a latent overflow in the reference; treated as 0.
inline uint32_t stage_avg(uint32_t limit, uint32_t count, unsigned i)
{
    const uint32_t v = (3u * i >= 32u) ? 0u : (count >> (3u * i));
    return std::min(v, limit);
}

// `count` after nb more segments (src/psd.rs:218-225): saturates at avg + 1.
// The reference's count is a u32 that `count += 1` overflows after 2^32 segments when avg = u32::MAX
// (panic in debug builds, wrap to 0 in release: psd() then drops the stage or divides by a zero gain) --
// hours on a CPU core, seconds of continuous ingest here.  The library counts in 64 bits: with
// avg = u32::MAX (plain sum) the count is unbounded and gain() follows it; the u32 the ABI reports
// saturates at u32::MAX instead of wrapping.  Below 2^32 segments everything is the reference's value.
inline uint64_t count_after64(uint64_t c0, uint32_t avg, uint64_t nb)
{
    if (nb == 0)
        return c0;
    if (avg == 0xFFFFFFFFu)
        return c0 + nb;
    return std::min<uint64_t>(c0 + nb, (uint64_t)avg + 1);
}
inline uint32_t count_report(uint64_t c) { return c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c; }
inline uint32_t count_after(uint32_t c0, uint32_t avg, uint64_t nb) { return count_report(count_after64(c0, avg, nb)); }

// EWMA over a batch of nb segments (steps i = 1..nb), src/psd.rs:218-233:
//   p <- g_i p + s_i,  g_i = avg/count if count > avg else 1.
// g_1 = avg/c0 if c0 > avg; for i >= 2, g_i = gamma = avg/(avg+1) iff
// i >= i_s, else 1.  Hence  p_end = g_total p_0 + sum_i W_i s_i  with
//   W_i = gamma^max(0, nb - max(i, i_s - 1)),  g_total = g_1 gamma^max(0, nb - i_s + 1).
struct EwmaPlan {
    bool ewma = false;   // any g_i != 1 in this batch
    float g1 = 1.0f;
    float gamma = 1.0f;
    int64_t i_s = 0;     // first step >= 2 with g = gamma
    int64_t nb = 0;
    double g_total = 1.0;
};

inline EwmaPlan plan_ewma(uint32_t c0, uint32_t avg, uint64_t nb)
{
    EwmaPlan p;
    p.nb = (int64_t)nb;
    if (nb == 0)
        return p;
    p.g1 = c0 > avg ? (float)avg / (float)c0 : 1.0f;                 // :218-221
    p.gamma = (float)avg / (float)((uint64_t)avg + 1);               // count == avg + 1
    const int64_t c1 = (int64_t)std::min(c0, avg) + 1;               // count after step 1
    const int64_t i_sat = (int64_t)avg + 3 - c1;                     // first i with c_{i-1} > avg
    p.i_s = std::max<int64_t>(2, i_sat);
    // (gamma rounds to 1.0f for avg >= 2^25: every weight is then exactly 1 -- a plain sum)
    p.ewma = (p.g1 != 1.0f) || (p.nb >= p.i_s && p.gamma != 1.0f);
    const int64_t ng = std::max<int64_t>(0, p.nb - p.i_s + 1);
    double gt = (double)p.g1;
    if (ng > 0)
        gt *= (p.gamma == 0.0f) ? 0.0 : std::pow((double)p.gamma, (double)ng);
    p.g_total = gt;
    return p;
}

// number of gamma factors applied after step i (1-based)
inline int64_t ewma_after(const EwmaPlan &p, int64_t step)
{
    return std::max<int64_t>(0, p.nb - std::max<int64_t>(step, p.i_s - 1));
}

inline double ewma_weight(const EwmaPlan &p, int64_t step)
{
    const int64_t na = ewma_after(p, step);
    if (na == 0)
        return 1.0;
    return p.gamma == 0.0f ? 0.0 : std::pow((double)p.gamma, (double)na);
}

} // namespace psdk
