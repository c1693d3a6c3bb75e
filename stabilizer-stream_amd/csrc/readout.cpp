// readout.cpp -- everything that reads a cascade out: PsdStage accessors (src/psd.rs:271-287), PsdCascade::psd and Break
// (:290-337, :479-543), clone, the packed read-out record, Var::eval / Trace::plot on the host, and the single-stage Psd<N>.
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

namespace psdrt {

int stitch_impl(uint32_t n, float nenbw, float power, uint32_t overlap, uint32_t n_stages,
                const uint32_t *counts, const uint32_t *avgs, const uint64_t *pendings,
                const float *spectra, int keep_overlap, uint32_t min_count, int keep_transition_band,
                float *psd_out, size_t psd_cap, size_t *psd_len, psdc_break *breaks,
                size_t breaks_cap, size_t *n_breaks, const uint64_t *counts64)
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

} // namespace psdrt

extern "C" {

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
    o->eager = h->eager;
    o->merge = h->merge;
    o->fold = h->fold;
    o->span_cap = h->span_cap;
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
    if (!out || n_channels < hd->n_channels || need == 0 || cap < need)
        return fail(nullptr, PSDC_ERR_ARG, "psdc_pack_pad: fewer channels than the record holds, or buffer too small (psdc_readout_bytes)");
    {   // [out, out + need) must not touch [rec, rec + len): the copy below is a memcpy and the header rewrite would corrupt the source
        const uintptr_t o0 = reinterpret_cast<uintptr_t>(out), r0 = reinterpret_cast<uintptr_t>(rec);
        if (o0 < r0 + len && r0 < o0 + need)
            return fail(nullptr, PSDC_ERR_ARG, "psdc_pack_pad: out overlaps rec");
    }
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
    try { // (nothing unwinds across the ABI; ns <= 16 and bins <= PACK_MAX_N / 2 + 1 = 65537 here, so this is ~4 MiB at most)
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
