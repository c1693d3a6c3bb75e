// psd_cascade.hpp -- header-only C++ mirror of the reference's PSD surface
// (quartiq/stabilizer-stream src/psd.rs) over the C ABI in include/psdcascade.h.
// Same names and argument meaning as the Rust types; misuse throws where the
// reference panics.  Link with -lpsdcascade.
#pragma once
#include <cstdint>
#include <span>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/psdcascade.h"

namespace stabilizer_stream {

enum class Detrend : int { None = 0, Midpoint = 1, Span = 2, Mean = 3, Linear = 4 }; // src/psd.rs:59-72

struct MergeOpts { // src/psd.rs:339-358
    bool keep_overlap = false;
    uint32_t min_count = 1;
    bool keep_transition_band = false;
};

struct AvgOpts { // src/psd.rs:360-376
    uint32_t limit = UINT32_MAX;
    uint32_t count = UINT32_MAX;
};

struct Break { // src/psd.rs:290-337
    size_t start;
    bool include;
    uint32_t count;
    uint32_t avg;
    std::pair<size_t, size_t> bins; // Range<usize>
    size_t fft_size;
    size_t decimation;
    size_t pending;
    size_t processed;

    size_t effective_fft_size() const { return fft_size * decimation; }
    float rbw() const { return 1.0f / static_cast<float>(effective_fft_size()); }

    static std::vector<float> frequencies(const std::vector<Break> &b)
    {
        std::vector<psdc_break> c(b.size());
        for (size_t i = 0; i < b.size(); ++i)
            c[i] = psdc_break{b[i].start, b[i].include ? 1u : 0u, b[i].count, b[i].avg, 0, b[i].bins.first,
                              b[i].bins.second, b[i].fft_size, b[i].decimation, b[i].pending, b[i].processed};
        std::vector<float> f(psdc_frequencies(c.data(), c.size(), nullptr, 0));
        psdc_frequencies(c.data(), c.size(), f.data(), f.size());
        return f;
    }
};

// Window<N> (src/psd.rs:12-56): a plain struct with public fields, so a caller may build any
template <size_t N>
struct Window {
    std::vector<float> win; // [f32; N]
    float power = 1.0f;     // src/psd.rs:15
    float nenbw = 1.0f;     // src/psd.rs:17
    size_t overlap = 0;     // src/psd.rs:19

    static Window rectangular() { return from_kind(PSDC_WINDOW_RECTANGULAR); } // src/psd.rs:24-32
    static Window hann() { return from_kind(PSDC_WINDOW_HANN); }               // src/psd.rs:42-55

private:
    static Window from_kind(int kind)
    {
        Window w;
        w.win.resize(N);
        if (psdc_window_table(N, kind, w.win.data(), &w.power, &w.nenbw, &w.overlap) < 0)
            throw std::runtime_error(psdc_last_error(nullptr));
        return w;
    }
};

// Psd<N> with the PsdStage trait (src/psd.rs:122-288): one stage, the decimated stream handed back.
template <size_t N>
class Psd {
public:
    // Psd::new(fft, win) (src/psd.rs:137-152): `fft_len` stands for the plan's fft.len() (the transform is the
    // library's own), `win` is any Window<N>; device -1 = $PSDC_DEVICE
    Psd(size_t fft_len, const Window<N> &win, int device = PSDC_DEVICE_DEFAULT)
    {
        static_assert(N >= 2, "Nyquist and DC distinction (src/psd.rs:138)");
        if (fft_len != N || win.win.size() != N)
            throw std::invalid_argument("assertion failed: N == fft.len() (src/psd.rs:139)");
        s_ = psdc_stage_create_window(N, win.win.data(), win.power, win.nenbw, win.overlap, device);
        if (!s_)
            throw std::runtime_error(psdc_last_error(nullptr));
    }
    Psd() : Psd(N, Window<N>::hann()) {}
    Psd(const Psd &o) : s_(psdc_stage_clone(o.s_))
    {
        if (!s_)
            throw std::runtime_error(psdc_last_error(nullptr));
    }
    Psd(Psd &&o) noexcept : s_(std::exchange(o.s_, nullptr)) {}
    Psd &operator=(Psd o) noexcept
    {
        std::swap(s_, o.s_);
        return *this;
    }
    ~Psd() { psdc_stage_destroy(s_); }

    void set_avg(uint32_t avg) { check(psdc_stage_set_avg(s_, avg)); }
    void set_detrend(Detrend d) { check(psdc_stage_set_detrend(s_, static_cast<int>(d))); }
    // PsdStage::process(x, y) -> &mut y[..n] (src/psd.rs:196-269)
    std::span<float> process(std::span<const float> x, std::span<float> y)
    {
        size_t n = 0;
        check(psdc_stage_process(s_, x.data(), x.size(), y.data(), y.size(), &n));
        return y.first(n);
    }
    std::vector<float> spectrum() const
    {
        std::vector<float> p(N / 2 + 1);
        check(psdc_stage_get_spectrum(s_, p.data()));
        return p;
    }
    float gain() const
    {
        float g = 0.0f;
        check(psdc_stage_get_gain(s_, &g));
        return g;
    }
    uint32_t count() const
    {
        uint32_t c = 0;
        check(psdc_stage_get_count(s_, &c));
        return c;
    }
    std::vector<float> buf() const
    {
        size_t len = 0;
        check(psdc_stage_get_buf(s_, nullptr, 0, &len));
        std::vector<float> b(len);
        if (len)
            check(psdc_stage_get_buf(s_, b.data(), b.size(), &len));
        return b;
    }

private:
    void check(int rc) const
    {
        if (rc < 0)
            throw std::runtime_error(std::string("psdcascade: ") + psdc_stage_last_error(s_));
    }
    psdc_stage *s_;
};

// Var (src/var.rs:4-45) with VarBuilder's defaults; eval on a merged PSD
struct Var {
    int x_exp = -2;
    int sinx_exp = 4;
    float clip = 3.4028234663852886e38f;
    size_t dc_cut = 2;
    float eval(std::span<const float> phase_psd, std::span<const float> frequencies, float tau) const
    {
        return psdc_var_eval(x_exp, sinx_exp, clip, dc_cut, phase_psd.data(), frequencies.data(), phase_psd.size(), tau);
    }
};

// Trace::plot (src/bin/psd.rs:125-157): integrated rms and plot points of a merged PSD
inline std::pair<float, std::vector<std::pair<double, double>>> trace_plot(std::span<const float> psd,
                                                                           std::span<const float> frequencies, float fs,
                                                                           bool integrate, float integral_start,
                                                                           float integral_end)
{
    std::vector<std::pair<double, double>> pts(psd.size());
    float rms = 0.0f;
    size_t np = 0;
    static_assert(sizeof(std::pair<double, double>) == 2 * sizeof(double));
    if (psdc_trace_plot(psd.data(), frequencies.data(), psd.size(), fs, integrate, integral_start, integral_end, &rms,
                        reinterpret_cast<double *>(pts.data()), pts.size(), &np) < 0)
        throw std::runtime_error(psdc_last_error(nullptr));
    pts.resize(np);
    return {rms, std::move(pts)};
}

template <size_t N>
class PsdCascade { // src/psd.rs:399-544
public:
    // PsdCascade::<N>::default() (src/psd.rs:408-423); device -1 = $PSDC_DEVICE
    explicit PsdCascade(int device = PSDC_DEVICE_DEFAULT) : h_(psdc_create(N, PSDC_WINDOW_HANN, 1, device))
    {
        if (!h_)
            throw std::runtime_error(psdc_last_error(nullptr));
    }
    // a cascade over a caller-built window (the reference's Default always plans Hann)
    explicit PsdCascade(const Window<N> &win, int device = PSDC_DEVICE_DEFAULT)
        : h_(win.win.size() == N ? psdc_create_window(N, win.win.data(), win.power, win.nenbw, win.overlap, 1, device) : nullptr)
    {
        if (!h_)
            throw std::runtime_error(win.win.size() == N ? psdc_last_error(nullptr) : "window table length != N");
    }
    PsdCascade(const PsdCascade &o) : h_(psdc_clone(o.h_))
    {
        if (!h_)
            throw std::runtime_error(psdc_last_error(nullptr));
    }
    PsdCascade(PsdCascade &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    PsdCascade &operator=(PsdCascade o) noexcept
    {
        std::swap(h_, o.h_);
        return *this;
    }
    ~PsdCascade() { psdc_destroy(h_); }

    float rbw() const { return psdc_rbw(h_); }
    void set_avg(AvgOpts a) { check(psdc_set_avg(h_, a.limit, a.count)); }
    void set_detrend(Detrend d) { check(psdc_set_detrend(h_, static_cast<int>(d))); }
    void process(std::span<const float> x) { check(psdc_process(h_, 0, x.data(), x.size())); }
    void process_device(const float *d_x, size_t len) { check(psdc_process_device(h_, 0, d_x, len)); }
    // how device spans share rounds (include/psdcascade.h): held until `n` spans / 2^29 samples or a call that cannot join -- a function
    // of the calls alone, so the same calls give the same bits; eager(true): also sent out when the device is seen idle (timing-dependent);
    // merge(false): a span that continues the held one in memory becomes a span of its own instead of extending it
    void coalesce(int n) { check(psdc_configure(h_, PSDC_OPT_COALESCE, n)); }
    void eager(bool on) { check(psdc_configure(h_, PSDC_OPT_EAGER, on ? 1 : 0)); }
    void merge(bool on) { check(psdc_configure(h_, PSDC_OPT_MERGE, on ? 1 : 0)); }
    psdc_handle *handle() const { return h_; } // for the batched feeders of source.hpp
    // the multi-GPU read-out record (psdc_pack_readout): gather with any transport, stitch with psd_from_readout
    std::vector<unsigned char> pack_readout() const
    {
        std::vector<unsigned char> rec(psdc_readout_bytes(N, 1));
        size_t len = 0;
        check(psdc_pack_readout(h_, rec.data(), rec.size(), &len));
        return rec;
    }

    std::pair<std::vector<float>, std::vector<Break>> psd(const MergeOpts &o = {}) const
    {
        const int ns = psdc_num_stages(h_, 0);
        check(ns);
        std::vector<float> p(static_cast<size_t>(ns) * (N / 2 + 1));
        std::vector<psdc_break> b(ns);
        size_t plen = 0, nb = 0;
        check(psdc_psd(h_, 0, o.keep_overlap, o.min_count, o.keep_transition_band, p.data(), p.size(), &plen,
                       b.data(), b.size(), &nb));
        p.resize(plen);
        std::vector<Break> out;
        for (size_t i = 0; i < nb; ++i)
            out.push_back(Break{static_cast<size_t>(b[i].start), b[i].include != 0, b[i].count, b[i].avg,
                                {b[i].bins_start, b[i].bins_end}, static_cast<size_t>(b[i].fft_size),
                                static_cast<size_t>(b[i].decimation), static_cast<size_t>(b[i].pending),
                                static_cast<size_t>(b[i].processed)});
        return {std::move(p), std::move(out)};
    }

private:
    void check(int rc) const
    {
        if (rc < 0)
            throw std::runtime_error(std::string("psdcascade: ") + psdc_last_error(h_));
    }
    psdc_handle *h_;
};

// a record as one of `n_channels` channels, the added ones empty (psdc_pack_pad): a gather moves equal blocks, so shards whose
// channel counts differ pad to the largest
inline std::vector<unsigned char> pad_readout(std::span<const unsigned char> rec, uint32_t n, uint32_t n_channels)
{
    std::vector<unsigned char> out(psdc_readout_bytes(n, n_channels));
    if (psdc_pack_pad(rec.data(), rec.size(), out.data(), out.size(), n_channels) < 0)
        throw std::runtime_error(psdc_last_error(nullptr));
    return out;
}

// PsdCascade::psd (src/psd.rs:479-543) of channel `channel` of a gathered read-out record
inline std::pair<std::vector<float>, std::vector<Break>> psd_from_readout(std::span<const unsigned char> rec, uint32_t channel = 0,
                                                                          const MergeOpts &o = {})
{
    uint32_t n = 0, nc = 0, ns = 0;
    if (psdc_unpack_info(rec.data(), rec.size(), channel, &n, &nc, &ns) < 0)
        throw std::runtime_error(psdc_last_error(nullptr));
    std::vector<float> p(static_cast<size_t>(ns) * (n / 2 + 1));
    std::vector<psdc_break> b(ns);
    size_t plen = 0, nb = 0;
    if (psdc_unpack_stitch(rec.data(), rec.size(), channel, o.keep_overlap, o.min_count, o.keep_transition_band, p.data(),
                           p.size(), &plen, b.data(), b.size(), &nb) < 0)
        throw std::runtime_error(psdc_last_error(nullptr));
    p.resize(plen);
    std::vector<Break> out;
    for (size_t i = 0; i < nb; ++i)
        out.push_back(Break{static_cast<size_t>(b[i].start), b[i].include != 0, b[i].count, b[i].avg,
                            {b[i].bins_start, b[i].bins_end}, static_cast<size_t>(b[i].fft_size),
                            static_cast<size_t>(b[i].decimation), static_cast<size_t>(b[i].pending),
                            static_cast<size_t>(b[i].processed)});
    return {std::move(p), std::move(out)};
}

} // namespace stabilizer_stream
