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

template <size_t N>
class PsdCascade { // src/psd.rs:399-544
public:
    explicit PsdCascade(int device = 0) : h_(psdc_create(N, PSDC_WINDOW_HANN, 1, device))
    {
        if (!h_)
            throw std::runtime_error(psdc_last_error(nullptr));
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

} // namespace stabilizer_stream
