// cpp_mirror_check.cpp -- the reference's stream_test flow (src/bin/stream_test.rs:38-66) written
// against the C++ mirror stabilizer-stream_amd/cpp/psd_cascade.hpp: four PsdCascade<512>, traces fed
// in small chunks, clone, psd(&MergeOpts::default()), Break::frequencies.  The numbers are those of
// the reference's own test (src/psd.rs:602-643): 2^16 samples of unit-variance white noise, default
// detrend, every included bin within 10/sqrt(count) of PSD = 2 (trace 0 runs Detrend::Midpoint as
// stream_test does, which is exercised but not held to that bound: on white noise it adds a random
// DC level by construction).
// Needs a GPU to run; building it is the check that the mirror matches the C ABI.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../stabilizer-stream_amd/cpp/psd_cascade.hpp"

using namespace stabilizer_stream;

int main()
{
    constexpr size_t N = 1 << 9;
    std::vector<PsdCascade<N>> dec;
    for (int i = 0; i < 4; ++i)
        dec.emplace_back();
    dec[0].set_detrend(Detrend::Midpoint); // src/bin/stream_test.rs:41
    std::mt19937 rng(0x7654321);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<float> x(176); // one dual-iir frame's worth per trace (22 batches x 8)
    const size_t total = size_t(1) << 16; // src/psd.rs:603
    for (size_t done = 0; done < total; done += x.size())
        for (auto &d : dec) {
            for (auto &v : x)
                v = (u(rng) - 0.5f) * std::sqrt(12.0f); // src/psd.rs:604-606
            d.process(x);
        }
    PsdCascade<N> copy = dec[1]; // Clone (src/psd.rs:399)
    const auto [y, b] = copy.psd(MergeOpts{});
    const auto f = Break::frequencies(b);
    if (b.empty() || y.size() != f.size() || f.front() != 0.0f || f.back() != 0.5f) {
        std::fprintf(stderr, "bad shape: %zu breaks, %zu bins, %zu freqs\n", b.size(), y.size(), f.size());
        return 1;
    }
    int bad = 0;
    for (const auto &br : b) {
        if (!br.include)
            continue;
        const double tol = 10.0 / std::sqrt((double)br.count); // the reference's own bound (src/psd.rs:639-642)
        for (size_t k = br.start; k < br.start + (br.bins.second - br.bins.first); ++k)
            if (std::fabs(y[k] * 0.5 - 1.0) > tol)
                ++bad;
        std::printf("stage fft %zu x%zu: count %u, bins %zu..%zu, rbw %.3g\n", br.fft_size, br.decimation, br.count,
                    br.bins.first, br.bins.second, br.rbw());
    }
    bool threw = false;
    try {
        copy.set_detrend(Detrend::Linear); // unimplemented!() in the reference (src/psd.rs:110)
    } catch (const std::runtime_error &) {
        threw = true;
    }
    std::printf("%zu stages, %zu bins, %d out of bound, Linear %s\n", b.size(), y.size(), bad, threw ? "throws" : "ACCEPTED");
    return (bad == 0 && threw) ? 0 : 1;
}
