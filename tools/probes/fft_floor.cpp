// fft_floor.cpp -- CPU probe: dynamic-range floor of the workgroup-level FFT (fft_block.h, twiddles from three seeds
// per lane) against a table-twiddle f32 radix-2 FFT (the oracle's f32 arithmetic, i.e. what rustfft-like precomputed
// twiddles give) on a frame that holds a large step under unit noise: rms relative error of |X[k]|^2 over bins far
// below the largest component.  g++ -O2 -std=c++17 -I../../stabilizer-stream_amd/csrc fft_floor.cpp -o fft_floor
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_block.h"
using namespace psdk;

template <int N>
static void run(double step)
{
    using T = BlockFft<N>;
    constexpr int TEAM = T::TEAM;
    std::vector<float> xa(N), xb(N);
    srand(5);
    for (int i = 0; i < N; ++i) {
        const float w = (float)pow(sin(M_PI * i / N), 2.0);
        const float na = ((float)rand() / RAND_MAX - 0.5f) * 3.4641f, nb = ((float)rand() / RAND_MAX - 0.5f) * 3.4641f;
        xa[i] = (na + (i >= N / 3 ? (float)step : -(float)step)) * w; // segment a: a step under noise
        xb[i] = nb * w;                                               // segment b: noise only
    }
    std::vector<cf> z(N), frame(T::FRAME), tw0(T::TW0_SIZE), twa(T::TWA_SIZE), twb(T::TWB_SIZE);
    for (int i = 0; i < N; ++i)
        z[i] = {xa[i], xb[i]};
    for (int c = 0; c < 4; ++c)
        for (int tl = 0; tl < TEAM; ++tl) {
            double a = -2.0 * M_PI * (double)(4 * tl + c) / (double)N;
            tw0[c * TEAM + tl] = {(float)cos(a), (float)sin(a)};
        }
    for (int q = 1; q < T::RA; ++q)
        for (int s = 0; s < T::SA; ++s) {
            double a = -2.0 * M_PI * (double)(s * q) / (double)T::L1;
            twa[(q - 1) * T::SA + s] = {(float)cos(a), (float)sin(a)};
        }
    for (int q = 1; q < T::RB; ++q)
        for (int s = 0; s < 16; ++s) {
            double a = -2.0 * M_PI * (double)(s * q) / (double)T::SA;
            twb[(q - 1) * 16 + s] = {(float)cos(a), (float)sin(a)};
        }
    std::vector<std::vector<cf>> regs(TEAM, std::vector<cf>(16));
    for (int t = 0; t < TEAM; ++t)
        for (int m = 0; m < 4; ++m)
            for (int c = 0; c < 4; ++c)
                regs[t][4 * m + c] = z[4 * t + c + (N / 4) * m];
    for (int t = 0; t < TEAM; ++t) T::pass0(t, regs[t].data(), T::load_seeds(t, tw0.data()));
    for (int t = 0; t < TEAM; ++t) T::store0(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadA(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passA(t, regs[t].data(), T::load_seeds_a(t, twa.data()));
    for (int t = 0; t < TEAM; ++t) T::storeA(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadB(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passB(t, regs[t].data(), twb.data());
    for (int t = 0; t < TEAM; ++t) T::storeB(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadC(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passC(regs[t].data());
    std::vector<double> q(N);
    for (int t = 0; t < TEAM; ++t)
        for (int s = 0; s < 16; ++s)
            q[T::freq_of(t, s)] = (double)regs[t][s].re * regs[t][s].re + (double)regs[t][s].im * regs[t][s].im;
    // two-for-one fold: P[k] = |Xa|^2 + |Xb|^2 = (Q[k] + Q[N-k]) / 2
    // f64 truth and f32 radix-2 table FFT of segment a and b separately (the reference's own arithmetic)
    auto fft = [&](auto &y, auto one) {
        using R = decltype(one);
        int bits = 0;
        while ((1 << bits) < N) ++bits;
        std::vector<std::complex<R>> t(N);
        for (int i = 0; i < N; ++i) {
            int r = 0;
            for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            t[r] = y[i];
        }
        for (int len = 2; len <= N; len <<= 1)
            for (int b = 0; b < N; b += len)
                for (int k = 0; k < len / 2; ++k) {
                    const double a = -2.0 * M_PI * k / len;
                    const std::complex<R> w((R)cos(a), (R)sin(a));
                    const auto u = t[b + k];
                    const std::complex<R> v(t[b + k + len / 2].real() * w.real() - t[b + k + len / 2].imag() * w.imag(),
                                            t[b + k + len / 2].real() * w.imag() + t[b + k + len / 2].imag() * w.real());
                    t[b + k] = u + v;
                    t[b + k + len / 2] = u - v;
                }
        y = t;
    };
    std::vector<std::complex<double>> da(N), db(N);
    std::vector<std::complex<float>> fa(N), fb(N);
    for (int i = 0; i < N; ++i) {
        da[i] = xa[i];
        db[i] = xb[i];
        fa[i] = xa[i];
        fb[i] = xb[i];
    }
    fft(da, 1.0);
    fft(db, 1.0);
    fft(fa, 1.0f);
    fft(fb, 1.0f);
    double pmax = 0;
    for (int k = 0; k <= N / 2; ++k) pmax = std::max(pmax, std::norm(da[k]) + std::norm(db[k]));
    double eg = 0, ef = 0;
    int cnt = 0;
    for (int k = N / 16; k <= N / 2; ++k) {
        const double ref = std::norm(da[k]) + std::norm(db[k]);
        const double g = 0.5 * (q[k] + q[(N - k) % N]);
        const double f = (double)std::norm(fa[k]) + (double)std::norm(fb[k]);
        eg += pow((g - ref) / ref, 2);
        ef += pow((f - ref) / ref, 2);
        ++cnt;
    }
    printf("N=%5d step %8g: rms rel err over bins N/16..N/2: block FFT (seed twiddles, two-for-one) %.3g, f32 radix-2 table FFT %.3g, ratio %.2f\n",
           N, step, sqrt(eg / cnt), sqrt(ef / cnt), sqrt(eg / ef));
}

int main()
{
    for (double s : {0.0, 10.0, 1e3, 5e4}) {
        run<2048>(s);
        run<4096>(s);
        run<16384>(s);
    }
}
