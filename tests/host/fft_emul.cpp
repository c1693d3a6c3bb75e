// Host emulation of stabilizer-stream_amd/csrc/fft_core.h: runs the device
// FFT code lane by lane on the CPU, checks |X[k]|^2 against a direct f64 DFT,
// and counts LDS bank conflicts with the gfx950 banking rules
// (ds_read_b64: two 32-lane groups, 64 banks x 4 B; ds_write_b64: four
// 16-lane groups, 32 banks x 4 B).  Build: g++ -O2 -std=c++17 -I<csrc>.
#include "fft_core.h"
#include "fft_wave1024.h"
#include <cmath>
#include <algorithm>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace psdk;

struct Conf {
    long rd = 0, wr = 0, rd_ideal = 0, wr_ideal = 0;
};

static int group_cycles(const std::vector<int> &slots, int nslots)
{
    // slots: 8-byte slot index per lane of one group; distinct addresses on the
    // same bank pair serialize, identical addresses broadcast
    std::vector<std::vector<int>> per(nslots);
    int worst = 1;
    for (int a : slots) {
        auto &v = per[a % nslots];
        bool dup = false;
        for (int b : v)
            if (b == a) dup = true;
        if (!dup) v.push_back(a);
        if ((int)v.size() > worst) worst = (int)v.size();
    }
    return worst;
}

template <int N, int P>
static void run_pass(std::vector<std::vector<cf>> &regs, std::vector<cf> &frame,
                     const std::vector<cf> &tw, Conf &conf, bool rotate)
{
    using Plan = FftPlan<N>;
    using PI = PassInfo<N, P>;
    constexpr int TEAM = Plan::TEAM;
    if constexpr (P > 0) {
        for (int t = 0; t < TEAM; ++t) {
            int rot = (rotate && PI::LAST) ? ((t >> 3) & (PI::R - 1)) : 0;
            pass_load<N, P>(t, regs[t].data(), frame.data(), rot);
        }
        // conflicts of the read instructions: one instruction per (i, m)
        for (int i = 0; i < PI::NB; ++i)
            for (int m = 0; m < PI::R; ++m)
                for (int w0 = 0; w0 < TEAM; w0 += 64)
                    for (int g = 0; g < 2; ++g) {
                        std::vector<int> s;
                        for (int l = 0; l < 32 && w0 + g * 32 + l < TEAM; ++l) {
                            int t = w0 + g * 32 + l;
                            int rot = (rotate && PI::LAST) ? ((t >> 3) & (PI::R - 1)) : 0;
                            int mm = PI::LAST ? (m + rot) % PI::R : m;
                            s.push_back(lds_swz<N>(PI::elem(t, i, mm)));
                        }
                        if (s.empty()) continue;
                        conf.rd += group_cycles(s, 32);
                        conf.rd_ideal += 1;
                    }
    }
    for (int t = 0; t < TEAM; ++t)
        pass_compute<N, P>(t, regs[t].data(), tw.data());
    if constexpr (!PI::LAST) {
        for (int t = 0; t < TEAM; ++t)
            pass_store<N, P>(t, regs[t].data(), frame.data());
        for (int i = 0; i < PI::NB; ++i)
            for (int q = 0; q < PI::R; ++q)
                for (int w0 = 0; w0 < TEAM; w0 += 16) {
                    std::vector<int> s;
                    for (int l = 0; l < 16 && w0 + l < TEAM; ++l)
                        s.push_back(lds_swz<N>(PI::elem(w0 + l, i, q)));
                    conf.wr += group_cycles(s, 16);
                    conf.wr_ideal += 1;
                }
        run_pass<N, P + 1>(regs, frame, tw, conf, rotate);
    }
}

template <int N>
static int check(bool rotate)
{
    using Plan = FftPlan<N>;
    constexpr int E = Plan::E, TEAM = Plan::TEAM;
    std::vector<cf> z(N), tw(N), frame(N);
    srand(1234 + N);
    for (int i = 0; i < N; ++i) {
        z[i].re = (float)rand() / RAND_MAX - 0.5f;
        z[i].im = (float)rand() / RAND_MAX - 0.5f;
        double a = -2.0 * M_PI * i / N;
        tw[i] = {(float)cos(a), (float)sin(a)};
    }
    std::vector<std::vector<cf>> regs(TEAM, std::vector<cf>(E));
    using P0 = PassInfo<N, 0>;
    for (int t = 0; t < TEAM; ++t)
        for (int i = 0; i < P0::NB; ++i)
            for (int m = 0; m < P0::R; ++m)
                regs[t][i * P0::R + m] = z[P0::elem(t, i, m)];
    Conf conf;
    run_pass<N, 0>(regs, frame, tw, conf, rotate);
    // reference: direct DFT in double (O(N^2), fine up to 16384 with a table)
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, -2.0 * M_PI * i / N);
    std::vector<double> pw(N), got(N, -1.0);
    double pmax = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        for (int j = 0; j < N; ++j)
            acc += std::complex<double>(z[j].re, z[j].im) * w[(int)(((long)j * k) % N)];
        pw[k] = std::norm(acc);
        if (pw[k] > pmax) pmax = pw[k];
    }
    int seen = 0;
    for (int t = 0; t < TEAM; ++t)
        for (int s = 0; s < E; ++s) {
            int k = freq_of_slot<N>(t, s);
            if (k < 0 || k >= N || got[k] >= 0) {
                printf("N=%d: bad/duplicate k=%d (t=%d slot=%d)\n", N, k, t, s);
                return 1;
            }
            got[k] = (double)regs[t][s].re * regs[t][s].re + (double)regs[t][s].im * regs[t][s].im;
            ++seen;
        }
    double err = 0;
    for (int k = 0; k < N; ++k) {
        double e = fabs(got[k] - pw[k]) / pmax;
        if (e > err) err = e;
    }
    printf("N=%5d rot=%d passes=%d max|dP|/Pmax=%.3g  lds read cycles %ld (ideal %ld)  write cycles %ld (ideal %ld)\n",
           N, (int)rotate, Plan::NPASS, err, conf.rd, conf.rd_ideal, conf.wr, conf.wr_ideal);
    return (seen == N && err < 2e-6) ? 0 : 1;
}

// the wave-level (4,16,16) FFT of the fused kernel
static int check_wave1024()
{
    using namespace w1024;
    std::vector<cf> z(N), frame(FRAME), tw0(TW0_SIZE), tw1(TW1_SIZE);
    srand(99);
    for (int i = 0; i < N; ++i) {
        z[i].re = (float)rand() / RAND_MAX - 0.5f;
        z[i].im = (float)rand() / RAND_MAX - 0.5f;
    }
    for (int q = 1; q < 4; ++q)
        for (int s = 0; s < 256; ++s) {
            double a = -2.0 * M_PI * (double)(s * q) / 1024.0;
            tw0[(q - 1) * 256 + s] = {(float)cos(a), (float)sin(a)};
        }
    for (int q = 1; q < 16; ++q)
        for (int s = 0; s < 16; ++s) {
            double a = -2.0 * M_PI * (double)(s * q) / 256.0;
            tw1[(q - 1) * 16 + s] = {(float)cos(a), (float)sin(a)};
        }
    std::vector<std::vector<cf>> regs(64, std::vector<cf>(16));
    for (int t = 0; t < 64; ++t)
        for (int m = 0; m < 4; ++m)
            for (int c = 0; c < 4; ++c)
                regs[t][4 * m + c] = z[4 * t + c + 256 * m];
    for (int t = 0; t < 64; ++t) pass0(t, regs[t].data(), tw0.data());
    for (int t = 0; t < 64; ++t) store0(t, regs[t].data(), frame.data());
    for (int t = 0; t < 64; ++t) load1(t, regs[t].data(), frame.data());
    for (int t = 0; t < 64; ++t) pass1(t, regs[t].data(), tw1.data());
    for (int t = 0; t < 64; ++t) store1(t, regs[t].data(), frame.data());
    for (int t = 0; t < 64; ++t) load2(t, regs[t].data(), frame.data());
    for (int t = 0; t < 64; ++t) pass2(regs[t].data());
    // conflicts
    long rd = 0, rdi = 0, wr = 0, wri = 0;
    for (int q = 0; q < 4; ++q)
        for (int c = 0; c < 4; ++c)
            for (int g = 0; g < 4; ++g) {
                std::vector<int> sl;
                for (int t = 16 * g; t < 16 * g + 16; ++t) sl.push_back(swz(256 * q + 4 * t + c));
                wr += group_cycles(sl, 16), ++wri;
            }
    for (int m = 0; m < 16; ++m) {
        for (int g = 0; g < 2; ++g) {
            std::vector<int> a, b;
            for (int t = 32 * g; t < 32 * g + 32; ++t) {
                a.push_back(swz(256 * (t >> 4) + (t & 15) + 16 * m));
                b.push_back(swz(16 * t + m));
            }
            rd += group_cycles(a, 32) + group_cycles(b, 32), rdi += 2;
        }
        for (int g = 0; g < 4; ++g) {
            std::vector<int> a;
            for (int t = 16 * g; t < 16 * g + 16; ++t) a.push_back(swz(256 * (t >> 4) + (t & 15) + 16 * m));
            wr += group_cycles(a, 16), ++wri;
        }
    }
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, -2.0 * M_PI * i / N);
    std::vector<double> got(N, -1.0);
    double err = 0, pmax = 0;
    std::vector<double> pw(N);
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        for (int j = 0; j < N; ++j)
            acc += std::complex<double>(z[j].re, z[j].im) * w[(int)(((long)j * k) % N)];
        pw[k] = std::norm(acc);
        pmax = std::max(pmax, pw[k]);
    }
    for (int t = 0; t < 64; ++t)
        for (int q = 0; q < 16; ++q) {
            int k = freq_of(t, q);
            if (k < 0 || k >= N || got[k] >= 0) {
                printf("wave1024: bad/duplicate k=%d\n", k);
                return 1;
            }
            got[k] = (double)regs[t][q].re * regs[t][q].re + (double)regs[t][q].im * regs[t][q].im;
        }
    for (int k = 0; k < N; ++k) err = std::max(err, fabs(got[k] - pw[k]) / pmax);
    printf("wave1024 (4,16,16): max|dP|/Pmax=%.3g  lds read cycles %ld (ideal %ld)  write cycles %ld (ideal %ld)\n",
           err, rd, rdi, wr, wri);
    return (err < 2e-6 && rd == rdi && wr == wri) ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad |= check<16>(false);
    bad |= check<32>(false);
    bad |= check<64>(false);
    bad |= check<128>(false);
    bad |= check<256>(false);
    bad |= check<512>(false);
    bad |= check<1024>(false);
    bad |= check<1024>(true);
    bad |= check<2048>(false);
    bad |= check<4096>(false);
    bad |= check<8192>(false);
    bad |= check<16384>(false);
    bad |= check_wave1024();
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
