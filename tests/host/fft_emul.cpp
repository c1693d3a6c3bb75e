// Host emulation of stabilizer-stream_amd/csrc/fft_core.h: runs the device
// FFT code lane by lane on the CPU, checks |X[k]|^2 against a direct f64 DFT,
// and counts LDS bank conflicts with the gfx950 banking rules
// (ds_read_b64: two 32-lane groups, 64 banks x 4 B; ds_write_b64: four
// 16-lane groups, 32 banks x 4 B).  Build: g++ -O2 -std=c++17 -I<csrc>.
#include "fft_core.h"
#include "fft_team.h"
#include "fft_block.h"
#include "fft_block3.h"
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
        // conflicts of the read instructions, one instruction per (i, m), over a WAVEFRONT: lane l belongs to team l / TEAM of the
        // wave (frames LdsFrame<N>::SIZE apart) when a team is smaller than the wave, to wave l / 64 of the team otherwise
        constexpr int LANES = TEAM < 64 ? 64 : TEAM;
        for (int i = 0; i < PI::NB; ++i)
            for (int m = 0; m < PI::R; ++m)
                for (int w0 = 0; w0 < LANES; w0 += 64)
                    for (int g = 0; g < 2; ++g) {
                        std::vector<int> s;
                        for (int l = 0; l < 32; ++l) {
                            const int lane = w0 + g * 32 + l;
                            const int j = TEAM < 64 ? lane / TEAM : 0, t = TEAM < 64 ? lane % TEAM : lane;
                            int rot = (rotate && PI::LAST) ? ((t >> 3) & (PI::R - 1)) : 0;
                            int mm = PI::LAST ? (m + rot) % PI::R : m;
                            s.push_back(j * LdsFrame<N>::SIZE + LdsFrame<N>::at(PI::elem(t, i, mm)));
                        }
                        conf.rd += group_cycles(s, 32);
                        conf.rd_ideal += 1;
                    }
    }
    for (int t = 0; t < TEAM; ++t)
        pass_compute<N, P>(t, regs[t].data(), tw.data());
    if constexpr (!PI::LAST) {
        for (int t = 0; t < TEAM; ++t)
            pass_store<N, P>(t, regs[t].data(), frame.data());
        constexpr int LANES = TEAM < 64 ? 64 : TEAM;
        for (int i = 0; i < PI::NB; ++i)
            for (int q = 0; q < PI::R; ++q)
                for (int w0 = 0; w0 < LANES; w0 += 16) {
                    std::vector<int> s;
                    for (int l = 0; l < 16; ++l) {
                        const int lane = w0 + l;
                        const int j = TEAM < 64 ? lane / TEAM : 0, t = TEAM < 64 ? lane % TEAM : lane;
                        s.push_back(j * LdsFrame<N>::SIZE + LdsFrame<N>::at(PI::elem(t, i, q)));
                    }
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
    std::vector<cf> z(N), tw(N), frame(LdsFrame<N>::SIZE);
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

// team FFT of the fused kernel: numerics per team and LDS conflicts of a whole wavefront
// (TPW teams side by side, frames contiguous)
template <int N>
static int check_team()
{
    using T = TeamFft<N>;
    constexpr int TEAM = T::TEAM, TPW = T::TPW, FR = T::FRAME;
    std::vector<cf> z(N), frame(FR), tw0(T::TW0_SIZE), tw1(T::TW1_SIZE > 0 ? T::TW1_SIZE : 1);
    srand(7 + N);
    for (int i = 0; i < N; ++i) {
        z[i].re = (float)rand() / RAND_MAX - 0.5f;
        z[i].im = (float)rand() / RAND_MAX - 0.5f;
    }
    for (int c = 0; c < 4; ++c)
        for (int tl = 0; tl < TEAM; ++tl) {
            double a = -2.0 * M_PI * (double)(4 * tl + c) / (double)N;
            tw0[c * TEAM + tl] = {(float)cos(a), (float)sin(a)};
        }
    for (int q = 1; q < T::R1; ++q)
        for (int s = 0; s < 16; ++s) {
            double a = -2.0 * M_PI * (double)(s * q) / (double)T::L1;
            tw1[(q - 1) * 16 + s] = {(float)cos(a), (float)sin(a)};
        }
    std::vector<std::vector<cf>> regs(TEAM, std::vector<cf>(16));
    for (int t = 0; t < TEAM; ++t)
        for (int m = 0; m < 4; ++m)
            for (int c = 0; c < 4; ++c)
                regs[t][4 * m + c] = z[4 * t + c + (N / 4) * m];
    for (int t = 0; t < TEAM; ++t) T::pass0(t, regs[t].data(), tw0.data());
    for (int t = 0; t < TEAM; ++t) T::store0(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::load1(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::pass1(t, regs[t].data(), tw1.data());
    for (int t = 0; t < TEAM; ++t) T::store1(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::load2(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::pass2(regs[t].data());
    std::vector<std::complex<double>> w(N);
    for (int i = 0; i < N; ++i) w[i] = std::polar(1.0, -2.0 * M_PI * i / N);
    std::vector<double> got(N, -1.0), pw(N);
    double err = 0, pmax = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        for (int j = 0; j < N; ++j)
            acc += std::complex<double>(z[j].re, z[j].im) * w[(int)(((long)j * k) % N)];
        pw[k] = std::norm(acc);
        pmax = std::max(pmax, pw[k]);
    }
    for (int t = 0; t < TEAM; ++t)
        for (int q = 0; q < 16; ++q) {
            int k = T::freq_of(t, q);
            if (k < 0 || k >= N || got[k] >= 0) {
                printf("team<%d>: bad/duplicate k=%d\n", N, k);
                return 1;
            }
            got[k] = (double)regs[t][q].re * regs[t][q].re + (double)regs[t][q].im * regs[t][q].im;
        }
    for (int k = 0; k < N; ++k) err = std::max(err, fabs(got[k] - pw[k]) / pmax);
    // conflicts over a wavefront: lane l -> team l / TEAM at frame offset team * FR
    auto phys = [&](int l, int idx) { return (l / TEAM) * FR + idx; };
    long rd = 0, rdi = 0, wr = 0, wri = 0;
    auto rd_instr = [&](auto addr) {
        for (int g = 0; g < 2; ++g) {
            std::vector<int> sl;
            for (int l = 32 * g; l < 32 * g + 32; ++l) sl.push_back(addr(l));
            rd += group_cycles(sl, 32), ++rdi;
        }
    };
    auto wr_instr = [&](auto addr) {
        for (int g = 0; g < 4; ++g) {
            std::vector<int> sl;
            for (int l = 16 * g; l < 16 * g + 16; ++l) sl.push_back(addr(l));
            wr += group_cycles(sl, 16), ++wri;
        }
    };
    for (int q = 0; q < 4; ++q)
        for (int c = 0; c < 4; ++c)
            wr_instr([&](int l) { int tl = l % TEAM; return phys(l, 4 * tl + (tl >> 2) + (T::L1 + T::L1 / 16) * q + c); });
    for (int i = 0; i < T::NB1; ++i)
        for (int m = 0; m < T::R1; ++m) {
            auto a = [&](int l) { int tl = l % TEAM; return phys(l, T::base1(tl) + T::STEP1 * i + 17 * m); };
            rd_instr(a);
            wr_instr(a);
        }
    for (int m = 0; m < 16; ++m)
        rd_instr([&](int l) { int tl = l % TEAM; return phys(l, 17 * tl + m); });
    printf("team<%4d> (4,%2d,16) x%d per wave: max|dP|/Pmax=%.3g  lds read cycles %ld (ideal %ld)  write cycles %ld (ideal %ld)\n",
           N, T::R1, TPW, err, rd, rdi, wr, wri);
    return (err < 2e-6) ? 0 : 1;
}

// workgroup FFT of the large-N fused kernel
template <int N>
static int check_block()
{
    using T = BlockFft<N>;
    constexpr int TEAM = T::TEAM;
    std::vector<cf> z(N), frame(T::FRAME), tw0(T::TW0_SIZE), twa(T::TWA_SIZE), twb(T::TWB_SIZE);
    srand(11 + N);
    for (int i = 0; i < N; ++i) {
        z[i].re = (float)rand() / RAND_MAX - 0.5f;
        z[i].im = (float)rand() / RAND_MAX - 0.5f;
    }
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
    std::vector<typename T::Seeds> seeds(TEAM); // what a lane reads of the two global tables (the rest by products)
    for (int t = 0; t < TEAM; ++t) seeds[t] = T::load_seeds(t, tw0.data());
    for (int t = 0; t < TEAM; ++t) T::pass0(t, regs[t].data(), seeds[t]);
    for (int t = 0; t < TEAM; ++t) T::store0(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadA(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passA(t, regs[t].data(), T::load_seeds_a(t, twa.data()));
    for (int t = 0; t < TEAM; ++t) T::storeA(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadB(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passB(t, regs[t].data(), twb.data());
    for (int t = 0; t < TEAM; ++t) T::storeB(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::loadC(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::passC(regs[t].data());
    // reference via the f64 radix-2 of a simple recursive DFT (O(N log N) to keep N = 16384 quick)
    std::vector<std::complex<double>> x(N), y(N);
    for (int i = 0; i < N; ++i) x[i] = {z[i].re, z[i].im};
    {   // iterative f64 FFT
        int bits = 0;
        while ((1 << bits) < N) ++bits;
        for (int i = 0; i < N; ++i) {
            int r = 0;
            for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            y[r] = x[i];
        }
        for (int len = 2; len <= N; len <<= 1)
            for (int b = 0; b < N; b += len)
                for (int k = 0; k < len / 2; ++k) {
                    auto w = std::polar(1.0, -2.0 * M_PI * k / len);
                    auto u = y[b + k], t = w * y[b + k + len / 2];
                    y[b + k] = u + t;
                    y[b + k + len / 2] = u - t;
                }
    }
    std::vector<double> got(N, -1.0);
    double err = 0, pmax = 0;
    for (int k = 0; k < N; ++k) pmax = std::max(pmax, std::norm(y[k]));
    for (int t = 0; t < TEAM; ++t)
        for (int q = 0; q < 16; ++q) {
            int k = T::freq_of(t, q);
            if (k < 0 || k >= N || got[k] >= 0) {
                printf("block<%d>: bad/duplicate k=%d\n", N, k);
                return 1;
            }
            got[k] = (double)regs[t][q].re * regs[t][q].re + (double)regs[t][q].im * regs[t][q].im;
        }
    for (int k = 0; k < N; ++k) err = std::max(err, fabs(got[k] - std::norm(y[k])) / pmax);
    long rd = 0, rdi = 0, wr = 0, wri = 0;
    auto rd_instr = [&](auto addr) {
        for (int w0 = 0; w0 < TEAM; w0 += 32) {
            std::vector<int> sl;
            for (int l = w0; l < w0 + 32; ++l) sl.push_back(addr(l));
            rd += group_cycles(sl, 32), ++rdi;
        }
    };
    auto wr_instr = [&](auto addr) {
        for (int w0 = 0; w0 < TEAM; w0 += 16) {
            std::vector<int> sl;
            for (int l = w0; l < w0 + 16; ++l) sl.push_back(addr(l));
            wr += group_cycles(sl, 16), ++wri;
        }
    };
    for (int q = 0; q < 4; ++q)
        for (int c = 0; c < 4; ++c)
            wr_instr([&](int tl) { return 4 * tl + (tl >> 2) + T::STEP0 * q + c; });
    long rd0 = rd, rdi0 = rdi;
    for (int i = 0; i < T::NBA; ++i)
        for (int m = 0; m < T::RA; ++m) {
            auto a = [&](int tl) { return T::baseA(tl) + T::STEP0 * (TEAM / T::SA) * i + T::STEPA * m; };
            rd_instr(a);
            wr_instr(a);
        }
    if (getenv("FFT_EMUL_VERBOSE")) printf("   pass A reads %ld (ideal %ld)\n", rd - rd0, rdi - rdi0);
    rd0 = rd, rdi0 = rdi;
    for (int i = 0; i < T::NBB; ++i)
        for (int m = 0; m < T::RB; ++m) {
            auto a = [&](int tl) { return T::baseB(tl) + T::STEPA * i + 17 * m; };
            rd_instr(a);
            wr_instr(a);
        }
    if (getenv("FFT_EMUL_VERBOSE")) printf("   pass B reads %ld (ideal %ld)\n", rd - rd0, rdi - rdi0);
    rd0 = rd, rdi0 = rdi;
    for (int m = 0; m < 16; ++m)
        rd_instr([&](int tl) { return 17 * tl + m; });
    if (getenv("FFT_EMUL_VERBOSE")) printf("   pass C reads %ld (ideal %ld)\n", rd - rd0, rdi - rdi0);
    printf("block<%5d> (4,%2d,%2d,16): max|dP|/Pmax=%.3g  lds read cycles %ld (ideal %ld)  write cycles %ld (ideal %ld)\n",
           N, T::RA, T::RB, err, rd, rdi, wr, wri);
    return (err < 3e-6) ? 0 : 1;
}

// three-pass workgroup FFT (fft_block3.h): values against an f64 FFT, every bin exactly once, LDS bank conflicts
template <int N>
static int check_block3()
{
    using T = BlockFft3<N>;
    constexpr int TEAM = T::TEAM;
    std::vector<cf> z(N), frame(T::FRAME), tw0(T::TW0_SIZE), tw1(T::TW1_SIZE);
    srand(17 + N);
    for (int i = 0; i < N; ++i) {
        z[i].re = (float)rand() / RAND_MAX - 0.5f;
        z[i].im = (float)rand() / RAND_MAX - 0.5f;
    }
    for (int tl = 0; tl < TEAM; ++tl) {
        const double a = -2.0 * M_PI * (double)tl / (double)N;
        tw0[tl] = {(float)cos(a), (float)sin(a)};
        tw0[TEAM + tl] = {(float)cos(4 * a), (float)sin(4 * a)};
    }
    for (int q = 1; q < T::R1; ++q)
        for (int s = 0; s < 16; ++s) {
            const double a = -2.0 * M_PI * (double)(s * q) / (double)T::L1;
            tw1[(q - 1) * 16 + s] = {(float)cos(a), (float)sin(a)};
        }
    std::vector<std::vector<cf>> regs(TEAM, std::vector<cf>(16));
    for (int t = 0; t < TEAM; ++t)
        for (int m = 0; m < 16; ++m)
            regs[t][m] = z[t + TEAM * m];
    for (int t = 0; t < TEAM; ++t) T::pass0(regs[t].data(), T::load_seeds(t, tw0.data()));
    for (int t = 0; t < TEAM; ++t) T::store0(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::load1(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::pass1(t, regs[t].data(), tw1.data());
    for (int t = 0; t < TEAM; ++t) T::store1(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::load2(t, regs[t].data(), frame.data());
    for (int t = 0; t < TEAM; ++t) T::pass2(regs[t].data());
    std::vector<std::complex<double>> x(N), y(N);
    for (int i = 0; i < N; ++i) x[i] = {z[i].re, z[i].im};
    {
        int bits = 0;
        while ((1 << bits) < N) ++bits;
        for (int i = 0; i < N; ++i) {
            int r = 0;
            for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            y[r] = x[i];
        }
        for (int len = 2; len <= N; len <<= 1)
            for (int b = 0; b < N; b += len)
                for (int k = 0; k < len / 2; ++k) {
                    auto w = std::polar(1.0, -2.0 * M_PI * k / len);
                    auto u = y[b + k], t = w * y[b + k + len / 2];
                    y[b + k] = u + t;
                    y[b + k + len / 2] = u - t;
                }
    }
    std::vector<double> got(N, -1.0);
    double err = 0, pmax = 0, cerr = 0;
    for (int k = 0; k < N; ++k) pmax = std::max(pmax, std::norm(y[k]));
    for (int t = 0; t < TEAM; ++t)
        for (int q = 0; q < 16; ++q) {
            const int k = T::freq_of(t, q);
            if (k < 0 || k >= N || got[k] >= 0) {
                printf("block3<%d>: bad/duplicate k=%d\n", N, k);
                return 1;
            }
            got[k] = (double)regs[t][q].re * regs[t][q].re + (double)regs[t][q].im * regs[t][q].im;
            cerr = std::max(cerr, std::abs(std::complex<double>(regs[t][q].re, regs[t][q].im) - y[k]) / std::sqrt(pmax));
        }
    for (int k = 0; k < N; ++k) err = std::max(err, fabs(got[k] - std::norm(y[k])) / pmax);
    long rd = 0, rdi = 0, wr = 0, wri = 0;
    auto rd_instr = [&](auto addr) {
        for (int w0 = 0; w0 < TEAM; w0 += 32) {
            std::vector<int> sl;
            for (int l = w0; l < w0 + 32; ++l) sl.push_back(addr(l));
            rd += group_cycles(sl, 32), ++rdi;
        }
    };
    auto wr_instr = [&](auto addr) {
        for (int w0 = 0; w0 < TEAM; w0 += 16) {
            std::vector<int> sl;
            for (int l = w0; l < w0 + 16; ++l) sl.push_back(addr(l));
            wr += group_cycles(sl, 16), ++wri;
        }
    };
    for (int q = 0; q < 16; ++q)
        wr_instr([&](int tl) { return tl + (tl >> 4) + T::STEP1 * q; });
    for (int i = 0; i < T::NB1; ++i)
        for (int m = 0; m < T::R1; ++m) {
            auto a = [&](int tl) { return T::base1(tl) + T::STEP1 * i + 17 * m; };
            rd_instr(a);
            wr_instr(a);
        }
    for (int m = 0; m < 16; ++m)
        rd_instr([&](int tl) { return 17 * tl + m; });
    printf("block3<%5d> (16,%2d,16): max|dP|/Pmax=%.3g max|dX|/sqrt(Pmax)=%.3g  lds read cycles %ld (ideal %ld)  write cycles %ld (ideal %ld)\n",
           N, T::R1, err, cerr, rd, rdi, wr, wri);
    return (err < 3e-6 && cerr < 3e-6) ? 0 : 1;
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
    if (getenv("FFT_EMUL_ROT_ALL")) {
        bad |= check<256>(true);
        bad |= check<512>(true);
        bad |= check<2048>(true);
        bad |= check<4096>(true);
        bad |= check<8192>(true);
        bad |= check<16384>(true);
    }
    bad |= check<2048>(false);
    bad |= check<4096>(false);
    bad |= check<8192>(false);
    bad |= check<16384>(false);
    bad |= check_team<256>();
    bad |= check_team<512>();
    bad |= check_team<1024>();
    bad |= check_block<2048>();
    bad |= check_block<4096>();
    bad |= check_block<8192>();
    bad |= check_block<16384>();
    bad |= check_block3<2048>();
    bad |= check_block3<4096>();
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
