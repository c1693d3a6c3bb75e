// Checks stabilizer-stream_amd/csrc/plan.h against a literal simulation of the
// reference's per-call segment loop (src/psd.rs:196-269) and EWMA recurrence
// (src/psd.rs:218-233).  Build: g++ -O2 -std=c++17 -I<csrc>.
#include "plan.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace psdk;

struct SimStage { // literal counters of Psd::process
    uint32_t n, ov;
    uint64_t idx = 0, count = 0, drain = 35, decimated = 0, emitted = 0;
    uint64_t process(uint64_t len)
    {
        uint64_t nout = 0;
        while (len > 0) {
            uint64_t take = std::min<uint64_t>(len, n - idx);
            len -= take;
            idx += take;
            if (idx < n) break;
            bool first = count == 0;
            count += 1;
            uint64_t start = first ? 0 : ov;
            uint64_t nb = (n - start) / 8;
            decimated += n - start;
            uint64_t skip = std::min(drain, nb);
            drain -= skip;
            nout += nb - skip;
            idx = ov;
        }
        emitted += nout;
        return nout;
    }
};

int main()
{
    std::mt19937_64 rng(7);
    int bad = 0;
    // stream bookkeeping
    for (int trial = 0; trial < 400; ++trial) {
        uint32_t n = 16u << (rng() % 8);
        bool hann = rng() & 1;
        Geometry g;
        g.n = n;
        g.overlap = hann ? n / 2 : 0;
        g.hop = n - g.overlap;
        SimStage s{n, g.overlap};
        uint64_t total = 0;
        for (int c = 0; c < 60; ++c) {
            uint64_t len = rng() % (5 * n);
            if (rng() % 7 == 0) len = 0;
            s.process(len);
            total += len;
            uint64_t J = segments_for(g, total), P = decimated_prefix(g, J);
            if (J != s.count || P != s.decimated || emitted_for(g, P) != s.emitted ||
                pending_for(g, total) != s.idx) {
                printf("bookkeeping mismatch n=%u ov=%u total=%lu: J %lu/%lu P %lu/%lu E %lu/%lu pend %lu/%lu\n",
                       n, g.overlap, (unsigned long)total, (unsigned long)J, (unsigned long)s.count,
                       (unsigned long)P, (unsigned long)s.decimated,
                       (unsigned long)emitted_for(g, P), (unsigned long)s.emitted,
                       (unsigned long)pending_for(g, total), (unsigned long)s.idx);
                bad = 1;
            }
        }
    }
    // EWMA plan
    std::uniform_real_distribution<double> U(0.1, 2.0);
    for (int trial = 0; trial < 20000; ++trial) {
        uint32_t avg = (trial % 5 == 0) ? (uint32_t)(rng() % 3) : (uint32_t)(rng() % 40);
        if (trial % 97 == 0) avg = 0xFFFFFFFFu;
        uint32_t c0 = (uint32_t)(rng() % 60);
        uint64_t nb = rng() % 70;
        EwmaPlan p = plan_ewma(c0, avg, nb);
        double p0 = U(rng), ref = p0, mine = p.g_total * p0;
        uint32_t count = c0;
        bool any = false;
        for (uint64_t i = 1; i <= nb; ++i) {
            double s = U(rng);
            float g;
            if (count > avg) { // src/psd.rs:218-224
                g = (float)avg / (float)count;
                count = avg;
            } else
                g = 1.0f;
            count += 1;
            if (g != 1.0f) any = true;
            ref = (double)g * ref + s;
            mine += ewma_weight(p, (int64_t)i) * s;
        }
        if (count != count_after(c0, avg, nb)) {
            printf("count mismatch c0=%u avg=%u nb=%lu: %u vs %u\n", c0, avg, (unsigned long)nb, count,
                   count_after(c0, avg, nb));
            bad = 1;
        }
        if (any != p.ewma && nb > 0) {
            printf("ewma flag mismatch c0=%u avg=%u nb=%lu\n", c0, avg, (unsigned long)nb);
            bad = 1;
        }
        if (std::fabs(ref - mine) > 1e-9 * std::fabs(ref) + 1e-12) {
            printf("ewma mismatch c0=%u avg=%u nb=%lu: %.12g vs %.12g\n", c0, avg, (unsigned long)nb, ref, mine);
            bad = 1;
        }
    }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
