// tests/host/devcall_probe.cpp -- psdc_process_device at call sizes from 2^26 down to 2^16 samples, from C++ (no Python in the
// loop): a live stream does not arrive in 256 MiB spans.  One device buffer of 2^26 samples (psdc_fill_noise_device) is fed as
// in-place spans of the given size, pass after pass, for `seconds` per size, in two orders: "contiguous" -- consecutive pieces, as a
// ring or capture buffer is handed over (each call continues the last one in memory and extends the held span: PSDC_OPT_MERGE) --
// and "scattered" -- the same pieces in an order in which no call continues the one before it (every call a span of its own, up to
// PSDC_OPT_COALESCE of them per round).  Prints ONE JSON line with MS/s to the drain (psdc_sync) and ns of host time per call.
// bench.py runs it after the timed region (`device_fed_calls`).
//   usage: devcall_probe [n = 1024] [seconds per size = 0.4] [device = 0] [eager = 0] [only log2 size = 0: all] [PSDC_OPT_COALESCE = 0: the library's own depth] [which = 0: both orders, 1: contiguous only, 2: scattered only]
#include "psdcascade.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 1024u;
    const double secs = argc > 2 ? atof(argv[2]) : 0.4;
    const int device = argc > 3 ? atoi(argv[3]) : 0;
    const int eager = argc > 4 ? atoi(argv[4]) : 0;
    const int only = argc > 5 ? atoi(argv[5]) : 0;
    const int coalesce = argc > 6 ? atoi(argv[6]) : 0;
    const int which = argc > 7 ? atoi(argv[7]) : 0;
    const size_t total = (size_t)1 << 26;
    float *d = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, total * sizeof(float)) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    if (psdc_fill_noise_device(device, d, total, 0x7654321, 0) != PSDC_OK || hipDeviceSynchronize() != hipSuccess) {
        fprintf(stderr, "fill failed\n");
        return 1;
    }
    printf("{\"n\": %u, \"unit\": \"MS/s\", \"eager\": %d", n, eager);
    for (int scattered = 0; scattered < 2; ++scattered) {
        if (which && which != scattered + 1)
            continue;
        printf(", \"%s\": {", scattered ? "scattered" : "contiguous");
        bool first = true;
        for (int lg : {26, 24, 22, 20, 18, 16}) {
            if (only && lg != only)
                continue;
            const size_t chunk = (size_t)1 << lg, nchunks = total / chunk;
            // scattered: piece i of a pass is chunk (i * step) mod nchunks, step odd and > 1: a permutation in which no piece follows
            // its predecessor in memory (one chunk: the same span again and again, which does not continue itself either)
            const size_t step = nchunks >= 4 ? (nchunks / 2 + 1) | 1 : 1;
            psdc_handle *h = psdc_create(n, PSDC_WINDOW_HANN, 1, device);
            if (!h) {
                fprintf(stderr, "psdc_create failed: %s\n", psdc_last_error(nullptr));
                return 1;
            }
            if (eager)
                psdc_configure(h, PSDC_OPT_EAGER, 1);
            if (coalesce)
                psdc_configure(h, PSDC_OPT_COALESCE, coalesce);
            auto piece = [&](size_t i) { return d + (scattered ? (i * step) % nchunks : i) * chunk; };
            for (int w = 0; w < 2; ++w) // first-use costs: stream buffers grown to the round size, clocks
                for (size_t i = 0; i < nchunks; ++i)
                    psdc_process_device(h, 0, piece(i), chunk);
            psdc_sync(h);
            const double t0 = now();
            size_t calls = 0, fed = 0;
            while (now() - t0 < secs)
                for (size_t i = 0; i < nchunks; ++i, ++calls, fed += chunk)
                    if (psdc_process_device(h, 0, piece(i), chunk) != PSDC_OK) {
                        fprintf(stderr, "psdc_process_device: %s\n", psdc_last_error(h));
                        return 1;
                    }
            const double t1 = now();
            psdc_sync(h);
            const double t2 = now();
            printf("%s\"2^%d\": {\"with_drain\": %.0f, \"host_ns_per_call\": %.0f}", first ? "" : ", ", lg, fed / (t2 - t0) / 1e6,
                   (t1 - t0) / calls * 1e9);
            first = false;
            psdc_destroy(h);
        }
        printf("}");
    }
    printf("}\n");
    (void)hipFree(d);
    return 0;
}
