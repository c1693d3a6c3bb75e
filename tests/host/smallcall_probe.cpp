// tests/host/smallcall_probe.cpp -- the drop-in boundary at the granularity the reference's own callers use it: Source::get hands
// 512 samples per call for Data::Raw (src/source.rs:150-157) and one frame's worth for Data::File (src/source.rs:136-142), and
// src/bin/psd.rs:172-181 passes each straight to PsdCascade::process; the reference's own `insn` bench feeds 65536-sample calls
// (src/psd.rs:554-559).  Feeds a host buffer through psdc_process (HOST memory: staging copy + upload + cascade) in calls of 512,
// 4096 and 65536 samples from plain C++ -- no Python in the loop -- and prints ONE JSON line: MS/s to the last call's return,
// MS/s with the drain (psdc_sync), ns per call.  bench.py runs it after the timed region (`host_fed_small`).
//   usage: smallcall_probe [n = 1024] [seconds per size = 0.5] [device = 0]
#include "psdcascade.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 1024u;
    const double secs = argc > 2 ? atof(argv[2]) : 0.5;
    const int device = argc > 3 ? atoi(argv[3]) : 0;
    const size_t total = (size_t)1 << 26;
    std::vector<float> x(total);
    unsigned s = 12345;
    for (auto &v : x) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4641016f;
    }
    printf("{\"n\": %u, \"unit\": \"MS/s\", \"calls\": {", n);
    bool first = true;
    for (size_t chunk : {(size_t)512, (size_t)4096, (size_t)65536, (size_t)1 << 22}) {
        psdc_handle *h = psdc_create(n, PSDC_WINDOW_HANN, 1, device);
        if (!h) {
            fprintf(stderr, "psdc_create failed: %s\n", psdc_last_error(nullptr));
            return 1;
        }
        psdc_process(h, 0, x.data(), total); // first-use costs (staging buffers, copy threads, stage pools)
        psdc_sync(h);
        const double t0 = now();
        size_t calls = 0, fed = 0;
        while (now() - t0 < secs) // whole passes over the buffer
            for (size_t a = 0; a < total; a += chunk, ++calls, fed += chunk)
                if (psdc_process(h, 0, x.data() + a, chunk) != PSDC_OK) {
                    fprintf(stderr, "psdc_process: %s\n", psdc_last_error(h));
                    return 1;
                }
        const double t1 = now();
        psdc_sync(h);
        const double t2 = now();
        printf("%s\"%zu\": {\"to_return\": %.0f, \"with_drain\": %.0f, \"ns_per_call\": %.1f}", first ? "" : ", ", chunk,
               fed / (t1 - t0) / 1e6, fed / (t2 - t0) / 1e6, (t1 - t0) / calls * 1e9);
        first = false;
        psdc_destroy(h);
    }
    printf("}}\n");
    return 0;
}
