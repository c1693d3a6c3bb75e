// Per-call cost of the host-fed boundary at the chunk sizes the reference's own callers use: Source::get hands
// 512 samples per call for Data::Raw (src/source.rs:150-157) and one frame per call for Data::File, and
// bin/psd.rs:181 passes each straight to PsdCascade::process.  Feeds 2^26 samples in chunks of 512, 4096,
// 65536 (the reference's `insn` bench, src/psd.rs:554-559) and 2^22 samples and prints MS/s and ns per call.
// g++ -O2 -I../include smallcall_probe.cpp -L../stabilizer-stream_amd -lpsdcascade -Wl,-rpath,... (see scripts_gpu_host.sh)
#include "psdcascade.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const size_t total = (size_t)1 << 26;
    std::vector<float> x(total);
    unsigned s = 12345;
    for (auto &v : x) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4641016f;
    }
    for (int n : {512, 1024}) {
        for (size_t chunk : {(size_t)512, (size_t)4096, (size_t)65536, (size_t)1 << 22}) {
            psdc_handle *h = psdc_create(n, PSDC_WINDOW_HANN, 1, 0);
            if (!h) {
                fprintf(stderr, "psdc_create failed\n");
                return 1;
            }
            psdc_process(h, 0, x.data(), total); // first-use costs (staging buffers, copy threads, stage pools)
            psdc_sync(h);
            const double t0 = now();
            size_t calls = 0, fed = 0;
            while (now() - t0 < 0.4) // whole passes over the buffer for at least 0.4 s
                for (size_t a = 0; a < total; a += chunk, ++calls, fed += chunk)
                    if (psdc_process(h, 0, x.data() + a, chunk) != PSDC_OK) {
                        fprintf(stderr, "psdc_process: %s\n", psdc_last_error(h));
                        return 1;
                    }
            const double t1 = now();
            psdc_sync(h);
            const double t2 = now();
            printf("N=%d chunk %8zu: %7.0f MS/s to the last call's return (%6.0f ns per call), %7.0f MS/s with the drain\n", n,
                   chunk, fed / (t1 - t0) / 1e6, (t1 - t0) / calls * 1e9, fed / (t2 - t0) / 1e6);
            psdc_destroy(h);
        }
    }
    return 0;
}
