// Where does the host-fed path spend its time?  memcpy into pinned staging (1..4 threads) and
// pinned H2D bandwidth, at the library's staging quantum (2^22 floats = 16 MiB).
// hipcc -O2 -o hostfed_probe hostfed_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const size_t q = (size_t)1 << 24; // bytes per quantum
    const size_t total = (size_t)1 << 28;
    std::vector<char> src(total, 1);
    char *pin[2];
    char *dev;
    hipHostMalloc((void **)&pin[0], q, hipHostMallocDefault);
    hipHostMalloc((void **)&pin[1], q, hipHostMallocDefault);
    hipMalloc((void **)&dev, q);
    memset(pin[0], 0, q);
    memset(pin[1], 0, q);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int nt = 1; nt <= 8; nt *= 2) {
        double t0 = now();
        for (int rep = 0; rep < 4; ++rep)
            for (size_t o = 0; o < total; o += q) {
                std::vector<std::thread> th;
                const size_t part = q / nt;
                for (int t = 1; t < nt; ++t)
                    th.emplace_back([&, t] { memcpy(pin[0] + t * part, src.data() + o + t * part, part); });
                memcpy(pin[0], src.data() + o, part);
                for (auto &x : th)
                    x.join();
            }
        double dt = now() - t0;
        printf("memcpy -> pinned, %d thread(s): %.1f GB/s\n", nt, 4.0 * total / dt / 1e9);
    }
    {
        hipMemcpyAsync(dev, pin[0], q, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < 64; ++i)
            hipMemcpyAsync(dev, pin[i & 1], q, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double dt = now() - t0;
        printf("H2D pinned 16 MiB x 64: %.1f GB/s\n", 64.0 * q / dt / 1e9);
    }
    {
        // pipelined: memcpy(1 thread) into one buffer while the other is in flight
        hipEvent_t ev[2];
        hipEventCreate(&ev[0]);
        hipEventCreate(&ev[1]);
        double t0 = now();
        int b = 0;
        for (int rep = 0; rep < 4; ++rep)
            for (size_t o = 0; o < total; o += q) {
                hipEventSynchronize(ev[b]);
                memcpy(pin[b], src.data() + o, q);
                hipMemcpyAsync(dev, pin[b], q, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[b], s);
                b ^= 1;
            }
        hipStreamSynchronize(s);
        double dt = now() - t0;
        printf("memcpy(1 thread) + H2D double-buffered: %.1f GB/s\n", 4.0 * total / dt / 1e9);
    }
    {
        // direct H2D from pageable memory
        double t0 = now();
        for (size_t o = 0; o < total; o += q)
            hipMemcpyAsync(dev, src.data() + o, q, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double dt = now() - t0;
        printf("H2D from pageable memory: %.1f GB/s\n", (double)total / dt / 1e9);
    }
    return 0;
}
