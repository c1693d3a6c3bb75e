// tools/probes/stream_query.cpp -- host cost of asking "is the stream idle" three ways: hipStreamQuery, hipEventQuery on the last
// recorded event, and a flag in coherent pinned host memory that a one-thread kernel behind the work writes.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/stream_query.cpp -o /tmp/stream_query && /tmp/stream_query
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long cycles, unsigned *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles)
        ;
    if (sink)
        *sink = 1;
}
__global__ void signal_kernel(volatile unsigned *flag, unsigned v)
{
    *flag = v;
    __threadfence_system();
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    unsigned *flag = nullptr;
    hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocCoherent);
    *flag = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const int reps = 20000;
    for (int busy = 0; busy < 2; ++busy) {
        if (busy) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 100000000LL * 3, nullptr); // ~3 s at 100 MHz wall clock
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flag, 2u);
            hipEventRecord(ev, s);
        } else {
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flag, 1u);
            hipEventRecord(ev, s);
            hipStreamSynchronize(s);
        }
        int notready = 0;
        auto t0 = now();
        for (int i = 0; i < reps; ++i)
            notready += hipStreamQuery(s) != hipSuccess;
        auto t1 = now();
        for (int i = 0; i < reps; ++i)
            notready += hipEventQuery(ev) != hipSuccess;
        auto t2 = now();
        unsigned seen = 0;
        for (int i = 0; i < reps; ++i)
            seen += *(volatile unsigned *)flag;
        auto t3 = now();
        printf("%s stream: hipStreamQuery %.2f us, hipEventQuery %.2f us, host flag %.4f us per call (not-ready answers %d, flag %u)\n",
               busy ? "busy" : "idle", us(t0, t1) / reps, us(t1, t2) / reps, us(t2, t3) / reps, notready, seen / reps);
    }
    // a launch of the one-thread signal kernel: host cost
    auto t0 = now();
    for (int i = 0; i < 2000; ++i)
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flag, 3u);
    auto t1 = now();
    printf("signal kernel launch: %.2f us host per launch (behind a busy stream)\n", us(t0, t1) / 2000);
    hipStreamSynchronize(s);
    printf("after sync flag = %u\n", *flag);
    return 0;
}
