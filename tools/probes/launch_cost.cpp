// tools/probes/launch_cost.cpp -- host time of one kernel launch against the size of its by-value argument (the fused and post launches
// carry their job tables in the kernel-argument segment: ~9 KiB and ~10 KiB).  hipcc -O2 launch_cost.cpp -o launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
template <int W>
struct Arg {
    int v[W];
};
template <int W>
__global__ void k(const Arg<W> a, int *out)
{
    if (a.v[0] == 12345 && threadIdx.x == 0)
        out[0] = a.v[W - 1];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int W>
void run(hipStream_t s, int *d, bool ext)
{
    Arg<W> a{};
    const int reps = 2000;
    for (int w = 0; w < 200; ++w)
        hipLaunchKernelGGL(k<W>, dim3(64), dim3(64), 0, s, a, d);
    (void)hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < reps; ++i) {
        if (ext)
            hipExtLaunchKernelGGL(k<W>, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, a, d);
        else
            hipLaunchKernelGGL(k<W>, dim3(64), dim3(64), 0, s, a, d);
        if ((i & 15) == 15)
            (void)hipStreamSynchronize(s); // (never queue-bound: the host cost of the call itself)
    }
    double t1 = now();
    (void)hipStreamSynchronize(s);
    printf("arg %6zu B %s: %.2f us per launch (sync every 16 included)\n", sizeof(a), ext ? "ext  " : "plain", 1e6 * (t1 - t0) / reps);
}
int main()
{
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d;
    (void)hipMalloc(&d, 64);
    for (int ext = 0; ext < 2; ++ext) {
        run<16>(s, d, ext);
        run<256>(s, d, ext);
        run<1024>(s, d, ext);
        run<2400>(s, d, ext);
        run<4000>(s, d, ext);
    }
    return 0;
}
