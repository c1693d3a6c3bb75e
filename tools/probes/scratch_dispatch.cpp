// tools/probes/scratch_dispatch.cpp -- does a kernel that uses scratch (a few spilled dwords) cost more to DISPATCH?  Two otherwise
// identical near-empty kernels on the headline's grid (520 workgroups of 512 threads, 80 KiB of LDS each), one with a private array the
// compiler must keep in scratch; run under `rocprofv3 --kernel-trace` and compare the kernels' durations (start to end of the dispatch).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_plain(int *out, int sel)
{
    __shared__ float lds[20000];
    lds[threadIdx.x] = (float)sel;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == 12345.0f)
        out[blockIdx.x] = 1;
}
__global__ __launch_bounds__(512) void k_scratch(int *out, int sel)
{
    __shared__ float lds[20000];
    volatile int priv[8];
    for (int i = 0; i < 8; ++i)
        priv[i] = sel + i;
    lds[threadIdx.x] = (float)priv[(sel + threadIdx.x) & 7]; // dynamic index: stays in scratch
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == 12345.0f)
        out[blockIdx.x] = 1;
}
int main()
{
    int *d;
    (void)hipMalloc(&d, 4096 * 4);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int r = 0; r < 200; ++r) {
        hipLaunchKernelGGL(k_plain, dim3(520), dim3(512), 0, s, d, r);
        hipLaunchKernelGGL(k_scratch, dim3(520), dim3(512), 0, s, d, r);
        if ((r & 7) == 7)
            (void)hipStreamSynchronize(s);
    }
    (void)hipStreamSynchronize(s);
    printf("done\n");
    return 0;
}
