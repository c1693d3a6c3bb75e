// Issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950: is a packed f32 op
// two results for the price of one (64 FLOP/clk/SIMD) or two passes?  Prints lane-results per clock per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.cpp -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 65536
#define ACC 8
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float seed)
{
    f2 a[ACC];
    for (int i = 0; i < ACC; ++i)
        a[i] = f2{seed + i + threadIdx.x, seed - i};
    f2 b = {seed * 0.5f, seed * 0.25f}, c = {seed * 0.125f, seed};
    const float sc = __builtin_amdgcn_readfirstlane(seed * 0.75f);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if constexpr (MODE == 0) { // two scalar fma
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
            } else if constexpr (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            } else if constexpr (MODE == 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if constexpr (MODE == 3) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if constexpr (MODE == 4) { // two scalar add
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if constexpr (MODE == 5) { // packed fma with op_sel swizzle, as a complex multiply would use it
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            } else if constexpr (MODE == 6) { // VOP2 accumulate form, three VGPR reads
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
            } else if constexpr (MODE == 7) { // one factor in an SGPR
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "s"(sc), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "s"(sc), "v"(c.y));
            } else if constexpr (MODE == 8) { // accumulate form, one factor in an SGPR
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "s"(sc), "v"(c.x));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "s"(sc), "v"(c.y));
            } else if constexpr (MODE == 9) { // x*x + acc: two distinct VGPRs
                asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if constexpr (MODE == 10) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if constexpr (MODE == 11) { // literal constant factor
                asm volatile("v_fmamk_f32 %0, %0, 0x3f6c835e, %1" : "+v"(a[i].x) : "v"(c.x));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f6c835e, %1" : "+v"(a[i].y) : "v"(c.y));
            } else { // sub then mul, the window's shape
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(c.y));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < ACC; ++i)
        s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> static void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256 CUs, 4 waves per block = 1 wave per SIMD per block
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int mhz = 0;
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0); // kHz
    const double clk = ms * 1e-3 * (mhz * 1e3);
    const double results = double(ITERS) * ACC * 2 * 64 * waves_per_simd; // lane results per SIMD
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f lane-results/clk/SIMD (at %d MHz nominal)\n", name, waves_per_simd, ms,
           results / clk, mhz / 1000);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("2 x v_fma_f32 (v,v,v)", w);
        run<6>("2 x v_fmac_f32 (v,v)", w);
        run<7>("2 x v_fma_f32 (v,s,v)", w);
        run<8>("2 x v_fmac_f32 (s,v)", w);
        run<11>("2 x v_fmamk_f32 (v,lit,v)", w);
        run<9>("2 x v_fmac_f32 (v,same v)", w);
        run<10>("2 x v_mul_f32", w);
        run<4>("2 x v_add_f32", w);
        run<12>("v_sub_f32 + v_mul_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<5>("v_pk_fma_f32 op_sel", w);
        run<2>("v_pk_mul_f32", w);
        run<3>("v_pk_add_f32", w);
    }
    return 0;
}
