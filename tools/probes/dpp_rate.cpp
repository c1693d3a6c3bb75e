// Issue cost of the DPP forms the N = 1024 decimator uses (wave_shr:1 / wave_ror:1 wavefront shifts) against row-level
// DPP and plain moves on gfx950, in shader cycles per instruction per wave (s_memtime around the loop), at 1, 2 and 4
// wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 dpp_rate.cpp -o dpp_rate && ./dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 8192
#define ACC 8
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float seed)
{
    float a[ACC], b[ACC];
    for (int i = 0; i < ACC; ++i) {
        a[i] = seed + i + threadIdx.x;
        b[i] = seed - i;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if constexpr (MODE == 0)
                asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 1)
                asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 2)
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 3)
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 4)
                asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 5)
                asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 6)
                asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 7)
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 8) // the pair the decimator issues per shifted value: ror, then shr filled from it
                asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\t"
                             "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % ACC]));
            else if constexpr (MODE == 9)
                asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            else if constexpr (MODE == 10)
                asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)\n\ts_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(b[i]));
            else
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < ACC; ++i)
        s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char *name, int per_instr)
{
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w;
        float *out;
        unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * 256);
        hipMallocManaged(&cyc, sizeof(unsigned long long) * blocks);
        k<MODE><<<blocks, 256>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        k<MODE><<<blocks, 256>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        double sum = 0;
        for (int i = 0; i < blocks; ++i)
            sum += (double)cyc[i];
        printf("%-34s waves/SIMD %d: %6.2f cycles per instruction per wave, %5.2f per SIMD\n", name, w,
               sum / blocks / ((double)ITERS * ACC * per_instr), sum / blocks / ((double)ITERS * ACC * per_instr) / w);
        hipFree(out);
        hipFree(cyc);
    }
}
int main()
{
    run<0>("v_mov_b32", 1);
    run<7>("v_add_f32", 1);
    run<1>("v_mov_b32_dpp quad_perm", 1);
    run<2>("v_mov_b32_dpp row_shr:1", 1);
    run<3>("v_mov_b32_dpp wave_shr:1", 1);
    run<4>("v_mov_b32_dpp wave_ror:1", 1);
    run<9>("v_mov_b32_dpp row_bcast:15", 1);
    run<5>("v_add_f32_dpp row_shr:1", 1);
    run<6>("v_add_f32_dpp wave_shr:1", 1);
    run<8>("ror + nop + shr (per pair)", 1);
    run<10>("ds_swizzle_b32 + wait", 1);
    run<11>("v_permlane32_swap_b32", 1);
    return 0;
}
