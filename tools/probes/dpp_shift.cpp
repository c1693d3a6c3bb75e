// Semantics probe: wave_shr:1 with a fill for lane 0 taken from a wave_ror:1 of another register
// (what the register-resident stage A of the decimator relies on).  hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float shr1(float fill, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float ror1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}
__global__ void k(const float *a, const float *b, float *o)
{
    const int l = threadIdx.x;
    const float prev = a[l], cur = b[l];
    const float s1 = shr1(ror1(prev), cur);
    const float s2 = shr1(ror1(shr1(0.0f, prev)), s1);
    o[l] = s1;
    o[64 + l] = s2;
}
int main()
{
    float ha[64], hb[64], ho[128];
    for (int i = 0; i < 64; ++i) {
        ha[i] = 100 + i;
        hb[i] = 200 + i;
    }
    float *a, *b, *o;
    hipMalloc(&a, 256);
    hipMalloc(&b, 256);
    hipMalloc(&o, 512);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float e1 = l >= 1 ? hb[l - 1] : ha[63];
        const float e2 = l >= 2 ? hb[l - 2] : ha[62 + l];
        if (ho[l] != e1 || ho[64 + l] != e2) {
            printf("lane %d: s1 %g (want %g) s2 %g (want %g)\n", l, ho[l], e1, ho[64 + l], e2);
            ++bad;
        }
    }
    printf(bad ? "MISMATCH\n" : "dpp shifts ok\n");
    return bad != 0;
}
