// frame_load.hip -- GPU probe of the AdcDac frame addressing of csrc/frames.h: decodes trace ch of a run of frames (a) sample
// by sample (frame_sample) and (b) by the fused kernels' 8-byte buffer loads of four wire words, against a host decode.
// hipcc --offload-arch=gfx950 -O3 -I../../stabilizer-stream_amd/csrc frame_load.hip -o frame_load && ./frame_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "frames.h"
using namespace psdk;

__global__ void k_sample(FrameSpan fs, int ch, float *out, unsigned n)
{
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        out[i] = frame_sample(fs, ch, i);
}
__global__ void k_buffer(FrameSpan fs, int ch, float *out, unsigned n)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fs.frames), 0, (int)fs.bytes, 0x00020000);
    const unsigned flip = ch >= 2 ? 0x80008000u : 0u;
    for (unsigned q = blockIdx.x * 256 + threadIdx.x; 4 * q < n; q += gridDim.x * 256) {
        const unsigned si = 4 * q;
        const unsigned off = frame_cell_offset(fs, si >> 3) + (unsigned)ch * 16u + (si & 4u) * 2u;
        const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
        const unsigned a = (unsigned)r[0] ^ flip, b = (unsigned)r[1] ^ flip;
        const float lsb = adcdac_lsb();
        out[si] = (float)(short)(unsigned short)(a & 0xffffu) * lsb;
        out[si + 1] = (float)(short)(unsigned short)(a >> 16) * lsb;
        out[si + 2] = (float)(short)(unsigned short)(b & 0xffffu) * lsb;
        out[si + 3] = (float)(short)(unsigned short)(b >> 16) * lsb;
    }
}

int main()
{
    int bad = 0;
    for (unsigned batches : {1u, 7u, 22u, 31u}) {
        const unsigned nframes = 5000, fsz = 8 + 64 * batches, per = batches * 8 * nframes;
        std::vector<uint8_t> h((size_t)nframes * fsz);
        std::vector<short> tr[4];
        srand(batches);
        for (int c = 0; c < 4; ++c) tr[c].resize(per);
        for (unsigned f = 0; f < nframes; ++f) {
            uint8_t *p = h.data() + (size_t)f * fsz;
            p[0] = 0x7b; p[1] = 5; p[2] = 1; p[3] = (uint8_t)batches;
            for (unsigned b = 0; b < batches; ++b)
                for (int c = 0; c < 4; ++c)
                    for (int i = 0; i < 8; ++i) {
                        const short v = (short)(rand() & 0xffff);
                        tr[c][(size_t)(f * batches + b) * 8 + i] = v;
                        p[8 + b * 64 + c * 16 + 2 * i] = (uint8_t)(v & 0xff);
                        p[8 + b * 64 + c * 16 + 2 * i + 1] = (uint8_t)((unsigned short)v >> 8);
                    }
        }
        uint8_t *d; float *o;
        hipMalloc(&d, h.size()); hipMalloc(&o, sizeof(float) * per);
        hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
        FrameSpan fs{d, (unsigned long long)h.size(), fsz, batches, batches >= 2 ? (unsigned)((0x100000000ull + batches - 1) / batches) : 0u, 0};
        std::vector<float> got(per);
        for (int ch = 0; ch < 4; ++ch)
            for (int mode = 0; mode < 2; ++mode) {
                hipMemset(o, 0xff, sizeof(float) * per);
                if (mode == 0) hipLaunchKernelGGL(k_sample, dim3(256), dim3(256), 0, 0, fs, ch, o, per);
                else hipLaunchKernelGGL(k_buffer, dim3(256), dim3(256), 0, 0, fs, ch, o, per);
                hipMemcpy(got.data(), o, sizeof(float) * per, hipMemcpyDeviceToHost);
                size_t nb = 0, first = 0;
                const float lsb = 4.096f * 2.5f / 32768.0f;
                for (size_t i = 0; i < per; ++i) {
                    short v = tr[ch][i];
                    if (ch >= 2) v = (short)((unsigned short)v ^ 0x8000u);
                    if (got[i] != (float)v * lsb) { if (!nb) first = i; ++nb; }
                }
                printf("batches %2u ch %d %s: %zu mismatches of %u%s\n", batches, ch, mode ? "buffer loads" : "frame_sample", nb, per,
                       nb ? "" : " OK");
                if (nb) { printf("   first at %zu: got %g\n", first, got[first]); ++bad; }
            }
        hipFree(d); hipFree(o);
    }
    return bad != 0;
}
