// frames.h -- device side of the AdcDac frame source (kernels.h FrameSpan): where a sample of a trace sits in a run of
// frames and how its wire word becomes f32 (src/de/frame.rs:5-9 header of 8 bytes, src/de/data.rs:11-82 payload).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace psdk {

// byte offset, within the span, of (batch, channel 0) cell `cell` (cell < 2^24)
__device__ __forceinline__ unsigned frame_cell_offset(const FrameSpan &fs, unsigned cell)
{
    const unsigned f = fs.batches == 1 ? cell : __umulhi(cell, fs.magic); // cell / batches
    // (24-bit multiplies -- full rate, where v_mul_lo_u32 is quarter rate: f < 2^24, batches < 2^8, frame_size < 2^24,
    // products below 2^32)
    const unsigned b = cell - __umul24(f, fs.batches);
    return __umul24(f, fs.frame_size) + 8u + b * 64u;
}

__device__ __forceinline__ float adcdac_lsb() { return 4.096f * 2.5f / 32768.0f; } // src/de/data.rs:28-35 (one constant, asserted equal)

// wire word -> volts: i16::from_le_bytes (ADC) / .wrapping_add(i16::MIN) (DAC: flips the sign bit) as f32 * LSB
__device__ __forceinline__ float adcdac_volts(unsigned word16, bool dac)
{
    if (dac)
        word16 ^= 0x8000u;
    return (float)(short)(unsigned short)word16 * adcdac_lsb();
}

// one sample of trace ch (generic kernels, seams, tails: off the hot path)
__device__ __forceinline__ float frame_sample(const FrameSpan &fs, int ch, unsigned long long i)
{
    const unsigned off = frame_cell_offset(fs, (unsigned)(i >> 3)) + (unsigned)ch * 16u + ((unsigned)i & 7u) * 2u;
    const unsigned short w = *reinterpret_cast<const unsigned short *>(fs.frames + off);
    return adcdac_volts(w, ch >= 2);
}

// four consecutive samples (i a multiple of 4) as one 8-byte load, converted
__device__ __forceinline__ float4 frame_sample4(const FrameSpan &fs, int ch, unsigned long long i)
{
    const unsigned off = frame_cell_offset(fs, (unsigned)(i >> 3)) + (unsigned)ch * 16u + ((unsigned)i & 4u) * 2u;
    const uint2 r = *reinterpret_cast<const uint2 *>(fs.frames + off);
    const bool dac = ch >= 2;
    return make_float4(adcdac_volts(r.x & 0xffffu, dac), adcdac_volts(r.x >> 16, dac), adcdac_volts(r.y & 0xffffu, dac),
                       adcdac_volts(r.y >> 16, dac));
}

} // namespace psdk
