// tests/host/sim/sim_device.h -- TEST INFRASTRUCTURE: state shared by the modelled kernels (sim_kernels.cpp) and the check
// program (tests/host/round_plan_check.cpp).
//
// The model tracks IDENTITY, not arithmetic: sample i of stage k of channel c carries the bit pattern
// (c * 16 + k) << 24 | i (24-bit index: streams stay below 2^24 samples).  A modelled kernel touches exactly the addresses the
// real kernel reads and writes (ASan then checks every pointer the planner computed), verifies that what it read IS the
// samples it was meant to read (a wrong offset, a stale tail, an unwritten seam all show as a wrong identity), counts every
// segment transformed and every decimator output produced per (channel, stage), and writes the next stage's identities.
// After a scenario each segment index and each decimator output index of every stream must have been produced EXACTLY ONCE --
// the invariants of src/psd.rs:196-269 (every segment transformed once, every new sample decimated once).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace sim {

struct World {
    std::vector<std::string> errors;
    std::map<int, std::vector<uint8_t>> seg, dec; // times produced, by tag = channel * 16 + stage, then by index
    long fused_launches = 0, fused_jobs = 0, fused_frame_jobs = 0, fused_pairs = 0, seg_jobs = 0, seg_segments = 0, dec_jobs = 0,
         tail_jobs = 0, tail_frame_jobs = 0, red_jobs = 0, max_run = 0, multi_block_jobs = 0, launches_over_cap = 0, aux_launches = 0,
         one_launch_rounds = 0, prologue_copies = 0;
    long fused_enqueued = 0; // launch_fused calls so far (counted when the HOST makes them: what has gone out, run or not)
    int cap_blocks = 0; // > 0: fused_max_blocks() override, so that small streams exercise run lengths > 1 and oversubscription
};
inline World &world()
{
    static World w;
    return w;
}
inline void error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (world().errors.size() < 40)
        world().errors.push_back(buf);
}
inline uint32_t ident(int tag, uint64_t idx) { return ((uint32_t)tag << 24) | (uint32_t)(idx & 0xFFFFFFu); }
inline uint32_t bits_of(const float *p)
{
    uint32_t b;
    memcpy(&b, p, 4);
    return b;
}
inline void put_bits(float *p, uint32_t b) { memcpy(p, &b, 4); }
inline void mark(std::map<int, std::vector<uint8_t>> &m, int tag, uint64_t idx)
{
    std::vector<uint8_t> &v = m[tag];
    if (idx >= v.size())
        v.resize(idx + 1 + v.size() / 2, 0);
    if (v[idx] < 255)
        v[idx] += 1;
}

} // namespace sim
