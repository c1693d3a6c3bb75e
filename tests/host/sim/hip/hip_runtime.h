// tests/host/sim/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A host model of the handful of HIP runtime calls the host runtime (stabilizer-stream_amd/csrc/{runtime,planner,frames_ingest,readout}.cpp) makes, so that the library's
// whole host runtime -- the round planner above all (advance_round: span splitting, seam regions, deferral, run sizing, tail
// carries, buffer growth) -- runs on the CPU under AddressSanitizer / UBSan (tests/host/round_plan_check.cpp).  "Device
// memory" is malloc'ed host memory, so every address the planner hands to a kernel is checked by ASan when the modelled
// kernel (sim_kernels.cpp) touches exactly what the real one reads and writes.
//
// Execution model: like the device, work is ENQUEUED and runs later.  One global FIFO holds the operations of all streams in
// the order the host issued them -- one legal schedule of the real semantics (per-stream order kept; an event wait is always
// enqueued after its record) -- and runs when the host synchronises (stream / event / device sync, a blocking copy, hipFree,
// as on the real runtime).  A host that recycles a staging buffer, frees a stream buffer or lets a span go before the device
// has consumed it therefore fails here as it would (sometimes) fail there.  hipStreamQuery reports "not ready" while that
// stream has operations queued: the coalescing of in-place spans on a busy device is exercised as on hardware.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;

struct sim_stream {
    long pending = 0; // operations of this stream still in the queue
};
struct sim_event {
    long seq = -1; // position of its latest record in the queue's numbering (-1: never recorded)
};
typedef sim_stream *hipStream_t;
typedef sim_event *hipEvent_t;

namespace sim {

struct Op {
    long seq;
    sim_stream *s;
    std::function<void()> fn;
};
struct Runtime {
    std::deque<Op> q;
    long next_seq = 0, done_seq = -1;
    sim_stream null_stream;
    size_t live_bytes = 0, live_blocks = 0, total_ops = 0;
    bool fail_next_malloc = false; // fault injection
};
inline Runtime &rt()
{
    static Runtime r;
    return r;
}
inline sim_stream *S(hipStream_t s) { return s ? s : &rt().null_stream; }
inline void enqueue(hipStream_t s, std::function<void()> fn)
{
    Runtime &r = rt();
    sim_stream *st = S(s);
    st->pending += 1;
    r.q.push_back({r.next_seq++, st, std::move(fn)});
    r.total_ops += 1;
}
inline void run_until(long seq) // run every queued operation numbered <= seq
{
    Runtime &r = rt();
    while (!r.q.empty() && r.q.front().seq <= seq) {
        Op op = std::move(r.q.front());
        r.q.pop_front();
        op.fn();
        op.s->pending -= 1;
        r.done_seq = op.seq;
    }
}
inline void drain() { run_until(rt().next_seq); }

} // namespace sim

inline const char *hipGetErrorString(hipError_t e)
{
    return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorNotReady ? "not ready" : "invalid value";
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize()
{
    sim::drain();
    return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi)
{
    *lo = 0;
    *hi = -1;
    return hipSuccess;
}

template <class T>
inline hipError_t hipMalloc(T **p, size_t bytes)
{
    if (sim::rt().fail_next_malloc) {
        sim::rt().fail_next_malloc = false;
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    // (exactly `bytes`: ASan's red zone starts at the first byte past what the library asked for)
    *p = static_cast<T *>(malloc(bytes ? bytes : 1));
    if (!*p)
        return hipErrorOutOfMemory;
    memset((void *)*p, 0xA5, bytes); // device memory is not zeroed: a read of something never written shows
    sim::rt().live_blocks += 1;
    return hipSuccess;
}
inline hipError_t hipFree(void *p)
{
    sim::drain(); // hipFree waits for outstanding work, like the real one
    if (p)
        sim::rt().live_blocks -= 1;
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned)
{
    *p = malloc(bytes ? bytes : 1);
    if (*p)
        sim::rt().live_blocks += 1;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void *p)
{
    sim::drain();
    if (p)
        sim::rt().live_blocks -= 1;
    free(p);
    return hipSuccess;
}

inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s)
{
    sim::enqueue(s, [=] { memmove(dst, src, bytes); });
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind)
{
    sim::drain();
    memmove(dst, src, bytes);
    return hipSuccess;
}
// (a null-stream copy does NOT wait for the handle's non-blocking streams: nothing is drained here -- what it reads is the caller's)
inline hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind)
{
    for (size_t r = 0; r < height; ++r)
        memmove(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t s)
{
    sim::enqueue(s, [=] { memset(dst, v, bytes); });
    return hipSuccess;
}
inline hipError_t hipMemset(void *dst, int v, size_t bytes)
{
    sim::drain();
    memset(dst, v, bytes);
    return hipSuccess;
}

inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = new sim_stream();
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
inline hipError_t hipStreamDestroy(hipStream_t s)
{
    sim::drain();
    delete s;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s)
{
    // everything this stream has queued, and (FIFO) whatever other streams queued before it
    sim::Runtime &r = sim::rt();
    long last = -1;
    for (const sim::Op &op : r.q)
        if (op.s == sim::S(s))
            last = op.seq;
    sim::run_until(last);
    return hipSuccess;
}
inline hipError_t hipStreamQuery(hipStream_t s) { return sim::S(s)->pending ? hipErrorNotReady : hipSuccess; }

inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    *e = new sim_event();
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e)
{
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    e->seq = sim::rt().next_seq; // the marker operation below gets this number
    sim::enqueue(s, [] {});
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    if (e->seq >= 0)
        sim::run_until(e->seq);
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    // FIFO execution: the record (already queued, or already run) precedes everything queued from here on
    (void)e;
    sim::enqueue(s, [] {});
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.0f;
    return hipSuccess;
}
