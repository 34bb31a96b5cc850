// hipsim.h — TEST-ONLY single-process SIMT emulator used to debug / sanitise the HIP kernels
// of deft4j_amd/csrc on a CPU (the build container has no GPU).  It is never part of
// libdeft4g.so: the product path has no CPU fallback.  Compiled with -DD4G_HOSTSIM, this
// header stands in for <hip/hip_runtime.h> and d4g_rt.h's HIP half.
//
// Model: every workgroup runs as `blockDim.x` ucontext fibers on one OS thread.
//   __syncthreads()           all live fibers of the workgroup rendezvous
//   __ballot/__shfl*          all live lanes of a 64-lane wave rendezvous (convergent use only)
//   atomics                   plain read-modify-write (single OS thread)
//   __shared__                static storage (workgroups run one after another)
// It checks convergence (a wave collective met by a workgroup barrier aborts) but cannot see
// data races; its purpose is functional parity debugging with ASan/UBSan.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)

struct SimDim3 { unsigned x, y, z; };
extern SimDim3 threadIdx, blockIdx, blockDim, gridDim;
struct uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }

namespace hipsim {
enum { READY = 0, AT_BARRIER = 1, AT_COLLECTIVE = 2, DONE = 3 };
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = READY;
    int tid = 0;
    int parity = 0;
};
struct WaveBuf {
    long long val[2][64];
    bool arrived[2][64];
};
struct Sim {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<WaveBuf> waves;
    Fiber* cur = nullptr;
    std::function<void()> body;
    long long launches = 0;
    const char* kname = "";
};
Sim& sim();
void yield_to_scheduler();
void run_grid(unsigned grid, unsigned block, const std::function<void()>& body);
long long collective(long long v);   // deposits v, waits for the wave, returns this lane's parity buffer index
}  // namespace hipsim

inline void __syncthreads() {
    hipsim::sim().cur->state = hipsim::AT_BARRIER;
    hipsim::yield_to_scheduler();
}
inline unsigned long long __ballot(int pred) {
    int p = (int)hipsim::collective(pred ? 1 : 0);
    auto& w = hipsim::sim().waves[threadIdx.x >> 6];
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++)
        if (w.arrived[p][l] && w.val[p][l]) m |= 1ULL << l;
    return m;
}
template <typename T>
inline T __shfl(T v, int src) {
    long long raw = 0;
    static_assert(sizeof(T) <= 8, "shfl payload");
    memcpy(&raw, &v, sizeof(T));
    int p = (int)hipsim::collective(raw);
    auto& w = hipsim::sim().waves[threadIdx.x >> 6];
    src &= 63;
    if (!w.arrived[p][src]) return v;
    T out;
    memcpy(&out, &w.val[p][src], sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_xor(T v, int mask) { return __shfl(v, (int)((threadIdx.x & 63) ^ mask)); }
template <typename T>
inline T __shfl_up(T v, int delta) {
    int lane = threadIdx.x & 63;
    T o = __shfl(v, lane >= delta ? lane - delta : lane);
    return o;
}
inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long expect, unsigned long long v) {
    unsigned long long o = *p;
    if (o == expect) *p = v;
    return o;
}
inline unsigned atomicCAS(unsigned* p, unsigned expect, unsigned v) {
    unsigned o = *p;
    if (o == expect) *p = v;
    return o;
}
inline void __threadfence() {}
inline long long clock64() { return 0; }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }

template <typename T>
inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T>
inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T>
inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
template <typename T>
inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T>
inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }

// ---- runtime half (mirrors d4g_rt.h) ----
#define RT_MAX_LANES 8
struct RtGlobals {
    void* stream = nullptr;
    int cur = 0;
    int device = 0;
    bool ready = false;
};
#define RT_MAX_CTX 16
inline int& rt_ctx() { static int c = 0; return c; }   // (the emulator has one device: every context is the same)
inline RtGlobals& rt() {
    static RtGlobals g;
    return g;
}
inline void* rt_malloc(size_t n) { return calloc(1, n ? n : 16); }
inline void rt_free(void* p) { free(p); }
inline void rt_h2d(void* d, const void* h, size_t n) { if (n) memcpy(d, h, n); }
inline void rt_d2h(void* h, const void* d, size_t n) { if (n) memcpy(h, d, n); }
inline void rt_d2d(void* d, const void* s, size_t n) { if (n) memmove(d, s, n); }
inline void rt_memset(void* d, int v, size_t n) { if (n) memset(d, v, n); }
inline bool& rt_low_priority_thread() { static bool v = false; return v; }
inline void rt_sync() {}
inline void rt_sync_all() {}
struct RtEvent { void record() {} void record2() {} };  // timing is meaningless in the emulator
inline void rt_stream2_wait(RtEvent&) {}
inline void rt_stream_wait(RtEvent&) {}
inline float rt_elapsed_ms(RtEvent&, RtEvent&) { return 0.f; }
inline int device_cus() { return 2; }
#define RT_CHECK(x) (x)
#define RT_LAUNCH2(kern, grid, block, ...) RT_LAUNCH(kern, grid, block, __VA_ARGS__)
#define RT_LAUNCH(kern, grid, block, ...) (hipsim::sim().kname = #kern, hipsim::run_grid((unsigned)(grid), (unsigned)(block), [&]() { kern(__VA_ARGS__); }))
