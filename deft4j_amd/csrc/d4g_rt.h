// d4g_rt.h — thin runtime layer used by the host orchestration (HIP stream, device memory,
// kernel launch).  The test-only host simulator (tests/hostsim) provides the same names so the
// kernels can be debugged and sanitised on a CPU; the product library always uses HIP.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <map>
#include <string>
#include <unordered_map>

#ifndef D4G_HOSTSIM
#include <hip/hip_runtime.h>

#define RT_CHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

#define RT_MAX_LANES 8
#include <mutex>
#include <vector>
// Process-wide state: the selected device and the lock of everything host threads share (the memory pool, the list of
// per-thread stream sets).  Batches are driven by whichever host thread calls in — CompressionUtil's pool threads
// (C/CompressionUtil.java:111-117), or two bench threads keeping two batches in flight — and each thread gets its own
// HIP streams, so independent batches overlap on the device instead of queueing behind one library-wide lock.
inline bool& rt_low_priority_thread() { thread_local bool v = false; return v; }   // set before the thread's first runtime call
struct RtGlobals;
// Device memory comes from a small caching pool: hipMalloc / hipFree cost milliseconds for the buffers a batch
// needs (and vary a lot from host to host), so freed blocks are kept by size class and handed out again.
// Callers free a block only after the work that used it has completed (they synchronise first), as hipFree's
// implicit device synchronisation used to guarantee.
struct RtPool {
    std::multimap<size_t, void*> freeBlocks;      // capacity -> block
    std::unordered_map<void*, size_t> capacity;   // every block the pool handed out or holds
    size_t heldBytes = 0;                         // bytes sitting in freeBlocks
    size_t maxHeldBytes = (size_t)64 << 30;       // D4G_POOL_MAX_MB overrides
    static size_t size_class(size_t n) {          // 1/8-octave steps: at most 12.5 % slack
        if (n < 4096) return 4096;
        int hb = 63 - __builtin_clzll((unsigned long long)n);
        size_t step = (size_t)1 << (hb - 3);
        return (n + step - 1) & ~(step - 1);
    }
};
// One process may drive several GPUs (a JVM is one process: CompressionUtil's pool, C/CompressionUtil.java:99-117, fans
// streams over the devices of a node): a *context* = one device with its own memory pool, its own copy of the search
// programs and, per host thread, its own streams.  The same device may back two contexts (tests on a one-GPU box).  The
// context a host thread works on is thread-local: the C ABI sets it from the batch / the caller's choice at every entry.
#define RT_MAX_CTX 16
struct RtContext {
    int device = -1;
    bool ready = false;
    RtPool pool;
};
struct RtProcess {
    RtContext ctx[RT_MAX_CTX];
    std::mutex mu;
    std::vector<RtGlobals*> threads;
};
inline int& rt_ctx() { thread_local int c = 0; return c; }
inline RtProcess& rtp() {
    static RtProcess p;
    return p;
}
struct RtGlobals {
    // "lane" k = a pair of HIP streams: state ops on a[k], header searches on b[k].  Groups of deflate
    // blocks run their level sequences on different lanes so that one group's launch tails overlap
    // another group's work.  Lane 0 is the default stream of every other phase.
    hipStream_t a[RT_MAX_LANES] = {nullptr};
    hipStream_t b[RT_MAX_LANES] = {nullptr};
    int cur = 0;
    bool made = false;
    int ctxIdx;
    int& device;
    bool& ready;
    explicit RtGlobals(int c) : ctxIdx(c), device(rtp().ctx[c].device), ready(rtp().ctx[c].ready) {
        std::lock_guard<std::mutex> lk(rtp().mu);
        rtp().threads.push_back(this);
    }
    ~RtGlobals() {
        destroy_streams();
        std::lock_guard<std::mutex> lk(rtp().mu);
        auto& v = rtp().threads;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == this) { v.erase(v.begin() + i); break; }
    }
    void ensure() {   // the calling thread's streams, created on first use after d4g_init
        if (made || !ready) return;
        RT_CHECK(hipSetDevice(device));
        // A thread that is about to run kernels lasting minutes (the Zopfli squeeze) asks for low-priority streams: HIP multiplexes
        // its streams onto a few hardware queues per priority level, and a queue stays busy until its kernel ends — normal-priority
        // streams of other threads that landed on the same queue would wait behind it (measured: a candidate search held up for 78 s).
        // Only lane 0 now, the other lanes when a level-executor round first uses them: HIP hands its hardware queues to streams
        // in creation order, and a thread that created sixteen streams up front pushed the next thread's main stream back onto
        // the first thread's queue — where that thread's candidate search (one launch of milliseconds) held up the other batch's
        // parse (measured: two batches in flight 9.7 ms per step when they collided, 7.0 when they did not).
        made = true;
        make_lane(0);
    }
    void make_lane(int k) {
        if (a[k]) return;
        int least = 0, greatest = 0;
        if (rt_low_priority_thread()) {
            RT_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            RT_CHECK(hipStreamCreateWithPriority(&a[k], hipStreamNonBlocking, least));
            RT_CHECK(hipStreamCreateWithPriority(&b[k], hipStreamNonBlocking, least));
        } else {
            RT_CHECK(hipStreamCreateWithFlags(&a[k], hipStreamNonBlocking));
            RT_CHECK(hipStreamCreateWithFlags(&b[k], hipStreamNonBlocking));
        }
    }
    hipStream_t sa() { make_lane(cur); return a[cur]; }
    hipStream_t sb() { make_lane(cur); return b[cur]; }
    void destroy_streams() {
        if (!made) return;
        for (int k = 0; k < RT_MAX_LANES; k++) {
            if (a[k]) { (void)hipStreamDestroy(a[k]); a[k] = nullptr; }
            if (b[k]) { (void)hipStreamDestroy(b[k]); b[k] = nullptr; }
        }
        made = false;
    }
    hipStream_t& stream_ref() { make_lane(cur); return a[cur]; }
};
inline RtGlobals& rt() {
    struct PerThread {
        RtGlobals* g[RT_MAX_CTX] = {nullptr};
        ~PerThread() { for (RtGlobals* p : g) delete p; }
    };
    thread_local PerThread t;
    const int c = rt_ctx();
    if (!t.g[c]) t.g[c] = new RtGlobals(c);
    t.g[c]->ensure();
    return *t.g[c];
}
inline RtPool& rt_pool() { return rtp().ctx[rt_ctx()].pool; }
inline void* rt_malloc(size_t n) {
    std::lock_guard<std::mutex> lk(rtp().mu);
    RtPool& P = rt_pool();
    size_t c = RtPool::size_class(n ? n : 16);
    auto it = P.freeBlocks.find(c);
    if (it != P.freeBlocks.end()) {
        void* p = it->second;
        P.freeBlocks.erase(it);
        P.heldBytes -= c;
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, c);
    if (e != hipSuccess && !P.freeBlocks.empty()) {   // out of memory: give the cached blocks back and retry
        (void)hipGetLastError();
        for (auto& kv : P.freeBlocks) { P.capacity.erase(kv.second); (void)hipFree(kv.second); }
        P.freeBlocks.clear();
        P.heldBytes = 0;
        e = hipMalloc(&p, c);
    }
    if (e != hipSuccess) throw std::runtime_error(std::string("hipMalloc: ") + hipGetErrorString(e));
    P.capacity[p] = c;
    return p;
}
inline void rt_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(rtp().mu);
    RtPool& P = rt_pool();
    auto it = P.capacity.find(p);
    if (it == P.capacity.end()) { (void)hipFree(p); return; }
    size_t c = it->second;
    if (P.heldBytes + c > P.maxHeldBytes) { P.capacity.erase(it); (void)hipFree(p); return; }
    P.freeBlocks.emplace(c, p);
    P.heldBytes += c;
}
inline void rt_pool_release() {   // d4g_shutdown
    std::lock_guard<std::mutex> lk(rtp().mu);
    RtPool& P = rt_pool();
    for (auto& kv : P.freeBlocks) { P.capacity.erase(kv.second); (void)hipFree(kv.second); }
    P.freeBlocks.clear();
    P.heldBytes = 0;
}
inline void rt_h2d(void* d, const void* h, size_t n) { if (n) RT_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, rt().sa())); }
inline void rt_d2h(void* h, const void* d, size_t n) {
    if (n) RT_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, rt().sa()));
    RT_CHECK(hipStreamSynchronize(rt().sa()));
}
inline void rt_d2d(void* d, const void* s, size_t n) { if (n) RT_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, rt().sa())); }
inline void rt_memset(void* d, int v, size_t n) { if (n) RT_CHECK(hipMemsetAsync(d, v, n, rt().sa())); }
inline void rt_sync() { RT_CHECK(hipStreamSynchronize(rt().sa())); }
inline void rt_sync_all() {
    for (int k = 0; k < RT_MAX_LANES; k++) {
        if (rt().a[k]) RT_CHECK(hipStreamSynchronize(rt().a[k]));
        if (rt().b[k]) RT_CHECK(hipStreamSynchronize(rt().b[k]));
    }
}
#define RT_LAUNCH(kern, grid, block, ...)                                                           \
    do {                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, rt().sa(), __VA_ARGS__); \
        RT_CHECK(hipGetLastError());                                                                \
    } while (0)

struct RtEvent {
    hipEvent_t e = nullptr;
    RtEvent() { RT_CHECK(hipEventCreate(&e)); }
    ~RtEvent() { if (e) (void)hipEventDestroy(e); }
    void record() { RT_CHECK(hipEventRecord(e, rt().sa())); }
    void record2() { RT_CHECK(hipEventRecord(e, rt().sb())); }
};
inline void rt_stream2_wait(RtEvent& ev) { RT_CHECK(hipStreamWaitEvent(rt().sb(), ev.e, 0)); }
inline void rt_stream_wait(RtEvent& ev) { RT_CHECK(hipStreamWaitEvent(rt().sa(), ev.e, 0)); }
#define RT_LAUNCH2(kern, grid, block, ...)                                                          \
    do {                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, rt().sb(), __VA_ARGS__); \
        RT_CHECK(hipGetLastError());                                                                \
    } while (0)
inline int device_cus() {
    hipDeviceProp_t prop;
    RT_CHECK(hipGetDeviceProperties(&prop, rt().device < 0 ? 0 : rt().device));
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}
inline float rt_elapsed_ms(RtEvent& a, RtEvent& b) {
    RT_CHECK(hipEventSynchronize(b.e));
    float ms = 0;
    RT_CHECK(hipEventElapsedTime(&ms, a.e, b.e));
    return ms;
}
#endif
