// d4g_rt.h — thin runtime layer used by the host orchestration (HIP stream, device memory,
// kernel launch).  The test-only host simulator (tests/hostsim) provides the same names so the
// kernels can be debugged and sanitised on a CPU; the product library always uses HIP.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#ifndef D4G_HOSTSIM
#include <hip/hip_runtime.h>

#define RT_CHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

#define RT_MAX_LANES 8
struct RtGlobals {
    // "lane" k = a pair of HIP streams: state ops on a[k], header searches on b[k].  Groups of deflate
    // blocks run their level sequences on different lanes so that one group's launch tails overlap
    // another group's work.  Lane 0 is the default stream of every other phase.
    hipStream_t a[RT_MAX_LANES] = {nullptr};
    hipStream_t b[RT_MAX_LANES] = {nullptr};
    int cur = 0;
    int device = -1;
    bool ready = false;
    hipStream_t& stream_ref() { return a[cur]; }
};
inline RtGlobals& rt() {
    static RtGlobals g;
    return g;
}
inline void* rt_malloc(size_t n) {
    void* p = nullptr;
    RT_CHECK(hipMalloc(&p, n ? n : 16));
    return p;
}
inline void rt_free(void* p) { if (p) (void)hipFree(p); }
inline void rt_h2d(void* d, const void* h, size_t n) { if (n) RT_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, rt().a[rt().cur])); }
inline void rt_d2h(void* h, const void* d, size_t n) {
    if (n) RT_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, rt().a[rt().cur]));
    RT_CHECK(hipStreamSynchronize(rt().a[rt().cur]));
}
inline void rt_d2d(void* d, const void* s, size_t n) { if (n) RT_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, rt().a[rt().cur])); }
inline void rt_memset(void* d, int v, size_t n) { if (n) RT_CHECK(hipMemsetAsync(d, v, n, rt().a[rt().cur])); }
inline void rt_sync() { RT_CHECK(hipStreamSynchronize(rt().a[rt().cur])); }
#define RT_LAUNCH(kern, grid, block, ...)                                                           \
    do {                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, rt().a[rt().cur], __VA_ARGS__); \
        RT_CHECK(hipGetLastError());                                                                \
    } while (0)

struct RtEvent {
    hipEvent_t e = nullptr;
    RtEvent() { RT_CHECK(hipEventCreate(&e)); }
    ~RtEvent() { if (e) (void)hipEventDestroy(e); }
    void record() { RT_CHECK(hipEventRecord(e, rt().a[rt().cur])); }
    void record2() { RT_CHECK(hipEventRecord(e, rt().b[rt().cur])); }
};
inline void rt_stream2_wait(RtEvent& ev) { RT_CHECK(hipStreamWaitEvent(rt().b[rt().cur], ev.e, 0)); }
inline void rt_stream_wait(RtEvent& ev) { RT_CHECK(hipStreamWaitEvent(rt().a[rt().cur], ev.e, 0)); }
#define RT_LAUNCH2(kern, grid, block, ...)                                                          \
    do {                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, rt().b[rt().cur], __VA_ARGS__); \
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
