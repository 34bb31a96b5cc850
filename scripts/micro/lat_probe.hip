// Dependent-chain latencies of the cross-lane primitives the wave-wide heap is made of (one wave, cycles per link).
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __device__ int llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
template <int T> __global__ void k(int* out, long long* ticks, int n) {
    const int lane = threadIdx.x;
    int v = out[lane];
    int s = 0;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (T == 0) { v = v * 3 + 1; }                                                      // dependent VALU
        if (T == 1) { v = __shfl(v, (lane * 2 + 1) & 63) + 1; }                             // ds_bpermute chain
        if (T == 2) { int u = __builtin_amdgcn_readlane(v, 5); v = v + u; }                 // readlane -> VALU
        if (T == 3) { int u = __builtin_amdgcn_readlane(v, 5); u = (u >> 3) & 63; u = __builtin_amdgcn_readlane(v, u); v = v + u; }  // readlane -> SALU -> readlane (lane select) -> VALU
        if (T == 4) { unsigned long long b = __ballot((v & 1) != 0); v = v + (int)__popcll(b); }   // cmp -> SGPR -> SALU -> VALU
        if (T == 5) { unsigned long long b = __ballot((v & 1) != 0); v = ((b >> lane) & 1) ? v + 3 : v + 1; }  // cmp -> SGPR -> VALU
        if (T == 6) { int u = __builtin_amdgcn_readlane(v, 5); if (u & 1) v += 1; else v += 3; }  // readlane -> scalar branch
        if (T == 7) { int u = __builtin_amdgcn_readlane(v, 5); v = llvm_writelane(u + 1, (u >> 2) & 63, v); }  // readlane -> writelane
        if (T == 8) { v = __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, false) + 1; }          // DPP row_shr:1 chain
        if (T == 9) { v = __builtin_amdgcn_ds_swizzle(v, 0x801f) + 1; }                        // ds_swizzle chain
        if (T == 10) { v = v * 3 + 1; s = s * 5 + 2; }                                          // VALU + independent VALU
      }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[lane] = v + s;
    if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}
template <int T> void run(const char* name, int* d, long long* t) {
    const int n = 100000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, d, t, n); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-52s %7.1f cycles  %6.1f ns per link\n", name, (double)h[0] / n, h[1] * 10.0 / n);
}
int main() {
    int* d; long long* t;
    hipMalloc(&d, 4096); hipMalloc(&t, 64); hipMemset(d, 0, 4096);
    run<0>("VALU -> VALU", d, t);
    run<1>("ds_bpermute -> VALU", d, t);
    run<2>("readlane -> VALU", d, t);
    run<3>("readlane -> SALU -> readlane(select) -> VALU", d, t);
    run<4>("ballot -> SALU popcount -> VALU", d, t);
    run<5>("ballot -> VALU (mask test)", d, t);
    run<6>("readlane -> scalar branch -> VALU", d, t);
    run<7>("readlane -> writelane", d, t);
    run<8>("DPP row_shr -> VALU", d, t);
    run<9>("ds_swizzle -> VALU", d, t);
    run<10>("two independent VALU chains", d, t);
    return 0;
}
