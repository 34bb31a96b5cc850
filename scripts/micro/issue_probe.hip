// Issue rate of one wave on an otherwise idle CU: dependent / independent VALU chains, v_cmp + v_cndmask pairs, SALU chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T> __global__ void k(unsigned* out, long long* ticks, int n) {
    const unsigned lane = threadIdx.x;
    unsigned a = out[lane], b = out[lane + 64], c = out[lane + 128], d = out[lane + 192];
    long long t0 = clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (T == 0) { a = (a ^ lane) + (a >> 3); }                                   // 2-3 dependent VALU
            if (T == 1) { a = (a ^ lane) + (a >> 3); b = (b ^ lane) + (b >> 3); c = (c ^ lane) + (c >> 3); d = (d ^ lane) + (d >> 3); }   // 4 independent chains
            if (T == 2) { a = a < b ? a + 1 : a + c; }                                   // v_cmp -> v_cndmask (VCC)
            if (T == 3) { bool p = a < b, q = (a ^ c) < d; a = (p && q) ? a + 1 : a + 3; }   // two masks combined (s_and_b64)
            if (T == 4) { unsigned p = a < b ? a + 1 : a + 3; a = (a ^ c) < d ? p : a + 3; }  // the same with two selects
            if (T == 5) { a = min(a, b) + 1; b = max(b, c) ^ a; }                         // min/max chain
        }
    }
    long long t1 = clock64();
    out[lane] = a + b + c + d;
    if (lane == 0) ticks[0] = t1 - t0;
}
template <int T> void run(const char* name, unsigned* d, long long* t, double instr) {
    const int n = 160000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, d, t, n); hipDeviceSynchronize(); }
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-48s %6.1f cycles per step (about %.0f instructions)\n", name, (double)h / n, instr);
}
int main() {
    unsigned* d; long long* t;
    hipMalloc(&d, 4096); hipMalloc(&t, 64); hipMemset(d, 1, 4096);
    run<0>("dependent xor / shift / add", d, t, 3);
    run<1>("four independent xor / shift / add chains", d, t, 12);
    run<2>("cmp -> cndmask -> add (VCC)", d, t, 4);
    run<3>("two compares, masks ANDed, select", d, t, 6);
    run<4>("two compares, two selects", d, t, 7);
    run<5>("min / max chain", d, t, 4);
    return 0;
}
