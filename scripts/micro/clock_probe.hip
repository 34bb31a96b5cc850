// How fast does one wave run a dependent chain?  (Is the shader clock up when the device is nearly idle?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, long long* ticks, int n) {
    float x = out[threadIdx.x];
    long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = x * 1.0001f + 0.5f;
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* d; long long* t;
    hipMalloc(&d, 4096); hipMalloc(&t, 8 * 4096); hipMemset(d, 0, 4096);
    for (int blocks : {1, 1, 256, 4096}) {
        for (int rep = 0; rep < 2; rep++) {
            int n = 200000;
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, t, n);
            hipDeviceSynchronize();
            long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            double instr = (double)n * 64;
            printf("blocks %d: %lld ticks (100 MHz) for %.0f dependent FMAs -> %.2f ns per FMA\n", blocks, h, instr, h * 10.0 / instr);
        }
    }
    return 0;
}
