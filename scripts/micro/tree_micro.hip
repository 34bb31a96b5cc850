// How long does one wave take to rebuild a literal/length tree of ~56 used symbols?  (the search's critical chain)
#include "../../deft4j_amd/csrc/d4g_device.h"
#include <cstdio>
#include <vector>
typedef TreeMem<uint64_t, uint16_t, D4G_NLIT, 16, true> LitTree;
#ifndef VARIANT
#define VARIANT 0
#endif
__global__ void k_tree(const uint32_t* histG, uint8_t* lensG, long long* ticks, int reps) {
    __shared__ alignas(16) unsigned char mem[(LitTree::bytes(1) + 15) & ~15];
    __shared__ uint32_t hist[D4G_NLIT];
    __shared__ uint8_t lens[D4G_NLIT];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < D4G_NLIT; i += 64) { hist[i] = histG[i]; lens[i] = 0; }
    __syncthreads();
    LitTree tm; tm.carve(mem, 1);
    long long t0 = clock64();
    int err = 0;
    for (int r = 0; r < reps; r++)
        err |= d4g_build_tree_wave64<(D4G_NLIT + 63) / 64>(tm, 286, 15, [&](int i) { return hist[i]; }, [&](int v, int len) { lens[v] = (uint8_t)len; });
    long long t1 = clock64();
    for (int i = lane; i < D4G_NLIT; i += 64) lensG[i] = lens[i];
    if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = err; }
}
int main() {
    std::vector<uint32_t> h(D4G_NLIT, 0);
    // 55 used literals with a text-like distribution + 256 + a few length symbols
    unsigned s = 12345;
    int used = 0;
    for (int i = 32; i < 127 && used < 48; i += 2) { s = s * 1664525u + 1013904223u; h[i] = 1 + (s >> 20) % 3000; used++; }
    h[256] = 1;
    for (int i = 257; i < 265; i++) { s = s * 1664525u + 1013904223u; h[i] = 1 + (s >> 20) % 500; }
    uint32_t* dh; uint8_t* dl; long long* dt;
    hipMalloc(&dh, 4 * D4G_NLIT); hipMalloc(&dl, D4G_NLIT); hipMalloc(&dt, 64);
    hipMemcpy(dh, h.data(), 4 * D4G_NLIT, hipMemcpyHostToDevice);
    for (int big = 0; big < 2; big++) {
    if (big) {   // a PNG-like block: every literal used, a few length symbols
        unsigned t = 99;
        for (int i = 0; i < 256; i++) { t = t * 1664525u + 1013904223u; h[i] = 20 + (t >> 20) % 400; }
        h[256] = 1;
        for (int i = 257; i < 280; i++) { t = t * 1664525u + 1013904223u; h[i] = 1 + (t >> 20) % 200; }
        hipMemcpy(dh, h.data(), 4 * D4G_NLIT, hipMemcpyHostToDevice);
    }
    for (int rep = 0; rep < 2; rep++) {
        const int reps = 200;
        hipLaunchKernelGGL(k_tree, dim3(1), dim3(64), 0, 0, dh, dl, dt, reps);
        hipDeviceSynchronize();
        long long t[2]; hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        std::vector<uint8_t> l(D4G_NLIT); hipMemcpy(l.data(), dl, D4G_NLIT, hipMemcpyDeviceToHost);
        unsigned ck = 0; for (int i = 0; i < D4G_NLIT; i++) ck = ck * 31 + l[i];
        int used = 0; for (int i = 0; i < D4G_NLIT; i++) used += h[i] != 0;
#ifdef D4G_PROFILE_OPS
        unsigned long long sec[4]; hipMemcpyFromSymbol(sec, HIP_SYMBOL(d4g_dbg_tree), sizeof(sec));
        unsigned long long zero[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(d4g_dbg_tree), zero, sizeof(zero));
        if (used > 64) printf("  sections (cycles per tree): leaves %.0f merges %.0f depths %.0f\n", (double)sec[0] / reps, (double)sec[1] / reps, (double)sec[2] / reps);
#endif
        printf("variant %d: %.0f cycles per tree (%d leaves), err %lld, lens checksum %08x\n", VARIANT, (double)t[0] / reps, used, t[1], ck);
    }
    }
    return 0;
}
