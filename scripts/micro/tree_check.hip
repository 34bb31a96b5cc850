// Checks the wave-wide tree builders against each other on many histograms: d4g_build_tree_wave64 (one-register queue up to 64
// leaves, path-per-lane queue for 65..127, general queue beyond) against the general queue alone; then times the 65..127 path.
#include "../../deft4j_amd/csrc/d4g_device.h"
#include <cstdio>
#include <vector>
#include <cstring>
typedef TreeMem<uint64_t, uint16_t, D4G_NLIT, 16, true> LitTree;
__global__ void k_both(const uint32_t* histG, uint8_t* lensA, uint8_t* lensB, int* errs, int nSym, int limit, long long* ticks, int reps) {
    __shared__ alignas(16) unsigned char mem[(LitTree::bytes(1) + 15) & ~15];
    __shared__ uint32_t hist[D4G_NLIT];
    __shared__ uint8_t lens[D4G_NLIT];
    const int lane = threadIdx.x & 63;
    const uint32_t* hg = histG + (size_t)blockIdx.x * D4G_NLIT;
    for (int i = lane; i < D4G_NLIT; i += 64) { hist[i] = hg[i]; lens[i] = 0; }
    __syncthreads();
    LitTree tm; tm.carve(mem, 1);
    int e0 = d4g_build_tree_wave<(D4G_NLIT + 63) / 64>(tm, nSym, limit, [&](int i) { return hist[i]; }, [&](int v, int len) { lens[v] = (uint8_t)len; });
    __syncthreads();
    for (int i = lane; i < D4G_NLIT; i += 64) { lensA[(size_t)blockIdx.x * D4G_NLIT + i] = lens[i]; lens[i] = 0; }
    __syncthreads();
    long long t0 = clock64();
    int e1 = 0;
    for (int r = 0; r < reps; r++) {
        for (int i = lane; i < D4G_NLIT; i += 64) lens[i] = 0;
        __syncthreads();
        e1 = d4g_build_tree_wave64<(D4G_NLIT + 63) / 64>(tm, nSym, limit, [&](int i) { return hist[i]; }, [&](int v, int len) { lens[v] = (uint8_t)len; });
    }
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    if (reps > 1)
        for (int r = 0; r < reps; r++) {
            for (int i = lane; i < D4G_NLIT; i += 64) lens[i] = 0;
            __syncthreads();
            e1 |= d4g_build_tree_wave<(D4G_NLIT + 63) / 64>(tm, nSym, limit, [&](int i) { return hist[i]; }, [&](int v, int len) { lens[v] = (uint8_t)len; });
        }
    long long t3 = clock64();
    if (lane == 0 && blockIdx.x == 0) ticks[1] = t3 - t2;
    __syncthreads();
    for (int i = lane; i < D4G_NLIT; i += 64) lensB[(size_t)blockIdx.x * D4G_NLIT + i] = lens[i];
    if (lane == 0) { errs[2 * blockIdx.x] = e0; errs[2 * blockIdx.x + 1] = e1; if (blockIdx.x == 0) ticks[0] = t1 - t0; }
}
int main() {
    const int N = 1024;
    std::vector<uint32_t> h((size_t)N * D4G_NLIT, 0);
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int t = 0; t < N; t++) {
        uint32_t* g = &h[(size_t)t * D4G_NLIT];
        int used = t == 0 ? 100 : t < 8 ? 60 + t : (int)(rnd() % 141);   // 0..140 used symbols: all three queues, and their borders
        int mode = t % 4;                                                // 0 text-like, 1 many ties, 2 all equal, 3 geometric (deep: limiter)
        for (int u = 0; u < used; u++) {
            int sym;
            do sym = (int)(rnd() % 286); while (g[sym]);
            g[sym] = mode == 0 ? 1 + rnd() % 3000 : mode == 1 ? 1 + rnd() % 3 : mode == 2 ? 5 : (u < 22 ? 1u << u : 1 + rnd() % 7);
        }
    }
    uint32_t* dh; uint8_t *dA, *dB; int* de; long long* dt;
    hipMalloc(&dh, h.size() * 4); hipMalloc(&dA, (size_t)N * D4G_NLIT); hipMalloc(&dB, (size_t)N * D4G_NLIT); hipMalloc(&de, N * 8); hipMalloc(&dt, 64);
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {
        const int reps = 100;
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_both, dim3(1), dim3(64), 0, 0, dh, dA, dB, de, 286, 15, dt, reps); hipDeviceSynchronize(); }
        long long t[2]; hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
        printf("100 leaves: %.0f cycles per tree with the path-per-lane queue, %.0f with the general queue (incl. clearing 288 lengths)\n", (double)t[0] / reps, (double)t[1] / reps);
    }
    hipLaunchKernelGGL(k_both, dim3(N), dim3(64), 0, 0, dh, dA, dB, de, 286, 15, dt, 1);
    hipDeviceSynchronize();
    std::vector<uint8_t> A((size_t)N * D4G_NLIT), B((size_t)N * D4G_NLIT); std::vector<int> e(2 * N);
    hipMemcpy(A.data(), dA, A.size(), hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, B.size(), hipMemcpyDeviceToHost); hipMemcpy(e.data(), de, N * 8, hipMemcpyDeviceToHost);
    int bad = 0, lim = 0, mid = 0;
    for (int t = 0; t < N; t++) {
        int used = 0; for (int i = 0; i < 286; i++) used += h[(size_t)t * D4G_NLIT + i] != 0;
        mid += used > 64 && used <= 127;
        bool diff = memcmp(&A[(size_t)t * D4G_NLIT], &B[(size_t)t * D4G_NLIT], D4G_NLIT) != 0 || e[2 * t] != e[2 * t + 1];
        if (diff && bad < 8) printf("  MISMATCH histogram %d (used %d, mode %d) err %d/%d\n", t, used, t % 4, e[2 * t], e[2 * t + 1]);
        bad += diff;
        int mx = 0; for (int i = 0; i < 286; i++) mx = A[(size_t)t * D4G_NLIT + i] > mx ? A[(size_t)t * D4G_NLIT + i] : mx;
        lim += mx == 15;
    }
    printf("%d of %d histograms differ (%d with 65..127 used symbols, %d reach the length limit)\n", bad, N, mid, lim);
    return bad != 0;
}
