__device__ long long g_t[8];
extern "C" __device__ int d4g_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
#define __builtin_amdgcn_writelane d4g_llvm_writelane
template <int NREG, typename H, typename I, int MAXN, int IDB_, bool OVL, typename FreqFn, typename OutFn>
__device__ int tree64_exp(TreeMem<H, I, MAXN, IDB_, OVL>& m, int numSymbols, int limit, FreqFn freq, OutFn outLen) {
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
    const int lane = threadIdx.x & 63;
    long long tStart = clock64();
    numSymbols = d4g_uniform(numSymbols);
    limit = d4g_uniform(limit);
    // does it fit?  used symbols (plus the dummy leaves that bring a queue of fewer than two up to two) and the total weight
    int used = 0;
    unsigned long long total = 0;
    for (int base = 0; base < numSymbols; base += 64) {
        const int i = base + lane;
        const unsigned f = i < numSymbols ? (unsigned)freq(i) : 0u;
        used += __popcll(__ballot(f != 0));
        total += f;
    }
    total = (unsigned long long)wave_sum_i64((long long)total);
    if (used > 64 || total >= (1ull << 24) - 4) return d4g_build_tree_wave<NREG>(m, numSymbols, limit, freq, outLen);
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(3);
#endif
    unsigned hv = 0;   // the queue: slot = lane
    int hs = 0;
    const int parentLane = (lane - 1) >> 1;   // (lane 0: -1, masked below)
    const int leftLane = 2 * lane + 1;
    // anc: this slot and its ancestors, the root aside (bit per slot) — a constant of the lane
    unsigned ancLo = 0, ancHi = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int A = ((lane + 1) >> j) - 1;
        if (A >= 1) { if (A < 32) ancLo |= 1u << A; else ancHi |= 1u << (A - 32); }
    }
    const unsigned bitLo = lane < 32 ? 1u << lane : 0u, bitHi = lane >= 32 ? 1u << (lane - 32) : 0u;
#if VAR_OFFER == 1
    auto offer = [&](unsigned x) D4G_LAMBDA_INLINE {
        int k = hs++;
        const unsigned xw = x >> 8;
        while (k > 0) {
            const int p = (k - 1) >> 1;
            const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)hv, p);
            if ((e >> 8) <= xw) break;
            hv = (unsigned)__builtin_amdgcn_writelane((int)e, k, (int)hv);
            k = p;
        }
        hv = (unsigned)__builtin_amdgcn_writelane((int)x, k, (int)hv);
    };
#else
    auto offer = [&](unsigned x) D4G_LAMBDA_INLINE {
        const int k = hs++;
        const unsigned xw = x >> 8;
        const unsigned pv = (unsigned)__shfl((int)hv, parentLane & 63);
        // the slots from the root down to k: k's ancestor set (a lane constant of lane k) and the root
        const unsigned pLo = (unsigned)d4g_readlane((int)ancLo, k) | 1u, pHi = (unsigned)d4g_readlane((int)ancHi, k);
        const bool onPath = ((pLo & bitLo) | (pHi & bitHi)) != 0;
        const bool pGreater = lane > 0 && (pv >> 8) > xw;     // the parent comes down into this slot
        const bool selfGreater = (hv >> 8) > xw;
        if (onPath) {
            if (pGreater) hv = pv;
            else if (lane == k || selfGreater) hv = x;
        }
    };
#endif
#if VAR_POLL == 1
    auto poll = [&]() D4G_LAMBDA_INLINE -> unsigned {
        const unsigned out = (unsigned)__builtin_amdgcn_readlane((int)hv, 0);
        const int s = --hs;
        if (s == 0) return out;
        const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)hv, s);
        const unsigned xw = x >> 8;
        int k = 0;
        const int half = s >> 1;
        while (k < half) {
            int child = 2 * k + 1;
            unsigned cvv = (unsigned)__builtin_amdgcn_readlane((int)hv, child);
            const unsigned rvv = (unsigned)__builtin_amdgcn_readlane((int)hv, (child + 1) & 63);
            if (child + 1 < s && (cvv >> 8) > (rvv >> 8)) { cvv = rvv; child++; }
            if (xw <= (cvv >> 8)) break;
            hv = (unsigned)__builtin_amdgcn_writelane((int)cvv, k, (int)hv);
            k = child;
        }
        hv = (unsigned)__builtin_amdgcn_writelane((int)x, k, (int)hv);
        return out;
    };
#else
    auto poll = [&]() D4G_LAMBDA_INLINE -> unsigned {
        const unsigned out = (unsigned)d4g_readlane((int)hv, 0);
        const int s = --hs;
        if (s == 0) return out;
        const unsigned x = (unsigned)d4g_readlane((int)hv, s);
        const unsigned xw = x >> 8;
        const unsigned lv = (unsigned)__shfl((int)hv, leftLane & 63), rv = (unsigned)__shfl((int)hv, (leftLane + 1) & 63);
        const bool hasL = leftLane < s;
        const bool right = leftLane + 1 < s && (lv >> 8) > (rv >> 8);
        const unsigned cv = right ? rv : lv;
        const unsigned long long Rm = __ballot(right), Hm = __ballot(hasL);
        // C: the slots that are the child their parent would hand up — slot 2P+1 when P hands up its left child, 2P+2 when
        // its right one (only P < 32 has children): the parents' bits spread to the even positions, on the scalar unit
        const unsigned long long C = (d4g_spread_bits((unsigned)(Hm & ~Rm)) << 1) | (d4g_spread_bits((unsigned)Rm) << 2);
        // on the root's hand-up path: this slot and all its ancestors (the root aside) are such children
        const bool onPath = lane < s && ((unsigned)C & ancLo) == ancLo && ((unsigned)(C >> 32) & ancHi) == ancHi;
        const bool pass = onPath && hasL && xw > (cv >> 8);
        const bool stop = onPath && !pass && (lane == 0 || xw > (hv >> 8));
        if (pass) hv = cv;
        else if (stop) hv = x;
        return out;
    };
#endif
    long long tA = clock64();
    // leaves in symbol order
    int nl = 0;
    for (int base = 0; base < numSymbols; base += 64) {
        const int i = base + lane;
        const int fv = i < numSymbols ? (int)freq(i) : 0;
        unsigned long long um = __ballot(fv != 0);
        while (um) {
            const int bpos = __ffsll((long long)um) - 1;
            um &= um - 1;
            const unsigned f = (unsigned)d4g_readlane(fv, bpos);
            if (lane == 0) m.value[nl] = (I)(base + bpos);
            offer((f << 8) | (unsigned)nl);
            nl++;
        }
    }
    int index = 0;
    while (hs < 2) {  // dummy leaves — HuffmanTree.java:50-58
        bool unused = index >= numSymbols;
        if (!unused) unused = d4g_uniform((int)freq(index)) == 0;
        if (unused) {
            if (lane == 0) m.value[nl] = (I)index;
            offer((1u << 8) | (unsigned)nl);
            nl++;
        }
        index++;
    }
    long long tB = clock64();
    int nn = nl;
    for (int i = 0; i < nl - 1; i++) {
        const unsigned a = poll(), b = poll();
        const int l = (int)(a & 255u), r = (int)(b & 255u);
        const int id = nn++;
        if (lane == 0) {
            m.left[id - nl] = (I)l;
            m.right[id - nl] = (I)r;
            m.parent[l] = (I)id;
            m.parent[r] = (I)(id | SIDE);
        }
        offer((((a >> 8) + (b >> 8)) << 8) | (unsigned)id);
    }
    const int root = (int)(poll() & 255u);
    d4g_wave_sync();
    long long tC = clock64();
    // leaf depths: one lane per leaf walks to the root
    int dep = 0;
    if (lane < nl) {
        int node = lane;
        while (node != root) { node = m.parent[node] & ~SIDE; dep++; }
    }
    const int maxDepth = wave_max_i32(dep);
    if (maxDepth > limit) {
        int err = 0;
        if (lane == 0) err = d4g_tree_finish(m, 1, 0, nl, root, numSymbols, limit, outLen);
#ifndef D4G_HOSTSIM
        __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
        return __shfl(err, 0);
    }
    if (lane < nl) {
        const int v = m.value[lane];
        if (v < numSymbols) outLen(v, dep);
    }
    if (lane == 0) { g_t[0] += tA - tStart; g_t[1] += tB - tA; g_t[2] += tC - tB; g_t[3] += clock64() - tC; }
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    return 0;
}

