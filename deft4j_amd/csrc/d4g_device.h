// d4g_device.h — device-side building blocks of the gfx950 optimiser kernels.
//
// Everything here is integer work on HBM-resident token arrays with LDS-resident code
// tables (no MFMA: the path is branchy integer/byte work, HBM/LDS bound).  Each function
// cites the reference code whose RESULT it reproduces (B/ = deft4j-base/src/main/java/
// com/github/NeRdTheNed/deft4j/).  The algorithms are re-designed for wave64 execution:
// token lists are bit masks over the block's token range, header RLE pairs are generated
// on the fly from code-length runs, and Huffman construction replays the JDK binary heap
// in LDS so ties break exactly as java.util.PriorityQueue does.
#pragma once
#include "d4g_types.h"

#ifndef D4G_HOSTSIM
#include <hip/hip_runtime.h>
#endif

#define D4G_DEV __device__ __forceinline__
// Issue priority of the optimiser's waves (s_setprio): above the default 0, so that they are not starved by an old, always-ready
// wave of another kernel on the same SIMD (the Zopfli squeeze runs for minutes); serial sections go to 3 and come back here.
#define D4G_BASE_PRIO 1

// ---------------------------------------------------------------------------------------
// RFC 1951 symbol arithmetic (B/deflate/Constants.java:9-23,65-128) — closed forms instead
// of the reference's lookup tables, so no divergent table reads are needed.
// ---------------------------------------------------------------------------------------
D4G_DEV int d4g_len2sym(int len, int edge) {
    if (edge) return 284;                 // Constants.len2litlen edge case (:15-20)
    if (len == 258) return 285;
    int x = len - 3;
    if (x < 8) return 257 + x;
    int e = (31 - __clz(x)) - 2;
    return 261 + 4 * e + ((x >> e) & 3);
}
D4G_DEV int d4g_lsym_ebits(int s) { return (s < 265 || s == 285) ? 0 : (s - 261) >> 2; }
D4G_DEV int d4g_lsym_base(int s) {
    if (s < 265) return s - 254;
    if (s == 285) return 258;
    int e = (s - 261) >> 2;
    return 3 + ((4 + ((s - 261) & 3)) << e);
}
D4G_DEV int d4g_dist2sym(int d) {
    int y = d - 1;
    if (y < 4) return y;
    int n = 31 - __clz(y);
    return 2 * n + ((y >> (n - 1)) & 1);
}
D4G_DEV int d4g_dsym_ebits(int s) { return s < 4 ? 0 : (s >> 1) - 1; }
D4G_DEV int d4g_dsym_base(int s) {
    if (s < 4) return s + 1;
    int e = (s >> 1) - 1;
    return 1 + ((2 + (s & 1)) << e);
}
__device__ const uint8_t D4G_CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// token word helpers
D4G_DEV int tok_val(uint32_t a) { return (int)(a & 0x1ffu); }
D4G_DEV int tok_edge(uint32_t a) { return (int)((a >> 15) & 1u); }
D4G_DEV int tok_dist(uint32_t a) { return (int)(a >> 16); }
// bytes [sh/8, sh/8+4) of the little-endian pair {lo, hi}; sh in {0, 8, 16, 24} (v_alignbit_b32)
D4G_DEV uint32_t d4g_alignbit(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
// back-reference record helpers (refs[r].x, see d4g_types.h)
D4G_DEV uint32_t d4g_ref_pack(int len, int lsym, int dsym, int ebits) {
    return (uint32_t)len | ((uint32_t)(lsym - 257) << 9) | ((uint32_t)dsym << 14) | ((uint32_t)ebits << 19);
}
D4G_DEV int ref_len(uint32_t a) { return (int)(a & 0x1ffu); }
D4G_DEV int ref_lsym(uint32_t a) { return 257 + (int)((a >> 9) & 31u); }
D4G_DEV int ref_dsym(uint32_t a) { return (int)((a >> 14) & 31u); }
D4G_DEV int ref_ebits(uint32_t a) { return (int)((a >> 19) & 31u); }

// ---------------------------------------------------------------------------------------
// wave64 / workgroup reductions
// ---------------------------------------------------------------------------------------
D4G_DEV long long wave_sum_i64(long long v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// 32-bit wave sum: row scans and row broadcasts on the DPP path (six v_add_u32), no LDS crossbar trips
D4G_DEV int wave_sum_i32(int v) {
#ifdef D4G_HOSTSIM
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
#else
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
#endif
}
D4G_DEV long long wave_min_i64(long long v) {
    for (int m = 32; m >= 1; m >>= 1) {
        long long o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}
D4G_DEV int wave_max_i32(int v) {
    for (int m = 32; m >= 1; m >>= 1) {
        int o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}
// Workgroup-wide sum; `red` is an LDS array of >= 17 long long.  All threads call.
D4G_DEV long long wg_sum_i64(long long v, long long* red) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_i64(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int i = 0; i < nw; i++) s += red[i];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}
// Two workgroup-wide sums in one exchange (`red`: >= 32 long long).  All threads call; every thread adds the per-wave
// partials itself, so there is no serial stage.
D4G_DEV void wg_sum2_i64(long long a, long long b, long long* red, long long& sa, long long& sb) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    a = wave_sum_i64(a);
    b = wave_sum_i64(b);
    __syncthreads();
    if (lane == 0) { red[wave] = a; red[16 + wave] = b; }
    __syncthreads();
    sa = 0; sb = 0;
    for (int i = 0; i < nw; i++) { sa += red[i]; sb += red[16 + i]; }
    __syncthreads();
}
D4G_DEV int wg_max_i32(int v, long long* red) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max_i32(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = red[0];
        for (int i = 1; i < nw; i++) s = red[i] > s ? red[i] : s;
        red[16] = s;
    }
    __syncthreads();
    return (int)red[16];
}

// ---------------------------------------------------------------------------------------
// Huffman tree with exact JDK PriorityQueue tie-breaking and the reference's depth limiter.
// Reproduces HuffmanTree(freq, limit).getTable().codeLen — B/huffman/HuffmanTree.java:36-128
// (constructor + limiter), :134-158 (traverse), :164-192 (getTable).  One lane builds one
// tree; storage is strided so 64 lanes can each build a small tree side by side in LDS.
//   W = weight type, I = node-index type, MAXN = max leaves.
//   element i of an array lives at arr[i * stride + lane].
// ---------------------------------------------------------------------------------------
// Heap entries carry their weight (weight << IDBITS | node id), so one LDS read per heap slot decides a
// sift step; node weights are never looked up separately.  H = heap entry type (u64 for the
// literal/length and distance trees, u32 for the 19-symbol code-length tree).
// IDB = bits of a heap entry that hold the node id; OVERLAY = firstAt / depth reuse the heap's memory (the
// queue is empty by the time depths are computed) — the 64-trees-per-wave header search uses both to stay small.
template <typename H, typename I, int MAXN, int IDB_ = 8 * (int)sizeof(I), bool OVERLAY = false>
struct TreeMem {
    H* heap;     // [MAXN+1]  the priority queue
    I* left;     // [MAXN]    children of internal node id live at index id - nl
    I* right;    // [MAXN]
    I* parent;   // [2*MAXN]  (bit (8*sizeof(I)-1) = side)
    I* value;    // [MAXN]    symbol of leaf
    I* firstAt;  // [MAXN+1]  first leaf (DFS order) at each depth
    I* depth;    // [MAXN]    depth of each leaf
    static constexpr int NONE = (1 << (8 * sizeof(I) - 1)) - 1;
    static constexpr int SIDE = 1 << (8 * sizeof(I) - 1);
    static constexpr int IDBITS = IDB_;
    static_assert(!OVERLAY || sizeof(H) * (MAXN + 1) >= sizeof(I) * (2 * MAXN + 1), "heap too small to hold firstAt + depth");
    static constexpr size_t bytes(int lanes) {
        return (size_t)lanes * (sizeof(H) * (MAXN + 1) + sizeof(I) * (MAXN * 2 + 2 * MAXN + MAXN + (OVERLAY ? 0 : (MAXN + 1) + MAXN)));
    }
    __device__ void carve(unsigned char* base, int lanes) {
        heap = (H*)base; base += sizeof(H) * (MAXN + 1) * lanes;
        left = (I*)base; base += sizeof(I) * MAXN * lanes;
        right = (I*)base; base += sizeof(I) * MAXN * lanes;
        parent = (I*)base; base += sizeof(I) * 2 * MAXN * lanes;
        value = (I*)base; base += sizeof(I) * MAXN * lanes;
        if (OVERLAY) {
            firstAt = (I*)heap;
            depth = firstAt + (MAXN + 1) * lanes;
        } else {
            firstAt = (I*)base; base += sizeof(I) * (MAXN + 1) * lanes;
            depth = (I*)base;
        }
    }
};

template <typename H, typename I, int MAXN, int IDB_, bool OVL, typename OutFn>
__device__ int d4g_tree_finish(TreeMem<H, I, MAXN, IDB_, OVL>& m, int stride, int lane, int nl, int root, int numSymbols, int limit,
                               OutFn outLen);

// Returns 0 on success, 1 if the limiter could not rebalance (the reference throws there).
// freq(i) reads symbol i's frequency; outLen(i, len) receives each symbol's code length.
template <typename H, typename I, int MAXN, int IDB_, bool OVL, typename FreqFn, typename OutFn>
__device__ int d4g_build_tree(TreeMem<H, I, MAXN, IDB_, OVL>& m, int stride, int lane, int numSymbols, int limit, FreqFn freq,
                              OutFn outLen) {
    const int NONE = TreeMem<H, I, MAXN, IDB_, OVL>::NONE;
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
    const int IDB = TreeMem<H, I, MAXN, IDB_, OVL>::IDBITS;
    const H IDMASK = ((H)1 << IDB) - 1;
#define TM(arr, i) m.arr[(i) * stride + lane]
    int nl = 0, hs = 0;
    // java.util.PriorityQueue.offer: append, sift up while key < parent (stop on >=); keys compare by weight only
    auto pq_add = [&](H x) {
        int k = hs++;
        H wx = x >> IDB;
        while (k > 0) {
            int p = (k - 1) >> 1;
            H e = TM(heap, p);
            if (wx >= (e >> IDB)) break;
            TM(heap, k) = e;
            k = p;
        }
        TM(heap, k) = x;
    };
    // PriorityQueue.poll: root out, last element sifts down; left child unless left > right; stop when key <= child
    auto pq_remove = [&]() -> H {
        H result = TM(heap, 0);
        int s = --hs;
        H x = TM(heap, s);
        if (s != 0) {
            H wx = x >> IDB;
            int k = 0, half = s >> 1;
            while (k < half) {
                int child = 2 * k + 1;
                H c = TM(heap, child);
                int r = child + 1;
                if (r < s) {
                    H cr = TM(heap, r);
                    if ((c >> IDB) > (cr >> IDB)) { c = cr; child = r; }
                }
                if (wx <= (c >> IDB)) break;
                TM(heap, k) = c;
                k = child;
            }
            TM(heap, k) = x;
        }
        return result;
    };
    for (int i = 0; i < numSymbols; i++) {
        unsigned f = freq(i);
        if (f > 0) {
            TM(value, nl) = (I)i;
            pq_add(((H)f << IDB) | (H)nl);
            nl++;
        }
    }
    int index = 0;
    while (hs < 2) {  // dummy leaves — HuffmanTree.java:50-58
        if (index >= numSymbols || freq(index) == 0) {
            TM(value, nl) = (I)index;
            pq_add(((H)1 << IDB) | (H)nl);
            nl++;
        }
        index++;
    }
    int nn = nl;
    for (int i = 0; i < nl - 1; i++) {
        H hl = pq_remove();
        H hr = pq_remove();
        int l = (int)(hl & IDMASK), r = (int)(hr & IDMASK);
        int id = nn++;
        TM(left, id - nl) = (I)l;
        TM(right, id - nl) = (I)r;
        TM(parent, l) = (I)id;
        TM(parent, r) = (I)(id | SIDE);
        pq_add((((hl >> IDB) + (hr >> IDB)) << IDB) | (H)id);
    }
    int root = (int)(pq_remove() & IDMASK);
#undef TM
    return d4g_tree_finish(m, stride, lane, nl, root, numSymbols, limit, outLen);
}

// Second half of the builder: depths by DFS (traverse :134-158), the depth limiter (:75-127) and the
// lengths (getTable :164-192) of a tree whose nodes are already in m.left / right / parent / value.
template <typename H, typename I, int MAXN, int IDB_, bool OVL, typename OutFn>
__device__ int d4g_tree_finish(TreeMem<H, I, MAXN, IDB_, OVL>& m, int stride, int lane, int nl, int root, int numSymbols, int limit,
                               OutFn outLen) {
    const int NONE = TreeMem<H, I, MAXN, IDB_, OVL>::NONE;
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
#define TM(arr, i) m.arr[(i) * stride + lane]
    int maxDepth = 0;
    // traverse — DFS, left before right; records each leaf's depth and the first leaf per depth
    auto traverse = [&]() {
        for (int d = 0; d <= nl; d++) TM(firstAt, d) = (I)NONE;
        maxDepth = 0;
        int node = root, depth = 0;
        while (true) {
            while (node >= nl) { node = TM(left, node - nl); depth++; }
            TM(depth, node) = (I)depth;
            if (TM(firstAt, depth) == (I)NONE) TM(firstAt, depth) = (I)node;
            if (depth > maxDepth) maxDepth = depth;
            while (node != root && (TM(parent, node) & SIDE)) { node = TM(parent, node) & ~SIDE; depth--; }
            if (node == root) break;
            node = TM(right, (TM(parent, node) & ~SIDE) - nl);
        }
    };
    traverse();
    int err = 0;
    while (maxDepth > limit) {  // HuffmanTree.java:75-127
        int leafA = TM(firstAt, maxDepth);
        int pa = TM(parent, leafA);
        int p1 = pa & ~SIDE;
        int leafB = (pa & SIDE) ? TM(left, p1 - nl) : TM(right, p1 - nl);
        int pp = TM(parent, p1);
        int p2 = pp & ~SIDE;
        if (pp & SIDE) { TM(right, p2 - nl) = (I)leafB; TM(parent, leafB) = (I)(p2 | SIDE); }
        else { TM(left, p2 - nl) = (I)leafB; TM(parent, leafB) = (I)p2; }
        bool moved = false;
        for (int i = maxDepth - 2; i >= 1; i--) {
            int leafC = TM(firstAt, i);
            if (leafC != NONE) {
                int pc = TM(parent, leafC);
                int p3 = pc & ~SIDE;
                int in = p1;  // the detached parent's slot is reused for new InternalNode(leafA, leafC)
                TM(left, in - nl) = (I)leafA;
                TM(right, in - nl) = (I)leafC;
                TM(parent, leafA) = (I)in;
                TM(parent, leafC) = (I)(in | SIDE);
                if (pc & SIDE) { TM(right, p3 - nl) = (I)in; TM(parent, in) = (I)(p3 | SIDE); }
                else { TM(left, p3 - nl) = (I)in; TM(parent, in) = (I)p3; }
                moved = true;
                break;
            }
        }
        if (!moved) { err = 1; break; }
        traverse();
    }
    // getTable: only lengths are needed (codes are canonical by (length, symbol))
    for (int i = 0; i < nl; i++) {
        int v = TM(value, i);
        if (v < numSymbols) outLen(v, (int)TM(depth, i));
    }
#undef TM
    return err;
}

// ---------------------------------------------------------------------------------------
// The same tree built by one whole wave.  The first 64 slots of the priority queue (its top six
// levels, where every sift passes) live in a register pair: slot k is lane k of `w0` (weight) and `i0`
// (node id), read with v_readlane at a wave-uniform index and written with a compare + select, so those
// sift steps cost no LDS round trip; deeper slots stay in LDS (m.heap).  Every lane runs the same code on
// the same wave-uniform values, which keeps the queue's control flow on the scalar unit; lane 0 records
// the tree's nodes in LDS.  Leaf depths are then counted by one lane per leaf; only a tree deeper than
// `limit` goes through the serial DFS + limiter of d4g_tree_finish.
// ---------------------------------------------------------------------------------------
#define D4G_LAMBDA_INLINE __attribute__((always_inline))
// The wave's vote on a condition.  (HIP's __ballot takes an int: a bool goes through 0 / 1 and a second compare.)
#ifdef D4G_HOSTSIM
D4G_DEV unsigned long long d4g_ballot(bool p) { return __ballot(p ? 1 : 0); }
#else
D4G_DEV unsigned long long d4g_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif
#ifdef D4G_HOSTSIM
D4G_DEV int d4g_readlane(int v, int k) { return __shfl(v, k); }
D4G_DEV int d4g_uniform(int v) { return __shfl(v, 0); }
D4G_DEV void d4g_wave_sync() { (void)__shfl(0, 0); }   // the emulator's lanes are not in lock step: rendezvous
#else
D4G_DEV int d4g_readlane(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
D4G_DEV int d4g_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
D4G_DEV void d4g_wave_sync() { __builtin_amdgcn_wave_barrier(); }  // LDS accesses of one wave are already in order
#endif

struct D4GWaveHeap {
    int w0, i0;
    uint64_t* deep;   // LDS, slots >= 64: weight << 32 | id
    // register slots (k < 64)
    D4G_DEV unsigned regW(int k) const { return (unsigned)d4g_readlane(w0, k); }
    D4G_DEV int regI(int k) const { return d4g_readlane(i0, k); }
    D4G_DEV void regPut(int k, unsigned ww, int ii) {
        const bool me = (int)(threadIdx.x & 63) == k;
        w0 = me ? (int)ww : w0;
        i0 = me ? ii : i0;
    }
    // any slot
    D4G_DEV void get(int k, unsigned& ww, int& ii) const {
        if (k < 64) { ww = regW(k); ii = regI(k); }
        else {
            uint64_t e = deep[k];
            ww = (unsigned)d4g_uniform((int)(e >> 32));
            ii = d4g_uniform((int)(uint32_t)e);
        }
    }
    D4G_DEV void put(int k, unsigned ww, int ii) {
        if (k < 64) regPut(k, ww, ii);
        else if ((threadIdx.x & 63) == 0) deep[k] = ((uint64_t)ww << 32) | (uint32_t)ii;
    }
    // slots k and k + 1 (k + 1 may be one past the queue's end: still inside the array, the caller ignores it): both LDS reads
    // go out before either is waited for
    D4G_DEV void get2(int k, unsigned& w0o, int& i0o, unsigned& w1o, int& i1o) const {
        if (k >= 64) {
            const uint64_t e0 = deep[k], e1 = deep[k + 1];
            w0o = (unsigned)d4g_uniform((int)(e0 >> 32)); i0o = d4g_uniform((int)(uint32_t)e0);
            w1o = (unsigned)d4g_uniform((int)(e1 >> 32)); i1o = d4g_uniform((int)(uint32_t)e1);
        } else { get(k, w0o, i0o); get(k + 1, w1o, i1o); }
    }
};

#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
__device__ unsigned long long d4g_dbg_tree[4];       // literal/length tree sections: leaves, merges, depths
__device__ unsigned long long d4g_dbg_counters[8];   // profile builds: [0] wave trees built, [1] of them through the serial limiter, [2] leaves
#endif
// All 64 lanes of one wave call this with the same arguments.  NREG = ceil(MAXN / 64).
template <int NREG, typename H, typename I, int MAXN, int IDB_, bool OVL, typename FreqFn, typename OutFn>
__device__ int d4g_build_tree_wave(TreeMem<H, I, MAXN, IDB_, OVL>& m, int numSymbols, int limit, FreqFn freq, OutFn outLen) {
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
    const int lane = threadIdx.x & 63;
    numSymbols = d4g_uniform(numSymbols);  // tell the compiler what is wave-uniform: the queue code then runs on the scalar unit
    limit = d4g_uniform(limit);
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(3);   // a serial section everyone in the workgroup waits for: issue ahead of the throughput-bound waves
#endif
    D4GWaveHeap hp;
    hp.w0 = 0; hp.i0 = 0;
    hp.deep = (uint64_t*)m.heap;
    int nl = 0, hs = 0;
    // java.util.PriorityQueue.offer / poll, as in d4g_build_tree
    auto pq_add = [&](unsigned wx, int idx) D4G_LAMBDA_INLINE {
        int k = hs++;
        while (k > 0) {
            int p = (k - 1) >> 1;
            unsigned pw; int pi;
            hp.get(p, pw, pi);
            if (wx >= pw) break;
            hp.put(k, pw, pi);
            k = p;
        }
        hp.put(k, wx, idx);
    };
    auto pq_remove = [&](unsigned& rw, int& ri) D4G_LAMBDA_INLINE {
        rw = hp.regW(0);
        ri = hp.regI(0);
        int s = --hs;
        if (s != 0) {
            unsigned xw; int xi;
            hp.get(s, xw, xi);
            int k = 0;
            const int half = s >> 1;
            // levels whose children are both register slots: no region checks, the child's id is read only once chosen
            const int regHalf = half < 31 ? half : 31;   // k < 31  =>  2k + 2 < 64
            bool placed = false;
            if (regHalf > 0) {
                // every lane k picks the child its slot would hand up (left unless left > right), once for the whole
                // path: the slots below the moving position are not touched until the walk reaches them
                const int li = 2 * lane + 1;
                const unsigned lwv = (unsigned)__shfl(hp.w0, li & 63), rwv = (unsigned)__shfl(hp.w0, (li + 1) & 63);
                const bool right = li + 1 < s && lwv > rwv;
                const int cidx = right ? li + 1 : li;
                const int cwv = (int)(right ? rwv : lwv);
                // (bounded and unrolled — at most five register levels, slots 0..30 — instead of a while loop: a taken
                // back-edge costs a lone wave ~45 cycles per level)
#pragma unroll
                for (int lev = 0; lev < 5; lev++) {
                    if (k >= regHalf) break;
                    int child = d4g_readlane(cidx, k);
                    unsigned cw = (unsigned)d4g_readlane(cwv, k);
                    if (xw <= cw) { placed = true; break; }
                    hp.regPut(k, cw, hp.regI(child));
                    k = child;
                }
            }
            if (!placed) {
#pragma unroll
                for (int lev = 0; lev < 6; lev++) {   // the rest of the path (slots in LDS): at most 2^11 slots
                    if (k >= half) break;
                    int child = 2 * k + 1;
                    unsigned cw, rw2; int ci, ri2;
                    hp.get2(child, cw, ci, rw2, ri2);
                    if (child + 1 < s && cw > rw2) { cw = rw2; ci = ri2; child = child + 1; }
                    if (xw <= cw) break;
                    hp.put(k, cw, ci);
                    k = child;
                }
            }
            hp.put(k, xw, xi);
        }
    };
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    long long tq0 = clock64();
#endif
    // leaves in symbol order: 64 frequencies per step, the used ones are offered one by one
    for (int base = 0; base < numSymbols; base += 64) {
        int i = base + lane;
        int fv = i < numSymbols ? (int)freq(i) : 0;
        unsigned long long um = d4g_ballot(fv != 0);
        while (um) {
            int bpos = __ffsll((long long)um) - 1;
            um &= um - 1;
            unsigned f = (unsigned)d4g_readlane(fv, bpos);
            if (lane == 0) m.value[nl] = (I)(base + bpos);
            pq_add(f, nl);
            nl++;
        }
    }
    int index = 0;
    while (hs < 2) {  // dummy leaves — HuffmanTree.java:50-58
        bool unused = index >= numSymbols;
        if (!unused) unused = d4g_uniform((int)freq(index)) == 0;
        if (unused) {
            if (lane == 0) m.value[nl] = (I)index;
            pq_add(1u, nl);
            nl++;
        }
        index++;
    }
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    long long tq1 = clock64();
#endif
    int nn = nl;
    for (int i = 0; i < nl - 1; i++) {
        unsigned lw, rw;
        int l, r;
        pq_remove(lw, l);
        pq_remove(rw, r);
        int id = nn++;
        if (lane == 0) {
            m.left[id - nl] = (I)l;
            m.right[id - nl] = (I)r;
            m.parent[l] = (I)id;
            m.parent[r] = (I)(id | SIDE);
        }
        pq_add(lw + rw, id);
    }
    unsigned rootW;
    int root;
    pq_remove(rootW, root);
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    long long tq2 = clock64();
    if (lane == 0 && MAXN > 100) {
        atomicAdd(&d4g_dbg_tree[0], (unsigned long long)(tq1 - tq0));
        atomicAdd(&d4g_dbg_tree[1], (unsigned long long)(tq2 - tq1));
    }
#endif
    // leaf depths: one lane per leaf walks to the root
    int dep[NREG];
    int maxDepth = 0;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        int leaf = r * 64 + lane, d = 0;
        if (leaf < nl) {
            int node = leaf;
            while (node != root) { node = m.parent[node] & ~SIDE; d++; }
        }
        dep[r] = d;
        maxDepth = d > maxDepth ? d : maxDepth;
    }
    maxDepth = wave_max_i32(maxDepth);
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    if (lane == 0 && MAXN > 100) atomicAdd(&d4g_dbg_tree[2], (unsigned long long)(clock64() - tq2));
    if (lane == 0) {
        int slot = MAXN > 100 ? 0 : (MAXN > 20 ? 3 : 5);   // literal/length, distance, code-length trees
        atomicAdd(&d4g_dbg_counters[slot], 1ULL);
        if (maxDepth > limit) atomicAdd(&d4g_dbg_counters[slot + 1], 1ULL);
        if (slot == 0) atomicAdd(&d4g_dbg_counters[2], (unsigned long long)nl);
    }
#endif
    if (maxDepth > limit) {
        int err = 0;
        if (lane == 0) err = d4g_tree_finish(m, 1, 0, nl, root, numSymbols, limit, outLen);
#ifndef D4G_HOSTSIM
        __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
        return __shfl(err, 0);
    }
#pragma unroll
    for (int r = 0; r < NREG; r++) {
        int leaf = r * 64 + lane;
        if (leaf < nl) {
            int v = m.value[leaf];
            if (v < numSymbols) outLen(v, dep[r]);
        }
    }
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    return 0;
}

// ---------------------------------------------------------------------------------------
// The same tree for 65 .. 127 leaves (text with capitals, digits and punctuation; many binary formats): the queue with every
// lane holding the root-to-leaf path of "its" leaf position.  Level d (0..6) of the implicit heap has 2^d slots; register R[d]
// of lane L holds slot 2^d - 1 + (L >> (6 - d)): a slot of level d is replicated over the 2^(6-d) lanes below it.  offer() is
// lane-local (a lane sees its slot and the slot's parent in its own registers); poll() needs one exchange per level with the
// lane that holds the sibling subtree (lane ^ 2^(5-d)).  Empty slots hold 0xffffffff; an entry is weight << 8 | node id, node
// ids below 255, weights below 2^24.  Slower than the one-register queue below for up to 64 leaves (twelve instructions per
// level of a poll), several times faster than the general queue above, whose deep slots live in LDS.
// ---------------------------------------------------------------------------------------
#define D4G_RP_INF 0xffffffffu
template <int D> D4G_DEV unsigned d4g_rp_sibling(unsigned v) {   // the value held by the lane 2^(5-D) away
#ifdef D4G_HOSTSIM
    return (unsigned)__shfl_xor((int)v, 32 >> D);
#else
    if (D == 0) return (unsigned)__shfl_xor((int)v, 32);
    if (D == 1) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                  // swap with lane ^ 16
    if (D == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, false);    // row_ror:8
    if (D == 3) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                  // swap with lane ^ 4
    if (D == 4) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false);                  // quad_perm [1,0,3,2]
#endif
}
struct D4GPathHeap {
    unsigned R[7];
    unsigned mA[6];   // 0xff where this lane's slot of level d + 1 is a right child
    int hs, lane;
    D4G_DEV void init() {
        lane = threadIdx.x & 63;
        hs = 0;
#pragma unroll
        for (int d = 0; d < 7; d++) R[d] = D4G_RP_INF;
#pragma unroll
        for (int d = 0; d < 6; d++) mA[d] = ((lane >> (5 - d)) & 1) ? 0xffu : 0u;
    }
    // x = weight << 8 | id with id larger than every id in the queue: equal weights then compare as "x is not smaller" (siftUp's
    // x >= e).  Every level reads its parent's old value: a parent heavier than x comes down, x lands in the topmost slot
    // heavier than it (the slot being filled holds 0xffffffff).
    D4G_DEV void offer(unsigned x) {
        const int k = hs++;
        const int D = 31 - __builtin_clz((unsigned)(k + 1));
        const int Lk = (k + 1 - (1 << D)) << (6 - D);
        const unsigned z = (unsigned)(lane ^ Lk);
#pragma unroll
        for (int d = 6; d >= 1; d--) {
            const unsigned xd = z < (1u << (6 - d)) ? x : D4G_RP_INF;
            const unsigned keep = R[d] < xd ? R[d] : xd;
            R[d] = R[d - 1] > xd ? R[d - 1] : keep;
        }
        R[0] = R[0] < x ? R[0] : x;
    }
    template <int D> D4G_DEV unsigned sib(unsigned v) { return d4g_rp_sibling<D>(v); }
    template <int D> D4G_DEV void sift(int s) {   // the last slot s (on level D >= 1) is taken out and sifted down from the root
        const int Ls = (s + 1 - (1 << D)) << (6 - D);
        const unsigned x = (unsigned)d4g_readlane((int)R[D], Ls);
        const unsigned zs = (unsigned)(lane ^ Ls);
        R[D] = zs < (1u << (6 - D)) ? D4G_RP_INF : R[D];
        unsigned y = x;   // what arrives at (or already sits in) this lane's slot of the level in hand
#define D4G_RP_LEVEL(d)                                                                                                        \
        if ((d) < D) {                                                                                                         \
            const unsigned A = R[(d) + 1], B = sib<(d) < 6 ? (d) : 5>(A);                                                      \
            const bool own = (A | mA[(d) < 6 ? (d) : 5]) < (B | (mA[(d) < 6 ? (d) : 5] ^ 0xffu));   /* left unless left > right */ \
            const unsigned cv = own ? A : B;                                                                                   \
            const bool pass = (cv | 0xffu) < y;   /* x heavier than the child handed up (never true where x has not arrived) */ \
            R[(d)] = pass ? cv : y;                                                                                            \
            y = (pass && own) ? y : A;                                                                                         \
        }
        D4G_RP_LEVEL(0) D4G_RP_LEVEL(1) D4G_RP_LEVEL(2) D4G_RP_LEVEL(3) D4G_RP_LEVEL(4) D4G_RP_LEVEL(5)
#undef D4G_RP_LEVEL
        R[D] = y;
    }
    D4G_DEV unsigned poll() {
        const unsigned out = (unsigned)d4g_readlane((int)R[0], 0);
        const int s = --hs;
        if (s == 0) { R[0] = D4G_RP_INF; return out; }
        const int D = 31 - __builtin_clz((unsigned)(s + 1));
        if (D == 6) sift<6>(s);
        else if (D == 5) sift<5>(s);
        else if (D == 4) sift<4>(s);
        else if (D == 3) sift<3>(s);
        else if (D == 2) sift<2>(s);
        else sift<1>(s);
        return out;
    }
};
// used leaves (with the dummies) at most 127, total weight below 2^24 - 4: checked by the caller
template <typename H, typename I, int MAXN, int IDB_, bool OVL, typename FreqFn, typename OutFn>
__device__ int d4g_build_tree_wave_path(TreeMem<H, I, MAXN, IDB_, OVL>& m, int numSymbols, int limit, FreqFn freq, OutFn outLen) {
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
    const int lane = threadIdx.x & 63;
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(3);
#endif
    D4GPathHeap hp;
    hp.init();
    int nl = 0;
    for (int base = 0; base < numSymbols; base += 64) {
        const int i = base + lane;
        const int fv = i < numSymbols ? (int)freq(i) : 0;
        unsigned long long um = d4g_ballot(fv != 0);
        while (um) {
            const int bpos = __ffsll((long long)um) - 1;
            um &= um - 1;
            const unsigned f = (unsigned)d4g_readlane(fv, bpos);
            if (lane == 0) m.value[nl] = (I)(base + bpos);
            hp.offer((f << 8) | (unsigned)nl);
            nl++;
        }
    }
    int index = 0;
    while (hp.hs < 2) {  // dummy leaves — HuffmanTree.java:50-58
        bool unused = index >= numSymbols;
        if (!unused) unused = d4g_uniform((int)freq(index)) == 0;
        if (unused) {
            if (lane == 0) m.value[nl] = (I)index;
            hp.offer((1u << 8) | (unsigned)nl);
            nl++;
        }
        index++;
    }
    int nn = nl;
    for (int i = 0; i < nl - 1; i++) {
        const unsigned a = hp.poll(), b = hp.poll();
        const int l = (int)(a & 255u), r = (int)(b & 255u);
        const int id = nn++;
        if (lane == 0) {
            m.left[id - nl] = (I)l;
            m.right[id - nl] = (I)r;
            m.parent[l] = (I)id;
            m.parent[r] = (I)(id | SIDE);
        }
        hp.offer((((a >> 8) + (b >> 8)) << 8) | (unsigned)id);
    }
    const int root = (int)(hp.poll() & 255u);
    d4g_wave_sync();
    // leaf depths: two leaves per lane walk to the root
    int dep[2] = {0, 0};
    int maxDepth = 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int leaf = r * 64 + lane;
        if (leaf < nl) {
            int node = leaf, d = 0;
            while (node != root) { node = m.parent[node] & ~SIDE; d++; }
            dep[r] = d;
        }
        maxDepth = dep[r] > maxDepth ? dep[r] : maxDepth;
    }
    maxDepth = wave_max_i32(maxDepth);
    if (maxDepth > limit) {
        int err = 0;
        if (lane == 0) err = d4g_tree_finish(m, 1, 0, nl, root, numSymbols, limit, outLen);
#ifndef D4G_HOSTSIM
        __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
        return __shfl(err, 0);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int leaf = r * 64 + lane;
        if (leaf < nl) {
            const int v = m.value[leaf];
            if (v < numSymbols) outLen(v, dep[r]);
        }
    }
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    return 0;
}

// ---------------------------------------------------------------------------------------
// The same tree again for the common case of at most 64 leaves and weights below 2^24 (a text block uses ~55 of the
// literal/length symbols; the distance and code-length alphabets always fit): the whole priority queue is ONE register,
// slot k = lane k, entry = weight << 8 | node id, and offer / poll are done by all lanes at once instead of one sift
// step at a time —
//   offer(x): the slots on the path from the new last slot to the root are known from the slot numbers alone; a slot
//             on it takes its parent's entry when that is heavier than x (JDK siftUp moves a parent down only when it
//             is strictly heavier), and x lands in the topmost slot whose own entry is heavier (or in the new slot);
//   poll():   every slot picks the child it would hand up (left unless left > right); the chosen-child bits of all
//             slots say which slots lie on the root's hand-up path; x (the former last entry) passes a slot while it
//             is heavier than the child handed up there (siftDown stops on <=) — weights do not decrease along the
//             path, so the passed slots are a prefix of it: each takes its chosen child's entry, the first slot not
//             passed takes x.
// No loop over sift levels, no branches: two cross-lane reads per poll, one per offer.  Results (node arrays in `m`,
// depths, the limiter path) are those of d4g_build_tree_wave, which is also what larger inputs fall back to.
// ---------------------------------------------------------------------------------------
// bit p of x -> bit 2p (p < 32)
D4G_DEV unsigned long long d4g_spread_bits(unsigned x) {
#ifdef D4G_HOSTSIM
    unsigned long long v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
#else
    unsigned long long o;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(o) : "s"(x));   // every bit doubled
    return o & 0x5555555555555555ull;
#endif
}
template <int NREG, typename H, typename I, int MAXN, int IDB_, bool OVL, typename FreqFn, typename OutFn>
__device__ int d4g_build_tree_wave64(TreeMem<H, I, MAXN, IDB_, OVL>& m, int numSymbols, int limit, FreqFn freq, OutFn outLen) {
    const int SIDE = TreeMem<H, I, MAXN, IDB_, OVL>::SIDE;
    const int lane = threadIdx.x & 63;
    numSymbols = d4g_uniform(numSymbols);
    limit = d4g_uniform(limit);
    // does it fit?  used symbols (plus the dummy leaves that bring a queue of fewer than two up to two) and the total weight
    int used = 0;
    unsigned long long total = 0;
    for (int base = 0; base < numSymbols; base += 64) {
        const int i = base + lane;
        const unsigned f = i < numSymbols ? (unsigned)freq(i) : 0u;
        used += __popcll(d4g_ballot(f != 0));
        total += f;
    }
    total = (unsigned long long)wave_sum_i64((long long)total);
    if (used > 127 || total >= (1ull << 24) - 4) return d4g_build_tree_wave<NREG>(m, numSymbols, limit, freq, outLen);
    if (used > 64) return d4g_build_tree_wave_path(m, numSymbols, limit, freq, outLen);   // (ids stay below 255: at most 127 leaves + 126 merges)
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
    auto offer = [&](unsigned x) D4G_LAMBDA_INLINE {
        const int k = hs++;
        const unsigned xw = x >> 8;
        const unsigned pv = (unsigned)__shfl((int)hv, parentLane & 63);
        // the slots from the root down to k: k's ancestor set (a lane constant of lane k) and the root
        const unsigned pLo = (unsigned)d4g_readlane((int)ancLo, k) | 1u, pHi = (unsigned)d4g_readlane((int)ancHi, k);
        const bool onPath = ((pLo & bitLo) | (pHi & bitHi)) != 0;
        const bool pGreater = (lane > 0) & ((pv >> 8) > xw);     // the parent comes down into this slot
        const bool selfGreater = (hv >> 8) > xw;
        if (onPath) {
            if (pGreater) hv = pv;
            else if (lane == k || selfGreater) hv = x;
        }
    };
    auto poll = [&]() D4G_LAMBDA_INLINE -> unsigned {
        const unsigned out = (unsigned)d4g_readlane((int)hv, 0);
        const int s = --hs;
        if (s == 0) return out;
        const unsigned x = (unsigned)d4g_readlane((int)hv, s);
        const unsigned xw = x >> 8;
        const unsigned lv = (unsigned)__shfl((int)hv, leftLane & 63), rv = (unsigned)__shfl((int)hv, (leftLane + 1) & 63);
        const bool hasL = leftLane < s;
        const bool right = (leftLane + 1 < s) & ((lv >> 8) > (rv >> 8));
        const unsigned cv = right ? rv : lv;
        const unsigned long long Rm = d4g_ballot(right), Hm = d4g_ballot(hasL);
        // C: the slots that are the child their parent would hand up — slot 2P+1 when P hands up its left child, 2P+2 when
        // its right one (only P < 32 has children): the parents' bits spread to the even positions, on the scalar unit
        const unsigned long long C = (d4g_spread_bits((unsigned)(Hm & ~Rm)) << 1) | (d4g_spread_bits((unsigned)Rm) << 2);
        // on the root's hand-up path: this slot and all its ancestors (the root aside) are such children
        // (bitwise on purpose: a short-circuit here becomes an exec-mask region with a branch around it)
        const bool onPath = (lane < s) & (((unsigned)C & ancLo) == ancLo) & (((unsigned)(C >> 32) & ancHi) == ancHi);
        const bool pass = onPath & hasL & (xw > (cv >> 8));
        const bool stop = onPath & !pass & ((lane == 0) | (xw > (hv >> 8)));
        if (pass) hv = cv;
        else if (stop) hv = x;
        return out;
    };
    // leaves in symbol order
    int nl = 0;
    for (int base = 0; base < numSymbols; base += 64) {
        const int i = base + lane;
        const int fv = i < numSymbols ? (int)freq(i) : 0;
        unsigned long long um = d4g_ballot(fv != 0);
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
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    return 0;
}

// ---------------------------------------------------------------------------------------
// Dynamic-header RLE pairs, generated run by run.
// HuffmanTable.pack — B/huffman/HuffmanTable.java:70-159.  `emit(sym, run, value)` is called
// for every pair in output order (run = 0 for a literal code length).
// flags: bit0 ohh, 1 use8, 2 use7, 3 alt8, 4 noRep, 5 noZRep, 6 noZRep2, 7 noRepZeros
// ---------------------------------------------------------------------------------------
#define F_OHH 1
#define F_USE8 2
#define F_USE7 4
#define F_ALT8 8
#define F_NOREP 16
#define F_NOZREP 32
#define F_NOZREP2 64
#define F_NOREPZEROS 128
#define F_DEFAULT (F_OHH | F_USE8 | F_USE7)  // DeflateBlockHuffman.rewriteHeader() :480-482

template <typename Emit>
__device__ void d4g_pack_run(int v, int r, int flags, Emit emit) {
    if (v == 0) {
        if (!(flags & F_NOZREP2)) {
            while (r >= 138) { emit(18, 138, 0); r -= 138; }
            if (r >= 11) { emit(18, r, 0); r = 0; }
        }
        if (!(flags & F_NOZREP)) {
            while (r >= 10) { emit(17, 10, 0); r -= 10; }
            if (r >= 3) { emit(17, r, 0); r = 0; }
        }
    }
    if (!(flags & F_NOREP) && r > 0 && (!(flags & F_NOREPZEROS) || v != 0)) {
        emit(v, 0, v);
        r--;
        int j = 6;
        while (j >= 3) {
            if (flags & F_OHH) {
                if ((flags & F_USE8) && r == 8) {
                    emit(16, (flags & F_ALT8) ? 5 : 4, v);
                    emit(16, (flags & F_ALT8) ? 3 : 4, v);
                    r -= 8;
                    break;
                }
                if ((flags & F_USE7) && r == 7) {
                    emit(16, 4, v);
                    emit(16, 3, v);
                    r -= 7;
                    break;
                }
            }
            if (r - j >= 0) { emit(16, j, v); r -= j; }
            else j--;
        }
    }
    while (r > 0) { emit(v, 0, v); r--; }
}

// The same packing summarised in closed form (no loops): rep(sym, run, value, count) once per distinct
// repeat pair (sym 16 / 17 / 18) the run produces, with its multiplicity, and lit(count) for the plain
// code lengths `v` it leaves.  For callers that only count symbols or sum per-pair gains.
template <typename Rep, typename Lit>
__device__ void d4g_pack_kinds(int v, int r, int flags, Rep rep, Lit lit) {
    if (v == 0) {
        if (!(flags & F_NOZREP2)) {
            int a = r / 138;
            if (a) { rep(18, 138, 0, a); r -= a * 138; }
            if (r >= 11) { rep(18, r, 0, 1); r = 0; }
        }
        if (!(flags & F_NOZREP)) {
            int b = r / 10;
            if (b) { rep(17, 10, 0, b); r -= b * 10; }
            if (r >= 3) { rep(17, r, 0, 1); r = 0; }
        }
    }
    int lits = 0;
    if (!(flags & F_NOREP) && r > 0 && (!(flags & F_NOREPZEROS) || v != 0)) {
        lits = 1;
        r--;
        // the j == 6 phase takes sixes until it meets the 8 / 7 special cases (when enabled) or fewer than six are left
        const bool s8 = (flags & F_OHH) && (flags & F_USE8), s7 = (flags & F_OHH) && (flags & F_USE7);
        if (s8 && r >= 8 && r % 6 == 2) {
            int q = (r - 8) / 6;
            if (q) rep(16, 6, v, q);
            if (flags & F_ALT8) { rep(16, 5, v, 1); rep(16, 3, v, 1); }
            else rep(16, 4, v, 2);
            r = 0;
        } else if (s7 && r >= 7 && r % 6 == 1) {
            int q = (r - 7) / 6;
            if (q) rep(16, 6, v, q);
            rep(16, 4, v, 1);
            rep(16, 3, v, 1);
            r = 0;
        } else {
            int q = r / 6;
            if (q) rep(16, 6, v, q);
            r -= q * 6;
            if (r >= 3) { rep(16, r, v, 1); r = 0; }   // one pair of 5, 4 or 3
        }
    }
    lits += r;
    if (lits) lit(lits);
}

// Runs of the concatenated code lengths (lit then dist; runs may span the boundary — A.4).
// len(i) reads combined length i.  Calls body(value, runLength) per run.
template <typename LenFn, typename Body>
__device__ void d4g_for_runs(int n, LenFn len, Body body) {
    int last = len(0), run = 1;
    for (int i = 1; i <= n; i++) {
        if (i < n && len(i) == last) { run++; }
        else {
            body(last, run);
            if (i < n) { last = len(i); run = 1; }
        }
    }
}

// pair encode/decode (see d4g_types.h)
D4G_DEV uint16_t pair_encode(int sym, int run, int value) {
    if (sym < 16) return (uint16_t)sym;
    if (sym == 16) return (uint16_t)(16 | ((((run - 3) | (value << 2)) & 0xff) << 5));
    return (uint16_t)(sym | (run << 5));
}
D4G_DEV void pair_decode(uint16_t p, int& sym, int& run, int& value) {
    sym = p & 31;
    int x = (p >> 5) & 0xff;
    if (sym < 16) { run = 0; value = sym; }
    else if (sym == 16) { run = (x & 3) + 3; value = x >> 2; }
    else { run = x; value = 0; }
}
D4G_DEV int pair_extra_bits(int sym) { return sym == 16 ? 2 : sym == 17 ? 3 : 7; }  // getRLEPairSize :133-163

// replaceWithLiteralsIfSmaller on one RLE run pair (DeflateBlockHuffman.java:222-296 with
// distDec == null): every byte of the run is `value`, so the early-exit loop reduces to one
// comparison.  Returns bits saved (>= 0) or -1 when the pair stays a run.
template <typename ClFn>
D4G_DEV int pair_replace_gain(int sym, int run, int value, bool prune, ClFn cl) {
    int checkSize = cl(sym) + pair_extra_bits(sym);
    int b = cl(value);
    if (b < 1) return -1;
    int total = run * b;
    if (prune ? total > checkSize : total >= checkSize) return -1;
    return checkSize - total;
}

// the same with the two code lengths already at hand (lsym = bits of the repeat symbol, lval = bits of the repeated value)
D4G_DEV int pair_gain_bits(int sym, int run, int lsym, int lval, bool prune) {
    int checkSize = lsym + pair_extra_bits(sym);
    if (lval < 1) return -1;
    int total = run * lval;
    if (prune ? total > checkSize : total >= checkSize) return -1;
    return checkSize - total;
}

// removeDynHeaderTrailingZeroLenCodelens — DeflateBlockHuffman.java:335-364 (iterative form)
template <typename ClFn>
D4G_DEV int trim_codelens(int nCl, ClFn cl) {
    while (true) {
        int lastZero = -1, lastNonZero = nCl;
        for (int i = 0; i < nCl; i++) {
            if (cl(D4G_CL_ORDER[i]) == 0) lastZero = i;
            else lastNonZero = i;
        }
        if (lastZero > lastNonZero) nCl = lastZero;
        else return nCl;
    }
}
