// d4g_ops.h — workgroup-cooperative operations on one candidate state held in LDS, and
// the kernels that execute the candidate-search program.  See d4g_device.h for the pieces.
#pragma once
#include "d4g_device.h"

// Candidate states and masks are handed from workgroup to workgroup inside one launch.  They are
// written with agent-scope write-through stores and read with agent-scope (L1-bypassing) loads, so the
// hand-off needs no L2 write-back / L1 invalidate per task (MI355X_MICROARCH.md: "sc1 stores + drained
// flag", every load of the handed-off bytes an sc1 load).  Tokens and decoded bytes are read-only and
// keep using plain cached loads.
#ifdef D4G_HOSTSIM
D4G_DEV uint32_t ld_sc1(const uint32_t* p) { return *p; }
D4G_DEV uint64_t ld_sc1(const uint64_t* p) { return *p; }
D4G_DEV void st_sc1(uint32_t* p, uint32_t v) { *p = v; }
D4G_DEV void st_sc1(uint64_t* p, uint64_t v) { *p = v; }
#else
D4G_DEV uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV uint64_t ld_sc1(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV void st_sc1(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
D4G_DEV int ld_state_i32(const int32_t* p) { return (int)ld_sc1((const uint32_t*)p); }
// flags that hand work from one workgroup to another inside a launch
#ifdef D4G_HOSTSIM
D4G_DEV int d4g_flag_load(const int32_t* p) { return *p; }
D4G_DEV void d4g_flag_store(int32_t* p, int v) { *p = v; }
D4G_DEV void d4g_release_agent() {}
D4G_DEV void d4g_acquire_agent() {}
D4G_DEV void d4g_drain_stores() {}
D4G_DEV void d4g_sleep() {}
#else
D4G_DEV int d4g_flag_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV void d4g_flag_store(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV void d4g_release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
D4G_DEV void d4g_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
D4G_DEV void d4g_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#ifndef D4G_SPIN_SLEEP
#define D4G_SPIN_SLEEP 8
#endif
D4G_DEV void d4g_sleep() { __builtin_amdgcn_s_sleep(D4G_SPIN_SLEEP); }
#endif


#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
__device__ unsigned long long d4g_dbg_pass[4];  // computed replace passes: lookup cycles, loop cycles, count, records
__device__ unsigned long long d4g_dbg_hdr[4];   // profile builds: header rewrite sections (RLE, code-length tree, tail), count
// profile builds: a clock read that is not overtaken by (and does not overtake) outstanding LDS / memory operations
D4G_DEV long long d4g_clock_drained() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    long long t = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
#endif

// Header-search memo.  The 56 header candidates of a state depend only on its code lengths, and most of a block's
// ~200 header searches per run see value-identical lengths (a Huffman code hardly moves when a few tokens change):
// the first search of a length set owns the entry and publishes (header bits, winning candidate); the others reuse it.
#ifndef D4G_HSMEMO_SLOTS
#define D4G_HSMEMO_SLOTS 64
#endif
struct D4GHsMemo {
    unsigned long long tag;     // first 64-bit hash of the length set, 0 = free
    unsigned long long check;   // second, independent hash (written by the owner before `state` turns 2)
    int32_t state;              // 0 free/claimed, 2 published
    int32_t hdr;                // header bits of the best candidate
    int32_t lane;               // which candidate
    int32_t nLit;               // the key: nLit, n and the n combined code lengths (zero padded) — hits are confirmed against it
    int32_t n, pad;
    uint32_t key[(D4G_NLIT + D4G_NDIST) / 4];
};

// Huffman-rebuild memo.  recodeHuffman's result (both codes, the default header's RLE pairs and code-length code,
// the bit sizes) is a function of the symbol histogram alone, and many of a block's candidate states share a
// histogram.  Same protocol as the header-search memo: the first rebuild of a histogram owns the entry and
// publishes the rebuilt part of the state (bytes [64, 1056): lengths and pairs); later ones copy it.
#ifndef D4G_RCMEMO_SLOTS
#define D4G_RCMEMO_SLOTS 128
#endif
#define D4G_RCMEMO_WORDS 248
struct D4GRecodeMemo {
    unsigned long long tag, check;
    int32_t state, nLit, nDist, nCl, nPairs, err;
    long long litlenBits, hdrBits;
    uint32_t body[D4G_RCMEMO_WORDS];
    uint32_t key[D4G_HIST];     // the histogram the entry was built from: hits are confirmed against it
};
static_assert(offsetof(D4GState, litLen) == 64 && offsetof(D4GState, hist) == 64 + 4 * D4G_RCMEMO_WORDS, "memo body covers lengths + pairs");

// Token-pass memo.  What replaceBackrefsWithLiteralsIfSmaller does to a state — which records it expands, the
// histogram changes and the bits saved — depends only on the two codes, the incoming mask and the strict/lenient
// comparison; many candidate states of a block share all of those.  Entry = header, the 320 histogram deltas, the
// outgoing mask (memo-owned copy, so entries stay valid across rounds).  Cleared when a merge arena is re-used.
#ifndef D4G_PASSMEMO_SLOTS
#define D4G_PASSMEMO_SLOTS 256   // a block sees ~250 distinct (codes, mask) pairs per run; 128 slots overflowed
#endif
struct D4GPassMemo {
    unsigned long long tag, check;
    int32_t state, pad;
    long long saved;
    int32_t delta[D4G_HIST];
    uint32_t keyCodes[(D4G_NLIT + D4G_NDIST) / 4];   // the key: both codes, the comparison mode ...
    int32_t keyKind, pad2;
    // followed by maskWords u64 (the outgoing mask) and maskWords u64 (the incoming mask: the rest of the key)
};
#define D4G_PASSMEMO_HDR_WORDS ((int)(sizeof(D4GPassMemo) / 8))

// occupancy target of a kernel (caps its VGPR budget); the emulator build has no such notion
#ifdef D4G_HOSTSIM
#define D4G_WAVES_PER_SIMD(n)
#else
#define D4G_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, 8)))
#endif

struct D4GCtx {
    const uint2* tok;         // {token word, decoded-byte offset}
    uint4* refs;              // back-reference records {packed symbols/length, decoded-byte offset, first eight bytes}
    const uint32_t* tokRef;   // token -> back-reference record index
    uint32_t* binStat;        // per block and length symbol: static statistics of its records (d4g_types.h)
    uint64_t* binMask;        // per block and length symbol: which records carry it
    struct D4GHsMemo* hsMemo; // per block: results of the header searches already done, by code-length set
    struct D4GRecodeMemo* rcMemo;  // per block: Huffman rebuilds already done, by histogram
    uint64_t* passMemo;            // token-pass memo pool (D4GBlock.passMemo / passMemoStride)
    const uint8_t* U;
    const D4GBlock* blocks;
    D4GState* states;     // [numBlocks * slotsPerBlock]
    uint64_t* masks;      // mask pools of all blocks
    long long* keys;      // [numBlocks * nOps] candidate keys of the current round
    const D4GOp* ops;     // program in the reference's enumeration order (op id = tie-break order)
    const uint8_t* hdrFlags;  // [56] pack flags of header candidate k  (addOptimisedRecoded loop order)
    const uint8_t* hdrPrune;  // [56] its `prune` loop variable
    const int32_t* active;    // [nActive] block indices to run
    int32_t* errors;          // device error counter
    long long* opStats;       // optional per-op-kind cycle accounting (D4G_PROFILE_OPS builds), else null
    int32_t nActive;
    int32_t nOps;
    int32_t slotsPerBlock;
    int32_t masksPerBlock;
    int32_t tileGroups;       // blocks per launch tile / 8 (d4g_map_wg)
};

// LDS working set of one state-op workgroup
struct D4GLds {
    D4GState st;
    long long red[32];
    int misc[64];
    alignas(16) unsigned char treeLit[TreeMem<uint64_t, uint16_t, D4G_NLIT>::bytes(1)];
    alignas(16) unsigned char treeDist[TreeMem<uint64_t, uint16_t, D4G_NDIST>::bytes(1)];
    alignas(16) unsigned char treeCl[TreeMem<uint32_t, uint8_t, 20>::bytes(1)];
    uint32_t clFreq[20];
    uint16_t litCost[256];    // bits of each literal under the state's code; absent symbols cost D4G_NO_CODE
};

D4G_DEV uint64_t* mask_ptr(const D4GCtx& c, const D4GBlock& b, int slot) { return c.masks + b.maskBase + (long long)slot * b.maskWords; }
// mask bit of the back-reference record `ref` (absolute index) of a block whose records start at refStart
D4G_DEV bool d4g_ref_expanded(const uint64_t* mask, uint32_t ref, long long refStart) {
    uint32_t r = ref - (uint32_t)refStart;
    return (mask[r >> 6] >> (r & 63)) & 1;
}
// the bits of mask word w that stand for records (a mask may be longer than its records: whole cache lines, zero padded)
D4G_DEV uint64_t d4g_valid_bits(int w, int nRef) {
    const long long lo = (long long)w * 64;
    if (lo + 64 <= nRef) return ~0ull;
    if (lo >= nRef) return 0ull;
    return (1ull << (nRef & 63)) - 1;
}
D4G_DEV D4GState* state_ptr(const D4GCtx& c, int blockIdx_, int slot) { return c.states + ((long long)blockIdx_ * c.slotsPerBlock + slot); }

D4G_DEV void wg_copy_words(uint32_t* dst, const uint32_t* src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
D4G_DEV void wg_load_state(D4GState* S, const D4GState* g);
D4G_DEV void wg_store_state(D4GState* g, const D4GState* S);
D4G_DEV void wg_copy_mask(uint64_t* dst, const uint64_t* src, long long words);

// Bits of one back-reference under the state's codes — getLitLenSize, DeflateBlockHuffman.java:112-131
D4G_DEV int backref_cost(const D4GState* S, int len, int edge, int dist, int& lsym, int& dsym) {
    lsym = d4g_len2sym(len, edge);
    dsym = d4g_dist2sym(dist);
    return S->litLen[lsym] + d4g_lsym_ebits(lsym) + S->distLen[dsym] + d4g_dsym_ebits(dsym);
}

// Walk the decoded bytes [p, p+len) with aligned 32-bit loads (the next word is requested before the
// current one is consumed); fn(byte) returns false to stop early.  U is padded, so the aligned
// word holding the last byte is always readable.
template <typename Fn>
D4G_DEV void for_bytes(const uint8_t* p, int len, Fn fn) {
    uintptr_t a = (uintptr_t)p;
    const uint32_t* w = (const uint32_t*)(a & ~(uintptr_t)3);
    int skip = (int)(a & 3);
    uint32_t cur = *w++ >> (8 * skip);
    int have = 4 - skip, k = 0;
    while (true) {
        int n = have < len - k ? have : len - k;
        bool more = k + n < len;
        uint32_t nxt = 0;
        if (more) nxt = *w++;
        for (int j = 0; j < n; j++) {
            if (!fn((int)(cur & 0xff))) return;
            cur >>= 8;
        }
        k += n;
        if (!more) return;
        cur = nxt;
        have = 4;
    }
}
#ifndef D4G_LONG_TOKEN
#define D4G_LONG_TOKEN 32  // back-references longer than this are summed by the whole wave
#endif
#define D4G_NO_CODE 0x4000 // literal cost of a byte whose symbol has no code: above any real sum (258 * 15)

// litCost[v] for the state's current literal/length code (ends with a barrier)
D4G_DEV void wg_fill_lit_cost(D4GLds* L) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        int l = L->st.litLen[i];
        L->litCost[i] = (uint16_t)(l ? l : D4G_NO_CODE);
    }
    __syncthreads();
}

// Literal cost of a back-reference's decoded bytes, four bytes per step: the bytes are taken from the
// block's decoded data with aligned words (Uw = the stream's region of U, 16-byte aligned and padded).
struct D4GLitWalk {
    uint32_t widx;   // next word to request
    uint32_t cur, nxt;
    int sh;          // bit offset of the first byte in `cur`
    int rem;         // bytes left
    int total;
};
D4G_DEV void lw_start(D4GLitWalk& w, const uint32_t* Uw, uint32_t off, int len) {
    w.widx = off >> 2;
    w.sh = (int)(off & 3u) * 8;
    w.cur = Uw[w.widx];
    w.nxt = Uw[w.widx + 1];
    w.widx += 2;
    w.rem = len;
    w.total = 0;
}
// The same walk starting from a record: bytes 0-7 come with the record, the words after them are requested at the
// start, so the first three steps never wait for U.
struct D4GRecWalk {
    uint32_t b0, b1;   // bytes 0-3 and 4-7 (from the record)
    uint32_t widx;     // next word of U to request
    uint32_t cur, nxt;
    int sh, step, rem, total;
};
D4G_DEV void rw_start(D4GRecWalk& w, const uint32_t* Uw, const uint4& rec, int len) {
    w.b0 = rec.z;
    w.b1 = rec.w;
    w.sh = (int)(rec.y & 3u) * 8;
    w.widx = (rec.y + 8) >> 2;
    w.cur = 0u; w.nxt = 0u;
    if (len > 8) { w.cur = Uw[w.widx]; w.nxt = Uw[w.widx + 1]; }
    w.widx += 2;
    w.step = 0;
    w.rem = len;
    w.total = 0;
}
D4G_DEV void rw_step(D4GRecWalk& w, const uint32_t* Uw, const uint16_t* lc) {
    uint32_t x;
    if (w.step == 0) x = w.b0;
    else if (w.step == 1) x = w.b1;
    else {
        x = d4g_alignbit(w.nxt, w.cur, w.sh);
        w.cur = w.nxt;
        w.nxt = Uw[w.widx++];
    }
    w.step++;
    int l0 = lc[x & 255u], l1 = lc[(x >> 8) & 255u], l2 = lc[(x >> 16) & 255u], l3 = lc[x >> 24];
    int n = w.rem;
    w.total += l0 + (n > 1 ? l1 : 0) + (n > 2 ? l2 : 0) + (n > 3 ? l3 : 0);
    w.rem = n - 4;
}
D4G_DEV void lw_step(D4GLitWalk& w, const uint32_t* Uw, const uint16_t* lc) {
    uint32_t x = d4g_alignbit(w.nxt, w.cur, w.sh);
    w.cur = w.nxt;
    w.nxt = Uw[w.widx++];
    int l0 = lc[x & 255u], l1 = lc[(x >> 8) & 255u], l2 = lc[(x >> 16) & 255u], l3 = lc[x >> 24];
    int n = w.rem;
    w.total += l0 + (n > 1 ? l1 : 0) + (n > 2 ? l2 : 0) + (n > 3 ? l3 : 0);
    w.rem = n - 4;
}

// Token-pass memo, shared by the replace pass and the least-expensive pass.  `kind` separates the functions that are
// memoised (0 / 1: replace, lenient / strict; 2 / 3: least-expensive, modes 0 / 1).  Returns 0: compute without the
// memo, 1: this workgroup owns `entry` and must publish it, 2: `entry` holds the result.
__device__ __forceinline__ int wg_passmemo_lookup(D4GLds* L, const D4GCtx& c, const D4GBlock& b, const uint64_t* maskIn, int kind,
                                                  D4GPassMemo*& entry, unsigned long long& h2out) {
    D4GState* S = &L->st;
    entry = nullptr;
    if (!c.passMemo || b.passMemo < 0) return 0;
    auto mix1 = [](unsigned long long i, unsigned long long w) {
        unsigned long long x = (w + 1) * 0x9e3779b97f4a7c15ULL + (i + 1) * 0xbf58476d1ce4e5b9ULL;
        x ^= x >> 29; x *= 0x94d049bb133111ebULL; x ^= x >> 32;
        return x;
    };
    auto mix2 = [](unsigned long long i, unsigned long long w) {
        unsigned long long y = (w + 0x632be59bd9b4e019ULL) * ((i + 7) * 0xd6e8feb86659fd93ULL | 1ULL);
        y ^= y >> 31; y *= 0xff51afd7ed558ccdULL; y ^= y >> 33;
        return y;
    };
    unsigned long long a1 = 0, a2 = 0;
    const uint32_t* lw = (const uint32_t*)S->litLen;   // litLen[288] + distLen[32], contiguous
    for (int i = threadIdx.x; i < (D4G_NLIT + D4G_NDIST) / 4; i += blockDim.x) { a1 += mix1(i, lw[i]); a2 += mix2(i, lw[i]); }
    for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) {
        unsigned long long m = ld_sc1(maskIn + w);
        a1 += mix1(1000 + w, m);
        a2 += mix2(1000 + w, m);
    }
    long long s1, s2;
    wg_sum2_i64((long long)a1, (long long)a2, L->red, s1, s2);
    unsigned long long h1 = (unsigned long long)s1 + (unsigned long long)kind * 0x51ed270b35a3ULL;
    const unsigned long long h2 = (unsigned long long)s2 ^ ((unsigned long long)kind * 0x9b05688c2b3e6c1fULL);
    h2out = h2;
    if (h1 == 0) h1 = 1;
    uint64_t* pool = c.passMemo + b.passMemo;
    if (threadIdx.x == 0) {
        long long role = 0, idx = 0;
        for (int probe = 0; probe < 8; probe++) {
            int k = (int)(((h1 >> 7) + probe) % D4G_PASSMEMO_SLOTS);
            D4GPassMemo* e = (D4GPassMemo*)(pool + (long long)k * b.passMemoStride);
            unsigned long long t = atomicCAS(&e->tag, 0ULL, h1);
            if (t == 0) { role = 1; idx = k; break; }
            if (t == h1) {
                int st = 0;
#ifdef D4G_PROFILE_OPS
                long long w0 = clock64();
#endif
                for (int spin = 0; spin < (1 << 16); spin++) {
                    st = d4g_flag_load(&e->state);
                    if (st == 2) break;
                    d4g_sleep();
                }
#ifdef D4G_PROFILE_OPS
                if (c.opStats) { atomicAdd((unsigned long long*)&c.opStats[32], (unsigned long long)(clock64() - w0)); atomicAdd((unsigned long long*)&c.opStats[33], 1ULL); }
#endif
                if (st == 2 && ld_sc1((const uint64_t*)&e->check) == (uint64_t)h2) { role = 2; idx = k; }
                break;
            }
        }
#ifdef D4G_PROFILE_OPS
        if (c.opStats && role != 2) atomicAdd((unsigned long long*)&c.opStats[19], 1ULL);   // passes actually computed
#endif
        L->misc[62] = (int)role;
        L->misc[63] = (int)idx;
    }
    __syncthreads();
    int role = L->misc[62];
    entry = (D4GPassMemo*)(pool + (long long)L->misc[63] * b.passMemoStride);
    __syncthreads();
    // The hashes only find the entry; what decides is the key itself: codes, comparison mode and incoming mask.
    uint64_t* keyMask = (uint64_t*)entry + D4G_PASSMEMO_HDR_WORDS + b.maskWords;
    if (role == 1) {   // ours: the key goes in now, the result and the flag follow when the pass is done
        for (int i = threadIdx.x; i < (D4G_NLIT + D4G_NDIST) / 4; i += blockDim.x) st_sc1(&entry->keyCodes[i], lw[i]);
        for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) st_sc1(keyMask + w, ld_sc1(maskIn + w));
        if (threadIdx.x == 0) st_sc1((uint32_t*)&entry->keyKind, (uint32_t)kind);
    } else if (role == 2) {
        int bad = 0;
        for (int i = threadIdx.x; i < (D4G_NLIT + D4G_NDIST) / 4; i += blockDim.x) bad |= ld_sc1(&entry->keyCodes[i]) != lw[i];
        for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) bad |= ld_sc1(keyMask + w) != ld_sc1(maskIn + w);
        if (threadIdx.x == 0) bad |= ld_state_i32(&entry->keyKind) != kind;
        if (wg_max_i32(bad, L->red)) { role = 0; entry = nullptr; }   // same hashes, different key: compute without the memo
    }
    return role;
}
// a published entry applied to the state in LDS: outgoing mask, histogram deltas, bits saved
__device__ __forceinline__ void wg_passmemo_apply(D4GLds* L, const D4GBlock& b, const D4GPassMemo* e, uint64_t* maskOut) {
    D4GState* S = &L->st;
    const uint64_t* mm = (const uint64_t*)e + D4G_PASSMEMO_HDR_WORDS;
    for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) st_sc1(maskOut + w, ld_sc1(mm + w));
    for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) S->hist[i] += (uint32_t)ld_state_i32(&e->delta[i]);
    if (threadIdx.x == 0) {
        long long saved = (long long)ld_sc1((const uint64_t*)&e->saved);
        S->sizeBits -= saved;
        S->litlenBits -= saved;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// replaceBackrefsWithLiteralsIfSmaller — DeflateBlockHuffman.java:312-319 over :222-296.
// One lane per back-reference record, 64 per wave step; the new mask word is the wave ballot.
// A back-reference is expanded when all its bytes have codes and their bits sum below its own
// cost (`prune`: not above).  The sum is monotone, so the reference's byte-by-byte early exit is
// the same as comparing the whole sum — lanes stop as soon as the bound is reached.
// The histogram follows the token list (back-reference symbols out, literal bytes in).
// ---------------------------------------------------------------------------------------
#ifndef D4G_TOK_ILP
#define D4G_TOK_ILP 1  // records per lane per step (2 and 4 measured slower now that a record carries its first bytes)
#endif
__device__ __forceinline__ void wg_replace_backrefs(D4GLds* L, const D4GCtx& c, const D4GBlock& b, const uint64_t* maskIn, uint64_t* maskOut,
                                    bool prune) {
    D4GState* S = &L->st;
    const int K = D4G_TOK_ILP;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    long long pp0 = d4g_clock_drained(), pp1 = 0, pp2 = 0;
#endif
    // ---- memo lookup: hashes over both codes, the incoming mask and the comparison mode ----
    D4GPassMemo* mine = nullptr;
    unsigned long long h2 = 0;
    int* histBefore = (int*)L->treeLit;   // (the tree memory is idle during a token pass)
    {
        const int role = wg_passmemo_lookup(L, c, b, maskIn, prune ? 1 : 0, mine, h2);
        if (role == 2) { wg_passmemo_apply(L, b, mine, maskOut); return; }
        if (role == 1) {
            for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) histBefore[i] = (int)S->hist[i];
            __syncthreads();
        } else {
            mine = nullptr;
        }
    }
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    pp1 = d4g_clock_drained();
#endif
    wg_fill_lit_cost(L);
    const uint16_t* lc = L->litCost;
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    const uint32_t* Uw = (const uint32_t*)Ub;
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;  // a block's counts fit 31 bits
    int savedLane = 0;
    // stage A, software-pipelined: the record and the mask word of the NEXT step are requested (branch-free, from
    // always-valid addresses, so both loads go out back to back) before this step's are used
    uint4 nrv[K];
    uint64_t nmw[K];
    auto fetch = [&](int wbase) D4G_LAMBDA_INLINE {
#pragma unroll
        for (int j = 0; j < K; j++) {
            int w = wbase + j, r = w * 64 + lane;
            int wc = w < nWords ? w : 0, rc = r < nRef ? r : 0;
            nmw[j] = nWords ? ld_sc1(maskIn + wc) : 0;
            nrv[j] = nRef ? rf[rc] : make_uint4(0u, 0u, 0u, 0u);
            if (r >= nRef) nrv[j] = make_uint4(0u, 0u, 0u, 0u);
            if (w >= nWords) nmw[j] = 0;
        }
    };
    fetch(wave * K);
    static_assert(D4G_TOK_ILP == 1, "the straight-line first two steps below are written for one record per lane");
    for (int w0 = wave; w0 < nWords; w0 += nw) {
        const uint4 rec = nrv[0];
        const uint64_t mw = nmw[0];
        fetch(w0 + nw);
        int bit = (int)((mw >> lane) & 1);
        const uint32_t a = rec.x;
        const int len = ref_len(a);
        const bool act = len > 0 && !bit;
        // cost of the back-reference (getLitLenSize — :112-131) and the bound its literals must stay under
        int cost = 0, lim = 0, total = 0;
        if (act) {
            cost = S->litLen[ref_lsym(a)] + S->distLen[ref_dsym(a)] + ref_ebits(a);
            lim = cost + (prune ? 1 : 0);
            // bytes 0-3 came with the record (a back-reference has at least three bytes)
            const uint32_t x = rec.z;
            total = lc[x & 255u] + lc[(x >> 8) & 255u] + lc[(x >> 16) & 255u] + (len > 3 ? lc[x >> 24] : 0);
        }
        bool undec = act && total < lim && len > 4;
        if (d4g_ballot(undec)) {
            if (undec) {   // bytes 4-7, also from the record
                const uint32_t x = rec.w;
                const int n = len - 4;
                total += lc[x & 255u] + (n > 1 ? lc[(x >> 8) & 255u] : 0) + (n > 2 ? lc[(x >> 16) & 255u] : 0) + (n > 3 ? lc[x >> 24] : 0);
                undec = total < lim && len > 8;
            }
            if (d4g_ballot(undec)) {
                if (undec) {   // the rare long walk: the rest comes from U, four bytes per step
                    D4GLitWalk lw;
                    lw_start(lw, Uw, rec.y + 8, len - 8);
                    lw.total = total;
                    while (lw.rem > 0 && lw.total < lim) lw_step(lw, Uw, lc);
                    total = lw.total;
                }
            }
        }
        // apply: expanded when all its bytes have codes and they sum below the bound
        if (act && total < lim) {
            bit = 1;
            savedLane += cost - total;
            atomicSub(&S->hist[ref_lsym(a)], 1u);
            atomicSub(&S->hist[D4G_NLIT + ref_dsym(a)], 1u);
            for_bytes(Ub + rec.y, len, [&](int by) { atomicAdd(&S->hist[by], 1u); return true; });
        }
        uint64_t nm = d4g_ballot(bit);
        if (lane == 0) {
            st_sc1(maskOut + w0, nm);
            if (mine) st_sc1((uint64_t*)mine + D4G_PASSMEMO_HDR_WORDS + w0, nm);
        }
    }
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    pp2 = d4g_clock_drained();
#endif
    long long saved = wg_sum_i64((long long)savedLane, L->red);
    if (threadIdx.x == 0) { S->sizeBits -= saved; S->litlenBits -= saved; }
    __syncthreads();
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    if (threadIdx.x == 0) {
        atomicAdd(&d4g_dbg_pass[0], (unsigned long long)(pp1 - pp0));
        atomicAdd(&d4g_dbg_pass[1], (unsigned long long)(pp2 - pp1));
        atomicAdd(&d4g_dbg_pass[2], 1ULL);
        atomicAdd(&d4g_dbg_pass[3], (unsigned long long)nRef);
    }
#endif
    if (mine) {   // publish: histogram deltas, the outgoing mask, then the flag
        for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) st_sc1((uint32_t*)&mine->delta[i], (uint32_t)((int)S->hist[i] - histBefore[i]));
        if (threadIdx.x == 0) {
            st_sc1((uint64_t*)&mine->saved, (uint64_t)saved);
            st_sc1((uint64_t*)&mine->check, (uint64_t)h2);
        }
        d4g_drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) d4g_flag_store(&mine->state, 2);
    }
}

// ---------------------------------------------------------------------------------------
// Static per-bin statistics of a block's back-reference records (D4G_NBINS rows, d4g_types.h) and the per-bin
// record masks.  One workgroup per block, once per parse.
// ---------------------------------------------------------------------------------------
#define D4G_BINS_SPLIT 4   // workgroups per block (each sums its share in LDS, then adds it to the block's zeroed table)
__global__ void __launch_bounds__(256) k_block_bins(D4GCtx c, const int32_t* blockList) {
    __shared__ uint32_t T[D4G_NBINS * D4G_BINSTRIDE];
    const D4GBlock b = c.blocks[blockList[blockIdx.x / D4G_BINS_SPLIT]];
    const int part = blockIdx.x % D4G_BINS_SPLIT;
    if (b.binStat < 0) return;
    for (int i = threadIdx.x; i < D4G_NBINS * D4G_BINSTRIDE; i += blockDim.x) T[i] = 0;
    __syncthreads();
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    uint64_t* bm = c.binMask + b.binMask;
    const int lane = threadIdx.x & 63;
    const int nRef = (int)b.refCount, nWords = (int)b.maskWords;
    for (int r0 = part * (int)blockDim.x; r0 < nWords * 64; r0 += D4G_BINS_SPLIT * (int)blockDim.x) {
        int r = r0 + threadIdx.x;
        uint4 rv = r < nRef ? rf[r] : make_uint4(0u, 0u, 0u, 0u);
        int len = ref_len(rv.x);
        int bin = ref_lsym(rv.x) - 257;
        if (len > 0) {
            // the record's copy of its first eight decoded bytes (U is padded)
            const uint32_t* uw = (const uint32_t*)Ub + (rv.y >> 2);
            const int sh = (int)(rv.y & 3u) * 8;
            uint32_t w0 = uw[0], w1 = uw[1], w2 = uw[2];
            c.refs[b.refStart + r].z = d4g_alignbit(w1, w0, sh);
            c.refs[b.refStart + r].w = d4g_alignbit(w2, w1, sh);
            uint32_t* row = T + bin * D4G_BINSTRIDE;
            atomicAdd(&row[D4G_BIN_DIST + ref_dsym(rv.x)], 1u);
            atomicAdd(&row[D4G_BIN_COUNT], 1u);
            atomicAdd(&row[D4G_BIN_EBITS], (unsigned)ref_ebits(rv.x));
            for_bytes(Ub + rv.y, len, [&](int by) { atomicAdd(&row[by], 1u); return true; });
        }
        // the bin masks: one ballot per bin present in this word of records
        unsigned long long todo = d4g_ballot(len > 0);
        while (todo) {
            int src = __ffsll((long long)todo) - 1;
            int bsel = __shfl(bin, src);
            unsigned long long same = d4g_ballot(len > 0 && bin == bsel);
            if (lane == src) bm[(long long)bsel * nWords + (r >> 6)] = same;
            todo &= ~same;
        }
    }
    __syncthreads();
    uint32_t* g = c.binStat + b.binStat;   // zeroed by the host
    for (int i = threadIdx.x; i < D4G_NBINS * D4G_BINSTRIDE; i += blockDim.x)
        if (T[i]) atomicAdd(&g[i], T[i]);
}

// ---------------------------------------------------------------------------------------
// removeDistLitLeastExpensive — DeflateBlockHuffman.java:373-458.
// For every length symbol ("bin") the reference sums, over the bin's back-references, literal bits minus
// back-reference bits, counts them, and disallows bins holding a byte without a code; it then expands the whole
// bin that is cheapest (mode 0) or rarest (mode 1).  All of those sums are linear in the block's static bin
// statistics (k_block_bins), so they are taken from the table under the state's code and corrected by walking
// only the records that are already expanded (few); a mask with more expanded than unexpanded records is walked
// the other way round.  Expanding the chosen bin is an OR with the bin's record mask plus the same kind of
// table-minus-correction update of the histogram.
// ---------------------------------------------------------------------------------------
// literal bits of one record with the NO_CODE bytes counted in the high part (total = bits + D4G_NO_CODE * missing)
D4G_DEV int d4g_ref_lit_total(const uint32_t* Uw, const uint16_t* lc, uint32_t off, int len) {
    D4GLitWalk lw;
    lw_start(lw, Uw, off, len);
    while (lw.rem > 0) lw_step(lw, Uw, lc);
    return lw.total;
}
// Calls fn(record index, record) for the records whose bit is set in sel(w) (w = mask word index), 64 at a time:
// set bits are queued in LDS until a full wave's worth is there, so the lanes stay busy on sparse selections.
template <typename Sel, typename Fn>
D4G_DEV void wave_for_selected(int wave, int nw, int nWords, int nRef, const uint4* rf, uint32_t* queue /* [128] per wave */, Sel sel,
                               Fn fn) {
    const int lane = threadIdx.x & 63;
    int pending = 0;
    for (int w = wave; w < nWords; w += nw) {
        unsigned long long m = sel(w);
        m &= d4g_valid_bits(w, nRef);
        int n = __popcll(m);
        if ((m >> lane) & 1) queue[pending + __popcll(m & ((1ULL << lane) - 1))] = (uint32_t)(w * 64 + lane);
        pending += n;
        d4g_wave_sync();
        if (pending >= 64) {
            uint32_t r = queue[lane];
            fn((int)r, rf[r]);
            d4g_wave_sync();
            uint32_t carry = lane < pending - 64 ? queue[64 + lane] : 0u;
            d4g_wave_sync();
            if (lane < pending - 64) queue[lane] = carry;
            pending -= 64;
            d4g_wave_sync();
        }
    }
    if (lane < pending) {
        uint32_t r = queue[lane];
        fn((int)r, rf[r]);
    }
}

__device__ __forceinline__ void wg_least(D4GLds* L, const D4GCtx& c, const D4GBlock& b, const uint64_t* maskIn, uint64_t* maskOut, int mode) {
    D4GState* S = &L->st;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int* binSize = L->misc;        // [32]
    int* binFreq = L->misc + 32;   // [32]
    unsigned* flags = (unsigned*)L->red;  // [0] noAllow bits, [1] seen bits, [2] chosen bin + 1
    // scratch in the (idle) literal/length tree memory
    int* binZ = (int*)L->treeLit;                      // [32] bytes without a code, per bin
    int* delta = binZ + 32;                            // [320] histogram correction of the chosen bin
    uint32_t* queues = (uint32_t*)(delta + 320);       // [8][128]: at most eight waves walk records
    unsigned* npop = (unsigned*)(queues + 128 * 8);    // [1] expanded records in the mask
    static_assert(sizeof(L->treeLit) >= (32 + 320 + 128 * 8 + 1) * 4, "scratch does not fit the tree memory");
    const int nwq = nw < 8 ? nw : 8;
    __syncthreads();
    // the whole pass is a function of (codes, mask, mode): shared through the token-pass memo
    D4GPassMemo* mine = nullptr;
    unsigned long long memoH2 = 0;
    {
        const int role = wg_passmemo_lookup(L, c, b, maskIn, 2 + (mode & 1), mine, memoH2);
        if (role == 2) { wg_passmemo_apply(L, b, mine, maskOut); return; }
        if (role != 1) mine = nullptr;
    }
    uint64_t* memoMask = mine ? (uint64_t*)mine + D4G_PASSMEMO_HDR_WORDS : nullptr;
    if (mine) for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) st_sc1((uint32_t*)&mine->delta[i], 0u);
    if (threadIdx.x < 64) L->misc[threadIdx.x] = 0;
    if (threadIdx.x < 32) binZ[threadIdx.x] = 0;
    if (threadIdx.x < 4) flags[threadIdx.x] = 0;
    if (threadIdx.x == 0) *npop = 0;
    wg_fill_lit_cost(L);
    const uint16_t* lc = L->litCost;
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    const uint32_t* Uw = (const uint32_t*)Ub;
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;
    const uint32_t* stat = c.binStat + b.binStat;
    const uint64_t* bmask = c.binMask + b.binMask;
    {
        unsigned pc = 0;
        for (int w = threadIdx.x; w < nWords; w += blockDim.x) pc += (unsigned)__popcll(ld_sc1(maskIn + w));
        if (pc) atomicAdd(npop, pc);
    }
    __syncthreads();
    bool viaExpanded = 2 * (int)*npop <= nRef;   // walk the smaller side
#ifdef D4G_HOSTSIM
    if (getenv("D4G_SIM_LEAST_DIRECT")) viaExpanded = false;   // tests: force the other side
#endif
    if (S->type == D4G_DYNAMIC) {
        if (viaExpanded) {
            // table part: every record of the bin, under the state's code
            for (int bin = wave; bin < D4G_NBINS; bin += nw) {
                const uint32_t* row = stat + bin * D4G_BINSTRIDE;
                int lit = 0, z = 0, cost = 0;
                for (int i = lane; i < D4G_BINSTRIDE; i += 64) {
                    int cnt = (int)row[i];
                    if (i < 256) { int l = S->litLen[i]; lit += cnt * l; z += l ? 0 : cnt; }
                    else if (i < D4G_BIN_COUNT) cost += cnt * S->distLen[i - D4G_BIN_DIST];
                    else if (i == D4G_BIN_COUNT) cost += cnt * S->litLen[257 + bin];
                    else if (i == D4G_BIN_EBITS) cost += cnt;
                }
                lit = wave_sum_i32(lit); z = wave_sum_i32(z); cost = wave_sum_i32(cost);
                if (lane == 0) { binSize[bin] = lit - cost; binFreq[bin] = (int)row[D4G_BIN_COUNT]; binZ[bin] = z; }
            }
            __syncthreads();
            // correction: the expanded records do not count
            if (wave < nwq) wave_for_selected(wave, nwq, nWords, nRef, rf, queues + 128 * wave, [&](int w) { return ld_sc1(maskIn + w); },
                              [&](int, uint4 rv) {
                                  uint32_t a = rv.x;
                                  int bin = ref_lsym(a) - 257;
                                  int cost = S->litLen[ref_lsym(a)] + S->distLen[ref_dsym(a)] + ref_ebits(a);
                                  int total = d4g_ref_lit_total(Uw, lc, rv.y, ref_len(a));
                                  atomicSub(&binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                                  atomicSub(&binFreq[bin], 1);
                                  if (total >= D4G_NO_CODE) atomicSub(&binZ[bin], total >> 14);
                              });
        } else {
            if (wave < nwq) wave_for_selected(wave, nwq, nWords, nRef, rf, queues + 128 * wave, [&](int w) { return ~ld_sc1(maskIn + w); },
                              [&](int, uint4 rv) {
                                  uint32_t a = rv.x;
                                  int bin = ref_lsym(a) - 257;
                                  int cost = S->litLen[ref_lsym(a)] + S->distLen[ref_dsym(a)] + ref_ebits(a);
                                  int total = d4g_ref_lit_total(Uw, lc, rv.y, ref_len(a));
                                  atomicAdd(&binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                                  atomicAdd(&binFreq[bin], 1);
                                  if (total >= D4G_NO_CODE) atomicAdd(&binZ[bin], total >> 14);
                              });
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int rem = -1, remSize = 0, remFreq = 0;
        if (S->type == D4G_DYNAMIC) {
            for (int i = 0; i < D4G_NBINS; i++) {
                // seen: the bin has an unexpanded record; allowed: none of them holds a byte without a code
                if (binFreq[i] > 0 && binZ[i] == 0) {
                    bool doRem = mode == 1 ? binFreq[i] < remFreq : binSize[i] < remSize;
                    if (rem == -1 || doRem) { rem = i; remSize = binSize[i]; remFreq = binFreq[i]; }
                }
            }
            S->sizeBits += remSize;
            S->litlenBits += remSize;
        }
        flags[2] = (unsigned)(rem + 1);
        flags[3] = (unsigned)remSize;
    }
    for (int i = threadIdx.x; i < 320; i += blockDim.x) delta[i] = 0;
    __syncthreads();
    const int rem = (int)flags[2] - 1;
    // publish the pass for the states that share it: the flag goes up once every store has drained
    auto publish = [&]() D4G_LAMBDA_INLINE {
        if (!mine) return;
        if (threadIdx.x == 0) {
            st_sc1((uint64_t*)&mine->saved, (uint64_t)(-(long long)(int)flags[3]));
            st_sc1((uint64_t*)&mine->check, (uint64_t)memoH2);
        }
        d4g_drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) d4g_flag_store(&mine->state, 2);
    };
    if (rem < 0) {
        for (int w = threadIdx.x; w < nWords; w += blockDim.x) {
            uint64_t v = ld_sc1(maskIn + w);
            st_sc1(maskOut + w, v);
            if (mine) st_sc1(memoMask + w, v);
        }
        __syncthreads();
        publish();
        return;
    }
    // expand the bin: new mask = old | bin mask; the histogram loses the bin's unexpanded records' symbols and gains their bytes
    const uint64_t* bm = bmask + (long long)rem * nWords;
    for (int w = threadIdx.x; w < nWords; w += blockDim.x) {
        uint64_t v = ld_sc1(maskIn + w) | bm[w];
        st_sc1(maskOut + w, v);
        if (mine) st_sc1(memoMask + w, v);
    }
    // what the already expanded records of the bin contributed to the static row (they were moved earlier)
    if (wave < nwq) wave_for_selected(wave, nwq, nWords, nRef, rf, queues + 128 * wave, [&](int w) { return ld_sc1(maskIn + w) & bm[w]; },
                      [&](int, uint4 rv) {
                          uint32_t a = rv.x;
                          atomicAdd(&delta[D4G_BIN_DIST + ref_dsym(a)], 1);
                          atomicAdd(&delta[D4G_BIN_COUNT], 1);
                          for_bytes(Ub + rv.y, ref_len(a), [&](int by) { atomicAdd(&delta[by], 1); return true; });
                      });
    __syncthreads();
    {
        const uint32_t* row = stat + rem * D4G_BINSTRIDE;
        for (int i = threadIdx.x; i <= D4G_BIN_COUNT; i += blockDim.x) {
            int moved = (int)row[i] - delta[i];
            if (!moved) continue;
            // (byte values, distance symbols and the bin's length symbol are distinct histogram slots)
            if (i < 256) { atomicAdd(&S->hist[i], (unsigned)moved); if (mine) st_sc1((uint32_t*)&mine->delta[i], (uint32_t)moved); }
            else if (i < D4G_BIN_COUNT) {
                atomicSub(&S->hist[D4G_NLIT + i - D4G_BIN_DIST], (unsigned)moved);
                if (mine) st_sc1((uint32_t*)&mine->delta[D4G_NLIT + i - D4G_BIN_DIST], (uint32_t)(-moved));
            } else {
                atomicSub(&S->hist[257 + rem], (unsigned)moved);
                if (mine) st_sc1((uint32_t*)&mine->delta[257 + rem], (uint32_t)(-moved));
            }
        }
    }
    __syncthreads();
    publish();
}

// ---------------------------------------------------------------------------------------
// Header operations on the LDS state
// ---------------------------------------------------------------------------------------
// code-length-code tree from L->clFreq — Huffman.ofRLEPacked, B/huffman/Huffman.java:117-134
__device__ __forceinline__ void t0_build_cl_tree(D4GLds* L) {
    D4GState* S = &L->st;
    TreeMem<uint32_t, uint8_t, 20> tm;
    tm.carve(L->treeCl, 1);
    for (int i = 0; i < 19; i++) S->clLen[i] = 0;
    int err = d4g_build_tree(tm, 1, 0, 19, 7, [&](int i) { return (unsigned)L->clFreq[i]; },
                             [&](int v, int len) { S->clLen[v] = (uint8_t)len; });
    if (err) S->flags |= 0x100;
}

// the same by all lanes of wave 0
__device__ __forceinline__ void w0_build_cl_tree(D4GLds* L) {
    D4GState* S = &L->st;
#if !defined(D4G_HOSTSIM) || defined(D4G_SIM_WAVE_HEAP)
    TreeMem<uint32_t, uint8_t, 20> tm;
    tm.carve(L->treeCl, 1);
    const int lane = threadIdx.x & 63;
    if (lane < 19) S->clLen[lane] = 0;
    int err = d4g_build_tree_wave<1>(tm, 19, 7, [&](int i) { return (unsigned)L->clFreq[i]; },
                                     [&](int v, int len) { S->clLen[v] = (uint8_t)len; });
    if (err && lane == 0) S->flags |= 0x100;
#else
    if ((threadIdx.x & 63) == 0) t0_build_cl_tree(L);
#endif
    d4g_wave_sync();
}

// removeTrailingHeaderCodes — DeflateBlockHuffman.java:366-370 (thread 0)
__device__ __forceinline__ void t0_remove_trailing_header_codes(D4GState* S) {
    if (S->type != D4G_DYNAMIC) return;
    int n = trim_codelens(S->nCl, [&](int s) { return (int)S->clLen[s]; });
    long long saved = 3LL * (S->nCl - n);
    S->nCl = n;
    S->sizeBits -= saved;
    S->hdrBits -= saved;
}

// removeTrailingHeaderCodes by all lanes of wave 0: trim_codelens drops trailing zero lengths (in code-length
// order) one at a time while a non-zero one exists, i.e. it keeps everything up to the last non-zero length.
__device__ __forceinline__ void w0_remove_trailing_header_codes(D4GState* S) {
    if (S->type != D4G_DYNAMIC) return;
    const int lane = threadIdx.x & 63;
    const int nCl = S->nCl;
    bool nz = lane < nCl && lane < 19 && S->clLen[D4G_CL_ORDER[lane]] != 0;
    unsigned long long m = d4g_ballot(nz);
    int n = m ? 64 - __clzll((long long)m) : nCl;
    if (lane == 0) {
        long long saved = 3LL * (nCl - n);
        S->nCl = n;
        S->sizeBits -= saved;
        S->hdrBits -= saved;
    }
}

// rewriteHeader — DeflateBlockHuffman.java:484-577 (HuffmanTable.packCodeLengths :100-186 inside).  All threads.
// Wave 0 finds the runs of the concatenated code lengths with ballots; the lane at each run start packs
// that run (the pair count first, then the pairs at their prefix-summed position).  Thread 0 finishes with
// the code-length code and the header size.
// (forced inline: inside a kernel the compiler then knows L is LDS and drops the generic-pointer checks)
__device__ __forceinline__ void wg_rewrite_header(D4GLds* L, int flags) {
    D4GState* S = &L->st;
    __syncthreads();
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
    long long r0 = d4g_clock_drained(), r1 = 0, r2 = 0;
#endif
    if (S->type != D4G_DYNAMIC) return;
    if (threadIdx.x < 20) L->clFreq[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
#ifndef D4G_HOSTSIM
        __builtin_amdgcn_s_setprio(3);   // the workgroup's other waves wait for this one
#endif
        const int lane = threadIdx.x;
        const int nLit = S->nLit, n = nLit + S->nDist;
        auto len = [&](int i) { return i < nLit ? (int)S->litLen[i] : (int)S->distLen[i - nLit]; };
        constexpr int NCH = (D4G_NLIT + D4G_NDIST) / 64;  // 5 chunks of 64 code lengths
        unsigned long long sm[NCH];
        int v[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int i = ch * 64 + lane;
            v[ch] = i < n ? len(i) : -1;
            int pv = (i > 0 && i < n) ? len(i - 1) : -2;
            sm[ch] = d4g_ballot(i < n && v[ch] != pv);   // runs may span the literal/distance boundary — A.4
        }
        int base = 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int i = ch * 64 + lane;
            bool start = (sm[ch] >> lane) & 1;
            int cnt = 0, run = 0;
            if (start) {
                int nx = -1;
                unsigned long long m = lane == 63 ? 0ULL : (sm[ch] >> (lane + 1));
                if (m) nx = i + __ffsll((long long)m);
#pragma unroll
                for (int c2 = ch + 1; c2 < NCH; c2++)
                    if (nx < 0 && sm[c2]) nx = c2 * 64 + __ffsll((long long)sm[c2]) - 1;
                if (nx < 0) nx = n;
                run = nx - i;
                d4g_pack_run(v[ch], run, flags, [&](int, int, int) { cnt++; });
            }
            int incl = cnt;
            for (int d = 1; d < 64; d <<= 1) {
                int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            int off = base + incl - cnt;
            if (start)
                d4g_pack_run(v[ch], run, flags, [&](int sym, int r, int value) {
                    S->pairs[off++] = pair_encode(sym, r, value);
                    atomicAdd(&L->clFreq[sym], 1u);
                });
            base += __shfl(incl, 63);
        }
        if (lane == 0) S->nPairs = base;
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
        r1 = d4g_clock_drained();
#endif
        // code-length code (Huffman.ofRLEPacked, B/huffman/Huffman.java:117-134) and the header's size
        w0_build_cl_tree(L);
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
        r2 = d4g_clock_drained();
#endif
        int hbl = 0;
        if (lane < 19) hbl = (int)L->clFreq[lane] * (S->clLen[lane] + (lane >= 16 ? pair_extra_bits(lane) : 0));
        long long hb = 5 + 5 + 4 + 19 * 3 + wave_sum_i32(hbl);
        if (lane == 0) {
            S->sizeBits += hb - S->hdrBits;
            S->hdrBits = hb;
            S->nCl = 19;
        }
        d4g_wave_sync();
        w0_remove_trailing_header_codes(S);
#ifndef D4G_HOSTSIM
        __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
#if defined(D4G_PROFILE_OPS) && !defined(D4G_HOSTSIM)
        if (lane == 0) {
            atomicAdd(&d4g_dbg_hdr[0], (unsigned long long)(r1 - r0));
            atomicAdd(&d4g_dbg_hdr[1], (unsigned long long)(r2 - r1));
            atomicAdd(&d4g_dbg_hdr[2], (unsigned long long)(d4g_clock_drained() - r2));
            atomicAdd(&d4g_dbg_hdr[3], 1ULL);
        }
#endif
    }
    __syncthreads();
}

// replaceRLERunsWithLiteralsIfSmaller — DeflateBlockHuffman.java:321-332.  All threads.
__device__ __forceinline__ void wg_replace_rle_runs(D4GLds* L, bool prune) {
    D4GState* S = &L->st;
    long long saved = 0;
    __syncthreads();
    if (S->type == D4G_DYNAMIC) {
        for (int i = threadIdx.x; i < S->nPairs; i += blockDim.x) {
            uint16_t p = S->pairs[i];
            if ((p & 31) >= 16 && !(p & D4G_PAIR_EXPANDED)) {
                int sym, run, value;
                pair_decode(p, sym, run, value);
                int g = pair_replace_gain(sym, run, value, prune, [&](int s) { return (int)S->clLen[s]; });
                if (g >= 0) { S->pairs[i] = p | D4G_PAIR_EXPANDED; saved += g; }
            }
        }
    }
    saved = wg_sum_i64(saved, L->red);
    if (threadIdx.x == 0) { S->sizeBits -= saved; S->hdrBits -= saved; }
    __syncthreads();
}

// recodeHeader — DeflateBlockHuffman.java:579-629 (numCodelenLens deliberately not reset).  Wave 0: the pairs are
// counted and priced 64 at a time, the code-length tree is built by the wave.
__device__ __forceinline__ void wg_recode_header(D4GLds* L) {
    D4GState* S = &L->st;
    __syncthreads();
    if (S->type != D4G_DYNAMIC) return;
    if (threadIdx.x < 20) L->clFreq[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int np = S->nPairs;
        for (int i = lane; i < np; i += 64) {
            int sym, run, value;
            uint16_t p = S->pairs[i];
            pair_decode(p, sym, run, value);
            if (p & D4G_PAIR_EXPANDED) atomicAdd(&L->clFreq[value], (unsigned)run);
            else atomicAdd(&L->clFreq[sym], 1u);
        }
        d4g_wave_sync();
        w0_build_cl_tree(L);
        // trim from the current count (trim_codelens: keep everything up to the last non-zero length)
        const int nCl0 = S->nCl;
        bool nz = lane < nCl0 && lane < 19 && S->clLen[D4G_CL_ORDER[lane]] != 0;
        unsigned long long m = d4g_ballot(nz);
        const int nCl = m ? 64 - __clzll((long long)m) : nCl0;
        int hbl = 0;
        for (int i = lane; i < np; i += 64) {
            int sym, run, value;
            uint16_t p = S->pairs[i];
            pair_decode(p, sym, run, value);
            if (p & D4G_PAIR_EXPANDED) hbl += run * S->clLen[value];
            else hbl += S->clLen[sym] + (sym >= 16 ? pair_extra_bits(sym) : 0);
        }
        long long hb = 5 + 5 + 4 + 3LL * nCl + wave_sum_i32(hbl);
        if (lane == 0) {
            S->nCl = nCl;
            S->sizeBits += hb - S->hdrBits;
            S->hdrBits = hb;
        }
    }
    __syncthreads();
}

// optimiseHeader — DeflateBlockHuffman.java:471-476
__device__ __forceinline__ void wg_optimise_header(D4GLds* L) {
    __syncthreads();
    if (threadIdx.x < 64) w0_remove_trailing_header_codes(&L->st);
    __syncthreads();
    wg_replace_rle_runs(L, false);
}

// Σ token bits from the histogram — recodeToHuffmanInternal, DeflateBlockHuffman.java:759-770
__device__ __forceinline__ void wg_litlen_bits_from_hist(D4GLds* L) {
    D4GState* S = &L->st;
    long long v = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) {
        unsigned h = S->hist[i];
        if (h) {
            if (i < D4G_NLIT) v += (long long)h * (S->litLen[i] + (i >= 257 ? d4g_lsym_ebits(i) : 0));
            else v += (long long)h * (S->distLen[i - D4G_NLIT] + d4g_dsym_ebits(i - D4G_NLIT));
        }
    }
    v = wg_sum_i64(v, L->red);
    if (threadIdx.x == 0) {
        S->sizeBits -= S->litlenBits;
        S->litlenBits = v;
        S->sizeBits += v;
    }
    __syncthreads();
}

// recodeHuffman — DeflateBlockHuffman.java:670-743 + recodeToHuffman :745-757
// (forced inline: inside a kernel the compiler then knows L is LDS and drops the generic-pointer checks)
__device__ __forceinline__ void wg_recode_huffman(D4GLds* L, D4GRecodeMemo* memoTab, long long* prof = nullptr) {
    D4GState* S = &L->st;
    __syncthreads();
    // ---- memo lookup by two position-keyed hashes of the histogram ----
    D4GRecodeMemo* mine = nullptr;
    unsigned long long h2 = 0;
    if (memoTab) {
        unsigned long long a1 = 0, a2 = 0;
        for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) {
            unsigned long long w = (unsigned long long)S->hist[i] + 1;
            unsigned long long x = w * 0x9e3779b97f4a7c15ULL + (unsigned long long)(i + 1) * 0xbf58476d1ce4e5b9ULL;
            x ^= x >> 29; x *= 0x94d049bb133111ebULL; x ^= x >> 32;
            unsigned long long y = (w + 0x632be59bd9b4e019ULL) * ((unsigned long long)(i + 7) * 0xd6e8feb86659fd93ULL | 1ULL);
            y ^= y >> 31; y *= 0xff51afd7ed558ccdULL; y ^= y >> 33;
            a1 += x;
            a2 += y;
        }
        long long s1, s2;
        wg_sum2_i64((long long)a1, (long long)a2, L->red, s1, s2);
        unsigned long long h1 = (unsigned long long)s1;
        h2 = (unsigned long long)s2;
        if (h1 == 0) h1 = 1;
        if (threadIdx.x == 0) {
            long long role = 0, idx = 0;   // 0 compute without the memo, 1 owner, 2 hit
            for (int probe = 0; probe < 8; probe++) {
                int k = (int)(((h1 >> 7) + probe) % D4G_RCMEMO_SLOTS);
                D4GRecodeMemo* e = memoTab + k;
                unsigned long long t = atomicCAS(&e->tag, 0ULL, h1);
                if (t == 0) { role = 1; idx = k; break; }
                if (t == h1) {
                    int st = 0;
                    for (int spin = 0; spin < (1 << 16); spin++) {
                        st = d4g_flag_load(&e->state);
                        if (st == 2) break;
                        d4g_sleep();
                    }
                    if (st == 2 && ld_sc1((const uint64_t*)&e->check) == (uint64_t)h2) { role = 2; idx = k; }
                    break;
                }
            }
            L->misc[62] = (int)role;
            L->misc[63] = (int)idx;
        }
        __syncthreads();
        int role = L->misc[62];
        D4GRecodeMemo* e = memoTab + L->misc[63];
        __syncthreads();
        // the hashes only find the entry: a hit is confirmed against the histogram it was built from
        if (role == 1) {
            for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) st_sc1(&e->key[i], S->hist[i]);
        } else if (role == 2) {
            int bad = 0;
            for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) bad |= ld_sc1(&e->key[i]) != S->hist[i];
            if (wg_max_i32(bad, L->red)) role = 0;
        }
        if (role == 2) {
            for (int i = threadIdx.x; i < D4G_RCMEMO_WORDS; i += blockDim.x) ((uint32_t*)S)[16 + i] = ld_sc1(&e->body[i]);
            if (threadIdx.x == 0) {
                S->type = D4G_DYNAMIC;
                S->nLit = ld_state_i32(&e->nLit); S->nDist = ld_state_i32(&e->nDist);
                S->nCl = ld_state_i32(&e->nCl); S->nPairs = ld_state_i32(&e->nPairs);
                S->litlenBits = (long long)ld_sc1((const uint64_t*)&e->litlenBits);
                S->hdrBits = (long long)ld_sc1((const uint64_t*)&e->hdrBits);
                S->sizeBits = S->litlenBits + S->hdrBits;
                if (ld_state_i32(&e->err)) S->flags |= 0x100;
            }
            __syncthreads();
            return;
        }
        if (role == 1) mine = e;
    }
#ifdef D4G_PROFILE_OPS
    long long pt0 = clock64(), pt1 = 0, pt2 = 0;
#endif
    int ml = 0, md = 0;
    for (int i = threadIdx.x; i < 286; i += blockDim.x)
        if (S->hist[i]) ml = i + 1 > ml ? i + 1 : ml;
    for (int i = threadIdx.x; i < 30; i += blockDim.x)
        if (S->hist[D4G_NLIT + i]) md = i + 1 > md ? i + 1 : md;
    int lastLit = wg_max_i32(ml, L->red);
    int lastDist = wg_max_i32(md, L->red);
    for (int i = threadIdx.x; i < D4G_NLIT; i += blockDim.x) S->litLen[i] = 0;
    for (int i = threadIdx.x; i < D4G_NDIST; i += blockDim.x) S->distLen[i] = 0;
    __syncthreads();
#if !defined(D4G_HOSTSIM) || defined(D4G_SIM_WAVE_HEAP)
    // wave 0 builds the literal/length tree while wave 1 (when there is one) builds the distance tree
    const int wv = d4g_uniform((int)(threadIdx.x >> 6)), ln = threadIdx.x & 63;
    if (wv == 0) {
        TreeMem<uint64_t, uint16_t, D4G_NLIT> tm;
        tm.carve(L->treeLit, 1);
        int err = d4g_build_tree_wave<(D4G_NLIT + 63) / 64>(tm, lastLit, 15, [&](int i) { return S->hist[i]; },
                                                            [&](int v, int len) { S->litLen[v] = (uint8_t)len; });
        if (ln == 0) {
            if (err) atomicOr((unsigned*)&S->flags, 0x100u);
            S->nLit = lastLit;
        }
#ifdef D4G_PROFILE_OPS
        pt1 = clock64();
#endif
    }
    if (wv == (blockDim.x > 64 ? 1 : 0)) {
        bool used = ln < lastDist && S->hist[D4G_NLIT + ln] != 0;
        int nz = __popcll(d4g_ballot(used));
        if (lastDist == 0) {  // handleZero: new HuffmanTable(1)
            if (ln == 0) S->nDist = 1;
        } else if (nz <= 1) {  // handleOne: one used distance code, length 1
            if (ln == 0) { S->nDist = lastDist; S->distLen[lastDist - 1] = 1; }
        } else {
            TreeMem<uint64_t, uint16_t, D4G_NDIST> tm;
            tm.carve(L->treeDist, 1);
            int err = d4g_build_tree_wave<1>(tm, lastDist, 15, [&](int i) { return S->hist[D4G_NLIT + i]; },
                                             [&](int v, int len) { S->distLen[v] = (uint8_t)len; });
            if (ln == 0) {
                if (err) atomicOr((unsigned*)&S->flags, 0x100u);
                S->nDist = lastDist;
            }
        }
    }
#else
    if (threadIdx.x == 0) {
        TreeMem<uint64_t, uint16_t, D4G_NLIT> tm;
        tm.carve(L->treeLit, 1);
        int err = d4g_build_tree(tm, 1, 0, lastLit, 15, [&](int i) { return S->hist[i]; },
                                 [&](int v, int len) { S->litLen[v] = (uint8_t)len; });
        if (err) atomicOr((unsigned*)&S->flags, 0x100u);
        S->nLit = lastLit;
#ifdef D4G_PROFILE_OPS
        pt1 = clock64();
#endif
    }
    if (threadIdx.x == 64 || (blockDim.x <= 64 && threadIdx.x == 0)) {
        int nz = 0;
        for (int i = 0; i < lastDist; i++) nz += S->hist[D4G_NLIT + i] != 0;
        if (lastDist == 0) {  // handleZero: new HuffmanTable(1)
            S->nDist = 1;
        } else if (nz <= 1) {  // handleOne: one used distance code, length 1
            S->nDist = lastDist;
            S->distLen[lastDist - 1] = 1;
        } else {
            TreeMem<uint64_t, uint16_t, D4G_NDIST> tm;
            tm.carve(L->treeDist, 1);
            int err = d4g_build_tree(tm, 1, 0, lastDist, 15, [&](int i) { return S->hist[D4G_NLIT + i]; },
                                     [&](int v, int len) { S->distLen[v] = (uint8_t)len; });
            if (err) atomicOr((unsigned*)&S->flags, 0x100u);
            S->nDist = lastDist;
        }
    }
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S->type != D4G_DYNAMIC) { S->type = D4G_DYNAMIC; S->hdrBits = 0; S->nPairs = 0; S->nCl = 0; }
    }
#ifdef D4G_PROFILE_OPS
    pt2 = clock64();
#endif
    wg_litlen_bits_from_hist(L);
    wg_rewrite_header(L, F_DEFAULT);
    if (mine) {   // publish the rebuilt part of the state, then the flag
        for (int i = threadIdx.x; i < D4G_RCMEMO_WORDS; i += blockDim.x) st_sc1(&mine->body[i], ((const uint32_t*)S)[16 + i]);
        if (threadIdx.x == 0) {
            st_sc1((uint32_t*)&mine->nLit, (uint32_t)S->nLit); st_sc1((uint32_t*)&mine->nDist, (uint32_t)S->nDist);
            st_sc1((uint32_t*)&mine->nCl, (uint32_t)S->nCl); st_sc1((uint32_t*)&mine->nPairs, (uint32_t)S->nPairs);
            st_sc1((uint32_t*)&mine->err, (uint32_t)((S->flags & 0x100) ? 1 : 0));
            st_sc1((uint64_t*)&mine->litlenBits, (uint64_t)S->litlenBits);
            st_sc1((uint64_t*)&mine->hdrBits, (uint64_t)S->hdrBits);
            st_sc1((uint64_t*)&mine->check, (uint64_t)h2);
        }
        d4g_drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) d4g_flag_store(&mine->state, 2);
    }
#ifdef D4G_PROFILE_OPS
    if (threadIdx.x == 0 && prof) {
        atomicAdd((unsigned long long*)&prof[24], (unsigned long long)(pt1 - pt0));          // literal/length tree (thread 0)
        atomicAdd((unsigned long long*)&prof[25], (unsigned long long)(pt2 - pt1));          // wait for the distance tree
        atomicAdd((unsigned long long*)&prof[26], (unsigned long long)(clock64() - pt2));    // sizes + header rewrite
        atomicAdd((unsigned long long*)&prof[27], 1ULL);
    }
#endif
}

// recodeToFixedHuffman — DeflateBlockHuffman.java:637-653
__device__ __forceinline__ void wg_recode_to_fixed(D4GLds* L) {
    D4GState* S = &L->st;
    __syncthreads();
    if (S->type == D4G_FIXED) return;
    for (int i = threadIdx.x; i < D4G_NLIT; i += blockDim.x)
        S->litLen[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 286 ? 8 : 0;  // HuffmanTable.LIT :166-209 (286 codes)
    for (int i = threadIdx.x; i < D4G_NDIST; i += blockDim.x) S->distLen[i] = i < 30 ? 5 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        S->sizeBits -= S->hdrBits;
        S->type = D4G_FIXED;
        S->hdrBits = 0;
        S->nLit = S->nDist = S->nCl = S->nPairs = 0;
    }
    wg_litlen_bits_from_hist(L);
}

// ---------------------------------------------------------------------------------------
// Dependency-driven execution.  Instead of one launch per program level, persistent workgroups pull
// (block, op) tasks from a queue ordered by level; an op waits for its source slot's ready flag and
// publishes its own.  Tasks are pulled in queue order, so a task's producer was always pulled earlier
// by a workgroup that is resident and running: waiting never deadlocks, whatever subset of the grid
// is resident.  Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility): producer stores ->
// every wave s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0 agent-scope release -> relaxed agent
// flag store; consumer lane 0 relaxed agent poll -> agent-scope acquire -> barrier -> plain loads.
// ---------------------------------------------------------------------------------------
struct D4GQueue {
    const int32_t* opFlat;   // op ids in level order
    int32_t nOpsFlat;
    unsigned* head;          // [8] next task index of each XCD's queue
    int32_t* ready;          // [numBlocks * slotsPerBlock]: epoch in which the slot was last produced
    int32_t epoch;
    int32_t xoff[9];         // the active list is grouped by XCD: group x = active[xoff[x], xoff[x+1])
    long long spinLimit;     // polls before a wait gives up (D4G_SPIN_LIMIT; < 0: every wait gives up at once — tests)
};

// Pull the next task: first from the queue of this workgroup's XCD (workgroup ids equal mod 8 share an XCD
// and its L2, so a deflate block's tokens and decoded bytes stay in one L2), then steal from the others.
// Returns false when every queue is empty.  Thread 0 only.
D4G_DEV bool d4g_pull_task(const D4GQueue& q, int& cursor, int& opIdx, int& actIdx) {
    for (; cursor < 8; cursor++) {
        int x = (int)((blockIdx.x + cursor) & 7);
        int nA = q.xoff[x + 1] - q.xoff[x];
        long long nTasks = (long long)q.nOpsFlat * nA;
        if (nTasks == 0) continue;
        long long t = atomicAdd(q.head + x, 1u);
        if (t < nTasks) {
            opIdx = (int)(t / nA);
            actIdx = q.xoff[x] + (int)(t % nA);
            return true;
        }
    }
    return false;
}


D4G_DEV void wg_load_state(D4GState* S, const D4GState* g) {
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(D4GState) / 4); i += blockDim.x) ((uint32_t*)S)[i] = ld_sc1((const uint32_t*)g + i);
    __syncthreads();
}
D4G_DEV void wg_store_state(D4GState* g, const D4GState* S) {
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(D4GState) / 4); i += blockDim.x) st_sc1((uint32_t*)g + i, ((const uint32_t*)S)[i]);
    __syncthreads();
}
D4G_DEV void wg_copy_mask(uint64_t* dst, const uint64_t* src, long long words) {
    for (long long i = threadIdx.x; i < words; i += blockDim.x) st_sc1(dst + i, ld_sc1(src + i));
    __syncthreads();
}

// All threads call.  Returns false if the slot never became ready: the producer runs in another workgroup (for header
// searches: another kernel), and HIP promises no co-residency — under a profiler that serialises kernels, or with the
// device full of someone else's long kernel, a wait can outlast any bound.  That is not an error: it is counted in
// c.errors[1], k_select then leaves the block untouched and the host runs the round with the level executor.
__device__ bool wg_wait_slot(const D4GCtx& c, const D4GQueue& q, int blk, int slot, int* lds) {
    if (slot == 0) return true;  // the block's current state was written by an earlier kernel
    __syncthreads();
    if (threadIdx.x == 0) {
        const int32_t* f = q.ready + (long long)blk * c.slotsPerBlock + slot;
        int ok = 0;
        // bounded: ~1 s in total by default, and every waiter gives up as soon as any other has
        for (long long spin = 0; spin < q.spinLimit; spin++) {
            if (d4g_flag_load(f) == q.epoch) { ok = 1; break; }
            if ((spin & 63) == 63 && d4g_flag_load(c.errors + 1) != 0) break;
            d4g_sleep();
        }
        if (!ok && q.spinLimit >= 0 && d4g_flag_load(f) == q.epoch) ok = 1;
        if (!ok) atomicAdd(c.errors + 1, 1);
        *lds = ok;
    }
    __syncthreads();
    return *lds != 0;
}
__device__ void wg_publish_slot(const D4GCtx& c, const D4GQueue& q, int blk, int slot) {
    d4g_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) d4g_flag_store(q.ready + (long long)blk * c.slotsPerBlock + slot, q.epoch);
}

// ---------------------------------------------------------------------------------------
// State-op executor: one workgroup = one op of the program on one block.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void d4g_exec_state_op(D4GLds* L, const D4GCtx& c, int blk, int opId) {
    const D4GOp op = c.ops[opId];
#ifdef D4G_PROFILE_OPS
    long long tStart = clock64();
#endif
    const D4GBlock b = c.blocks[blk];
    D4GState* S = &L->st;
    const D4GState* src = state_ptr(c, blk, op.src);
    long long* keyp = c.keys + (long long)blk * c.nOps + opId;
    if (op.kind == OP_CAND) {
        if (threadIdx.x == 0) {
            long long sz = (long long)ld_sc1((const uint64_t*)&src->sizeBits);
            *keyp = ld_state_i32(&src->valid) ? D4G_MAKE_KEY(sz, (long long)opId * 64) : D4G_KEY_NONE;
        }
        return;
    }
    D4GState* dst = state_ptr(c, blk, op.dst);
    if (!ld_state_i32(&src->valid)) {  // the reference never builds this candidate (null / skipped branch)
        if (threadIdx.x == 0) { st_sc1((uint32_t*)&dst->valid, 0u); *keyp = D4G_KEY_NONE; }
        return;
    }
    wg_load_state(S, src);
    const uint64_t* maskIn = mask_ptr(c, b, S->maskSlot);
    switch (op.kind) {
    case OP_OPT: {
        long long orig = S->sizeBits;
        uint64_t* mo = mask_ptr(c, b, op.maskSlot);
#ifdef D4G_PROFILE_OPS
        long long tA = clock64();
#endif
        wg_replace_backrefs(L, c, b, maskIn, mo, false);
#ifdef D4G_PROFILE_OPS
        long long tB = clock64();
#endif
        wg_optimise_header(L);
#ifdef D4G_PROFILE_OPS
        if (threadIdx.x == 0 && c.opStats) {
            atomicAdd((unsigned long long*)&c.opStats[20], (unsigned long long)(tA - tStart));
            atomicAdd((unsigned long long*)&c.opStats[21], (unsigned long long)(tB - tA));
            atomicAdd((unsigned long long*)&c.opStats[22], (unsigned long long)(clock64() - tB));
            atomicAdd((unsigned long long*)&c.opStats[23], 1ULL);
        }
#endif
        if (threadIdx.x == 0) {
            S->maskSlot = op.maskSlot;
            S->valid = (op.arg & 1) ? (orig - S->sizeBits > 0) : 1;
        }
        break;
    }
    case OP_RECODE: {
        if (op.arg & 1) {  // recodeHuffmanLessMatches — :655-658
            uint64_t* mo = mask_ptr(c, b, op.maskSlot);
            wg_replace_backrefs(L, c, b, maskIn, mo, true);
            if (threadIdx.x == 0) S->maskSlot = op.maskSlot;
        }
        wg_recode_huffman(L, c.rcMemo ? c.rcMemo + (long long)blk * D4G_RCMEMO_SLOTS : nullptr, c.opStats);
        break;
    }
    case OP_RECODE_FULL: {  // recodedHuffmanFull — DeflateStream.java:212-229
        long long prevSize = S->sizeBits;
        int curSlot = op.src;
        int same = 1;
        for (int it = 0;; it++) {
            int ms = op.scratchMask + (it & 1);
            const uint64_t* mi = mask_ptr(c, b, S->maskSlot);
            wg_replace_backrefs(L, c, b, mi, mask_ptr(c, b, ms), true);
            wg_recode_huffman(L, c.rcMemo ? c.rcMemo + (long long)blk * D4G_RCMEMO_SLOTS : nullptr, c.opStats);
            __syncthreads();
            long long thisSize = S->sizeBits;
            __syncthreads();
            if (thisSize >= prevSize) break;
            if (threadIdx.x == 0) S->maskSlot = ms;
            curSlot = op.scratch + (it & 1);
            wg_store_state(state_ptr(c, blk, curSlot), S);
            prevSize = thisSize;
            same = 0;
        }
        wg_load_state(S, state_ptr(c, blk, curSlot));
        wg_copy_mask(mask_ptr(c, b, op.maskSlot), mask_ptr(c, b, S->maskSlot), b.maskWords);
        if (threadIdx.x == 0) { S->maskSlot = op.maskSlot; S->valid = !same; }
        break;
    }
    case OP_LEAST: {
        uint64_t* mo = mask_ptr(c, b, op.maskSlot);
        wg_least(L, c, b, maskIn, mo, op.arg);
        if (threadIdx.x == 0) S->maskSlot = op.maskSlot;
        break;
    }
    case OP_POST:
        wg_recode_header(L);
        break;
    case OP_PRUNEHDR:
        wg_replace_rle_runs(L, true);
        wg_recode_header(L);
        break;
    case OP_TOFIXED_OPT: {
        wg_recode_to_fixed(L);
        uint64_t* mo = mask_ptr(c, b, op.maskSlot);
        wg_replace_backrefs(L, c, b, maskIn, mo, false);
        wg_optimise_header(L);
        if (threadIdx.x == 0) S->maskSlot = op.maskSlot;
        break;
    }
    default:
        break;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S->flags & 0x100) {
#ifdef D4G_HOSTSIM
            fprintf(stderr, "state op %d kind %d: tree limiter failed\n", opId, op.kind);
#endif
            atomicAdd(c.errors, 1);
        }
        *keyp = (op.seq >= 0 && S->valid) ? D4G_MAKE_KEY(S->sizeBits, (long long)opId * 64) : D4G_KEY_NONE;
    }
    wg_store_state(dst, S);
#ifdef D4G_PROFILE_OPS
    if (threadIdx.x == 0 && c.opStats) {
        atomicAdd((unsigned long long*)&c.opStats[op.kind * 2 + (op.arg & 1) * 32], (unsigned long long)(clock64() - tStart));
        atomicAdd((unsigned long long*)&c.opStats[op.kind * 2 + 1 + (op.arg & 1) * 32], 1ULL);
    }
#endif
}

// XCD-aware (block, op) mapping: workgroups g and g+8 share an XCD (and its L2), so all ops
// of one deflate block are given workgroup ids that are equal mod 8 — they re-read the same
// tokens and decoded bytes.  Speed only; nothing depends on placement.
// Blocks are taken in tiles of 8 x tileGroups (D4G_TILE_GROUPS, default: the whole launch is one tile); inside a tile
// the order is op-major: the level's ops are sorted longest first, so every block's long ops start before anyone's
// short ones and the short ones fill the tail.  Smaller tiles keep a block's ops closer in time (less L2 re-fetch,
// measured 152 -> 53 MB per launch with 64-block tiles) but cost ~7 % in time: HBM is nowhere near the limit here.
D4G_DEV bool d4g_map_wg(int nActive, int nOpsLevel, int tileGroups, int& blkSlot, int& opRel) {
    int g = blockIdx.x;
    int x = g & 7, j = g >> 3;
    int perTile = tileGroups * nOpsLevel;
    int tile = j / perTile;
    int rem = j - tile * perTile;
    opRel = rem / tileGroups;
    int local = tile * tileGroups + (rem - opRel * tileGroups);
    blkSlot = local * 8 + x;
    return blkSlot < nActive && opRel < nOpsLevel;
}

#ifndef D4G_STATE_WAVES
#define D4G_STATE_WAVES 8
#endif
__global__ void __launch_bounds__(256) D4G_WAVES_PER_SIMD(D4G_STATE_WAVES) k_exec_state_ops(D4GCtx c, const int32_t* opList, int nOpsLevel) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    __shared__ D4GLds L;
    int bs, orel;
    if (!d4g_map_wg(c.nActive, nOpsLevel, c.tileGroups, bs, orel)) return;
    d4g_exec_state_op(&L, c, c.active[bs], opList[orel]);
}

// Same body for the token-pass-only ops (optimise / least-expensive pruning): they have no single-lane
// section, so they run with wide workgroups (up to 16 waves sweep the block's tokens together).
__global__ void __launch_bounds__(1024) k_exec_state_ops_wide(D4GCtx c, const int32_t* opList, int nOpsLevel) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    __shared__ D4GLds L;
    int bs, orel;
    if (!d4g_map_wg(c.nActive, nOpsLevel, c.tileGroups, bs, orel)) return;
    d4g_exec_state_op(&L, c, c.active[bs], opList[orel]);
}

// The persistent executor serves small launches (merge rounds, few-block batches) where a workgroup's own latency is
// what is waited for, not the occupancy: a 128-VGPR budget removes its spills (merge phase of an 8 MiB stream -9 %).
#ifndef D4G_PERSIST_WAVES
#define D4G_PERSIST_WAVES 4
#endif
__global__ void __launch_bounds__(256) D4G_WAVES_PER_SIMD(D4G_PERSIST_WAVES) k_persist_state_ops(D4GCtx c, D4GQueue q) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    __shared__ D4GLds L;
    __shared__ int sTask[3], sOk;
    int cursor = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int oi = 0, ai = 0;
            sTask[0] = d4g_pull_task(q, cursor, oi, ai) ? 1 : 0;
            sTask[1] = oi;
            sTask[2] = ai;
        }
        __syncthreads();
        if (!sTask[0]) break;
        int opId = q.opFlat[sTask[1]];
        int blk = c.active[sTask[2]];
        const D4GOp op = c.ops[opId];
        bool ok = wg_wait_slot(c, q, blk, op.src, &sOk);
        if (ok) d4g_exec_state_op(&L, c, blk, opId);
        else if (threadIdx.x == 0) {
            c.keys[(long long)blk * c.nOps + opId] = D4G_KEY_NONE;
            if (op.dst >= 0) st_sc1((uint32_t*)&state_ptr(c, blk, op.dst)->valid, 0u);
        }
        if (op.dst >= 0) wg_publish_slot(c, q, blk, op.dst);
    }
}

// ---------------------------------------------------------------------------------------
// Header search: the 56 optimiseBlockDynBlock candidates of one base block, one lane each.
// DeflateStream.java:184-198 (rewriteHeader(flags) -> [recodeHeaderToLessRLEMatches] ->
// optimiseHeader) enumerated in addOptimisedRecoded's loop order (:281-315).
// Pairs are never stored: each lane re-generates them from the shared code-length runs and
// keeps only the 19 symbol counts, two code-length-code tables and its tree scratch in LDS.
// ---------------------------------------------------------------------------------------
// one code-length tree per lane: 16-bit queue entries (weight < 1024, node id < 64), depths in the queue's memory
typedef TreeMem<uint16_t, uint8_t, 20, 6, true> D4GHdrTree;
struct D4GHdrLds {
    alignas(16) uint8_t lens[D4G_NLIT + D4G_NDIST];
    uint8_t runV[D4G_MAXPAIRS];
    uint16_t runL[D4G_MAXPAIRS];
    int nRuns;              // runs whose packing depends on the flags (the others are summed in baseFreq)
    uint32_t baseFreq[20];  // code-length symbols contributed by the runs every candidate packs as plain literals
    alignas(16) unsigned char tree[D4GHdrTree::bytes(64)];
    uint16_t freq[19 * 64];
    uint8_t cl0[19 * 64];
    uint8_t cl1[19 * 64];
};

__device__ __forceinline__ long long d4g_hdr_candidate_body(D4GHdrLds* H, int lane, int flags, int prune, long long litlenBits, long long* prof = nullptr) {
#ifdef D4G_PROFILE_OPS
    long long q0 = d4g_clock_drained();
#endif
    D4GHdrTree tm;
    tm.carve(H->tree, 64);
#define FQ(s) H->freq[(s) * 64 + lane]
#define C0(s) H->cl0[(s) * 64 + lane]
#define C1(s) H->cl1[(s) * 64 + lane]
    int nRuns = H->nRuns;
    for (int s = 0; s < 19; s++) { FQ(s) = (uint16_t)H->baseFreq[s]; C0(s) = 0; C1(s) = 0; }
    // rewriteHeader(flags): symbol counts of the packed lengths
    // The repeat symbols are counted in registers; each run touches LDS once, for its own (wave-uniform) value.
    {
        int c16 = 0, c17 = 0, c18 = 0;
        for (int r = 0; r < nRuns; r++) {
            const int v = H->runV[r];
            d4g_pack_kinds(v, H->runL[r], flags,
                           [&](int sym, int, int, int cnt) { if (sym == 16) c16 += cnt; else if (sym == 17) c17 += cnt; else c18 += cnt; },
                           [&](int cnt) { FQ(v) += (uint16_t)cnt; });
        }
        FQ(16) += (uint16_t)c16; FQ(17) += (uint16_t)c17; FQ(18) += (uint16_t)c18;
    }
#ifdef D4G_PROFILE_OPS
    long long q1 = d4g_clock_drained();
#endif
    d4g_build_tree(tm, 64, lane, 19, 7, [&](int i) { return (unsigned)FQ(i); }, [&](int v, int len) { C0(v) = (uint8_t)len; });
#ifdef D4G_PROFILE_OPS
    long long q2 = d4g_clock_drained();
#endif
    long long hdr = 5 + 5 + 4 + 19 * 3;
    for (int s = 0; s < 19; s++) hdr += (long long)FQ(s) * (C0(s) + (s >= 16 ? pair_extra_bits(s) : 0));
    int nCl = trim_codelens(19, [&](int s) { return (int)C0(s); });
    hdr -= 3 * (19 - nCl);
    bool useC1 = false;
    if (prune) {
        // recodeHeaderToLessRLEMatches: expand runs that are not shorter than literals, then re-derive the code
        for (int s = 0; s < 19; s++) FQ(s) = (uint16_t)H->baseFreq[s];
        {
            const int l16 = C0(16), l17 = C0(17), l18 = C0(18), l0 = C0(0);
            int c16 = 0, c17 = 0, c18 = 0, z = 0;
            for (int r = 0; r < nRuns; r++) {
                const int v = H->runV[r];
                const int lv = C0(v);
                int addV = 0;
                d4g_pack_kinds(v, H->runL[r], flags,
                               [&](int sym, int run, int, int cnt) {
                                   // sym 16 repeats `v`, 17 / 18 repeat zero
                                   int lsym = sym == 16 ? l16 : sym == 17 ? l17 : l18;
                                   int lval = sym == 16 ? lv : l0;
                                   bool expand = pair_gain_bits(sym, run, lsym, lval, true) >= 0;
                                   if (expand) { if (sym == 16) addV += run * cnt; else z += run * cnt; }
                                   else if (sym == 16) c16 += cnt;
                                   else if (sym == 17) c17 += cnt;
                                   else c18 += cnt;
                               },
                               [&](int cnt) { addV += cnt; });
                if (addV) FQ(v) += (uint16_t)addV;
            }
            FQ(16) += (uint16_t)c16; FQ(17) += (uint16_t)c17; FQ(18) += (uint16_t)c18; FQ(0) += (uint16_t)z;
        }
        d4g_build_tree(tm, 64, lane, 19, 7, [&](int i) { return (unsigned)FQ(i); }, [&](int v, int len) { C1(v) = (uint8_t)len; });
        nCl = trim_codelens(nCl, [&](int s) { return (int)C1(s); });
        hdr = 5 + 5 + 4 + 3 * nCl;
        for (int s = 0; s < 19; s++) hdr += (long long)FQ(s) * (C1(s) + (s >= 16 ? pair_extra_bits(s) : 0));
        useC1 = true;
    }
    // optimiseHeader: trim again, then expand runs that are strictly longer than literals
    {
        int n2 = useC1 ? trim_codelens(nCl, [&](int s) { return (int)C1(s); }) : trim_codelens(nCl, [&](int s) { return (int)C0(s); });
        hdr -= 3 * (nCl - n2);
        nCl = n2;
    }
    long long saved = 0;
    {
        const int a16 = C0(16), a17 = C0(17), a18 = C0(18), a0 = C0(0);
        const int b16 = C1(16), b17 = C1(17), b18 = C1(18), b0 = C1(0);
        for (int r = 0; r < nRuns; r++) {
            const int v = H->runV[r];
            const int av = C0(v), bv = C1(v);
            d4g_pack_kinds(v, H->runL[r], flags,
                           [&](int sym, int run, int, int cnt) {
                               int as = sym == 16 ? a16 : sym == 17 ? a17 : a18, al = sym == 16 ? av : a0;
                               int g;
                               if (useC1) {
                                   if (pair_gain_bits(sym, run, as, al, true) >= 0) return;  // already literals
                                   int bs = sym == 16 ? b16 : sym == 17 ? b17 : b18, bl = sym == 16 ? bv : b0;
                                   g = pair_gain_bits(sym, run, bs, bl, false);
                               } else {
                                   g = pair_gain_bits(sym, run, as, al, false);
                               }
                               if (g >= 0) saved += (long long)g * cnt;
                           },
                           [&](int) {});
        }
    }
    hdr -= saved;
#ifdef D4G_PROFILE_OPS
    if (prof && lane == 0) {
        atomicAdd((unsigned long long*)&prof[48], (unsigned long long)(q1 - q0));          // first count pass
        atomicAdd((unsigned long long*)&prof[49], (unsigned long long)(q2 - q1));          // first tree
        atomicAdd((unsigned long long*)&prof[50], (unsigned long long)(d4g_clock_drained() - q2));   // rest (lane 0 has no prune)
        atomicAdd((unsigned long long*)&prof[51], 1ULL);
    }
#endif
#undef FQ
#undef C0
#undef C1
    return litlenBits + hdr;
}
// (a call in the header-search kernels; the fused executor inlines the body under its own register budget)
__device__ long long d4g_hdr_candidate(D4GHdrLds* H, int lane, int flags, int prune, long long litlenBits, long long* prof = nullptr) {
    return d4g_hdr_candidate_body(H, lane, flags, prune, litlenBits, prof);
}

__device__ void d4g_exec_hdr_search(D4GHdrLds& H, uint8_t* comb, const D4GCtx& c, int blk, int opId) {
    const D4GOp op = c.ops[opId];
    const D4GState* base = state_ptr(c, blk, op.src);
    long long* keyp = c.keys + (long long)blk * c.nOps + opId;
    int lane = threadIdx.x & 63;
#ifdef D4G_PROFILE_OPS
    long long hs0 = d4g_clock_drained();
#endif
    __syncthreads();
    if (!ld_state_i32(&base->valid) || ld_state_i32(&base->type) != D4G_DYNAMIC) {
        if (lane == 0) *keyp = D4G_KEY_NONE;
        return;
    }
    int nLit = ld_state_i32(&base->nLit), n = nLit + ld_state_i32(&base->nDist);
    // code lengths word by word (litLen and distLen are 4-byte aligned, contiguous in the state)
    for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) ((uint32_t*)H.lens)[i] = ld_sc1((const uint32_t*)base->litLen + i);
    __syncthreads();
    for (int i = lane; i < D4G_NLIT + D4G_NDIST; i += 64) comb[i] = i >= n ? 0 : i < nLit ? H.lens[i] : H.lens[D4G_NLIT + i - nLit];
    __syncthreads();
    long long baseLitlenBits = (long long)ld_sc1((const uint64_t*)&base->litlenBits);
    // ---- memo lookup: two position-keyed 64-bit hashes of the length set ----
    D4GHsMemo* mine = nullptr;   // the entry this search owns and must publish
    unsigned long long h1 = 0, h2 = 0;
    if (c.hsMemo) {
        for (int i = lane; i < n; i += 64) {
            unsigned long long w = (unsigned long long)comb[i] + 1;
            unsigned long long x = w * 0x9e3779b97f4a7c15ULL + (unsigned long long)(i + 1) * 0xbf58476d1ce4e5b9ULL;
            x ^= x >> 29; x *= 0x94d049bb133111ebULL; x ^= x >> 32;
            unsigned long long y = (w + 0x632be59bd9b4e019ULL) * ((unsigned long long)(i + 7) * 0xd6e8feb86659fd93ULL | 1ULL);
            y ^= y >> 31; y *= 0xff51afd7ed558ccdULL; y ^= y >> 33;
            h1 += x;
            h2 += y;
        }
        h1 = (unsigned long long)wave_sum_i64((long long)h1) + (unsigned long long)nLit * 0x100000001b3ULL;
        h2 = (unsigned long long)wave_sum_i64((long long)h2) ^ ((unsigned long long)n << 40);
        if (h1 == 0) h1 = 1;
        D4GHsMemo* tab = c.hsMemo + (long long)blk * D4G_HSMEMO_SLOTS;
        long long found = -1;   // >= 0: published result (hdr << 8 | lane)
        int slot = -1;          // the entry claimed (mine) or hit
        if (lane == 0) {
            for (int probe = 0; probe < 8; probe++) {
                D4GHsMemo* e = tab + ((h1 >> 7) + probe) % D4G_HSMEMO_SLOTS;
                unsigned long long t = atomicCAS(&e->tag, 0ULL, h1);
                if (t == 0) { mine = e; slot = (int)(e - tab); break; }           // ours to compute
                if (t == h1) {                             // someone has it or is computing it: wait for the result
                    int st = 0;
                    for (int spin = 0; spin < (1 << 18); spin++) {
                        st = d4g_flag_load(&e->state);
                        if (st == 2) break;
                        d4g_sleep();
                    }
                    if (st == 2 && ld_sc1((const uint64_t*)&e->check) == (uint64_t)h2) {
                        found = ((long long)ld_state_i32(&e->hdr) << 8) | (long long)ld_state_i32(&e->lane);
                        slot = (int)(e - tab);
                    }
                    break;                                 // (timeout or a hash clash: compute without the memo)
                }
            }
        }
        found = __shfl(found, 0);
        slot = __shfl(slot, 0);
        const bool owner = __shfl(mine != nullptr ? 1 : 0, 0) != 0;
        // the hashes only find the entry: the key (nLit, n, the code lengths) is written by the owner at claim time and
        // a hit is confirmed against it.  comb[] is zero beyond n (cleared above), so whole words compare.
        if (slot >= 0) {
            D4GHsMemo* e = tab + slot;
            const uint32_t* cw = (const uint32_t*)comb;
            if (owner) {
                for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) st_sc1(&e->key[i], cw[i]);
                if (lane == 0) { st_sc1((uint32_t*)&e->nLit, (uint32_t)nLit); st_sc1((uint32_t*)&e->n, (uint32_t)n); }
            } else if (found >= 0) {
                bool bad = false;
                for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) bad |= ld_sc1(&e->key[i]) != cw[i];
                if (lane == 0) bad |= ld_state_i32(&e->nLit) != nLit || ld_state_i32(&e->n) != n;
                if (d4g_ballot(bad)) found = -1;             // same hashes, different lengths: compute without the memo
            }
        }
        if (found >= 0) {
            if (lane == 0) *keyp = D4G_MAKE_KEY(baseLitlenBits + (found >> 8), (long long)opId * 64 + (found & 255));
            return;
        }
    }
    // Runs of the combined code lengths, found with ballots.  HuffmanTable.pack turns a run of up to three equal
    // non-zero lengths (up to two zeros) into plain literals whatever the flags: those only add to the symbol
    // counts, once for all 56 candidates.  The other runs are listed for the candidates' own packing (their
    // order does not matter: candidates only count symbols and sum savings).
    if (lane < 20) H.baseFreq[lane] = 0;
    __syncthreads();
    {
        constexpr int NCH = (D4G_NLIT + D4G_NDIST) / 64;
        unsigned long long sm[NCH];
        int v[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int i = ch * 64 + lane;
            v[ch] = i < n ? (int)comb[i] : -1;
            int pv = (i > 0 && i < n) ? (int)comb[i - 1] : -2;
            sm[ch] = d4g_ballot(i < n && v[ch] != pv);
        }
        int ncx = 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int i = ch * 64 + lane;
            bool start = (sm[ch] >> lane) & 1;
            int run = 0;
            if (start) {
                int nx = -1;
                unsigned long long m = lane == 63 ? 0ULL : (sm[ch] >> (lane + 1));
                if (m) nx = i + __ffsll((long long)m);
#pragma unroll
                for (int c2 = ch + 1; c2 < NCH; c2++)
                    if (nx < 0 && sm[c2]) nx = c2 * 64 + __ffsll((long long)sm[c2]) - 1;
                if (nx < 0) nx = n;
                run = nx - i;
            }
            bool simple = start && (v[ch] != 0 ? run <= 3 : run <= 2);
            bool cx = start && !simple;
            if (simple) atomicAdd(&H.baseFreq[v[ch]], (unsigned)run);
            unsigned long long cm = d4g_ballot(cx);
            if (cx) {
                int idx = ncx + __popcll(cm & ((1ULL << lane) - 1));
                H.runV[idx] = (uint8_t)v[ch];
                H.runL[idx] = (uint16_t)run;
            }
            ncx += __popcll(cm);
        }
        if (lane == 0) H.nRuns = ncx;
    }
    __syncthreads();
#ifdef D4G_PROFILE_OPS
    long long hs1 = d4g_clock_drained();
#endif
    long long key = D4G_KEY_NONE;
    if (lane < 56) {
        long long size = d4g_hdr_candidate(&H, lane, c.hdrFlags[lane], c.hdrPrune[lane], baseLitlenBits, c.opStats);
        key = D4G_MAKE_KEY(size, (long long)opId * 64 + lane);
    }
    key = wave_min_i64(key);
    if (lane == 0) {
        *keyp = key;
        if (mine) {   // publish: result first, then the flag
            long long bestSize = key >> D4G_KEY_SEQ_BITS;
            st_sc1((uint32_t*)&mine->hdr, (uint32_t)(int32_t)(bestSize - baseLitlenBits));
            st_sc1((uint32_t*)&mine->lane, (uint32_t)(key & 63));
            st_sc1((uint64_t*)&mine->check, (uint64_t)h2);
            d4g_drain_stores();
            d4g_flag_store(&mine->state, 2);
        }
    }
#ifdef D4G_PROFILE_OPS
    if (c.opStats && lane == 0) {
        atomicAdd((unsigned long long*)&c.opStats[52], (unsigned long long)(hs1 - hs0));          // load + runs
        atomicAdd((unsigned long long*)&c.opStats[53], (unsigned long long)(d4g_clock_drained() - hs1));   // candidates (all lanes)
        atomicAdd((unsigned long long*)&c.opStats[54], 1ULL);
        atomicAdd((unsigned long long*)&c.opStats[55], (unsigned long long)H.nRuns);
    }
#endif
}

__global__ void __launch_bounds__(64) k_exec_hdr_search(D4GCtx c, const int32_t* opList, int nOpsLevel) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    __shared__ D4GHdrLds H;
    __shared__ __attribute__((aligned(16))) uint8_t comb[D4G_NLIT + D4G_NDIST];   // (compared word-wise against the memo's key)
    int bs, orel;
    if (!d4g_map_wg(c.nActive, nOpsLevel, c.tileGroups, bs, orel)) return;
    d4g_exec_hdr_search(H, comb, c, c.active[bs], opList[orel]);
}

__global__ void __launch_bounds__(64) k_persist_hdr_search(D4GCtx c, D4GQueue q) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    __shared__ D4GHdrLds H;
    __shared__ __attribute__((aligned(16))) uint8_t comb[D4G_NLIT + D4G_NDIST];   // (compared word-wise against the memo's key)
    __shared__ int sTask[3], sOk;
    int cursor = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int oi = 0, ai = 0;
            sTask[0] = d4g_pull_task(q, cursor, oi, ai) ? 1 : 0;
            sTask[1] = oi;
            sTask[2] = ai;
        }
        __syncthreads();
        if (!sTask[0]) break;
        int opId = q.opFlat[sTask[1]];
        int blk = c.active[sTask[2]];
        bool ok = wg_wait_slot(c, q, blk, c.ops[opId].src, &sOk);
        if (ok) d4g_exec_hdr_search(H, comb, c, blk, opId);
        else if (threadIdx.x == 0) c.keys[(long long)blk * c.nOps + opId] = D4G_KEY_NONE;
    }
}

// ---------------------------------------------------------------------------------------
// Selection: first strict minimum over the round's candidates (DeflateStream.java:349-368),
// winner copied into the block's slot 0.  A header-search winner is materialised by running
// the reference's own sequence of header operations on its base block.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_select(D4GCtx c, D4GRoundResult* results) {
    __shared__ D4GLds L;
    if ((int)blockIdx.x >= c.nActive) return;
    int blk = c.active[blockIdx.x];
    const D4GBlock b = c.blocks[blk];
    D4GState* S = &L.st;
    D4GState* cur = state_ptr(c, blk, 0);
    if (d4g_flag_load(c.errors + 1) != 0) {   // a wait of the persistent executor gave up: the keys are incomplete — nothing is selected
        if (threadIdx.x == 0) {
            D4GRoundResult r;
            r.curSize = cur->sizeBits; r.bestSize = cur->sizeBits; r.bestSeq = -1; r.improved = -1; r.newType = cur->type; r.pad = 0;
            results[blockIdx.x] = r;
        }
        return;
    }
    const long long* keys = c.keys + (long long)blk * c.nOps;
    long long best = D4G_KEY_NONE;
    for (int i = threadIdx.x; i < c.nOps; i += blockDim.x) best = keys[i] < best ? keys[i] : best;
    best = wave_min_i64(best);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) L.red[threadIdx.x >> 6] = best;
    __syncthreads();
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) best = L.red[i] < best ? L.red[i] : best;
    __syncthreads();
    long long curSize = cur->sizeBits;
    long long bestSize = best >> D4G_KEY_SEQ_BITS;
    int seq = (int)(best & ((1 << D4G_KEY_SEQ_BITS) - 1));
    bool improved = best != D4G_KEY_NONE && bestSize < curSize;
    if (improved) {
        int opId = seq >> 6, lane = seq & 63;
        const D4GOp op = c.ops[opId];
        int slot = (op.kind == OP_CAND || op.kind == OP_HDRSEARCH) ? op.src : op.dst;
        wg_load_state(S, state_ptr(c, blk, slot));
        if (op.kind == OP_HDRSEARCH) {
            wg_rewrite_header(&L, c.hdrFlags[lane]);
            if (c.hdrPrune[lane]) { wg_replace_rle_runs(&L, true); wg_recode_header(&L); }
            wg_optimise_header(&L);
            if (threadIdx.x == 0 && S->sizeBits != bestSize) {
#ifdef D4G_HOSTSIM
                fprintf(stderr, "select: header candidate op %d lane %d materialised to %lld bits (litlen %lld hdr %lld nPairs %d nCl %d nLit %d nDist %d), search said %lld; base size %lld litlen %lld hdr %lld type %d\n", opId, lane, (long long)S->sizeBits, (long long)S->litlenBits, (long long)S->hdrBits, S->nPairs, S->nCl, S->nLit, S->nDist, bestSize,
                        (long long)state_ptr(c, blk, slot)->sizeBits, (long long)state_ptr(c, blk, slot)->litlenBits, (long long)state_ptr(c, blk, slot)->hdrBits, state_ptr(c, blk, slot)->type);
#endif
                atomicAdd(c.errors, 1);
            }
        }
        if (S->maskSlot != 0) wg_copy_mask(mask_ptr(c, b, 0), mask_ptr(c, b, S->maskSlot), b.maskWords);
        __syncthreads();
        if (threadIdx.x == 0) { S->maskSlot = 0; S->valid = 1; }
        wg_store_state(cur, S);
    }
    if (threadIdx.x == 0) {
        D4GRoundResult r;
        r.curSize = curSize;
        r.bestSize = improved ? bestSize : curSize;
        r.bestSeq = improved ? (seq >> 6) : -1;
        r.improved = improved ? 1 : 0;
        r.newType = improved ? S->type : cur->type;
        r.pad = 0;
        results[blockIdx.x] = r;
    }
}
