// d4g_zopfli.h — gfx950 kernels of the Zopfli deflate encoder deft4j's recompress modes ZOPFLI* call
// (C/MultiCafeUndZopfliCompressor.java:19-25,33,48-52 — CafeUndZopfli 5cdf283e67; C/MultiJZopfliCompressor.java:18-60,78-85 —
// jzopfli 0.0.4; both absent from /root/reference, so the RESULT reproduced is that of the published algorithm as
// oracle/zopfli_oracle.c restates and pins it: byte-identical to libzopfli 1.0.3 and to the reference's own
// test/asyoulik/asyoulik-zopfli.txt.gz).
//
// Re-designed for the GPU rather than translated:
//  * Zopfli's hash chains (3-byte hash + the run-length "second hash") become sorted buckets: the positions of every 32 Ki
//    sort block are ordered by (hash, position) once per hash kind, so a position's chain is the part of its bucket below its
//    own slot plus the previous block's bucket.  A wave takes 64 chain nodes per step and replays the chain walk (first-hash
//    chain, the switch to the second chain, the 8192-hit cap, the quick reject, the stop at `limit`) with ballots and a prefix
//    maximum.  The result of every search — the (length, distance) change points of `sublen` — is stored once per input
//    position ("match table") and shared by every block-splitting option and squeeze iteration; Zopfli's longest-match cache
//    memoises the same function per block.  Positions whose result depends on where their block ends (the last 258 bytes, or
//    a byte run that crosses the end) are searched again per block ("tail tables") by a window scan that keeps the keys of
//    the 32 Ki positions before a tile in LDS and applies the block's end to the run lengths on the fly.
//  * The squeeze (iterated shortest path) runs one wave per block with the cost window and the chosen lengths in LDS
//    rings; lanes relax the 3..258 match lengths of a position together.  Costs are the same IEEE doubles / floats in
//    the same order as the CPU code; log() is zopf_portable_log restated (+,-,*,/ only, no contraction).
//  * Length-limited code lengths come from a level-by-level package-merge whose ties break as Zopfli's boundary
//    package-merge does (a package before a leaf of equal weight); on the GPU every level is merged by the whole wave with binary searches.
//  * Block splitting evaluates the nine probes of a round from ten segment histograms counted in one sweep.
#pragma once
#include "d4g_device.h"
#include "d4g_lz77.h"

#pragma clang fp contract(off)

#define ZF_WSIZE 32768
#define ZF_MAXM 258
#define ZF_NUM_LL 288
#define ZF_NUM_D 32
#define ZF_MAX_HITS 8192
#define ZF_TILE 512            // positions per match-table workgroup
#define ZF_MATCH_THREADS 512
#define ZF_KEY_TILE 1024
#define ZF_SPLIT_FIRST 0
#define ZF_SPLIT_LAST 1
#define ZF_SPLIT_NONE 2
#define ZF_POOL_LINK 0x80000000u

struct ZfInput {
    const uint8_t* data;     // 16-byte aligned, >= 320 zero bytes after the end (as LzStream::data)
    long long n;
    uint16_t* val;           // 3-byte hash of every position
    uint16_t* same;          // following bytes equal to this one (capped at 65535), relative to the input's end
    uint16_t* lead;          // per ZF_KEY_TILE tile: length of its leading run
    uint16_t* val2;          // the second hash ((same - 3) & 255) ^ val, with `same` relative to the input's end
    // per 32 Ki-position sort block and per hash kind (0: val, 1: val2): positions ordered by (key, position)
    uint16_t* sorted[2];     // [position slot] position within its sort block, bucket after bucket
    uint16_t* rank[2];       // [position] its slot within the sort block
    uint16_t* bstart[2];     // [sort block * 32768 + key] first slot of the key's bucket
    uint32_t* table;         // 8 words per position: change points (len << 16 | dist), ascending; 0 = unused;
                             // word 7 = ZF_POOL_LINK | index of the remaining points in `pool`
    uint32_t* best;          // (longest length << 16) | its distance; 0 when shorter than 3
};
struct ZfPool { uint32_t* words; uint32_t cap; uint32_t* used; int32_t* error; };
struct ZfKeyJob { int32_t input, tile; };
struct ZfMatchJob {
    int32_t input, count;    // positions [first, first + count), count <= ZF_TILE
    long long first, end;    // `end` = end of the block the results are for (the input's end for the shared table)
    uint32_t* table;         // entry of position `first`
    uint32_t* best;
};

D4G_DEV int zf_hash3(const uint8_t* d, long long p) { return (((d[p] << 5) ^ d[p + 1]) << 5 ^ d[p + 2]) & 32767; }

// ---------------------------------------------------------------------------------------------------------------
// keys: hash value and run length of every position (hash.c's val / same, with `end` = the input's end)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_zf_keys_a(const ZfInput* inputs, const ZfKeyJob* jobs) {
    const ZfKeyJob job = jobs[blockIdx.x];
    const ZfInput in = inputs[job.input];
    const long long t0 = (long long)job.tile * ZF_KEY_TILE;
    __shared__ int firstDiff;
    if (threadIdx.x == 0) firstDiff = ZF_KEY_TILE;
    __syncthreads();
    const int b0 = in.data[t0];
    for (int k = threadIdx.x; k < ZF_KEY_TILE; k += blockDim.x) {
        const long long p = t0 + k;
        if (p >= in.n) { atomicMin(&firstDiff, k); break; }
        in.val[p] = (uint16_t)zf_hash3(in.data, p);     // the bytes past the end are zero, as the published code substitutes
        if (in.data[p] != b0) atomicMin(&firstDiff, k);
    }
    __syncthreads();
    if (threadIdx.x == 0) in.lead[job.tile] = (uint16_t)firstDiff;
}
__global__ void __launch_bounds__(256) k_zf_keys_b(const ZfInput* inputs, const ZfKeyJob* jobs) {
    const ZfKeyJob job = jobs[blockIdx.x];
    const ZfInput in = inputs[job.input];
    const long long t0 = (long long)job.tile * ZF_KEY_TILE;
    const int cnt = (int)((in.n - t0) < ZF_KEY_TILE ? (in.n - t0) : ZF_KEY_TILE);
    const long long nTiles = (in.n + ZF_KEY_TILE - 1) / ZF_KEY_TILE;
    __shared__ int nb[ZF_KEY_TILE];     // index of the first position >= k after which the byte changes (or the input ends)
    for (int k = threadIdx.x; k < ZF_KEY_TILE; k += blockDim.x) {
        int v = 1 << 20;
        if (k < cnt) { const long long p = t0 + k; if (p + 1 >= in.n || in.data[p + 1] != in.data[p]) v = k; }
        nb[k] = v;
    }
    __syncthreads();
    for (int d = 1; d < ZF_KEY_TILE; d <<= 1) {   // suffix minimum
        int v[ZF_KEY_TILE / 256];
        for (int r = 0, k = threadIdx.x; k < ZF_KEY_TILE; k += blockDim.x, r++) { int a = nb[k], b = k + d < ZF_KEY_TILE ? nb[k + d] : (1 << 20); v[r] = a < b ? a : b; }
        __syncthreads();
        for (int r = 0, k = threadIdx.x; k < ZF_KEY_TILE; k += blockDim.x, r++) nb[k] = v[r];
        __syncthreads();
    }
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
        const long long p = t0 + k;
        long long s;
        if (nb[k] < (1 << 20)) s = nb[k] - k;
        else {   // the run leaves the tile: add the leading runs of the following tiles
            s = cnt - 1 - k;
            for (long long t = job.tile + 1; t < nTiles && s < 65535; t++) {
                if (in.data[t * ZF_KEY_TILE] != in.data[p]) break;
                const int l = in.lead[t];
                s += l;
                const long long tl = (in.n - t * ZF_KEY_TILE) < ZF_KEY_TILE ? (in.n - t * ZF_KEY_TILE) : ZF_KEY_TILE;
                if (l < tl) break;
            }
        }
        const int sm = (int)(s > 65535 ? 65535 : s);
        in.same[p] = (uint16_t)sm;
        in.val2[p] = (uint16_t)((((sm - 3) & 255) ^ zf_hash3(in.data, p)) & 32767);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// match table (lz77.c ZopfliFindLongestMatch for every position, limit = 258, with sublen)
// ---------------------------------------------------------------------------------------------------------------
D4G_DEV unsigned long long zf_load8(const uint8_t* base, long long off) {   // bytes [off, off + 8) of a 4-byte aligned array
    const uint32_t* w = (const uint32_t*)base;
    const long long i = off >> 2;
    const int sh = (int)(off & 3) * 8;
    const unsigned long long lo = ((unsigned long long)w[i + 1] << 32) | w[i];
    return sh ? (lo >> sh) | ((unsigned long long)w[i + 2] << (64 - sh)) : lo;
}
D4G_DEV int zf_match_len(const uint8_t* d, long long p, long long q, int limit) {
    int k = 0;
    while (k < limit) {
        unsigned long long x = zf_load8(d, p + k) ^ zf_load8(d, q + k);
        if (x) { k += (__ffsll((long long)x) - 1) >> 3; break; }
        k += 8;
    }
    return k < limit ? k : limit;
}
D4G_DEV int wave_incl_max_i32(int v) {
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d);
        if (lane >= d && o > v) v = o;
    }
    return v;
}
D4G_DEV int zf_samecap(int same, long long end, long long x) {   // hash.c: the run is cut at the block's end
    const long long c = end - x - 1;
    return same < c ? same : (int)c;
}

struct ZfMatchLds {
    uint32_t key[ZF_WSIZE + ZF_TILE];          // val | same << 16 of the window and the tile
    uint32_t cps[ZF_MATCH_THREADS / 64][264];   // change points of the position a wave is searching
};

__global__ void __launch_bounds__(ZF_MATCH_THREADS) k_zf_match(const ZfInput* inputs, const ZfMatchJob* jobs, ZfPool pool) {
    __shared__ ZfMatchLds L;
    const ZfMatchJob job = jobs[blockIdx.x];
    const ZfInput in = inputs[job.input];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long w0 = job.first - ZF_WSIZE;                 // position of key[0]
    const long long lo0 = w0 < 0 ? 0 : w0;
    for (long long x = lo0 + threadIdx.x; x < job.first + job.count; x += blockDim.x)
        L.key[x - w0] = (uint32_t)in.val[x] | ((uint32_t)in.same[x] << 16);
    __syncthreads();
    uint32_t* cps = L.cps[wave];
    for (int pi = wave; pi < job.count; pi += nw) {
        const long long p = job.first + pi;
        int limit = (int)((job.end - p) < ZF_MAXM ? (job.end - p) : ZF_MAXM);
        int best = 1, bestDist = 0, ncp = 0;
        if (limit >= 3) {
            const uint32_t kp = L.key[p - w0];
            const int key1 = (int)(kp & 0xffff);
            const int sp = zf_samecap((int)(kp >> 16), job.end, p);
            const int key2 = ((sp - 3) & 255) ^ key1;
            const long long lo = p - (ZF_WSIZE - 1) < 0 ? 0 : p - (ZF_WSIZE - 1);
            int mode = 1, hits = 0;
            bool done = false;
            for (long long base = p - 1; base >= lo && !done; base -= 64) {
                const long long q = base - lane;
                const bool valid = q >= lo;
                bool m1 = false, m2 = false;
                if (valid) {
                    const uint32_t kq = L.key[q - w0];
                    const int v = (int)(kq & 0xffff);
                    m1 = v == key1;
                    m2 = ((((zf_samecap((int)(kq >> 16), job.end, q) - 3) & 255) ^ v) == key2);
                }
                const unsigned long long M1 = __ballot(m1), M2 = __ballot(m2);
                if ((mode == 1 ? M1 : M2) == 0ull) continue;               // the switch to the second chain is only tested at first-chain nodes
                // lz77.c's quick reject: a candidate that differs at offset `bestlength` cannot set a record, and only records matter
                int len = 0;
                if ((m2 || (mode == 1 && m1)) && in.data[q + best] == in.data[p + best]) len = zf_match_len(in.data, p, q, limit);
                int from = 0;
                while (!done) {
                    unsigned long long memb = (mode == 1 ? M1 : M2);
                    memb = from >= 64 ? 0ull : (memb >> from) << from;
                    if (!memb) break;
                    const bool inM = (memb >> lane) & 1ull;
                    const int pm = wave_incl_max_i32(inM ? len : 0);
                    const int run = pm > best ? pm : best;                 // bestlength after this node
                    const int idx = __popcll(memb & ((1ull << lane) - 1ull));
                    const bool isBreak = inM && len >= limit;
                    const bool hitStop = inM && (hits + idx + 1 >= ZF_MAX_HITS);
                    const bool isSwitch = mode == 1 && inM && run >= sp && m2;
                    const unsigned long long stopM = __ballot(isBreak || hitStop), swM = __ballot(isSwitch);
                    const int ls = stopM ? __ffsll((long long)stopM) - 1 : 64, lw = swM ? __ffsll((long long)swM) - 1 : 64;
                    const int e = ls < lw ? ls : lw;                        // last node processed in this segment (64 = all)
                    const unsigned long long upto = e >= 63 ? ~0ull : ((2ull << e) - 1ull);
                    // a node sets a new record iff its length exceeds everything before it
                    int ex = __shfl_up(pm, 1);
                    if (lane == 0) ex = 0;
                    const int before = ex > best ? ex : best;
                    const bool rec = inM && ((upto >> lane) & 1ull) && len > before;
                    const unsigned long long recM = __ballot(rec);
                    if (rec) {
                        const int r = ncp + __popcll(recM & ((1ull << lane) - 1ull));
                        if (r < 264) cps[r] = ((uint32_t)len << 16) | (uint32_t)(p - q);
                    }
                    if (recM) {
                        const int last = 63 - __clzll((long long)recM);
                        best = __shfl(len, last);
                        bestDist = (int)__shfl((int)(p - q), last);
                        ncp += __popcll(recM);
                    }
                    hits += __popcll(memb & upto);
                    if (ls <= lw && ls < 64) { done = true; break; }
                    if (lw < 64) { mode = 2; from = lw + 1; continue; }
                    break;
                }
            }
        }
        LZ_WAVE_SYNC();
        // entry: the change points of length >= 3
        int skip = 0;
        while (skip < ncp && (int)(cps[skip] >> 16) < 3) skip++;
        const int n3 = ncp - skip;
        uint32_t* ent = job.table + (long long)pi * 8;
        uint32_t link = 0;
        if (n3 > 8) {
            uint32_t at = 0;
            if (lane == 0) {
                at = atomicAdd(pool.used, (uint32_t)(n3 - 7 + 1));
                if (at + (uint32_t)(n3 - 7 + 1) > pool.cap) { atomicAdd(pool.error, 1); at = 0xffffffffu; }
            }
            at = __shfl(at, 0);
            if (at != 0xffffffffu) {
                for (int k = lane; k < n3 - 7; k += 64) pool.words[at + k] = cps[skip + 7 + k];
                if (lane == 0) pool.words[at + n3 - 7] = 0;
                link = ZF_POOL_LINK | at;
            }
        }
        if (lane < 8) {
            uint32_t v = 0;
            if (lane < n3 && (lane < 7 || n3 <= 8)) v = cps[skip + lane];
            if (lane == 7 && n3 > 8) v = link;
            ent[lane] = v;
        }
        if (lane == 0) job.best[pi] = best >= 3 ? (((uint32_t)best << 16) | (uint32_t)bestDist) : 0u;
        LZ_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The shared table (block end = input end) without the window scan: both hash chains are enumerated from buckets.
// k_zf_sort orders the positions of a 32 Ki sort block by (key, position) — k_lz_sort's counting sort with the key read from an
// array — once for the first hash and once for the second; a position's chain is then the part of its bucket below its own slot,
// followed by the previous sort block's bucket from the top (positions further back are out of the window).
// ---------------------------------------------------------------------------------------------------------------
#define ZF_SORT_BLOCK 32768
#define ZF_SORT_THREADS 512
struct ZfSortJob { int32_t input, blk, kind, pad; };
__global__ void __launch_bounds__(ZF_SORT_THREADS) k_zf_sort(const ZfInput* inputs, const ZfSortJob* jobs) {
    __shared__ uint16_t tbl[32768];          // counts, then bucket cursors
    __shared__ unsigned part[ZF_SORT_THREADS];
    const ZfSortJob job = jobs[blockIdx.x];
    const ZfInput in = inputs[job.input];
    const long long p0 = (long long)job.blk * ZF_SORT_BLOCK;
    const int nIns = (int)((in.n - p0) < ZF_SORT_BLOCK ? (in.n - p0) : ZF_SORT_BLOCK);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint16_t* key = (job.kind ? in.val2 : in.val) + p0;
    unsigned* tw = (unsigned*)tbl;
    for (int i = tid; i < 16384; i += ZF_SORT_THREADS) tw[i] = 0;
    __syncthreads();
    for (int i = tid; i < nIns; i += ZF_SORT_THREADS) {
        const unsigned h = key[i];
        atomicAdd(tw + (h >> 1), 1u << ((h & 1) * 16));   // two 16-bit counters per word; a count is at most 32768
    }
    __syncthreads();
    unsigned sum = 0;
    for (int k = 0; k < 32; k++) { unsigned w = tw[tid * 32 + k]; sum += (w & 0xffffu) + (w >> 16); }
    part[tid] = sum;
    __syncthreads();
    if (tid < 64) {
        unsigned loc[8], sacc = 0;
        for (int k = 0; k < 8; k++) { loc[k] = sacc; sacc += part[tid * 8 + k]; }
        unsigned inc = sacc;
        for (int dd = 1; dd < 64; dd <<= 1) { unsigned o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
        const unsigned excl = inc - sacc;
        for (int k = 0; k < 8; k++) part[tid * 8 + k] = excl + loc[k];
    }
    __syncthreads();
    {
        unsigned run = part[tid];
        uint16_t* bs = in.bstart[job.kind] + (long long)job.blk * ZF_SORT_BLOCK;
        for (int k = 0; k < 32; k++) {
            const unsigned w = tw[tid * 32 + k];
            const unsigned a = run, b = run + (w & 0xffffu);
            run = b + (w >> 16);
            tw[tid * 32 + k] = (a & 0xffffu) | (b << 16);
            bs[tid * 64 + 2 * k] = (uint16_t)a;
            bs[tid * 64 + 2 * k + 1] = (uint16_t)b;
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    // placement in position order by one wave, 64 positions per step (as k_lz_sort: lane tags find the lanes that share a key)
    volatile uint16_t* vt = tbl;
    uint16_t* Sout = in.sorted[job.kind] + p0;
    uint16_t* Rout = in.rank[job.kind] + p0;
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (int i0 = 0; i0 < nIns; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < nIns;
        const unsigned h = valid ? key[i] : 0u;
        const unsigned cur = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        if (valid) vt[h] = (uint16_t)(0x8000u | (unsigned)lane);
        LZ_WAVE_SYNC();
        const unsigned w = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        if (valid && (w & 63u) != (unsigned)lane) vt[h] = (uint16_t)(0xC000u | (unsigned)lane);
        LZ_WAVE_SYNC();
        const unsigned w2 = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        const bool dup = valid && (w2 & 0x4000u);
        unsigned rank = cur;
        if (valid && !dup) vt[h] = (uint16_t)(cur + 1);
        unsigned long long rem = __ballot(dup);
        while (rem) {
            const int leader = __ffsll((long long)rem) - 1;
            const unsigned hh = __shfl(h, leader);
            const unsigned long long m = __ballot(dup && h == hh);
            if (dup && h == hh) {
                rank = cur + (unsigned)__popcll(m & below);
                if ((m >> lane) == 1ULL) vt[h] = (uint16_t)(cur + (unsigned)__popcll(m));
            }
            rem &= ~m;
        }
        LZ_WAVE_SYNC();
        if (valid) {
            Rout[i] = (uint16_t)rank;
            Sout[rank] = (uint16_t)i;
        }
    }
}

// slot range [lo, hi) of `key`'s bucket in sort block `blk`
D4G_DEV void zf_bucket(const ZfInput& in, int kind, long long blk, int key, int& lo, int& hi) {
    const uint16_t* bs = in.bstart[kind] + blk * ZF_SORT_BLOCK;
    const long long p0 = blk * ZF_SORT_BLOCK;
    const int nIns = (int)((in.n - p0) < ZF_SORT_BLOCK ? (in.n - p0) : ZF_SORT_BLOCK);
    lo = bs[key];
    hi = key == 32767 ? nIns : bs[key + 1];
    if (hi < lo) hi = nIns;      // (a bucket start of 32768 does not fit 16 bits: only the empty buckets behind a full block's single key)
}

#define ZF_MS_THREADS 256
struct ZfMatchSortedLds { uint32_t cps[ZF_MS_THREADS / 64][264]; };
__global__ void __launch_bounds__(ZF_MS_THREADS) k_zf_match_sorted(const ZfInput* inputs, const ZfMatchJob* jobs, ZfPool pool) {
    __shared__ ZfMatchSortedLds L;
    const ZfMatchJob job = jobs[blockIdx.x];
    const ZfInput in = inputs[job.input];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t* cps = L.cps[wave];
    for (int pi = wave; pi < job.count; pi += nw) {
        const long long p = job.first + pi;
        const int limit = (int)((in.n - p) < ZF_MAXM ? (in.n - p) : ZF_MAXM);
        int best = 1, bestDist = 0, ncp = 0;
        if (limit >= 3) {
            const int key1 = in.val[p], key2 = in.val2[p], sp = in.same[p];
            const long long lo = p - (ZF_WSIZE - 1) < 0 ? 0 : p - (ZF_WSIZE - 1);
            int mode = 1, hits = 0;
            // the current chain as two descending runs: slots [aLo, aHi) of sort block blkA, then [bLo, bHi) of the block before it
            long long blkA = p >> 15;
            int aLo, aHi, bLo = 0, bHi = 0;
            zf_bucket(in, 0, blkA, key1, aLo, aHi);
            aHi = in.rank[0][p];                                  // below the position's own slot
            if (blkA > 0) zf_bucket(in, 0, blkA - 1, key1, bLo, bHi);
            bool done = false;
            int taken = 0;                                         // candidates of the current chain consumed so far
            while (!done) {
                const int nA = aHi - aLo, nB = bHi - bLo;
                const int i = taken + lane;
                long long q = -1;
                if (i < nA) q = blkA * ZF_SORT_BLOCK + in.sorted[0 + (mode == 2)][blkA * ZF_SORT_BLOCK + (aHi - 1 - i)];
                else if (i < nA + nB) q = (blkA - 1) * ZF_SORT_BLOCK + in.sorted[0 + (mode == 2)][(blkA - 1) * ZF_SORT_BLOCK + (bHi - 1 - (i - nA))];
                const bool member = q >= lo;                        // (q = -1 when the chain has ended; positions descend, so the rest is out too)
                const unsigned long long memb = __ballot(member);
                if (!memb) break;
                const bool m2 = mode == 1 && member && in.val2[q] == key2;
                // lz77.c's quick reject: a candidate that differs at offset `bestlength` cannot set a record, and only records matter
                int len = 0;
                if (member && in.data[q + best] == in.data[p + best]) len = zf_match_len(in.data, p, q, limit);
                const int pm = wave_incl_max_i32(member ? len : 0);
                const int run = pm > best ? pm : best;
                const int idx = __popcll(memb & ((1ull << lane) - 1ull));
                const bool isBreak = member && len >= limit;
                const bool hitStop = member && (hits + idx + 1 >= ZF_MAX_HITS);
                const bool isSwitch = m2 && run >= sp;
                const unsigned long long stopM = __ballot(isBreak || hitStop), swM = __ballot(isSwitch);
                const int ls = stopM ? __ffsll((long long)stopM) - 1 : 64, lw = swM ? __ffsll((long long)swM) - 1 : 64;
                const int e = ls < lw ? ls : lw;
                const unsigned long long upto = e >= 63 ? ~0ull : ((2ull << e) - 1ull);
                int ex = __shfl_up(pm, 1);
                if (lane == 0) ex = 0;
                const int before = ex > best ? ex : best;
                const bool rec = member && ((upto >> lane) & 1ull) && len > before;
                const unsigned long long recM = __ballot(rec);
                if (rec) {
                    const int r = ncp + __popcll(recM & ((1ull << lane) - 1ull));
                    if (r < 264) cps[r] = ((uint32_t)len << 16) | (uint32_t)(p - q);
                }
                if (recM) {
                    const int last = 63 - __clzll((long long)recM);
                    best = __shfl(len, last);
                    bestDist = (int)__shfl((int)(p - q), last);
                    ncp += __popcll(recM);
                }
                hits += __popcll(memb & upto);
                if (ls <= lw && ls < 64) { done = true; break; }
                if (lw < 64) {
                    // the walk moves to the second chain at this node: what lies below it in its bucket of the second hash
                    const long long qs = __shfl(q, lw);
                    mode = 2;
                    blkA = qs >> 15;
                    zf_bucket(in, 1, blkA, key2, aLo, aHi);
                    aHi = in.rank[1][qs];
                    bLo = bHi = 0;
                    if (blkA > 0 && blkA == (p >> 15)) zf_bucket(in, 1, blkA - 1, key2, bLo, bHi);
                    taken = 0;
                    continue;
                }
                if (memb != ~0ull) break;                            // the chain ended inside this step
                taken += 64;
            }
        }
        LZ_WAVE_SYNC();
        int skip = 0;
        while (skip < ncp && (int)(cps[skip] >> 16) < 3) skip++;
        const int n3 = ncp - skip;
        uint32_t* ent = job.table + (long long)pi * 8;
        uint32_t link = 0;
        if (n3 > 8) {
            uint32_t at = 0;
            if (lane == 0) {
                at = atomicAdd(pool.used, (uint32_t)(n3 - 7 + 1));
                if (at + (uint32_t)(n3 - 7 + 1) > pool.cap) { atomicAdd(pool.error, 1); at = 0xffffffffu; }
            }
            at = __shfl(at, 0);
            if (at != 0xffffffffu) {
                for (int k = lane; k < n3 - 7; k += 64) pool.words[at + k] = cps[skip + 7 + k];
                if (lane == 0) pool.words[at + n3 - 7] = 0;
                link = ZF_POOL_LINK | at;
            }
        }
        if (lane < 8) {
            uint32_t v = 0;
            if (lane < n3 && (lane < 7 || n3 <= 8)) v = cps[skip + lane];
            if (lane == 7 && n3 > 8) v = link;
            ent[lane] = v;
        }
        if (lane == 0) job.best[pi] = best >= 3 ? (((uint32_t)best << 16) | (uint32_t)bestDist) : 0u;
        LZ_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// looking matches up: the shared table, or the block's tail table for the positions its end affects
// ---------------------------------------------------------------------------------------------------------------
struct ZfView {
    const uint8_t* data;
    const uint16_t* same;
    const uint32_t* table;
    const uint32_t* best;
    const uint32_t* pool;
    long long start, end;        // the block [start, end) being encoded (lz77.c's instart / inend)
    long long tailStart;         // positions >= tailStart read the tail tables (index 0 = tailStart)
    const uint32_t* tailTable;
    const uint32_t* tailBest;
};
D4G_DEV uint32_t zf_best(const ZfView& v, long long p) { return p >= v.tailStart ? v.tailBest[p - v.tailStart] : v.best[p]; }
D4G_DEV const uint32_t* zf_entry(const ZfView& v, long long p) { return p >= v.tailStart ? v.tailTable + (p - v.tailStart) * 8 : v.table + p * 8; }
// distance Zopfli's sublen[] holds for length k at a position whose entry is e[0..8) (k <= the entry's longest length)
D4G_DEV int zf_sublen(const ZfView& v, const uint32_t* e, int k) {
    for (int c = 0; c < 7; c++) { const uint32_t w = e[c]; if ((int)(w >> 16) >= k) return (int)(w & 0xffff); }
    uint32_t w = e[7];
    if (!(w & ZF_POOL_LINK)) return (int)(w & 0xffff);
    const uint32_t* q = v.pool + (w & 0x7fffffffu);
    while ((int)(*q >> 16) < k) q++;
    return (int)(*q & 0xffff);
}

// ---------------------------------------------------------------------------------------------------------------
// greedy parse with lazy matching (lz77.c ZopfliLZ77Greedy): the block splitter's input and the squeeze's first statistics
// ---------------------------------------------------------------------------------------------------------------
struct ZfGreedyJob { ZfView v; uint16_t* lit; uint16_t* dist; uint32_t* pos; uint32_t* count; };

// One wave; every lane follows the walk, lane 0 stores.  `hist` (LDS, 320 counters: 288 lit/len then 32 dist) may be null.
template <bool STORE>
D4G_DEV uint32_t zf_greedy_walk(const ZfView& v, uint32_t* gb, uint16_t* oLit, uint16_t* oDist, uint32_t* oPos, uint32_t* hist) {
    const int lane = threadIdx.x & 63;
    uint32_t n = 0;
    long long bbase = -(1LL << 40);
    unsigned prevLength = 0, prevMatch = 0;
    bool avail = false;
    auto emit = [&](int litlen, int dist, long long pos) {
        if (lane == 0) {
            if (STORE) { oLit[n] = (uint16_t)litlen; oDist[n] = (uint16_t)dist; oPos[n] = (uint32_t)pos; }
            if (hist) {
                if (dist == 0) hist[litlen]++;
                else { hist[d4g_len2sym(litlen, 0)]++; hist[ZF_NUM_LL + d4g_dist2sym(dist)]++; }
            }
        }
        n++;
    };
    for (long long i = v.start; i < v.end; i++) {
        if (i >= bbase + 256) {
            LZ_WAVE_SYNC();
            for (int k = lane; k < 256; k += 64) gb[k] = i + k < v.end ? zf_best(v, i + k) : 0u;
            bbase = i;
            LZ_WAVE_SYNC();
        }
        const uint32_t w = gb[i - bbase];
        int leng = (int)(w >> 16), dist = (int)(w & 0xffff);
        int score = dist > 1024 ? leng - 1 : leng;
        const int prevScore = prevMatch > 1024 ? (int)prevLength - 1 : (int)prevLength;
        if (avail) {
            avail = false;
            if (score > prevScore + 1) {
                emit(v.data[i - 1], 0, i - 1);
                if (score >= 3 && leng < ZF_MAXM) { avail = true; prevLength = leng; prevMatch = dist; continue; }
            } else {
                leng = (int)prevLength; dist = (int)prevMatch;
                emit(leng, dist, i - 1);
                i += leng - 2;
                continue;
            }
        } else if (score >= 3 && leng < ZF_MAXM) {
            avail = true; prevLength = leng; prevMatch = dist;
            continue;
        }
        if (score >= 3) emit(leng, dist, i);
        else { leng = 1; emit(v.data[i], 0, i); }
        i += leng - 1;
    }
    return n;
}
__global__ void __launch_bounds__(64) k_zf_greedy(const ZfGreedyJob* jobs) {
    __shared__ uint32_t gb[256];
    const ZfGreedyJob job = jobs[blockIdx.x];
    const uint32_t n = zf_greedy_walk<true>(job.v, gb, job.lit, job.dist, job.pos, nullptr);
    if (threadIdx.x == 0) *job.count = n;
}

// ---------------------------------------------------------------------------------------------------------------
// block size evaluation (deflate.c ZopfliCalculateBlockSize*, GetDynamicLengths, TryOptimizeHuffmanForRle,
// CalculateTreeSize/EncodeTree; katajainen.c ZopfliLengthLimitedCodeLengths)
// ---------------------------------------------------------------------------------------------------------------
struct ZfPmBig { uint32_t w[288]; uint16_t sym[288]; uint32_t list[2][576]; uint32_t bits[15][18]; };
struct ZfPmSmall { uint32_t w[32]; uint16_t sym[32]; uint32_t list[2][64]; uint32_t bits[15][2]; };
struct ZfClInst { uint32_t w[19]; uint32_t list[2][38]; uint32_t bits[7][2]; int32_t lvl[8]; uint16_t clc[19]; uint16_t sym[19]; uint8_t len[19]; uint8_t pad; };
struct ZfEvalLds {
    uint32_t llc[ZF_NUM_LL], dc[ZF_NUM_D];      // symbol counts of the range (no end symbol)
    uint32_t llc2[ZF_NUM_LL], dc2[ZF_NUM_D];    // the RLE-smoothed copies
    uint8_t good[ZF_NUM_LL], goodD[ZF_NUM_D];
    uint8_t ll[ZF_NUM_LL], d[ZF_NUM_D], ll2[ZF_NUM_LL], d2[ZF_NUM_D];
    int32_t lvl[2][16];
    union {
        struct { ZfPmBig big[1]; ZfPmSmall small[1]; } pm;
        ZfClInst cl[16];
        uint16_t chunk[4096];                    // the squeeze's trace window
    } u;
};
struct ZfPmRef { uint32_t* w; uint16_t* sym; uint32_t* list0; uint32_t* list1; uint32_t* bits; int stride; int m; int32_t* lvl; };

// sorted leaves (weight, then symbol) of the non-zero counts; all lanes of the wave.  The used symbols are compacted first
// (tmp: 2 n words), so the rank of each costs one pass over the used symbols, not over the alphabet.
D4G_DEV int zf_sort_leaves(const uint32_t* counts, int n, uint32_t* w, uint16_t* sym, uint32_t* tmp) {
    const int lane = threadIdx.x & 63;
    uint32_t* tc = tmp;
    uint32_t* ti = tmp + n;
    int nz = 0;
    LZ_WAVE_SYNC();
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const uint32_t c = i < n ? counts[i] : 0u;
        const unsigned long long m = __ballot(c != 0u);
        if (c) {
            const int pos = nz + __popcll(m & ((1ull << lane) - 1ull));
            tc[pos] = c;
            ti[pos] = (uint32_t)i;
        }
        nz += __popcll(m);
    }
    LZ_WAVE_SYNC();
    for (int p = lane; p < nz; p += 64) {
        const uint32_t c = tc[p];
        int rank = 0;
        for (int q = 0; q < nz; q++) { const uint32_t cq = tc[q]; rank += (cq < c || (cq == c && q < p)) ? 1 : 0; }
        w[rank] = c;
        sym[rank] = (uint16_t)ti[p];
    }
    LZ_WAVE_SYNC();
    return nz;
}
// Package-merge, one lane: level l's list = leaves merged with the pairs of level l-1's list, a pair before a leaf of
// equal weight (BoundaryPM's `sum > leaf weight` test); the code length of the r-th lightest leaf = the number of
// levels whose selected prefix holds more than r leaves (ExtractBitLengths).  `out` (indexed by symbol) must be zero.
D4G_DEV void zf_pm_serial(const ZfPmRef& r, int maxbits, uint8_t* out) {
    const int m = r.m;
    if (m == 0) return;
    if (m == 1) { out[r.sym[0]] = 1; return; }
    if (m == 2) { out[r.sym[0]] = 1; out[r.sym[1]] = 1; return; }
    const int mb = maxbits < m - 1 ? maxbits : m - 1;
    uint32_t* prev = r.list0;
    uint32_t* cur = r.list1;
    for (int i = 0; i < m; i++) prev[i] = r.w[i];
    int lenPrev = m;
    for (int l = 1; l < mb; l++) {
        uint32_t* bits = r.bits + l * r.stride;
        for (int k = 0; k < r.stride; k++) bits[k] = 0;
        const int np = lenPrev >> 1;
        int i = 0, k = 0, cnt = 0;
        while (i < m || k < np) {
            bool takeP = false;
            uint32_t pk = 0;
            if (k < np) { pk = prev[2 * k] + prev[2 * k + 1]; takeP = i >= m || pk <= r.w[i]; }
            if (takeP) { cur[cnt] = pk; k++; }
            else { cur[cnt] = r.w[i]; i++; bits[cnt >> 5] |= 1u << (cnt & 31); }
            cnt++;
        }
        lenPrev = cnt;
        uint32_t* t = prev; prev = cur; cur = t;
    }
    int t = 2 * m - 2;
    for (int l = mb - 1; l >= 1; l--) {
        const uint32_t* bits = r.bits + l * r.stride;
        int a = 0;
        for (int k = 0; k * 32 < t; k++) {
            uint32_t wd = bits[k];
            if (t - k * 32 < 32) wd &= (1u << (t - k * 32)) - 1u;
            a += __popcll((unsigned long long)wd);
        }
        r.lvl[l] = a;
        t = 2 * (t - a);
    }
    r.lvl[0] = t;
    for (int q = 0; q < m; q++) {
        int len = 0;
        for (int l = 0; l < mb; l++) len += r.lvl[l] > q ? 1 : 0;
        out[r.sym[q]] = (uint8_t)len;
    }
}
// The same package-merge by a whole wave: a level's merged order comes from two binary searches per item (a leaf goes behind
// the pairs that weigh at most as much, a pair behind the leaves that weigh strictly less), the leaf bits are OR-ed into the
// level's mask, and the top-down pass counts mask prefixes with popcounts.  All 64 lanes call it with the same arguments.
D4G_DEV void zf_pm_wave(const ZfPmRef& r, int maxbits, uint8_t* out) {
    const int lane = threadIdx.x & 63;
    const int m = r.m;
    if (m == 0) return;
    if (m <= 2) { if (lane < m) out[r.sym[lane]] = 1; LZ_WAVE_SYNC(); return; }
    const int mb = maxbits < m - 1 ? maxbits : m - 1;
    uint32_t* prev = r.list0;
    uint32_t* cur = r.list1;
    for (int i = lane; i < m; i += 64) prev[i] = r.w[i];
    int lenPrev = m;
    LZ_WAVE_SYNC();
    for (int l = 1; l < mb; l++) {
        uint32_t* bits = r.bits + l * r.stride;
        for (int k = lane; k < r.stride; k += 64) bits[k] = 0;
        LZ_WAVE_SYNC();
        const int np = lenPrev >> 1;
        for (int i = lane; i < m; i += 64) {
            const uint32_t x = r.w[i];
            int lo = 0, hi = np;                       // first pair heavier than the leaf
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (prev[2 * mid] + prev[2 * mid + 1] <= x) lo = mid + 1; else hi = mid; }
            const int pos = i + lo;
            cur[pos] = x;
            atomicOr(&bits[pos >> 5], 1u << (pos & 31));
        }
        for (int k = lane; k < np; k += 64) {
            const uint32_t x = prev[2 * k] + prev[2 * k + 1];
            int lo = 0, hi = m;                        // first leaf at least as heavy as the pair
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (r.w[mid] < x) lo = mid + 1; else hi = mid; }
            cur[k + lo] = x;
        }
        lenPrev = m + np;
        uint32_t* t = prev; prev = cur; cur = t;
        LZ_WAVE_SYNC();
    }
    int t = 2 * m - 2;
    for (int l = mb - 1; l >= 1; l--) {
        const uint32_t* bits = r.bits + l * r.stride;
        int part = 0;
        for (int k = lane; k * 32 < t; k += 64) {
            uint32_t wd = bits[k];
            if (t - k * 32 < 32) wd &= (1u << (t - k * 32)) - 1u;
            part += __popcll((unsigned long long)wd);
        }
        const int a = wave_sum_i32(part);
        if (lane == 0) r.lvl[l] = a;
        t = 2 * (t - a);
    }
    if (lane == 0) r.lvl[0] = t;
    LZ_WAVE_SYNC();
    for (int q = lane; q < m; q += 64) {
        int len = 0;
        for (int l = 0; l < mb; l++) len += r.lvl[l] > q ? 1 : 0;
        out[r.sym[q]] = (uint8_t)len;
    }
    LZ_WAVE_SYNC();
}
D4G_DEV void zf_patch_dist(uint8_t* d) {   // PatchDistanceCodesForBuggyDecoders
    int num = 0;
    for (int i = 0; i < 30; i++) { if (d[i]) num++; if (num >= 2) return; }
    if (num == 0) d[0] = d[1] = 1;
    else if (num == 1) d[d[0] ? 1 : 0] = 1;
}
D4G_DEV void zf_optimize_rle(int length, uint32_t* counts, uint8_t* good) {   // OptimizeHuffmanForRle, one lane
    for (; length >= 0; --length) {
        if (length == 0) return;
        if (counts[length - 1] != 0) break;
    }
    for (int i = 0; i < length; i++) good[i] = 0;
    uint32_t symbol = counts[0];
    int stride = 0;
    for (int i = 0; i < length + 1; ++i) {
        if (i == length || counts[i] != symbol) {
            if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
                for (int k = 0; k < stride; ++k) good[i - k - 1] = 1;
            stride = 1;
            if (i != length) symbol = counts[i];
        } else ++stride;
    }
    stride = 0;
    uint32_t limit = counts[0];
    unsigned long long sum = 0;
    for (int i = 0; i < length + 1; ++i) {
        uint32_t ad = 0;
        if (i != length) ad = counts[i] > limit ? counts[i] - limit : limit - counts[i];
        if (i == length || good[i] || ad >= 4) {
            if (stride >= 4 || (stride >= 3 && sum == 0)) {
                int count = (int)((sum + (unsigned)(stride / 2)) / (unsigned)stride);
                if (count < 1) count = 1;
                if (sum == 0) count = 0;
                for (int k = 0; k < stride; ++k) counts[i - k - 1] = (uint32_t)count;
            }
            stride = 0;
            sum = 0;
            if (i < length - 3) limit = (uint32_t)(((unsigned long long)counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4);
            else if (i < length) limit = counts[i];
            else limit = 0;
        }
        ++stride;
        if (i != length) sum += counts[i];
    }
}
// EncodeTree's size for one use_16/17/18 combination, one lane (clc / PM scratch in `c`)
D4G_DEV int zf_tree_size_one(const uint8_t* ll, const uint8_t* d, int combo, ZfClInst& c) {
    const bool use16 = combo & 1, use17 = combo & 2, use18 = combo & 4;
    int hlit = 29, hdist = 29;
    for (int i = 0; i < 19; i++) c.clc[i] = 0;
    while (hlit > 0 && ll[257 + hlit - 1] == 0) hlit--;
    while (hdist > 0 && d[1 + hdist - 1] == 0) hdist--;
    const int hlit2 = hlit + 257, total = hlit2 + hdist + 1;
    for (int i = 0; i < total; i++) {
        const int symbol = i < hlit2 ? ll[i] : d[i - hlit2];
        int count = 1;
        if (use16 || (symbol == 0 && (use17 || use18)))
            for (int j = i + 1; j < total && symbol == (j < hlit2 ? ll[j] : d[j - hlit2]); j++) count++;
        i += count - 1;
        if (symbol == 0 && count >= 3) {
            if (use18) while (count >= 11) { const int c2 = count > 138 ? 138 : count; c.clc[18]++; count -= c2; }
            if (use17) while (count >= 3) { const int c2 = count > 10 ? 10 : count; c.clc[17]++; count -= c2; }
        }
        if (use16 && count >= 4) {
            count--;
            c.clc[symbol]++;
            while (count >= 3) { const int c2 = count > 6 ? 6 : count; c.clc[16]++; count -= c2; }
        }
        c.clc[symbol] += (uint16_t)count;
    }
    // 19-symbol code, limit 7
    int m = 0;
    for (int i = 0; i < 19; i++) {
        c.len[i] = 0;
        if (!c.clc[i]) continue;
        int k = m++;
        while (k > 0 && c.w[k - 1] > c.clc[i]) { c.w[k] = c.w[k - 1]; c.sym[k] = c.sym[k - 1]; k--; }   // stable: equal weights keep symbol order
        c.w[k] = c.clc[i]; c.sym[k] = (uint16_t)i;
    }
    ZfPmRef r = {c.w, c.sym, c.list[0], c.list[1], &c.bits[0][0], 2, m, c.lvl};
    zf_pm_serial(r, 7, c.len);
    int hclen = 15;
    while (hclen > 0 && c.clc[D4G_CL_ORDER[hclen + 4 - 1]] == 0) hclen--;
    int size = 14 + (hclen + 4) * 3;
    for (int i = 0; i < 19; i++) size += c.len[i] * c.clc[i];
    size += c.clc[16] * 2 + c.clc[17] * 3 + c.clc[18] * 7;
    return size;
}
D4G_DEV long long zf_data_size(const uint32_t* llc, const uint32_t* dc, const uint8_t* ll, const uint8_t* d) {   // all lanes
    const int lane = threadIdx.x & 63;
    long long s = 0;
    for (int i = lane; i < 286; i += 64) {
        if (i == 256) continue;
        s += (long long)(ll[i] + (i > 256 ? d4g_lsym_ebits(i) : 0)) * llc[i];
    }
    if (lane < 30) s += (long long)(d[lane] + d4g_dsym_ebits(lane)) * dc[lane];
    return wave_sum_i64(s) + ll[256];
}
// GetDynamicLengths: E.llc / E.dc hold the range's counts with llc[256] = 1.  Leaves the chosen lengths in E.ll / E.d, the
// header's best use_16/17/18 combination in *combo, and returns tree size + data size.  All lanes of one wave.
D4G_DEV long long zf_dynamic_lengths(ZfEvalLds& E, int* combo) {
    const int lane = threadIdx.x & 63;
    LZ_WAVE_SYNC();
    for (int i = lane; i < ZF_NUM_LL; i += 64) { E.llc2[i] = E.llc[i]; E.ll[i] = 0; E.ll2[i] = 0; }
    if (lane < ZF_NUM_D) { E.dc2[lane] = E.dc[lane]; E.d[lane] = 0; E.d2[lane] = 0; }
    LZ_WAVE_SYNC();
    if (lane == 2) zf_optimize_rle(ZF_NUM_LL, E.llc2, E.good);
    if (lane == 3) zf_optimize_rle(ZF_NUM_D, E.dc2, E.goodD);
    LZ_WAVE_SYNC();
    // two passes over one set of package-merge scratch: the plain counts' trees (lit/len on lane 0, distance on lane 1), then the
    // smoothed counts' (a second set would let all four run at once, but the scratch is what bounds the waves per CU)
    for (int pass = 0; pass < 2; pass++) {
        const int mBig = zf_sort_leaves(pass ? E.llc2 : E.llc, ZF_NUM_LL, E.u.pm.big[0].w, E.u.pm.big[0].sym, E.u.pm.big[0].list[1]);
        const int mSmall = zf_sort_leaves(pass ? E.dc2 : E.dc, ZF_NUM_D, E.u.pm.small[0].w, E.u.pm.small[0].sym, E.u.pm.small[0].list[1]);
        LZ_WAVE_SYNC();
#ifdef D4G_HOSTSIM
        // (the CPU emulation keeps its test time down with the one-lane builder here; the wave-wide one is compared with the oracle
        // through d4g_debug_zopfli_code_lengths in the same test, and on the GPU every stream is)
        if (lane < 2) {
            ZfPmRef r;
            uint8_t* out;
            if (lane == 0) { r = {E.u.pm.big[0].w, E.u.pm.big[0].sym, E.u.pm.big[0].list[0], E.u.pm.big[0].list[1], &E.u.pm.big[0].bits[0][0], 18, mBig, E.lvl[0]}; out = pass ? E.ll2 : E.ll; }
            else { r = {E.u.pm.small[0].w, E.u.pm.small[0].sym, E.u.pm.small[0].list[0], E.u.pm.small[0].list[1], &E.u.pm.small[0].bits[0][0], 2, mSmall, E.lvl[1]}; out = pass ? E.d2 : E.d; }
            zf_pm_serial(r, 15, out);
        }
#else
        {
            ZfPmRef rb = {E.u.pm.big[0].w, E.u.pm.big[0].sym, E.u.pm.big[0].list[0], E.u.pm.big[0].list[1], &E.u.pm.big[0].bits[0][0], 18, mBig, E.lvl[0]};
            ZfPmRef rs = {E.u.pm.small[0].w, E.u.pm.small[0].sym, E.u.pm.small[0].list[0], E.u.pm.small[0].list[1], &E.u.pm.small[0].bits[0][0], 2, mSmall, E.lvl[1]};
            zf_pm_wave(rb, 15, pass ? E.ll2 : E.ll);
            zf_pm_wave(rs, 15, pass ? E.d2 : E.d);
        }
#endif
        LZ_WAVE_SYNC();
    }
    LZ_WAVE_SYNC();
    if (lane == 0) zf_patch_dist(E.d);
    if (lane == 1) zf_patch_dist(E.d2);
    LZ_WAVE_SYNC();
    int ts = 0x7fffffff;
    if (lane < 16) {
        const bool second = lane >= 8;
        ts = zf_tree_size_one(second ? E.ll2 : E.ll, second ? E.d2 : E.d, lane & 7, E.u.cl[lane]);
    }
    LZ_WAVE_SYNC();
    int k1 = lane < 8 ? ((ts << 4) | lane) : 0x7fffffff, k2 = (lane >= 8 && lane < 16) ? ((ts << 4) | (lane & 7)) : 0x7fffffff;
    k1 = -wave_max_i32(-k1);
    k2 = -wave_max_i32(-k2);
    const long long d1 = zf_data_size(E.llc, E.dc, E.ll, E.d), d2 = zf_data_size(E.llc, E.dc, E.ll2, E.d2);
    const long long c1 = (k1 >> 4) + d1, c2 = (k2 >> 4) + d2;
    LZ_WAVE_SYNC();
    if (c2 < c1) {
        for (int i = lane; i < ZF_NUM_LL; i += 64) E.ll[i] = E.ll2[i];
        if (lane < ZF_NUM_D) E.d[lane] = E.d2[lane];
        LZ_WAVE_SYNC();
        if (combo) *combo = k2 & 7;
        return c2;
    }
    if (combo) *combo = k1 & 7;
    return c1;
}

// ZopfliCalculateBlockSize for the three block types.  E.llc / E.dc = the range's counts (llc[256] is set to 1 here);
// byteLen = the bytes the range decodes to.  Costs are whole bits, so integers stand in for the published doubles.
D4G_DEV void zf_block_costs(ZfEvalLds& E, long long byteLen, bool wantFixed, long long& unc, long long& fixedc, long long& dyn, int* combo) {
    const int lane = threadIdx.x & 63;
    const long long rem = byteLen % 65535, blocks = byteLen / 65535 + (rem ? 1 : 0);
    unc = blocks * 5 * 8 + byteLen * 8;
    fixedc = unc;
    if (wantFixed) {
        long long s = 0;
        for (int i = lane; i < 286; i += 64) {
            if (i == 256) continue;
            const int bl = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
            s += (long long)(bl + (i > 256 ? d4g_lsym_ebits(i) : 0)) * E.llc[i];
        }
        if (lane < 30) s += (long long)(5 + d4g_dsym_ebits(lane)) * E.dc[lane];
        fixedc = 3 + wave_sum_i64(s) + 7;
    }
    LZ_WAVE_SYNC();
    if (lane == 0) E.llc[256] = 1;
    LZ_WAVE_SYNC();
    dyn = 3 + zf_dynamic_lengths(E, combo);
}
// ZopfliCalculateBlockSizeAutoType: the fixed tree is only priced for stores of at most 1000 symbols
D4G_DEV long long zf_block_cost_auto(ZfEvalLds& E, long long byteLen, uint32_t storeSize) {
    long long unc, fixedc, dyn;
    zf_block_costs(E, byteLen, storeSize <= 1000, unc, fixedc, dyn, nullptr);
    return (unc < fixedc && unc < dyn) ? unc : (fixedc < dyn ? fixedc : dyn);
}
// counts of the store symbols [a, b) into E.llc / E.dc; all lanes of one wave
struct ZfStore { const uint16_t* lit; const uint16_t* dist; const uint32_t* pos; uint32_t size; };
D4G_DEV void zf_count_range(ZfEvalLds& E, const ZfStore& s, uint32_t a, uint32_t b) {
    const int lane = threadIdx.x & 63;
    LZ_WAVE_SYNC();
    for (int i = lane; i < ZF_NUM_LL; i += 64) E.llc[i] = 0;
    if (lane < ZF_NUM_D) E.dc[lane] = 0;
    LZ_WAVE_SYNC();
    for (uint32_t i = a + lane; i < b; i += 64) {
        const int dd = s.dist[i], l = s.lit[i];
        if (dd == 0) atomicAdd(&E.llc[l], 1u);
        else { atomicAdd(&E.llc[d4g_len2sym(l, 0)], 1u); atomicAdd(&E.dc[d4g_dist2sym(dd)], 1u); }
    }
    LZ_WAVE_SYNC();
}
D4G_DEV long long zf_byte_range(const ZfStore& s, uint32_t a, uint32_t b) {   // ZopfliLZ77GetByteRange
    if (a == b) return 0;
    const uint32_t l = b - 1;
    return (long long)s.pos[l] + (s.dist[l] == 0 ? 1 : s.lit[l]) - (long long)s.pos[a];
}

// cost of a list of store ranges (one wave each): the totals the two splitting attempts compare, and the final blocks' three costs
struct ZfRangeJob { ZfStore s; uint32_t a, b; };
struct ZfRangeOut { long long unc, fixedc, dyn, autoCost, firstPos, byteLen; };
__global__ void __launch_bounds__(64) k_zf_range_cost(const ZfRangeJob* jobs, ZfRangeOut* out) {
    __shared__ ZfEvalLds E;
    const ZfRangeJob job = jobs[blockIdx.x];
    zf_count_range(E, job.s, job.a, job.b);
    long long unc, fixedc, dyn;
    const long long bytes = zf_byte_range(job.s, job.a, job.b);
    zf_block_costs(E, bytes, true, unc, fixedc, dyn, nullptr);
    if (threadIdx.x == 0) {
        const long long f2 = job.s.size > 1000 ? unc : fixedc;
        out[blockIdx.x] = {unc, fixedc, dyn, (unc < f2 && unc < dyn) ? unc : (f2 < dyn ? f2 : dyn), job.a < job.b ? (long long)job.s.pos[job.a] : 0, bytes};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// block splitting (blocksplitter.c ZopfliBlockSplitLZ77: FindMinimum over SplitCost, largest splittable block next)
// ---------------------------------------------------------------------------------------------------------------
#define ZF_SPLIT_WAVES 5
struct ZfSplitJob {
    ZfStore s;
    uint32_t maxblocks;      // 0 = unlimited
    uint32_t cap;            // capacity of points[]
    uint8_t* done;           // s.size bytes, zero
    uint32_t* points;        // out: split points (store indices), ascending
    uint32_t* bytePos;       // out: the byte position each point starts at
    uint32_t* npoints;
    int32_t* error;
};
struct ZfSplitLds {
    ZfEvalLds E[ZF_SPLIT_WAVES];
    uint32_t seg[10][ZF_NUM_LL + ZF_NUM_D];
    uint32_t HL[ZF_NUM_LL + ZF_NUM_D], HR[ZF_NUM_LL + ZF_NUM_D];
    long long vp[9];
    long long waveBest[ZF_SPLIT_WAVES];
    uint32_t waveBestAt[ZF_SPLIT_WAVES];
    uint32_t p[9];
    uint32_t lstart, lend, fs, fe, pos, numblocks, np;
    long long lastbest, origcost;
    int32_t besti, stop, outer;
};
D4G_DEV void zf_sym_of(const ZfStore& s, uint32_t i, int& a, int& b) {   // histogram slots of store symbol i (b = -1: literal)
    const int dd = s.dist[i], l = s.lit[i];
    if (dd == 0) { a = l; b = -1; }
    else { a = d4g_len2sym(l, 0); b = ZF_NUM_LL + d4g_dist2sym(dd); }
}
__global__ void __launch_bounds__(ZF_SPLIT_WAVES * 64) k_zf_split(const ZfSplitJob* jobs) {
    __shared__ ZfSplitLds L;
    const ZfSplitJob job = jobs[blockIdx.x];
    const ZfStore s = job.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NH = ZF_NUM_LL + ZF_NUM_D;
    if (s.size < 10) { if (tid == 0) *job.npoints = 0; return; }
    if (tid == 0) { L.lstart = 0; L.lend = s.size; L.numblocks = 1; L.np = 0; L.outer = 1; }
    __syncthreads();
    ZfEvalLds& E = L.E[wave];
    while (true) {
        if (tid == 0 && job.maxblocks > 0 && L.numblocks >= job.maxblocks) L.outer = 0;
        __syncthreads();
        if (!L.outer) break;
        const uint32_t lstart = L.lstart, lend = L.lend;
        // ---- FindMinimum(SplitCost, lstart + 1, lend) ----
        if (lend - (lstart + 1) < 1024) {
            long long best = 0x7fffffffffffffffLL;
            uint32_t bestAt = lstart + 1;
            for (uint32_t i = lstart + 1 + wave; i < lend; i += ZF_SPLIT_WAVES) {
                zf_count_range(E, s, lstart, i);
                long long v = zf_block_cost_auto(E, zf_byte_range(s, lstart, i), s.size);
                zf_count_range(E, s, i, lend);
                v += zf_block_cost_auto(E, zf_byte_range(s, i, lend), s.size);
                if (v < best) { best = v; bestAt = i; }
            }
            if (lane == 0) { L.waveBest[wave] = best; L.waveBestAt[wave] = bestAt; }
            __syncthreads();
            if (tid == 0) {
                long long b = 0x7fffffffffffffffLL;
                uint32_t at = lstart + 1;
                for (int w = 0; w < ZF_SPLIT_WAVES; w++)
                    if (L.waveBest[w] < b || (L.waveBest[w] == b && L.waveBest[w] != 0x7fffffffffffffffLL && L.waveBestAt[w] < at)) { b = L.waveBest[w]; at = L.waveBestAt[w]; }
                L.pos = at; L.lastbest = b;
            }
            __syncthreads();
        } else {
            if (tid == 0) { L.fs = lstart + 1; L.fe = lend; L.lastbest = 0x7fffffffffffffffLL; L.pos = lstart + 1; L.stop = 0; }
            for (int k = tid; k < NH; k += blockDim.x) { L.HL[k] = 0; L.HR[k] = 0; }
            __syncthreads();
            if (tid == 0) { int a, b; zf_sym_of(s, lstart, a, b); L.HL[a]++; if (b >= 0) L.HL[b]++; }   // [lstart, lstart + 1)
            __syncthreads();
            while (true) {
                const uint32_t fs = L.fs, fe = L.fe;
                if (fe - fs <= 9) break;
                if (tid < 9) L.p[tid] = fs + (uint32_t)(tid + 1) * ((fe - fs) / 10);
                for (int k = tid; k < 10 * NH; k += blockDim.x) (&L.seg[0][0])[k] = 0;
                __syncthreads();
                for (uint32_t x = fs + tid; x < fe; x += blockDim.x) {
                    int g = 0;
                    for (int q = 0; q < 9; q++) g += L.p[q] <= x ? 1 : 0;
                    int a, b;
                    zf_sym_of(s, x, a, b);
                    atomicAdd(&L.seg[g][a], 1u);
                    if (b >= 0) atomicAdd(&L.seg[g][b], 1u);
                }
                __syncthreads();
                for (int pi = wave; pi < 9; pi += ZF_SPLIT_WAVES) {
                    const uint32_t at = L.p[pi];
                    LZ_WAVE_SYNC();
                    for (int k = lane; k < NH; k += 64) {
                        uint32_t v = L.HL[k];
                        for (int g = 0; g <= pi; g++) v += L.seg[g][k];
                        if (k < ZF_NUM_LL) E.llc[k] = v; else E.dc[k - ZF_NUM_LL] = v;
                    }
                    LZ_WAVE_SYNC();
                    long long v = zf_block_cost_auto(E, zf_byte_range(s, lstart, at), s.size);
                    LZ_WAVE_SYNC();
                    for (int k = lane; k < NH; k += 64) {
                        uint32_t c = L.HR[k];
                        for (int g = pi + 1; g < 10; g++) c += L.seg[g][k];
                        if (k < ZF_NUM_LL) E.llc[k] = c; else E.dc[k - ZF_NUM_LL] = c;
                    }
                    LZ_WAVE_SYNC();
                    v += zf_block_cost_auto(E, zf_byte_range(s, at, lend), s.size);
                    if (lane == 0) L.vp[pi] = v;
                }
                __syncthreads();
                if (tid == 0) {
                    int besti = 0;
                    long long best = L.vp[0];
                    for (int q = 1; q < 9; q++) if (L.vp[q] < best) { best = L.vp[q]; besti = q; }
                    if (best > L.lastbest) L.stop = 1;
                    else { L.besti = besti; L.pos = L.p[besti]; L.lastbest = best; }
                }
                __syncthreads();
                if (L.stop) break;
                const int besti = L.besti;
                for (int k = tid; k < NH; k += blockDim.x) {
                    uint32_t a = 0, b = 0;
                    for (int g = 0; g < besti; g++) a += L.seg[g][k];           // segments left of the new start p[besti - 1]
                    for (int g = besti + 2; g < 10; g++) b += L.seg[g][k];      // segments right of the new end p[besti + 1]
                    L.HL[k] += a; L.HR[k] += b;
                }
                __syncthreads();
                if (tid == 0) {
                    const uint32_t nfs = besti == 0 ? fs : L.p[besti - 1], nfe = besti == 8 ? fe : L.p[besti + 1];
                    L.fs = nfs; L.fe = nfe;
                }
                __syncthreads();
            }
            __syncthreads();
        }
        // ---- the block's cost unsplit, then the decision ----
        if (wave == 0) {
            zf_count_range(E, s, lstart, lend);
            const long long oc = zf_block_cost_auto(E, zf_byte_range(s, lstart, lend), s.size);
            if (lane == 0) L.origcost = oc;
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t llpos = L.pos;
            if (L.lastbest > L.origcost || llpos == lstart + 1 || llpos == lend) job.done[lstart] = 1;
            else if (L.np >= job.cap) { atomicAdd(job.error, 1); L.outer = 0; }
            else {
                uint32_t k = L.np;
                while (k > 0 && job.points[k - 1] > llpos) { job.points[k] = job.points[k - 1]; k--; }
                job.points[k] = llpos;
                L.np++;
                L.numblocks++;
            }
            // FindLargestSplittableBlock
            uint32_t longest = 0;
            bool found = false;
            for (uint32_t i = 0; i <= L.np; i++) {
                const uint32_t st = i == 0 ? 0 : job.points[i - 1], en = i == L.np ? s.size - 1 : job.points[i];
                if (!job.done[st] && en - st > longest) { L.lstart = st; L.lend = en; found = true; longest = en - st; }
            }
            if (!found || L.lend - L.lstart < 10) L.outer = 0;
        }
        __syncthreads();
    }
    if (tid == 0) *job.npoints = L.np;
    __syncthreads();
    for (uint32_t k = tid; k < L.np; k += blockDim.x) job.bytePos[k] = s.pos[job.points[k]];
}

// ---------------------------------------------------------------------------------------------------------------
// squeeze (squeeze.c ZopfliLZ77Optimal / ZopfliLZ77OptimalFixed): one wave per block
// ---------------------------------------------------------------------------------------------------------------
D4G_DEV double zf_u2d(unsigned long long u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
D4G_DEV unsigned long long zf_d2u(double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; }
// oracle/zopfli_oracle.c zopf_portable_log, operation for operation
D4G_DEV double zf_log(double x) {
    unsigned long long u = zf_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = zf_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 27.0;
    p = p * z + 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
struct ZfSqJob {
    ZfView v;
    uint16_t* lit[2]; uint16_t* dist[2]; uint32_t* pos[2];   // two store buffers (capacity end - start): the run being made and the best so far
    uint16_t* lengthArray;                                   // end - start + 1
    uint32_t* path;                                          // end - start
    int32_t iterations;
    int32_t fixedModel;                                      // 1: one run with the fixed tree's costs (ZopfliLZ77OptimalFixed) into buffer 0
};
struct ZfSqOut { int32_t bestBuf; uint32_t bestSize; long long bestCost; long long cyc[6]; };   // cyc: greedy, DP, trace+follow, cost, statistics (device clock ticks)
struct ZfSqLds {
    ZfEvalLds E;
    double llTab[260];              // per match length: the cost of its length symbol (this iteration's model)
    double cpDc[64][8];             //   per change point: the cost of its distance symbol,
    uint8_t cpDb[64][8];            //   its distance extra bits
    uint8_t lbTab[260];             // per match length: length extra bits (+ the fixed tree's code lengths in the fixed model)
    double llsym[ZF_NUM_LL], dsym[ZF_NUM_D];
    uint32_t f[ZF_NUM_LL + ZF_NUM_D], fbest[ZF_NUM_LL + ZF_NUM_D], flast[ZF_NUM_LL + ZF_NUM_D];
};
D4G_DEV double zf_model(const ZfSqLds& S, bool fixedModel, int k, int dist) {   // GetCostStat / GetCostFixed for a match
    const int lsym = d4g_len2sym(k, 0), lbits = d4g_lsym_ebits(lsym), dsym = d4g_dist2sym(dist), dbits = d4g_dsym_ebits(dsym);
    if (fixedModel) return (double)((lsym <= 279 ? 7 : 8) + 5 + dbits + lbits);
    return (double)(lbits + dbits) + S.llsym[lsym] + S.dsym[dsym];
}
D4G_DEV double zf_model_lit(const ZfSqLds& S, bool fixedModel, int c) { return fixedModel ? (c <= 143 ? 8.0 : 9.0) : S.llsym[c]; }
D4G_DEV void zf_calc_entropy(const uint32_t* count, int n, double* out) {   // ZopfliCalculateEntropy; all lanes
    const int lane = threadIdx.x & 63;
    int part = 0;
    for (int i = lane; i < n; i += 64) part += (int)count[i];
    const uint32_t sum = (uint32_t)wave_sum_i32(part);
    const double kInvLog2 = 1.4426950408889;
    const double log2sum = (sum == 0 ? zf_log((double)n) : zf_log((double)sum)) * kInvLog2;
    for (int i = lane; i < n; i += 64) {
        double b = count[i] == 0 ? log2sum : log2sum - zf_log((double)count[i]) * kInvLog2;
        if (b < 0 && b > -1e-5) b = 0;
        out[i] = b;
    }
}
D4G_DEV void zf_calc_stats(ZfSqLds& S) {
    LZ_WAVE_SYNC();
    zf_calc_entropy(S.f, ZF_NUM_LL, S.llsym);
    zf_calc_entropy(S.f + ZF_NUM_LL, ZF_NUM_D, S.dsym);
    LZ_WAVE_SYNC();
}
struct ZfRan { uint32_t w, z; };
D4G_DEV uint32_t zf_ran(ZfRan& r) {
    r.z = 36969u * (r.z & 65535u) + (r.z >> 16);
    r.w = 18000u * (r.w & 65535u) + (r.w >> 16);
    return (r.z << 16) + r.w;
}
// first minimum of (value, index) over the wave
D4G_DEV void zf_wave_argmin(double& v, int& idx) {
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m);
        const int oi = __shfl_xor(idx, m);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
D4G_DEV double zf_min_cost(const ZfSqLds& S, bool fixedModel) {   // GetCostModelMinCost
    const int lane = threadIdx.x & 63;
    double bv = 1e30;
    int bk = 1 << 20;
    for (int k = 3 + lane; k < 259; k += 64) { const double c = zf_model(S, fixedModel, k, 1); if (c < bv) { bv = c; bk = k; } }
    zf_wave_argmin(bv, bk);
    const int bestlength = bv < 1e30 ? bk : 0;
    double dv = 1e30;
    int di = 1 << 20;
    if (lane < 30) { dv = zf_model(S, fixedModel, 3, d4g_dsym_base(lane)); di = lane; }
    zf_wave_argmin(dv, di);
    const int bestdist = dv < 1e30 ? d4g_dsym_base(di) : 0;
    return zf_model(S, fixedModel, bestlength, bestdist);
}

// GetBestLengths: fills job.lengthArray[0 .. blocksize].
//
// The costs of the next 640 positions live in registers: lane L, register q holds index B + 64 q + L (B = the 64-aligned
// base of the window).  The position being expanded reads its own cost with v_readlane, every lane relaxes the targets it
// owns (the literal's and the match lengths'), and once per 64 positions the finished lengths are written out and the
// registers shift.  Everything that does not depend on the running cost — the literal's cost, the change points with the
// distance part of their cost, the long-run flag — is prepared for 64 positions at a time by all lanes.
D4G_DEV uint32_t zf_rl(uint32_t v, int l) {
#ifdef D4G_HOSTSIM
    return __shfl(v, l);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
#endif
}
D4G_DEV float zf_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
D4G_DEV uint32_t zf_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
#define ZF_NQ 10
// Change points of a position: up to eight match lengths (ascending, at most 258), two to a word; an unused place holds ZF_NOCP.
// zf_count_below = how many of them are below k: both halves of a word at once with the packed 16-bit instructions (a place is
// below k when place - k is negative; nothing overflows: every value is below 2^15).
#define ZF_NOCP 0x7fffu
#define ZF_NOCP2 0x7fff7fffu
D4G_DEV int zf_count_below(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int k) {
#ifdef D4G_HOSTSIM
    return ((int)(c0 & 0xffff) < k) + ((int)(c0 >> 16) < k) + ((int)(c1 & 0xffff) < k) + ((int)(c1 >> 16) < k) +
           ((int)(c2 & 0xffff) < k) + ((int)(c2 >> 16) < k) + ((int)(c3 & 0xffff) < k) + ((int)(c3 >> 16) < k);
#else
    typedef short zf_s2 __attribute__((ext_vector_type(2)));
    const zf_s2 kk = {(short)k, (short)k};
    zf_s2 a, b, c, d;
    __builtin_memcpy(&a, &c0, 4); __builtin_memcpy(&b, &c1, 4); __builtin_memcpy(&c, &c2, 4); __builtin_memcpy(&d, &c3, 4);
    const zf_s2 t = ((a - kk) >> 15) + ((b - kk) >> 15) + ((c - kk) >> 15) + ((d - kk) >> 15);   // -1 per place below k
    return -((int)t.x + (int)t.y);
#endif
}
D4G_DEV void zf_best_lengths(ZfSqLds& S, const ZfSqJob& job, bool fixedModel) {
    const ZfView& v = job.v;
    const int lane = threadIdx.x & 63;
    const long long start = v.start, end = v.end;
    const int size = (int)(end - start);
    LZ_WAVE_SYNC();
    for (int k = 3 + lane; k < 259; k += 64) {
        const int lsym = d4g_len2sym(k, 0), lb = d4g_lsym_ebits(lsym);
        S.lbTab[k] = (uint8_t)(fixedModel ? (lsym <= 279 ? 7 : 8) + 5 + lb : lb);
        S.llTab[k] = fixedModel ? 0.0 : S.llsym[lsym];
    }
    const double mincost = zf_min_cost(S, fixedModel);
    LZ_WAVE_SYNC();
    if (lane == 0) job.lengthArray[0] = 0;
    float C[ZF_NQ];
    uint32_t Ln[ZF_NQ];
#pragma unroll
    for (int q = 0; q < ZF_NQ; q++) { C[q] = 1e30f; Ln[q] = 0; }
    if (lane == 0) C[0] = 0.f;
    int B = 0;                  // index of register 0, lane 0
    int loaded = -64;           // base the batch registers were prepared for
    uint32_t rMeta = 0, rCl0 = ZF_NOCP2, rCl1 = ZF_NOCP2, rCl2 = ZF_NOCP2, rCl3 = ZF_NOCP2, rLitLo = 0, rLitHi = 0;
    bool afterShortcut = false;
    int j = 0;
    while (j < size) {
        while (j >= B + 64) {   // the window moves on: 64 lengths are final
            const int x = B + lane;
            if (x > 0 && x <= size) job.lengthArray[x] = (uint16_t)Ln[0];
#pragma unroll
            for (int q = 0; q + 1 < ZF_NQ; q++) { C[q] = C[q + 1]; Ln[q] = Ln[q + 1]; }
            C[ZF_NQ - 1] = 1e30f; Ln[ZF_NQ - 1] = 0;
            B += 64;
        }
        if (loaded != B) {
            LZ_WAVE_SYNC();
            const long long p = start + B + lane;
            rMeta = 0; rCl0 = rCl1 = rCl2 = rCl3 = ZF_NOCP2; rLitLo = rLitHi = 0;
            if (p < end) {
                const uint32_t* e = zf_entry(v, p);
                const uint4 a = *(const uint4*)e, b = *(const uint4*)(e + 4);
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                uint32_t cl[8];
                int n = 0;
                const bool link = (b.w & ZF_POOL_LINK) != 0;
                for (int c = 0; c < 8; c++) {
                    cl[c] = ZF_NOCP;
                    if (w[c] && !link) {
                        const int ds = d4g_dist2sym((int)(w[c] & 0xffff));
                        cl[c] = w[c] >> 16;
                        S.cpDb[lane][c] = (uint8_t)d4g_dsym_ebits(ds);
                        S.cpDc[lane][c] = fixedModel ? 0.0 : S.dsym[ds];
                        n++;
                    }
                }
                if (link) n = 255;
                rCl0 = cl[0] | (cl[1] << 16); rCl1 = cl[2] | (cl[3] << 16); rCl2 = cl[4] | (cl[5] << 16); rCl3 = cl[6] | (cl[7] << 16);
                const unsigned long long lu = zf_d2u(zf_model_lit(S, fixedModel, v.data[p]));
                rLitLo = (uint32_t)lu; rLitHi = (uint32_t)(lu >> 32);
                bool fl = false;
                if (p > start + ZF_MAXM + 1 && p + ZF_MAXM * 2 + 1 < end)
                    fl = zf_samecap(v.same[p], end, p) > ZF_MAXM * 2 && zf_samecap(v.same[p - ZF_MAXM], end, p - ZF_MAXM) > ZF_MAXM;
                rMeta = (zf_best(v, p) >> 16) | ((uint32_t)n << 16) | (fl ? 1u << 24 : 0u) | (n == 255 ? 1u << 25 : 0u);
            }
            loaded = B;
            LZ_WAVE_SYNC();
        }
        // the usual positions of this 64-block in a tight loop over registers 0..5: no long-run flag, at most eight change
        // points.  Anything else leaves the loop for the general code below.
        if (!afterShortcut) {
            int o = j - B;
            const int oEnd = size - B < 64 ? size - B : 64;
            float cr[6];
            uint32_t lr[6];
#pragma unroll
            for (int q = 0; q < 6; q++) { cr[q] = C[q]; lr[q] = Ln[q]; }
            const int oTight = oEnd < 63 ? oEnd : 63;
            for (; o < oTight; o++) {
                const uint32_t meta = zf_rl(rMeta, o);
                if (__builtin_expect((meta >> 24) != 0u, 0)) break;
                const int leng = (int)(meta & 0xffff);
                const int kend = leng >= 3 ? (leng < size - B - o ? leng : size - B - o) : 0;
                const double cj = (double)zf_u2f(zf_rl(zf_f2u(cr[0]), o));
                const double ncLit = zf_u2d((unsigned long long)zf_rl(rLitLo, o) | ((unsigned long long)zf_rl(rLitHi, o) << 32)) + cj;
                // the literal: index o + 1 sits in register 0 (position 63, whose literal lands in register 1, is left to the
                // general code below).  The matches' targets are other lanes (k >= 3): register 0's costs as doubles serve both.
                const double a0 = (double)cr[0];
                if (lane == o + 1 && ncLit < a0) { cr[0] = (float)ncLit; lr[0] = 1; }
                if (__builtin_expect(kend >= 3, 1)) {
                    const double mca = mincost + cj;
                    const uint32_t c0 = zf_rl(rCl0, o), c1 = zf_rl(rCl1, o), c2 = zf_rl(rCl2, o), c3 = zf_rl(rCl3, o);
                    auto relax = [&](int q, float& crq, uint32_t& lrq) D4G_LAMBDA_INLINE {
                        const int k = 64 * q + lane - o;
                        const double cq = q == 0 ? a0 : (double)crq;
                        const bool isM = (unsigned)(k - 3) <= (unsigned)(kend - 3) && !(cq <= mca);   // 3 <= k <= kend (kend >= 3 here)
                        if (__ballot(isM) != 0ull) {   // (worth its wait: dropping the test for register 0 costs 14 %)
                            const int kk = isM ? k : 3;
                            const int ci = zf_count_below(c0, c1, c2, c3, kk);
                            const int ib = S.lbTab[kk] + S.cpDb[o][ci];
                            const double nc = fixedModel ? (double)ib + cj : (((double)ib + S.llTab[kk]) + S.cpDc[o][ci]) + cj;
                            if (isM && nc < cq) { crq = (float)nc; lrq = (uint32_t)k; }
                        }
                    };
                    relax(0, cr[0], lr[0]);
                    if (__builtin_expect(o + kend >= 64, 0)) {   // (one test for the usual case — every target in register 0 — instead of one per register)
                        relax(1, cr[1], lr[1]);
                        if (o + kend >= 128) {
#pragma unroll
                            for (int q = 2; q < 6; q++)
                                if (o + kend >= 64 * q) relax(q, cr[q], lr[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 6; q++) { C[q] = cr[q]; Ln[q] = lr[q]; }
            j = B + o;
            if (o >= oEnd) continue;
        }
        const int o = j - B;
        const uint32_t meta = zf_rl(rMeta, o);
        if (((meta >> 24) & 1u) && !afterShortcut) {
            // inside a long run of one byte: 258 positions take a 258-byte match at distance 1 without searching.
            // index t = s + 258 takes cost[s] + c for the 258 sources s = j .. j + 257: (q, L) <- (q - 4, L - 2) or (q - 5, L + 62)
            const double symbolcost = zf_model(S, fixedModel, ZF_MAXM, 1);
#pragma unroll
            for (int q = ZF_NQ - 1; q >= 4; q--) {
                const int srel = 64 * q + lane - ZF_MAXM;       // the source's index relative to B
                const float a = __shfl(C[q - 4], (lane + 62) & 63);
                const float b = q >= 5 ? __shfl(C[q - 5], (lane + 62) & 63) : 0.f;
                const float src = lane >= 2 ? a : b;
                if (srel >= o && srel < o + ZF_MAXM) { C[q] = (float)((double)src + symbolcost); Ln[q] = ZF_MAXM; }
            }
            j += ZF_MAXM;
            afterShortcut = true;
            continue;
        }
        afterShortcut = false;
        const int leng = (int)(meta & 0xffff), ncp = (int)((meta >> 16) & 0xff);
        const double cj = (double)zf_u2f(zf_rl(zf_f2u(C[0]), o));
        const double ncLit = zf_u2d((unsigned long long)zf_rl(rLitLo, o) | ((unsigned long long)zf_rl(rLitHi, o) << 32)) + cj;
        const int kend = leng >= 3 ? (leng < size - j ? leng : size - j) : 0;
        const double mca = mincost + cj;
        {   // the literal: index o + 1 sits in register 0, or in register 1 when o == 63
            const double c0d = (double)C[0], c1d = (double)C[1];
            if (lane == o + 1 && ncLit < c0d) { C[0] = (float)ncLit; Ln[0] = 1; }
            if (lane == o - 63 && ncLit < c1d) { C[1] = (float)ncLit; Ln[1] = 1; }
        }
        if (kend >= 3) {
            if (__builtin_expect(ncp != 255 && o + kend < 128, 1)) {
                // the usual case: at most eight change points, targets within registers 0 and 1
                const uint32_t c0 = zf_rl(rCl0, o), c1 = zf_rl(rCl1, o), c2 = zf_rl(rCl2, o), c3 = zf_rl(rCl3, o);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (q == 0 || o + kend >= 64) {
                        const int k = 64 * q + lane - o;
                        const double cq = (double)C[q];
                        const bool isM = k >= 3 && k <= kend && !(cq <= mca);      // (targets already at the model's minimum are skipped, as published)
                        if (__ballot(isM) != 0ull) {
                            const int kk = isM ? k : 3;
                            const int ci = zf_count_below(c0, c1, c2, c3, kk);
                            const int ib = S.lbTab[kk] + S.cpDb[o][ci];
                            const double nc = fixedModel ? (double)ib + cj : (((double)ib + S.llTab[kk]) + S.cpDc[o][ci]) + cj;
                            if (isM && nc < cq) { C[q] = (float)nc; Ln[q] = (uint32_t)k; }
                        }
                    }
                }
            } else {
                const int qmax = (o + kend) >> 6;
                uint32_t c0 = ZF_NOCP2, c1 = ZF_NOCP2, c2 = ZF_NOCP2, c3 = ZF_NOCP2;
                if (ncp != 255) { c0 = zf_rl(rCl0, o); c1 = zf_rl(rCl1, o); c2 = zf_rl(rCl2, o); c3 = zf_rl(rCl3, o); }
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    if (q <= qmax) {
                        const int k = 64 * q + lane - o;
                        const double cq = (double)C[q];
                        const bool isM = k >= 3 && k <= kend && !(cq <= mca);
                        if (__ballot(isM) == 0ull) continue;
                        const int kk = isM ? k : 3;
                        double nc;
                        if (ncp != 255) {
                            const int ci = zf_count_below(c0, c1, c2, c3, kk);
                            const int ib = S.lbTab[kk] + S.cpDb[o][ci];
                            nc = fixedModel ? (double)ib + cj : (((double)ib + S.llTab[kk]) + S.cpDc[o][ci]) + cj;
                        } else {
                            nc = isM ? zf_model(S, fixedModel, kk, zf_sublen(v, zf_entry(v, start + j), kk)) + cj : 0.0;
                        }
                        if (isM && nc < cq) { C[q] = (float)nc; Ln[q] = (uint32_t)k; }
                    }
                }
            }
        }
        j++;
    }
    // the lengths still in registers
    while (B <= size) {
        const int x = B + lane;
        if (x > 0 && x <= size) job.lengthArray[x] = (uint16_t)Ln[0];
#pragma unroll
        for (int q = 0; q + 1 < ZF_NQ; q++) { C[q] = C[q + 1]; Ln[q] = Ln[q + 1]; }
        B += 64;
    }
    LZ_WAVE_SYNC();
}

// TraceBackwards + FollowPath: walk lengthArray back from the end (lane 0, through an LDS window), then look every
// chosen length's distance up in the match table (all lanes) and count the symbols into S.E.llc / S.E.dc.
D4G_DEV uint32_t zf_trace_follow(ZfSqLds& S, const ZfSqJob& job, int buf) {
    const ZfView& v = job.v;
    const int lane = threadIdx.x & 63;
    const long long size = v.end - v.start;
    if (size == 0) return 0;
    __threadfence();
    long long idx = size;
    uint32_t npath = 0;
    const long long cap = size;
    while (idx > 0) {
        const long long base = idx > 4095 ? idx - 4095 : 0;
        LZ_WAVE_SYNC();
        for (long long k = lane; k <= idx - base; k += 64) S.E.u.chunk[k] = job.lengthArray[base + k];
        LZ_WAVE_SYNC();
        if (lane == 0) {
            while (idx > 0 && idx >= base) {
                const int l = S.E.u.chunk[idx - base];
                job.path[cap - 1 - npath] = ((uint32_t)(idx - l) << 9) | (uint32_t)l;
                npath++;
                idx -= l;
            }
        }
        idx = __shfl(idx, 0);
        npath = __shfl(npath, 0);
    }
    __threadfence();
    LZ_WAVE_SYNC();
    for (int i = lane; i < ZF_NUM_LL; i += 64) S.E.llc[i] = 0;
    if (lane < ZF_NUM_D) S.E.dc[lane] = 0;
    LZ_WAVE_SYNC();
    uint16_t* oLit = buf ? job.lit[1] : job.lit[0];       // (no dynamic indexing: it would pin the whole job struct in scratch memory)
    uint16_t* oDist = buf ? job.dist[1] : job.dist[0];
    uint32_t* oPos = buf ? job.pos[1] : job.pos[0];
    for (uint32_t f = lane; f < npath; f += 64) {
        const uint32_t w = job.path[cap - npath + f];
        const long long p = v.start + (w >> 9);
        const int l = (int)(w & 511);
        int litlen, dist = 0;
        if (l >= 3) { litlen = l; dist = zf_sublen(v, zf_entry(v, p), l); }
        else litlen = v.data[p];
        oLit[f] = (uint16_t)litlen; oDist[f] = (uint16_t)dist; oPos[f] = (uint32_t)p;
        if (dist == 0) atomicAdd(&S.E.llc[litlen], 1u);
        else { atomicAdd(&S.E.llc[d4g_len2sym(litlen, 0)], 1u); atomicAdd(&S.E.dc[d4g_dist2sym(dist)], 1u); }
    }
    LZ_WAVE_SYNC();
    return npath;
}

__global__ void __launch_bounds__(64) k_zf_squeeze(const ZfSqJob* __restrict__ jobs, ZfSqOut* __restrict__ outs) {
    __shared__ ZfSqLds S;
    const ZfSqJob& job = jobs[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int NH = ZF_NUM_LL + ZF_NUM_D;
    if (job.fixedModel) {
        zf_best_lengths(S, job, true);
        const uint32_t n = zf_trace_follow(S, job, 0);
        if (lane == 0) outs[blockIdx.x] = {0, n, 0, {0, 0, 0, 0, 0, 0}};
        return;
    }
#ifdef D4G_HOSTSIM
#define ZF_CLOCK() 0LL
#else
#define ZF_CLOCK() (long long)wall_clock64()
#endif
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    long long tc = ZF_CLOCK(), tn;
    // first statistics: the greedy parse
    for (int k = lane; k < NH; k += 64) S.f[k] = 0;
    LZ_WAVE_SYNC();
    zf_greedy_walk<false>(job.v, (uint32_t*)S.E.u.chunk, nullptr, nullptr, nullptr, S.f);
    LZ_WAVE_SYNC();
    if (lane == 0) S.f[256] = 1;
    zf_calc_stats(S);
    tn = ZF_CLOCK(); cyc[0] += tn - tc; tc = tn;
    for (int k = lane; k < NH; k += 64) S.fbest[k] = 0;
    ZfRan ran = {1, 2};
    int cur = 0, bestBuf = 0, lastrandomstep = -1;
    uint32_t bestSize = 0;
    long long bestcost = 0x7fffffffffffffffLL, lastcost = 0;
    for (int it = 0; it < job.iterations; it++) {
        zf_best_lengths(S, job, false);
        tn = ZF_CLOCK(); cyc[1] += tn - tc; tc = tn;
        const uint32_t n = zf_trace_follow(S, job, cur);
        tn = ZF_CLOCK(); cyc[2] += tn - tc; tc = tn;
        if (lane == 0) S.E.llc[256] = 1;
        LZ_WAVE_SYNC();
        const long long cost = 3 + zf_dynamic_lengths(S.E, nullptr);
        LZ_WAVE_SYNC();
        tn = ZF_CLOCK(); cyc[3] += tn - tc; tc = tn;
        if (cost < bestcost) {
            for (int k = lane; k < NH; k += 64) S.fbest[k] = S.f[k];
            bestBuf = cur; bestSize = n; bestcost = cost;
            cur ^= 1;
        }
        for (int k = lane; k < NH; k += 64) {
            S.flast[k] = S.f[k];
            S.f[k] = k < ZF_NUM_LL ? S.E.llc[k] : S.E.dc[k - ZF_NUM_LL];     // GetStatistics of the run just made (llc[256] is already 1)
        }
        zf_calc_stats(S);
        if (lastrandomstep != -1) {
            for (int k = lane; k < NH; k += 64) S.f[k] = (uint32_t)((double)S.f[k] * 1.0 + (double)S.flast[k] * 0.5);
            LZ_WAVE_SYNC();
            if (lane == 0) S.f[256] = 1;
            zf_calc_stats(S);
        }
        if (it > 5 && cost == lastcost) {
            for (int k = lane; k < NH; k += 64) S.f[k] = S.fbest[k];
            LZ_WAVE_SYNC();
            if (lane == 0) {
                for (int i = 0; i < ZF_NUM_LL; i++) if ((zf_ran(ran) >> 4) % 3 == 0) S.f[i] = S.f[zf_ran(ran) % ZF_NUM_LL];
                for (int i = 0; i < ZF_NUM_D; i++) if ((zf_ran(ran) >> 4) % 3 == 0) S.f[ZF_NUM_LL + i] = S.f[ZF_NUM_LL + zf_ran(ran) % ZF_NUM_D];
                S.f[256] = 1;
            }
            ran.w = __shfl(ran.w, 0);
            ran.z = __shfl(ran.z, 0);
            zf_calc_stats(S);
            lastrandomstep = it;
        }
        lastcost = cost;
        tn = ZF_CLOCK(); cyc[4] += tn - tc; tc = tn;
    }
    if (lane == 0) outs[blockIdx.x] = {bestBuf, bestSize, bestcost, {cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], 0}};
}

// ---------------------------------------------------------------------------------------------------------------
// emission (deflate.c AddLZ77Block / AddDynamicTree / AddNonCompressedBlock): bits are OR-ed into a zeroed word array
// ---------------------------------------------------------------------------------------------------------------
struct ZfEmitJob {
    ZfStore s;
    uint32_t a, b;           // store range (types 1, 2)
    int32_t btype, final;
    long long bitPos;        // absolute bit position of the block's first header bit in `out`
    const uint8_t* src;      // type 0: the bytes of this stored chunk
    uint32_t srcLen;
    uint32_t pad;
    uint32_t* out;
};
D4G_DEV void zf_put_bits(uint32_t* out, long long pos, unsigned long long v, int n) {   // n <= 57 bits, LSB first
    if (n == 0) return;
    const long long w = pos >> 5;
    const int sh = (int)(pos & 31);
    const unsigned long long lo = v << sh;
    if ((uint32_t)lo) atomicOr(&out[w], (uint32_t)lo);
    if ((uint32_t)(lo >> 32)) atomicOr(&out[w + 1], (uint32_t)(lo >> 32));
    if (sh && sh + n > 64) { const uint32_t hi = (uint32_t)(v >> (64 - sh)); if (hi) atomicOr(&out[w + 2], hi); }
}
D4G_DEV uint32_t zf_rev(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i); return r; }
struct ZfEmitLds {
    ZfEvalLds E;
    uint16_t llcode[ZF_NUM_LL], dcode[ZF_NUM_D];   // bit-reversed canonical codes
    uint8_t ll[ZF_NUM_LL], d[ZF_NUM_D];
    uint32_t waveTot[4];
    long long base;
};
// canonical codes (ZopfliLengthsToSymbols), reversed for LSB-first output; one lane
D4G_DEV void zf_codes(const uint8_t* len, int n, uint16_t* code) {
    int blc[16] = {0}, next[16];
    for (int i = 0; i < n; i++) blc[len[i]]++;
    blc[0] = 0;
    int c = 0;
    next[0] = 0;
    for (int b = 1; b < 16; b++) { c = (c + blc[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; i++) code[i] = len[i] ? (uint16_t)zf_rev((uint32_t)next[len[i]]++, len[i]) : 0;
}
__global__ void __launch_bounds__(256) k_zf_emit(const ZfEmitJob* jobs) {
    __shared__ ZfEmitLds L;
    const ZfEmitJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* out = job.out;
    if (job.btype == 0) {
        if (tid == 0) {
            zf_put_bits(out, job.bitPos, (unsigned)job.final, 3);        // BFINAL, BTYPE 00
            const long long byte0 = (job.bitPos + 3 + 7) >> 3;
            const uint32_t nlen = (~job.srcLen) & 0xffffu;
            zf_put_bits(out, byte0 * 8, (unsigned long long)job.srcLen | ((unsigned long long)nlen << 16), 32);
        }
        const long long d0 = ((job.bitPos + 3 + 7) >> 3) + 4;
        for (uint32_t k = tid; k < job.srcLen; k += blockDim.x) zf_put_bits(out, (d0 + k) * 8, job.src[k], 8);
        return;
    }
    if (wave == 0) {
        long long hdrBits = 3;
        int combo = 0;
        if (job.btype == 2) {
            zf_count_range(L.E, job.s, job.a, job.b);
            if (lane == 0) L.E.llc[256] = 1;
            LZ_WAVE_SYNC();
            zf_dynamic_lengths(L.E, &combo);
            for (int i = lane; i < ZF_NUM_LL; i += 64) L.ll[i] = L.E.ll[i];
            if (lane < ZF_NUM_D) L.d[lane] = L.E.d[lane];
        } else {
            for (int i = lane; i < ZF_NUM_LL; i += 64) L.ll[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
            if (lane < ZF_NUM_D) L.d[lane] = 5;
        }
        LZ_WAVE_SYNC();
        if (lane == 0) zf_codes(L.ll, ZF_NUM_LL, L.llcode);
        if (lane == 1) zf_codes(L.d, ZF_NUM_D, L.dcode);
        if (lane == 0) {
            long long bp = job.bitPos;
            zf_put_bits(out, bp, (unsigned)job.final | ((unsigned)job.btype << 1), 3);
            bp += 3;
            if (job.btype == 2) {   // EncodeTree with the winning combination
                const bool use16 = combo & 1, use17 = combo & 2, use18 = combo & 4;
                const uint8_t* ll = L.ll;
                const uint8_t* d = L.d;
                ZfClInst& c = L.E.u.cl[0];
                int hlit = 29, hdist = 29;
                zf_tree_size_one(ll, d, combo, c);     // leaves the code-length code's lengths in c.len
                uint16_t clcode[19];
                zf_codes(c.len, 19, clcode);
                while (hlit > 0 && ll[257 + hlit - 1] == 0) hlit--;
                while (hdist > 0 && d[1 + hdist - 1] == 0) hdist--;
                const int hlit2 = hlit + 257, total = hlit2 + hdist + 1;
                int hclen = 15;
                while (hclen > 0 && c.clc[D4G_CL_ORDER[hclen + 4 - 1]] == 0) hclen--;
                zf_put_bits(out, bp, (unsigned)hlit | ((unsigned)hdist << 5) | ((unsigned)hclen << 10), 14);
                bp += 14;
                for (int i = 0; i < hclen + 4; i++) { zf_put_bits(out, bp, c.len[D4G_CL_ORDER[i]], 3); bp += 3; }
                auto put = [&](int sym, int extra, int ebits) {
                    zf_put_bits(out, bp, clcode[sym], c.len[sym]); bp += c.len[sym];
                    if (ebits) { zf_put_bits(out, bp, (unsigned)extra, ebits); bp += ebits; }
                };
                for (int i = 0; i < total; i++) {
                    const int symbol = i < hlit2 ? ll[i] : d[i - hlit2];
                    int count = 1;
                    if (use16 || (symbol == 0 && (use17 || use18)))
                        for (int j = i + 1; j < total && symbol == (j < hlit2 ? ll[j] : d[j - hlit2]); j++) count++;
                    i += count - 1;
                    if (symbol == 0 && count >= 3) {
                        if (use18) while (count >= 11) { const int c2 = count > 138 ? 138 : count; put(18, c2 - 11, 7); count -= c2; }
                        if (use17) while (count >= 3) { const int c2 = count > 10 ? 10 : count; put(17, c2 - 3, 3); count -= c2; }
                    }
                    if (use16 && count >= 4) {
                        count--;
                        put(symbol, 0, 0);
                        while (count >= 3) { const int c2 = count > 6 ? 6 : count; put(16, c2 - 3, 2); count -= c2; }
                    }
                    while (count > 0) { put(symbol, 0, 0); count--; }
                }
            }
            hdrBits = bp - job.bitPos;
            L.base = job.bitPos + hdrBits;
        }
    }
    __syncthreads();
    // the symbols: a tile of 256 per step, bit offsets from a workgroup scan
    const uint32_t n = job.b - job.a;
    for (uint32_t t0 = 0; t0 < n; t0 += 256) {
        const uint32_t i = t0 + tid;
        unsigned long long bits = 0;
        int nb = 0;
        if (i < n) {
            const int dd = job.s.dist[job.a + i], l = job.s.lit[job.a + i];
            if (dd == 0) { bits = L.llcode[l]; nb = L.ll[l]; }
            else {
                const int ls = d4g_len2sym(l, 0), ds = d4g_dist2sym(dd);
                const int le = d4g_lsym_ebits(ls), de = d4g_dsym_ebits(ds);
                bits = L.llcode[ls]; nb = L.ll[ls];
                bits |= (unsigned long long)(l - d4g_lsym_base(ls)) << nb; nb += le;
                bits |= (unsigned long long)L.dcode[ds] << nb; nb += L.d[ds];
                bits |= (unsigned long long)(dd - d4g_dsym_base(ds)) << nb; nb += de;
            }
        }
        int inc = nb;
        for (int dlt = 1; dlt < 64; dlt <<= 1) { const int o = __shfl_up(inc, dlt); if (lane >= dlt) inc += o; }
        if (lane == 63) L.waveTot[wave] = (uint32_t)inc;
        __syncthreads();
        long long off = L.base;
        for (int w = 0; w < wave; w++) off += L.waveTot[w];
        if (nb) zf_put_bits(out, off + inc - nb, bits, nb);
        __syncthreads();
        if (tid == 0) L.base += (long long)L.waveTot[0] + L.waveTot[1] + L.waveTot[2] + L.waveTot[3];
        __syncthreads();
    }
    if (tid == 0) zf_put_bits(out, L.base, L.llcode[256], L.ll[256]);   // end of block
}

// length of the tail a block end needs searched again: max(258, the byte run that crosses `end`); one wave per query
struct ZfTailQuery { int32_t input; int32_t pad; long long end; };
__global__ void __launch_bounds__(64) k_zf_tail_len(const ZfInput* inputs, const ZfTailQuery* qs, uint32_t* out) {
    const ZfTailQuery q = qs[blockIdx.x];
    const ZfInput in = inputs[q.input];
    const int lane = threadIdx.x & 63;
    uint32_t r = 0;
    for (long long base = q.end - 1; base >= 0 && r < 65536; base -= 64) {
        const long long x = base - lane;
        const bool crosses = x >= 0 && x + (long long)in.same[x] >= q.end;
        const unsigned long long m = __ballot(crosses);
        if (m == ~0ull) { r += 64; continue; }
        r += (uint32_t)(__ffsll((long long)~m) - 1);
        break;
    }
    if (lane == 0) out[blockIdx.x] = r < ZF_MAXM ? ZF_MAXM : r;
}
