// d4g_lz77.h — LZ77 match search + lazy parse + block emit, bit-compatible with zlib level 9.
//
// The reference owns no match finder: its recompress modes call java.util.zip.Deflater(BEST_COMPRESSION, nowrap) and
// jzlib at level 9 with strategies DEFAULT / FILTERED / HUFFMAN_ONLY (C/JavaCompressor.java:36-49,
// C/JZLibCompressor.java:29-41; C/ = deft4j-compress/src/main/java/com/github/NeRdTheNed/deft4j/util/compression/).
// Those encoders are a sequential hash-chain search with lazy evaluation; what they compute is nevertheless a pure
// function of the data, and this file computes that function the MI355X way:
//
//   k_lz_sort    every position's 3-byte hash (15 bits, zlib's UPDATE_HASH with shift 5) is bucketed per 32 Ki-position
//                *sort block*: S16 = the block's positions ordered by (hash, position), rank16 = where each position
//                sits in that order, bstart = where each hash's bucket begins.  A position's hash chain — every earlier
//                position with the same hash, nearest first — is then two CONTIGUOUS runs (its own block's bucket
//                below its rank, then the previous block's bucket from the top), so 64 lanes can test 64 chain
//                candidates at once instead of chasing prev[] pointers.
//   k_lz_parse   one wave per 2 Ki-position chunk runs deflate_slow's state machine (match_available, prev_length,
//                prev_match — wave-uniform, on the scalar unit).  At each position the machine actually visits, the wave
//                searches the chain cooperatively (lz_search): lane j takes candidate j, the 32 KiB window + chunk are
//                staged in LDS, a wave max-reduce on (length, nearest-first) picks exactly the candidate zlib's
//                sequential scan would have kept (first one to reach the maximum; stop at nice_length).  Chunks start
//                from a guessed state 512 bytes early; k_lz_check compares every chunk's entry state with its
//                predecessor's exit state and the few that differ are re-run from the exact state until all agree —
//                by induction from chunk 0 the parse is then the sequential one.
//   k_lz_fill    chunk token buffers -> the optimiser's token / back-reference record arrays (d4g_types.h), with an
//                end-of-block token after every 16383 symbols (lit_bufsize - 1; jzlib's early flush: k_lz_split).
//   k_lz_blocks  per block: symbol histogram, zlib's build_tree / gen_bitlen / scan_tree (the heap's tie-break on
//                depth included), stored / fixed / dynamic decision of _tr_flush_block, and the block's D4GState —
//                the same state the parser would produce for the emitted block, so the candidate search and the
//                bit packer (k_write) take over without a re-parse.
// All integer / byte work; bound by LDS latency and VALU issue, not HBM.
#pragma once
#include <cmath>

#include "d4g_device.h"

#define LZ_SORT_BLOCK 32768
#define LZ_CHUNK 2048          // positions per parse chunk (one wave)
#define LZ_WARM 512            // a speculative chunk starts this many positions early
#define LZ_MAX_DIST 32506      // w_size - MIN_LOOKAHEAD
#define LZ_SYMS_PER_BLOCK 16383
#define LZ_WIN_SLACK 304       // bytes staged past the last chunk (MAX_MATCH + the 8-byte compare's over-read)

#define LZ_DEFAULT 0
#define LZ_FILTERED 1
#define LZ_HUFFMAN_ONLY 2
#define LZ_FLAVOR_ZLIB 0
#define LZ_FLAVOR_JZLIB 1

#ifdef D4G_HOSTSIM
#define LZ_WAVE_SYNC() ((void)__ballot(1))
#else
#define LZ_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

struct LzStream {           // one uncompressed input
    const uint8_t* data;    // in U (16-byte aligned, >= 320 zero bytes after the end)
    long long len;
    long long posBase;      // index of position 0 in S16 / rank16 (a multiple of LZ_SORT_BLOCK)
    int32_t sortBlock0;     // first sort block (index into bstart / LZ_SORT_BLOCK)
    int32_t nChunks;
};
struct LzSortJob { int32_t stream, blk; };
struct LzParseJob { int32_t stream, firstChunk, metaBase, strategy; };   // meta index of chunk c = metaBase + c
struct LzChunkMeta {
    unsigned long long entry, exit;   // packed parse states (lz_pack_state)
    uint32_t ntok, nmatch;
    uint32_t firstPos;                // position of the first byte the chunk's first token covers
    uint32_t pad;                     // the chunk's last token
    uint32_t dcost;                   // sum over the chunk's matches of 5 + extra bits of the distance code (jzlib's early-flush bound)
    uint32_t pad2;
};

// parse state at a loop top of deflate_slow: position, match_available, match_length carried from the previous
// position (2 = none) and the distance of that pending match
D4G_DEV unsigned long long lz_pack_state(long long p, int ma, int ml, int md) {
    return (unsigned long long)p | ((unsigned long long)ml << 32) | ((unsigned long long)ma << 41) | ((unsigned long long)(ml >= 3 ? md : 0) << 42);
}

D4G_DEV unsigned lz_hash3(unsigned b0, unsigned b1, unsigned b2) { return ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7fffu; }
D4G_DEV int lz_block_inserted(long long len, int blk) {   // positions of sort block blk that zlib inserts (p + 3 <= len)
    long long n = len - 2 - (long long)blk * LZ_SORT_BLOCK;
    return n < 0 ? 0 : n > LZ_SORT_BLOCK ? LZ_SORT_BLOCK : (int)n;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_sort: stable counting sort of one sort block's positions by hash.
// ---------------------------------------------------------------------------------------------------------------------
#define LZ_SORT_THREADS 512
__global__ void __launch_bounds__(LZ_SORT_THREADS) k_lz_sort(const LzStream* streams, const LzSortJob* jobs, uint16_t* S16, uint16_t* rank16,
                                                             uint16_t* bstart) {
    __shared__ uint16_t tbl[32768];          // counts, then bucket cursors
    __shared__ unsigned part[LZ_SORT_THREADS];
    const LzSortJob job = jobs[blockIdx.x];
    const LzStream st = streams[job.stream];
    const long long p0 = (long long)job.blk * LZ_SORT_BLOCK;
    const int nIns = lz_block_inserted(st.len, job.blk);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t* d = st.data + p0;
    unsigned* tw = (unsigned*)tbl;
    for (int i = tid; i < 16384; i += LZ_SORT_THREADS) tw[i] = 0;
    __syncthreads();
    for (int i = tid; i < nIns; i += LZ_SORT_THREADS) {
        unsigned h = lz_hash3(d[i], d[i + 1], d[i + 2]);
        atomicAdd(tw + (h >> 1), 1u << ((h & 1) * 16));   // two 16-bit counters per word; a count is at most 32768
    }
    __syncthreads();
    // exclusive scan of the 32768 counters: 64 per thread
    unsigned sum = 0;
    for (int k = 0; k < 32; k++) { unsigned w = tw[tid * 32 + k]; sum += (w & 0xffffu) + (w >> 16); }
    part[tid] = sum;
    __syncthreads();
    if (tid < 64) {   // scan of 512 partial sums by one wave: 8 per lane
        unsigned loc[8], s = 0;
        for (int k = 0; k < 8; k++) { loc[k] = s; s += part[tid * 8 + k]; }
        unsigned inc = s;
        for (int dd = 1; dd < 64; dd <<= 1) { unsigned o = __shfl_up(inc, dd); if (lane >= dd) inc += o; }
        unsigned excl = inc - s;
        for (int k = 0; k < 8; k++) part[tid * 8 + k] = excl + loc[k];
    }
    __syncthreads();
    {
        unsigned run = part[tid];
        uint16_t* bs = bstart + ((long long)st.sortBlock0 + job.blk) * LZ_SORT_BLOCK;
        for (int k = 0; k < 32; k++) {
            unsigned w = tw[tid * 32 + k];
            unsigned a = run, b = run + (w & 0xffffu);
            run = b + (w >> 16);
            tw[tid * 32 + k] = (a & 0xffffu) | (b << 16);
            bs[tid * 64 + 2 * k] = (uint16_t)a;
            bs[tid * 64 + 2 * k + 1] = (uint16_t)b;
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    // Placement in position order by one wave, 64 positions per step.  Lanes whose hash is unique within the step take
    // their bucket's cursor directly; lanes that share a hash are found by a write / read-back of lane tags in the
    // cursor table itself and ranked by ballots (a handful of groups per step on text).
    volatile uint16_t* vt = tbl;
    uint16_t* Sout = S16 + st.posBase + p0;
    uint16_t* Rout = rank16 + st.posBase + p0;
    const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (int i0 = 0; i0 < nIns; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < nIns;
        unsigned h = 0;
        if (valid) h = lz_hash3(d[i], d[i + 1], d[i + 2]);
        unsigned cur = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        if (valid) vt[h] = (uint16_t)(0x8000u | (unsigned)lane);
        LZ_WAVE_SYNC();
        unsigned w = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        if (valid && (w & 63u) != (unsigned)lane) vt[h] = (uint16_t)(0xC000u | (unsigned)lane);
        LZ_WAVE_SYNC();
        unsigned w2 = valid ? vt[h] : 0u;
        LZ_WAVE_SYNC();
        bool dup = valid && (w2 & 0x4000u);
        unsigned rank = cur;
        if (valid && !dup) vt[h] = (uint16_t)(cur + 1);
        unsigned long long rem = __ballot(dup);
        while (rem) {
            int leader = __ffsll((long long)rem) - 1;
            unsigned hh = __shfl(h, leader);
            unsigned long long m = __ballot(dup && h == hh);
            if (dup && h == hh) {
                rank = cur + (unsigned)__popcll(m & below);
                if ((m >> lane) == 1ULL) vt[h] = (uint16_t)(cur + (unsigned)__popcll(m));   // the group's last lane
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

// ---------------------------------------------------------------------------------------------------------------------
// lz_search: longest_match for position p by the whole wave.  Returns the best length (>= bestInit) and, when it
// improved, the distance — the FIRST chain candidate that reaches the maximum, as zlib's scan keeps it.
// ---------------------------------------------------------------------------------------------------------------------
struct LzCtx {
    const LzStream* streams;
    const uint16_t* S16;
    const uint16_t* rank16;
    const uint16_t* bstart;
    LzChunkMeta* meta;
    uint32_t* chunkTok;       // LZ_CHUNK + 2 token words per chunk
    int32_t* errors;
};

D4G_DEV uint32_t lz_load4(const uint32_t* win, int off) {   // bytes [off, off+4) of the staged window (little endian)
    uint32_t lo = win[off >> 2], hi = win[(off >> 2) + 1];
    int sh = (off & 3) * 8;
    return (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh);
}

D4G_DEV unsigned long long lz_load8(const uint32_t* win, int off) {   // bytes [off, off+8)
    const int i = off >> 2, sh = (off & 3) * 8;
    const unsigned long long lo = ((unsigned long long)win[i + 1] << 32) | win[i];
    return sh ? (lo >> sh) | ((unsigned long long)win[i + 2] << (64 - sh)) : lo;
}

D4G_DEV void lz_search(const LzCtx& c, const LzStream& st, const uint32_t* win, long long w0, long long p, int bestInit, int maxChain,
                       int& bestLen, int& bestDist) {
    const int lane = threadIdx.x & 63;
    const int blk = (int)(p >> 15);
    const int r = c.rank16[st.posBase + p];
    const int po = (int)(p - w0);
    const unsigned w3 = lz_load4(win, po);
    const unsigned h = lz_hash3(w3 & 0xffu, (w3 >> 8) & 0xffu, (w3 >> 16) & 0xffu);
    const uint16_t* bs = c.bstart + ((long long)st.sortBlock0 + blk) * LZ_SORT_BLOCK;
    const int cntA = (r - (int)bs[h]) & 0xffff;
    int cntB = 0, endB = 0;
    if (blk > 0) {
        const uint16_t* bp = bs - LZ_SORT_BLOCK;
        int B2 = bp[h];
        int E2 = h == 32767u ? lz_block_inserted(st.len, blk - 1) : (int)bp[h + 1];
        cntB = (E2 - B2) & 0xffff;
        endB = B2 + cntB;
    }
    int total = cntA + cntB;
    if (total > maxChain) total = maxChain;
    const long long rem = st.len - p;
    const int maxcmp = rem < 258 ? (int)rem : 258;
    const uint16_t* SA = c.S16 + st.posBase + (long long)blk * LZ_SORT_BLOCK;
    int best = bestInit, bd = 0;
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        bool ok = i < total;
        long long cpos = 0;
        if (ok) {
            if (i < cntA) cpos = (long long)blk * LZ_SORT_BLOCK + SA[r - 1 - i];
            else cpos = (long long)(blk - 1) * LZ_SORT_BLOCK + (SA - LZ_SORT_BLOCK)[endB - 1 - (i - cntA)];
        }
        const int dist = (int)(p - cpos);
        // zlib: the chain head may be MAX_DIST away, later candidates must be nearer; stream position 0 is NIL
        ok = ok && dist <= (i == 0 ? LZ_MAX_DIST : LZ_MAX_DIST - 1) && cpos >= 1;
        int len = 0;
        if (ok) {
            const int co = po - dist;
            // can this candidate beat `best` at all?  (bytes best-3 .. best must agree; at best == 2 only bytes 0..2)
            const int off = best >= 3 ? best - 3 : 0;
            const unsigned msk = best >= 3 ? 0xffffffffu : 0x00ffffffu;
            if (((lz_load4(win, co + off) ^ lz_load4(win, po + off)) & msk) == 0) {
                int k = 0;
                while (k < maxcmp) {   // eight bytes per step
                    unsigned long long x = lz_load8(win, co + k) ^ lz_load8(win, po + k);
                    if (x) { k += (__ffsll((long long)x) - 1) >> 3; break; }
                    k += 8;
                }
                len = k < maxcmp ? k : maxcmp;
            }
        }
        int key = (ok && len > best) ? ((len << 8) | (63 - lane)) : 0;
        key = wave_max_i32(key);
        if (key) {
            best = key >> 8;
            bd = __shfl(dist, 63 - (key & 63));
        }
        if (best >= maxcmp) break;            // nice_length (258, or what is left of the input) reached
        if (__ballot(!ok && i < total)) break;   // the chain left the window
    }
    bestLen = best;
    bestDist = bd;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_parse: deflate_slow over chunks.  blockDim.x / 64 consecutive chunks of one stream share the staged window.
// exact = 0: every chunk but a stream's first starts LZ_WARM positions early from a clean state and records from its
//            first loop top inside the chunk;  exact = 1 (one wave per job): the chunk starts from its predecessor's
//            recorded exit state and the wave carries on into the following chunks while they disagree.
// ---------------------------------------------------------------------------------------------------------------------
#define LZ_PARSE_MAXWAVES 16   // 16 chunks share one staged window: 66 KB of LDS, two workgroups = 32 waves per CU
// `heads` (exact mode): per chunk, 1 = another workgroup re-runs this chunk in the same pass.  A re-run chunk whose new exit
// state differs from what its successor started from goes straight on into the successor (re-staging the window), and on,
// until a chunk's recorded entry equals the exit just produced — or the successor is somebody else's job.  Inputs whose
// speculative parses never fall into step (one long run: every chunk starts a 258-byte match at a different phase) are
// thus parsed by one wave in one pass instead of one pass per chunk.
__global__ void __launch_bounds__(64 * LZ_PARSE_MAXWAVES) k_lz_parse(LzCtx c, const LzParseJob* jobs, int exact, const uint8_t* heads) {
    __shared__ uint32_t win[(32768 + LZ_WARM + LZ_PARSE_MAXWAVES * LZ_CHUNK + LZ_WIN_SLACK + 64) / 4];
    const LzParseJob job = jobs[blockIdx.x];
    const LzStream st = c.streams[job.stream];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint8_t* wb = (const uint8_t*)win;
    int tileChunk = job.firstChunk;                 // first chunk of the staged tile
    unsigned long long carry = 0;                   // exact mode, from the second chunk of a chain on: the state to start from
    bool chained = false;
    for (;;) {
    const long long tile0 = (long long)tileChunk * LZ_CHUNK;
    long long w0 = tile0 - 32768 - LZ_WARM;
    if (w0 < 0) w0 = 0;
    w0 &= ~15LL;
    const int span = exact ? LZ_PARSE_MAXWAVES : nw;   // chunks the staged window serves (exact mode: the one wave walks through them)
    long long w1 = tile0 + (long long)span * LZ_CHUNK + LZ_WIN_SLACK;
    __syncthreads();
    {   // stage [w0, w1): the input is padded with zero bytes, so reading past st.len is harmless
        long long lim = ((st.len + 15) & ~15LL) + 320;
        if (w1 > lim) w1 = lim;
        const int nvec = (int)((w1 - w0 + 15) >> 4);
        const uint4* src = (const uint4*)(st.data + w0);
        uint4* dst = (uint4*)win;
        for (int i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    int chunk = tileChunk + wave;
    for (;;) {                                      // (exact mode: the chunks of the staged span, one after the other)
    if (chunk >= st.nChunks) return;                // (speculative pass: the tile's last waves may have no chunk; exact mode has one wave)
    const long long c0 = (long long)chunk * LZ_CHUNK;
    long long c1 = c0 + LZ_CHUNK;
    if (c1 > st.len) c1 = st.len;
    LzChunkMeta* M = c.meta + job.metaBase + chunk;
    uint32_t* tokOut = c.chunkTok + (long long)(job.metaBase + chunk) * (LZ_CHUNK + 2);

    long long p;
    int ma = 0, ml = 2, md = 0;
    bool rec;
    if (chunk == 0) { p = 0; rec = true; }
    else if (exact) {
        unsigned long long e = chained ? carry : (M - 1)->exit;
        p = (long long)(e & 0xffffffffULL);
        ml = (int)((e >> 32) & 0x1ff);
        ma = (int)((e >> 41) & 1);
        md = (int)((e >> 42) & 0xffff);
        rec = true;
    } else {
        p = c0 - LZ_WARM;
        if (p < 0) p = 0;
        rec = false;
    }
    unsigned long long entry = lz_pack_state(p, ma, ml, md);
    unsigned ntok = 0, nmatch = 0, tokbuf = 0, lastTok = 0, dcost = 0;
    long long firstPos = -1;
    if (rec && p >= c1 && p < st.len) {   // (cannot happen: a match is shorter than a chunk)
        if (lane == 0) atomicAdd(c.errors, 1);
        return;
    }
    auto emit = [&](unsigned t, long long startPos) D4G_LAMBDA_INLINE {
        if (!rec) return;
        if (firstPos < 0) firstPos = startPos;
        if (lane == (int)(ntok & 63u)) tokbuf = t;
        lastTok = t;
        ntok++;
        if ((ntok & 63u) == 0) tokOut[ntok - 64 + lane] = tokbuf;
    };
    while (p < c1) {
        if (!rec && p >= c0) { rec = true; entry = lz_pack_state(p, ma, ml, md); }
        const int prevLen = ml, prevDist = md;
        ml = 2;
        if (st.len - p >= 3 && prevLen < 258 && job.strategy != LZ_HUFFMAN_ONLY) {
            int bl, bd;
            lz_search(c, st, win, w0, p, prevLen, prevLen >= 32 ? 1024 : 4096, bl, bd);
            if (bl > prevLen) {
                ml = bl;
                md = bd;
                if (ml <= 5 && (job.strategy == LZ_FILTERED || (ml == 3 && md > 4096))) ml = 2;   // TOO_FAR
            }
        }
        if (prevLen >= 3 && ml <= prevLen) {
            emit((unsigned)prevLen | ((unsigned)prevDist << 9), p - 1);
            if (rec) { nmatch++; dcost += 5u + (unsigned)d4g_dsym_ebits(d4g_dist2sym(prevDist)); }
            p += prevLen - 1;
            ma = 0;
            ml = 2;
        } else if (ma) {
            emit((unsigned)wb[p - 1 - w0], p - 1);
            p++;
        } else {
            ma = 1;
            p++;
        }
    }
    if (!rec) { rec = true; entry = lz_pack_state(p, ma, ml, md); }   // (the warm-up ran past the whole chunk: only at a stream's end)
    if (p >= st.len) {   // end of input: the pending literal is flushed (deflate_slow's epilogue)
        if (ma) emit((unsigned)wb[st.len - 1 - w0], st.len - 1);
        ma = 0; ml = 2; md = 0;
        p = st.len;
    }
    if (ntok & 63u) { if (lane < (int)(ntok & 63u)) tokOut[(ntok & ~63u) + lane] = tokbuf; }
    const unsigned long long exitState = lz_pack_state(p, ma, ml, md);
    if (lane == 0) {
        M->entry = entry;
        M->exit = exitState;
        M->ntok = ntok;
        M->nmatch = nmatch;
        M->firstPos = (uint32_t)(firstPos < 0 ? p : firstPos);
        M->pad = lastTok;   // the chunk's last token (the host needs the stream's last one: match or literal)
        M->dcost = dcost;
        M->pad2 = 0;
    }
    if (!exact) return;
    // exact mode (one wave): does the successor have to follow?
    const int next = chunk + 1;
    if (next >= st.nChunks || heads[job.metaBase + next]) return;
    const unsigned long long nextEntry = (M + 1)->entry;      // (the successor is nobody's job in this pass: its meta is stable)
    if (nextEntry == exitState) return;
    carry = exitState;
    chained = true;
    chunk = next;
    if (next >= tileChunk + span) { tileChunk = next; break; }   // past the staged window: stage the next one
    }
    }
}

// chunks whose entry state is not their predecessor's exit state (one thread per chunk; list compacted by atomics —
// order does not matter, every listed chunk is re-run independently)
__global__ void k_lz_check(const LzChunkMeta* meta, const int32_t* chunkIndex, int nChunksTotal, int32_t* redo, unsigned* nRedo) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nChunksTotal) return;
    if (chunkIndex[i] == 0) return;
    if (meta[i].entry != meta[i - 1].exit) redo[atomicAdd(nRedo, 1u)] = i;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_split: block boundaries of the jzlib flavour.  jzlib 1.1.x keeps zlib's TRUNCATE_BLOCK heuristic in _tr_tally:
// when a block reaches 8192 symbols (level > 2) it is flushed early if matches < symbols / 2 and
// (8192 * 8 + sum over matches of (5 + distance extra bits)) / 8 < (strstart - block_start) / 2.  The decision chain is
// sequential over a stream's blocks; each decision is three prefix sums at a symbol index (chunk prefix from the host +
// a partial wave scan inside one chunk).  One wave per output stream.  PARITY UNPINNED (no jzlib in this environment).
// ---------------------------------------------------------------------------------------------------------------------
struct LzSplitJob {
    int32_t stream, metaBase;          // input stream; chunk metas of the parse (-1: HUFFMAN_ONLY)
    long long preBase;                 // first entry of this stream's chunk prefix tables (nChunks + 1 entries each)
    long long nSyms, nRefs, dcostTotal;
    long long outBase;                 // first slot of the stream in the boundary list
    int32_t maxBlocks, lastIsMatch;
};
struct LzSplitOut { long long symStart, symCount; int32_t isLast, pad; };

// (matches, distance cost, start position) of symbol t of the stream: sums over symbols [0, t)
D4G_DEV void lz_split_prefix(const LzCtx& c, const LzSplitJob& J, const LzStream& st, const long long* symPre, const long long* refPre,
                             const long long* dcPre, long long t, long long& m, long long& dc, long long& pos) {
    const int lane = threadIdx.x & 63;
    if (t >= J.nSyms) { m = J.nRefs; dc = J.dcostTotal; pos = st.len; return; }
    if (J.metaBase < 0) { m = 0; dc = 0; pos = t; return; }   // every byte a literal
    int lo = 0, hi = st.nChunks - 1;
    while (lo < hi) {   // last chunk whose first symbol is <= t (chunks without symbols are skipped: equal prefixes)
        int mid = (lo + hi + 1) >> 1;
        if (symPre[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const LzChunkMeta M = c.meta[J.metaBase + lo];
    const uint32_t* src = c.chunkTok + (long long)(J.metaBase + lo) * (LZ_CHUNK + 2);
    const int cnt = (int)(t - symPre[lo]);
    long long acc = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        int j = j0 + lane;
        long long v = 0;
        if (j < cnt) {
            unsigned w = src[j];
            int dist = (int)(w >> 9);
            v = dist ? ((long long)(w & 0x1ff) | ((long long)(5 + d4g_dsym_ebits(d4g_dist2sym(dist))) << 24) | (1LL << 44)) : 1LL;
        }
        acc += wave_sum_i64(v);
    }
    pos = (long long)M.firstPos + (acc & 0xffffff);
    dc = dcPre[lo] + ((acc >> 24) & 0xfffff);
    m = refPre[lo] + (acc >> 44);
}

__global__ void __launch_bounds__(64) k_lz_split(LzCtx c, const LzSplitJob* jobs, const long long* symPreAll, const long long* refPreAll,
                                                 const long long* dcPreAll, LzSplitOut* outList, int32_t* outCount) {
    const LzSplitJob J = jobs[blockIdx.x];
    const LzStream st = c.streams[J.stream];
    const long long *symPre = symPreAll + J.preBase, *refPre = refPreAll + J.preBase, *dcPre = dcPreAll + J.preBase;
    const int lane = threadIdx.x;
    LzSplitOut* out = outList + J.outBase;
    int nb = 0;
    long long cur = 0, mCur = 0, dCur = 0, pCur = 0;
    const long long N = J.nSyms;
    const bool epilogueLiteral = !J.lastIsMatch;   // deflate_slow's epilogue emits the last literal; its flush flag is ignored
    while (true) {
        long long take = N - cur < LZ_SYMS_PER_BLOCK ? N - cur : LZ_SYMS_PER_BLOCK;
        bool fired = take == LZ_SYMS_PER_BLOCK;   // the tally of the block's last symbol asked for a flush
        if (N - cur >= 8192) {
            const long long t = cur + 8192;
            long long m1, d1, p1, m0, d0, p0;
            lz_split_prefix(c, J, st, symPre, refPre, dcPre, t - 1, m0, d0, p0);   // p0 = start of the 8192nd symbol
            lz_split_prefix(c, J, st, symPre, refPre, dcPre, t, m1, d1, p1);
            const long long matches = m1 - mCur;
            const long long outLength = (8192LL * 8 + (d1 - dCur)) >> 3;
            const long long inLength = p0 + 1 - pCur;
            if (matches < 8192 / 2 && outLength < inLength / 2) { take = 8192; fired = true; }
        }
        const bool endsStream = cur + take == N;
        // a flush asked for by the input's very last symbol is ignored when that symbol is the epilogue's literal
        // (the block is then closed as the last one); asked for inside the loop it leaves an empty last block behind
        const bool last = endsStream && !(fired && take > 0 && !epilogueLiteral);
        if (nb < J.maxBlocks && lane == 0) { out[nb].symStart = cur; out[nb].symCount = take; out[nb].isLast = last; out[nb].pad = 0; }
        nb++;
        cur += take;
        if (last) break;
        if (endsStream) {
            if (nb < J.maxBlocks && lane == 0) { out[nb].symStart = N; out[nb].symCount = 0; out[nb].isLast = 1; out[nb].pad = 0; }
            nb++;
            break;
        }
        lz_split_prefix(c, J, st, symPre, refPre, dcPre, cur, mCur, dCur, pCur);
    }
    if (lane == 0) outCount[blockIdx.x] = nb;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_fill: one wave per (output stream, chunk): the chunk's tokens go to their final place in tok / refs / tokRef,
// and an end-of-block token follows the last symbol of every block.
// ---------------------------------------------------------------------------------------------------------------------
struct LzOutStream {           // one emitted deflate stream = (input, strategy, flavour)
    int32_t stream;            // LzStream index
    int32_t metaBase;          // chunk metas of its parse (-1: HUFFMAN_ONLY, every byte a literal)
    long long tokBase;         // first token of the stream in tok (symbols and end-of-block tokens)
    long long refBase;         // first back-reference record
    long long blkBase;         // first entry in the block tables
    int32_t nBlocks;
    int32_t pad;
};
struct LzFillJob { int32_t out, chunk; long long symBase, refBase; };   // symbols / records before this chunk in its stream
struct LzBlockDesc {           // per emitted block
    long long symStart, symCount;   // symbols (tokens without the end-of-block) of the stream that the block holds
    long long uStart, uLen;         // decoded byte range (written by k_lz_fill)
    long long refStart, refCount;   // back-reference records, relative to the stream (written by k_lz_fill / host)
    long long lastTop;              // loop-top position of the iteration that flushed the block (window base rule)
    int32_t out, isLast;
};

D4G_DEV int lz_find_block(const LzBlockDesc* bd, int nb, long long sym) {   // block holding symbol `sym`
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (bd[mid].symStart <= sym) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(64) k_lz_fill(LzCtx c, const LzOutStream* outs, const LzFillJob* jobs, LzBlockDesc* blocks, uint2* tok,
                                                uint4* refs, uint32_t* tokRef) {
    const LzFillJob job = jobs[blockIdx.x];
    const LzOutStream o = outs[job.out];
    const LzStream st = c.streams[o.stream];
    const int lane = threadIdx.x;
    LzBlockDesc* bd = blocks + o.blkBase;
    unsigned ntok;
    long long pos;
    const uint32_t* src = nullptr;
    if (job.chunk >= st.nChunks) {   // (an empty input still gets one job: its empty block's end-of-block token)
        ntok = 0;
        pos = 0;
    } else if (o.metaBase >= 0) {
        const LzChunkMeta m = c.meta[o.metaBase + job.chunk];
        ntok = m.ntok;
        pos = m.firstPos;
        src = c.chunkTok + (long long)(o.metaBase + job.chunk) * (LZ_CHUNK + 2);
    } else {
        long long c0 = (long long)job.chunk * LZ_CHUNK, c1 = c0 + LZ_CHUNK;
        if (c1 > st.len) c1 = st.len;
        ntok = (unsigned)(c1 - c0);
        pos = c0;
    }
    if (job.chunk == 0 && lane == 0 && bd[o.nBlocks - 1].symCount == 0)   // an empty last block holds just its end-of-block token
        tok[o.tokBase + bd[o.nBlocks - 1].symStart + (o.nBlocks - 1)] = make_uint2(256u, (uint32_t)st.len);
    if (ntok == 0) return;
    int blk = lz_find_block(bd, o.nBlocks, job.symBase);
    long long nrefBefore = job.refBase;
    for (unsigned t0 = 0; t0 < ntok; t0 += 64) {
        const unsigned t = t0 + lane;
        const bool ok = t < ntok;
        unsigned w = 0;
        if (ok) w = src ? src[t] : (unsigned)st.data[pos + lane];
        const int dist = (int)(w >> 9), len = dist ? (int)(w & 0x1ff) : 1;
        // exclusive scans over the 64 tokens: bytes covered, back-references
        int sl = ok ? len : 0, sr = (ok && dist) ? 1 : 0;
        int il = sl, ir = sr;
        for (int d = 1; d < 64; d <<= 1) {
            int a = __shfl_up(il, d), b = __shfl_up(ir, d);
            if (lane >= d) { il += a; ir += b; }
        }
        const long long myPos = pos + (il - sl);
        const long long myRef = nrefBefore + (ir - sr);
        const long long sym = job.symBase + t;
        if (ok) {
            int b = blk;
            while (b + 1 < o.nBlocks && bd[b + 1].symStart <= sym) b++;
            const LzBlockDesc& B = bd[b];
            const long long ti = o.tokBase + sym + b;          // b end-of-block tokens precede this symbol
            if (dist) {
                const int lsym = d4g_len2sym(len, 0), dsym = d4g_dist2sym(dist);
                tok[ti] = make_uint2((uint32_t)len | ((uint32_t)dist << 16), (uint32_t)myPos);
                refs[o.refBase + myRef] = make_uint4(d4g_ref_pack(len, lsym, dsym, d4g_lsym_ebits(lsym) + d4g_dsym_ebits(dsym)), (uint32_t)myPos, 0u, 0u);
                tokRef[ti] = (uint32_t)(o.refBase + myRef);
            } else {
                tok[ti] = make_uint2(w & 0xffu, (uint32_t)myPos);
            }
            if (sym == B.symStart) { blocks[o.blkBase + b].uStart = myPos; blocks[o.blkBase + b].refStart = myRef; }
            if (sym == B.symStart + B.symCount - 1) {
                tok[ti + 1] = make_uint2(256u, (uint32_t)(myPos + len));
                blocks[o.blkBase + b].uLen = myPos + len;         // end position; the host subtracts uStart
                blocks[o.blkBase + b].refCount = myRef + (dist ? 1 : 0);   // end record; the host subtracts refStart
                blocks[o.blkBase + b].lastTop = myPos + 1;
            }
        }
        pos += __shfl(il, 63);
        nrefBefore += __shfl(ir, 63);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_blocks: trees.c for one block — histogram, build_tree x3 (heap tie-break on depth), gen_bitlen with zlib's
// overflow repair, scan_tree, the stored / fixed / dynamic choice of _tr_flush_block, and the block's D4GState.
// One wave per block; the tree construction itself is sequential work on lane 0's scalar path.
// ---------------------------------------------------------------------------------------------------------------------
struct LzBlockOut { int32_t type; int32_t pad; long long sizeBits; long long optLen, staticLen; };

#define LZ_HEAP 573
struct LzTreeLds {
    uint16_t freq[LZ_HEAP], dad[LZ_HEAP], len[LZ_HEAP];
    int16_t heap[LZ_HEAP];
    uint8_t depth[LZ_HEAP];
    uint16_t blCount[16];
    int heapLen, heapMax;
    long long optLen, staticLen;
};

D4G_DEV bool lz_smaller(const LzTreeLds& T, int n, int m) {
    return T.freq[n] < T.freq[m] || (T.freq[n] == T.freq[m] && T.depth[n] <= T.depth[m]);
}
__device__ inline void lz_pqdownheap(LzTreeLds& T, int k) {
    int v = T.heap[k], j = k << 1;
    while (j <= T.heapLen) {
        if (j < T.heapLen && lz_smaller(T, T.heap[j + 1], T.heap[j])) j++;
        if (lz_smaller(T, v, T.heap[j])) break;
        T.heap[k] = T.heap[j];
        k = j;
        j <<= 1;
    }
    T.heap[k] = (int16_t)v;
}
// build_tree + gen_bitlen for the alphabet whose frequencies sit in T.freq[0, elems); returns max_code.
// kind: 0 literal/length (extra bits from 257, static lengths of the fixed code), 1 distance, 2 code-length code.
__device__ inline int lz_build_tree(LzTreeLds& T, int elems, int kind, int maxLength) {
    int maxCode = -1, node;
    T.heapLen = 0;
    T.heapMax = LZ_HEAP;
    for (int n = 0; n < elems; n++) {
        if (T.freq[n] != 0) { T.heap[++T.heapLen] = (int16_t)(maxCode = n); T.depth[n] = 0; }
        else T.len[n] = 0;
    }
    auto stLen = [&](int n) D4G_LAMBDA_INLINE { return kind == 0 ? (n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8) : 5; };
    while (T.heapLen < 2) {
        node = T.heap[++T.heapLen] = (int16_t)(maxCode < 2 ? ++maxCode : 0);
        T.freq[node] = 1;
        T.depth[node] = 0;
        T.optLen--;
        if (kind != 2) T.staticLen -= stLen(node);
    }
    for (int n = T.heapLen / 2; n >= 1; n--) lz_pqdownheap(T, n);
    node = elems;
    do {
        int n = T.heap[1];
        T.heap[1] = T.heap[T.heapLen--];
        lz_pqdownheap(T, 1);
        int m = T.heap[1];
        T.heap[--T.heapMax] = (int16_t)n;
        T.heap[--T.heapMax] = (int16_t)m;
        T.freq[node] = (uint16_t)(T.freq[n] + T.freq[m]);
        T.depth[node] = (uint8_t)((T.depth[n] >= T.depth[m] ? T.depth[n] : T.depth[m]) + 1);
        T.dad[n] = T.dad[m] = (uint16_t)node;
        T.heap[1] = (int16_t)node++;
        lz_pqdownheap(T, 1);
    } while (T.heapLen >= 2);
    T.heap[--T.heapMax] = T.heap[1];
    // gen_bitlen
    int h, overflow = 0;
    for (int b = 0; b <= 15; b++) T.blCount[b] = 0;
    T.len[T.heap[T.heapMax]] = 0;
    for (h = T.heapMax + 1; h < LZ_HEAP; h++) {
        int n = T.heap[h];
        int bits = T.len[T.dad[n]] + 1;
        if (bits > maxLength) { bits = maxLength; overflow++; }
        T.len[n] = (uint16_t)bits;
        if (n > maxCode) continue;
        T.blCount[bits]++;
        int xbits = 0;
        if (kind == 0) { if (n >= 257) xbits = d4g_lsym_ebits(n); }
        else if (kind == 1) xbits = d4g_dsym_ebits(n);
        else xbits = n == 16 ? 2 : n == 17 ? 3 : n == 18 ? 7 : 0;
        T.optLen += (long long)T.freq[n] * (bits + xbits);
        if (kind != 2) T.staticLen += (long long)T.freq[n] * (stLen(n) + xbits);
    }
    if (overflow > 0) {
        do {
            int bits = maxLength - 1;
            while (T.blCount[bits] == 0) bits--;
            T.blCount[bits]--;
            T.blCount[bits + 1] += 2;
            T.blCount[maxLength]--;
            overflow -= 2;
        } while (overflow > 0);
        for (int bits = maxLength; bits != 0; bits--) {
            int n = T.blCount[bits];
            while (n != 0) {
                int m = T.heap[--h];
                if (m > maxCode) continue;
                if (T.len[m] != (unsigned)bits) {
                    T.optLen += ((long long)bits - T.len[m]) * T.freq[m];
                    T.len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    return maxCode;
}

// scan_tree / send_tree: the run-length pairs zlib sends for lens[0..maxCode]; out == nullptr only counts the
// code-length-code frequencies into blFreq.
D4G_DEV int lz_scan_tree(const uint8_t* lens, int maxCode, uint16_t* blFreq, uint16_t* pairsOut, int np) {
    int prevlen = -1, nextlen = lens[0], count = 0, maxCount = 7, minCount = 4;
    if (nextlen == 0) { maxCount = 138; minCount = 3; }
    for (int n = 0; n <= maxCode; n++) {
        int curlen = nextlen;
        nextlen = n + 1 <= maxCode ? lens[n + 1] : 0xffff;   // zlib's guard value
        if (++count < maxCount && curlen == nextlen) continue;
        else if (count < minCount) {
            if (blFreq) blFreq[curlen] += (uint16_t)count;
            if (pairsOut) for (int k = 0; k < count; k++) pairsOut[np++] = pair_encode(curlen, 0, curlen);
        } else if (curlen != 0) {
            if (curlen != prevlen) {
                if (blFreq) blFreq[curlen]++;
                if (pairsOut) pairsOut[np++] = pair_encode(curlen, 0, curlen);
                count--;
            }
            if (blFreq) blFreq[16]++;
            if (pairsOut) pairsOut[np++] = pair_encode(16, count, curlen);
        } else if (count <= 10) {
            if (blFreq) blFreq[17]++;
            if (pairsOut) pairsOut[np++] = pair_encode(17, count, 0);
        } else {
            if (blFreq) blFreq[18]++;
            if (pairsOut) pairsOut[np++] = pair_encode(18, count, 0);
        }
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { maxCount = 138; minCount = 3; }
        else if (curlen == nextlen) { maxCount = 6; minCount = 3; }
        else { maxCount = 7; minCount = 4; }
    }
    return np;
}

__global__ void __launch_bounds__(64) k_lz_blocks(const LzStream* streams, const LzOutStream* outs, const LzBlockDesc* blocks, int nBlocks,
                                                  const uint2* tok, D4GState* states, LzBlockOut* res) {
    __shared__ D4GState S;
    __shared__ LzTreeLds T;
    __shared__ uint16_t blFreq[19];
    if ((int)blockIdx.x >= nBlocks) return;
    const LzBlockDesc B = blocks[blockIdx.x];
    const LzOutStream o = outs[B.out];
    const LzStream st = streams[o.stream];
    const int lane = threadIdx.x;
    const int bIdx = (int)(blockIdx.x - o.blkBase);
    for (int i = lane; i < (int)(sizeof(D4GState) / 4); i += 64) ((uint32_t*)&S)[i] = 0;
    __syncthreads();
    const uint2* tk = tok + o.tokBase + B.symStart + bIdx;
    for (long long t = lane; t < B.symCount; t += 64) {
        uint32_t x = tk[t].x;
        int dist = (int)(x >> 16);
        if (dist) {
            atomicAdd(&S.hist[d4g_len2sym((int)(x & 0x1ffu), 0)], 1u);
            atomicAdd(&S.hist[D4G_NLIT + d4g_dist2sym(dist)], 1u);
        } else atomicAdd(&S.hist[x & 0xffu], 1u);
    }
    if (lane == 0) S.hist[256] = 1;
    __syncthreads();
    if (lane == 0) {
        T.optLen = 0; T.staticLen = 0;
        // literal/length tree
        for (int n = 0; n < 286; n++) T.freq[n] = (uint16_t)S.hist[n];
        int lMax = lz_build_tree(T, 286, 0, 15);
        for (int n = 0; n <= lMax; n++) S.litLen[n] = (uint8_t)T.len[n];
        // distance tree
        for (int n = 0; n < 30; n++) T.freq[n] = (uint16_t)S.hist[D4G_NLIT + n];
        int dMax = lz_build_tree(T, 30, 1, 15);
        for (int n = 0; n <= dMax; n++) S.distLen[n] = (uint8_t)T.len[n];
        // code-length code
        for (int n = 0; n < 19; n++) blFreq[n] = 0;
        lz_scan_tree(S.litLen, lMax, blFreq, nullptr, 0);
        lz_scan_tree(S.distLen, dMax, blFreq, nullptr, 0);
        for (int n = 0; n < 19; n++) T.freq[n] = blFreq[n];
        lz_build_tree(T, 19, 2, 7);
        for (int n = 0; n < 19; n++) S.clLen[n] = (uint8_t)T.len[n];
        int maxBl;
        for (maxBl = 18; maxBl >= 3; maxBl--) if (S.clLen[D4G_CL_ORDER[maxBl]] != 0) break;
        T.optLen += 3 * ((long long)maxBl + 1) + 5 + 5 + 4;
        const long long optLenb = (T.optLen + 3 + 7) >> 3, staticLenb = (T.staticLen + 3 + 7) >> 3;
        const long long best = staticLenb <= optLenb ? staticLenb : optLenb;
        const long long storedLen = B.uLen;
        // zlib can only store a block whose first byte is still in its window (block_start >= 0).  The window base
        // when the block is flushed follows from the loop-top position of that iteration (fill_window slides by
        // 32 KiB whenever a loop top finds strstart >= base + 65274 with fewer than 262 bytes of lookahead).
        long long base = 0;
        {
            const long long top = B.isLast ? st.len : B.lastTop;
            while (true) {
                long long thr = base + 65275;                       // first loop top with lookahead < 262 in a full window
                if (st.len - base < 65536) { long long a = base + 65274, b2 = st.len - 261; thr = a > b2 ? a : b2; }
                if (top >= thr) base += 32768; else break;
            }
        }
        int type;
        if (storedLen + 4 <= best && B.uStart >= base) type = D4G_STORED;
        else if (staticLenb == best) type = D4G_FIXED;
        else type = D4G_DYNAMIC;
        long long litBits = 0;
        if (type == D4G_FIXED) {
            for (int i = 0; i < D4G_NLIT; i++) S.litLen[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 286 ? 8 : 0;
            for (int i = 0; i < D4G_NDIST; i++) S.distLen[i] = i < 30 ? 5 : 0;
            for (int i = 0; i < 32; i++) S.clLen[i] = 0;
        } else {
            S.nLit = lMax + 1;
            S.nDist = dMax + 1;
            S.nCl = maxBl + 1;
            int np = lz_scan_tree(S.litLen, lMax, nullptr, S.pairs, 0);
            np = lz_scan_tree(S.distLen, dMax, nullptr, S.pairs, np);
            S.nPairs = np;
            long long hb = 5 + 5 + 4 + 3 * S.nCl;
            for (int i = 0; i < np; i++) {
                int sym = S.pairs[i] & 31;
                hb += S.clLen[sym] + (sym >= 16 ? pair_extra_bits(sym) : 0);
            }
            S.hdrBits = hb;
        }
        for (int n = 0; n < 286; n++) if (S.hist[n]) litBits += (long long)S.hist[n] * (S.litLen[n] + (n >= 257 ? d4g_lsym_ebits(n) : 0));
        for (int n = 0; n < 30; n++) if (S.hist[D4G_NLIT + n]) litBits += (long long)S.hist[D4G_NLIT + n] * (S.distLen[n] + d4g_dsym_ebits(n));
        S.type = type == D4G_STORED ? D4G_DYNAMIC : type;
        S.litlenBits = litBits;
        S.sizeBits = S.hdrBits + litBits;
        S.valid = 1;
        S.maskSlot = 0;
        res[blockIdx.x].type = type;
        res[blockIdx.x].pad = 0;
        res[blockIdx.x].sizeBits = S.sizeBits;
        res[blockIdx.x].optLen = T.optLen;
        res[blockIdx.x].staticLen = T.staticLen;
    }
    __syncthreads();
    D4GState* g = states + blockIdx.x;
    for (int i = lane; i < (int)(sizeof(D4GState) / 4); i += 64) ((uint32_t*)g)[i] = ((uint32_t*)&S)[i];
}

// copies the temporary per-block states to the optimiser's state slots (slot 0 of every Huffman block)
__global__ void k_lz_place_states(const D4GState* tmp, const long long* dstIdx, int n, D4GState* states) {
    int b = blockIdx.x;
    if (b >= n || dstIdx[b] < 0) return;
    const uint32_t* s = (const uint32_t*)(tmp + b);
    uint32_t* d = (uint32_t*)(states + dstIdx[b]);
    for (int i = threadIdx.x; i < (int)(sizeof(D4GState) / 4); i += blockDim.x) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lz_entropy_bound: a lower bound on what ANY all-literal encoding of an input can cost — the sum, over the
// 16383-byte slices a HUFFMAN_ONLY compressor cuts it into, of the slice's zeroth-order entropy.  The optimiser never
// creates a back-reference, and merging blocks or storing them only raises that sum, so a HUFFMAN_ONLY candidate whose
// bound already exceeds the best optimised DEFAULT / FILTERED candidate cannot win CompressionUtil's strict minimum:
// it is then not optimised at all (the winner and its bytes are unaffected).  One workgroup per slice.
// ---------------------------------------------------------------------------------------------------------------------
struct LzBoundJob { const uint8_t* data; int32_t len, input; };
__global__ void __launch_bounds__(256) k_lz_entropy_bound(const LzBoundJob* jobs, double* perInput) {
    __shared__ unsigned hist[256];
    const LzBoundJob J = jobs[blockIdx.x];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < J.len; i += 256) atomicAdd(&hist[J.data[i]], 1u);
    __syncthreads();
    const unsigned f = hist[threadIdx.x];
    double v = f ? (double)f * log2((double)J.len / (double)f) : 0.0;
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0 && v > 0.0) atomicAdd(&perInput[J.input], v);
}
