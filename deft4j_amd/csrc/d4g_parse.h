// d4g_parse.h — inflate-to-tokens on the GPU, parallel across deflate blocks.
//
// Results are those of DeflateStream.parse (B/deflate/DeflateStream.java:72-126),
// DeflateBlockHuffman.initDynamicDecoder (:892-1010) and decodeStream (:778-890): token arrays,
// decoded bytes, and one initial state per Huffman block (code lengths, header RLE pairs,
// symbol histogram, bit sizes).  The reference walks the stream serially; here
//   1. k_scan_headers     tests EVERY bit position for a plausible dynamic-block header
//                         (BTYPE, HLIT/HDIST range, complete code-length code) — brute force
//                         that a 256-CU chip does in well under a millisecond per 10 MB;
//   2. k_probe_blocks     one wave per candidate decodes the block without output and reports
//                         where it ends (speculative candidates must also have complete codes);
//                         the host links candidates into the one chain that starts at bit 0 and
//                         probes the few positions the scan cannot see (fixed / stored blocks);
//   3. k_emit_blocks      one wave per confirmed block decodes again, now writing tokens at
//                         their final offsets and the block's initial state;
//   4. k_fill_src / k_jump_streams / k_resolve_streams   decoded bytes by pointer jumping: every
//                         byte points at the byte it copies (src < position), doubling collapses
//                         the chains onto literals, then one gather fills U.  No 32 KiB window is
//                         ever walked serially.
// Decoding itself keeps the reference's semantics (LUT fast path; bit-serial canonical fallback
// identical to Huffman.readSymbol, B/huffman/Huffman.java:170-197; reads past EOF fail).
#pragma once
#include <type_traits>

#include "d4g_device.h"

struct D4GStreamDesc {
    const uint8_t* data;   // compressed bytes (device, 16-byte aligned, readable 16 KiB past len)
    long long len;
    long long uBase;       // start of this stream's region in U / src
    long long uLen;
};

struct D4GProbeIn {
    int32_t stream;
    int32_t strict;        // 1: speculative candidate (dynamic only, complete codes required)
    long long bitPos;
};
struct D4GProbeOut {
    int32_t status;        // 0 ok, -1 not a (valid) block
    int32_t type;
    int32_t bfinal;
    int32_t eofHit;        // stored block ran past the end of input (bytes read as 0xff)
    long long endBit;      // first bit after the block
    long long nTok, uLen, sizeBits;
    long long needHist;    // max over back-references of (distance - bytes produced so far in the block)
    int32_t nRef;          // back-reference tokens in the block
    int32_t firstBatch;    // first record of the block's verified chunk starts (D4GChunkBatch chain), < 0: none
};
// What the probe learned about a batch of 64 chunks, kept for the emit pass: per lane the verified start (bits from
// the block's first token) and its counts, so the emit decodes every chunk exactly once.
struct D4GChunkBatch {
    int32_t next;          // following batch of the same block, -1: last
    int32_t pad[3];
    uint4 rec[64];         // .x start, .y tokens | records << 10 | stop flag << 20, .z decoded bytes, .w token bits
};
struct D4GChunkPool { D4GChunkBatch* batches; unsigned* next; unsigned cap; };
struct D4GEmitIn {
    int32_t stream;
    int32_t type;
    long long bitPos;
    long long tokStart;    // absolute index into tok
    long long uStart;      // stream-relative offset of the block's decoded bytes
    long long uLen;
    long long stateIdx;    // absolute index into the state pool (slot 0 of the block), -1 for stored
    long long sizeBits;    // from the probe
    long long refStart;    // absolute index of the block's first back-reference record
    int32_t firstBatch;    // the probe's chunk records for this block (< 0: decode speculatively again)
    int32_t pad;
};
#define D4G_LUT_BITS 10
#define D4G_INCH 33024   // staged input: one batch of 512 chunks of D4G_CHUNK_BITS (32 KiB) plus the overshoot of the last token

struct D4GDecTab {          // canonical decoder of one alphabet
    uint16_t lut[1 << D4G_LUT_BITS];  // sym | len<<9, 0xffff = use the bit-serial path
    uint16_t sorted[D4G_NLIT];        // symbols ordered by (length, index)
    int first[16], count[16], offs[16];
    int useLut;
    int complete;           // Kraft sum == 1
    int nCodes;
};

#define D4G_PARSE_MAXTHREADS 512
struct D4GParseLds {
    D4GState st;
    D4GDecTab lit, dist, cl;
    alignas(16) uint8_t inbuf[D4G_INCH + 16];
    // workgroup-wide exchange of the chunk decoders: exits / stop flags, per-wave scan totals, broadcasts
    int xExit[D4G_PARSE_MAXTHREADS], xFlag[D4G_PARSE_MAXTHREADS];
    unsigned wsN[8], wsU[8], wsR[8], wsB[8];
    int wsM[8];
    int anyDirty, firstStop;
    long long bc;
};

// Bit reader owned by lane 0 (B/io/BitInputStream.java:59-82: LSB-first).
struct D4GBitReader {
    long long inBase;      // absolute byte index of inbuf[0] (multiple of 16)
    long long nbits;       // total bits of the stream
    // chunk-relative 32-bit state: the decode loop does no 64-bit position arithmetic
    int rel;               // offset in inbuf of the next byte to load into buf
    int posRel;            // bits consumed, relative to bit 8*inBase
    int limitRel;          // stream end, relative to bit 8*inBase (clamped)
    uint64_t buf;
    int cnt;
    __device__ long long pos() const { return inBase * 8 + posRel; }
    __device__ void reset_to(const uint8_t* inbuf, long long bitpos) {
        long long lim = nbits - inBase * 8;
        limitRel = lim > 0x3fffffff ? 0x3fffffff : (int)lim;
        posRel = (int)(bitpos - inBase * 8);
        rel = posRel >> 3;
        buf = 0;
        cnt = 0;
        // byte loads up to the next 4-byte boundary of the staged chunk, then aligned words
        while (rel & 3) {
            uint64_t v = (rel >= 0 && rel < D4G_INCH + 16) ? inbuf[rel] : 0;
            buf |= v << cnt;
            cnt += 8;
            rel++;
        }
        fill(inbuf);
        int sh = posRel & 7;
        buf >>= sh;
        cnt -= sh;
        fill(inbuf);
    }
    // Guarantees at least 33 valid bits (enough for one code + its extra bits).
    __device__ void fill(const uint8_t* inbuf) {
        while (cnt <= 32) {
            uint64_t v = (rel >= 0 && rel + 4 <= D4G_INCH + 16) ? *(const uint32_t*)(inbuf + rel) : 0;
            buf |= v << cnt;
            cnt += 32;
            rel += 4;
        }
    }
    __device__ int avail() const { return limitRel - posRel; }
    __device__ bool have(int n) const { return posRel + n <= limitRel; }
    __device__ void skip(int n) { buf >>= n; cnt -= n; posRel += n; }
    __device__ bool near_end() const { return rel + 1024 > D4G_INCH; }
};

#ifndef D4G_CHUNK_BITS
#define D4G_CHUNK_BITS 512   // bits per lane and pass of the wave-wide token decoder
#endif
static_assert(512 * D4G_CHUNK_BITS + 160 <= (D4G_INCH + 16) * 8, "a batch of 512 chunks must fit the staged input window");
static_assert(D4G_CHUNK_BITS + 48 < 1024, "tokens per chunk must fit the 10-bit field of a chunk record");
// 64 bits of the staged input starting at bit `posRel` (any lane, any position inside the staged chunk)
__device__ __forceinline__ uint64_t d4g_peek64(const uint8_t* inbuf, int posRel) {
    int a = (posRel >> 3) & ~3;
    int sh = posRel - a * 8;   // 0..31
    const uint32_t* w = (const uint32_t*)(inbuf + a);
    uint64_t lo = ((uint64_t)w[1] << 32) | w[0];
    uint64_t hi = w[2];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// Huffman.buildCodes (B/huffman/Huffman.java:35-64) + decoder tables.  Lane 0 prepares the
// canonical structure, all lanes fill the LUT.
__device__ void d4g_build_decoder(D4GDecTab* T, const uint8_t* lens, int n) {
    const int lane = threadIdx.x, nl = blockDim.x;   // every thread of the workgroup helps
    __syncthreads();
    if (lane < 16) { T->count[lane] = 0; T->first[lane] = 0; T->offs[lane] = 0; }
    __syncthreads();
    for (int i = lane; i < n; i += nl)
        if (lens[i] > 0 && lens[i] < 16) atomicAdd(&T->count[lens[i]], 1);
    __syncthreads();
    if (lane == 0) {
        int nc = 0, next = 0, lastShift = 0, o = 0;
        long long kraft = 0;
        for (int l = 1; l <= 15; l++) {
            T->offs[l] = o;
            o += T->count[l];
            nc += T->count[l];
            if (T->count[l]) {
                next <<= (l - lastShift);
                lastShift = l;
                T->first[l] = next;
                next += T->count[l];
                kraft += (long long)T->count[l] << (15 - l);
            }
        }
        T->useLut = kraft <= (1 << 15);
        T->complete = kraft == (1 << 15);
        T->nCodes = nc;
    }
    __syncthreads();
    // symbols ordered by (length, index): symbol i sits after the lower-numbered symbols of its length
    for (int i = lane; i < n; i += nl) {
        const int l = lens[i];
        if (l > 0 && l < 16) {
            int rank = 0;
            for (int j = 0; j < i; j++) rank += lens[j] == l;
            T->sorted[T->offs[l] + rank] = (uint16_t)i;
        }
    }
    __syncthreads();
    for (int i = lane; i < (1 << D4G_LUT_BITS); i += nl) T->lut[i] = 0xffff;
    __syncthreads();
    if (T->useLut) {
        for (int l = 1; l <= D4G_LUT_BITS; l++) {
            int cnt = T->count[l];
            for (int k = lane; k < cnt; k += nl) {
                int code = T->first[l] + k;
                int sym = T->sorted[T->offs[l] + k];
                unsigned r = 0;
                for (int bI = 0; bI < l; bI++) r |= ((code >> bI) & 1u) << (l - 1 - bI);
                for (unsigned e = r; e < (1u << D4G_LUT_BITS); e += (1u << l)) T->lut[e] = (uint16_t)(sym | (l << 9));
            }
        }
    }
    __syncthreads();
}

// Decode one symbol from `bits` (lane 0), `avail` = bits left in the stream.
// Returns sym | len << 16, or -1 on failure.  (Packed return instead of an out-pointer: a generic
// pointer to a private variable next to LDS table reads trips a gfx950 backend assertion in ROCm 7.2.)
__device__ __forceinline__ int d4g_decode_sym_packed(const D4GDecTab* T, uint64_t bits, int avail) {
    unsigned e = T->lut[bits & ((1u << D4G_LUT_BITS) - 1)];
    if (e != 0xffff) {
        int l = e >> 9;
        if (l > avail) return -1;
        return (int)(e & 511) | (l << 16);
    }
    int code = 0;
    for (int l = 1; l <= 15; l++) {  // Huffman.readSymbol, bit-serial
        if (l > avail) return -1;
        code = (code << 1) | (int)((bits >> (l - 1)) & 1);
        if (T->count[l] && code >= T->first[l] && code < T->first[l] + T->count[l])
            return (int)T->sorted[T->offs[l] + code - T->first[l]] | (l << 16);
    }
    return -1;
}
#define D4G_DECODE(T, bits, avail, symVar, lenVar)            \
    do {                                                      \
        int _r = d4g_decode_sym_packed(T, bits, avail);       \
        symVar = _r < 0 ? -1 : (_r & 0xffff);                 \
        lenVar = _r < 0 ? 0 : (_r >> 16);                     \
    } while (0)

// ---------------------------------------------------------------------------------------
// 1. Header scan.  One thread per input byte tests its 8 bit positions.
// ---------------------------------------------------------------------------------------
struct D4GScanTile { int32_t stream; int32_t pad; long long byteStart; };
#define D4G_SCAN_TILE 2048
#define D4G_SCAN_LOCAL 1024   // candidates a workgroup collects before it touches the global list

__global__ void __launch_bounds__(256) k_scan_headers(const D4GStreamDesc* streams, const D4GScanTile* tiles, D4GProbeIn* cands,
                                                      unsigned* nCands, unsigned capCands) {
    alignas(16) __shared__ uint8_t buf[D4G_SCAN_TILE + 32];
    __shared__ uint32_t lpos[D4G_SCAN_LOCAL];   // bit offsets (inside the tile) of this workgroup's candidates
    __shared__ unsigned lcount, gbase;
    if (threadIdx.x == 0) lcount = 0;
    const D4GScanTile tile = tiles[blockIdx.x];
    const D4GStreamDesc sd = streams[tile.stream];
    for (int i = threadIdx.x * 16; i < D4G_SCAN_TILE + 32; i += blockDim.x * 16)
        *(uint4*)(buf + i) = *(const uint4*)(sd.data + tile.byteStart + i);  // the input buffer is padded past len
    __syncthreads();
    long long nbits = sd.len * 8;
    int lane = threadIdx.x & 63;
    for (int b = threadIdx.x; b < D4G_SCAN_TILE; b += blockDim.x) {
        // 96 bits starting at byte b: four aligned words of the tile, shifted into place
        const uint32_t* bw = (const uint32_t*)buf + (b >> 2);
        const int bsh = (b & 3) * 8;
        const uint32_t w0 = bw[0], w1 = bw[1], w2 = bw[2], w3 = bw[3];
        uint64_t lo = (uint64_t)d4g_alignbit(w1, w0, bsh) | ((uint64_t)d4g_alignbit(w2, w1, bsh) << 32);
        uint32_t hi = d4g_alignbit(w3, w2, bsh);
        long long bit0 = (tile.byteStart + b) * 8;
        for (int s = 0; s < 8; s++) {
            uint64_t v = s ? ((lo >> s) | ((uint64_t)hi << (64 - s))) : lo;   // window bits 0..63
            uint64_t w = (v >> 32) | ((uint64_t)(hi >> s) << 32);             // window bits 32..95
            long long p = bit0 + s;
            bool ok = ((v >> 1) & 3) == 2 && ((v >> 3) & 31) <= 29 && ((v >> 8) & 31) <= 29;
            int ncl = (int)((v >> 13) & 15) + 4;
            ok = ok && (p + 17 + 3 * ncl <= nbits);
            if (ok) {
                // the code-length code must be complete: sum 2^(7-l) == 128, at least two codes.  Branch-free over the
                // 3-bit fields: bit planes of the (masked) 57-bit field, one popcount per length value
                uint64_t f = (v >> 17) | ((w >> 32) << 47);           // window bits 17..80
                f &= (1ULL << (3 * ncl)) - 1;
                const uint64_t M = 0x1249249249249249ULL;              // bit 0 of every 3-bit field
                const uint64_t p0 = f & M, p1 = (f >> 1) & M, p2 = (f >> 2) & M;
                const uint64_t n0 = ~p0, n1 = ~p1, n2 = ~p2;
                int kraft = 64 * __popcll(p0 & n1 & n2) + 32 * __popcll(n0 & p1 & n2) + 16 * __popcll(p0 & p1 & n2) +
                            8 * __popcll(n0 & n1 & p2) + 4 * __popcll(p0 & n1 & p2) + 2 * __popcll(n0 & p1 & p2) + __popcll(p0 & p1 & p2);
                int used = __popcll(p0 | p1 | p2);
                ok = kraft == 128 && used >= 2;
            }
            // candidates are collected per workgroup (one global atomic per tile instead of one per hit: tens of
            // thousands of atomics on a single counter were most of this kernel's time)
            unsigned long long m = __ballot(ok);
            if (m) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&lcount, (unsigned)__popcll(m));
                base = __shfl(base, 0);
                if (ok) {
                    unsigned idx = base + (unsigned)__popcll(m & ((1ULL << lane) - 1));
                    if (idx < D4G_SCAN_LOCAL) lpos[idx] = (uint32_t)(p - tile.byteStart * 8);
                    else {   // (a tile with more plausible headers than the local list holds: straight to the global list)
                        unsigned g = atomicAdd(nCands, 1u);
                        if (g < capCands) { D4GProbeIn c; c.stream = tile.stream; c.strict = 1; c.bitPos = p; cands[g] = c; }
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned nl = lcount < D4G_SCAN_LOCAL ? lcount : D4G_SCAN_LOCAL;
    if (threadIdx.x == 0) gbase = nl ? atomicAdd(nCands, nl) : 0u;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nl; i += blockDim.x) {
        unsigned g = gbase + i;
        if (g < capCands) { D4GProbeIn c; c.stream = tile.stream; c.strict = 1; c.bitPos = tile.byteStart * 8 + lpos[i]; cands[g] = c; }
    }
}

// ---------------------------------------------------------------------------------------
// 1b. Header pre-filter.  The scan's candidates are almost all false positives (a plausible prolog and a
// complete code-length code at a random bit position); the probe would spend a whole wave and ~300 serial symbol
// decodes on each.  Here one LANE per candidate walks the coded code lengths straight from global memory with a
// private 128-entry decode table, keeping only the Kraft sums: a candidate survives when its literal/length code is
// complete with an end-of-block code and its distance code is complete or has at most one code — the conditions
// the strict probe applies (d4g_parse_block), so nothing the probe would accept is dropped.  Survivors are
// compacted for the probe.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_prefilter_headers(const D4GStreamDesc* streams, const D4GProbeIn* in, unsigned n, D4GProbeIn* kept,
                                                          unsigned* nKept) {
    __shared__ uint8_t lut[64 * 128];   // per lane: 7-bit window -> sym | len << 5
    const int lane = threadIdx.x & 63;
    const unsigned idx = blockIdx.x * 64 + lane;
    bool ok = idx < n;
    D4GProbeIn pi;
    pi.stream = 0; pi.strict = 1; pi.bitPos = 0;
    if (ok) pi = in[idx];
    const D4GStreamDesc sd = streams[pi.stream];
    const long long nbits = sd.len * 8;
    long long pos = pi.bitPos;
    const uint32_t* words = (const uint32_t*)sd.data;   // 16-byte aligned, readable past len
    auto peek = [&](long long p) -> uint64_t {          // 57+ valid bits at bit p
        long long w = p >> 5;
        int sh = (int)(p & 31);
        uint64_t lo = words[w] | ((uint64_t)words[w + 1] << 32);
        uint64_t hi = words[w + 2];
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    };
    __shared__ uint8_t clens[64 * 20];
    uint8_t* T = lut + lane * 128;
    uint8_t* CLN = clens + lane * 20;
    int nLit = 0, nDist = 0;
    if (ok) {
        uint64_t b = peek(pos);
        nLit = (int)((b >> 3) & 31) + 257;
        nDist = (int)((b >> 8) & 31) + 1;
        int nCl = (int)((b >> 13) & 15) + 4;
        ok = ((b >> 1) & 3) == 2 && nLit <= 286 && nDist <= 30 && pos + 17 + 3 * nCl <= nbits;
        if (ok) {
            pos += 17;
            // 19 x 3 bits = 57 bits: one window.  Counts and next codes per length live in 8-bit fields of a word
            // (no dynamically indexed register arrays); the lengths themselves in the lane's LDS row.
            uint64_t c = peek(pos);
            uint64_t cnt64 = 0;
            for (int k = 0; k < 19; k++) CLN[k] = 0;
            for (int i = 0; i < nCl; i++) {
                int l = (int)((c >> (3 * i)) & 7);
                CLN[D4G_CL_ORDER[i]] = (uint8_t)l;
                if (l) cnt64 += 1ULL << (8 * l);
            }
            pos += 3 * nCl;
            // canonical codes (Huffman.buildCodes), LUT over 7 bits (LSB-first windows: codes are bit-reversed)
            uint64_t next64 = 0;
            int code = 0;
            for (int l = 1; l <= 7; l++) {
                code = (code + (l > 1 ? (int)((cnt64 >> (8 * (l - 1))) & 255) : 0)) << 1;
                next64 |= (uint64_t)(code & 255) << (8 * l);
            }
            for (int e = 0; e < 128; e++) T[e] = 0xff;
            for (int k = 0; k < 19; k++) {
                int l = CLN[k];
                if (l) {
                    int cd = (int)((next64 >> (8 * l)) & 255);
                    next64 += 1ULL << (8 * l);
                    unsigned r = 0;
                    for (int bI = 0; bI < l; bI++) r |= ((cd >> bI) & 1u) << (l - 1 - bI);
                    for (unsigned e = r; e < 128; e += (1u << l)) T[e] = (uint8_t)(k | (l << 5));
                }
            }
        }
    }
    if (ok) {
        const int combined = nLit + nDist;
        long long kraftLit = 0, kraftDist = 0;
        int i = 0, prev = 0, nDistCodes = 0, eob = 0;
        while (i < combined) {
            if (pos + 14 > nbits + 64) { ok = false; break; }
            uint64_t b = peek(pos);
            unsigned e = T[b & 127];
            if (e == 0xff) { ok = false; break; }
            int sym = e & 31, l = e >> 5;
            pos += l;
            b >>= l;
            int run = 1, value = sym;
            if (sym == 16) {
                if (i < 1) { ok = false; break; }
                run = (int)(b & 3) + 3; pos += 2; value = prev;
            } else if (sym == 17) {
                run = (int)(b & 7) + 3; pos += 3; value = 0;
            } else if (sym == 18) {
                run = (int)(b & 127) + 11; pos += 7; value = 0;
            }
            if (pos > nbits || i + run > combined) { ok = false; break; }
            if (value) {
                // the run may span the literal/distance boundary
                int inLit = i < nLit ? (i + run <= nLit ? run : nLit - i) : 0;
                kraftLit += (long long)inLit << (15 - value);
                kraftDist += (long long)(run - inLit) << (15 - value);
                nDistCodes += run - inLit;
                if (i <= 256 && i + run > 256) eob = 1;
            }
            prev = value;
            i += run;
        }
        ok = ok && kraftLit == (1 << 15) && eob && (kraftDist == (1 << 15) || nDistCodes <= 1);
    }
    unsigned long long m = __ballot(ok);
    if (m) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(nKept, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (ok) kept[base + (unsigned)__popcll(m & ((1ULL << lane) - 1))] = pi;
    }
}

// ---------------------------------------------------------------------------------------
// 2./3. Block parser: one wave per block.  EMIT = false probes (counts, end position),
// EMIT = true writes tokens and the block's initial state.
// ---------------------------------------------------------------------------------------
struct D4GParseOut {
    uint2* tok;
    uint8_t* U;
    D4GState* states;
    uint4* refs;        // back-reference records (d4g_types.h), in token order; .z/.w are filled in after the bytes are resolved
    uint32_t* tokRef;   // per token: index of its back-reference record (written for back-references only)
};

template <bool EMIT>
__device__ __forceinline__ void d4g_parse_block(const D4GStreamDesc& sd, long long bitPos, int strict, D4GProbeOut& po,
                                const D4GEmitIn* em, const D4GParseOut& out, const D4GChunkPool& pool) {
    __shared__ D4GParseLds L;
    const int tid = threadIdx.x, NL = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = NL >> 6;
    D4GBitReader br;   // thread 0's
    br.inBase = 0;
    br.nbits = sd.len * 8;
    br.rel = 0; br.posRel = 0; br.limitRel = 0; br.buf = 0; br.cnt = 0;
    D4GState* S = &L.st;
    // workgroup-wide exchange of small values (all threads call; the value of thread 0 / a reduction comes back to all)
    auto bcast = [&](long long v) D4G_LAMBDA_INLINE {
        __syncthreads();
        if (tid == 0) L.bc = v;
        __syncthreads();
        return L.bc;
    };
    auto stage = [&](long long bitpos) {
        long long base = (bitpos >> 3) & ~15LL;
        if (base > sd.len) base = sd.len & ~15LL;
        __syncthreads();
        for (int i = tid * 16; i < D4G_INCH + 16; i += NL * 16) *(uint4*)(L.inbuf + i) = *(const uint4*)(sd.data + base + i);
        __syncthreads();
        br.inBase = base;
        if (tid == 0) br.reset_to(L.inbuf, bitpos);
    };
    po.nRef = 0; po.firstBatch = -1;
    po.status = -1; po.type = 0; po.bfinal = 0; po.eofHit = 0; po.endBit = 0; po.nTok = 0; po.uLen = 0; po.sizeBits = 0; po.needHist = 0;
    stage(bitPos);
    long long pk = 0;
    if (tid == 0) {
        if (!br.have(3)) pk = -1;
        else { pk = (long long)(br.buf & 7); br.skip(3); }
    }
    pk = bcast(pk);
    if (pk < 0) return;
    po.bfinal = (int)(pk & 1);
    int btype = (int)(pk >> 1);
    po.type = btype;
    if (btype == 3) return;
    if (strict && btype != 2) return;
    if (btype == 0) {
        // DeflateBlockUncompressed.parse — B/deflate/DeflateBlockUncompressed.java:23-36
        long long r = 0, p = 0;
        if (tid == 0) {
            p = (br.pos() + 7) & ~7LL;
            if (p + 32 > br.nbits) r = -1;
            else {
                br.reset_to(L.inbuf, p);
                int len = (int)(br.buf & 0xffff), nlen = (int)((br.buf >> 16) & 0xffff);
                r = nlen != ((~len) & 0xffff) ? -1 : len;
            }
        }
        r = bcast(r);
        p = bcast(p);
        if (r < 0) return;
        int len = (int)r;
        long long bytePos = (p + 32) >> 3;
        if (EMIT) {
            for (int k = tid; k < len; k += NL) {
                // bytes past the end of input read as (byte)-1 in the reference (BitInputStreamUtil.readFromBIS)
                uint8_t v = (bytePos + k < sd.len) ? sd.data[bytePos + k] : 0xff;
                out.U[sd.uBase + em->uStart + k] = v;
            }
        }
        long long np = (bytePos + len) * 8;
        if (np > br.nbits) { np = br.nbits; po.eofHit = 1; }
        po.endBit = np;
        po.uLen = len;
        po.status = 0;
        return;
    }
    for (int i = tid; i < (int)(sizeof(D4GState) / 4); i += NL) ((uint32_t*)S)[i] = 0;
    __syncthreads();
    if (btype == 1) {
        // The fixed code (HuffmanTable.LIT, B/huffman/HuffmanTable.java:166-209) is the RFC 1951 code over
        // 288 symbols; 286/287 take code space but are not decodable symbols (decodeStream rejects > 285).
        for (int i = tid; i < D4G_NLIT; i += NL) S->litLen[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = tid; i < D4G_NDIST; i += NL) S->distLen[i] = i < 30 ? 5 : 0;
        if (tid == 0) S->type = D4G_FIXED;
        __syncthreads();
    } else {
        // initDynamicDecoder — DeflateBlockHuffman.java:892-1010
        long long r = 0;
        if (tid == 0) {
            if (!br.have(14)) r = -1;
            else {
                br.fill(L.inbuf);
                S->nLit = (int)(br.buf & 31) + 257;
                S->nDist = (int)((br.buf >> 5) & 31) + 1;
                S->nCl = (int)((br.buf >> 10) & 15) + 4;
                br.skip(14);
                if (S->nLit > 288) r = -1;
                if (strict && (S->nLit > 286 || S->nDist > 30)) r = -1;
                if (r == 0 && !br.have(3 * S->nCl)) r = -1;
                if (r == 0) {
                    for (int i = 0; i < S->nCl; i++) {
                        br.fill(L.inbuf);
                        S->clLen[D4G_CL_ORDER[i]] = (uint8_t)(br.buf & 7);
                        br.skip(3);
                    }
                }
                S->type = D4G_DYNAMIC;
                S->hdrBits = 5 + 5 + 4 + 3 * S->nCl;
            }
        }
        r = bcast(r);
        if (r < 0) return;
        d4g_build_decoder(&L.cl, S->clLen, 19);
        if (tid == 0) {
            int i = 0, np = 0;
            int combined = S->nLit + S->nDist;
            while (i < combined && r == 0) {
                br.fill(L.inbuf);
                int cl = 0, sym;
                D4G_DECODE(&L.cl, br.buf, br.avail(), sym, cl);
                if (sym < 0 || sym > 18) { r = -1; break; }
                br.skip(cl);
                S->hdrBits += cl;
                int run = 0, value = sym;
                if (sym == 16) {
                    if (i < 1 || !br.have(2)) { r = -1; break; }
                    run = (int)(br.buf & 3) + 3;
                    br.skip(2); S->hdrBits += 2;
                    value = (i - 1 < S->nLit) ? S->litLen[i - 1] : S->distLen[i - 1 - S->nLit];
                } else if (sym == 17) {
                    if (!br.have(3)) { r = -1; break; }
                    run = (int)(br.buf & 7) + 3;
                    br.skip(3); S->hdrBits += 3;
                    value = 0;
                } else if (sym == 18) {
                    if (!br.have(7)) { r = -1; break; }
                    run = (int)(br.buf & 127) + 11;
                    br.skip(7); S->hdrBits += 7;
                    value = 0;
                }
                int cnt = run ? run : 1;
                if (i + cnt > combined) { r = -1; break; }
                for (int k = 0; k < cnt; k++, i++) {
                    if (i < S->nLit) S->litLen[i] = (uint8_t)value;
                    else S->distLen[i - S->nLit] = (uint8_t)value;
                }
                S->pairs[np++] = pair_encode(sym, run, value);
            }
            S->nPairs = np;
        }
        r = bcast(r);
        if (r < 0) return;
    }
    d4g_build_decoder(&L.lit, S->litLen, btype == 1 ? 288 : S->nLit);
    d4g_build_decoder(&L.dist, S->distLen, btype == 1 ? 30 : S->nDist);
    if (btype == 1 && tid == 0) { S->litLen[286] = 0; S->litLen[287] = 0; }
    __syncthreads();
    if (strict) {
        // speculative candidates must look like an encoder's output: complete literal/length code with an
        // EOB code, complete (or at most one-code) distance code
        bool good = L.lit.complete && S->litLen[256] > 0 && (L.dist.complete || L.dist.nCodes <= 1);
        if (!good) return;
    }
    // ---- decodeStream — DeflateBlockHuffman.java:778-890, by the whole workgroup ----
    // The block's bits are cut into chunks of D4G_CHUNK_BITS; NL (= workgroup size) chunks make a batch.  Thread i
    // decodes chunk i from a start position: thread 0 from the known token boundary, the others from a guess (their
    // chunk's first bit).  A decoder started off a token boundary usually falls into step with the true token
    // sequence after a few codes, so its exit position (first token boundary past the chunk's end) is usually
    // right.  Each pass hands every thread its left neighbour's exit as the new start and re-decodes the chunks
    // whose start changed; when a pass changes nothing, start(i+1) == exit(i) for all i and start(0) is true, so
    // every start is a true token boundary — the check is exact, at worst after NL passes.  With 512 threads a batch
    // is 32 KiB of input: most blocks are one batch, decoded by eight waves at once instead of one wave eight times.
    const long long tokensStart = bcast(br.pos());
    long long s0 = tokensStart;   // true token boundary where the batch begins
    long long G = s0;             // grid origin of the batch: chunk i covers [G + i*C, G + (i+1)*C)
    unsigned nTok = 0, nU = 0, nRef = 0, litlenBits = 0;
    int needHist = 0;
    uint2* tokOut = EMIT ? out.tok + em->tokStart : nullptr;
    uint4* refOut = EMIT ? out.refs + em->refStart : nullptr;
    uint32_t* tokRefOut = EMIT ? out.tokRef + em->tokStart : nullptr;
    const uint32_t refBase32 = EMIT ? (uint32_t)em->refStart : 0u;
    const unsigned uStart32 = EMIT ? (unsigned)em->uStart : 0u;
    long long endBit = 0;
    constexpr int C = D4G_CHUNK_BITS;
    // one chunk: WRITE = false counts, WRITE = true also stores the tokens (the thread's output offsets are known then)
    auto decode_chunk = [&](auto writeTag, int start, int endc, int limRel, unsigned tokAt, unsigned uAt, unsigned refAt, int& exitp,
                            unsigned& n, unsigned& u, unsigned& r, unsigned& lb, int& need, int& fl) D4G_LAMBDA_INLINE {
        constexpr bool WRITE = decltype(writeTag)::value;
        int pos = start;
        n = 0; u = 0; r = 0; lb = 0; need = -0x40000000; fl = 0;
        while (pos < endc) {
            uint64_t bits = d4g_peek64(L.inbuf, pos);
            int avail = limRel - pos;
            int cl = 0, sym;
            D4G_DECODE(&L.lit, bits, avail, sym, cl);
            if (sym < 0 || sym > 285) { fl = 2; break; }
            int used = cl;
            if (sym <= 256) {
                if (WRITE) {
                    atomicAdd(&S->hist[sym], 1u);
                    tokOut[tokAt + n] = make_uint2((uint32_t)sym, uStart32 + uAt + u);
                }
                n++;
                lb += (unsigned)used;
                pos += used;
                if (sym == 256) { fl = 1; break; }
                u++;
            } else {
                int eb = d4g_lsym_ebits(sym);
                int len = d4g_lsym_base(sym) + (int)((bits >> cl) & ((1u << eb) - 1));
                used += eb;
                int edge = (len == 258 && sym == 284);
                int dcl = 0, ds;
                D4G_DECODE(&L.dist, bits >> used, avail - used, ds, dcl);
                if (ds < 0 || ds > 29) { fl = 2; break; }
                int deb = d4g_dsym_ebits(ds);
                int dist = d4g_dsym_base(ds) + (int)((bits >> (used + dcl)) & ((1u << deb) - 1));
                used += dcl + deb;
                if (used > avail) { fl = 2; break; }
                if (dist - (int)u > need) need = dist - (int)u;
                if (WRITE) {
                    atomicAdd(&S->hist[sym], 1u);
                    atomicAdd(&S->hist[D4G_NLIT + ds], 1u);
                    tokOut[tokAt + n] = make_uint2((uint32_t)len | ((uint32_t)edge << 15) | ((uint32_t)dist << 16), uStart32 + uAt + u);
                    refOut[refAt + r] = make_uint4(d4g_ref_pack(len, sym, ds, eb + deb), uStart32 + uAt + u, 0u, 0u);
                    tokRefOut[tokAt + n] = refBase32 + refAt + r;
                }
                n++; r++;
                lb += (unsigned)used;
                pos += used;
                u += (unsigned)len;
            }
        }
        exitp = pos;
    };
    int recBatch = EMIT ? em->firstBatch : -1;   // emit: next recorded batch (its first record); probe: last recorded batch
    int firstBatch = -1;
    bool recording = !EMIT && pool.batches != nullptr;
    const bool replay = EMIT && recBatch >= 0 && pool.batches != nullptr;
    while (true) {
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (replay) {   // the probe's verified starts and counts of this batch: one record per wave, nw consecutive records
            rec = pool.batches[recBatch + wave].rec[lane];
            __syncthreads();
            if (tid == 0) L.bc = (long long)rec.x;
            __syncthreads();
            s0 = tokensStart + L.bc;
        }
        // the staged input must cover the batch: NL chunks, one token of overshoot, the 12-byte window of a peek
        {
            long long baseBits = br.inBase * 8;
            bool covers = s0 >= baseBits && (G - baseBits) + (long long)NL * C + 64 + 96 <= (long long)(D4G_INCH + 16) * 8;
            if (!covers) stage(s0);   // workgroup-uniform decision
        }
        const long long baseBits = br.inBase * 8;
        const int Grel = (int)(G - baseBits), s0rel = (int)(s0 - baseBits);
        long long lim = br.nbits - baseBits;
        const int limRel = lim > 0x3fffffff ? 0x3fffffff : (int)lim;
        int start = tid == 0 ? s0rel : Grel + tid * C;
        const int endc = Grel + (tid + 1) * C;
        int exitp = start, need = 0, fl = 0;
        unsigned n = 0, u = 0, r = 0, lb = 0;
        bool dirty = true;
        if (replay) {
            start = (int)(tokensStart + (long long)rec.x - baseBits);
            n = rec.y & 1023u; r = (rec.y >> 10) & 1023u; fl = (int)(rec.y >> 20); u = rec.z; lb = rec.w;
            need = -0x40000000;   // (the host checked the distances at probe time)
        } else {
            for (int pass = 0; pass < NL + 2; pass++) {
                if (dirty) decode_chunk(std::false_type{}, start, endc, limRel, 0u, 0u, 0u, exitp, n, u, r, lb, need, fl);
                // every thread takes its left neighbour's exit (and stop flag)
                __syncthreads();
                L.xExit[tid] = exitp;
                L.xFlag[tid] = fl;
                if (tid == 0) L.anyDirty = 0;
                __syncthreads();
                const int pe = tid ? L.xExit[tid - 1] : exitp, pfl = tid ? L.xFlag[tid - 1] : fl;
                dirty = tid > 0 && pfl == 0 && pe != start;
                if (dirty) { start = pe; L.anyDirty = 1; }
                __syncthreads();
                if (!L.anyDirty) break;
            }
        }
        // the block ends (or fails) in the first thread that stopped early; threads up to it hold true tokens
        __syncthreads();
        if (tid == 0) L.firstStop = NL;
        L.xExit[tid] = exitp;
        L.xFlag[tid] = fl;
        __syncthreads();
        if (fl != 0) atomicMin(&L.firstStop, tid);
        __syncthreads();
        const int f = L.firstStop;
        const bool valid = tid <= f;
        if (f < NL && L.xFlag[f] == 2) return;   // invalid code or out of input: the block does not parse
        unsigned pn = valid ? n : 0u, pu = valid ? u : 0u, pr = valid ? r : 0u, plb = valid ? lb : 0u;
        unsigned sn = pn, su = pu, sr = pr;   // inclusive scans: inside the wave, then across waves
        for (int d = 1; d < 64; d <<= 1) {
            unsigned a = __shfl_up(sn, d), b = __shfl_up(su, d), c2 = __shfl_up(sr, d);
            if (lane >= d) { sn += a; su += b; sr += c2; }
        }
        const int plbw = wave_sum_i32((int)plb);
        if (lane == 63) { L.wsN[wave] = sn; L.wsU[wave] = su; L.wsR[wave] = sr; }
        if (lane == 0) L.wsB[wave] = (unsigned)plbw;
        __syncthreads();
        unsigned offN = 0, offU = 0, offR = 0, totN = 0, totU = 0, totR = 0, totB = 0;
        for (int w = 0; w < nw; w++) {
            if (w < wave) { offN += L.wsN[w]; offU += L.wsU[w]; offR += L.wsR[w]; }
            totN += L.wsN[w]; totU += L.wsU[w]; totR += L.wsR[w]; totB += L.wsB[w];
        }
        sn += offN; su += offU; sr += offR;
        {
            int nd = valid && r ? need - (int)(nU + (su - pu)) : -0x40000000;
            nd = wave_max_i32(nd);
            __syncthreads();
            if (lane == 0) L.wsM[wave] = nd;
            __syncthreads();
            for (int w = 0; w < nw; w++) if (L.wsM[w] > needHist) needHist = L.wsM[w];
        }
        if (EMIT && valid) {
            int e2, need2, fl2;
            unsigned n2, u2, r2, lb2;
            decode_chunk(std::true_type{}, start, endc, limRel, nTok + (sn - pn), nU + (su - pu), nRef + (sr - pr), e2, n2, u2, r2, lb2, need2,
                         fl2);
            if (replay) exitp = e2;
        }
        if (replay) {   // (the replayed exits were not exchanged yet)
            __syncthreads();
            L.xExit[tid] = exitp;
            __syncthreads();
        }
        if (recording) {   // leave the verified chunks to the emit pass: nw consecutive records, one per wave
            __syncthreads();
            if (tid == 0) L.bc = (long long)atomicAdd(pool.next, (unsigned)nw);
            __syncthreads();
            const unsigned idx = (unsigned)L.bc;
            if (idx + (unsigned)nw <= pool.cap) {
                D4GChunkBatch* bt = pool.batches + idx + wave;
                bt->rec[lane] = make_uint4((uint32_t)(baseBits + start - tokensStart), n | (r << 10) | ((unsigned)fl << 20), u, lb);
                if (lane == 0) bt->next = -1;
                if (tid == 0 && recBatch >= 0) pool.batches[recBatch].next = (int32_t)idx;
                if (firstBatch < 0) firstBatch = (int)idx;
                recBatch = (int)idx;
            } else {
                recording = false;   // pool exhausted: the emit pass decodes this block speculatively
                firstBatch = -2;
            }
        }
        if (replay) recBatch = pool.batches[recBatch].next;
        nTok += totN;
        nU += totU;
        nRef += totR;
        litlenBits += totB;
        if (f < NL) { endBit = baseBits + L.xExit[f]; break; }
        if (replay && recBatch < 0) return;   // (a chain that ends before the block does: cannot happen for a recorded block)
        s0 = baseBits + L.xExit[NL - 1];
        G += (long long)NL * C;
        __syncthreads();   // (xExit / wave sums are rewritten by the next batch)
    }
    __syncthreads();
    if (tid == 0) {
        S->litlenBits = (long long)litlenBits;
        S->sizeBits = S->hdrBits + (long long)litlenBits;
        S->valid = 1;
        S->maskSlot = 0;
    }
    __syncthreads();
    po.status = 0;
    po.endBit = endBit;
    po.nTok = (long long)nTok;
    po.uLen = (long long)nU;
    po.sizeBits = S->sizeBits;
    po.needHist = (long long)needHist;
    po.nRef = (int32_t)nRef;
    po.firstBatch = firstBatch == -2 ? -1 : firstBatch;
    if (EMIT) {
        D4GState* g = out.states + em->stateIdx;
        for (int i = tid; i < (int)(sizeof(D4GState) / 4); i += NL) ((uint32_t*)g)[i] = ((uint32_t*)S)[i];
    }
}

// With `hits` set, only the candidates that parse are reported, compacted (the scan's candidates are mostly
// false positives; the host reads back a few hundred records instead of all of them).
struct D4GProbeHit { D4GProbeIn in; D4GProbeOut out; };
__global__ void __launch_bounds__(D4G_PARSE_MAXTHREADS) k_probe_blocks(const D4GStreamDesc* streams, const D4GProbeIn* in, D4GProbeOut* outp, unsigned n,
                                                     D4GProbeHit* hits, unsigned* nHits, D4GChunkPool pool) {
    if (blockIdx.x >= n) return;
    const D4GProbeIn pi = in[blockIdx.x];
    D4GProbeOut po;
    D4GParseOut none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    d4g_parse_block<false>(streams[pi.stream], pi.bitPos, pi.strict, po, nullptr, none, pool);
    if (threadIdx.x == 0) {
        if (hits) {
            if (po.status == 0) {
                unsigned k = atomicAdd(nHits, 1u);
                hits[k].in = pi;
                hits[k].out = po;
            }
        } else {
            outp[blockIdx.x] = po;
        }
    }
}

// Emit: one wave per block decodes it again, now writing tokens, back-reference records, the block's
// initial state (stored blocks: their bytes).
__global__ void __launch_bounds__(D4G_PARSE_MAXTHREADS) k_emit_blocks(const D4GStreamDesc* streams, const D4GEmitIn* in, D4GParseOut out, int32_t* errors,
                                                    D4GChunkPool pool) {
    const D4GEmitIn em = in[blockIdx.x];
    D4GProbeOut po;
    d4g_parse_block<true>(streams[em.stream], em.bitPos, 0, po, &em, out, pool);
    if (threadIdx.x == 0 && (po.status != 0 || po.uLen != em.uLen)) atomicAdd(errors, 1);
}

// ---------------------------------------------------------------------------------------
// 4. Decoded bytes by pointer jumping.  src[q] (u32, stream-relative) is the position byte q
// copies from; literals and stored bytes point at themselves.
// ---------------------------------------------------------------------------------------
struct D4GTokRange { int32_t stream; int32_t stored; long long tokStart, tokCount, uStart, uLen; };
// An entry whose top bit is set is final: the position in its low 31 bits holds the byte itself (a literal, a stored byte).
// Final entries are never looked up again — by the last rounds that is most of them.  (A stream decodes to less than 2 GiB.)
#define D4G_SRC_FINAL 0x80000000u

__global__ void __launch_bounds__(256) k_fill_src(const D4GStreamDesc* streams, const D4GTokRange* ranges, const uint2* tok,
                                                  uint8_t* U, uint32_t* src, int32_t* badDist, int G) {
    const D4GTokRange r = ranges[blockIdx.x / G];
    const D4GStreamDesc sd = streams[r.stream];
    uint32_t* s = src + sd.uBase;
    uint8_t* u = U + sd.uBase;
    long long stride = (long long)G * blockDim.x;
    long long t0 = (long long)(blockIdx.x % G) * blockDim.x + threadIdx.x;
    if (r.stored) {
        for (long long k = t0; k < r.uLen; k += stride) s[r.uStart + k] = (uint32_t)(r.uStart + k) | D4G_SRC_FINAL;
        return;
    }
    for (long long t = t0; t < r.tokCount; t += stride) {
        uint2 tk = tok[r.tokStart + t];
        uint32_t a = tk.x, pos = tk.y;
        int dist = tok_dist(a), val = tok_val(a);
        if (dist == 0) {
            if (val < 256) { u[pos] = (uint8_t)val; s[pos] = pos | D4G_SRC_FINAL; }
        } else if ((uint32_t)dist > pos) {
            badDist[r.stream] = 1;  // reference: readSlice walks off the first block (NullPointerException)
            for (int k = 0; k < val; k++) s[pos + k] = (pos + k) | D4G_SRC_FINAL;
        } else {
            for (int k = 0; k < val; k++) s[pos + k] = pos + k - dist;
        }
    }
}

// One stream per blockIdx.y.  In place: reading a concurrently updated entry still yields an ancestor.
// `prev` (the previous round's moved-entries counter, nullptr for the first round): the rounds of a batch are launched
// back to back and a round whose predecessor moved fewer than stopNum entries does nothing — the host reads the counters
// once per batch instead of once per round.
__global__ void __launch_bounds__(256) k_jump_streams(const D4GStreamDesc* streams, uint32_t* src, unsigned long long* changed, int G,
                                                      const unsigned long long* prev, unsigned long long stopNum) {
    if (prev && *prev < stopNum) return;
    const D4GStreamDesc sd = streams[blockIdx.x / G];
    uint32_t* s = src + sd.uBase;   // uBase is a multiple of 16: four entries per 16-byte load
    const long long n4 = sd.uLen >> 2;
    long long stride = (long long)G * blockDim.x;
    int any = 0;   // entries this thread moved
    // four consecutive entries per thread and step: one coalesced 16-byte load, four gathers in flight
    for (long long q4 = (long long)(blockIdx.x % G) * blockDim.x + threadIdx.x; q4 < n4; q4 += stride) {
        const uint32_t q = (uint32_t)(q4 << 2);
        uint4 a = *(const uint4*)(s + q);
        uint32_t b0 = (a.x & D4G_SRC_FINAL) ? a.x : s[a.x], b1 = (a.y & D4G_SRC_FINAL) ? a.y : s[a.y];
        uint32_t b2 = (a.z & D4G_SRC_FINAL) ? a.z : s[a.z], b3 = (a.w & D4G_SRC_FINAL) ? a.w : s[a.w];
        bool c0 = b0 != a.x, c1 = b1 != a.y, c2 = b2 != a.z, c3 = b3 != a.w;
        if (c0 | c1 | c2 | c3) {
            *(uint4*)(s + q) = make_uint4(b0, b1, b2, b3);
            any += (int)c0 + (int)c1 + (int)c2 + (int)c3;
        }
    }
    if (blockIdx.x % G == 0 && threadIdx.x < (sd.uLen & 3)) {   // the last one to three entries
        long long q = (n4 << 2) + threadIdx.x;
        uint32_t a = s[q];
        uint32_t b = (a & D4G_SRC_FINAL) ? a : s[a];
        if (b != a) { s[q] = b; any++; }
    }
    // how many entries moved this round (the host stops doubling once few do; k_resolve_streams walks the rest);
    // one global atomic per workgroup
    __shared__ unsigned wgMoved;
    if (threadIdx.x == 0) wgMoved = 0;
    __syncthreads();
    int tot = wave_sum_i32(any);
    if (tot && (threadIdx.x & 63) == 0) atomicAdd(&wgMoved, (unsigned)tot);
    __syncthreads();
    if (threadIdx.x == 0 && wgMoved) atomicAdd((unsigned long long*)changed, (unsigned long long)wgMoved);
}
// The first doubling rounds, tile by tile.  A round's gathers reach 32 KiB * 2^round back: for the first few rounds that is the
// tile itself and a handful of tiles before it.  One workgroup owns one 32 Ki-entry tile and runs `reps` rounds over it in a
// row while its neighbours (same XCD: consecutive tiles get workgroup ids that are equal mod 8) do the same to theirs, so the
// entries and their targets are served by that XCD's L2 after the first touch instead of crossing to HBM every round — as far as
// the tiles in flight fit: the launch is 1024 threads per tile and asks for 70 KiB of (unused) LDS, two tiles per CU at most.  In
// place and unsynchronised like k_jump_streams: any value read is a position further up the same copy chain.
#define D4G_JUMP_TILE 32768
struct D4GJumpTile { int32_t stream, pad; long long first; };
__global__ void __launch_bounds__(1024) k_jump_tiles(const D4GStreamDesc* streams, const D4GJumpTile* tiles, uint32_t* src, int reps,
                                                    unsigned long long* changed) {
    // (the launch asks for LDS it does not use — d4g_host.h: that bounds the workgroups per CU, so that the tiles in flight stay in L2)
    const D4GJumpTile t = tiles[blockIdx.x];
    const D4GStreamDesc sd = streams[t.stream];
    uint32_t* s = src + sd.uBase;
    const long long q0 = t.first, q1 = q0 + D4G_JUMP_TILE < sd.uLen ? q0 + D4G_JUMP_TILE : sd.uLen;
    const long long n4 = (q1 - q0) >> 2;
    int any = 0;
    for (int rep = 0; rep < reps; rep++) {
        any = 0;
        for (long long k = threadIdx.x; k < n4; k += blockDim.x) {
            const long long q = q0 + (k << 2);
            uint4 a = *(const uint4*)(s + q);
            uint32_t b0 = (a.x & D4G_SRC_FINAL) ? a.x : s[a.x], b1 = (a.y & D4G_SRC_FINAL) ? a.y : s[a.y];
            uint32_t b2 = (a.z & D4G_SRC_FINAL) ? a.z : s[a.z], b3 = (a.w & D4G_SRC_FINAL) ? a.w : s[a.w];
            bool c0 = b0 != a.x, c1 = b1 != a.y, c2 = b2 != a.z, c3 = b3 != a.w;
            if (c0 | c1 | c2 | c3) {
                *(uint4*)(s + q) = make_uint4(b0, b1, b2, b3);
                any += (int)c0 + (int)c1 + (int)c2 + (int)c3;
            }
        }
        if ((long long)threadIdx.x < ((q1 - q0) & 3)) {   // the last one to three entries of the stream
            const long long q = q0 + (n4 << 2) + threadIdx.x;
            uint32_t a = s[q];
            uint32_t b = (a & D4G_SRC_FINAL) ? a : s[a];
            if (b != a) { s[q] = b; any++; }
        }
    }
    __shared__ unsigned wgMoved;
    if (threadIdx.x == 0) wgMoved = 0;
    __syncthreads();
    int tot = wave_sum_i32(any);
    if (tot && (threadIdx.x & 63) == 0) atomicAdd(&wgMoved, (unsigned)tot);
    __syncthreads();
    if (threadIdx.x == 0 && wgMoved) atomicAdd((unsigned long long*)changed, (unsigned long long)wgMoved);
}
__global__ void __launch_bounds__(256) k_resolve_streams(const D4GStreamDesc* streams, const uint32_t* src, uint8_t* U, int G) {
    const D4GStreamDesc sd = streams[blockIdx.x / G];
    const uint32_t* s = src + sd.uBase;
    uint8_t* u = U + sd.uBase;
    long long stride = (long long)G * blockDim.x;
    for (long long q = (long long)(blockIdx.x % G) * blockDim.x + threadIdx.x; q < sd.uLen; q += stride) {
        uint32_t a = s[q];
        if (a != ((uint32_t)q | D4G_SRC_FINAL)) {
            // (pointer doubling stops early: follow what is left of the chain — every hop lands further up it)
            while (!(a & D4G_SRC_FINAL)) a = s[a];
            u[q] = u[a & ~D4G_SRC_FINAL];
        }
    }
}
