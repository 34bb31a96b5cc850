// d4g_write.h — canonical-Huffman bit packing of the final blocks, and the block-merge kernel.
//
// Writer reproduces DeflateStream.write (B/deflate/DeflateStream.java:128-145),
// DeflateBlockHuffman.writeHuffCode (:1033-1103) / writeBackref (:1110-1130) / writeSym and
// DeflateBlockUncompressed.write (B/deflate/DeflateBlockUncompressed.java:39-56).
// One workgroup per block: per-token bit lengths -> workgroup prefix scan -> every lane ORs
// its code words into the zero-initialised output at its bit offset (LSB-first bit order,
// codes bit-reversed as Util.rev does, B/util/Util.java:244-254).
#pragma once
#include "d4g_ops.h"

// LDS writes of a wave's lanes must be visible to its other lanes before they are read back (the LDS executes one wave's
// operations in order; the barrier keeps the compiler from moving them across)
#ifdef D4G_HOSTSIM
#define D4G_WAVE_LDS_SYNC() ((void)__ballot(1))
#else
#define D4G_WAVE_LDS_SYNC() __builtin_amdgcn_wave_barrier()
#endif

struct D4GWriteJob {
    int32_t blk;        // block index (Huffman: its slot 0 is written; stored: only the data range is used)
    int32_t type;       // final type to write
    int32_t isFinal;    // last block of its stream
    int32_t pad;
    long long bitStart; // absolute bit offset (in the output word array) of the block's 3 prolog bits
    long long uAbs;     // stored: absolute offset of the data in U
    long long uLen;     // stored: byte count
};

D4G_DEV void put_bits(uint32_t* out, long long bitpos, uint64_t bits, int n) {
    if (n <= 0) return;
    long long w = bitpos >> 5;
    int sh = (int)(bitpos & 31);
    atomicOr(&out[w], (uint32_t)(bits << sh));
    if (sh + n > 32) atomicOr(&out[w + 1], (uint32_t)(bits >> (32 - sh)));
    if (sh + n > 64) atomicOr(&out[w + 2], (uint32_t)(bits >> (64 - sh)));
}

// canonical codes by (length, symbol), stored bit-reversed — Huffman.buildCodes + Huffman.getSym
__device__ void t0_canonical_codes(const uint8_t* lens, int n, uint16_t* codes) {
    int count[16];
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[lens[i]]++;
    int nextc[16];
    int next = 0, lastShift = 0;
    for (int l = 1; l <= 15; l++) {
        nextc[l] = 0;
        if (count[l]) {
            next <<= (l - lastShift);
            lastShift = l;
            nextc[l] = next;
            next += count[l];
        }
    }
    for (int i = 0; i < n; i++) {
        int l = lens[i];
        unsigned r = 0;
        if (l) {
            unsigned code = (unsigned)nextc[l]++;
            for (int b = 0; b < l; b++) r |= ((code >> b) & 1u) << (l - 1 - b);
        }
        codes[i] = (uint16_t)r;
    }
}

// the same by the whole workgroup: symbol i's code = first code of its length + how many lower-numbered symbols share the length
__device__ __forceinline__ void wg_canonical_codes(const uint8_t* lens, int n, uint16_t* codes, unsigned* cnt, unsigned* nextc) {
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[lens[i]], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned next = 0;
        int lastShift = 0;
        nextc[0] = 0;
        for (int l = 1; l <= 15; l++) {
            nextc[l] = 0;
            if (cnt[l]) {
                next <<= (l - lastShift);
                lastShift = l;
                nextc[l] = next;
                next += cnt[l];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int l = lens[i];
        unsigned r = 0;
        if (l) {
            unsigned rank = 0;
            for (int j = 0; j < i; j++) rank += lens[j] == l;
            const unsigned code = nextc[l] + rank;
            for (int b = 0; b < l; b++) r |= ((code >> b) & 1u) << (l - 1 - b);
        }
        codes[i] = (uint16_t)r;
    }
    __syncthreads();
}

// Output words a wave assembles in LDS per step of 64 tokens before it touches memory: interior words are written with one
// plain store each, only the two words it may share with its neighbours go through atomicOr.  64 tokens x 48 bits fit;
// a step that holds more (several long back-references expanded to literals) falls back to per-token atomics.
#define D4G_STAGE_WORDS 128
D4G_DEV void stage_bits(uint32_t* st, int relBit, uint64_t bits, int n) {   // relBit counted from the first staged word's bit 0
    if (n <= 0) return;
    int w = relBit >> 5, sh = relBit & 31;
    atomicOr(&st[w], (uint32_t)(bits << sh));
    if (sh + n > 32) atomicOr(&st[w + 1], (uint32_t)(bits >> (32 - sh)));
    if (sh + n > 64) atomicOr(&st[w + 2], (uint32_t)(bits >> (64 - sh)));
}

struct D4GWriteLds {
    D4GState st;
    uint16_t litCode[D4G_NLIT];
    uint16_t distCode[D4G_NDIST];
    uint16_t clCode[20];
    long long scan[16];
    long long base;
    unsigned cnt[3][16], next[3][16];   // per alphabet: symbols per code length, first code of each length
    uint32_t stage[16][D4G_STAGE_WORDS + 2];   // per wave: the output words of its current 64 tokens
};

__global__ void __launch_bounds__(1024) k_write(D4GCtx c, const D4GWriteJob* jobs, uint32_t* out) {
    __shared__ D4GWriteLds W;
    const D4GWriteJob job = jobs[blockIdx.x];
    if (job.type == D4G_STORED) {
        long long dataByte = ((job.bitStart + 3 + 7) >> 3);  // flushToByteAligned
        if (threadIdx.x == 0) {
            put_bits(out, job.bitStart, (uint64_t)(job.isFinal ? 1 : 0), 3);
            unsigned len = (unsigned)job.uLen;
            put_bits(out, dataByte * 8, (uint64_t)(len & 0xffff) | ((uint64_t)((~len) & 0xffff) << 16), 32);
        }
        // the payload: whole output words get one plain store, the (up to two) partial ones an atomicOr
        const uint8_t* src = c.U + job.uAbs;
        const long long o0 = dataByte + 4, o1 = o0 + job.uLen;
        for (long long w = (o0 >> 2) + threadIdx.x; w <= ((o1 - 1) >> 2) && job.uLen > 0; w += blockDim.x) {
            long long lo = w * 4 < o0 ? o0 : w * 4, hi = w * 4 + 4 > o1 ? o1 : w * 4 + 4;
            uint32_t v = 0;
            for (long long q = lo; q < hi; q++) v |= (uint32_t)src[q - o0] << (8 * (q & 3));
            if (hi - lo == 4) out[w] = v;
            else atomicOr(&out[w], v);
        }
        return;
    }
    const D4GBlock b = c.blocks[job.blk];
    D4GState* S = &W.st;
    __syncthreads();
    wg_copy_words((uint32_t*)S, (const uint32_t*)state_ptr(c, job.blk, 0), (int)(sizeof(D4GState) / 4));
    __syncthreads();
    // ---- codes and header, by the whole workgroup ----
    if (threadIdx.x == 0 && S->type == D4G_FIXED) { S->litLen[286] = 8; S->litLen[287] = 8; }  // RFC 1951 fixed code is canonical over 288 symbols
    __syncthreads();
    wg_canonical_codes(S->litLen, D4G_NLIT, W.litCode, W.cnt[0], W.next[0]);
    wg_canonical_codes(S->distLen, D4G_NDIST, W.distCode, W.cnt[1], W.next[1]);
    if (S->type == D4G_DYNAMIC) wg_canonical_codes(S->clLen, 19, W.clCode, W.cnt[2], W.next[2]);
    {
        // header fields: prolog, then (dynamic) HLIT / HDIST / HCLEN, the code-length code lengths and the RLE pairs; every
        // thread owns one field, a workgroup scan gives its bit offset
        const int nCl = S->type == D4G_DYNAMIC ? S->nCl : 0, nPairs = S->type == D4G_DYNAMIC ? S->nPairs : 0;
        const int nFields = 1 + (S->type == D4G_DYNAMIC ? 3 : 0) + nCl + nPairs;
        long long hdrTotal = 0;
        for (int f0 = 0; f0 < nFields; f0 += blockDim.x) {
            const int f = f0 + threadIdx.x;
            uint64_t bits = 0;
            int n = 0, expRun = 0, expVal = 0;
            if (f < nFields) {
                if (f == 0) { bits = (uint64_t)((S->type << 1) | (job.isFinal ? 1 : 0)); n = 3; }
                else if (S->type == D4G_DYNAMIC) {
                    if (f == 1) { bits = (uint64_t)(S->nLit - 257); n = 5; }
                    else if (f == 2) { bits = (uint64_t)(S->nDist - 1); n = 5; }
                    else if (f == 3) { bits = (uint64_t)(S->nCl - 4); n = 4; }
                    else if (f < 4 + nCl) { bits = S->clLen[D4G_CL_ORDER[f - 4]]; n = 3; }
                    else {
                        int sym, run, value;
                        const uint16_t pr = S->pairs[f - 4 - nCl];
                        pair_decode(pr, sym, run, value);
                        if (pr & D4G_PAIR_EXPANDED) { expRun = run; expVal = value; n = run * S->clLen[value]; }
                        else {
                            bits = W.clCode[sym];
                            n = S->clLen[sym];
                            if (sym == 16) { bits |= (uint64_t)(run - 3) << n; n += 2; }
                            else if (sym == 17) { bits |= (uint64_t)(run - 3) << n; n += 3; }
                            else if (sym == 18) { bits |= (uint64_t)(run - 11) << n; n += 7; }
                        }
                    }
                }
            }
            long long incl = n;
            for (int d = 1; d < 64; d <<= 1) {
                long long o = __shfl_up(incl, d);
                if ((int)(threadIdx.x & 63) >= d) incl += o;
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 63) W.scan[threadIdx.x >> 6] = incl;
            __syncthreads();
            long long waveBase = 0, total = 0;
            for (int i = 0; i < (int)(blockDim.x >> 6); i++) { if (i < (int)(threadIdx.x >> 6)) waveBase += W.scan[i]; total += W.scan[i]; }
            long long pos = job.bitStart + hdrTotal + waveBase + incl - n;
            if (expRun) {
                for (int k = 0; k < expRun; k++) { put_bits(out, pos, W.clCode[expVal], S->clLen[expVal]); pos += S->clLen[expVal]; }
            } else put_bits(out, pos, bits, n);
            hdrTotal += total;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (S->type == D4G_DYNAMIC && hdrTotal != 3 + S->hdrBits) {
#ifdef D4G_HOSTSIM
                fprintf(stderr, "write: header wrote %lld bits, state says %lld\n", hdrTotal - 3, (long long)S->hdrBits);
#endif
                atomicAdd(c.errors, 1);
            }
            W.base = job.bitStart + hdrTotal;
        }
    }
    __syncthreads();
    const uint64_t* mask = mask_ptr(c, b, S->maskSlot);
    const uint8_t* Ub = c.U + b.uBase;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (long long t0 = 0; t0 < b.tokCount; t0 += blockDim.x) {
        long long t = t0 + threadIdx.x;
        long long nbits = 0;
        uint32_t a = 0, toff = 0;
        int kind = 0;  // 0 none, 1 single symbol, 2 back-reference, 3 expanded back-reference
        if (t < b.tokCount) {
            uint2 tkw = c.tok[b.tokStart + t];
            a = tkw.x;
            toff = tkw.y;
            int dist = tok_dist(a), val = tok_val(a);
            if (dist == 0) {
                if (val == 256 && t != b.tokCount - 1) kind = 0;  // EOB of a merged-away block
                else { kind = 1; nbits = S->litLen[val]; }
            } else if (d4g_ref_expanded(mask, c.tokRef[b.tokStart + t], b.refStart)) {
                kind = 3;
                const uint8_t* p = Ub + toff;
                for_bytes(p, val, [&](int by) { nbits += S->litLen[by]; return true; });
            } else {
                kind = 2;
                int ls, ds;
                nbits = backref_cost(S, val, tok_edge(a), dist, ls, ds);
            }
        }
        // workgroup exclusive scan of nbits
        long long incl = nbits;
        for (int d = 1; d < 64; d <<= 1) {
            long long o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        __syncthreads();
        if (lane == 63) W.scan[wave] = incl;
        __syncthreads();
        long long waveBase = 0, total = 0;
        for (int i = 0; i < nw; i++) { if (i < wave) waveBase += W.scan[i]; total += W.scan[i]; }
        long long pos = W.base + waveBase + incl - nbits;
        // this wave's 64 tokens cover bits [wStart, wStart + wBits): assemble them in LDS, then store whole words
        const long long wStart = W.base + waveBase;
        const long long wBits = __shfl(incl, 63);
        const long long word0 = wStart >> 5;
        const int nWords = wBits > 0 ? (int)(((wStart + wBits - 1) >> 5) - word0 + 1) : 0;
        const bool staged = nWords <= D4G_STAGE_WORDS;
        uint32_t* stg = W.stage[wave];
        if (staged) for (int k = lane; k < nWords + 2; k += 64) stg[k] = 0;
        D4G_WAVE_LDS_SYNC();
        const int rel = (int)(pos - (word0 << 5));
        uint64_t bits = 0;
        int n = 0;
        if (kind == 1) {
            int val = tok_val(a);
            bits = W.litCode[val];
            n = S->litLen[val];
        } else if (kind == 2) {
            int len = tok_val(a), dist = tok_dist(a);
            int ls = d4g_len2sym(len, tok_edge(a)), ds = d4g_dist2sym(dist);
            bits = W.litCode[ls];
            n = S->litLen[ls];
            bits |= (uint64_t)(len - d4g_lsym_base(ls)) << n;
            n += d4g_lsym_ebits(ls);
            bits |= (uint64_t)W.distCode[ds] << n;
            n += S->distLen[ds];
            bits |= (uint64_t)(dist - d4g_dsym_base(ds)) << n;
            n += d4g_dsym_ebits(ds);
        }
        if (kind == 1 || kind == 2) {
            if (staged) stage_bits(stg, rel, bits, n);
            else put_bits(out, pos, bits, n);
        } else if (kind == 3) {
            int len = tok_val(a);
            const uint8_t* p = Ub + toff;
            long long pp = pos;
            for_bytes(p, len, [&](int by) {
                int l = S->litLen[by];
                if (staged) stage_bits(stg, (int)(pp - (word0 << 5)), W.litCode[by], l);
                else put_bits(out, pp, W.litCode[by], l);
                pp += l;
                return true;
            });
        }
        D4G_WAVE_LDS_SYNC();
        if (staged) {
            const bool headShared = (wStart & 31) != 0, tailShared = ((wStart + wBits) & 31) != 0;
            for (int k = lane; k < nWords; k += 64) {
                uint32_t v = stg[k];
                if ((k == 0 && headShared) || (k == nWords - 1 && tailShared)) { if (v) atomicOr(&out[word0 + k], v); }
                else out[word0 + k] = v;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) W.base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && W.base != job.bitStart + 3 + S->sizeBits) {
#ifdef D4G_HOSTSIM
        fprintf(stderr, "write: block wrote %lld bits, state says %lld (type %d)\n", W.base - job.bitStart - 3, (long long)S->sizeBits, S->type);
#endif
        atomicAdd(c.errors, 1);
    }
}

// ---------------------------------------------------------------------------------------
// DeflateBlockHuffman.merge — DeflateBlockHuffman.java:1233-1271: both blocks recoded to the
// fixed code, token lists concatenated (first EOB dropped).  Builds slot 0 / mask 0 of `blkM`
// from slot 0 / current masks of blkA and blkB.  blkM's descriptor is already on the device.
// ---------------------------------------------------------------------------------------
struct D4GMergeJob { int32_t blkA, blkB, blkM, pad; };

__global__ void __launch_bounds__(256) k_make_merged(D4GCtx c, const D4GMergeJob* jobs) {
    __shared__ D4GLds L;
    const D4GMergeJob job = jobs[blockIdx.x];
    const D4GBlock bA = c.blocks[job.blkA], bB = c.blocks[job.blkB], bM = c.blocks[job.blkM];
    D4GState* S = &L.st;
    const D4GState* sA = state_ptr(c, job.blkA, 0);
    const D4GState* sB = state_ptr(c, job.blkB, 0);
    wg_load_state(S, sA);
    for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) S->hist[i] += sB->hist[i];
    __syncthreads();
    if (threadIdx.x == 0) S->hist[256] -= 1;  // the first block's EOB is removed
    __syncthreads();
    int wasType = S->type;
    __syncthreads();
    if (wasType == D4G_FIXED && threadIdx.x == 0) S->type = D4G_DYNAMIC;  // force the full recompute below
    __syncthreads();
    wg_recode_to_fixed(&L);
    // masks: back-references of A then those of B
    const uint64_t* mA = mask_ptr(c, bA, sA->maskSlot);
    const uint64_t* mB = mask_ptr(c, bB, sB->maskSlot);
    uint64_t* mM = mask_ptr(c, bM, 0);
    long long q = bA.refCount >> 6;
    int s = (int)(bA.refCount & 63);
    for (long long w = threadIdx.x; w < bM.maskWords; w += blockDim.x) {
        uint64_t v;
        if (w < q) v = mA[w];
        else {
            long long j = w - q;
            uint64_t lo = j < bB.maskWords ? mB[j] : 0;
            uint64_t prev = (j >= 1 && j - 1 < bB.maskWords) ? mB[j - 1] : 0;
            v = s ? ((lo << s) | (prev >> (64 - s))) : lo;
            if (j == 0 && s) v = (mA[q] & ((1ULL << s) - 1)) | (lo << s);
        }
        mM[w] = v;
    }
    // the arena block is a new token range: forget the token passes memoised for its previous contents
    if (c.passMemo && bM.passMemo >= 0)
        for (int k = threadIdx.x; k < D4G_PASSMEMO_SLOTS; k += blockDim.x) {
            D4GPassMemo* e = (D4GPassMemo*)(c.passMemo + bM.passMemo + (long long)k * bM.passMemoStride);
            e->tag = 0;
            e->state = 0;
        }
    // static bin statistics and bin masks of the merged block: rows add, masks concatenate like the token masks
    if (bM.binStat >= 0) {
        const uint32_t* gA = c.binStat + bA.binStat;
        const uint32_t* gB = c.binStat + bB.binStat;
        uint32_t* gM = c.binStat + bM.binStat;
        for (int i = threadIdx.x; i < D4G_NBINS * D4G_BINSTRIDE; i += blockDim.x) gM[i] = gA[i] + gB[i];
        for (int bin = 0; bin < D4G_NBINS; bin++) {
            const uint64_t* xA = c.binMask + bA.binMask + (long long)bin * bA.maskWords;
            const uint64_t* xB = c.binMask + bB.binMask + (long long)bin * bB.maskWords;
            uint64_t* xM = c.binMask + bM.binMask + (long long)bin * bM.maskWords;
            for (long long w = threadIdx.x; w < bM.maskWords; w += blockDim.x) {
                uint64_t v;
                if (w < q) v = xA[w];
                else {
                    long long j = w - q;
                    uint64_t lo = j < bB.maskWords ? xB[j] : 0;
                    uint64_t prev = (j >= 1 && j - 1 < bB.maskWords) ? xB[j - 1] : 0;
                    v = s ? ((lo << s) | (prev >> (64 - s))) : lo;
                    if (j == 0 && s) v = (xA[q] & ((1ULL << s) - 1)) | (lo << s);
                }
                xM[w] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { S->maskSlot = 0; S->valid = 1; S->flags = 0; }
    wg_store_state(state_ptr(c, job.blkM, 0), S);
}

// A finished merged block leaves its arena: state slot 0 and the current mask of `blkA` are copied to slot 0 / mask 0
// of `blkM` (whose descriptor the host has already rewritten to the merged token range).
// block descriptors the host changed (merge arenas, committed merged blocks): one upload + one scatter per round instead
// of one small copy per descriptor
__global__ void __launch_bounds__(64) k_patch_blocks(D4GBlock* blocks, const int32_t* idx, const D4GBlock* src, int n) {
    if ((int)blockIdx.x >= n) return;
    const uint32_t* s = (const uint32_t*)(src + blockIdx.x);
    uint32_t* d = (uint32_t*)(blocks + idx[blockIdx.x]);
    for (int i = threadIdx.x; i < (int)(sizeof(D4GBlock) / 4); i += blockDim.x) d[i] = s[i];
}

__global__ void __launch_bounds__(256) k_commit_merged(D4GCtx c, const D4GMergeJob* jobs) {
    const D4GMergeJob job = jobs[blockIdx.x];
    const D4GBlock bA = c.blocks[job.blkA], bM = c.blocks[job.blkM];
    const D4GState* sA = state_ptr(c, job.blkA, 0);
    D4GState* sM = state_ptr(c, job.blkM, 0);
    const uint64_t* mA = mask_ptr(c, bA, sA->maskSlot);
    uint64_t* mM = mask_ptr(c, bM, 0);
    for (long long w = threadIdx.x; w < bM.maskWords; w += blockDim.x) mM[w] = mA[w];
    for (int i = threadIdx.x; i < (int)(sizeof(D4GState) / 4); i += blockDim.x) ((uint32_t*)sM)[i] = ((const uint32_t*)sA)[i];
    __syncthreads();
    if (threadIdx.x == 0) sM->maskSlot = 0;
}

// ---------------------------------------------------------------------------------------
// Trailer checksums over the decoded bytes: CRC-32 + ISIZE for gzip members
// (K/GZFile.java:129-145, java.util.zip.CRC32) and Adler-32 for zlib streams (K/ZLibFile.java:41-51).
// k_csum_tiles: one workgroup per 128 KiB tile; thread t checksums 512 bytes (slice-by-4 CRC tables in
// LDS, 16-byte loads), then the 256 results are concatenated pairwise in order;
// k_csum_combine: one workgroup per stream folds the tile records in order.  Concatenation uses zlib's
// algebra: CRCs by multiplying with x^(8n) mod P (crc32_combine), Adler sums by the closed-form block rule.
// ---------------------------------------------------------------------------------------
#define D4G_CSUM_CHUNK 512                       // bytes per thread
#define D4G_CSUM_TILE (D4G_CSUM_CHUNK * 256)     // bytes per workgroup (128 KiB)
struct D4GCsumRec { uint32_t crc; uint32_t s1; uint32_t s2; uint32_t pad; unsigned long long len; };
struct D4GCsumOut { uint32_t crc32; uint32_t adler32; long long isize; };

D4G_DEV uint32_t crc_multmodp(uint32_t a, uint32_t b) {  // zlib crc32.c multmodp (reflected CRC-32 polynomial)
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
D4G_DEV uint32_t crc_x8nmodp(unsigned long long nbytes, const uint32_t* x2n) {  // x^(8*nbytes) mod P
    uint32_t p = 1u << 31;
    unsigned k = 3;
    while (nbytes) {
        if (nbytes & 1) p = crc_multmodp(x2n[k & 31], p);
        nbytes >>= 1;
        k++;
    }
    return p;
}
// ordered concatenation A ++ B of two checksum records (zlib crc32_combine / adler32_combine algebra)
D4G_DEV void csum_concat(uint32_t& crc, uint32_t& s1, uint32_t& s2, unsigned long long& len, uint32_t crcB, uint32_t s1B, uint32_t s2B,
                         unsigned long long lenB, const uint32_t* x2n) {
    if (lenB == 0) return;
    crc = len ? (crc_multmodp(crc_x8nmodp(lenB, x2n), crc) ^ crcB) : crcB;
    s2 = (uint32_t)((s2 + (lenB % 65521u) * s1 + s2B) % 65521u);
    s1 = (s1 + s1B) % 65521u;
    len += lenB;
}

__global__ void __launch_bounds__(256) k_csum_tiles(const D4GStreamDesc* streams, const long long* tileBase, int nStreams,
                                                    const uint8_t* U, const uint32_t* crcTab, D4GCsumRec* out) {
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t sx2n[32];
    __shared__ uint32_t sCrc[256], sS1[256], sS2[256];
    __shared__ unsigned long long sLen[256];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) T[i >> 8][i & 255] = crcTab[i];
    if (threadIdx.x < 32) sx2n[threadIdx.x] = crcTab[1024 + threadIdx.x];
    __syncthreads();
    long long tile = blockIdx.x;
    int lo = 0, hi = nStreams - 1;  // stream owning this tile (tileBase = exclusive prefix of tile counts)
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (tileBase[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    const D4GStreamDesc sd = streams[lo];
    long long off = (tile - tileBase[lo]) * D4G_CSUM_TILE + (long long)threadIdx.x * D4G_CSUM_CHUNK;
    long long len = sd.uLen - off;
    if (len > D4G_CSUM_CHUNK) len = D4G_CSUM_CHUNK;
    if (len < 0) len = 0;
    const uint8_t* p = U + sd.uBase + off;  // 16-byte aligned: uBase is, offsets are multiples of 512
    uint32_t crc = 0xffffffffu, s1 = 0, s2 = 0;
    long long i = 0;
    for (; i + 16 <= len; i += 16) {
        uint4 v = *(const uint4*)(p + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t x = crc ^ w[j];
            crc = T[3][x & 0xff] ^ T[2][(x >> 8) & 0xff] ^ T[1][(x >> 16) & 0xff] ^ T[0][x >> 24];
            uint32_t b0 = w[j] & 0xff, b1 = (w[j] >> 8) & 0xff, b2 = (w[j] >> 16) & 0xff, b3 = w[j] >> 24;
            uint32_t rem = (uint32_t)(len - i) - 4 * j;  // weight of b0 = bytes from it to the end of the chunk
            s1 += b0 + b1 + b2 + b3;
            s2 += rem * b0 + (rem - 1) * b1 + (rem - 2) * b2 + (rem - 3) * b3;
        }
    }
    for (; i < len; i++) {
        uint32_t b = p[i];
        crc = T[0][(crc ^ b) & 0xff] ^ (crc >> 8);
        s1 += b;
        s2 += (uint32_t)(len - i) * b;
    }
    sCrc[threadIdx.x] = len ? crc ^ 0xffffffffu : 0u;
    sS1[threadIdx.x] = s1 % 65521u;
    sS2[threadIdx.x] = s2 % 65521u;
    sLen[threadIdx.x] = (unsigned long long)len;
    __syncthreads();
    for (int step = 1; step < 256; step <<= 1) {
        int t = threadIdx.x;
        if ((t & (2 * step - 1)) == 0) csum_concat(sCrc[t], sS1[t], sS2[t], sLen[t], sCrc[t + step], sS1[t + step], sS2[t + step], sLen[t + step], sx2n);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        D4GCsumRec r;
        r.crc = sCrc[0]; r.s1 = sS1[0]; r.s2 = sS2[0]; r.pad = 0; r.len = sLen[0];
        out[tile] = r;
    }
}

__global__ void __launch_bounds__(256) k_csum_combine(const D4GStreamDesc* streams, const long long* tileBase, const D4GCsumRec* recs,
                                                      const uint32_t* x2n, D4GCsumOut* outs) {
    __shared__ uint32_t sCrc[256], sS1[256], sS2[256];
    __shared__ unsigned long long sLen[256];
    __shared__ uint32_t sx2n[32];
    const D4GStreamDesc sd = streams[blockIdx.x];
    long long c0 = tileBase[blockIdx.x], c1 = tileBase[blockIdx.x + 1];
    if (threadIdx.x < 32) sx2n[threadIdx.x] = x2n[threadIdx.x];
    __syncthreads();
    long long n = c1 - c0, per = (n + blockDim.x - 1) / blockDim.x;
    long long a = c0 + (long long)threadIdx.x * per, e = a + per;
    if (a > c1) a = c1;
    if (e > c1) e = c1;
    uint32_t crc = 0, s1 = 0, s2 = 0;
    unsigned long long len = 0;
    for (long long c = a; c < e; c++) {
        D4GCsumRec k = recs[c];
        csum_concat(crc, s1, s2, len, k.crc, k.s1, k.s2, k.len, sx2n);
    }
    sCrc[threadIdx.x] = crc; sS1[threadIdx.x] = s1; sS2[threadIdx.x] = s2; sLen[threadIdx.x] = len;
    __syncthreads();
    for (int step = 1; step < 256; step <<= 1) {
        int t = threadIdx.x;
        if ((t & (2 * step - 1)) == 0) csum_concat(sCrc[t], sS1[t], sS2[t], sLen[t], sCrc[t + step], sS1[t + step], sS2[t + step], sLen[t + step], sx2n);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // Adler-32 starts from a = 1, b = 0: a = 1 + S1, b = len*1 + S2 (mod 65521)
        unsigned long long L = sLen[0];
        uint32_t A = (1u + sS1[0]) % 65521u;
        uint32_t B = (uint32_t)((L % 65521u + sS2[0]) % 65521u);
        D4GCsumOut o;
        o.crc32 = L ? sCrc[0] : 0u;
        o.adler32 = (B << 16) | A;
        o.isize = (long long)sd.uLen;
        outs[blockIdx.x] = o;
    }
}
