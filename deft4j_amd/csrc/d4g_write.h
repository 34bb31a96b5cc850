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

struct D4GWriteLds {
    D4GState st;
    uint16_t litCode[D4G_NLIT];
    uint16_t distCode[D4G_NDIST];
    uint16_t clCode[20];
    long long scan[8];
    long long base;
};

__global__ void __launch_bounds__(256) k_write(D4GCtx c, const D4GWriteJob* jobs, uint32_t* out) {
    __shared__ D4GWriteLds W;
    const D4GWriteJob job = jobs[blockIdx.x];
    if (job.type == D4G_STORED) {
        long long dataByte = ((job.bitStart + 3 + 7) >> 3);  // flushToByteAligned
        if (threadIdx.x == 0) {
            put_bits(out, job.bitStart, (uint64_t)(job.isFinal ? 1 : 0), 3);
            unsigned len = (unsigned)job.uLen;
            put_bits(out, dataByte * 8, (uint64_t)(len & 0xffff) | ((uint64_t)((~len) & 0xffff) << 16), 32);
        }
        const uint8_t* src = c.U + job.uAbs;
        for (long long i = threadIdx.x; i < job.uLen; i += blockDim.x) {
            long long ob = dataByte + 4 + i;
            atomicOr(&out[ob >> 2], (uint32_t)src[i] << (8 * (ob & 3)));
        }
        return;
    }
    const D4GBlock b = c.blocks[job.blk];
    D4GState* S = &W.st;
    __syncthreads();
    wg_copy_words((uint32_t*)S, (const uint32_t*)state_ptr(c, job.blk, 0), (int)(sizeof(D4GState) / 4));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S->type == D4G_FIXED) { S->litLen[286] = 8; S->litLen[287] = 8; }  // RFC 1951 fixed code is canonical over 288 symbols
        t0_canonical_codes(S->litLen, D4G_NLIT, W.litCode);
        t0_canonical_codes(S->distLen, D4G_NDIST, W.distCode);
        long long pos = job.bitStart;
        put_bits(out, pos, (uint64_t)((S->type << 1) | (job.isFinal ? 1 : 0)), 3);
        pos += 3;
        if (S->type == D4G_DYNAMIC) {
            t0_canonical_codes(S->clLen, 19, W.clCode);
            put_bits(out, pos, (uint64_t)(S->nLit - 257), 5); pos += 5;
            put_bits(out, pos, (uint64_t)(S->nDist - 1), 5); pos += 5;
            put_bits(out, pos, (uint64_t)(S->nCl - 4), 4); pos += 4;
            for (int i = 0; i < S->nCl; i++) { put_bits(out, pos, S->clLen[D4G_CL_ORDER[i]], 3); pos += 3; }
            for (int i = 0; i < S->nPairs; i++) {
                int sym, run, value;
                uint16_t p = S->pairs[i];
                pair_decode(p, sym, run, value);
                if (p & D4G_PAIR_EXPANDED) {
                    for (int k = 0; k < run; k++) { put_bits(out, pos, W.clCode[value], S->clLen[value]); pos += S->clLen[value]; }
                } else {
                    put_bits(out, pos, W.clCode[sym], S->clLen[sym]); pos += S->clLen[sym];
                    if (sym == 16) { put_bits(out, pos, (uint64_t)(run - 3), 2); pos += 2; }
                    else if (sym == 17) { put_bits(out, pos, (uint64_t)(run - 3), 3); pos += 3; }
                    else if (sym == 18) { put_bits(out, pos, (uint64_t)(run - 11), 7); pos += 7; }
                }
            }
            if (pos != job.bitStart + 3 + S->hdrBits) {
#ifdef D4G_HOSTSIM
                fprintf(stderr, "write: header wrote %lld bits, state says %lld\n", pos - job.bitStart - 3, (long long)S->hdrBits);
#endif
                atomicAdd(c.errors, 1);
            }
        }
        W.base = pos;
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
            } else if ((mask[t >> 6] >> (t & 63)) & 1) {
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
        if (kind == 1) {
            int val = tok_val(a);
            put_bits(out, pos, W.litCode[val], S->litLen[val]);
        } else if (kind == 2) {
            int len = tok_val(a), dist = tok_dist(a);
            int ls = d4g_len2sym(len, tok_edge(a)), ds = d4g_dist2sym(dist);
            uint64_t bits = W.litCode[ls];
            int n = S->litLen[ls];
            bits |= (uint64_t)(len - d4g_lsym_base(ls)) << n;
            n += d4g_lsym_ebits(ls);
            bits |= (uint64_t)W.distCode[ds] << n;
            n += S->distLen[ds];
            bits |= (uint64_t)(dist - d4g_dsym_base(ds)) << n;
            n += d4g_dsym_ebits(ds);
            put_bits(out, pos, bits, n);
        } else if (kind == 3) {
            int len = tok_val(a);
            const uint8_t* p = Ub + toff;
            for_bytes(p, len, [&](int by) {
                int l = S->litLen[by];
                put_bits(out, pos, W.litCode[by], l);
                pos += l;
                return true;
            });
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
    // masks: tokens of A then tokens of B
    const uint64_t* mA = mask_ptr(c, bA, sA->maskSlot);
    const uint64_t* mB = mask_ptr(c, bB, sB->maskSlot);
    uint64_t* mM = mask_ptr(c, bM, 0);
    long long q = bA.tokCount >> 6;
    int s = (int)(bA.tokCount & 63);
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
    __syncthreads();
    if (threadIdx.x == 0) { S->maskSlot = 0; S->valid = 1; S->flags = 0; }
    wg_store_state(state_ptr(c, job.blkM, 0), S);
}
