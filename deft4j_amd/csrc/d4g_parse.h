// d4g_parse.h — inflate-to-tokens kernel (one wave per stream).
//
// Reproduces DeflateStream.parse (B/deflate/DeflateStream.java:72-126),
// DeflateBlockHuffman.initDynamicDecoder (:892-1010) and decodeStream (:778-890): it emits
// the token arrays, the decoded bytes, and one initial state per Huffman block (code
// lengths, header RLE pairs, symbol histogram, bit sizes).  The 32 KiB sliding window lives
// in a 64 KiB LDS ring so back-reference copies never wait on HBM; the compressed input is
// staged through an 8 KiB LDS chunk; lane 0 decodes symbols (LUT + canonical fallback with
// the reference's bit-serial semantics, Huffman.java:170-197), all 64 lanes copy match bytes.
#pragma once
#include "d4g_device.h"

struct D4GStreamIn {
    const uint8_t* data;   // compressed bytes (device, 16-byte aligned, readable 16 KiB past len)
    long long len;
    long long tokBase;     // this stream's region in tokA/tokOff
    long long tokCap;
    long long uBase;       // region in U
    long long uCap;
    long long blkBase;     // region in the parsed-block array
    long long blkCap;
    long long stBase;      // region in the parsed-state array
    long long stCap;
};
struct D4GStreamOut {
    int32_t status;        // 0 ok, -1 parse failure (reference returns false / throws), 1 capacity overflow (retry)
    int32_t nBlocks;
    long long nTok, nU, nStates;  // exact needs (valid also on overflow)
    long long consumedBytes;
    long long sizeBits;    // DeflateStream.getSizeBits — :171-182
};
struct D4GParsedBlock {
    int32_t type;
    int32_t stateIdx;      // index into the stream's parsed-state region (-1 for stored)
    long long tokStart, tokCount, uStart, uLen, sizeBits;
};

#define D4G_LUT_BITS 10
#define D4G_WIN 65536
#define D4G_INCH 8192

struct D4GDecTab {          // canonical decoder of one alphabet
    uint16_t lut[1 << D4G_LUT_BITS];  // sym | len<<9, 0xffff = use the bit-serial path
    uint16_t sorted[D4G_NLIT];        // symbols ordered by (length, index)
    int first[16], count[16], offs[16];
    int useLut;
};

struct D4GParseLds {
    uint8_t win[D4G_WIN];
    uint8_t inbuf[D4G_INCH + 16];
    D4GDecTab lit, dist, cl;
    D4GState st;
};

// Bit reader owned by lane 0 (B/io/BitInputStream.java:59-82: LSB-first).
struct D4GBitReader {
    const uint8_t* inbuf;  // LDS chunk
    long long inBase;      // absolute byte index of inbuf[0]
    long long nbits;       // total bits of the stream
    long long pos;         // bits consumed
    uint64_t buf;
    int cnt;
    long long nextByte;    // absolute index of the next byte to load into buf
    __device__ void reset_to(long long bitpos) {
        pos = bitpos;
        buf = 0;
        cnt = 0;
        nextByte = bitpos >> 3;
        fill();
        int sh = (int)(bitpos & 7);
        buf >>= sh;
        cnt -= sh;
    }
    __device__ void fill() {
        while (cnt <= 56) {
            long long o = nextByte - inBase;
            uint64_t v = (o >= 0 && o < D4G_INCH + 16) ? inbuf[o] : 0;
            buf |= v << cnt;
            cnt += 8;
            nextByte++;
        }
    }
    __device__ bool have(int n) const { return pos + n <= nbits; }
    __device__ void skip(int n) { buf >>= n; cnt -= n; pos += n; }
};

// Huffman.buildCodes (B/huffman/Huffman.java:35-64) + decoder tables.  Lane 0 prepares the
// canonical structure, all lanes fill the LUT.
__device__ void d4g_build_decoder(D4GDecTab* T, const uint8_t* lens, int n) {
    int lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
        for (int l = 0; l < 16; l++) { T->count[l] = 0; T->first[l] = 0; T->offs[l] = 0; }
        for (int i = 0; i < n; i++) if (lens[i] > 0 && lens[i] < 16) T->count[lens[i]]++;
        int next = 0, lastShift = 0, o = 0;
        long long kraft = 0;
        for (int l = 1; l <= 15; l++) {
            T->offs[l] = o;
            o += T->count[l];
            if (T->count[l]) {
                next <<= (l - lastShift);
                lastShift = l;
                T->first[l] = next;
                next += T->count[l];
                kraft += (long long)T->count[l] << (15 - l);
            }
        }
        T->useLut = kraft <= (1 << 15);
        int fill[16];
        for (int l = 0; l < 16; l++) fill[l] = T->offs[l];
        for (int i = 0; i < n; i++) if (lens[i] > 0 && lens[i] < 16) T->sorted[fill[lens[i]]++] = (uint16_t)i;
    }
    __syncthreads();
    for (int i = lane; i < (1 << D4G_LUT_BITS); i += 64) T->lut[i] = 0xffff;
    __syncthreads();
    if (T->useLut) {
        for (int l = 1; l <= D4G_LUT_BITS; l++) {
            int cnt = T->count[l];
            for (int k = lane; k < cnt; k += 64) {
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

// Decode one symbol from `bits` (lane 0), `avail` = bits left in the stream.  -1 on failure.
__device__ int d4g_decode_sym(const D4GDecTab* T, uint64_t bits, long long avail, int* len) {
    unsigned e = T->lut[bits & ((1u << D4G_LUT_BITS) - 1)];
    if (e != 0xffff) {
        int l = e >> 9;
        if (l > avail) return -1;
        *len = l;
        return e & 511;
    }
    int code = 0;
    for (int l = 1; l <= 15; l++) {  // Huffman.readSymbol, bit-serial
        if (l > avail) return -1;
        code = (code << 1) | (int)((bits >> (l - 1)) & 1);
        if (T->count[l] && code >= T->first[l] && code < T->first[l] + T->count[l]) {
            *len = l;
            return T->sorted[T->offs[l] + code - T->first[l]];
        }
    }
    return -1;
}

__global__ void __launch_bounds__(64) k_parse(const D4GStreamIn* ins, D4GStreamOut* outs, uint32_t* tokA, uint32_t* tokOff, uint8_t* U,
                                              D4GParsedBlock* pblocks, D4GState* pstates) {
    __shared__ D4GParseLds L;
    const D4GStreamIn in = ins[blockIdx.x];
    int lane = threadIdx.x & 63;
    D4GBitReader br;
    br.inbuf = L.inbuf;
    br.inBase = 0;
    br.nbits = in.len * 8;
    br.pos = 0; br.buf = 0; br.cnt = 0; br.nextByte = 0;
    long long nTok = 0, nU = 0, nStates = 0, streamBits = 0;
    int nBlocks = 0;
    int status = 0;
    D4GState* S = &L.st;
    bool fin = false;
    // stage the first input chunk
    auto stage = [&](long long base) {
        __syncthreads();
        for (int i = lane * 16; i < D4G_INCH + 16; i += 64 * 16) {
            const uint4* g = (const uint4*)(in.data + base + i);
            *(uint4*)(L.inbuf + i) = *g;
        }
        __syncthreads();
    };
    stage(0);
    if (lane == 0) br.reset_to(0);
    // refill when lane 0's reader is within 1 KiB of the chunk end (decided wave-uniformly)
    auto maybe_refill = [&]() {
        long long nb = __shfl((long long)br.nextByte, 0);
        long long base = __shfl((long long)br.inBase, 0);
        if (nb + 1024 > base + D4G_INCH && base + D4G_INCH < in.len + 16) {
            long long bitpos = __shfl((long long)br.pos, 0);
            long long nbase = (bitpos >> 3) & ~15LL;
            stage(nbase);
            br.inBase = nbase;
            if (lane == 0) br.reset_to(bitpos);
        }
    };
    while (!fin && status == 0) {
        maybe_refill();
        // ---- block prolog ----
        long long pk = 0;
        if (lane == 0) {
            if (!br.have(3)) pk = -1;
            else { br.fill(); pk = (long long)(br.buf & 7); br.skip(3); }
        }
        pk = __shfl(pk, 0);
        if (pk < 0) { status = -1; break; }
        fin = (pk & 1) != 0;
        int btype = (int)(pk >> 1);
        if (btype == 3) { status = -1; break; }
        long long blkTok0 = nTok, blkU0 = nU;
        long long blkSize = 0;
        int stateIdx = -1;
        if (btype == 0) {
            // DeflateBlockUncompressed.parse — B/deflate/DeflateBlockUncompressed.java:23-36
            long long r = 0;
            if (lane == 0) {
                long long p = (br.pos + 7) & ~7LL;
                if (p + 32 > br.nbits) r = -1;
                else {
                    br.reset_to(p);
                    int len = (int)(br.buf & 0xffff), nlen = (int)((br.buf >> 16) & 0xffff);
                    if (nlen != ((~len) & 0xffff)) r = -1;
                    else r = len;
                    br.skip(32);
                }
            }
            r = __shfl(r, 0);
            if (r < 0) { status = -1; break; }
            int len = (int)r;
            long long bytePos = __shfl((long long)br.pos, 0) >> 3;
            for (int k = lane; k < len; k += 64) {
                // bytes past the end of input read as (byte)-1 in the reference (BitInputStreamUtil.readFromBIS)
                uint8_t v = (bytePos + k < in.len) ? in.data[bytePos + k] : 0xff;
                L.win[(nU + k) & (D4G_WIN - 1)] = v;
                if (nU + k < in.uCap) U[in.uBase + nU + k] = v;
            }
            long long np = (bytePos + len) * 8;
            if (np > br.nbits) np = br.nbits + 8;  // EOF was hit: every later read fails
            // re-stage the input at the new position
            {
                long long nbase = np > br.nbits ? (br.nbits >> 3) & ~15LL : (np >> 3) & ~15LL;
                stage(nbase);
                br.inBase = nbase;
                if (lane == 0) { br.reset_to(np > br.nbits ? br.nbits : np); br.pos = np; }
            }
            nU += len;
            blkSize = 0;  // stored size depends on bit position (host computes it)
        } else {
            for (int i = lane; i < (int)(sizeof(D4GState) / 4); i += 64) ((uint32_t*)S)[i] = 0;
            __syncthreads();
            if (btype == 1) {
                for (int i = lane; i < D4G_NLIT; i += 64) S->litLen[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 286 ? 8 : 0;
                for (int i = lane; i < D4G_NDIST; i += 64) S->distLen[i] = i < 30 ? 5 : 0;
                if (lane == 0) S->type = D4G_FIXED;
                __syncthreads();
            } else {
                // initDynamicDecoder — DeflateBlockHuffman.java:892-1010
                long long r = 0;
                if (lane == 0) {
                    if (!br.have(14)) r = -1;
                    else {
                        br.fill();
                        S->nLit = (int)(br.buf & 31) + 257;
                        S->nDist = (int)((br.buf >> 5) & 31) + 1;
                        S->nCl = (int)((br.buf >> 10) & 15) + 4;
                        br.skip(14);
                        if (S->nLit > 288) r = -1;
                        if (r == 0 && !br.have(3 * S->nCl)) r = -1;
                        if (r == 0) {
                            for (int i = 0; i < S->nCl; i++) {
                                br.fill();
                                S->clLen[D4G_CL_ORDER[i]] = (uint8_t)(br.buf & 7);
                                br.skip(3);
                            }
                        }
                        S->type = D4G_DYNAMIC;
                        S->hdrBits = 5 + 5 + 4 + 3 * S->nCl;
                    }
                }
                r = __shfl(r, 0);
                if (r < 0) { status = -1; break; }
                d4g_build_decoder(&L.cl, S->clLen, 19);
                if (lane == 0) {
                    int i = 0, np = 0;
                    int combined = S->nLit + S->nDist;
                    while (i < combined && r == 0) {
                        br.fill();
                        int cl = 0;
                        int sym = d4g_decode_sym(&L.cl, br.buf, br.nbits - br.pos, &cl);
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
                r = __shfl(r, 0);
                if (r < 0) { status = -1; break; }
            }
            // The fixed code (HuffmanTable.LIT, B/huffman/HuffmanTable.java:166-209) is the RFC 1951 code over
            // 288 symbols; 286/287 take code space but are not decodable symbols (decodeStream rejects > 285).
            if (btype == 1 && lane == 0) { S->litLen[286] = 8; S->litLen[287] = 8; }
            d4g_build_decoder(&L.lit, S->litLen, btype == 1 ? 288 : S->nLit);
            if (btype == 1 && lane == 0) { S->litLen[286] = 0; S->litLen[287] = 0; }
            d4g_build_decoder(&L.dist, S->distLen, btype == 1 ? 30 : S->nDist);
            // ---- decodeStream — DeflateBlockHuffman.java:778-890 ----
            long long litlenBits = 0;
            while (true) {
                maybe_refill();
                // packed result: bit0 fail, bit1 eob, bits 8.. val (9 bits), bits 20.. dist (16 bits)
                long long res = 0;
                if (lane == 0) {
                    br.fill();
                    uint64_t bits = br.buf;
                    long long avail = br.nbits - br.pos;
                    int cl = 0, val = 0, dist = 0, edge = 0, used = 0;
                    bool ok = true;
                    int sym = d4g_decode_sym(&L.lit, bits, avail, &cl);
                    if (sym < 0 || sym > 285) ok = false;
                    else if (sym <= 256) { val = sym; used = cl; S->hist[sym]++; }
                    else {
                        int eb = d4g_lsym_ebits(sym);
                        int len = d4g_lsym_base(sym);
                        used = cl + eb;
                        if (used > avail) ok = false;
                        else {
                            len += (int)((bits >> cl) & ((1u << eb) - 1));
                            edge = (len == 258 && sym == 284);
                            int dcl = 0;
                            int ds = d4g_decode_sym(&L.dist, bits >> used, avail - used, &dcl);
                            if (ds < 0 || ds > 29) ok = false;
                            else {
                                int deb = d4g_dsym_ebits(ds);
                                if (used + dcl + deb > avail) ok = false;
                                else {
                                    dist = d4g_dsym_base(ds) + (int)((bits >> (used + dcl)) & ((1u << deb) - 1));
                                    used += dcl + deb;
                                    if (dist > nU) ok = false;  // reference: walks off the first block (NullPointerException)
                                    S->hist[sym]++;
                                    S->hist[D4G_NLIT + ds]++;
                                    val = len;
                                }
                            }
                        }
                    }
                    if (ok) {
                        br.skip(used);
                        litlenBits += used;
                        if (nTok < in.tokCap) {
                            tokA[in.tokBase + nTok] = (uint32_t)val | ((uint32_t)edge << 15) | ((uint32_t)dist << 16);
                            tokOff[in.tokBase + nTok] = (uint32_t)nU;
                        }
                        res = ((long long)(sym == 256) << 1) | ((long long)val << 8) | ((long long)dist << 20);
                    } else {
#ifdef D4G_HOSTSIM
                        if (getenv("D4G_DEBUG")) fprintf(stderr, "parse fail: tok %lld nU %lld pos %lld sym %d val %d dist %d used %d avail %lld cl %d\n", nTok, nU, br.pos, sym, val, dist, used, avail, cl);
#endif
                        res = 1;
                    }
                }
                res = __shfl(res, 0);
                if (res & 1) { status = -1; break; }
                nTok++;
                if (res & 2) break;
                int val = (int)((res >> 8) & 0x1ff), dist = (int)(res >> 20);
                if (dist == 0) {
                    if (lane == 0) {
                        L.win[nU & (D4G_WIN - 1)] = (uint8_t)val;
                        if (nU < in.uCap) U[in.uBase + nU] = (uint8_t)val;
                    }
                    nU++;
                } else {
                    // overlapping copies are periodic in `dist`: byte k = window[nU - dist + k % dist];
                    // all sources precede nU, all destinations follow it, so lanes never race.
                    long long src = nU - dist;
                    for (int k = lane; k < val; k += 64) {
                        int kk = k < dist ? k : k % dist;
                        uint8_t v = L.win[(src + kk) & (D4G_WIN - 1)];
                        L.win[(nU + k) & (D4G_WIN - 1)] = v;
                        if (nU + k < in.uCap) U[in.uBase + nU + k] = v;
                    }
                    nU += val;
                }
            }
            if (status != 0) break;
            if (lane == 0) {
                S->litlenBits = litlenBits;
                S->sizeBits = S->hdrBits + litlenBits;
                S->valid = 1;
                S->maskSlot = 0;
            }
            __syncthreads();
            blkSize = S->sizeBits;
            if (nStates < in.stCap) {
                D4GState* g = pstates + in.stBase + nStates;
                for (int i = lane; i < (int)(sizeof(D4GState) / 4); i += 64) ((uint32_t*)g)[i] = ((uint32_t*)S)[i];
                stateIdx = (int)nStates;
            }
            nStates++;
            __syncthreads();
        }
        if (lane == 0 && nBlocks < in.blkCap) {
            D4GParsedBlock pb;
            pb.type = btype;
            pb.stateIdx = stateIdx;
            pb.tokStart = blkTok0;
            pb.tokCount = nTok - blkTok0;
            pb.uStart = blkU0;
            pb.uLen = nU - blkU0;
            pb.sizeBits = blkSize;
            pblocks[in.blkBase + nBlocks] = pb;
        }
        // DeflateStream.getSizeBits; DeflateBlockUncompressed.getSizeBits alignment (:70-74)
        streamBits += 3;
        if (btype == 0) {
            long long c = streamBits % 8;
            c = c == 0 ? 0 : 8 - c;
            streamBits += ((nU - blkU0) + 4) * 8 + c;
        } else {
            streamBits += blkSize;
        }
        nBlocks++;
    }
    bool overflow = nTok > in.tokCap || nU > in.uCap || nBlocks > in.blkCap || nStates > in.stCap;
    long long pos = __shfl((long long)br.pos, 0);
    if (lane == 0) {
        D4GStreamOut o;
        o.status = status < 0 ? -1 : (overflow ? 1 : 0);
        o.nBlocks = nBlocks;
        o.nTok = nTok;
        o.nU = nU;
        o.nStates = nStates;
        if (pos > br.nbits) pos = br.nbits;
        o.consumedBytes = (pos + 7) >> 3;
        o.sizeBits = streamBits;
        outs[blockIdx.x] = o;
    }
}
