// d4g_types.h — data layout shared by the host orchestration and the HIP kernels of libdeft4g.
//
// Vocabulary follows the reference (deft4j): a *stream* is one raw DEFLATE stream, made of
// *blocks*; a Huffman block is a list of *tokens* (literal / back-reference / EOB, the
// reference's LitLen, B/deflate/LitLen.java:29-47) plus, when DYNAMIC, a header made of
// *RLE pairs* (DeflateBlockHuffman.java:43).  A candidate encoding of a block is a *state*.
//
// HBM layout (all arrays live in device memory for the whole call):
//   tok[t]    2 x u32 (one 8-byte load per token):
//             .x  literal byte | 256 (EOB) | match length, dist << 16 (0 for literals/EOB),
//                 bit 15 = the 284-as-258 edge case (DeflateBlockHuffman.java:843)
//             .y  offset of the token's decoded bytes in U (the stream's decoded data)
//   refs[r]   4 x u32 (one 16-byte load), one record per back-reference token, in token order (the search's
//             token passes read only these — literals never change):
//             .x  length | (length symbol - 257) << 9 | distance symbol << 14 | extra bits << 19
//             .y  offset of its decoded bytes in U
//             .z .w  its first eight decoded bytes (most literal-cost comparisons are decided within them, so a
//                    pass iteration needs no dependent load from U)
//   tokRef[t] u32  index into refs of back-reference token t (the writer's way from a token to its mask bit)
//   binStat / binMask  per block and length symbol: byte / distance-symbol / record statistics of its records and
//             the bit mask of its records (static; see D4G_NBINS below)
//   U[]       u8   decoded bytes of the whole stream, block after block
//   State     ~3.3 KB per candidate: code lengths, header RLE pairs, symbol histogram, sizes
//   masks     1 bit per back-reference: "this back-reference is expanded to literals".  The reference
//             only ever turns back-references into literals (replaceWithLiteralsIfSmaller
//             :222-296, removeDistLitLeastExpensive :373-458), never the reverse, so every
//             token list the search visits is the block's token range plus such a mask.
#pragma once
#include <stdint.h>

#define D4G_STORED 0
#define D4G_FIXED 1
#define D4G_DYNAMIC 2

#define D4G_NLIT 288   // padded literal/length alphabet (286 used)
#define D4G_NDIST 32   // padded distance alphabet (30 used)
#define D4G_HIST (D4G_NLIT + D4G_NDIST)
#define D4G_MAXPAIRS 320
// Static statistics of a block's back-reference records, one row per length symbol 257+b ("bin", the unit
// removeDistLitLeastExpensive works in): [0,256) how often each byte value occurs in the bin's records' decoded
// bytes, [256,286) how many of its records use each distance symbol, [286] its record count, [287] the sum of its
// records' extra bits.  Every per-bin sum that pass needs is linear in these.
#define D4G_NBINS 29
#define D4G_BINSTRIDE 320
#define D4G_BIN_DIST 256
#define D4G_BIN_COUNT 286
#define D4G_BIN_EBITS 287

// Encoded RLE pair (u16): bits 0-4 sym (0..18); bits 5-12 X; bit 13 expanded-to-literals.
//   literal length : X = 0, value = sym
//   sym 16 (copy)  : X = (run-3) | value<<2      (run 3..6, value 0..15)
//   sym 17 / 18    : X = run                      (3..10 / 11..138), value 0
#define D4G_PAIR_EXPANDED 0x2000u

struct D4GState {
    int32_t valid;     // 0 = "null" in the reference (e.g. optimiseBlockNormal returned null)
    int32_t type;      // D4G_FIXED / D4G_DYNAMIC (stored blocks never enter the search)
    int32_t nLit, nDist, nCl, nPairs;
    int32_t maskSlot;  // index of this state's token mask in the block's mask pool
    int32_t flags;     // op-specific (bit0: recodedHuffmanFull returned its input)
    int64_t sizeBits, litlenBits, hdrBits;  // sizeBits == litlenBits + hdrBits (3 prolog bits excluded)
    int64_t pad0;
    uint8_t litLen[D4G_NLIT];
    uint8_t distLen[D4G_NDIST];
    uint8_t clLen[32];
    uint16_t pairs[D4G_MAXPAIRS];
    uint32_t hist[D4G_HIST];  // [0,286) literal/length symbols, [288,318) distance symbols
};

// One deflate block of a stream as the parser found it / as the optimiser left it.
struct D4GBlock {
    int32_t type;        // D4G_STORED / FIXED / DYNAMIC
    int32_t stream;      // owning stream
    int64_t tokStart;    // first token (index into tok); tokCount includes the final EOB
    int64_t tokCount;
    int64_t uBase;       // start of the owning stream's region in U (tokOff values are relative to it)
    int64_t uStart;      // decoded bytes [uStart, uStart+uLen) of the stream's region
    int64_t uLen;
    int64_t sizeBits;    // encoded size without the 3 prolog bits (stored: without alignment)
    int64_t stateIdx;    // index of the block's current state in the block-state array
    int64_t maskBase;    // first u64 word of this block's mask pool
    int64_t maskWords;   // u64 words per mask = ceil(refCount / 64)
    int64_t refStart;    // first back-reference record (index into refs); adjacent blocks are contiguous
    int64_t refCount;
    int64_t binStat;     // first u32 of the block's per-length-symbol statistics (D4G_NBINS x D4G_BINSTRIDE), -1: none
    int64_t binMask;     // first u64 of the block's per-length-symbol record masks (D4G_NBINS x maskWords)
    int64_t passMemo;    // first u64 of the block's token-pass memo (D4G_PASSMEMO_SLOTS entries), -1: none
    int64_t passMemoStride;  // u64 words per entry: header + histogram delta + one mask
};

// Ops of the candidate-search program (one optimiseBlock call = one program run per block).
enum D4GOpKind {
    OP_OPT = 1,         // dst = copy(src).optimise()                      DeflateBlockHuffman.java:460-469
    OP_RECODE = 2,      // dst = recodedHuffman(src, prune)                DeflateStream.java:200-210
    OP_RECODE_FULL = 3, // dst = recodedHuffmanFull(src)                   DeflateStream.java:212-229
    OP_LEAST = 4,       // dst = leastExpPruned / leastSeenPruned(src)     DeflateStream.java:231-241
    OP_POST = 5,        // dst = copy(src).recodeHeader()                  DeflateBlockHuffman.java:579-629
    OP_PRUNEHDR = 6,    // dst = copy(src).recodeHeaderToLessRLEMatches()  :632-635
    OP_TOFIXED_OPT = 7, // dst = toFixedHuffman(src).optimise()            DeflateStream.java:329-337,470-478
    OP_CAND = 8,        // offer src as a candidate                        DeflateStream.java:349-368
    OP_HDRSEARCH = 9,   // the 56 optimiseBlockDynBlock candidates of one base block  :265-317
};

struct D4GOp {
    int32_t kind;
    int32_t src;       // state slot
    int32_t dst;       // state slot (-1: none)
    int32_t arg;       // prune / mode / requireSaved
    int32_t seq;       // candidate sequence number in the reference's enumeration order (-1: not a candidate)
    int32_t maskSlot;  // mask slot the op may write (token-changing ops), -1 otherwise
    int32_t scratch;   // OP_RECODE_FULL: first of two scratch state slots
    int32_t scratchMask;  // OP_RECODE_FULL: first of two scratch mask slots
};

#define D4G_KEY_SEQ_BITS 20
#define D4G_KEY_NONE 0x7fffffffffffffffLL
// candidate key: smaller size first, then earlier enumeration order ("strict <, first wins")
#define D4G_MAKE_KEY(size, seq) ((((int64_t)(size)) << D4G_KEY_SEQ_BITS) | (int64_t)(seq))

// Per-block result of one optimiseBlock round (read back by the host).
struct D4GRoundResult {
    int64_t curSize;   // size of the block's state before the round
    int64_t bestSize;  // size of the best non-stored candidate (== curSize when nothing beat it)
    int32_t bestSeq;   // op id of the winner (-1: the current state itself; 0: the "optimised" candidate)
    int32_t improved;  // bestSize < curSize
    int32_t newType;   // block type of the state now in slot 0
    int32_t pad;
};
