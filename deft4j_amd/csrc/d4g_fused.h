// d4g_fused.h — the candidate search (DeflateStream.optimiseBlock, B/deflate/DeflateStream.java:343-490) of one deflate
// block as ONE persistent workgroup that keeps the whole search in LDS and runs round after round of the reference's
// per-block fixpoint (DeflateStream.optimise :496-566) without going back to the host.
//
// What the reference's ~570 candidate states of a block are made of is very little distinct data: a candidate is
// (token mask, Huffman code, header), and
//   * which back-references a code would turn into literals does not depend on the mask: E(code) = { r : literal bits of
//     r < (<=) bits of r } is swept ONCE per distinct code (both comparison modes in one sweep over the block's
//     records), and replaceWithLiteralsIfSmaller (DeflateBlockHuffman.java:222-296) on any state with that code is
//     mask | E(code) plus a walk over the few records in E \ mask;
//   * a Huffman rebuild (recodeHuffman :670-743) is a function of the mask's histogram, its default header of the code;
//   * the header operations (:471-476, :579-635) touch the header only, the 56 header candidates of a base block
//     (DeflateStream.java:265-317) only its code lengths.
// So every state here is three small ids (mask, code, header) plus its literal/length bit count; every operation is a
// function on ids, memoised in LDS hash tables under its exact key, and equal values are given equal ids (codes by
// content, masks by content hash confirmed word by word).  The ~570 ops of the program become a few dozen distinct
// sweeps / tree builds / header searches per round; the ops themselves are bookkeeping done by one thread each.
//
// Execution is bulk-synchronous inside the workgroup: `advance` lets every op take the steps whose inputs exist and
// queue the tasks it is missing; then the queued tasks of one kind run side by side on the workgroup's waves (sweep:
// all waves; token-mask updates, least-expensive-length pruning, Huffman trees, header operations, header searches:
// one wave per task); repeat until every op has finished.  The JDK-exact tree builder (one wave) therefore no longer
// stalls a workgroup's other waves at a barrier for one op at a time: every tree of a step is built at once, and ops
// that do not need it are not waiting behind it.  The winner is assembled into the block's state slot 0 / mask 0
// exactly as k_select leaves them, so everything downstream (merge, bit packing) is unchanged.
//
// Anything that does not fit the LDS tables (ids, queues) abandons the round with the block untouched and reports it;
// the host then runs that block through the level executor (k_exec_state_ops), which has no such limits.
#pragma once
#include "d4g_ops.h"

#define D4F_MAXSLOTS 576
#define D4F_MAXOPS 640
#define D4F_MAXM 256      // distinct token masks of a round (mask pool slots [0, 256); 0 = the block's own mask)
#define D4F_MAXC 96       // distinct codes of a round (E-sets live in mask pool slots [256, 256 + 2 * 96))
#define D4F_MAXH 384      // distinct headers of a round
#define D4F_PASSN 1024
#define D4F_LEASTN 64
#define D4F_HDRN 256
#define D4F_SWEEP_K 8     // codes evaluated per sweep over the records
#define D4F_CODE_FIXED 1  // reserved code id: the fixed Huffman code
#define D4F_MAXROUNDS 16
#define D4F_HS_SLOTS 3      // header searches running side by side (one wave each)
#define D4F_TREE_SLOTS 4   // Huffman rebuilds running side by side (two waves each: literal/length and distance tree)

#ifdef D4G_HOSTSIM
D4G_DEV void d4f_fence_block() {}
#else
D4G_DEV void d4f_fence_block() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
#endif

// ---- per-block tables in HBM (carved from the block's state slots 1.. — the level executor's slots, unused here) ----
struct D4FCode {            // 384 B
    uint8_t lens[D4G_NLIT + D4G_NDIST];
    int32_t nLit, nDist, type, err;
    int32_t pad[12];
};
struct D4FPairs {           // 656 B: a header's RLE pairs without their expanded-to-literals flags
    uint16_t pairs[D4G_MAXPAIRS];
    int32_t nPairs, pad[3];
};
struct D4FHdr {             // 96 B
    uint8_t clLen[32];
    uint32_t flags[10];     // pair i expanded to literals
    int32_t nCl, bits, base, pad;   // base: index into the pairs table
    int32_t pad2[2];
};
struct D4FGlob {
    uint32_t* hist;         // [D4F_MAXM][D4G_HIST]
    D4FCode* code;          // [D4F_MAXC]
    D4FPairs* pairs;        // [D4F_MAXC + 1]: [0] the block's incoming header, [1 + c] the default header of code c
    D4FHdr* hdr;            // [D4F_MAXH]
    int32_t* binBase;       // [D4F_MAXC][2][32]: per length symbol, literal minus back-reference bits / bytes without a code, over all records
};
#define D4F_GLOB_BYTES ((size_t)D4F_MAXM * D4G_HIST * 4 + (size_t)D4F_MAXC * sizeof(D4FCode) + (size_t)(D4F_MAXC + 1) * sizeof(D4FPairs) + \
                        (size_t)D4F_MAXH * sizeof(D4FHdr) + (size_t)D4F_MAXC * 64 * 4)

struct D4FParams {
    const D4GOp* ops[2];    // [0] program of a DYNAMIC block, [1] of a FIXED block
    int32_t nOps[2];
    int32_t maxRounds;      // optimiseBlock rounds to run while the block keeps improving (1 for a merge attempt)
    int32_t regWords;       // blocks of up to this many mask words take the register forms of the mask tasks (64 * D4F_NWR; tests: 0)
    D4GRoundResult* results;   // [nActive][D4F_MAXROUNDS]
    int32_t* roundInfo;        // [nActive]: rounds completed | D4F_INFO_*
    long long* stats;          // optional [32] counters
};
// ---- cluster mode: many workgroups on ONE long block (the merge chain of a big stream) ----
// Workgroup 0 runs the search exactly as k_search_fused does; what scales with the block's length — the sweeps over the
// records, the mask updates, the record walks of the least-expensive pruning — it posts as *commands* that every workgroup
// of the launch, itself included, works off in items of a few thousand records.  The control workgroup never depends on a
// helper being resident: it takes items itself until none is left and then waits only for items already taken.
// Hand-off discipline (per-XCD L2s are not coherent): every command uses a FRESH slot of the arena (no workgroup has ever
// read those lines); parameters are written through (sc1), accumulators are device-scope atomics; bulk results (E-set and
// mask words) are written once to lines nobody has read, then released; readers acquire before their first read.
#define D4F_CL_MAXT 48        // tasks per command
#define D4F_CL_SLOTS 96       // command slots of a launch (more commands than that: the control workgroup works alone)
#define D4F_CL_CHUNK (64 * D4F_NWR)   // mask words per item of the mask commands
#define D4F_CL_SWEEP_WORDS 16 // mask words per item of a sweep
enum { D4F_CMD_SWEEP = 1, D4F_CMD_APPLY = 2, D4F_CMD_LEAST_WALK = 3, D4F_CMD_LEAST_EXPAND = 4 };
struct D4FClTask {
    int32_t m, code, prune, leaf, mNew, mode, rem, via;
    uint32_t full;
    int32_t pad0[3];
    int32_t cnt, saved;                 // accumulators from here on (atomics)
    unsigned long long h1, h2;
    int32_t bin[3][32];                 // size, freq, bytes without a code — per length symbol
    int32_t delta[D4G_HIST];
    int32_t pad1[14];
};
static_assert(sizeof(D4FClTask) % 128 == 0, "tasks on cache lines of their own");
struct D4FClCmd {
    int32_t cmd, nItems, nTasks, nk, nChunks, pad0[3];
    int32_t codes[D4F_SWEEP_K];
    int32_t next, done, pad1[14];       // atomics (a line of their own)
    uint32_t neq[D4F_SWEEP_K];
    int32_t pad2[24];
    D4FClTask task[D4F_CL_MAXT];
};
struct D4FClArena {
    int32_t epoch;                      // commands posted (atomic); -1: the launch is over
    int32_t pad[31];
    D4FClCmd slot[D4F_CL_SLOTS];
};

#define D4F_INFO_FALLBACK 0x10000   // the round after the completed ones did not fit the tables: run it with the level executor
#define D4F_INFO_MORE 0x20000       // still improving when maxRounds was reached

// ---- LDS ----
struct D4FSlot { int16_t m, c, h; uint8_t valid, type; int32_t lazy; long long litlen; };   // lazy: the tokens are mask m with E0(code c) still to be expanded
struct D4FPassE { uint32_t key; int32_t saved; int16_t mOut; uint8_t st, pad; };
struct D4FLeastE { uint32_t key; int32_t delta; int16_t mOut; uint8_t st, pad; };
struct D4FHdrE { uint32_t key; int16_t hOut; uint8_t st, pad; };

enum { D4F_Q_SWEEP = 0, D4F_Q_APPLY, D4F_Q_BINBASE, D4F_Q_LEAST, D4F_Q_TREE, D4F_Q_HDR, D4F_Q_HS, D4F_Q_FIXDOT, D4F_NQ };
// task queues of one step: capacities and offsets into one array (tasks are ids / memo entry indices)
#define D4F_QCAP_SWEEP D4F_MAXC
#define D4F_QCAP_APPLY 256
#define D4F_QCAP_BINBASE D4F_MAXC
#define D4F_QCAP_LEAST 64
#define D4F_QCAP_TREE 64
#define D4F_QCAP_HDR 256
#define D4F_QCAP_HS D4F_MAXC
#define D4F_QCAP_FIXDOT 4
#define D4F_QTOTAL (D4F_QCAP_SWEEP + D4F_QCAP_APPLY + D4F_QCAP_BINBASE + D4F_QCAP_LEAST + D4F_QCAP_TREE + D4F_QCAP_HDR + D4F_QCAP_HS + D4F_QCAP_FIXDOT)
D4G_DEV int d4f_qcap(int q) {
    return q == D4F_Q_SWEEP ? D4F_QCAP_SWEEP : q == D4F_Q_APPLY ? D4F_QCAP_APPLY : q == D4F_Q_BINBASE ? D4F_QCAP_BINBASE : q == D4F_Q_LEAST ? D4F_QCAP_LEAST
         : q == D4F_Q_TREE ? D4F_QCAP_TREE : q == D4F_Q_HDR ? D4F_QCAP_HDR : q == D4F_Q_HS ? D4F_QCAP_HS : D4F_QCAP_FIXDOT;
}
D4G_DEV int d4f_qoff(int q) {
    int o = 0;
    if (q > D4F_Q_SWEEP) o += D4F_QCAP_SWEEP;
    if (q > D4F_Q_APPLY) o += D4F_QCAP_APPLY;
    if (q > D4F_Q_BINBASE) o += D4F_QCAP_BINBASE;
    if (q > D4F_Q_LEAST) o += D4F_QCAP_LEAST;
    if (q > D4F_Q_TREE) o += D4F_QCAP_TREE;
    if (q > D4F_Q_HDR) o += D4F_QCAP_HDR;
    if (q > D4F_Q_HS) o += D4F_QCAP_HS;
    return o;
}
enum { D4F_H_OPT = 1, D4F_H_POST = 2, D4F_H_PRUNE = 3 };

// scratch of one Huffman-rebuild task (one wave): the three trees are built one after the other in the same memory
typedef TreeMem<uint64_t, uint16_t, D4G_NLIT, 16, true> D4FLitTree;    // (depth scratch of the limiter path overlays the queue)
typedef TreeMem<uint64_t, uint16_t, D4G_NDIST, 16, true> D4FDistTree;
struct D4FTreeScr {
    alignas(16) unsigned char tree[(D4FLitTree::bytes(1) + 15) & ~15];       // literal/length tree, then the code-length tree
    alignas(16) unsigned char treeD[(D4FDistTree::bytes(1) + 15) & ~15];     // distance tree (built beside the literal/length tree)
    uint32_t hist[D4G_HIST];
    alignas(16) uint8_t lens[D4G_NLIT + D4G_NDIST];
    uint16_t pairs[D4G_MAXPAIRS];
    uint8_t clLen[32];
    uint32_t clFreq[20];
    int32_t nLit, nDist, nCl, nPairs, hdrBits, err, m, pad;
    long long litlen;
};
// scratch of one wave for mask updates, least-expensive pruning and header operations
struct D4FWaveScr {
    uint32_t hist[D4G_HIST];
    uint16_t lc[256];
    uint8_t cl[64];          // [0, 29) length-symbol bits, [32, 62) distance-symbol bits
    uint32_t queue[128];
    int32_t binSize[32], binFreq[32], binZ[32];
    int32_t delta[D4G_HIST];
    int32_t misc[16];
};
struct D4FHdrScr {           // header operation of one wave (overlays D4FWaveScr)
    uint16_t pairs[D4G_MAXPAIRS];
    uint8_t clLen[32];
    uint32_t clFreq[20];
    alignas(16) unsigned char tree[256];
    int32_t nCl, nPairs, bits, pad;
};
struct D4FSweepScr {
    uint16_t lc[D4F_SWEEP_K][256];
    uint8_t cl[D4F_SWEEP_K][64];
    uint32_t neq[D4F_SWEEP_K];
    unsigned long long stage[8][D4F_SWEEP_K][2][16];   // per wave: the E-set words of the group in hand (d4f_sweep_group)
};
struct D4FHsScr {
    D4GHdrLds H;
    alignas(16) uint8_t comb[D4G_NLIT + D4G_NDIST];
};
union D4FScratch {
    D4FTreeScr tree[D4F_TREE_SLOTS];
    D4FWaveScr wave[8];
    D4FSweepScr sweep;
    D4GLds legacy;           // assembling the winner (the level executor's header functions)
};
static_assert(sizeof(D4FHdrScr) <= sizeof(D4FWaveScr), "header scratch overlays the wave scratch");

struct D4FLds {
    D4FSlot slot[D4F_MAXSLOTS];
    uint8_t slotReady[D4F_MAXSLOTS];
    uint8_t opStage[D4F_MAXOPS];
    uint16_t opReq[D4F_MAXOPS];
    long long treeLit[D4F_MAXM];      // Huffman rebuild of mask id m: literal/length bits under the rebuilt code ...
    uint8_t treeC[D4F_MAXM];          // ... its code id ...
    alignas(4) uint8_t treeSt[D4F_MAXM];         // ... 0 not asked, 1 queued, 2 there
    int32_t hdrBits[D4F_MAXH];
    unsigned long long maskH1[D4F_MAXM], maskH2[D4F_MAXM];   // content hashes of the masks (relative to mask 0)
    int32_t maskPop[D4F_MAXM];        // expanded records
    uint32_t maskFull[D4F_MAXM];      // length symbols known to have every record expanded (bit per symbol): they take no part in least-expensive pruning
    alignas(4) uint16_t maskHash[512];           // id + 1 by maskH1
    uint16_t maskStep[D4F_MAXM];      // step in which the mask was published (content is compared only with masks of earlier steps: their words are visible)
    uint16_t maskHH[D4F_MAXM];        // hash of the mask's histogram: masks with equal histograms share one Huffman rebuild (checked entry by entry)
    int32_t treeCand[2][D4F_QCAP_TREE]; // tree step: per queued mask, an equal-histogram mask whose rebuild stands / an earlier queue position
    int16_t defHdr[D4F_MAXC];         // header id of the code's default header, -1: not built
    uint16_t qAll[D4F_QTOTAL];
    int32_t qn[D4F_NQ];
    int32_t nMask, nHdr, nDone, fallback, progress[2], fixdotSt, fixdotM, step, pad0;
    long long fixdotLit;
    long long red[32];
    int32_t misc[32];
    // The header searches run last in a round, when the three tables below and the scratch are dead: their scratch
    // (D4F_HS_SLOTS searches side by side) overlays all four.
    alignas(16) D4FLeastE least[D4F_LEASTN];
    D4FHdrE hdrReq[D4F_HDRN];
    D4FPassE pass[D4F_PASSN];
    D4FScratch scr;
    // ---- not cleared between rounds ----
    // The code table: what is known about a code is a function of its lengths (and the block's records), not of the
    // round — E-sets, per-length-symbol sums, header-search results stay valid while the launch runs its rounds.
    unsigned long long codeH[D4F_MAXC];
    alignas(4) uint8_t codeHash[256];            // id + 1 by codeH
    alignas(4) uint8_t eState[D4F_MAXC];         // E-sets of the code: 0 none, 1 queued, 2 there
    alignas(4) uint8_t eEq[D4F_MAXC];            // both comparison modes expand the same records
    alignas(4) uint8_t bbState[D4F_MAXC];        // per-length-symbol sums of the code
    alignas(4) uint8_t hsState[D4F_MAXC];
    int32_t hsBits[D4F_MAXC];
    alignas(4) uint8_t hsLane[D4F_MAXC];
    int32_t nCode, pad2;
    // what the kernel was called with, the block, the round in progress
    D4GCtx c;
    D4GBlock b;
    D4FGlob G;
    const D4GOp* ops;
    int32_t nOps, curType, rounds, info, improved, regWords;
    struct D4FClArena* cl;    // cluster mode (k_search_cluster): the command area shared with the helper workgroups, else null
    int32_t clEpoch, clDoneWg; // commands posted so far / items this workgroup finished in the command in hand
    int32_t which, pad3;       // the block's index in the launch's active list
    long long curSize;
};

static_assert(sizeof(D4FLds) <= 80 * 1024, "two workgroups per CU: 80 KiB of LDS each");
static_assert(offsetof(D4FLds, codeH) - offsetof(D4FLds, least) >= D4F_HS_SLOTS * sizeof(D4FHsScr), "header-search scratch overlay");
// The workgroup's LDS, at namespace scope: the task functions below are real calls (one register budget each instead of
// one for the whole kernel) and still address it with LDS instructions.
__shared__ D4FLds d4fLds;
#define D4F_TASK __device__ __attribute__((noinline))
#define D4F_CTX const D4GCtx& c = d4fLds.c; const D4GBlock& b = d4fLds.b; const D4FGlob& G = d4fLds.G; (void)c; (void)b; (void)G

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
D4G_DEV unsigned long long d4f_mix1(unsigned long long i, unsigned long long w) {
    unsigned long long x = (w + 1) * 0x9e3779b97f4a7c15ULL + (i + 1) * 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 29; x *= 0x94d049bb133111ebULL; x ^= x >> 32;
    return x;
}
D4G_DEV unsigned long long d4f_mix2(unsigned long long i, unsigned long long w) {
    unsigned long long y = (w + 0x632be59bd9b4e019ULL) * ((i + 7) * 0xd6e8feb86659fd93ULL | 1ULL);
    y ^= y >> 31; y *= 0xff51afd7ed558ccdULL; y ^= y >> 33;
    return y;
}
D4G_DEV uint64_t* d4f_mask(const D4GCtx& c, const D4GBlock& b, int id) { return mask_ptr(c, b, id); }
D4G_DEV uint64_t* d4f_eset(const D4GCtx& c, const D4GBlock& b, int code, int prune) { return mask_ptr(c, b, D4F_MAXM + 2 * code + (prune ? 1 : 0)); }
D4G_DEV long long d4f_slot_size(const D4FSlot& s) {
    const D4FLds& F = d4fLds; return s.litlen + (s.type == D4G_DYNAMIC ? (long long)F.hdrBits[s.h] : 0LL); }

D4G_DEV void d4f_push(int q, int v) {
    D4FLds& F = d4fLds;
    int k = atomicAdd(&F.qn[q], 1);
    if (k < d4f_qcap(q)) F.qAll[d4f_qoff(q) + k] = (uint16_t)v;
    else F.fallback = 1;
}
// a wave's tree (all lanes call): the wave-wide JDK heap on the GPU; the emulator runs the one-lane form unless asked
template <int NREG, typename TM, typename FreqFn, typename OutFn>
__device__ __forceinline__ int d4f_wave_tree(TM& tm, int numSymbols, int limit, FreqFn freq, OutFn out) {
#if !defined(D4G_HOSTSIM) || defined(D4G_SIM_WAVE_HEAP)
    return d4g_build_tree_wave64<NREG>(tm, numSymbols, limit, freq, out);
#else
    int err = 0;
    if ((threadIdx.x & 63) == 0) err = d4g_build_tree(tm, 1, 0, numSymbols, limit, freq, out);
    return __shfl(err, 0);
#endif
}

// ---------------------------------------------------------------------------------------
// memo requests (one thread): find or create the entry of an exact key; the creator queues the task
// ---------------------------------------------------------------------------------------
D4G_DEV int d4f_req_eset(int code) {
    D4FLds& F = d4fLds;
    if (F.eState[code] == 0) {
        // (several threads may get here in one step: the byte-wide state is claimed through the 32-bit word that holds it)
        unsigned* w = (unsigned*)&F.eState[code & ~3];
        const unsigned sh = (unsigned)(code & 3) * 8;
        unsigned old = atomicOr(w, 1u << sh);
        if (((old >> sh) & 0xff) == 0) d4f_push(D4F_Q_SWEEP, code);
    }
    return code;
}
D4G_DEV int d4f_req_pass(int m, int code, int prune, int leaf) {
    D4FLds& F = d4fLds;
    const uint32_t key = 1u + ((uint32_t)m | ((uint32_t)code << 8) | ((uint32_t)prune << 15) | ((uint32_t)leaf << 16));
    uint32_t k = (key * 0x9e3779b1u) >> 22;   // 10 bits
    for (int probe = 0; probe < D4F_PASSN; probe++, k = (k + 1) & (D4F_PASSN - 1)) {
        uint32_t old = atomicCAS(&F.pass[k].key, 0u, key);
        if (old == 0) {
            d4f_req_eset(code);
            d4f_push(D4F_Q_APPLY, (int)k);
            return (int)k;
        }
        if (old == key) return (int)k;
    }
    F.fallback = 1;
    return 0;
}
D4G_DEV int d4f_req_binbase(int code) {
    D4FLds& F = d4fLds;
    if (F.bbState[code] == 0) {
        unsigned* w = (unsigned*)&F.bbState[code & ~3];
        const unsigned sh = (unsigned)(code & 3) * 8;
        unsigned old = atomicOr(w, 1u << sh);
        if (((old >> sh) & 0xff) == 0) d4f_push(D4F_Q_BINBASE, code);
    }
    return code;
}
D4G_DEV int d4f_req_least(int m, int code, int mode) {
    D4FLds& F = d4fLds;
    const uint32_t key = 1u + ((uint32_t)m | ((uint32_t)code << 8) | ((uint32_t)mode << 15));
    uint32_t k = (key * 0x9e3779b1u) >> 26;   // 6 bits
    for (int probe = 0; probe < D4F_LEASTN; probe++, k = (k + 1) & (D4F_LEASTN - 1)) {
        uint32_t old = atomicCAS(&F.least[k].key, 0u, key);
        if (old == 0) {
            d4f_req_binbase(code);
            d4f_push(D4F_Q_LEAST, (int)k);
            return (int)k;
        }
        if (old == key) return (int)k;
    }
    F.fallback = 1;
    return 0;
}
D4G_DEV int d4f_req_hdr(int h, int op) {
    D4FLds& F = d4fLds;
    const uint32_t key = 1u + ((uint32_t)h | ((uint32_t)op << 12));
    uint32_t k = (key * 0x9e3779b1u) >> 24;   // 8 bits
    for (int probe = 0; probe < D4F_HDRN; probe++, k = (k + 1) & (D4F_HDRN - 1)) {
        uint32_t old = atomicCAS(&F.hdrReq[k].key, 0u, key);
        if (old == 0) { d4f_push(D4F_Q_HDR, (int)k); return (int)k; }
        if (old == key) return (int)k;
    }
    F.fallback = 1;
    return 0;
}
D4G_DEV void d4f_req_tree(int m) {
    D4FLds& F = d4fLds;
    if (F.treeSt[m] == 0) {
        unsigned* w = (unsigned*)&F.treeSt[m & ~3];
        const unsigned sh = (unsigned)(m & 3) * 8;
        unsigned old = atomicOr(w, 1u << sh);
        if (((old >> sh) & 0xff) == 0) d4f_push(D4F_Q_TREE, m);
    }
}
D4G_DEV void d4f_req_hs(int code) {
    D4FLds& F = d4fLds;
    if (F.hsState[code] == 0) {
        unsigned* w = (unsigned*)&F.hsState[code & ~3];
        const unsigned sh = (unsigned)(code & 3) * 8;
        unsigned old = atomicOr(w, 1u << sh);
        if (((old >> sh) & 0xff) == 0) d4f_push(D4F_Q_HS, code);
    }
}

// ---------------------------------------------------------------------------------------
// advance: one op takes the steps whose inputs exist (one thread).  Returns the op's candidate key through `key` when
// it finishes.  Stage 255 = finished.
// ---------------------------------------------------------------------------------------
D4G_DEV void d4f_finish(int dst, int m, int c, int h, int type, int valid, long long litlen, int lazy = 0) {
    D4FLds& F = d4fLds;
    if (dst >= 0) {
        D4FSlot s;
        s.m = (int16_t)m; s.c = (int16_t)c; s.h = (int16_t)h; s.valid = (uint8_t)valid; s.type = (uint8_t)type; s.lazy = lazy; s.litlen = litlen;
        F.slot[dst] = s;
        d4f_fence_block();
        F.slotReady[dst] = 1;
    }
}

D4F_TASK bool d4f_advance_op(const D4GOp op, int opId, long long* bestKeyP) {
    D4FLds& F = d4fLds;
    long long& bestKey = *bestKeyP;
    int stage = F.opStage[opId];
    if (stage == 255) return false;
    if (!F.slotReady[op.src]) return false;
    d4f_fence_block();
    const D4FSlot src = F.slot[op.src];
    bool progressed = false;
    auto done = [&](long long key) {
        if (key < bestKey) bestKey = key;
        F.opStage[opId] = 255;
        atomicAdd(&F.nDone, 1);
        progressed = true;
    };
    auto cand_key = [&](int valid, long long size) -> long long {
        return (op.seq >= 0 && valid) ? D4G_MAKE_KEY(size, (long long)opId * 64) : D4G_KEY_NONE;
    };
    if (op.kind == OP_CAND) {
        done(src.valid ? D4G_MAKE_KEY(d4f_slot_size(src), (long long)opId * 64) : D4G_KEY_NONE);
        return true;
    }
    if (op.kind == OP_HDRSEARCH) {
        if (!src.valid || src.type != D4G_DYNAMIC) { done(D4G_KEY_NONE); return true; }
        if (stage == 0) { d4f_req_hs(src.c); F.opStage[opId] = 1; stage = 1; progressed = true; }
        if (F.hsState[src.c] == 2) {
            d4f_fence_block();
            done(D4G_MAKE_KEY(src.litlen + (long long)F.hsBits[src.c], (long long)opId * 64 + F.hsLane[src.c]));
        }
        return progressed;
    }
    if (!src.valid) {   // the reference never builds this candidate (null / skipped branch)
        d4f_finish(op.dst, src.m, src.c, src.h, src.type, 0, src.litlen);
        done(D4G_KEY_NONE);
        return true;
    }
    const bool dyn = src.type == D4G_DYNAMIC;
    switch (op.kind) {
    case OP_POST:
    case OP_PRUNEHDR: {
        if (!dyn) {   // the header functions leave a block without a dynamic header alone
            d4f_finish(op.dst, src.m, src.c, src.h, src.type, 1, src.litlen);
            done(cand_key(1, d4f_slot_size(src)));
            break;
        }
        if (stage == 0) {
            F.opReq[opId] = (uint16_t)d4f_req_hdr(src.h, op.kind == OP_POST ? D4F_H_POST : D4F_H_PRUNE);
            F.opStage[opId] = 1; stage = 1; progressed = true;
        }
        const D4FHdrE& e = F.hdrReq[F.opReq[opId]];
        if (e.st == 2) {
            d4f_fence_block();
            d4f_finish(op.dst, src.m, src.c, e.hOut, src.type, 1, src.litlen);
            done(cand_key(1, src.litlen + F.hdrBits[e.hOut]));
        }
        break;
    }
    case OP_OPT: {
        const int leaf = (op.arg >> 8) & 1;
        if (stage == 0) {
            F.opReq[opId] = (uint16_t)d4f_req_pass(src.m, src.c, 0, leaf);
            if (dyn) d4f_req_hdr(src.h, D4F_H_OPT);
            F.opStage[opId] = 1; stage = 1; progressed = true;
        }
        const D4FPassE& e = F.pass[F.opReq[opId]];
        if (e.st != 2) break;
        int hOut = src.h;
        if (dyn) {
            const D4FHdrE& he = F.hdrReq[d4f_req_hdr(src.h, D4F_H_OPT)];
            if (he.st != 2) break;
            hOut = he.hOut;
        }
        d4f_fence_block();
        const long long lit = src.litlen - e.saved;
        const long long size = lit + (dyn ? (long long)F.hdrBits[hOut] : 0LL);
        const int valid = (op.arg & 1) ? (d4f_slot_size(src) - size > 0) : 1;
        d4f_finish(op.dst, leaf ? src.m : e.mOut, src.c, hOut, src.type, valid, lit, leaf && e.saved > 0 ? 1 : 0);   // (every record of E0 saves at least a bit)
        done(cand_key(valid, size));
        break;
    }
    case OP_RECODE: {
        if (stage == 0) {
            if (op.arg & 1) { F.opReq[opId] = (uint16_t)d4f_req_pass(src.m, src.c, 1, 0); stage = 1; }
            else { F.opReq[opId] = (uint16_t)src.m; d4f_req_tree(src.m); stage = 2; }
            F.opStage[opId] = (uint8_t)stage; progressed = true;
        }
        if (stage == 1) {
            const D4FPassE& e = F.pass[F.opReq[opId]];
            if (e.st != 2) break;
            d4f_fence_block();
            const int m2 = e.mOut;
            F.opReq[opId] = (uint16_t)m2;
            d4f_req_tree(m2);
            stage = 2; F.opStage[opId] = 2; progressed = true;
        }
        if (stage == 2) {
            const int m2 = F.opReq[opId];
            if (F.treeSt[m2] != 2) break;
            d4f_fence_block();
            const int c2 = F.treeC[m2];
            const int h2 = F.defHdr[c2];
            const long long lit = F.treeLit[m2];
            d4f_finish(op.dst, m2, c2, h2, D4G_DYNAMIC, 1, lit);
            done(cand_key(1, lit + F.hdrBits[h2]));
        }
        break;
    }
    case OP_RECODE_FULL: {   // recodedHuffmanFull — DeflateStream.java:212-229; the running state lives in the op's scratch slots
        D4FSlot& cur = F.slot[op.scratch];
        D4FSlot& aux = F.slot[op.scratch + 1];   // litlen = size to beat, valid = nothing kept so far
        if (stage == 0) {
            cur = src;
            aux.litlen = d4f_slot_size(src);
            aux.valid = 1;
            F.opReq[opId] = (uint16_t)d4f_req_pass(src.m, src.c, 1, 0);
            stage = 1; F.opStage[opId] = 1; progressed = true;
        }
        for (;;) {
            if (stage == 1) {
                const D4FPassE& e = F.pass[F.opReq[opId]];
                if (e.st != 2) break;
                d4f_fence_block();
                const int m2 = e.mOut;
                F.opReq[opId] = (uint16_t)m2;
                d4f_req_tree(m2);
                stage = 2; F.opStage[opId] = 2; progressed = true;
            }
            if (stage == 2) {
                const int m2 = F.opReq[opId];
                if (F.treeSt[m2] != 2) break;
                d4f_fence_block();
                const int c2 = F.treeC[m2];
                const int h2 = F.defHdr[c2];
                const long long lit = F.treeLit[m2];
                const long long size2 = lit + F.hdrBits[h2];
                if (size2 >= aux.litlen) {
                    const int same = aux.valid;
                    d4f_finish(op.dst, cur.m, cur.c, cur.h, cur.type, !same, cur.litlen);
                    done(cand_key(!same, d4f_slot_size(cur)));
                    break;
                }
                cur.m = (int16_t)m2; cur.c = (int16_t)c2; cur.h = (int16_t)h2; cur.type = D4G_DYNAMIC; cur.litlen = lit;
                aux.litlen = size2;
                aux.valid = 0;
                F.opReq[opId] = (uint16_t)d4f_req_pass(m2, c2, 1, 0);
                stage = 1; F.opStage[opId] = 1; progressed = true;
            }
        }
        break;
    }
    case OP_LEAST: {
        if (!dyn) {
            d4f_finish(op.dst, src.m, src.c, src.h, src.type, 1, src.litlen);
            done(cand_key(1, d4f_slot_size(src)));
            break;
        }
        if (stage == 0) {
            F.opReq[opId] = (uint16_t)d4f_req_least(src.m, src.c, op.arg & 1);
            F.opStage[opId] = 1; stage = 1; progressed = true;
        }
        const D4FLeastE& e = F.least[F.opReq[opId]];
        if (e.st == 2) {
            d4f_fence_block();
            const long long lit = src.litlen + e.delta;
            d4f_finish(op.dst, e.mOut, src.c, src.h, src.type, 1, lit);
            done(cand_key(1, lit + F.hdrBits[src.h]));
        }
        break;
    }
    case OP_TOFIXED_OPT: {   // toFixedHuffman(src).optimise() — DeflateStream.java:329-337,470-478; offered only, never a source
        if (stage == 0) {
            if (src.type == D4G_FIXED) { stage = 2; }
            else {
                if (atomicCAS((unsigned*)&F.fixdotSt, 0u, 1u) == 0u) { F.fixdotM = src.m; d4f_push(D4F_Q_FIXDOT, src.m); }
                stage = 1;
            }
            F.opReq[opId] = (uint16_t)d4f_req_pass(src.m, D4F_CODE_FIXED, 0, 1);
            F.opStage[opId] = (uint8_t)stage; progressed = true;
        }
        if (stage == 1) {
            if (F.fixdotSt != 2 || F.fixdotM != src.m) break;
            stage = 2; F.opStage[opId] = 2; progressed = true;
        }
        if (stage == 2) {
            const D4FPassE& e = F.pass[F.opReq[opId]];
            if (e.st != 2) break;
            d4f_fence_block();
            const long long lit = (src.type == D4G_FIXED ? src.litlen : F.fixdotLit) - e.saved;
            d4f_finish(op.dst, src.m, D4F_CODE_FIXED, 0, D4G_FIXED, 1, lit, 1);
            done(cand_key(1, lit));
        }
        break;
    }
    default:
        F.fallback = 1;
        break;
    }
    return progressed;
}

// ---------------------------------------------------------------------------------------
// Sweep (all waves): E0 / E1 of up to D4F_SWEEP_K codes in one pass over the block's back-reference records.
// Record r is in E1(code) when all its bytes have codes and their bits sum to no more than the back-reference's own
// bits, in E0 when they sum to less (replaceWithLiteralsIfSmaller's `prune` / plain comparison, :222-296).
// ---------------------------------------------------------------------------------------
// the sweep's LDS tables for `nk` codes (all threads; ends with a barrier)
__device__ __forceinline__ void d4f_sweep_tables(const int32_t* codes, int nk) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FSweepScr& S = F.scr.sweep;
    __syncthreads();
    for (int i = threadIdx.x; i < nk * 256; i += blockDim.x) {
        const int k = i >> 8, v = i & 255;
        const int l = G.code[codes[k]].lens[v];
        S.lc[k][v] = (uint16_t)(l ? l : D4G_NO_CODE);
    }
    for (int i = threadIdx.x; i < nk * 64; i += blockDim.x) {
        const int k = i >> 6, j = i & 63;
        const uint8_t* ln = G.code[codes[k]].lens;
        S.cl[k][j] = j < 29 ? ln[257 + j] : (j >= 32 && j < 62) ? ln[D4G_NLIT + j - 32] : 0;
    }
    if (threadIdx.x < D4F_SWEEP_K) S.neq[threadIdx.x] = 0;
    __syncthreads();
}
// one wave: the E-set words [w0, w0 + nW) (nW <= 16 consecutive words) of the table's codes (S.neq[k] is set where E0 != E1).
// The words of a group are collected in registers — lane i keeps word w0 + i of every code — and leave as one contiguous
// store per code and comparison mode instead of one 8-byte store per word.
#define D4F_SWEEP_GROUP 16
__device__ __forceinline__ void d4f_sweep_group(const int32_t* codes, int nk, int w0, int nW) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int lane = threadIdx.x & 63;
    D4FSweepScr& S = F.scr.sweep;
    const uint4* rf = c.refs + b.refStart;
    const uint32_t* Uw = (const uint32_t*)(c.U + b.uBase);
    const int nRef = (int)b.refCount;
    unsigned long long (*stage)[2][16] = S.stage[(threadIdx.x >> 6) & 7];
    uint4 nrec = make_uint4(0u, 0u, 0u, 0u);
    if (nW > 0) { int r = w0 * 64 + lane; if (r < nRef) nrec = rf[r]; }
    for (int i = 0; i < nW; i++) {
        const uint4 rec = nrec;
        nrec = make_uint4(0u, 0u, 0u, 0u);
        if (i + 1 < nW) { int r = (w0 + i + 1) * 64 + lane; if (r < nRef) nrec = rf[r]; }
        const uint32_t a = rec.x;
        const int len = ref_len(a);
        const int ls = ref_lsym(a) - 257, ds = ref_dsym(a), eb = ref_ebits(a);
        for (int k = 0; k < nk; k++) {
            const uint16_t* lc = S.lc[k];
            int e0 = 0, e1 = 0;
            if (len > 0) {
                const int cost = S.cl[k][ls] + S.cl[k][32 + ds] + eb;
                uint32_t x = rec.z;
                int t = lc[x & 255u] + lc[(x >> 8) & 255u] + lc[(x >> 16) & 255u] + (len > 3 ? lc[x >> 24] : 0);
                if (len > 4 && t <= cost) {
                    x = rec.w;
                    const int n = len - 4;
                    t += lc[x & 255u] + (n > 1 ? lc[(x >> 8) & 255u] : 0) + (n > 2 ? lc[(x >> 16) & 255u] : 0) + (n > 3 ? lc[x >> 24] : 0);
                    if (len > 8 && t <= cost) {   // the rare long walk: the rest comes from U, four bytes per step
                        D4GLitWalk lw;
                        lw_start(lw, Uw, rec.y + 8, len - 8);
                        lw.total = t;
                        while (lw.rem > 0 && lw.total <= cost) lw_step(lw, Uw, lc);
                        t = lw.total;
                    }
                }
                e1 = t <= cost;
                e0 = t < cost;
            }
            const unsigned long long b1 = d4g_ballot(e1), b0 = d4g_ballot(e0);
            if (lane == 0) { stage[k][0][i] = b0; stage[k][1][i] = b1; }
        }
    }
    d4g_wave_sync();
    for (int k = 0; k < nk; k++) {
        unsigned long long v0 = 0, v1 = 0;
        if (lane < nW) {
            v0 = stage[k][0][lane]; v1 = stage[k][1][lane];
            d4f_eset(c, b, codes[k], 0)[w0 + lane] = v0;
            d4f_eset(c, b, codes[k], 1)[w0 + lane] = v1;
        }
        if (d4g_ballot(v0 != v1) && lane == 0) S.neq[k] = 1;
    }
    d4g_wave_sync();
}
D4F_TASK void d4f_sweep() {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nq = F.qn[D4F_Q_SWEEP] < D4F_QCAP_SWEEP ? F.qn[D4F_Q_SWEEP] : D4F_QCAP_SWEEP;
    D4FSweepScr& S = F.scr.sweep;
    const int nWords = (int)b.maskWords;
    for (int base = 0; base < nq; base += D4F_SWEEP_K) {
        const int nk = nq - base < D4F_SWEEP_K ? nq - base : D4F_SWEEP_K;
        int32_t* codes = &F.misc[16];
        __syncthreads();
        if ((int)threadIdx.x < nk) codes[threadIdx.x] = F.qAll[d4f_qoff(D4F_Q_SWEEP) + base + threadIdx.x];
        d4f_sweep_tables(codes, nk);
        // groups of consecutive words, as many as keep every wave busy (whole cache lines when the block is long enough)
        const int gw = nWords >= nw * D4F_SWEEP_GROUP ? D4F_SWEEP_GROUP : nWords >= nw * 8 ? 8 : 4;
        for (int w0 = wave * gw; w0 < nWords; w0 += nw * gw) d4f_sweep_group(codes, nk, w0, nWords - w0 < gw ? nWords - w0 : gw);
        __syncthreads();
        if ((int)threadIdx.x < nk) {
            F.eEq[codes[threadIdx.x]] = S.neq[threadIdx.x] ? 0 : 1;
            F.eState[codes[threadIdx.x]] = 2;
        }
    }
    __syncthreads();
}

// code tables of one wave's scratch (all lanes of the wave call)
D4G_DEV void d4f_wave_load_code(D4FWaveScr& W, const D4FCode& cd) {
    const int lane = threadIdx.x & 63;
    for (int v = lane; v < 256; v += 64) { const int l = cd.lens[v]; W.lc[v] = (uint16_t)(l ? l : D4G_NO_CODE); }
    W.cl[lane] = lane < 29 ? cd.lens[257 + lane] : (lane >= 32 && lane < 62) ? cd.lens[D4G_NLIT + lane - 32] : 0;
    d4g_wave_sync();
}
// literal bits of one record (NO_CODE bytes counted in the high part)
D4G_DEV int d4f_rec_lit_total(const uint32_t* Uw, const uint16_t* lc, const uint4& rv, int len) {
    uint32_t x = rv.z;
    int t = lc[x & 255u] + lc[(x >> 8) & 255u] + lc[(x >> 16) & 255u] + (len > 3 ? lc[x >> 24] : 0);
    if (len > 4) {
        x = rv.w;
        const int n = len - 4;
        t += lc[x & 255u] + (n > 1 ? lc[(x >> 8) & 255u] : 0) + (n > 2 ? lc[(x >> 16) & 255u] : 0) + (n > 3 ? lc[x >> 24] : 0);
        if (len > 8) {
            D4GLitWalk lw;
            lw_start(lw, Uw, rv.y + 8, len - 8);
            while (lw.rem > 0) lw_step(lw, Uw, lc);
            t += lw.total;
        }
    }
    return t;
}
// histogram contribution of expanding one record: its symbols go, its bytes come
D4G_DEV void d4f_rec_to_hist(uint32_t* hist, const uint8_t* Ub, const uint4& rv) {
    const uint32_t a = rv.x;
    const int len = ref_len(a);
    atomicSub(&hist[ref_lsym(a)], 1u);
    atomicSub(&hist[D4G_NLIT + ref_dsym(a)], 1u);
    uint32_t x = rv.z;
    atomicAdd(&hist[x & 255u], 1u); atomicAdd(&hist[(x >> 8) & 255u], 1u); atomicAdd(&hist[(x >> 16) & 255u], 1u);
    if (len > 3) atomicAdd(&hist[x >> 24], 1u);
    if (len > 4) {
        x = rv.w;
        const int n = len - 4;
        atomicAdd(&hist[x & 255u], 1u);
        if (n > 1) atomicAdd(&hist[(x >> 8) & 255u], 1u);
        if (n > 2) atomicAdd(&hist[(x >> 16) & 255u], 1u);
        if (n > 3) atomicAdd(&hist[x >> 24], 1u);
        if (len > 8) for_bytes(Ub + rv.y + 8, len - 8, [&](int by) { atomicAdd(&hist[by], 1u); return true; });
    }
}

// hash of a histogram (one wave; hist[] readable by every lane)
D4G_DEV uint16_t d4f_hist_hash(const uint32_t* hist) {
    const int lane = threadIdx.x & 63;
    unsigned h = 0;
    for (int i = lane; i < D4G_HIST; i += 64) h += (hist[i] + 0x9e37u) * (2u * (unsigned)i + 1u) * 0x9E3779B1u;
    h = (unsigned)wave_sum_i32((int)h);
    return (uint16_t)(h ^ (h >> 16));
}
// A new mask (words already written to pool slot `mNew`, hashes relative to mask 0 in h1/h2, histogram in W.hist) gets
// its id: an earlier mask with the same content (hashes, then every word) is reused.  One wave; returns the id.
D4G_DEV int d4f_publish_mask(const D4GCtx& c, const D4GBlock& b, const D4FGlob& G, int mNew, unsigned long long h1, unsigned long long h2,
                             int pop, const uint32_t* hist) {
    D4FLds& F = d4fLds;
    const int lane = threadIdx.x & 63;
    const int nWords = (int)b.maskWords;
    int found = -1;
    uint32_t k = (uint32_t)(h1 >> 40) & 511u;
    const uint64_t* mine = d4f_mask(c, b, mNew);
    for (int probe = 0; probe < 512; probe++, k = (k + 1) & 511u) {
        const int e = F.maskHash[k];
        if (e == 0) break;
        const int id = e - 1;
        if (F.maskH1[id] == h1 && F.maskH2[id] == h2 && F.maskPop[id] == pop && (int)F.maskStep[id] < F.step) {
            const uint64_t* other = d4f_mask(c, b, id);
            int bad = 0;
            for (int w = lane; w < nWords; w += 64) bad |= mine[w] != other[w];
            if (!d4g_ballot(bad)) { found = id; break; }
        }
    }
    if (found >= 0) return found;
    for (int i = lane; i < D4G_HIST; i += 64) G.hist[(size_t)mNew * D4G_HIST + i] = hist[i];
    const uint16_t hh = d4f_hist_hash(hist);
    if (lane == 0) {
        F.maskHH[mNew] = hh;
        F.maskH1[mNew] = h1; F.maskH2[mNew] = h2; F.maskPop[mNew] = pop; F.maskStep[mNew] = (uint16_t)F.step;
        uint32_t kk = (uint32_t)(h1 >> 40) & 511u;
        for (int probe = 0; probe < 512; probe++, kk = (kk + 1) & 511u) {
            // (two waves may publish at once: a slot is taken through the 32-bit word that holds it)
            unsigned* w32 = (unsigned*)&F.maskHash[kk & ~1u];
            const unsigned sh = (kk & 1u) * 16;
            unsigned cur = *(volatile unsigned*)w32;
            if (((cur >> sh) & 0xffffu) != 0) continue;
            unsigned old = atomicCAS(w32, cur, cur | ((unsigned)(mNew + 1) << sh));
            if (old == cur) break;
            kk = (kk - 1) & 511u;   // the word changed under us: look at this slot again
        }
    }
    return mNew;
}

// Record walk over a selection held in registers: lane L holds the selection words L, L + 64, ... (NW of them).  Calls
// fn(record) for every selected record, 128 at a time through the wave's LDS queue: the records of a batch are requested
// together (one round trip), not one mask word after the other.
#define D4F_NWR 4   // selection words per lane in the register forms: blocks of up to 64 * 4 * 64 = 16384 back-references
template <typename Fn>
D4G_DEV void d4f_wave_for_bits(const uint64_t (&d)[D4F_NWR], const uint4* rf, uint32_t* queue /* [128] */, Fn fn) {
    const int lane = threadIdx.x & 63;
    int pc = 0;
#pragma unroll
    for (int j = 0; j < D4F_NWR; j++) pc += __popcll(d[j]);
    int incl = pc;
    for (int dd = 1; dd < 64; dd <<= 1) {
        int o = __shfl_up(incl, dd);
        if (lane >= dd) incl += o;
    }
    const int total = __shfl(incl, 63), off = incl - pc;
    for (int base = 0; base < total; base += 128) {
        if (off < base + 128 && off + pc > base) {
            int o = off;
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) {
                uint64_t x = d[j];
                while (x) {
                    const int bit = __ffsll((long long)x) - 1;
                    x &= x - 1;
                    if (o >= base && o < base + 128) queue[o - base] = (uint32_t)((lane + 64 * j) * 64 + bit);
                    o++;
                }
            }
        }
        d4g_wave_sync();
        const int n = total - base < 128 ? total - base : 128;
        const bool v0 = lane < n, v1 = lane + 64 < n;
        const uint32_t r0 = v0 ? queue[lane] : 0u, r1 = v1 ? queue[lane + 64] : 0u;
        uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = make_uint4(0u, 0u, 0u, 0u);
        if (v0) a0 = rf[r0];
        if (v1) a1 = rf[r1];
        if (v0) fn(a0);
        if (v1) fn(a1);
        d4g_wave_sync();
    }
}
// The same for a selection of any length (sel(w) = selection word w, read by the lane that owns it): 64 * D4F_NWR words a pass.
template <typename Sel, typename Fn>
D4G_DEV void d4f_wave_for_words(int nWords, int nRef, const uint4* rf, uint32_t* queue, Sel sel, Fn fn) {
    const int lane = threadIdx.x & 63;
    for (int w0 = 0; w0 < nWords; w0 += 64 * D4F_NWR) {
        uint64_t d[D4F_NWR];
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) {
            const int w = w0 + lane + 64 * j;
            uint64_t v = w < nWords ? sel(w) : 0ull;
            v &= d4g_valid_bits(w, nRef);
            d[j] = v;
        }
        d4f_wave_for_bits(d, rf + (size_t)w0 * 64, queue, fn);
    }
}
// the wave's code tables from the code's lengths, words requested by the caller beforehand: w0 = word `lane` of lens, w1 = word 64 + lane (lanes < 16)
D4G_DEV void d4f_wave_code_from_words(D4FWaveScr& W, uint32_t w0, uint32_t w1) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int l = (int)((w0 >> (8 * j)) & 255u); W.lc[4 * lane + j] = (uint16_t)(l ? l : D4G_NO_CODE); }
    W.cl[lane] = 0;
    d4g_wave_sync();
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sidx = 256 + 4 * lane + j, l = (int)((w1 >> (8 * j)) & 255u);
            if (sidx >= 257 && sidx < 286) W.cl[sidx - 257] = (uint8_t)l;
            if (sidx >= D4G_NLIT && sidx < D4G_NLIT + 30) W.cl[32 + sidx - D4G_NLIT] = (uint8_t)l;
        }
    }
    d4g_wave_sync();
}

// ---------------------------------------------------------------------------------------
// Mask update (one wave per task): replaceBackrefsWithLiteralsIfSmaller of a state (mask m, code) = expand E(code) \ m.
// leaf: only the bits saved are wanted (the state is offered as a candidate / searched for headers, nothing builds on
// its tokens); otherwise the new mask, its histogram and its id.
// ---------------------------------------------------------------------------------------
D4F_TASK void d4f_apply_task(int idx) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FWaveScr& W = F.scr.wave[(threadIdx.x >> 6) & 7];
    const int lane = threadIdx.x & 63;
    D4FPassE& e = F.pass[idx];
    const uint32_t key = e.key - 1;
    const int m = (int)(key & 255u), code = (int)((key >> 8) & 127u), prune = (int)((key >> 15) & 1u), leaf = (int)((key >> 16) & 1u);
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;
    const uint64_t* E = d4f_eset(c, b, code, prune);
    const uint64_t* M = d4f_mask(c, b, m);
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    const uint32_t* Uw = (const uint32_t*)Ub;
    if (nWords <= F.regWords) {
        // register form: everything the task reads up front goes out in one round trip (mask and E words, the histogram,
        // the code's lengths), the selected records in a second one
        uint64_t ew[D4F_NWR], mw[D4F_NWR], d[D4F_NWR];
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) {
            const int w = lane + 64 * j;
            ew[j] = w < nWords ? E[w] : 0ull;
            mw[j] = w < nWords ? M[w] : 0ull;
        }
        uint32_t hr[D4G_HIST / 64];
        if (!leaf) {
#pragma unroll
            for (int i = 0; i < D4G_HIST / 64; i++) hr[i] = G.hist[(size_t)m * D4G_HIST + lane + 64 * i];
        }
        const uint32_t* lw = (const uint32_t*)G.code[code].lens;
        const uint32_t cw0 = lw[lane], cw1 = lane < 16 ? lw[64 + lane] : 0u;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) { d[j] = ew[j] & ~mw[j]; cnt += __popcll(d[j]); }
        cnt = wave_sum_i32(cnt);
        if (cnt == 0) {
            if (lane == 0) { e.mOut = (int16_t)m; e.saved = 0; d4f_fence_block(); e.st = 2; }
            return;
        }
        int mNew = m;
        if (!leaf) {
            if (lane == 0) { int id = atomicAdd(&F.nMask, 1); if (id >= D4F_MAXM) { F.fallback = 1; id = 0; } W.misc[0] = id; }
            d4g_wave_sync();
            mNew = W.misc[0];
            d4g_wave_sync();
            if (mNew == 0) { if (lane == 0) { e.mOut = (int16_t)m; e.saved = 0; d4f_fence_block(); e.st = 2; } return; }
#pragma unroll
            for (int i = 0; i < D4G_HIST / 64; i++) W.hist[lane + 64 * i] = hr[i];
        }
        d4f_wave_code_from_words(W, cw0, cw1);
        int savedLane = 0, bad = 0;
        d4f_wave_for_bits(d, rf, W.queue, [&](const uint4& rv) {
            const uint32_t a = rv.x;
            const int cost = W.cl[ref_lsym(a) - 257] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
            const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
            const int gain = cost - total;
            if (gain < (prune ? 0 : 1)) bad = 1;
            savedLane += gain;
            if (!leaf) d4f_rec_to_hist(W.hist, Ub, rv);
        });
        const int saved = wave_sum_i32(savedLane);
        if (d4g_ballot(bad) && lane == 0) atomicAdd(c.errors, 1);
        int mOut = m;
        if (!leaf) {
            uint64_t* O = d4f_mask(c, b, mNew);
            unsigned long long h1 = 0, h2 = 0;
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) {
                const int w = lane + 64 * j;
                if (w < nWords) {
                    const uint64_t o = mw[j], n = o | ew[j];
                    O[w] = n;
                    if (n != o) { h1 += d4f_mix1(w, n) - d4f_mix1(w, o); h2 += d4f_mix2(w, n) - d4f_mix2(w, o); }
                }
            }
            h1 = (unsigned long long)wave_sum_i64((long long)h1) + F.maskH1[m];
            h2 = (unsigned long long)wave_sum_i64((long long)h2) + F.maskH2[m];
            d4g_wave_sync();
            if (lane == 0) F.maskFull[mNew] = F.maskFull[m];
            mOut = d4f_publish_mask(c, b, G, mNew, h1, h2, F.maskPop[m] + cnt, W.hist);
        }
        if (lane == 0) { e.mOut = (int16_t)mOut; e.saved = saved; d4f_fence_block(); e.st = 2; }
        return;
    }
    int cnt = 0;
    for (int w = lane; w < nWords; w += 64) cnt += __popcll(E[w] & ~M[w]);
    cnt = wave_sum_i32(cnt);
    if (cnt == 0) {
        if (lane == 0) { e.mOut = (int16_t)m; e.saved = 0; d4f_fence_block(); e.st = 2; }
        return;
    }
    int mNew = m;
    if (!leaf) {
        if (lane == 0) { int id = atomicAdd(&F.nMask, 1); if (id >= D4F_MAXM) { F.fallback = 1; id = 0; } W.misc[0] = id; }
        d4g_wave_sync();
        mNew = W.misc[0];
        d4g_wave_sync();
        if (mNew == 0) { if (lane == 0) { e.mOut = (int16_t)m; e.saved = 0; d4f_fence_block(); e.st = 2; } return; }
        for (int i = lane; i < D4G_HIST; i += 64) W.hist[i] = G.hist[(size_t)m * D4G_HIST + i];
    }
    d4f_wave_load_code(W, G.code[code]);
    int savedLane = 0, bad = 0;
    d4f_wave_for_words(nWords, nRef, rf, W.queue, [&](int w) { return E[w] & ~M[w]; },
                      [&](const uint4& rv) {
                          const uint32_t a = rv.x;
                          const int cost = W.cl[ref_lsym(a) - 257] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                          const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                          const int gain = cost - total;
                          if (gain < (prune ? 0 : 1)) bad = 1;
                          savedLane += gain;
                          if (!leaf) d4f_rec_to_hist(W.hist, Ub, rv);
                      });
    const int saved = wave_sum_i32(savedLane);
    if (d4g_ballot(bad) && lane == 0) atomicAdd(c.errors, 1);
    int mOut = m;
    if (!leaf) {
        uint64_t* O = d4f_mask(c, b, mNew);
        unsigned long long h1 = 0, h2 = 0;
        for (int w = lane; w < nWords; w += 64) {
            const uint64_t o = M[w], n = o | E[w];
            O[w] = n;
            if (n != o) { h1 += d4f_mix1(w, n) - d4f_mix1(w, o); h2 += d4f_mix2(w, n) - d4f_mix2(w, o); }
        }
        h1 = (unsigned long long)wave_sum_i64((long long)h1) + F.maskH1[m];
        h2 = (unsigned long long)wave_sum_i64((long long)h2) + F.maskH2[m];
        d4g_wave_sync();
        if (lane == 0) F.maskFull[mNew] = F.maskFull[m];
        mOut = d4f_publish_mask(c, b, G, mNew, h1, h2, F.maskPop[m] + cnt, W.hist);
    }
    if (lane == 0) { e.mOut = (int16_t)mOut; e.saved = saved; d4f_fence_block(); e.st = 2; }
}

// ---------------------------------------------------------------------------------------
// Per-code sums of removeDistLitLeastExpensive (DeflateBlockHuffman.java:373-458) over ALL records of each length
// symbol, from the block's static statistics (k_block_bins): literal minus back-reference bits, bytes without a code.
// All waves; one wave per (code, length symbol).
// ---------------------------------------------------------------------------------------
D4F_TASK void d4f_binbase() {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nq = F.qn[D4F_Q_BINBASE] < D4F_QCAP_BINBASE ? F.qn[D4F_Q_BINBASE] : D4F_QCAP_BINBASE;
    const uint32_t* stat = c.binStat + b.binStat;
    for (int t = wave; t < nq * D4G_NBINS; t += nw) {
        const int code = F.qAll[d4f_qoff(D4F_Q_BINBASE) + t / D4G_NBINS], bin = t % D4G_NBINS;
        const uint8_t* ln = G.code[code].lens;
        const uint32_t* row = stat + bin * D4G_BINSTRIDE;
        int lit = 0, z = 0, cost = 0;
        for (int i = lane; i < D4G_BINSTRIDE; i += 64) {
            const int cnt = (int)row[i];
            if (i < 256) { const int l = ln[i]; lit += cnt * l; z += l ? 0 : cnt; }
            else if (i < D4G_BIN_COUNT) cost += cnt * ln[D4G_NLIT + i - D4G_BIN_DIST];
            else if (i == D4G_BIN_COUNT) cost += cnt * ln[257 + bin];
            else if (i == D4G_BIN_EBITS) cost += cnt;
        }
        lit = wave_sum_i32(lit); z = wave_sum_i32(z); cost = wave_sum_i32(cost);
        if (lane == 0) {
            G.binBase[(size_t)code * 64 + bin] = lit - cost;
            G.binBase[(size_t)code * 64 + 32 + bin] = z;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq; i += blockDim.x) F.bbState[F.qAll[d4f_qoff(D4F_Q_BINBASE) + i]] = 2;
    __syncthreads();
}

// removeDistLitLeastExpensive of a state (mask m, code), one wave — see wg_least (d4g_ops.h) for the method
D4F_TASK void d4f_least_task(int idx) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FWaveScr& W = F.scr.wave[(threadIdx.x >> 6) & 7];
    const int lane = threadIdx.x & 63;
    D4FLeastE& e = F.least[idx];
    const uint32_t key = e.key - 1;
    const int m = (int)(key & 255u), code = (int)((key >> 8) & 127u), mode = (int)((key >> 15) & 1u);
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;
    const uint64_t* M = d4f_mask(c, b, m);
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    const uint32_t* Uw = (const uint32_t*)Ub;
    const uint32_t* stat = c.binStat + b.binStat;
    const uint64_t* bmask = c.binMask + b.binMask;
    if (nWords <= F.regWords) {
        // register form (see d4f_apply_task).  Length symbols whose records are ALL expanded already (an earlier pruning
        // step took them) have no unexpanded record, so they are never chosen whatever their sums: their records — often
        // thousands — are left out of the walk.
        const uint32_t full = F.maskFull[m];
        uint64_t mw[D4F_NWR], fw[D4F_NWR], d[D4F_NWR];
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) {
            const int w = lane + 64 * j;
            mw[j] = w < nWords ? M[w] : 0ull;
            fw[j] = 0ull;
        }
        for (uint32_t fb = full; fb; fb &= fb - 1) {
            const uint64_t* fm = bmask + (long long)(__ffs((int)fb) - 1) * nWords;
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) { const int w = lane + 64 * j; if (w < nWords) fw[j] |= fm[w]; }
        }
        const uint32_t* lw = (const uint32_t*)G.code[code].lens;
        const uint32_t cw0 = lw[lane], cw1 = lane < 16 ? lw[64 + lane] : 0u;
        int bSize = 0, bZ = 0, bCount = 0;
        if (lane < D4G_NBINS) {
            bSize = G.binBase[(size_t)code * 64 + lane];
            bZ = G.binBase[(size_t)code * 64 + 32 + lane];
            bCount = (int)stat[lane * D4G_BINSTRIDE + D4G_BIN_COUNT];
        }
        const bool isFull = lane < D4G_NBINS && ((full >> lane) & 1u);
        const int popFull = wave_sum_i32(isFull ? bCount : 0);
        const bool viaExpanded = F.maskPop[m] - popFull <= nRef - F.maskPop[m];   // walk the smaller side
        if (lane < 32) {
            W.binSize[lane] = viaExpanded ? bSize : 0;
            W.binZ[lane] = viaExpanded ? bZ : 0;
            W.binFreq[lane] = viaExpanded && !isFull ? bCount : 0;
        }
        d4f_wave_code_from_words(W, cw0, cw1);
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) {
            const int w = lane + 64 * j;
            const uint64_t valid = w < nWords ? d4g_valid_bits(w, nRef) : 0ull;
            d[j] = viaExpanded ? (mw[j] & ~fw[j]) : (~mw[j] & valid);
        }
        if (viaExpanded) {
            d4f_wave_for_bits(d, rf, W.queue, [&](const uint4& rv) {
                const uint32_t a = rv.x;
                const int bin = ref_lsym(a) - 257;
                const int cost = W.cl[bin] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                atomicSub(&W.binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                atomicSub(&W.binFreq[bin], 1);
                if (total >= D4G_NO_CODE) atomicSub(&W.binZ[bin], total >> 14);
            });
        } else {
            d4f_wave_for_bits(d, rf, W.queue, [&](const uint4& rv) {
                const uint32_t a = rv.x;
                const int bin = ref_lsym(a) - 257;
                const int cost = W.cl[bin] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                atomicAdd(&W.binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                atomicAdd(&W.binFreq[bin], 1);
                if (total >= D4G_NO_CODE) atomicAdd(&W.binZ[bin], total >> 14);
            });
        }
        d4g_wave_sync();
        if (lane == 0) {
            int rem = -1, remSize = 0, remFreq = 0;
            for (int i = 0; i < D4G_NBINS; i++) {
                // seen: the bin has an unexpanded record; allowed: none of them holds a byte without a code
                if (W.binFreq[i] > 0 && W.binZ[i] == 0) {
                    bool doRem = mode == 1 ? W.binFreq[i] < remFreq : W.binSize[i] < remSize;
                    if (rem == -1 || doRem) { rem = i; remSize = W.binSize[i]; remFreq = W.binFreq[i]; }
                }
            }
            W.misc[1] = rem; W.misc[2] = remSize;
            int id = 0;
            if (rem >= 0) { id = atomicAdd(&F.nMask, 1); if (id >= D4F_MAXM) { F.fallback = 1; id = 0; } }
            W.misc[0] = id;
        }
        d4g_wave_sync();
        const int rem = W.misc[1], remSize = W.misc[2], mNew = W.misc[0];
        d4g_wave_sync();
        if (rem < 0 || mNew == 0) {
            if (lane == 0) { e.mOut = (int16_t)m; e.delta = 0; d4f_fence_block(); e.st = 2; }
            return;
        }
        // expand the bin: new mask = old | bin mask; the histogram loses the bin's unexpanded records' symbols and gains their bytes
        const uint64_t* bm = bmask + (long long)rem * nWords;
        uint64_t bw[D4F_NWR];
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) { const int w = lane + 64 * j; bw[j] = w < nWords ? bm[w] : 0ull; }
        const uint32_t* row = stat + rem * D4G_BINSTRIDE;
        uint32_t hr[D4G_HIST / 64], rr[D4G_HIST / 64];
#pragma unroll
        for (int i = 0; i < D4G_HIST / 64; i++) { hr[i] = G.hist[(size_t)m * D4G_HIST + lane + 64 * i]; rr[i] = row[lane + 64 * i]; }
#pragma unroll
        for (int i = 0; i < D4G_HIST / 64; i++) { W.hist[lane + 64 * i] = hr[i]; W.delta[lane + 64 * i] = 0; }
        d4g_wave_sync();
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) d[j] = mw[j] & bw[j];
        // what the already expanded records of the bin contributed to the static row (they were moved earlier)
        d4f_wave_for_bits(d, rf, W.queue, [&](const uint4& rv) {
            const uint32_t a = rv.x;
            atomicAdd(&W.delta[D4G_BIN_DIST + ref_dsym(a)], 1);
            atomicAdd(&W.delta[D4G_BIN_COUNT], 1);
            for_bytes(Ub + rv.y, ref_len(a), [&](int by) { atomicAdd(&W.delta[by], 1); return true; });
        });
        d4g_wave_sync();
#pragma unroll
        for (int k = 0; k < D4G_HIST / 64; k++) {
            const int i = lane + 64 * k;
            if (i <= D4G_BIN_COUNT) {
                const int moved = (int)rr[k] - W.delta[i];
                if (moved) {
                    if (i < 256) atomicAdd(&W.hist[i], (unsigned)moved);
                    else if (i < D4G_BIN_COUNT) atomicSub(&W.hist[D4G_NLIT + i - D4G_BIN_DIST], (unsigned)moved);
                    else atomicSub(&W.hist[257 + rem], (unsigned)moved);
                }
            }
        }
        d4g_wave_sync();
        uint64_t* O = d4f_mask(c, b, mNew);
        unsigned long long h1 = 0, h2 = 0;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < D4F_NWR; j++) {
            const int w = lane + 64 * j;
            if (w < nWords) {
                const uint64_t o = mw[j], n = o | bw[j];
                O[w] = n;
                cnt += __popcll(n & ~o);
                if (n != o) { h1 += d4f_mix1(w, n) - d4f_mix1(w, o); h2 += d4f_mix2(w, n) - d4f_mix2(w, o); }
            }
        }
        cnt = wave_sum_i32(cnt);
        h1 = (unsigned long long)wave_sum_i64((long long)h1) + F.maskH1[m];
        h2 = (unsigned long long)wave_sum_i64((long long)h2) + F.maskH2[m];
        d4g_wave_sync();
        if (lane == 0) F.maskFull[mNew] = full | (1u << rem);
        const int mOut = d4f_publish_mask(c, b, G, mNew, h1, h2, F.maskPop[m] + cnt, W.hist);
        if (lane == 0) { e.mOut = (int16_t)mOut; e.delta = remSize; d4f_fence_block(); e.st = 2; }
        return;
    }
    d4f_wave_load_code(W, G.code[code]);
    const bool viaExpanded = 2 * F.maskPop[m] <= nRef;   // walk the smaller side (maskPop counts every expanded record: mask 0's included)
    if (lane < 32) {
        if (viaExpanded && lane < D4G_NBINS) {
            W.binSize[lane] = G.binBase[(size_t)code * 64 + lane];
            W.binZ[lane] = G.binBase[(size_t)code * 64 + 32 + lane];
            W.binFreq[lane] = (int)stat[lane * D4G_BINSTRIDE + D4G_BIN_COUNT];
        } else { W.binSize[lane] = 0; W.binZ[lane] = 0; W.binFreq[lane] = 0; }
    }
    d4g_wave_sync();
    if (viaExpanded) {
        d4f_wave_for_words(nWords, nRef, rf, W.queue, [&](int w) { return M[w]; },
                      [&](const uint4& rv) {
                              const uint32_t a = rv.x;
                              const int bin = ref_lsym(a) - 257;
                              const int cost = W.cl[bin] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                              const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                              atomicSub(&W.binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                              atomicSub(&W.binFreq[bin], 1);
                              if (total >= D4G_NO_CODE) atomicSub(&W.binZ[bin], total >> 14);
                          });
    } else {
        d4f_wave_for_words(nWords, nRef, rf, W.queue, [&](int w) { return ~M[w]; },
                      [&](const uint4& rv) {
                              const uint32_t a = rv.x;
                              const int bin = ref_lsym(a) - 257;
                              const int cost = W.cl[bin] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                              const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                              atomicAdd(&W.binSize[bin], (total & (D4G_NO_CODE - 1)) - cost);
                              atomicAdd(&W.binFreq[bin], 1);
                              if (total >= D4G_NO_CODE) atomicAdd(&W.binZ[bin], total >> 14);
                          });
    }
    d4g_wave_sync();
    if (lane == 0) {
        int rem = -1, remSize = 0, remFreq = 0;
        for (int i = 0; i < D4G_NBINS; i++) {
            // seen: the bin has an unexpanded record; allowed: none of them holds a byte without a code
            if (W.binFreq[i] > 0 && W.binZ[i] == 0) {
                bool doRem = mode == 1 ? W.binFreq[i] < remFreq : W.binSize[i] < remSize;
                if (rem == -1 || doRem) { rem = i; remSize = W.binSize[i]; remFreq = W.binFreq[i]; }
            }
        }
        W.misc[1] = rem; W.misc[2] = remSize;
        int id = 0;
        if (rem >= 0) { id = atomicAdd(&F.nMask, 1); if (id >= D4F_MAXM) { F.fallback = 1; id = 0; } }
        W.misc[0] = id;
    }
    d4g_wave_sync();
    const int rem = W.misc[1], remSize = W.misc[2], mNew = W.misc[0];
    d4g_wave_sync();
    if (rem < 0 || mNew == 0) {
        if (lane == 0) { e.mOut = (int16_t)m; e.delta = 0; d4f_fence_block(); e.st = 2; }
        return;
    }
    // expand the bin: new mask = old | bin mask; the histogram loses the bin's unexpanded records' symbols and gains their bytes
    const uint64_t* bm = bmask + (long long)rem * nWords;
    for (int i = lane; i < D4G_HIST; i += 64) { W.hist[i] = G.hist[(size_t)m * D4G_HIST + i]; W.delta[i] = 0; }
    d4g_wave_sync();
    // what the already expanded records of the bin contributed to the static row (they were moved earlier)
    d4f_wave_for_words(nWords, nRef, rf, W.queue, [&](int w) { return M[w] & bm[w]; },
                      [&](const uint4& rv) {
                          const uint32_t a = rv.x;
                          atomicAdd(&W.delta[D4G_BIN_DIST + ref_dsym(a)], 1);
                          atomicAdd(&W.delta[D4G_BIN_COUNT], 1);
                          for_bytes(Ub + rv.y, ref_len(a), [&](int by) { atomicAdd(&W.delta[by], 1); return true; });
                      });
    d4g_wave_sync();
    {
        const uint32_t* row = stat + rem * D4G_BINSTRIDE;
        for (int i = lane; i <= D4G_BIN_COUNT; i += 64) {
            const int moved = (int)row[i] - W.delta[i];
            if (!moved) continue;
            if (i < 256) atomicAdd(&W.hist[i], (unsigned)moved);
            else if (i < D4G_BIN_COUNT) atomicSub(&W.hist[D4G_NLIT + i - D4G_BIN_DIST], (unsigned)moved);
            else atomicSub(&W.hist[257 + rem], (unsigned)moved);
        }
    }
    d4g_wave_sync();
    uint64_t* O = d4f_mask(c, b, mNew);
    unsigned long long h1 = 0, h2 = 0;
    int cnt = 0;
    for (int w = lane; w < nWords; w += 64) {
        const uint64_t o = M[w], n = o | bm[w];
        O[w] = n;
        cnt += __popcll(n & ~o);
        if (n != o) { h1 += d4f_mix1(w, n) - d4f_mix1(w, o); h2 += d4f_mix2(w, n) - d4f_mix2(w, o); }
    }
    cnt = wave_sum_i32(cnt);
    h1 = (unsigned long long)wave_sum_i64((long long)h1) + F.maskH1[m];
    h2 = (unsigned long long)wave_sum_i64((long long)h2) + F.maskH2[m];
    d4g_wave_sync();
    if (lane == 0) F.maskFull[mNew] = F.maskFull[m] | (1u << rem);
    const int mOut = d4f_publish_mask(c, b, G, mNew, h1, h2, F.maskPop[m] + cnt, W.hist);
    if (lane == 0) { e.mOut = (int16_t)mOut; e.delta = remSize; d4f_fence_block(); e.st = 2; }
}

// ---------------------------------------------------------------------------------------
// Header pieces by one wave (the workgroup forms are in d4g_ops.h)
// ---------------------------------------------------------------------------------------
// code-length code of clFreq — Huffman.ofRLEPacked, B/huffman/Huffman.java:117-134
D4G_DEV int d4f_wave_cl_tree(unsigned char* treeMem, const uint32_t* clFreq, uint8_t* clLen) {
    TreeMem<uint32_t, uint8_t, 20> tm;
    tm.carve(treeMem, 1);
    const int lane = threadIdx.x & 63;
    if (lane < 19) clLen[lane] = 0;
    d4g_wave_sync();
    int err = d4f_wave_tree<1>(tm, 19, 7, [&](int i) { return (unsigned)clFreq[i]; }, [&](int v, int len) { clLen[v] = (uint8_t)len; });
    d4g_wave_sync();
    return err;
}
// trailing zero code-length-code lengths in transmit order — removeDynHeaderTrailingZeroLenCodelens, :335-364
D4G_DEV int d4f_wave_trim(const uint8_t* clLen, int nCl) {
    const int lane = threadIdx.x & 63;
    const bool nz = lane < nCl && lane < 19 && clLen[D4G_CL_ORDER[lane]] != 0;
    const unsigned long long m = d4g_ballot(nz);
    return m ? 64 - __clzll((long long)m) : nCl;
}
// rewriteHeader with the default flags — DeflateBlockHuffman.java:484-577 (wg_rewrite_header's wave part): pairs,
// code-length code, size; nCl trimmed.  lens = litLen[288] + distLen[32].
__device__ __forceinline__ int d4f_wave_default_header(const uint8_t* lens, int nLit, int nDist, uint16_t* pairs, uint32_t* clFreq, uint8_t* clLen,
                                                        unsigned char* treeMem, int& nPairsOut, int& nClOut, int& bitsOut) {
    const int lane = threadIdx.x & 63;
    const int flags = F_DEFAULT;
    if (lane < 20) clFreq[lane] = 0;
    d4g_wave_sync();
    const int n = nLit + nDist;
    auto len = [&](int i) { return i < nLit ? (int)lens[i] : (int)lens[D4G_NLIT + i - nLit]; };
    constexpr int NCH = (D4G_NLIT + D4G_NDIST) / 64;
    unsigned long long sm[NCH];
    int v[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        int i = ch * 64 + lane;
        v[ch] = i < n ? len(i) : -1;
        int pv = (i > 0 && i < n) ? len(i - 1) : -2;
        sm[ch] = d4g_ballot(i < n && v[ch] != pv);
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
                pairs[off++] = pair_encode(sym, r, value);
                atomicAdd(&clFreq[sym], 1u);
            });
        base += __shfl(incl, 63);
    }
    d4g_wave_sync();
    const int err = d4f_wave_cl_tree(treeMem, clFreq, clLen);
    int hbl = 0;
    if (lane < 19) hbl = (int)clFreq[lane] * (clLen[lane] + (lane >= 16 ? pair_extra_bits(lane) : 0));
    int hb = 5 + 5 + 4 + 19 * 3 + wave_sum_i32(hbl);
    const int nCl = d4f_wave_trim(clLen, 19);
    hb -= 3 * (19 - nCl);
    nPairsOut = base; nClOut = nCl; bitsOut = hb;
    return err;
}
// replaceRLERunsWithLiteralsIfSmaller — :321-332: returns the bits saved
D4G_DEV int d4f_wave_replace_runs(uint16_t* pairs, int nPairs, const uint8_t* clLen, bool prune) {
    const int lane = threadIdx.x & 63;
    int saved = 0;
    for (int i = lane; i < nPairs; i += 64) {
        uint16_t p = pairs[i];
        if ((p & 31) >= 16 && !(p & D4G_PAIR_EXPANDED)) {
            int sym, run, value;
            pair_decode(p, sym, run, value);
            int g = pair_replace_gain(sym, run, value, prune, [&](int s) { return (int)clLen[s]; });
            if (g >= 0) { pairs[i] = p | D4G_PAIR_EXPANDED; saved += g; }
        }
    }
    saved = wave_sum_i32(saved);
    d4g_wave_sync();
    return saved;
}
// recodeHeader — :579-629 (numCodelenLens deliberately not reset): new code-length code for the pairs as they stand
D4G_DEV int d4f_wave_recode_header(D4FHdrScr& H) {
    const int lane = threadIdx.x & 63;
    if (lane < 20) H.clFreq[lane] = 0;
    d4g_wave_sync();
    const int np = H.nPairs;
    for (int i = lane; i < np; i += 64) {
        int sym, run, value;
        uint16_t p = H.pairs[i];
        pair_decode(p, sym, run, value);
        if (p & D4G_PAIR_EXPANDED) atomicAdd(&H.clFreq[value], (unsigned)run);
        else atomicAdd(&H.clFreq[sym], 1u);
    }
    d4g_wave_sync();
    const int err = d4f_wave_cl_tree(H.tree, H.clFreq, H.clLen);
    const int nCl = d4f_wave_trim(H.clLen, H.nCl);
    int hbl = 0;
    for (int i = lane; i < np; i += 64) {
        int sym, run, value;
        uint16_t p = H.pairs[i];
        pair_decode(p, sym, run, value);
        if (p & D4G_PAIR_EXPANDED) hbl += run * H.clLen[value];
        else hbl += H.clLen[sym] + (sym >= 16 ? pair_extra_bits(sym) : 0);
    }
    const int hb = 5 + 5 + 4 + 3 * nCl + wave_sum_i32(hbl);
    d4g_wave_sync();
    if (lane == 0) { H.nCl = nCl; H.bits = hb; }
    d4g_wave_sync();
    return err;
}

// one header operation (one wave): optimiseHeader (:471-476), recodeHeader (:579-629), recodeHeaderToLessRLEMatches (:632-635)
D4F_TASK void d4f_hdr_task(int idx) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FHdrScr& H = *(D4FHdrScr*)&F.scr.wave[(threadIdx.x >> 6) & 7];
    const int lane = threadIdx.x & 63;
    D4FHdrE& e = F.hdrReq[idx];
    const uint32_t key = e.key - 1;
    const int h = (int)(key & 0xfffu), op = (int)(key >> 12);
    const D4FHdr& src = G.hdr[h];
    const D4FPairs& pb = G.pairs[src.base];
    const int nPairs = pb.nPairs;
    for (int i = lane; i < nPairs; i += 64) H.pairs[i] = (uint16_t)(pb.pairs[i] | (((src.flags[i >> 5] >> (i & 31)) & 1u) ? D4G_PAIR_EXPANDED : 0u));
    if (lane < 32) H.clLen[lane] = src.clLen[lane];
    if (lane == 0) { H.nCl = src.nCl; H.nPairs = nPairs; H.bits = src.bits; }
    d4g_wave_sync();
    int err = 0;
    if (op == D4F_H_OPT) {
        const int n = d4f_wave_trim(H.clLen, H.nCl);
        const int saved = d4f_wave_replace_runs(H.pairs, nPairs, H.clLen, false);
        if (lane == 0) { H.bits -= 3 * (H.nCl - n) + saved; H.nCl = n; }
    } else if (op == D4F_H_POST) {
        err = d4f_wave_recode_header(H);
    } else {
        (void)d4f_wave_replace_runs(H.pairs, nPairs, H.clLen, true);   // (the bits are summed again by recodeHeader)
        err = d4f_wave_recode_header(H);
    }
    d4g_wave_sync();
    if (err && lane == 0) atomicAdd(c.errors, 1);
    if (lane == 0) { int id = atomicAdd(&F.nHdr, 1); if (id >= D4F_MAXH) { F.fallback = 1; id = 0; } H.pad = id; }
    d4g_wave_sync();
    const int hNew = H.pad;
    D4FHdr& dst = G.hdr[hNew];
    if (hNew != 0) {
        if (lane < 32) dst.clLen[lane] = H.clLen[lane];
        if (lane < 10) {
            uint32_t fl = 0;
            for (int j = 0; j < 32; j++) { const int i = lane * 32 + j; if (i < nPairs && (H.pairs[i] & D4G_PAIR_EXPANDED)) fl |= 1u << j; }
            dst.flags[lane] = fl;
        }
        if (lane == 0) { dst.nCl = H.nCl; dst.bits = H.bits; dst.base = src.base; F.hdrBits[hNew] = H.bits; }
    }
    d4g_wave_sync();
    if (lane == 0) { e.hOut = (int16_t)hNew; d4f_fence_block(); e.st = 2; }
}

// ---------------------------------------------------------------------------------------
// Huffman rebuild of a mask's histogram (one wave): recodeHuffman — DeflateBlockHuffman.java:670-743 — into the task's
// scratch; ids are given afterwards by d4f_tree_publish.
// ---------------------------------------------------------------------------------------
D4F_TASK void d4f_tree_lit(int slotIdx, int m) {      // the slot's first wave: literal/length tree
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FTreeScr& T = F.scr.tree[slotIdx];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < D4G_NLIT; i += 64) T.hist[i] = G.hist[(size_t)m * D4G_HIST + i];
    for (int i = lane; i < D4G_NLIT / 4; i += 64) ((uint32_t*)T.lens)[i] = 0;
    d4g_wave_sync();
    int ml = 0;
    for (int i = lane; i < 286; i += 64) if (T.hist[i]) ml = i + 1 > ml ? i + 1 : ml;
    const int lastLit = wave_max_i32(ml);
    D4FLitTree tm;
    tm.carve(T.tree, 1);
    const int err = d4f_wave_tree<(D4G_NLIT + 63) / 64>(tm, lastLit, 15, [&](int i) { return T.hist[i]; }, [&](int v, int len) { T.lens[v] = (uint8_t)len; });
    if (lane == 0) { T.nLit = lastLit; T.err = err; T.m = m; }
    d4g_wave_sync();
}
D4F_TASK void d4f_tree_dist(int slotIdx, int m) {     // the slot's second wave: distance tree
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FTreeScr& T = F.scr.tree[slotIdx];
    const int lane = threadIdx.x & 63;
    if (lane < D4G_NDIST) { T.hist[D4G_NLIT + lane] = G.hist[(size_t)m * D4G_HIST + D4G_NLIT + lane]; T.lens[D4G_NLIT + lane] = 0; }
    d4g_wave_sync();
    int md = 0;
    if (lane < 30 && T.hist[D4G_NLIT + lane]) md = lane + 1;
    const int lastDist = wave_max_i32(md);
    const bool used = lane < lastDist && T.hist[D4G_NLIT + lane] != 0;
    const int nz = __popcll(d4g_ballot(used));
    int nDist, err = 0;
    if (lastDist == 0) nDist = 1;                                  // handleZero: new HuffmanTable(1)
    else if (nz <= 1) { nDist = lastDist; if (lane == 0) T.lens[D4G_NLIT + lastDist - 1] = 1; }   // handleOne: one used distance code, length 1
    else {
        D4FDistTree tm;
        tm.carve(T.treeD, 1);
        err = d4f_wave_tree<1>(tm, lastDist, 15, [&](int i) { return T.hist[D4G_NLIT + i]; }, [&](int v, int len) { T.lens[D4G_NLIT + v] = (uint8_t)len; });
        nDist = lastDist;
    }
    if (lane == 0) { T.nDist = nDist; T.pad = err; }
    d4g_wave_sync();
}
D4F_TASK void d4f_tree_header(int slotIdx) {          // wave A again, once both trees stand: token bits and the default header
    D4FLds& F = d4fLds;
    D4FTreeScr& T = F.scr.tree[slotIdx];
    const int lane = threadIdx.x & 63;
    // Σ token bits from the histogram — recodeToHuffmanInternal, :759-770
    long long v = 0;
    for (int i = lane; i < D4G_HIST; i += 64) {
        const unsigned h = T.hist[i];
        if (h) {
            if (i < D4G_NLIT) v += (long long)h * (T.lens[i] + (i >= 257 ? d4g_lsym_ebits(i) : 0));
            else v += (long long)h * (T.lens[i] + d4g_dsym_ebits(i - D4G_NLIT));
        }
    }
    v = wave_sum_i64(v);
    int nPairs, nCl, bits;
    const int err = d4f_wave_default_header(T.lens, T.nLit, T.nDist, T.pairs, T.clFreq, T.clLen, T.tree, nPairs, nCl, bits);
    if (lane == 0) { T.nCl = nCl; T.nPairs = nPairs; T.hdrBits = bits; T.err |= err | T.pad; T.litlen = v; }
    d4g_wave_sync();
}
// ids of a finished rebuild (one wave, tasks one after the other): an earlier code with the same lengths is reused
D4F_TASK void d4f_tree_publish(int slotIdx) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FTreeScr& T = F.scr.tree[slotIdx];
    const int lane = threadIdx.x & 63;
    unsigned long long h = 0;
    const uint32_t* lw = (const uint32_t*)T.lens;
    for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) h += d4f_mix1(i, lw[i]);
    h = (unsigned long long)wave_sum_i64((long long)h) + (unsigned long long)T.nLit * 0x100000001b3ULL + ((unsigned long long)T.nDist << 48);
    if (h == 0) h = 1;
    int found = -1;
    uint32_t k = (uint32_t)(h >> 40) & 255u;
    for (int probe = 0; probe < 256; probe++, k = (k + 1) & 255u) {
        const int e = F.codeHash[k];
        if (e == 0) break;
        const int id = e - 1;
        if (F.codeH[id] == h) {
            const D4FCode& cd = G.code[id];
            int bad = 0;
            for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) bad |= ((const uint32_t*)cd.lens)[i] != lw[i];
            if (lane == 0) bad |= cd.nLit != T.nLit || cd.nDist != T.nDist || cd.type != D4G_DYNAMIC;
            if (!d4g_ballot(bad)) { found = id; break; }
        }
    }
    int code = found;
    const int nCodeNow = __shfl(F.nCode, 0);   // (every lane reads before lane 0 writes)
    if (code < 0) {
        code = nCodeNow;
        if (code >= D4F_MAXC) { if (lane == 0) F.fallback = 1; code = 0; }
        else {
            D4FCode& cd = G.code[code];
            for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) ((uint32_t*)cd.lens)[i] = lw[i];
            if (lane == 0) {
                cd.nLit = T.nLit; cd.nDist = T.nDist; cd.type = D4G_DYNAMIC; cd.err = T.err;
                F.nCode = code + 1;
                F.codeH[code] = h;
                F.defHdr[code] = -1;
                F.eState[code] = 0; F.eEq[code] = 0; F.bbState[code] = 0; F.hsState[code] = 0;
                F.codeHash[k] = (uint8_t)(code + 1);   // k: the first empty slot of the probe above
            }
        }
    }
    d4g_wave_sync();
    if (T.err && lane == 0) atomicAdd(c.errors, 1);
    const int defNow = __shfl((int)F.defHdr[code], 0);
    const int nHdrNow = __shfl(F.nHdr, 0);
    if (defNow < 0) {   // the code's default header (rewriteHeader()) — a function of the lengths
        int hid = nHdrNow;
        if (hid >= D4F_MAXH) { if (lane == 0) F.fallback = 1; hid = 0; }
        else {
            D4FPairs& pb = G.pairs[1 + code];
            for (int i = lane; i < T.nPairs; i += 64) pb.pairs[i] = T.pairs[i];
            D4FHdr& hd = G.hdr[hid];
            if (lane < 32) hd.clLen[lane] = lane < 19 ? T.clLen[lane] : 0;
            if (lane < 10) hd.flags[lane] = 0;
            if (lane == 0) {
                pb.nPairs = T.nPairs;
                hd.nCl = T.nCl; hd.bits = T.hdrBits; hd.base = 1 + code;
                F.hdrBits[hid] = T.hdrBits;
                F.nHdr = hid + 1;
                F.defHdr[code] = (int16_t)hid;
            }
        }
    }
    d4g_wave_sync();
    if (lane == 0) { F.treeC[T.m] = (uint8_t)code; F.treeLit[T.m] = T.litlen; d4f_fence_block(); F.treeSt[T.m] = 2; }
    d4g_wave_sync();
}

// ---------------------------------------------------------------------------------------
// Header search of one code (one wave): the 56 optimiseBlockDynBlock candidates — see d4g_exec_hdr_search
// ---------------------------------------------------------------------------------------
D4F_TASK void d4f_hs_task(int slotIdx, int code) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FHsScr& X = ((D4FHsScr*)&F.least)[slotIdx];
    const int lane = threadIdx.x & 63;
    D4GHdrLds& H = X.H;
    uint8_t* comb = X.comb;
    const D4FCode& cd = G.code[code];
    const int nLit = cd.nLit, n = nLit + cd.nDist;
    for (int i = lane; i < (D4G_NLIT + D4G_NDIST) / 4; i += 64) ((uint32_t*)H.lens)[i] = ((const uint32_t*)cd.lens)[i];
    d4g_wave_sync();
    for (int i = lane; i < D4G_NLIT + D4G_NDIST; i += 64) comb[i] = i >= n ? 0 : i < nLit ? H.lens[i] : H.lens[D4G_NLIT + i - nLit];
    if (lane < 20) H.baseFreq[lane] = 0;
    d4g_wave_sync();
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
    d4g_wave_sync();
    long long key = D4G_KEY_NONE;
    if (lane < 56) {
        long long size = d4g_hdr_candidate_body(&H, lane, c.hdrFlags[lane], c.hdrPrune[lane], 0LL);
        key = D4G_MAKE_KEY(size, lane);
    }
    key = wave_min_i64(key);
    if (lane == 0) {
        F.hsBits[code] = (int32_t)(key >> D4G_KEY_SEQ_BITS);
        F.hsLane[code] = (uint8_t)(key & 63);
        d4f_fence_block();
        F.hsState[code] = 2;
    }
    d4g_wave_sync();
}

// ---------------------------------------------------------------------------------------
// Cluster mode (see the notes at D4FClArena): the worker every workgroup runs on a posted command, and the control
// workgroup's side of the three kinds of step that scale with the block's length.
// ---------------------------------------------------------------------------------------
#ifdef D4G_HOSTSIM
D4G_DEV int d4f_atomic_ld(const int32_t* p) { return *p; }
D4G_DEV void d4f_atomic_st(int32_t* p, int v) { *p = v; }
#else
D4G_DEV int d4f_atomic_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
D4G_DEV void d4f_atomic_st(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
D4G_DEV D4FClCmd& d4f_cl_cmd() { return d4fLds.cl->slot[(d4fLds.clEpoch - 1) % D4F_CL_SLOTS]; }

// Work on the command in hand until no item is left.  All threads of a workgroup call; any workgroup of the launch may.
D4F_TASK void d4f_cl_work() {
    D4FLds& F = d4fLds;
    D4F_CTX;
    D4FClCmd& C = d4f_cl_cmd();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cmd = C.cmd, nItems = C.nItems, nChunks = C.nChunks, nk = C.nk;
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;
    if (threadIdx.x == 0) F.clDoneWg = 0;
    if (cmd == D4F_CMD_SWEEP) d4f_sweep_tables(C.codes, nk);
    else __syncthreads();
    D4FWaveScr& W = F.scr.wave[wave & 7];
    uint32_t* Wd = (uint32_t*)W.delta;
    const uint4* rf = c.refs + b.refStart;
    const uint8_t* Ub = c.U + b.uBase;
    const uint32_t* Uw = (const uint32_t*)Ub;
    const uint64_t* bmask = c.binMask + b.binMask;
    int myDone = 0, cur = -1;
    // the task in hand (wave-uniform) and this wave's share of its sums
    int tm = 0, tcode = 0, tprune = 0, tleaf = 0, tnew = 0, trem = -1, tvia = 0;
    uint32_t tfull = 0;
    int cntLane = 0, savedLane = 0, badLane = 0;
    unsigned long long h1Lane = 0, h2Lane = 0;
    auto flush = [&]() {
        if (cur < 0) return;
        D4FClTask& T = C.task[cur];
        const int cnt = wave_sum_i32(cntLane), saved = wave_sum_i32(savedLane);
        const unsigned long long h1 = (unsigned long long)wave_sum_i64((long long)h1Lane), h2 = (unsigned long long)wave_sum_i64((long long)h2Lane);
        if (lane == 0) {
            if (cnt) atomicAdd(&T.cnt, cnt);
            if (saved) atomicAdd(&T.saved, saved);
            if (h1) atomicAdd(&T.h1, h1);
            if (h2) atomicAdd(&T.h2, h2);
        }
        d4g_wave_sync();
        if (cmd == D4F_CMD_LEAST_WALK) {
            if (lane < D4G_NBINS) {
                if (W.binSize[lane]) atomicAdd(&T.bin[0][lane], W.binSize[lane]);
                if (W.binFreq[lane]) atomicAdd(&T.bin[1][lane], W.binFreq[lane]);
                if (W.binZ[lane]) atomicAdd(&T.bin[2][lane], W.binZ[lane]);
            }
        } else {
            for (int i = lane; i < D4G_HIST; i += 64) { const int v = (int)Wd[i]; if (v) atomicAdd(&T.delta[i], v); }
        }
        if (d4g_ballot(badLane) && lane == 0) atomicAdd(c.errors, 1);
        d4g_wave_sync();
    };
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&C.next, 1);
        item = __shfl(item, 0);
        if (item >= nItems) break;
        myDone++;
        if (cmd == D4F_CMD_SWEEP) {
            const int w0 = item * D4F_CL_SWEEP_WORDS;
            d4f_sweep_group(C.codes, nk, w0, w0 + D4F_CL_SWEEP_WORDS < nWords ? D4F_CL_SWEEP_WORDS : nWords - w0);
            continue;
        }
        const int t = item / nChunks, w0 = (item - t * nChunks) * D4F_CL_CHUNK;
        if (t != cur) {
            flush();
            cur = t;
            const D4FClTask& T = C.task[t];
            tm = T.m; tcode = T.code; tprune = T.prune; tleaf = T.leaf; tnew = T.mNew; trem = T.rem; tvia = T.via; tfull = T.full;
            cntLane = 0; savedLane = 0; badLane = 0; h1Lane = 0; h2Lane = 0;
            for (int i = lane; i < D4G_HIST; i += 64) Wd[i] = 0;
            if (lane < 32) { W.binSize[lane] = 0; W.binFreq[lane] = 0; W.binZ[lane] = 0; }
            if (cmd != D4F_CMD_LEAST_EXPAND) {
                const uint32_t* lw = (const uint32_t*)G.code[tcode].lens;
                d4f_wave_code_from_words(W, lw[lane], lane < 16 ? lw[64 + lane] : 0u);
            }
            d4g_wave_sync();
        }
        const uint64_t* M = d4f_mask(c, b, tm);
        uint64_t mw[D4F_NWR], xw[D4F_NWR], d[D4F_NWR];
        if (cmd == D4F_CMD_APPLY) {
            const uint64_t* E = d4f_eset(c, b, tcode, tprune);
            uint64_t* O = (!tleaf && tnew) ? d4f_mask(c, b, tnew) : nullptr;
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) {
                const int w = w0 + lane + 64 * j;
                mw[j] = w < nWords ? M[w] : 0ull;
                xw[j] = w < nWords ? E[w] : 0ull;
                d[j] = xw[j] & ~mw[j];
                cntLane += __popcll(d[j]);
                if (O && w < nWords) {
                    const uint64_t o = mw[j], n = o | xw[j];
                    O[w] = n;
                    if (n != o) { h1Lane += d4f_mix1(w, n) - d4f_mix1(w, o); h2Lane += d4f_mix2(w, n) - d4f_mix2(w, o); }
                }
            }
            d4f_wave_for_bits(d, rf + (size_t)w0 * 64, W.queue, [&](const uint4& rv) {
                const uint32_t a = rv.x;
                const int cost = W.cl[ref_lsym(a) - 257] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                const int gain = cost - total;
                if (gain < (tprune ? 0 : 1)) badLane = 1;
                savedLane += gain;
                if (!tleaf) d4f_rec_to_hist(Wd, Ub, rv);
            });
        } else if (cmd == D4F_CMD_LEAST_WALK) {
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) {
                const int w = w0 + lane + 64 * j;
                mw[j] = w < nWords ? M[w] : 0ull;
                xw[j] = 0ull;
            }
            if (tvia)
                for (uint32_t fb = tfull; fb; fb &= fb - 1) {
                    const uint64_t* fm = bmask + (long long)(__ffs((int)fb) - 1) * nWords;
#pragma unroll
                    for (int j = 0; j < D4F_NWR; j++) { const int w = w0 + lane + 64 * j; if (w < nWords) xw[j] |= fm[w]; }
                }
#pragma unroll
            for (int j = 0; j < D4F_NWR; j++) {
                const int w = w0 + lane + 64 * j;
                const uint64_t valid = w < nWords ? d4g_valid_bits(w, nRef) : 0ull;
                d[j] = tvia ? (mw[j] & ~xw[j]) : (~mw[j] & valid);
            }
            const int sgn = tvia ? -1 : 1;
            d4f_wave_for_bits(d, rf + (size_t)w0 * 64, W.queue, [&](const uint4& rv) {
                const uint32_t a = rv.x;
                const int bin = ref_lsym(a) - 257;
                const int cost = W.cl[bin] + W.cl[32 + ref_dsym(a)] + ref_ebits(a);
                const int total = d4f_rec_lit_total(Uw, W.lc, rv, ref_len(a));
                atomicAdd(&W.binSize[bin], sgn * ((total & (D4G_NO_CODE - 1)) - cost));
                atomicAdd(&W.binFreq[bin], sgn);
                if (total >= D4G_NO_CODE) atomicAdd(&W.binZ[bin], sgn * (total >> 14));
            });
        } else if (cmd == D4F_CMD_LEAST_EXPAND) {
            if (trem >= 0 && tnew) {
                const uint64_t* bm = bmask + (long long)trem * nWords;
                uint64_t* O = d4f_mask(c, b, tnew);
#pragma unroll
                for (int j = 0; j < D4F_NWR; j++) {
                    const int w = w0 + lane + 64 * j;
                    mw[j] = w < nWords ? M[w] : 0ull;
                    xw[j] = w < nWords ? bm[w] : 0ull;
                    d[j] = mw[j] & xw[j];
                    if (w < nWords) {
                        const uint64_t o = mw[j], n = o | xw[j];
                        O[w] = n;
                        cntLane += __popcll(n & ~o);
                        if (n != o) { h1Lane += d4f_mix1(w, n) - d4f_mix1(w, o); h2Lane += d4f_mix2(w, n) - d4f_mix2(w, o); }
                    }
                }
                // what the already expanded records of the bin contributed to the static row (they were moved earlier)
                d4f_wave_for_bits(d, rf + (size_t)w0 * 64, W.queue, [&](const uint4& rv) {
                    const uint32_t a = rv.x;
                    atomicAdd(&Wd[D4G_BIN_DIST + ref_dsym(a)], 1u);
                    atomicAdd(&Wd[D4G_BIN_COUNT], 1u);
                    for_bytes(Ub + rv.y, ref_len(a), [&](int by) { atomicAdd(&Wd[by], 1u); return true; });
                });
            }
        }
    }
    flush();
    __syncthreads();
    if (cmd == D4F_CMD_SWEEP && (int)threadIdx.x < nk && F.scr.sweep.neq[threadIdx.x]) atomicOr(&C.neq[threadIdx.x], 1u);
    d4g_drain_stores();
    if (lane == 0 && myDone) atomicAdd(&F.clDoneWg, myDone);
    __syncthreads();
    if (threadIdx.x == 0 && F.clDoneWg) { d4g_release_agent(); atomicAdd(&C.done, F.clDoneWg); }
    __syncthreads();
}

// control: a fresh command slot (all threads; the slot's header and `n` tasks' accumulators are cleared, written through)
D4F_TASK void d4f_cl_begin(int n) {
    D4FLds& F = d4fLds;
    __syncthreads();
    if (threadIdx.x == 0) F.clEpoch++;
    __syncthreads();
    D4FClCmd& C = d4f_cl_cmd();
    for (int i = threadIdx.x; i < (int)(offsetof(D4FClCmd, task) / 4); i += blockDim.x) st_sc1((uint32_t*)&C + i, 0u);
    const int words = (int)(sizeof(D4FClTask) / 4);
    for (int i = threadIdx.x; i < n * words; i += blockDim.x) st_sc1((uint32_t*)&C.task[0] + i, 0u);
    d4g_drain_stores();
    __syncthreads();
}
// control: post the command and work it off with whoever is there; returns when every item is done (all threads)
D4F_TASK void d4f_cl_run() {
    D4FLds& F = d4fLds;
    D4FClCmd& C = d4f_cl_cmd();
    d4g_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        d4g_release_agent();
        if (F.clEpoch <= D4F_CL_SLOTS) d4f_atomic_st(&F.cl->epoch, F.clEpoch);   // (beyond the slots: nobody is told, the control workgroup works alone)
    }
    __syncthreads();
    d4f_cl_work();
    if (threadIdx.x == 0) {
        const int nItems = C.nItems;
        long long spin = 0;   // (only items some resident workgroup has already taken are outstanding: microseconds; the bound is for a device gone wrong)
        while (d4f_atomic_ld(&C.done) < nItems && ++spin < (1LL << 25)) d4g_sleep();
        if (d4f_atomic_ld(&C.done) < nItems) { atomicAdd(F.c.errors, 1); F.fallback = 1; }
        d4g_acquire_agent();
    }
    __syncthreads();
}
D4G_DEV void d4f_cl_set(int32_t* p, int v) { st_sc1((uint32_t*)p, (uint32_t)v); }

D4F_TASK void d4f_cl_sweep() {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int nq = F.qn[D4F_Q_SWEEP] < D4F_QCAP_SWEEP ? F.qn[D4F_Q_SWEEP] : D4F_QCAP_SWEEP;
    const int nWords = (int)b.maskWords;
    for (int base = 0; base < nq; base += D4F_SWEEP_K) {
        const int nk = nq - base < D4F_SWEEP_K ? nq - base : D4F_SWEEP_K;
        d4f_cl_begin(0);
        D4FClCmd& C = d4f_cl_cmd();
        if ((int)threadIdx.x < nk) d4f_cl_set(&C.codes[threadIdx.x], F.qAll[d4f_qoff(D4F_Q_SWEEP) + base + threadIdx.x]);
        if (threadIdx.x == 0) {
            d4f_cl_set(&C.cmd, D4F_CMD_SWEEP); d4f_cl_set(&C.nk, nk); d4f_cl_set(&C.nChunks, 1);
            d4f_cl_set(&C.nItems, (nWords + D4F_CL_SWEEP_WORDS - 1) / D4F_CL_SWEEP_WORDS);
        }
        d4f_cl_run();
        if ((int)threadIdx.x < nk) {
            const int code = F.qAll[d4f_qoff(D4F_Q_SWEEP) + base + threadIdx.x];
            F.eEq[code] = C.neq[threadIdx.x] ? 0 : 1;
            F.eState[code] = 2;
        }
        __syncthreads();
    }
}

D4F_TASK void d4f_cl_apply_phase(int nq) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nWords = (int)b.maskWords;
    const int nChunks = (nWords + D4F_CL_CHUNK - 1) / D4F_CL_CHUNK;
    for (int base = 0; base < nq; base += D4F_CL_MAXT) {
        const int n = nq - base < D4F_CL_MAXT ? nq - base : D4F_CL_MAXT;
        d4f_cl_begin(n);
        D4FClCmd& C = d4f_cl_cmd();
        if ((int)threadIdx.x < n) {
            const uint32_t key = F.pass[F.qAll[d4f_qoff(D4F_Q_APPLY) + base + threadIdx.x]].key - 1;
            const int leaf = (int)((key >> 16) & 1u);
            int mNew = 0;
            if (!leaf) { mNew = atomicAdd(&F.nMask, 1); if (mNew >= D4F_MAXM) { F.fallback = 1; mNew = 0; } }
            D4FClTask& T = C.task[threadIdx.x];
            d4f_cl_set(&T.m, (int)(key & 255u)); d4f_cl_set(&T.code, (int)((key >> 8) & 127u)); d4f_cl_set(&T.prune, (int)((key >> 15) & 1u));
            d4f_cl_set(&T.leaf, leaf); d4f_cl_set(&T.mNew, mNew); d4f_cl_set(&T.rem, -1);
        }
        if (threadIdx.x == 0) { d4f_cl_set(&C.cmd, D4F_CMD_APPLY); d4f_cl_set(&C.nTasks, n); d4f_cl_set(&C.nChunks, nChunks); d4f_cl_set(&C.nItems, n * nChunks); }
        d4f_cl_run();
        for (int t = wave; t < n; t += nw) {
            D4FWaveScr& W = F.scr.wave[wave & 7];
            const D4FClTask& T = C.task[t];
            D4FPassE& e = F.pass[F.qAll[d4f_qoff(D4F_Q_APPLY) + base + t]];
            const int m = T.m, cnt = T.cnt;
            int mOut = m, saved = 0;
            if (cnt != 0 && (T.leaf || T.mNew != 0)) {
                saved = T.saved;
                if (!T.leaf) {
                    for (int i = lane; i < D4G_HIST; i += 64) W.hist[i] = G.hist[(size_t)m * D4G_HIST + i] + (uint32_t)T.delta[i];
                    d4g_wave_sync();
                    if (lane == 0) F.maskFull[T.mNew] = F.maskFull[m];
                    mOut = d4f_publish_mask(c, b, G, T.mNew, T.h1 + F.maskH1[m], T.h2 + F.maskH2[m], F.maskPop[m] + cnt, W.hist);
                }
            }
            if (lane == 0) { e.mOut = (int16_t)mOut; e.saved = saved; d4f_fence_block(); e.st = 2; }
        }
        __syncthreads();
    }
}

D4F_TASK void d4f_cl_least_phase(int nq) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nWords = (int)b.maskWords, nRef = (int)b.refCount;
    const int nChunks = (nWords + D4F_CL_CHUNK - 1) / D4F_CL_CHUNK;
    const uint32_t* stat = c.binStat + b.binStat;
    for (int base = 0; base < nq; base += D4F_CL_MAXT) {
        const int n = nq - base < D4F_CL_MAXT ? nq - base : D4F_CL_MAXT;
        // ---- the walk: per length symbol, literal minus back-reference bits / records / bytes without a code of the unexpanded records ----
        d4f_cl_begin(n);
        D4FClCmd* C1 = &d4f_cl_cmd();
        for (int t = wave; t < n; t += nw) {
            const uint32_t key = F.least[F.qAll[d4f_qoff(D4F_Q_LEAST) + base + t]].key - 1;
            const int m = (int)(key & 255u), code = (int)((key >> 8) & 127u), mode = (int)((key >> 15) & 1u);
            const uint32_t full = F.maskFull[m];
            int bSize = 0, bZ = 0, bCount = 0;
            if (lane < D4G_NBINS) {
                bSize = G.binBase[(size_t)code * 64 + lane];
                bZ = G.binBase[(size_t)code * 64 + 32 + lane];
                bCount = (int)stat[lane * D4G_BINSTRIDE + D4G_BIN_COUNT];
            }
            const bool isFull = lane < D4G_NBINS && ((full >> lane) & 1u);
            const int popFull = wave_sum_i32(isFull ? bCount : 0);
            const int via = F.maskPop[m] - popFull <= nRef - F.maskPop[m];
            D4FClTask& T = C1->task[t];
            if (lane < 32) {
                d4f_cl_set(&T.bin[0][lane], via ? bSize : 0);
                d4f_cl_set(&T.bin[2][lane], via ? bZ : 0);
                d4f_cl_set(&T.bin[1][lane], via && !isFull ? bCount : 0);
            }
            if (lane == 0) {
                d4f_cl_set(&T.m, m); d4f_cl_set(&T.code, code); d4f_cl_set(&T.mode, mode); d4f_cl_set(&T.via, via);
                d4f_cl_set((int32_t*)&T.full, (int)full); d4f_cl_set(&T.rem, -1);
            }
        }
        if (threadIdx.x == 0) { d4f_cl_set(&C1->cmd, D4F_CMD_LEAST_WALK); d4f_cl_set(&C1->nTasks, n); d4f_cl_set(&C1->nChunks, nChunks); d4f_cl_set(&C1->nItems, n * nChunks); }
        d4f_cl_run();
        // ---- the choice (DeflateBlockHuffman.java:373-458), then the expansion of the chosen symbol's records ----
        d4f_cl_begin(n);
        D4FClCmd* C2 = &d4f_cl_cmd();
        if ((int)threadIdx.x < n) {
            const D4FClTask& T = C1->task[threadIdx.x];
            const int mode = T.mode;
            int rem = -1, remSize = 0, remFreq = 0;
            for (int i = 0; i < D4G_NBINS; i++) {
                if (T.bin[1][i] > 0 && T.bin[2][i] == 0) {
                    const bool doRem = mode == 1 ? T.bin[1][i] < remFreq : T.bin[0][i] < remSize;
                    if (rem == -1 || doRem) { rem = i; remSize = T.bin[0][i]; remFreq = T.bin[1][i]; }
                }
            }
            int mNew = 0;
            if (rem >= 0) { mNew = atomicAdd(&F.nMask, 1); if (mNew >= D4F_MAXM) { F.fallback = 1; mNew = 0; } }
            D4FClTask& U = C2->task[threadIdx.x];
            d4f_cl_set(&U.m, T.m); d4f_cl_set(&U.code, T.code); d4f_cl_set(&U.mode, mode); d4f_cl_set((int32_t*)&U.full, (int)T.full);
            d4f_cl_set(&U.rem, rem); d4f_cl_set(&U.mNew, mNew); d4f_cl_set(&U.pad0[0], remSize);
        }
        if (threadIdx.x == 0) { d4f_cl_set(&C2->cmd, D4F_CMD_LEAST_EXPAND); d4f_cl_set(&C2->nTasks, n); d4f_cl_set(&C2->nChunks, nChunks); d4f_cl_set(&C2->nItems, n * nChunks); }
        d4f_cl_run();
        for (int t = wave; t < n; t += nw) {
            D4FWaveScr& W = F.scr.wave[wave & 7];
            const D4FClTask& T = C2->task[t];
            D4FLeastE& e = F.least[F.qAll[d4f_qoff(D4F_Q_LEAST) + base + t]];
            const int m = T.m, rem = T.rem, mNew = T.mNew;
            int mOut = m, dl = 0;
            if (rem >= 0 && mNew != 0) {
                const uint32_t* row = stat + rem * D4G_BINSTRIDE;
                for (int i = lane; i < D4G_HIST; i += 64) W.hist[i] = G.hist[(size_t)m * D4G_HIST + i];
                d4g_wave_sync();
                for (int i = lane; i <= D4G_BIN_COUNT; i += 64) {
                    const int moved = (int)row[i] - T.delta[i];
                    if (!moved) continue;
                    if (i < 256) atomicAdd(&W.hist[i], (unsigned)moved);
                    else if (i < D4G_BIN_COUNT) atomicSub(&W.hist[D4G_NLIT + i - D4G_BIN_DIST], (unsigned)moved);
                    else atomicSub(&W.hist[257 + rem], (unsigned)moved);
                }
                d4g_wave_sync();
                if (lane == 0) F.maskFull[mNew] = T.full | (1u << rem);
                mOut = d4f_publish_mask(c, b, G, mNew, T.h1 + F.maskH1[m], T.h2 + F.maskH2[m], F.maskPop[m] + T.cnt, W.hist);
                dl = T.pad0[0];
            }
            if (lane == 0) { e.mOut = (int16_t)mOut; e.delta = dl; d4f_fence_block(); e.st = 2; }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// The kernel: one workgroup = one block, all its rounds
// ---------------------------------------------------------------------------------------
D4G_DEV D4FGlob d4f_glob(const D4GCtx& c, int blk) {
    D4FGlob G;
    uint8_t* p = (uint8_t*)state_ptr(c, blk, 1);
    G.hist = (uint32_t*)p; p += (size_t)D4F_MAXM * D4G_HIST * 4;
    G.code = (D4FCode*)p; p += (size_t)D4F_MAXC * sizeof(D4FCode);
    G.pairs = (D4FPairs*)p; p += (size_t)(D4F_MAXC + 1) * sizeof(D4FPairs);
    G.hdr = (D4FHdr*)p; p += (size_t)D4F_MAXH * sizeof(D4FHdr);
    G.binBase = (int32_t*)p;
    return G;
}

// state slot 0 (LDS copy in L.st) from ids: lengths of code `cid`, header `hid`, histogram of mask `mid`
__device__ __forceinline__ void d4f_assemble(const D4FGlob& G, D4GState* S, int mid, int cid, int hid, int type, long long litlen) {
    const D4FCode& cd = G.code[cid];
    __syncthreads();
    for (int i = threadIdx.x; i < (D4G_NLIT + D4G_NDIST) / 4; i += blockDim.x) ((uint32_t*)S->litLen)[i] = ((const uint32_t*)cd.lens)[i];
    for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) S->hist[i] = G.hist[(size_t)mid * D4G_HIST + i];
    for (int i = threadIdx.x; i < 8; i += blockDim.x) ((uint32_t*)S->clLen)[i] = 0;
    if (type == D4G_DYNAMIC) {
        const D4FHdr& hd = G.hdr[hid];
        const D4FPairs& pb = G.pairs[hd.base];
        const int np = pb.nPairs;
        for (int i = threadIdx.x; i < np; i += blockDim.x) S->pairs[i] = (uint16_t)(pb.pairs[i] | (((hd.flags[i >> 5] >> (i & 31)) & 1u) ? D4G_PAIR_EXPANDED : 0u));
        __syncthreads();
        if (threadIdx.x < 32) S->clLen[threadIdx.x] = hd.clLen[threadIdx.x];
        if (threadIdx.x == 0) { S->nCl = hd.nCl; S->nPairs = np; S->hdrBits = hd.bits; S->nLit = cd.nLit; S->nDist = cd.nDist; }
    } else {
        if (threadIdx.x == 0) { S->nCl = 0; S->nPairs = 0; S->hdrBits = 0; S->nLit = 0; S->nDist = 0; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S->valid = 1; S->type = type; S->maskSlot = 0; S->flags = 0; S->pad0 = 0;
        S->litlenBits = litlen;
        S->sizeBits = litlen + S->hdrBits;
    }
    __syncthreads();
}

// round set-up (all threads): ids 0 = the block's current mask / code / header.  Returns false when the program does not fit.
D4F_TASK bool d4f_round_setup(const D4GOp* ops0, const D4GOp* ops1, int nOps0, int nOps1) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const D4GState* cur = state_ptr(c, c.active[d4fLds.which], 0);
    __syncthreads();
    const int curType = cur->type;
    const int prog = curType == D4G_FIXED ? 1 : 0;
    const int nOps = prog ? nOps1 : nOps0;
    for (int i = threadIdx.x; i < (int)(offsetof(D4FLds, scr) / 4); i += blockDim.x) ((uint32_t*)&F)[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { F.ops = prog ? ops1 : ops0; F.nOps = nOps; F.curType = curType; F.curSize = cur->sizeBits; }
    if (nOps > D4F_MAXOPS || c.slotsPerBlock > D4F_MAXSLOTS) { __syncthreads(); return false; }
    if (F.nCode > D4F_MAXC - 32) {   // the code table is nearly full: start it afresh (all threads see the same count)
        __syncthreads();
        for (int i = threadIdx.x; i < D4F_MAXC; i += blockDim.x) { F.eState[i] = 0; F.eEq[i] = 0; F.bbState[i] = 0; F.hsState[i] = 0; }
        for (int i = threadIdx.x; i < 256; i += blockDim.x) F.codeHash[i] = 0;
        if (threadIdx.x == 0) F.nCode = 2;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < D4F_MAXC; i += blockDim.x) F.defHdr[i] = -1;
    D4FCode& c1 = G.code[D4F_CODE_FIXED];
    for (int i = threadIdx.x; i < D4G_NLIT + D4G_NDIST; i += blockDim.x)
        c1.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 286 ? 8 : i < D4G_NLIT ? 0 : i < D4G_NLIT + 30 ? 5 : 0);   // HuffmanTable.LIT / DIST :166-209
    for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) G.hist[i] = cur->hist[i];
    D4FPairs& p0 = G.pairs[0];
    D4FHdr& h0 = G.hdr[0];
    const int np = curType == D4G_DYNAMIC ? cur->nPairs : 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) p0.pairs[i] = (uint16_t)(cur->pairs[i] & ~D4G_PAIR_EXPANDED);
    if (threadIdx.x < 10) {
        uint32_t fl = 0;
        for (int j = 0; j < 32; j++) { const int i = threadIdx.x * 32 + j; if (i < np && (cur->pairs[i] & D4G_PAIR_EXPANDED)) fl |= 1u << j; }
        h0.flags[threadIdx.x] = fl;
    }
    if (threadIdx.x < 32) h0.clLen[threadIdx.x] = curType == D4G_DYNAMIC ? cur->clLen[threadIdx.x] : 0;
    int pc = 0;
    const uint64_t* m0 = d4f_mask(c, b, 0);
    for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) pc += __popcll(m0[w]);
    const long long pop = wg_sum_i64(pc, F.red);
    {   // length symbols whose records are all expanded in the incoming mask (F.misc[1]: symbols with an unexpanded record)
        const uint64_t* bmask = c.binMask + b.binMask;
        const int nWords = (int)b.maskWords;
        unsigned nf = 0;
        if (nWords > 4096) nf = (1u << D4G_NBINS) - 1;   // (a long block: not worth the pass — no symbol is taken to be used up)
        else for (int i = threadIdx.x; i < D4G_NBINS * nWords; i += blockDim.x) {
            const int bin = i / nWords, w = i - bin * nWords;
            if (bmask[(long long)bin * nWords + w] & ~m0[w]) nf |= 1u << bin;
        }
        if (nf) atomicOr((unsigned*)&F.misc[1], nf);
        __syncthreads();
        if (threadIdx.x == 0) F.maskFull[0] = ~(unsigned)F.misc[1] & ((1u << D4G_NBINS) - 1);
    }
    // the incoming state's code: found in / added to the code table
    if (wave == 0) {
        const uint32_t* cw = (const uint32_t*)cur->litLen;
        const uint32_t w0 = cw[lane], w1 = lane < 16 ? cw[64 + lane] : 0u;
        const int nLit0 = cur->nLit, nDist0 = cur->nDist;
        unsigned long long h = d4f_mix1(lane, w0) + (lane < 16 ? d4f_mix1(64 + lane, w1) : 0ull);
        h = (unsigned long long)wave_sum_i64((long long)h) + (unsigned long long)nLit0 * 0x100000001b3ULL + ((unsigned long long)nDist0 << 48) +
            (curType == D4G_DYNAMIC ? 0ull : 0x9e3779b97f4a7c15ULL);
        if (h == 0) h = 1;
        int found = -1;
        uint32_t k = (uint32_t)(h >> 40) & 255u;
        for (int probe = 0; probe < 256; probe++, k = (k + 1) & 255u) {
            const int e = F.codeHash[k];
            if (e == 0) break;
            const int id = e - 1;
            if (F.codeH[id] == h) {
                const D4FCode& cd = G.code[id];
                int bad = ((const uint32_t*)cd.lens)[lane] != w0;
                if (lane < 16) bad |= ((const uint32_t*)cd.lens)[64 + lane] != w1;
                if (lane == 0) bad |= cd.nLit != nLit0 || cd.nDist != nDist0 || cd.type != curType;
                if (!d4g_ballot(bad)) { found = id; break; }
            }
        }
        const int nCodeNow = __shfl(F.nCode, 0);
        int code = found;
        if (code < 0) {
            code = nCodeNow;   // (the table was emptied above when it was nearly full: there is room)
            D4FCode& cd = G.code[code];
            ((uint32_t*)cd.lens)[lane] = w0;
            if (lane < 16) ((uint32_t*)cd.lens)[64 + lane] = w1;
            if (lane == 0) {
                cd.nLit = nLit0; cd.nDist = nDist0; cd.type = curType; cd.err = 0;
                F.nCode = code + 1;
                F.codeH[code] = h;
                F.codeHash[k] = (uint8_t)(code + 1);
                F.eState[code] = 0; F.eEq[code] = 0; F.bbState[code] = 0; F.hsState[code] = 0;
            }
        }
        if (lane == 0) F.misc[2] = code;
    }
    if (wave == 1 || blockDim.x == 64) {
        const uint16_t hh = d4f_hist_hash(cur->hist);
        if (lane == 0) F.maskHH[0] = hh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1.nLit = 0; c1.nDist = 0; c1.type = D4G_FIXED; c1.err = 0;
        p0.nPairs = np;
        h0.nCl = cur->nCl; h0.bits = (int32_t)cur->hdrBits; h0.base = 0;
        F.hdrBits[0] = curType == D4G_DYNAMIC ? (int32_t)cur->hdrBits : 0;
        F.nMask = 1; F.nHdr = 1; F.step = 1;
        F.maskPop[0] = (int)pop;
        D4FSlot s;
        s.m = 0; s.c = (int16_t)F.misc[2]; s.h = 0; s.valid = 1; s.type = (uint8_t)curType; s.lazy = 0; s.litlen = cur->litlenBits;
        F.slot[0] = s;
        F.slotReady[0] = 1;
    }
    __syncthreads();
    return true;
}

// selection (all threads): first strict minimum (DeflateStream.java:349-368), winner into slot 0 / mask 0, the round's
// result record.  `best` = the workgroup's smallest candidate key.  Returns false when the winner could not be written.
D4F_TASK bool d4f_select(long long best, D4GRoundResult* out) {
    D4FLds& F = d4fLds;
    D4F_CTX;
    const int wave = threadIdx.x >> 6;
    D4GState* cur = state_ptr(c, c.active[d4fLds.which], 0);
    const long long curSize = F.curSize;
    const long long bestSize = best >> D4G_KEY_SEQ_BITS;
    const int seq = (int)(best & ((1 << D4G_KEY_SEQ_BITS) - 1));
    const bool improved = best != D4G_KEY_NONE && bestSize < curSize;
    int newType = F.curType;
    __syncthreads();
    if (improved) {
        const int opId = seq >> 6, hl = seq & 63;
        const D4GOp op = F.ops[opId];
        const bool fromSrc = op.kind == OP_CAND || op.kind == OP_HDRSEARCH;
        const D4FSlot ws = F.slot[fromSrc ? op.src : op.dst];
        int mid = ws.m;
        __syncthreads();
        if (ws.lazy) {   // the winner's tokens were never written down: expand E0(code) \ mask now
            // (the memo table may have been overlaid by the header searches: entry 0 is written afresh; E0 of the code exists,
            // the state's size came from it)
            if (threadIdx.x == 0) {
                F.pass[0].key = 1u + ((uint32_t)ws.m | ((uint32_t)ws.c << 8));
                F.pass[0].st = 0;
            }
            __syncthreads();
            if (wave == 0) d4f_apply_task(0);
            __syncthreads();
            mid = F.pass[0].mOut;
            if (F.fallback) return false;
        }
        __syncthreads();
        D4GLds& L = F.scr.legacy;
        D4GState* S = &L.st;
        d4f_assemble(G, S, mid, ws.c, ws.h, ws.type, ws.litlen);
        if (op.kind == OP_HDRSEARCH) {
            wg_rewrite_header(&L, c.hdrFlags[hl]);
            if (c.hdrPrune[hl]) { wg_replace_rle_runs(&L, true); wg_recode_header(&L); }
            wg_optimise_header(&L);
        }
        __syncthreads();
        if (threadIdx.x == 0 && S->sizeBits != bestSize) {
#ifdef D4G_HOSTSIM
            fprintf(stderr, "fused select: op %d kind %d lane %d assembled to %lld bits (litlen %lld hdr %lld), the search said %lld\n", opId, op.kind, hl,
                    (long long)S->sizeBits, (long long)S->litlenBits, (long long)S->hdrBits, bestSize);
#endif
            atomicAdd(c.errors, 1);
        }
        if (mid != 0) {
            const uint64_t* ms = d4f_mask(c, b, mid);
            uint64_t* md = d4f_mask(c, b, 0);
            for (int w = threadIdx.x; w < (int)b.maskWords; w += blockDim.x) md[w] = ms[w];
        }
        __syncthreads();
        newType = S->type;
        for (int i = threadIdx.x; i < (int)(sizeof(D4GState) / 4); i += blockDim.x) ((uint32_t*)cur)[i] = ((const uint32_t*)S)[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        D4GRoundResult r;
        r.curSize = curSize;
        r.bestSize = improved ? bestSize : curSize;
        r.bestSeq = improved ? (seq >> 6) : -1;
        r.improved = improved ? 1 : 0;
        r.newType = newType;
        r.pad = 0;
        *out = r;
        F.improved = improved ? 1 : 0;
    }
    __syncthreads();
    return true;
}

// what every workgroup of either kernel starts with: the block and its tables (`which` = index into the active list)
__device__ __forceinline__ void d4f_block_init(const D4GCtx& cArg, const D4FParams& P, int which, D4FClArena* arena) {
    D4FLds& F = d4fLds;
    for (int i = threadIdx.x; i < D4F_MAXC; i += blockDim.x) { F.eState[i] = 0; F.eEq[i] = 0; F.bbState[i] = 0; F.hsState[i] = 0; }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) F.codeHash[i] = 0;
    if (threadIdx.x == 0) {
        const int blk = cArg.active[which];
        F.nCode = 2;
        F.regWords = P.regWords < 64 * D4F_NWR ? P.regWords : 64 * D4F_NWR;
        F.c = cArg;
        F.b = cArg.blocks[blk];
        F.G = d4f_glob(cArg, blk);
        F.cl = arena;
        F.clEpoch = 0;
        F.which = which;
    }
    __syncthreads();
}

// the rounds of the block (all threads of its — control — workgroup)
__device__ __forceinline__ void d4f_block_rounds(const D4FParams& P) {
    D4FLds& F = d4fLds;
    const int which = F.which;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int rounds = 0, info = 0;
    // optional accounting (D4G_FUSED_STATS): [0,8) tasks per kind, [8,16) steps per kind, [16,24) cycles per kind, 24 advance,
    // 25 set-up, 26 selection, 27 rounds, 28 steps, 29 cycles in total, 30 / 31 / 32 masks / codes / headers made
    const bool prof = P.stats != nullptr && threadIdx.x == 0;
    auto acc = [&](int k, long long v) { if (prof) atomicAdd((unsigned long long*)&P.stats[k], (unsigned long long)v); };
    const long long tKernel = prof ? clock64() : 0;
    for (;;) {
        long long tp = prof ? clock64() : 0;
        if (!d4f_round_setup(P.ops[0], P.ops[1], P.nOps[0], P.nOps[1])) { info |= D4F_INFO_FALLBACK; break; }
        if (prof) { acc(25, clock64() - tp); acc(27, 1); }
        const D4GOp* ops = F.ops;
        const int nOps = F.nOps;
        // this thread's ops (op i belongs to thread i mod blockDim): the first two stay in registers for the whole round
        D4GOp myOp0, myOp1;
        memset(&myOp0, 0, sizeof(myOp0));
        memset(&myOp1, 0, sizeof(myOp1));
        if ((int)threadIdx.x < nOps) myOp0 = ops[threadIdx.x];
        if ((int)(threadIdx.x + blockDim.x) < nOps) myOp1 = ops[threadIdx.x + blockDim.x];
        // ---- the program ----
        long long bestKey = D4G_KEY_NONE;
        int guard = 0;
        int lastWhich = -1;   // the kind of task that ran since the last advance (-1: look at every op)
        // An op past its first step waits for a memo entry, and entries only fill in when tasks of their kind run: such an op
        // is looked at again only after a step of a kind it can be waiting for.
        auto may_move = [&](int kind, int stage) -> bool {
            if (stage == 0 || lastWhich < 0) return true;
            switch (kind) {
            case OP_HDRSEARCH: return lastWhich == D4F_Q_HS;
            case OP_POST: case OP_PRUNEHDR: return lastWhich == D4F_Q_HDR;
            case OP_OPT: return lastWhich == D4F_Q_APPLY || lastWhich == D4F_Q_HDR;
            case OP_RECODE: case OP_RECODE_FULL: return stage == 1 ? lastWhich == D4F_Q_APPLY : lastWhich == D4F_Q_TREE;
            case OP_LEAST: return lastWhich == D4F_Q_LEAST;
            case OP_TOFIXED_OPT: return stage == 1 ? lastWhich == D4F_Q_FIXDOT : lastWhich == D4F_Q_APPLY;
            default: return true;
            }
        };
        for (int iter = 0;; iter++) {
            bool prog2 = false;
            tp = prof ? clock64() : 0;
            for (int i = threadIdx.x, n = 0; i < nOps; i += blockDim.x, n++) {
                const int st = F.opStage[i];
                if (st == 255) continue;
                const D4GOp& o = n == 0 ? myOp0 : n == 1 ? myOp1 : ops[i];
                if (F.slotReady[o.src] && may_move(o.kind, st)) prog2 |= d4f_advance_op(o, i, &bestKey);
            }
            if (prog2) F.progress[iter & 1] = 1;
            __syncthreads();
            if (prof) { acc(24, clock64() - tp); acc(28, 1); }
            const int nDone = F.nDone, fb = F.fallback, progressed = F.progress[iter & 1];
            int which = -1;
            for (int q = 0; q < D4F_NQ; q++)
                if (q != D4F_Q_HS && F.qn[q] > 0) { which = q; break; }
            // the header searches wait until nothing else can move (their results only rank candidates): all of them at once
            if (which < 0 && !progressed && F.qn[D4F_Q_HS] > 0) which = D4F_Q_HS;
            if (threadIdx.x == 0) { F.step++; F.progress[(iter + 1) & 1] = 0; }   // (the other flag: read before the previous step's barrier, written after this one)
            __syncthreads();
            if (fb) break;
            if (nDone >= nOps) break;
            lastWhich = which;
            if (which < 0) {
                if (!progressed && ++guard > 4) { if (threadIdx.x == 0) { atomicAdd(F.c.errors, 1); F.fallback = 1; } __syncthreads(); break; }
                continue;
            }
            guard = 0;
            const int nq = F.qn[which] < d4f_qcap(which) ? F.qn[which] : d4f_qcap(which);
            if (prof) { acc(which, nq); acc(8 + which, 1); tp = clock64(); }
            switch (which) {
            case D4F_Q_SWEEP: if (F.cl) d4f_cl_sweep(); else d4f_sweep(); break;
            case D4F_Q_APPLY:
                if (F.cl) { d4f_cl_apply_phase(nq); break; }
                for (int t = wave; t < nq; t += nw) d4f_apply_task(F.qAll[d4f_qoff(D4F_Q_APPLY) + t]);
                break;
            case D4F_Q_BINBASE: d4f_binbase(); break;
            case D4F_Q_LEAST:
                if (F.cl) { d4f_cl_least_phase(nq); break; }
                for (int t = wave; t < nq; t += nw) d4f_least_task(F.qAll[d4f_qoff(D4F_Q_LEAST) + t]);
                break;
            case D4F_Q_TREE: {
                // Masks with the same histogram have the same rebuild: a queued mask whose histogram equals that of a mask whose
                // rebuild stands, or of a mask earlier in the queue, takes that one's result instead of being built.
                uint16_t* Q = &F.qAll[d4f_qoff(D4F_Q_TREE)];
                const int nMaskNow = F.nMask;
                if ((int)threadIdx.x < nq) { F.treeCand[0][threadIdx.x] = 0x7fffffff; F.treeCand[1][threadIdx.x] = 0x7fffffff; }
                __syncthreads();
                for (int i = threadIdx.x; i < nq * nMaskNow; i += blockDim.x) {
                    const int t = i / nMaskNow, mp = i - t * nMaskNow, m = Q[t];
                    if (mp != m && F.treeSt[mp] == 2 && F.maskHH[mp] == F.maskHH[m]) atomicMin(&F.treeCand[0][t], mp);
                }
                for (int i = threadIdx.x; i < nq * nq; i += blockDim.x) {
                    const int t = i / nq, j = i - t * nq;
                    if (j < t && F.maskHH[Q[j]] == F.maskHH[Q[t]]) atomicMin(&F.treeCand[1][t], j);
                }
                __syncthreads();
                // the candidate's histogram entry by entry (one wave per queued mask); treeCand[0][t] becomes the mask to copy from, or -1
                for (int t = wave; t < nq; t += nw) {
                    const int m = Q[t];
                    int from = F.treeCand[0][t] != 0x7fffffff ? F.treeCand[0][t] : F.treeCand[1][t] != 0x7fffffff ? (int)Q[F.treeCand[1][t]] : -1;
                    if (from >= 0) {
                        int diff = 0;
                        for (int i = lane; i < D4G_HIST; i += 64) diff |= F.G.hist[(size_t)m * D4G_HIST + i] != F.G.hist[(size_t)from * D4G_HIST + i];
                        if (d4g_ballot(diff)) from = -1;
                    }
                    d4g_wave_sync();
                    if (lane == 0) F.treeCand[0][t] = from;
                }
                __syncthreads();
                // the masks to build move to the front of the queue; the others are listed with the mask they copy, in queue order
                if (threadIdx.x == 0) {   // (treeCand[1] is free again: mask << 16 | the mask it copies)
                    int nu = 0, na = 0;
                    for (int t = 0; t < nq; t++) {
                        const int m = Q[t], from = F.treeCand[0][t];
                        if (from < 0) Q[nu++] = (uint16_t)m;
                        else F.treeCand[1][na++] = (m << 16) | from;
                    }
                    F.misc[3] = nu;
                }
                __syncthreads();
                const int nu = F.misc[3];
                if (prof) acc(38, nq - nu);
                const int nts = (nw >> 1) < D4F_TREE_SLOTS ? (nw >> 1) : D4F_TREE_SLOTS;   // (at least two waves per workgroup)
                // waves 0 .. nts-1 build the literal/length trees, waves nts .. 2 nts-1 the distance trees: the long builds
                // then sit on different SIMDs (wave w runs on SIMD w mod 4) instead of two to a SIMD
                const bool distWave = wave >= nts;
                const int ts = distWave ? wave - nts : wave;
                for (int t0 = 0; t0 < nu; t0 += nts) {
                    long long tq = prof ? clock64() : 0;
                    if (ts < nts && t0 + ts < nu) {
                        const int m = Q[t0 + ts];
                        if (distWave) d4f_tree_dist(ts, m); else d4f_tree_lit(ts, m);
                    }
                    if (prof) { acc(33, clock64() - tq); tq = clock64(); }
                    __syncthreads();
                    if (prof) { acc(34, clock64() - tq); tq = clock64(); }
                    if (ts < nts && t0 + ts < nu && !distWave) d4f_tree_header(ts);
                    __syncthreads();
                    if (prof) { acc(35, clock64() - tq); tq = clock64(); }
                    if (wave == 0)
                        for (int t = 0; t < nts && t0 + t < nu; t++) d4f_tree_publish(t);
                    __syncthreads();
                    if (prof) { acc(36, clock64() - tq); acc(37, 1); }
                }
                if (threadIdx.x == 0) {   // in queue order: a mask copied from an earlier one of this queue finds it finished
                    for (int k = 0; k < nq - nu; k++) {
                        const int m = F.treeCand[1][k] >> 16, from = F.treeCand[1][k] & 0xffff;
                        F.treeC[m] = F.treeC[from]; F.treeLit[m] = F.treeLit[from];
                        d4f_fence_block();
                        F.treeSt[m] = 2;
                    }
                }
                break;
            }
            case D4F_Q_HDR:
                for (int t = wave; t < nq; t += nw) d4f_hdr_task(F.qAll[d4f_qoff(D4F_Q_HDR) + t]);
                break;
            case D4F_Q_HS: {
                const int nhs = nw < D4F_HS_SLOTS ? nw : D4F_HS_SLOTS;
                for (int t0 = 0; t0 < nq; t0 += nhs) {
                    if (wave < nhs && t0 + wave < nq) d4f_hs_task(wave, F.qAll[d4f_qoff(D4F_Q_HS) + t0 + wave]);
                    __syncthreads();
                }
                break;
            }
            case D4F_Q_FIXDOT: {   // recodeToFixedHuffman's Σ token bits of mask m under the fixed code — :637-653
                const int m = F.qAll[d4f_qoff(D4F_Q_FIXDOT) + 0];
                long long v = 0;
                const uint8_t* ln = F.G.code[D4F_CODE_FIXED].lens;
                for (int i = threadIdx.x; i < D4G_HIST; i += blockDim.x) {
                    const unsigned h = F.G.hist[(size_t)m * D4G_HIST + i];
                    if (h) v += (long long)h * (ln[i] + (i < D4G_NLIT ? (i >= 257 ? d4g_lsym_ebits(i) : 0) : d4g_dsym_ebits(i - D4G_NLIT)));
                }
                v = wg_sum_i64(v, F.red);
                if (threadIdx.x == 0) { F.fixdotLit = v; F.fixdotSt = 2; }
                break;
            }
            default: break;
            }
            __syncthreads();
            if (prof) acc(16 + which, clock64() - tp);
            if (threadIdx.x == 0) F.qn[which] = 0;
            __syncthreads();
        }
        if (prof) { acc(30, F.nMask); acc(31, F.nCode); acc(32, F.nHdr); }
        if (F.fallback) { info |= D4F_INFO_FALLBACK; break; }
        // ---- selection ----
        long long best = wave_min_i64(bestKey);
        __syncthreads();
        if (lane == 0) F.red[wave] = best;
        __syncthreads();
        for (int i = 0; i < nw; i++) best = F.red[i] < best ? F.red[i] : best;
        __syncthreads();
        tp = prof ? clock64() : 0;
        if (!d4f_select(best, &P.results[(size_t)which * D4F_MAXROUNDS + rounds])) { info |= D4F_INFO_FALLBACK; break; }
        if (prof) { acc(26, clock64() - tp); if (rounds == 0) acc(63, clock64() - tKernel); }   // [63]: cycles of the blocks' first rounds
        rounds++;
        if (!F.improved) break;
        if (rounds >= P.maxRounds || rounds >= D4F_MAXROUNDS) { info |= D4F_INFO_MORE; break; }
    }
    __syncthreads();
    if (prof) {   // [39] the slowest block, [40, 63) blocks by duration (0.5 M cycles per class)
        const long long dt = clock64() - tKernel;
        acc(29, dt);
        atomicMax((unsigned long long*)&P.stats[39], (unsigned long long)dt);
        const int cls = (int)(dt / 500000) < 22 ? (int)(dt / 500000) : 22;
        acc(40 + cls, 1);
    }
    if (threadIdx.x == 0) P.roundInfo[which] = rounds | info;
}

#ifndef D4F_WAVES_PER_SIMD
#define D4F_WAVES_PER_SIMD 4
#endif
__global__ void __launch_bounds__(512) D4G_WAVES_PER_SIMD(D4F_WAVES_PER_SIMD) k_search_fused(D4GCtx cArg, D4FParams P) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    if ((int)blockIdx.x >= cArg.nActive) return;
    d4f_block_init(cArg, P, (int)blockIdx.x, nullptr);
    d4f_block_rounds(P);
}

// One long block, many workgroups: workgroup 0 runs the search, the others work off the commands it posts (D4FClArena).
__global__ void __launch_bounds__(512) D4G_WAVES_PER_SIMD(D4F_WAVES_PER_SIMD) k_search_cluster(D4GCtx cArg, D4FParams P, D4FClArena* arena, int which) {
#ifndef D4G_HOSTSIM
    __builtin_amdgcn_s_setprio(D4G_BASE_PRIO);
#endif
    D4FLds& F = d4fLds;
    d4f_block_init(cArg, P, which, arena);
    if (blockIdx.x == 0) {
        d4f_block_rounds(P);
        __syncthreads();
        if (threadIdx.x == 0) { d4g_release_agent(); d4f_atomic_st(&arena->epoch, -1); }
        return;
    }
    int seen = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            int e = seen;
            for (long long spin = 0; spin < (1LL << 23); spin++) {   // (a helper that waits this long — seconds — leaves: the control workgroup does not need it)
                e = d4f_atomic_ld(&arena->epoch);
                if (e != seen) break;
                d4g_sleep();
            }
            if (e == seen) e = -1;
            if (e > 0) d4g_acquire_agent();
            F.misc[0] = e;
        }
        __syncthreads();
        const int e = F.misc[0];
        __syncthreads();
        if (e < 0) break;
        if (threadIdx.x == 0) F.clEpoch = e;   // (commands it missed were finished without it)
        __syncthreads();
        d4f_cl_work();
        seen = e;
    }
}
