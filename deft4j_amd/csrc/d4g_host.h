// d4g_host.h — host orchestration of libdeft4g: the candidate-search program generator, the
// stream-level loops of the reference (DeflateStream.optimise / mergeBlocks) and the batch
// object behind the C ABI.  The host only sequences kernels and resolves the few decisions
// that depend on a stream-wide bit position (stored-block alignment); all token, Huffman and
// header arithmetic runs in the HIP kernels.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <set>
#include <vector>

#include "d4g_ops.h"
#include "d4g_fused.h"
#include "d4g_parse.h"
#include "d4g_rt.h"
#include "d4g_write.h"
#include "../../include/deft4g.h"

namespace d4g {

typedef long long i64;

// threads per state-op workgroup (the kernels work for any multiple of 64; the CPU emulator
// in tests/hostsim lowers it to keep fiber switching cheap)
static inline int state_block() {
#ifdef D4G_HOSTSIM
    const char* e = getenv("D4G_SIM_BLOCK");
    if (e) return atoi(e);
#endif
    static int v = 0;
    if (!v) {
        const char* t = getenv("D4G_STATE_BLOCK");  // tuning knob: 64, 128 or 256
        v = t ? atoi(t) : 256;   // four waves per op: with the memos a level is bound by its longest ops, not by throughput
        if (v != 64 && v != 128 && v != 256) v = 256;
    }
    return v;
}

static inline int wide_block() {
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("D4G_WIDE_BLOCK");  // threads per workgroup for token-pass-only ops; 0 = same launch as the others
        v = t ? atoi(t) : 0;
        if (v != 0 && v != 256 && v != 512 && v != 1024) v = 0;
    }
#ifdef D4G_HOSTSIM
    return 0;
#endif
    return v;
}
static inline int lanes() {
    // tuning knob, read per call: block groups running their level sequences concurrently.  bench.py's roofline leg
    // sets it to 1 so that a launch's event-timed duration is not inflated by the neighbouring lane.
    const char* t = getenv("D4G_LANES");
    int v = t ? atoi(t) : 2;  // measured on MI355X: 1 -> 587, 2 -> 627, 4 -> 494, 8 -> 349 MB/s on config 2
    if (v < 1) v = 1;
    if (v > RT_MAX_LANES) v = RT_MAX_LANES;
    return v;
}

// Executor choice per round.  "persistent": dependency-driven work-queue kernels (no launch per level, no
// tails) — wins while the round is latency-bound (few active blocks: nerd.png 115 -> 68 ms).  "levels": one
// launch per program level over all active blocks — wins once there are enough blocks to fill the chip
// (config 2, 332 blocks: 79 vs 87 ms).  D4G_EXEC=persistent|levels forces one; default switches at
// D4G_PERSIST_MAX_BLOCKS active blocks.
// "fused" (the default): one workgroup per block runs the whole search, all rounds, out of LDS (d4g_fused.h); the two
// executors below remain for blocks it does not take (very large merged blocks, table overflows) and as cross-checks.
static inline bool exec_fused() {
    const char* t = getenv("D4G_EXEC");
    return !t || !strcmp(t, "fused");
}
// Blocks with more back-references than this go to the level / persistent executors.  The fused executor is one workgroup
// per block: unbeatable while there are blocks enough to fill the device, but a lone long block (the merge chain of one
// big stream: 3.8 ms per round at 100 k back-references with the persistent executor's many workgroups, 7.5 ms fused) is
// better served by op-level parallelism.  Measured on config 2 with merge on / 64 x 1 MiB with merge on.
static inline long long fused_max_refs(size_t nLong) {   // nLong: blocks of the round with more than 16384 back-references
    const char* t = getenv("D4G_FUSED_MAX_REFS");
    if (t) return atoll(t);
    return nLong >= 8 ? (1LL << 17) : (1LL << 14);
}
// A lone long block (more back-references than this) gets the whole device: k_search_cluster.  D4G_CLUSTER=0: never.
static inline long long cluster_min_refs() {
    const char* t = getenv("D4G_CLUSTER_MIN_REFS");   // (read per call: the tests switch it inside one process)
    return t ? atoll(t) : (1LL << 14);
}
static inline bool cluster_enabled() {
    const char* t = getenv("D4G_CLUSTER");
    return !(t && t[0] == '0');
}
static inline int exec_persistent(int nActive = 0) {
    // read per call (not cached): the parity tests switch executors inside one process
    const char* t = getenv("D4G_EXEC");
    const int mode = !t ? 2 : !strcmp(t, "levels") ? 0 : !strcmp(t, "persistent") ? 1 : 2;
    const char* m = getenv("D4G_PERSIST_MAX_BLOCKS");
    const int maxBlocks = m ? atoi(m) : 128;
#ifdef D4G_HOSTSIM
    return mode == 0 ? 0 : 1;
#endif
    if (mode == 2) return nActive <= maxBlocks ? 1 : 0;
    return mode;
}
// threads per workgroup of the block decoders (probe / emit): a batch is that many 512-bit chunks decoded side by side
static inline int parse_threads() {
#ifdef D4G_HOSTSIM
    const char* e = getenv("D4G_SIM_PARSE_THREADS");   // the emulator's fibers are slow: one wave unless a test asks for more
    int v = e ? atoi(e) : 64;
#else
    const char* e = getenv("D4G_PARSE_THREADS");
    int v = e ? atoi(e) : 512;
#endif
    if (v != 64 && v != 128 && v != 256 && v != 512) v = 64;
    return v;
}
static inline int env_int(const char* name, int def) {
    const char* t = getenv(name);
    return t ? atoi(t) : def;
}

static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------
// Candidate-search program = DeflateStream.optimiseBlock (B/deflate/DeflateStream.java:343-490)
// unrolled into ops over state slots.  Op ids follow the reference's enumeration order, so
// "first strict minimum" is min over (size, op id, lane).
// ---------------------------------------------------------------------------------------
struct Program {
    std::vector<D4GOp> ops;
    std::vector<int> opLevel;
    std::vector<int> slotLevel;
    int nSlots = 1, nMasks = 1, nLevels = 0;
    std::vector<std::vector<int>> stateLevels, hdrLevels;
    D4GOp* dOps = nullptr;
    int32_t* dLists = nullptr;
    int32_t *dStateFlat = nullptr, *dHdrFlat = nullptr;  // level-ordered op ids for the persistent executor
    int nStateFlat = 0, nHdrFlat = 0;
    std::vector<std::pair<size_t, int>> stateOff, hdrOff, wideOff;  // per level: (offset into dLists, count)
    int nRequested = 0;  // ops the plain unrolling would have emitted (for the record)
    std::set<int> hsCodes;  // distinct code-length sets among the header searches (for the record)

    // ---- symbolic identity of a state, used to emit every distinct computation once ----
    // A state is determined by (m, c, h): token mask, code lengths, header — each the result of a
    // symbolic function application of the ops below — plus g, the guard under which the reference
    // builds it at all (a null optimiseBlockNormal result or an unchanged recodedHuffmanFull removes
    // whole subtrees).  Two requests with the same key are the same computation on the same data, so
    // the later one can only repeat an earlier candidate of equal size and never wins the strict `<`.
    struct Sym { int m, c, h, g; };
    std::vector<Sym> sym;               // per slot
    std::map<std::vector<int>, int> ids;      // symbolic function application -> id
    std::map<std::vector<int>, int> slotOf;   // op key -> slot
    std::set<std::vector<int>> offered, searched;
    int nextId = 1;
    int id_of(std::vector<int> key) {
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        return ids[key] = nextId++;
    }

    int new_slot(int level, Sym sy) { slotLevel.push_back(level); sym.push_back(sy); return nSlots++; }
    int emit_raw(int kind, int src, int dst, int arg, bool cand, bool needMask, int level) {
        D4GOp op;
        memset(&op, 0, sizeof(op));
        op.kind = kind;
        op.src = src;
        op.dst = dst;
        op.arg = arg;
        op.seq = cand ? (int)ops.size() : -1;
        op.maskSlot = needMask ? nMasks++ : -1;
        op.scratch = -1;
        op.scratchMask = -1;
        if (kind == OP_RECODE_FULL) {
            op.scratch = new_slot(level, sym[src]); new_slot(level, sym[src]);
            op.scratchMask = nMasks; nMasks += 2;
        }
        ops.push_back(op);
        opLevel.push_back(level);
        return (int)ops.size() - 1;
    }
    // offer `slot` as a candidate at this point of the enumeration unless an equal state was offered before
    void offer(int slot, int cond, int opIdx) {
        const Sym& y = sym[slot];
        std::vector<int> uncond = {y.m, y.c, y.h, y.g, 0}, withc = {y.m, y.c, y.h, y.g, cond};
        bool dup = offered.count(uncond) || offered.count(withc);
        if (dup) { if (opIdx >= 0) ops[opIdx].seq = -1; return; }
        offered.insert(withc);
        if (opIdx < 0) emit_raw(OP_CAND, slot, -1, 0, true, false, slotLevel[slot] + 1);
    }
    // generic state op: kind/arg applied to src; `cand` offers the result
    int state_op(int kind, int src, int arg, bool cand, bool needMask) {
        nRequested++;
        const Sym x = sym[src];
        Sym y = x;
        int cond = 0;
        switch (kind) {
        case OP_RECODE:
            if (arg & 1) y.m = id_of({OP_RECODE, x.m, x.c});
            y.c = id_of({-1, y.m});          // code rebuilt from the histogram of mask y.m
            y.h = id_of({-2, y.c});          // rewriteHeader(default flags) of those lengths
            break;
        case OP_OPT:
            y.m = id_of({OP_OPT, x.m, x.c});
            y.h = id_of({OP_OPT, x.h});
            if (arg & 1) { cond = id_of({-3, x.m, x.c, x.h}); y.g = id_of({-4, x.g, cond}); }
            break;
        case OP_LEAST:
            y.m = id_of({OP_LEAST, arg, x.m, x.c});
            break;
        case OP_POST: y.h = id_of({OP_POST, x.h}); break;
        case OP_PRUNEHDR: y.h = id_of({OP_PRUNEHDR, x.h}); break;
        case OP_RECODE_FULL:
            y.m = id_of({OP_RECODE_FULL, 0, x.m, x.c, x.h});
            y.c = id_of({OP_RECODE_FULL, 1, x.m, x.c, x.h});
            y.h = id_of({OP_RECODE_FULL, 2, x.m, x.c, x.h});
            cond = id_of({-5, x.m, x.c, x.h});
            y.g = id_of({-4, x.g, cond});
            break;
        case OP_TOFIXED_OPT:
            y.m = id_of({OP_TOFIXED_OPT, x.m});
            y.c = id_of({-6});
            y.h = 0;
            break;
        default: break;
        }
        std::vector<int> key = {kind, arg, x.m, x.c, x.h, x.g};
        auto it = slotOf.find(key);
        if (it != slotOf.end()) {
            if (cand) offer(it->second, cond, -1);
            return it->second;
        }
        int level = slotLevel[src] + 1;
        int dst = new_slot(level, y);
        int opIdx = emit_raw(kind, src, dst, arg, cand, needMask, level);
        slotOf[key] = dst;
        if (cand) offer(dst, cond, opIdx);
        return dst;
    }
    int OPT(int src, bool requireSaved, bool cand) { return state_op(OP_OPT, src, requireSaved ? 1 : 0, cand, true); }
    int RECODE(int src, bool prune, bool cand) { return state_op(OP_RECODE, src, prune ? 1 : 0, cand, prune); }
    int FULL(int src, bool cand) { return state_op(OP_RECODE_FULL, src, 0, cand, true); }
    int LEAST(int src, int mode) { return state_op(OP_LEAST, src, mode, false, true); }
    void HS(int base) {  // the 56 header candidates depend only on the base's token bits and code lengths
        nRequested++;
        const Sym& y = sym[base];
        std::vector<int> key = {y.m, y.c, y.g};
        if (!searched.insert(key).second) return;
        hsCodes.insert(y.c);
        emit_raw(OP_HDRSEARCH, base, -1, 0, true, false, slotLevel[base] + 1);
    }

    void aor(int t) {  // addOptimisedRecoded — DeflateStream.java:265-317
        int b1 = OPT(t, false, false);
        int b2 = OPT(RECODE(t, false, false), false, false);
        int pruned = RECODE(t, true, false);
        int b3 = OPT(pruned, false, false);
        int b4 = OPT(FULL(pruned, false), false, false);
        HS(b1);
        HS(b2);
        HS(b3);
        HS(b4);
    }
    void run(int x) {  // runOptimisationsCallback — :400-442
        int post = state_op(OP_POST, x, 0, true, false);
        OPT(post, true, true);
        aor(post);
        int prune = state_op(OP_PRUNEHDR, x, 0, true, false);
        OPT(prune, true, true);
        aor(prune);
        aor(LEAST(x, 0));
        aor(LEAST(x, 1));
    }
    void multi(int e) {  // runOptimisationsCallbackMulti — :443-463
        nRequested++;
        offer(e, 0, -1);
        run(e);
        int hr = RECODE(e, false, true);
        run(hr);
        int hp = RECODE(e, true, true);
        run(hp);
        int hpf = FULL(hp, true);
        run(hpf);
    }
    void build(bool fixedOrigin) {
        slotLevel.assign(1, 0);
        sym.assign(1, Sym{id_of({-10}), id_of({-11}), id_of({-12}), 0});
        int T = 0;
        // the current block itself is the incumbent: candidates equal to it can never be strictly smaller
        offered.insert({sym[0].m, sym[0].c, sym[0].h, 0, 0});
        int optimised = OPT(T, true, true);  // op 0: "optimised"; the stored candidate (host) ranks right after it
        int H, OH;
        if (fixedOrigin) {
            H = RECODE(T, false, false);
            OH = OPT(H, true, false);
        } else {
            H = T;
            OH = optimised;
        }
        multi(H);
        multi(OH);
        if (!fixedOrigin) state_op(OP_TOFIXED_OPT, H, 0, true, true);  // "default fixed-huffman"
        multi(LEAST(H, 0));
        multi(LEAST(H, 1));
        // drop ops whose result feeds no candidate and no header search (e.g. bases of a repeated search)
        {
            std::vector<int> producer(nSlots, -1);
            for (size_t i = 0; i < ops.size(); i++)
                if (ops[i].dst >= 0) producer[ops[i].dst] = (int)i;
            std::vector<char> live(ops.size(), 0);
            std::vector<int> stack;
            for (size_t i = 0; i < ops.size(); i++)
                if (ops[i].seq >= 0 || ops[i].kind == OP_HDRSEARCH || ops[i].kind == OP_CAND) { live[i] = 1; stack.push_back((int)i); }
            while (!stack.empty()) {
                int i = stack.back();
                stack.pop_back();
                int pr = producer[ops[i].src];
                if (pr >= 0 && !live[pr]) { live[pr] = 1; stack.push_back(pr); }
            }
            std::vector<D4GOp> kept;
            std::vector<int> keptLevel;
            for (size_t i = 0; i < ops.size(); i++)
                if (live[i]) {
                    D4GOp o = ops[i];
                    if (o.seq >= 0) o.seq = (int)kept.size();
                    kept.push_back(o);
                    keptLevel.push_back(opLevel[i]);
                }
            ops.swap(kept);
            opLevel.swap(keptLevel);
        }
        // optimise() results nothing builds on (they are offered / searched for headers only): the fused executor computes
        // their size without writing their tokens down (arg bit 8; the other executors read bit 0 only)
        {
            std::vector<int> stateUses(nSlots, 0);
            for (const D4GOp& o : ops)
                if (o.kind != OP_HDRSEARCH && o.kind != OP_CAND) stateUses[o.src]++;
            for (D4GOp& o : ops)
                if (o.kind == OP_OPT && stateUses[o.dst] == 0) o.arg |= 0x100;
        }
        nLevels = 0;
        for (int l : opLevel) nLevels = std::max(nLevels, l + 1);
        stateLevels.assign(nLevels, {});
        hdrLevels.assign(nLevels, {});
        for (size_t i = 0; i < ops.size(); i++)
            (ops[i].kind == OP_HDRSEARCH ? hdrLevels : stateLevels)[opLevel[i]].push_back((int)i);
        // Within a level the ops are independent; the long ones are dispatched first so that the short ones fill the
        // launch's tail (execution order only — candidate ranking goes by op id).
        auto cost = [&](int id) {
            switch (ops[id].kind) {
            case OP_RECODE_FULL: return 8;
            case OP_RECODE: return (ops[id].arg & 1) ? 6 : 4;
            case OP_OPT: case OP_TOFIXED_OPT: case OP_LEAST: return 2;
            default: return 1;
            }
        };
        for (auto& v : stateLevels) std::stable_sort(v.begin(), v.end(), [&](int x, int y) { return cost(x) > cost(y); });
    }
    void release() {
        rt_free(dOps); rt_free(dLists); rt_free(dStateFlat); rt_free(dHdrFlat);
        dOps = nullptr; dLists = nullptr; dStateFlat = nullptr; dHdrFlat = nullptr;
        stateOff.clear(); hdrOff.clear(); wideOff.clear();
    }
    void upload() {
        dOps = (D4GOp*)rt_malloc(ops.size() * sizeof(D4GOp));
        rt_h2d(dOps, ops.data(), ops.size() * sizeof(D4GOp));
        std::vector<int32_t> lists;
        for (int l = 0; l < nLevels; l++) {
            // token-pass-only ops (no single-lane section) go to the wide-workgroup launch
            std::vector<int> narrow, wide;
            for (int id : stateLevels[l])
                {
                    // D4G_WIDE_KINDS: bit k set = ops of kind k run in the wide launch (default: the token-pass-only kinds)
                    static const int wideKinds = env_int("D4G_WIDE_KINDS", (1 << OP_OPT) | (1 << OP_LEAST));
                    (wide_block() > 0 && ((wideKinds >> ops[id].kind) & 1) ? wide : narrow).push_back(id);
                }
            stateOff.push_back({lists.size(), (int)narrow.size()});
            lists.insert(lists.end(), narrow.begin(), narrow.end());
            wideOff.push_back({lists.size(), (int)wide.size()});
            lists.insert(lists.end(), wide.begin(), wide.end());
            hdrOff.push_back({lists.size(), (int)hdrLevels[l].size()});
            lists.insert(lists.end(), hdrLevels[l].begin(), hdrLevels[l].end());
        }
        dLists = (int32_t*)rt_malloc(lists.size() * sizeof(int32_t));
        rt_h2d(dLists, lists.data(), lists.size() * sizeof(int32_t));
        std::vector<int32_t> sf, hf;
        for (int l = 0; l < nLevels; l++) {
            sf.insert(sf.end(), stateLevels[l].begin(), stateLevels[l].end());
            hf.insert(hf.end(), hdrLevels[l].begin(), hdrLevels[l].end());
        }
        nStateFlat = (int)sf.size();
        nHdrFlat = (int)hf.size();
        dStateFlat = (int32_t*)rt_malloc(sf.size() * sizeof(int32_t) + 16);
        dHdrFlat = (int32_t*)rt_malloc(hf.size() * sizeof(int32_t) + 16);
        rt_h2d(dStateFlat, sf.data(), sf.size() * sizeof(int32_t));
        rt_h2d(dHdrFlat, hf.data(), hf.size() * sizeof(int32_t));
        rt_sync();
    }
};

// The 56 (flags, prune) pairs in addOptimisedRecoded's loop order — DeflateStream.java:281-315
static void build_hdr_tables(uint8_t* flags, uint8_t* prune) {
    int k = 0;
    for (int noRepZeros = 0; noRepZeros < 2; noRepZeros++)
        for (int pr = 0; pr < 2; pr++)
            for (int noRep = 0; noRep < (noRepZeros ? 1 : 2); noRep++)
                for (int noZRep = (noRepZeros ? 1 : 0); noZRep < 2; noZRep++)
                    for (int noZRep2 = 0; noZRep2 < 2; noZRep2++)
                        for (int ohh = 1; ohh >= 0; ohh--) {
                            int base = (noRep ? F_NOREP : 0) | (noZRep ? F_NOZREP : 0) | (noZRep2 ? F_NOZREP2 : 0) | (noRepZeros ? F_NOREPZEROS : 0);
                            if (ohh) {
                                if (noRep) continue;
                                for (int use8 = 1; use8 >= 0; use8--)
                                    for (int use7 = 1; use7 >= 0; use7--) {
                                        if (!use8 && !use7) continue;
                                        flags[k] = (uint8_t)(base | F_OHH | (use8 ? F_USE8 : 0) | (use7 ? F_USE7 : 0));
                                        prune[k] = (uint8_t)pr;
                                        k++;
                                    }
                            } else {
                                flags[k] = (uint8_t)base;
                                prune[k] = (uint8_t)pr;
                                k++;
                            }
                        }
    if (k != 56) throw std::runtime_error("header flag table: expected 56 candidates");
}

struct Engine {  // per-process device objects shared by all batches
    Program progDyn, progFixed;
    uint8_t* dHdrTables = nullptr;  // flags[64] + prune[64]
    long long* dOpStats = nullptr;
    uint32_t* dCrcTab = nullptr;   // [1024] slice-by-4 CRC-32 tables, then [32] x^(2^k) mod P
    int slotsPerBlock = 0, masksPerBlock = 0, maxOps = 0;
    bool ready = false, built = false;
    // d4g_shutdown: the device objects go back (a later d4g_init may pick another device)
    void release() {
        if (!ready) return;
        progDyn.release(); progFixed.release();
        rt_free(dHdrTables); rt_free(dOpStats); rt_free(dCrcTab);
        dHdrTables = nullptr; dOpStats = nullptr; dCrcTab = nullptr;
        ready = false;
    }
    void init() {
        static std::mutex initMu;   // several host threads may arrive with the first batches
        std::lock_guard<std::mutex> lk(initMu);
        if (ready) return;
        if (!built) {
            progDyn.build(false);
            progFixed.build(true);
            built = true;
        }
        if (getenv("D4G_DEBUG_PROGRAM"))
            fprintf(stderr, "program: %d ops requested, %zu emitted (%zu header searches over %zu distinct code-length sets), %d levels, %d slots, %d masks\n",
                    progDyn.nRequested, progDyn.ops.size(), (size_t)std::count_if(progDyn.ops.begin(), progDyn.ops.end(), [](const D4GOp& o) { return o.kind == OP_HDRSEARCH; }),
                    progDyn.hsCodes.size(), progDyn.nLevels, progDyn.nSlots, progDyn.nMasks);
        if (const char* dp = getenv("D4G_DEBUG_PROGRAM"))
            if (atoi(dp) >= 2)
                for (int l = 0; l < progDyn.nLevels; l++) {
                    int kinds[16] = {0};
                    for (int id : progDyn.stateLevels[l]) kinds[progDyn.ops[id].kind * 1 + 0]++;
                    fprintf(stderr, "level %2d: OPT %d RECODE %d FULL %d LEAST %d POST %d PRUNEHDR %d TOFIXED %d CAND %d | hdr searches %zu\n", l, kinds[1], kinds[2],
                            kinds[3], kinds[4], kinds[5], kinds[6], kinds[7], kinds[8], progDyn.hdrLevels[l].size());
                }
        progDyn.upload();
        progFixed.upload();
        uint8_t tab[128];
        memset(tab, 0, sizeof(tab));
        build_hdr_tables(tab, tab + 64);
        dHdrTables = (uint8_t*)rt_malloc(128);
        rt_h2d(dHdrTables, tab, 128);
        dOpStats = (long long*)rt_malloc(64 * 8);
        rt_memset(dOpStats, 0, 64 * 8);
        {
            std::vector<uint32_t> t(1024 + 32);
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = i;
                for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
                t[i] = c;
            }
            for (int k = 1; k < 4; k++)
                for (uint32_t i = 0; i < 256; i++) t[k * 256 + i] = (t[(k - 1) * 256 + i] >> 8) ^ t[t[(k - 1) * 256 + i] & 0xff];
            auto mul = [](uint32_t a, uint32_t b) {
                uint32_t m = 1u << 31, p = 0;
                for (;;) {
                    if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
                    m >>= 1;
                    b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
                }
                return p;
            };
            uint32_t p = 1u << 30;  // x^1
            t[1024] = p;
            for (int k = 1; k < 32; k++) t[1024 + k] = p = mul(p, p);
            dCrcTab = (uint32_t*)rt_malloc(t.size() * 4);
            rt_h2d(dCrcTab, t.data(), t.size() * 4);
        }
        rt_sync();
        slotsPerBlock = std::max(progDyn.nSlots, progFixed.nSlots);
        masksPerBlock = std::max(progDyn.nMasks, progFixed.nMasks);
        // the fused executor carves its per-block tables out of the same pools (d4g_fused.h: d4f_glob, d4f_eset)
        slotsPerBlock = std::max<int>(slotsPerBlock, 2 + (int)((D4F_GLOB_BYTES + sizeof(D4GState) - 1) / sizeof(D4GState)));
        masksPerBlock = std::max<int>(masksPerBlock, D4F_MAXM + 2 * D4F_MAXC);
        maxOps = (int)std::max(progDyn.ops.size(), progFixed.ops.size());
        if ((i64)maxOps * 64 >= (1LL << D4G_KEY_SEQ_BITS)) throw std::runtime_error("program too long for the key layout");
        ready = true;
    }
};
inline Engine& engine() {   // one per context (d4g_rt.h): the programs live in that device's memory
    static Engine e[RT_MAX_CTX];
    return e[rt_ctx()];
}

// host view of one block of a stream
struct HBlock {
    int type = 0;        // current type (a Huffman block may have become STORED)
    int gpu = -1;        // device block index (Huffman blocks and merge arenas)
    int homeGpu = -1;    // device block of the first parsed block this one covers: where a merged block is committed
    int ordinal = 0;     // that first block's position among the stream's parsed blocks
    i64 tokStart = 0, tokCount = 0, uStart = 0, uLen = 0;
    i64 refStart = 0, refCount = 0;   // back-reference records of the block's tokens
    i64 size = 0;        // Huffman: sizeBits of the current state
    std::vector<D4GRoundResult> chain;  // phase-1 optimiseBlock rounds
    i64 size_at(i64 alignment) const {  // getSizeBits(alignment)
        if (type != D4G_STORED) return size;
        i64 c = alignment % 8;
        c = c == 0 ? 0 : 8 - c;
        return (uLen + 4) * 8 + c;
    }
};

struct HStream {
    int status = 0;
    std::vector<HBlock> blocks;
    i64 consumed = 0, sizeBitsIn = 0, saved = 0;
    i64 inOff = 0, inLen = 0;
    i64 tokBase = 0, uBase = 0, nTok = 0, nU = 0, refBase = 0, nRef = 0;
    i64 outWordBase = 0, outBits = 0;
    int arena[2] = {-1, -1};
    i64 commitMaskBase = 0;   // mask words where finished merged blocks keep their final mask (see commit_block)
    // mergeBlocks state machine
    size_t mIdx = 0;
    i64 mPos = 0, mSaved = 0;
    bool mFirst = true, mDone = false, mWaiting = false;
    int mArenaUsed = -1;
};

struct Batch {
    std::vector<std::vector<uint8_t>> inputs;
    std::vector<HStream> streams;
    d4g_stats stats;
    // device
    uint8_t* dIn = nullptr;
    uint2* dTok = nullptr;
    uint4* dRefs = nullptr;       // back-reference records
    uint32_t* dTokRef = nullptr;  // token -> record index
    uint32_t* dBinStat = nullptr; // per block: static bin statistics (d4g_types.h)
    uint64_t* dBinMask = nullptr; // per block: bin record masks
    D4GHsMemo* dHsMemo = nullptr; // per block: header-search memo
    D4GRecodeMemo* dRcMemo = nullptr;  // per block: Huffman-rebuild memo
    uint64_t* dPassMemo = nullptr;     // per block: token-pass memo entries
    uint8_t* dU = nullptr;
    D4GBlock* dBlocks = nullptr;
    D4GState* dStates = nullptr;
    uint64_t* dMasks = nullptr;
    long long* dKeys = nullptr;
    int32_t* dErr = nullptr;       // device consistency counter of THIS batch (kernels add to it; checked after each phase)
    int32_t* dActive = nullptr;
    D4GRoundResult* dResults = nullptr;
    uint32_t* dOut = nullptr;
    int32_t* dReady = nullptr;   // per (block, slot): epoch of the round that produced it (persistent executor)
    unsigned* dHeads = nullptr;
    int epoch = 0;
    std::vector<D4GBlock> hBlocks;  // device block descriptors (host copy)
    std::vector<int> gpuType;       // current state type per device block
    i64 outWords = 0;
    bool ran = false;

    ~Batch() {
        try { rt_sync_all(); } catch (...) {}   // nothing may still be running on a block that goes back to the pool
        rt_free(dIn); rt_free(dTok); rt_free(dRefs); rt_free(dTokRef); rt_free(dBinStat); rt_free(dBinMask); rt_free(dHsMemo); rt_free(dRcMemo); rt_free(dPassMemo); rt_free(chunkPool.batches); rt_free(chunkPool.next); rt_free(dU); rt_free(dBlocks); rt_free(dStates);
        rt_free(dErr);
        rt_free(dMasks); rt_free(dKeys); rt_free(dActive); rt_free(dResults); rt_free(dOut); rt_free(dStreams); rt_free(dSrc); rt_free(dReady); rt_free(dHeads); rt_free(dClArena);
    }

    int32_t* errors() {
        if (!dErr) { dErr = (int32_t*)rt_malloc(16); rt_memset(dErr, 0, 16); }
        return dErr;
    }
    // What only the level / persistent executors use (candidate keys, slot epochs, queue heads, the three memo tables): made
    // on their first use — a batch the fused executor handles alone never allocates or clears them.
    size_t legacyBlocks = 0;
    long long legacyPassMemoWords = 0;
    void ensure_legacy_tables() {
        if (dKeys || !legacyBlocks) return;
        Engine& E = engine();
        const size_t nb = legacyBlocks;
        dKeys = (long long*)rt_malloc(nb * (size_t)E.maxOps * sizeof(long long));
        dReady = (int32_t*)rt_malloc(nb * (size_t)slotsAlloc * sizeof(int32_t));
        rt_memset(dReady, 0, nb * (size_t)slotsAlloc * sizeof(int32_t));
        dHeads = (unsigned*)rt_malloc(64);
        dHsMemo = (D4GHsMemo*)rt_malloc(nb * (size_t)D4G_HSMEMO_SLOTS * sizeof(D4GHsMemo));
        rt_memset(dHsMemo, 0, nb * (size_t)D4G_HSMEMO_SLOTS * sizeof(D4GHsMemo));
        dPassMemo = (uint64_t*)rt_malloc((size_t)legacyPassMemoWords * 8 + 64);
        rt_memset(dPassMemo, 0, (size_t)legacyPassMemoWords * 8 + 64);
        dRcMemo = (D4GRecodeMemo*)rt_malloc(nb * (size_t)D4G_RCMEMO_SLOTS * sizeof(D4GRecodeMemo));
        rt_memset(dRcMemo, 0, nb * (size_t)D4G_RCMEMO_SLOTS * sizeof(D4GRecodeMemo));
    }
    D4GCtx make_ctx(const Program& P, int nActive) {
        Engine& E = engine();
        D4GCtx c;
        c.tok = dTok; c.refs = dRefs; c.tokRef = dTokRef; c.binStat = dBinStat; c.binMask = dBinMask; c.hsMemo = dHsMemo; c.rcMemo = dRcMemo; c.passMemo = dPassMemo;
        if (const char* m = getenv("D4G_MEMO")) {   // D4G_MEMO=0: every op computes (the memos are an optimisation only)
            if (m[0] == '0') { c.hsMemo = nullptr; c.rcMemo = nullptr; c.passMemo = nullptr; }
        } c.U = dU; c.blocks = dBlocks; c.states = dStates; c.masks = dMasks;
        c.keys = dKeys; c.ops = P.dOps; c.hdrFlags = E.dHdrTables; c.hdrPrune = E.dHdrTables + 64;
        c.active = dActive; c.errors = errors(); c.opStats = E.dOpStats; c.nActive = nActive; c.nOps = (int)P.ops.size();
        c.slotsPerBlock = slotsAlloc; c.masksPerBlock = E.masksPerBlock;
        c.tileGroups = nActive > 0 ? (nActive + 7) / 8 : 1;
        return c;
    }

    // fromDevice: the `in` pointers are device addresses (another batch's outputs): chained stages stay in HBM
    void create(size_t n, const uint8_t* const* in, const size_t* len, bool fromDevice = false) {
        memset(&stats, 0, sizeof(stats));
        double t0 = now_ms();
        streams.resize(n);
        i64 off = 0;
        for (size_t i = 0; i < n; i++) {
            streams[i].inOff = off;
            streams[i].inLen = (i64)len[i];
            off += ((i64)len[i] + 15) & ~15LL;
            off += 16;
            stats.bytes_in += (i64)len[i];
        }
        i64 total = off + D4G_INCH + 64;
        dIn = (uint8_t*)rt_malloc((size_t)total);
        rt_memset(dIn, 0, (size_t)total);
        for (size_t i = 0; i < n; i++) {
            if (fromDevice) rt_d2d(dIn + streams[i].inOff, in[i], len[i]);
            else rt_h2d(dIn + streams[i].inOff, in[i], len[i]);
        }
        rt_sync();
        stats.n_streams = (i64)n;
        stats.ms_upload = now_ms() - t0;
    }

    // ---- parse: header scan -> block probes -> chain -> emit -> pointer jumping ----
    struct PBlock { int type, bfinal; i64 bitPos, endBit, nTok, uLen, sizeBits, nRef; int firstBatch; i64 refSpan = -1; };   // refSpan: records the block occupies in refs even when it is STORED (LZ77 front end)
    struct PStream { int status = 0; std::vector<PBlock> blocks; i64 nTok = 0, nU = 0, consumed = 0, sizeBits = 0; i64 uBaseFixed = -1; };   // uBaseFixed: the decoded bytes already sit in U (LZ77 front end: the raw input)
    std::vector<PStream> ps;
    D4GStreamDesc* dStreams = nullptr;
    D4GChunkPool chunkPool = {nullptr, nullptr, 0};   // the probe's verified chunk starts, replayed by the emit pass
    uint32_t* dSrc = nullptr;
    int slotsAlloc = 0;
    double msParseKernels = 0;

    // Steps 1-2 + host chain walk: fills `ps` (block list per stream, exact token/byte counts).
    void parse_probe() {
        size_t n = streams.size();
        ps.assign(n, PStream());
        std::vector<D4GStreamDesc> sd(n);
        std::vector<D4GScanTile> tiles;
        i64 totalBytes = 0;
        for (size_t i = 0; i < n; i++) {
            sd[i].data = dIn + streams[i].inOff;
            sd[i].len = streams[i].inLen;
            sd[i].uBase = 0;
            sd[i].uLen = 0;
            for (i64 b = 0; b < streams[i].inLen; b += D4G_SCAN_TILE) tiles.push_back({(int32_t)i, 0, b});
            totalBytes += streams[i].inLen;
        }
        dStreams = (D4GStreamDesc*)rt_malloc(n * sizeof(D4GStreamDesc));
        rt_h2d(dStreams, sd.data(), n * sizeof(D4GStreamDesc));
        RtEvent e0, e1;
        e0.record();
        // 1. scan
        std::vector<D4GProbeIn> cands;
        std::vector<D4GProbeOut> pout;
        if (!tiles.empty()) {
            D4GScanTile* dTiles = (D4GScanTile*)rt_malloc(tiles.size() * sizeof(D4GScanTile));
            rt_h2d(dTiles, tiles.data(), tiles.size() * sizeof(D4GScanTile));
            unsigned cap = (unsigned)std::max<i64>(65536, totalBytes / 4);
            unsigned* dN = (unsigned*)rt_malloc(4);
            D4GProbeIn* dCands = nullptr;
            unsigned nc = 0;
            for (int attempt = 0; attempt < 2; attempt++) {
                rt_free(dCands);
                dCands = (D4GProbeIn*)rt_malloc((size_t)cap * sizeof(D4GProbeIn));
                rt_memset(dN, 0, 4);
                RT_LAUNCH(k_scan_headers, tiles.size(), 256, dStreams, dTiles, dCands, dN, cap);
                stats.kernel_launches++;
                rt_d2h(&nc, dN, 4);
                if (nc <= cap) break;
                cap = nc + 1024;
            }
            // 1b. header pre-filter (one lane per candidate) — 2. speculative probes of the survivors; only the candidates
            // that parse come back
            stats.scan_candidates = (i64)nc;
            if (nc) {
                D4GProbeIn* dKept = (D4GProbeIn*)rt_malloc((size_t)nc * sizeof(D4GProbeIn));
                rt_memset(dN, 0, 4);
                RT_LAUNCH(k_prefilter_headers, (nc + 63) / 64, 64, dStreams, dCands, nc, dKept, dN);
                stats.kernel_launches++;
                rt_d2h(&nc, dN, 4);
                rt_free(dCands);
                dCands = dKept;
            }
            if (nc) {
                D4GProbeHit* dHits = (D4GProbeHit*)rt_malloc((size_t)nc * sizeof(D4GProbeHit));
                rt_memset(dN, 0, 4);
                chunkPool.cap = (unsigned)std::min<i64>(1 << 30, totalBytes * 8 / (64 * D4G_CHUNK_BITS) + 2 * (i64)nc * (parse_threads() / 64) + 64);   // one record per wave and batch
                chunkPool.batches = (D4GChunkBatch*)rt_malloc((size_t)chunkPool.cap * sizeof(D4GChunkBatch));
                chunkPool.next = (unsigned*)rt_malloc(16);
                rt_memset(chunkPool.next, 0, 16);
                RT_LAUNCH(k_probe_blocks, nc, parse_threads(), dStreams, dCands, (D4GProbeOut*)nullptr, nc, dHits, dN, chunkPool);
                stats.kernel_launches++;
                unsigned nh = 0;
                rt_d2h(&nh, dN, 4);
                std::vector<D4GProbeHit> hits(nh);
                rt_d2h(hits.data(), dHits, (size_t)nh * sizeof(D4GProbeHit));
                rt_free(dHits);
                cands.resize(nh);
                pout.resize(nh);
                for (unsigned k = 0; k < nh; k++) { cands[k] = hits[k].in; pout[k] = hits[k].out; }
            }
            rt_free(dCands); rt_free(dN); rt_free(dTiles);
        }
        // candidate maps: bit position -> probe result
        std::vector<std::vector<std::pair<i64, int>>> byStream(n);
        for (size_t k = 0; k < cands.size(); k++)
            if (pout[k].status == 0) { byStream[cands[k].stream].push_back({cands[k].bitPos, (int)k}); stats.scan_confirmed++; }
        for (auto& v : byStream) std::sort(v.begin(), v.end());
        // chain walk; positions the scan cannot see (fixed / stored / unusual dynamic blocks) are probed exactly
        std::vector<i64> cur(n, 0), upos(n, 0), spos(n, 0);
        std::vector<char> done(n, 0);
        D4GProbeIn* dEx = (D4GProbeIn*)rt_malloc(n * sizeof(D4GProbeIn) + 16);
        D4GProbeOut* dExOut = (D4GProbeOut*)rt_malloc(n * sizeof(D4GProbeOut) + 16);
        auto accept = [&](size_t i, i64 bitPos, const D4GProbeOut& o, bool fromScan) {
            PStream& P = ps[i];
            if (o.status != 0 || o.needHist > upos[i]) { P.status = -1; done[i] = 1; return; }
            P.blocks.push_back({o.type, o.bfinal, bitPos, o.endBit, o.nTok, o.uLen, o.sizeBits, (i64)o.nRef, fromScan ? o.firstBatch : -1});
            upos[i] += o.uLen;
            P.nTok += o.nTok;
            spos[i] += 3;  // DeflateStream.getSizeBits — :171-182
            if (o.type == D4G_STORED) {
                i64 c = spos[i] % 8;
                c = c == 0 ? 0 : 8 - c;
                spos[i] += (o.uLen + 4) * 8 + c;
            } else {
                spos[i] += o.sizeBits;
            }
            cur[i] = o.endBit;
            if (o.bfinal) { done[i] = 1; P.consumed = (o.endBit + 7) / 8; }
            else if (o.eofHit) { P.status = -1; done[i] = 1; }  // the next 3-bit read hits EOF
        };
        while (true) {
            std::vector<D4GProbeIn> ex;
            std::vector<size_t> exStream;
            for (size_t i = 0; i < n; i++) {
                while (!done[i]) {
                    auto& v = byStream[i];
                    auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(cur[i], -1));
                    if (it != v.end() && it->first == cur[i]) accept(i, cur[i], pout[it->second], true);
                    else { ex.push_back({(int32_t)i, 0, cur[i]}); exStream.push_back(i); break; }
                }
            }
            if (ex.empty()) break;
            rt_h2d(dEx, ex.data(), ex.size() * sizeof(D4GProbeIn));
            RT_LAUNCH(k_probe_blocks, ex.size(), parse_threads(), dStreams, dEx, dExOut, (unsigned)ex.size(), (D4GProbeHit*)nullptr, (unsigned*)nullptr,
                      D4GChunkPool{nullptr, nullptr, 0u});
            stats.kernel_launches++;
            stats.exact_probes += (i64)ex.size();
            std::vector<D4GProbeOut> eo(ex.size());
            rt_d2h(eo.data(), dExOut, ex.size() * sizeof(D4GProbeOut));
            for (size_t k = 0; k < ex.size(); k++) accept(exStream[k], ex[k].bitPos, eo[k], false);
        }
        rt_free(dEx); rt_free(dExOut);
        e1.record();
        msParseKernels += rt_elapsed_ms(e0, e1);
        for (size_t i = 0; i < n; i++) {
            ps[i].nU = upos[i];
            ps[i].sizeBits = spos[i];
            if (ps[i].status != 0) { ps[i].blocks.clear(); ps[i].nTok = 0; ps[i].nU = 0; }
        }
    }

    // ---- device block table: host block lists, device descriptors and every per-block array, from `ps` ----
    struct Layout {
        std::vector<int32_t> realBlocks;   // device blocks that come straight from the parse (not merge arenas)
        std::vector<D4GEmitIn> emits;
        std::vector<D4GTokRange> ranges;
    };
    void layout_blocks(bool merge, bool needSlots, Layout& LY) {
        Engine& E = engine();
        size_t n = streams.size();
        slotsAlloc = needSlots ? E.slotsPerBlock : 1;
        int masksAlloc = needSlots ? E.masksPerBlock : 1;
        hBlocks.clear();
        gpuType.clear();
        i64 maskWordsTotal = 0, tokTot = 0, uTot = 0, refTot = 0, binMaskWords = 0, passMemoWords = 0;
        std::vector<int32_t>& realBlocks = LY.realBlocks;
        std::vector<D4GStreamDesc> sd(n);
        std::vector<D4GEmitIn>& emits = LY.emits;
        std::vector<D4GTokRange>& ranges = LY.ranges;
        auto add_block = [&](int stream, i64 tokStart, i64 tokCount, i64 refStart, i64 refCount, i64 uStart, i64 uLen, i64 maskWordsCap,
                             int type) {
            D4GBlock b;
            memset(&b, 0, sizeof(b));
            b.type = type;
            b.stream = stream;
            b.tokStart = tokStart;
            b.tokCount = tokCount;
            b.uBase = streams[stream].uBase;
            b.uStart = uStart;
            b.uLen = uLen;
            b.stateIdx = (i64)hBlocks.size() * slotsAlloc;
            b.maskBase = maskWordsTotal;
            b.maskWords = (refCount + 63) / 64;
            b.refStart = refStart;
            b.refCount = refCount;
            b.binStat = needSlots ? (i64)hBlocks.size() * D4G_NBINS * D4G_BINSTRIDE : -1;
            b.binMask = binMaskWords;
            if (needSlots) binMaskWords += (i64)D4G_NBINS * maskWordsCap;
            b.passMemo = needSlots ? passMemoWords : -1;
            b.passMemoStride = D4G_PASSMEMO_HDR_WORDS + 2 * maskWordsCap;   // header + key codes, outgoing mask, incoming mask (key)
            if (needSlots) passMemoWords += (i64)D4G_PASSMEMO_SLOTS * b.passMemoStride;
            maskWordsTotal += maskWordsCap * masksAlloc;
            hBlocks.push_back(b);
            gpuType.push_back(type);
            return (int)hBlocks.size() - 1;
        };
        for (size_t si = 0; si < n; si++) {
            HStream& s = streams[si];
            const PStream& P = ps[si];
            s.status = P.status;
            s.consumed = P.consumed;
            s.sizeBitsIn = P.sizeBits;
            s.nTok = P.nTok;
            s.nU = P.nU;
            s.tokBase = tokTot;
            s.refBase = refTot;
            s.uBase = P.uBaseFixed >= 0 ? P.uBaseFixed : uTot;
            sd[si].data = dIn ? dIn + s.inOff : nullptr;
            sd[si].len = s.inLen;
            sd[si].uBase = s.uBase;
            sd[si].uLen = P.nU;
            tokTot += P.nTok;
            if (P.uBaseFixed < 0) uTot += (P.nU + 15) & ~15LL;
            if (P.status != 0) continue;
            int nHuff = 0;
            i64 tpos = 0, upos = 0, rpos = 0;
            for (const PBlock& pb : P.blocks) {
                HBlock hb;
                hb.type = pb.type;
                hb.tokStart = s.tokBase + tpos;
                hb.tokCount = pb.nTok;
                hb.uStart = upos;
                hb.uLen = pb.uLen;
                hb.refStart = s.refBase + rpos;
                hb.refCount = pb.type == D4G_STORED ? 0 : pb.nRef;
                hb.size = pb.sizeBits;
                D4GEmitIn em;
                memset(&em, 0, sizeof(em));
                em.stream = (int32_t)si;
                em.type = pb.type;
                em.bitPos = pb.bitPos;
                em.tokStart = hb.tokStart;
                em.uStart = upos;
                em.uLen = pb.uLen;
                em.stateIdx = -1;
                em.sizeBits = pb.sizeBits;
                em.refStart = hb.refStart;
                em.firstBatch = pb.firstBatch;
                if (pb.type != D4G_STORED) {
                    hb.gpu = add_block((int)si, hb.tokStart, hb.tokCount, hb.refStart, hb.refCount, hb.uStart, hb.uLen, (hb.refCount + 63) / 64,
                                       pb.type);
                    em.stateIdx = hBlocks[hb.gpu].stateIdx;
                    hb.homeGpu = hb.gpu;
                    realBlocks.push_back(hb.gpu);
                    nHuff++;
                }
                hb.ordinal = (int)s.blocks.size();
                emits.push_back(em);
                ranges.push_back({(int32_t)si, pb.type == D4G_STORED ? 1 : 0, hb.tokStart, hb.tokCount, upos, pb.uLen});
                s.blocks.push_back(hb);
                tpos += pb.nTok;
                upos += pb.uLen;
                rpos += pb.refSpan >= 0 ? pb.refSpan : hb.refCount;
                stats.n_blocks++;
            }
            s.nRef = rpos;
            refTot += rpos;
            stats.n_tokens += P.nTok;
            stats.bytes_decoded += P.nU;
            if (merge && needSlots && nHuff >= 2) {
                // (mask slots of an arena start on 128-byte lines and are whole lines long: the cluster kernel's workgroups hand
                // mask words to each other and must never share a line between a slot already read and one still to be written)
                for (int a = 0; a < 2; a++) {
                    maskWordsTotal = (maskWordsTotal + 15) & ~15LL;
                    binMaskWords = (binMaskWords + 15) & ~15LL;
                    s.arena[a] = add_block((int)si, s.tokBase, 0, s.refBase, 0, 0, 0, (((s.nRef + 63) / 64 + 1) + 15) & ~15LL, D4G_FIXED);
                }
                // a finished merged block moves out of its arena (the two arenas are re-used by the next chain of merges)
                s.commitMaskBase = maskWordsTotal;
                maskWordsTotal += (s.nRef + 63) / 64 + (i64)P.blocks.size() + 2;
            }
        }
        size_t nb = hBlocks.size();
        if (!dStreams) dStreams = (D4GStreamDesc*)rt_malloc(n * sizeof(D4GStreamDesc) + 16);
        rt_h2d(dStreams, sd.data(), n * sizeof(D4GStreamDesc));
        if (refTot >= (1LL << 32)) throw std::runtime_error("batch holds 2^32 or more back-references: split it");
        uTotal = uTot;
        // (the LZ77 front end has filled tok / refs / tokRef / U already, with the same numbering)
        if (!dTok) dTok = (uint2*)rt_malloc((size_t)tokTot * 8 + 64);
        if (!dRefs) dRefs = (uint4*)rt_malloc((size_t)refTot * 16 + 64);
        if (!dTokRef) dTokRef = (uint32_t*)rt_malloc((size_t)tokTot * 4 + 64);
        if (!dU) dU = (uint8_t*)rt_malloc((size_t)uTot + 64);
        if (nb) {
            dBlocks = (D4GBlock*)rt_malloc(nb * sizeof(D4GBlock));
            rt_h2d(dBlocks, hBlocks.data(), nb * sizeof(D4GBlock));
            dStates = (D4GState*)rt_malloc(nb * (size_t)slotsAlloc * sizeof(D4GState));
            dMasks = (uint64_t*)rt_malloc((size_t)maskWordsTotal * 8 + 64);
            legacyBlocks = needSlots ? nb : 0;
            legacyPassMemoWords = passMemoWords;
            if (needSlots && !exec_fused()) ensure_legacy_tables();   // (the fused executor's batches make them when a block first falls back)
            dActive = (int32_t*)rt_malloc(nb * sizeof(int32_t));
            dResults = (D4GRoundResult*)rt_malloc(nb * sizeof(D4GRoundResult));
            // mask 0 of every block starts empty (no back-reference expanded); the writer reads it even when no search runs
            rt_memset(dMasks, 0, (size_t)maskWordsTotal * 8 + 64);   // one fill instead of one per block
            if (needSlots) {
                dBinStat = (uint32_t*)rt_malloc(nb * (size_t)D4G_NBINS * D4G_BINSTRIDE * 4);
                dBinMask = (uint64_t*)rt_malloc((size_t)binMaskWords * 8 + 64);
                rt_memset(dBinStat, 0, nb * (size_t)D4G_NBINS * D4G_BINSTRIDE * 4);
                rt_memset(dBinMask, 0, (size_t)binMaskWords * 8 + 64);
            }
        }
    }

    // ---- steps 3-5 of the parse: emit tokens/states, resolve decoded bytes, bin statistics ----
    i64 uTotal = 0;
    void build_blocks(bool merge, bool needSlots) {
        Engine& E = engine();
        size_t n = streams.size();
        Layout LY;
        layout_blocks(merge, needSlots, LY);
        std::vector<D4GEmitIn>& emits = LY.emits;
        std::vector<D4GTokRange>& ranges = LY.ranges;
        dSrc = (uint32_t*)rt_malloc((size_t)uTotal * 4 + 64);
        RtEvent e0, e1;
        e0.record();
        int32_t* dBadFlags = nullptr;     // per stream: a back-reference reached before the start of the stream
        std::vector<void*> later;         // device buffers the queued kernels still read: freed after the wait below
        if (!emits.empty()) {
            // 3. emit
            D4GEmitIn* dEm = (D4GEmitIn*)rt_malloc(emits.size() * sizeof(D4GEmitIn));
            rt_h2d(dEm, emits.data(), emits.size() * sizeof(D4GEmitIn));
            D4GParseOut po = {dTok, dU, dStates, dRefs, dTokRef};
            RT_LAUNCH(k_emit_blocks, emits.size(), parse_threads(), dStreams, dEm, po, errors(), chunkPool);
            stats.kernel_launches++;
            // 4. decoded bytes
            D4GTokRange* dRanges = (D4GTokRange*)rt_malloc(ranges.size() * sizeof(D4GTokRange));
            rt_h2d(dRanges, ranges.data(), ranges.size() * sizeof(D4GTokRange));
            int32_t* dBad = (int32_t*)rt_malloc(n * 4 + 16);
            rt_memset(dBad, 0, n * 4 + 16);
            const int GF = 8;
            RT_LAUNCH(k_fill_src, ranges.size() * GF, 256, dStreams, dRanges, dTok, dU, dSrc, dBad, GF);
            stats.kernel_launches++;
            i64 maxU = 0;
            for (size_t i = 0; i < n; i++) maxU = std::max(maxU, ps[i].nU);
            if (maxU >= (1LL << 31)) throw std::runtime_error("a stream decodes to 2 GiB or more");
            int G = (int)std::min<i64>(2048, std::max<i64>(1, (maxU + 4095) / 4096));
            unsigned long long* dChanged = (unsigned long long*)rt_malloc(40 * 8);   // one counter per round, zeroed once
            rt_memset(dChanged, 0, 40 * 8);
            i64 totalU = 0;
            for (size_t i = 0; i < n; i++) totalU += ps[i].nU;
            static int stopPct = -1;   // D4G_JUMP_STOP_PCT: stop doubling once fewer than this share of the bytes still moves
            if (stopPct < 0) { const char* t = getenv("D4G_JUMP_STOP_PCT"); stopPct = t ? atoi(t) : 50; }
            const unsigned long long stopNum = std::max<unsigned long long>(1, (unsigned long long)((totalU * stopPct + 99) / 100));   // the resolve pass walks what is left of the chains
            // the first rounds tile by tile, out of the XCDs' L2 (k_jump_tiles); D4G_JUMP_TILE_REPS=0: plain rounds only
            static const int tileReps = env_int("D4G_JUMP_TILE_REPS", 6);
            void* jumpTmp[2] = {nullptr, nullptr};
            if (tileReps > 0) {
                std::vector<D4GJumpTile> tl;
                for (size_t i = 0; i < n; i++)
                    for (i64 q = 0; q < ps[i].nU; q += D4G_JUMP_TILE) tl.push_back({(int32_t)i, 0, q});
                if (!tl.empty()) {
                    // consecutive tiles -> workgroup ids equal mod 8 (one XCD, one L2): region x of the tile list goes to ids x, x + 8, ...
                    const size_t nt = tl.size(), per = (nt + 7) / 8;
                    std::vector<D4GJumpTile> ord(nt);
                    size_t w = 0;
                    for (size_t j = 0; j < per; j++)
                        for (size_t x = 0; x < 8; x++) {
                            const size_t t = x * per + j;
                            if (t < nt) ord[w++] = tl[t];
                        }
                    D4GJumpTile* dTl = (D4GJumpTile*)rt_malloc(nt * sizeof(D4GJumpTile));
                    rt_h2d(dTl, ord.data(), nt * sizeof(D4GJumpTile));
                    unsigned long long* dCh0 = (unsigned long long*)rt_malloc(16);
                    rt_memset(dCh0, 0, 16);
#ifdef D4G_HOSTSIM
                    RT_LAUNCH(k_jump_tiles, nt, 256, dStreams, dTl, dSrc, tileReps, dCh0);
#else
                    {
                        // 1024 threads per tile and 70 KiB of LDS the kernel never touches: at most two tiles per CU (one beside a
                        // resident search workgroup), so the tiles of an XCD's CUs and their neighbours stay in its L2 over the
                        // rounds (2.89 -> 2.66 ms of parse kernels at one tile per CU, 100 KiB; 70 KiB is what several batches in
                        // flight like best: 12.5-13.0 -> 13.3-13.4 GB/s on config 2)
                        static const int jt = env_int("D4G_JUMP_THREADS", 1024), jl = env_int("D4G_JUMP_LDS_KB", 70);
                        if (jl > 64) {   // (more than 64 KiB of dynamic LDS has to be allowed once per device)
                            static std::atomic<unsigned long long> allowed{0};
                            int dev = 0;
                            RT_CHECK(hipGetDevice(&dev));
                            const unsigned long long bit = 1ull << (dev & 63);
                            if (!(allowed.load() & bit)) {
                                RT_CHECK(hipFuncSetAttribute((const void*)k_jump_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, jl * 1024));
                                allowed.fetch_or(bit);
                            }
                        }
                        hipLaunchKernelGGL(k_jump_tiles, dim3((unsigned)nt), dim3((unsigned)jt), (size_t)jl * 1024, rt().sa(), dStreams, dTl, dSrc, tileReps, dCh0);
                        RT_CHECK(hipGetLastError());
                    }
#endif
                    stats.kernel_launches++;
                    jumpTmp[0] = dTl; jumpTmp[1] = dCh0;   // (freed with the other parse buffers, after the next synchronisation)
                }
            }
            const int JB = tileReps > 0 ? 4 : 10;   // rounds per batch: launched back to back, counters read once (after the tile rounds one or two are left)
            for (int base = 0; base < 40; base += JB) {
                for (int round = base; round < base + JB; round++) {
                    RT_LAUNCH(k_jump_streams, n * (size_t)G, 256, dStreams, dSrc, dChanged + round, G,
                              round == 0 ? (const unsigned long long*)nullptr : dChanged + round - 1, stopNum);
                    stats.kernel_launches++;
                }
                unsigned long long ch[10];
                rt_d2h(ch, dChanged + base, JB * 8);
                if (getenv("D4G_DEBUG_JUMP")) {
                    fprintf(stderr, "jump rounds %d..%d of %lld bytes, moved:", base, base + JB - 1, (long long)totalU);
                    for (int k = 0; k < JB; k++) fprintf(stderr, " %llu", ch[k]);
                    fprintf(stderr, "\n");
                }
                bool done = false;
                for (int k = 0; k < JB; k++) {
                    stats.jump_rounds++;                   // round base + k ran (its predecessor moved enough)
                    if (ch[k] < stopNum) { done = true; break; }
                }
                if (done) break;
            }
            RT_LAUNCH(k_resolve_streams, n * (size_t)G, 256, dStreams, dSrc, dU, G);
            stats.kernel_launches++;
            // (no wait here: the bin statistics follow on the same stream; the flags come back behind them, one wait for both)
            dBadFlags = dBad;
            later = {dEm, dRanges, dChanged, jumpTmp[0], jumpTmp[1]};
        }
        later.push_back(block_bins(LY.realBlocks, needSlots));
        e1.record();
        if (dBadFlags) {
            std::vector<int32_t> bad(n);
            rt_d2h(bad.data(), dBadFlags, n * 4);
            rt_free(dBadFlags);
            for (void* q : later) rt_free(q);
            for (size_t i = 0; i < n; i++)
                if (bad[i]) throw std::runtime_error("parse: back-reference before the start of stream (host check missed it)");
        } else {
            rt_sync();
            for (void* q : later) rt_free(q);
        }
        msParseKernels += rt_elapsed_ms(e0, e1);
        rt_free(dSrc);
        dSrc = nullptr;
        rt_free(chunkPool.batches); rt_free(chunkPool.next);
        chunkPool = {nullptr, nullptr, 0};
        check_device_errors();
    }
    // 5. static bin statistics of every block's back-reference records (the least-expensive pass works from them);
    //    also fills in the records' first decoded bytes
    // (queued, not waited for: the caller frees the returned list after its next wait on the stream)
    void* block_bins(const std::vector<int32_t>& realBlocks, bool needSlots) {
        if (!needSlots || realBlocks.empty()) return nullptr;
        int32_t* dReal = (int32_t*)rt_malloc(realBlocks.size() * 4);
        rt_h2d(dReal, realBlocks.data(), realBlocks.size() * 4);
        D4GCtx c = make_ctx(engine().progDyn, 0);
        RT_LAUNCH(k_block_bins, realBlocks.size() * D4G_BINS_SPLIT, 256, c, dReal);
        stats.kernel_launches++;
        return dReal;
    }

    // One optimiseBlock call on every block of `act` (device block indices): runs the program
    // matching each block's current type and returns the per-block results in `act` order.
    double msSearch = 0;
    // One optimiseBlock call per block of `act`: the fused executor takes the blocks it can hold, the level / persistent
    // executors the rest.
    std::vector<D4GRoundResult> run_round(const std::vector<int>& act) {
        if (!exec_fused()) return run_round_legacy(act);
        std::vector<D4GRoundResult> res(act.size());
        std::vector<int> small, big;
        std::vector<size_t> smallPos, bigPos;
        // Blocks of up to 16384 back-references: one workgroup each (fused).  Longer ones: when there are many of them they
        // still fill the device one workgroup each; a few long blocks get the whole device one after the other (cluster).
        size_t nLong = 0;
        for (int k : act) nLong += hBlocks[k].refCount > (1LL << 14);
        for (size_t i = 0; i < act.size(); i++) {
            if (hBlocks[act[i]].refCount <= fused_max_refs(nLong)) { small.push_back(act[i]); smallPos.push_back(i); }
            else { big.push_back(act[i]); bigPos.push_back(i); }
        }
        if (!small.empty()) {
            std::vector<std::vector<D4GRoundResult>> ch = run_fused(small, 1);
            for (size_t i = 0; i < small.size(); i++) res[smallPos[i]] = ch[i].at(0);
        }
        if (!big.empty() && cluster_enabled()) {   // long merged blocks, one launch of the whole device each
            std::vector<int> rest;
            std::vector<size_t> restPos;
            for (size_t i = 0; i < big.size(); i++) {
                const D4GBlock& d = hBlocks[big[i]];
                D4GRoundResult r;
                if (d.refCount > cluster_min_refs() && (d.maskBase & 15) == 0 && (d.maskWords & 15) == 0 && (d.binMask & 15) == 0 && run_cluster(big[i], &r)) res[bigPos[i]] = r;
                else { rest.push_back(big[i]); restPos.push_back(bigPos[i]); }
            }
            big.swap(rest);
            bigPos.swap(restPos);
        }
        if (!big.empty()) {
            std::vector<D4GRoundResult> r = run_round_legacy(big);
            for (size_t i = 0; i < big.size(); i++) res[bigPos[i]] = r[i];
        } else {
            stats.rounds++;
        }
        return res;
    }
    // One optimiseBlock round of one long block with every workgroup of the device (k_search_cluster).  false: the round did
    // not fit the kernel's tables — the block is untouched and the caller uses another executor.
    D4FClArena* dClArena = nullptr;
    bool run_cluster(int blk, D4GRoundResult* out) {
        Engine& E = engine();
        rt().cur = 0;
        if (!dClArena) dClArena = (D4FClArena*)rt_malloc(sizeof(D4FClArena));
        rt_memset(dClArena, 0, 128);   // the epoch counter; every command slot is cleared by the control workgroup before use
        int32_t one = blk;
        rt_h2d(dActive, &one, sizeof(one));
        D4GRoundResult* dRes = (D4GRoundResult*)rt_malloc(D4F_MAXROUNDS * sizeof(D4GRoundResult));
        int32_t* dInfo = (int32_t*)rt_malloc(16);
        D4GCtx c = make_ctx(E.progDyn, 1);
        D4FParams P;
        memset(&P, 0, sizeof(P));
        P.ops[0] = E.progDyn.dOps; P.ops[1] = E.progFixed.dOps;
        P.nOps[0] = (int)E.progDyn.ops.size(); P.nOps[1] = (int)E.progFixed.ops.size();
        P.maxRounds = 1;
        P.regWords = env_int("D4G_FUSED_REG_WORDS", 64 * D4F_NWR);
        P.results = dRes;
        P.roundInfo = dInfo;
        P.stats = getenv("D4G_FUSED_STATS") ? E.dOpStats : nullptr;
#ifdef D4G_HOSTSIM
        const int wgs = 3, threads = std::max(128, state_block());
#else
        static const int wgs = env_int("D4G_CLUSTER_WGS", device_cus());
        const int threads = 512;
#endif
        RtEvent e0, e1;
        e0.record();
        RT_LAUNCH(k_search_cluster, wgs, threads, c, P, dClArena, 0);
        e1.record();
        stats.kernel_launches++;
        stats.state_launches++;
        int32_t info = 0;
        rt_d2h(&info, dInfo, sizeof(info));
        D4GRoundResult r;
        rt_d2h(&r, dRes, sizeof(r));
        const float ms = rt_elapsed_ms(e0, e1);
        msSearch += ms;
        stats.ms_state_kernels += ms;
        stats.state_tokens_per_round += hBlocks[blk].tokCount;
        stats.state_bytes_per_round += hBlocks[blk].uLen;
        rt_free(dRes); rt_free(dInfo);
        if (getenv("D4G_DEBUG_ROUNDS")) fprintf(stderr, "cluster search: block of %lld back-references, %.3f ms%s\n", (long long)hBlocks[blk].refCount, ms, (info & D4F_INFO_FALLBACK) ? " (did not fit)" : "");
        if ((info & D4F_INFO_FALLBACK) || (info & 0xffff) < 1) return false;
        gpuType[blk] = r.newType;
        stats.rounds_fused += 1;
        stats.rounds_cluster += 1;
        *out = r;
        return true;
    }
    // Fused executor (k_search_fused): every block of `act` runs up to maxRounds optimiseBlock rounds, while it keeps
    // improving, inside one workgroup.  Returns each block's chain of round results.  A round that does not fit the
    // kernel's tables comes back untouched and is run by the level executor; the block then goes on here.
    std::vector<std::vector<D4GRoundResult>> run_fused(const std::vector<int>& act, int maxRounds) {
        Engine& E = engine();
        std::vector<std::vector<D4GRoundResult>> chains(act.size());
        std::vector<int> todo(act.size());
        for (size_t i = 0; i < act.size(); i++) todo[i] = (int)i;
        D4GRoundResult* dRes = nullptr;
        int32_t* dInfo = nullptr;
        while (!todo.empty()) {
            const int nA = (int)todo.size();
            std::vector<int32_t> sub(nA);
            for (int i = 0; i < nA; i++) sub[i] = act[todo[i]];
            rt().cur = 0;
            rt_h2d(dActive, sub.data(), sub.size() * sizeof(int32_t));
            if (!dRes) {
                dRes = (D4GRoundResult*)rt_malloc(act.size() * (size_t)D4F_MAXROUNDS * sizeof(D4GRoundResult));
                dInfo = (int32_t*)rt_malloc(act.size() * sizeof(int32_t) + 16);
            }
            D4GCtx c = make_ctx(E.progDyn, nA);
            D4FParams P;
            memset(&P, 0, sizeof(P));
            P.ops[0] = E.progDyn.dOps; P.ops[1] = E.progFixed.dOps;
            P.nOps[0] = (int)E.progDyn.ops.size(); P.nOps[1] = (int)E.progFixed.ops.size();
            P.regWords = env_int("D4G_FUSED_REG_WORDS", 64 * D4F_NWR);
            P.results = dRes;
            P.roundInfo = dInfo;
            P.stats = getenv("D4G_FUSED_STATS") ? E.dOpStats : nullptr;
            std::vector<int> left(nA);
            int cap = 0;
            for (int i = 0; i < nA; i++) { left[i] = maxRounds - (int)chains[todo[i]].size(); cap = std::max(cap, left[i]); }
            P.maxRounds = std::min(cap, (int)D4F_MAXROUNDS);
            RtEvent e0, e1;
            e0.record();
#ifdef D4G_HOSTSIM
            const int fusedBlock = std::max(128, state_block());
#else
            static const int fusedBlock = env_int("D4G_FUSED_BLOCK", 512);
#endif
            RT_LAUNCH(k_search_fused, nA, fusedBlock, c, P);
            e1.record();
            stats.kernel_launches++;
            stats.state_launches++;
            std::vector<int32_t> info(nA);
            rt_d2h(info.data(), dInfo, (size_t)nA * sizeof(int32_t));
            std::vector<D4GRoundResult> r((size_t)nA * D4F_MAXROUNDS);
            rt_d2h(r.data(), dRes, r.size() * sizeof(D4GRoundResult));
            const float ms = rt_elapsed_ms(e0, e1);
            msSearch += ms;
            stats.ms_state_kernels += ms;
            for (int k : sub) { stats.state_tokens_per_round += hBlocks[k].tokCount; stats.state_bytes_per_round += hBlocks[k].uLen; }
            if (getenv("D4G_DEBUG_ROUNDS")) fprintf(stderr, "fused search: %d blocks, up to %d rounds, %.3f ms\n", nA, P.maxRounds, ms);
            std::vector<int> next, fb;
            for (int i = 0; i < nA; i++) {
                const int n = info[i] & 0xffff;
                std::vector<D4GRoundResult>& ch = chains[todo[i]];
                for (int k = 0; k < n; k++) ch.push_back(r[(size_t)i * D4F_MAXROUNDS + k]);
                if (n) gpuType[sub[i]] = ch.back().newType;
                stats.rounds_fused += n;
                if (info[i] & D4F_INFO_FALLBACK) fb.push_back(todo[i]);
                else if ((info[i] & D4F_INFO_MORE) && (int)ch.size() < maxRounds) next.push_back(todo[i]);
            }
            if (!fb.empty()) {   // one round with the level executor, then back here if it improved
                std::vector<int> fbAct(fb.size());
                for (size_t i = 0; i < fb.size(); i++) fbAct[i] = act[fb[i]];
                std::vector<D4GRoundResult> rr = run_round_legacy(fbAct);
                stats.fused_fallbacks += (int64_t)fb.size();
                for (size_t i = 0; i < fb.size(); i++) {
                    chains[fb[i]].push_back(rr[i]);
                    if (rr[i].improved && (int)chains[fb[i]].size() < maxRounds) next.push_back(fb[i]);
                }
            }
            std::sort(next.begin(), next.end());
            todo.swap(next);
        }
        rt_free(dRes); rt_free(dInfo);
        return chains;
    }
    bool forceLevels = false;   // the round in hand fell back from the persistent executor
    std::vector<D4GRoundResult> run_round_legacy(const std::vector<int>& act) {
        Engine& E = engine();
        ensure_legacy_tables();
        std::vector<D4GRoundResult> res(act.size());
        for (int pass = 0; pass < 2; pass++) {
            const Program& P = pass == 0 ? E.progDyn : E.progFixed;
            int wantType = pass == 0 ? D4G_DYNAMIC : D4G_FIXED;
            std::vector<int32_t> sub;
            std::vector<size_t> subPos;
            for (size_t i = 0; i < act.size(); i++)
                if (gpuType[act[i]] == wantType) { sub.push_back(act[i]); subPos.push_back(i); }
            if (sub.empty()) continue;
            int nA = (int)sub.size();
            rt().cur = 0;
            int xoff[9] = {0};
            const bool persist = !forceLevels && exec_persistent(nA) != 0;
            if (persist) {
                // group the active blocks by (position mod 8): one task queue per XCD
                std::vector<int32_t> g;
                std::vector<size_t> gp;
                for (int x = 0; x < 8; x++) {
                    xoff[x] = (int)g.size();
                    for (size_t i = x; i < sub.size(); i += 8) { g.push_back(sub[i]); gp.push_back(subPos[i]); }
                }
                xoff[8] = (int)g.size();
                sub.swap(g);
                subPos.swap(gp);
            }
            rt_h2d(dActive, sub.data(), sub.size() * sizeof(int32_t));
            RtEvent e0, e1, uploaded;
            uploaded.record();
            std::vector<std::unique_ptr<RtEvent>> evs, keep;
            e0.record();
            if (persist) {
                D4GCtx c = make_ctx(P, nA);
                epoch++;
                rt_memset(dHeads, 0, 64);
                const char* sl = getenv("D4G_SPIN_LIMIT");
                const long long spinLimit = sl ? atoll(sl) : (1LL << 21);
                D4GQueue qs = {P.dStateFlat, P.nStateFlat, dHeads, dReady, epoch, {0}, spinLimit};
                D4GQueue qh = {P.dHdrFlat, P.nHdrFlat, dHeads + 8, dReady, epoch, {0}, spinLimit};
                for (int x = 0; x < 9; x++) { qs.xoff[x] = xoff[x]; qh.xoff[x] = xoff[x]; }
                RtEvent ready;
                ready.record();
                rt_stream2_wait(ready);
                static const int cus = device_cus();
                static const int sPerCu = env_int("D4G_STATE_WGS_PER_CU", 8), hPerCu = env_int("D4G_HS_WGS_PER_CU", 4);
                i64 ns = (i64)P.nStateFlat * nA, nh = (i64)P.nHdrFlat * nA;
                i64 gs = std::min<i64>(ns, (i64)cus * sPerCu), gh = std::min<i64>(nh, (i64)cus * hPerCu);
                evs.emplace_back(new RtEvent());
                evs.back()->record();
                RT_LAUNCH(k_persist_state_ops, gs, state_block(), c, qs);
                evs.emplace_back(new RtEvent());
                evs.back()->record();
                stats.kernel_launches++;
                stats.state_launches++;
                for (int k : sub) { stats.state_tokens_per_round += hBlocks[k].tokCount; stats.state_bytes_per_round += hBlocks[k].uLen; }
                if (gh > 0) {
                    RT_LAUNCH2(k_persist_hdr_search, gh, 64, c, qh);
                    stats.kernel_launches++;
                }
                RtEvent hsDone;
                hsDone.record2();
                rt_stream_wait(hsDone);
                RT_LAUNCH(k_select, nA, state_block(), c, dResults);
                stats.kernel_launches++;
                stats.search_lanes = std::max<int64_t>(stats.search_lanes, 1);
            } else {
            // Split the active blocks into groups, one stream lane each: the launch tail of one group's level
            // (a few long recode/tree ops) overlaps the other groups' levels.
            int G = std::min(lanes(), std::max(1, nA / 16));
            std::vector<std::unique_ptr<RtEvent>> laneDone;
            for (int g = 0; g < G; g++) {
                int lo = (int)((i64)nA * g / G), hi = (int)((i64)nA * (g + 1) / G);
                if (hi <= lo) continue;
                rt().cur = g;
                i64 tokSum = 0, uSum = 0;
                for (int k = lo; k < hi; k++) { tokSum += hBlocks[sub[k]].tokCount; uSum += hBlocks[sub[k]].uLen; }
                D4GCtx c = make_ctx(P, hi - lo);
                c.active = dActive + lo;
                rt_stream_wait(uploaded);
                i64 groups = (hi - lo + 7) / 8;
                {   // launch tiles (d4g_map_wg): the whole group by default
                    static int tg = -1;
                    if (tg < 0) { const char* t = getenv("D4G_TILE_GROUPS"); tg = t ? atoi(t) : 0; }
                    c.tileGroups = tg > 0 && tg < groups ? tg : (int)groups;
                    groups = (groups + c.tileGroups - 1) / c.tileGroups * c.tileGroups;
                }
                // Level l's header searches read bases produced at level l-1, so they run on the lane's second
                // stream beside level l's state ops (the searches are LDS-bound at low occupancy).
                RtEvent* lvlPrev = nullptr;
                rt_stream2_wait(uploaded);
                for (int l = 0; l < P.nLevels; l++) {
                    if (P.hdrOff[l].second) {
                        if (lvlPrev) rt_stream2_wait(*lvlPrev);
                        i64 grid = 8 * groups * P.hdrOff[l].second;
                        RT_LAUNCH2(k_exec_hdr_search, grid, 64, c, P.dLists + P.hdrOff[l].first, P.hdrOff[l].second);
                        stats.kernel_launches++;
                    }
                    if (P.stateOff[l].second) {
                        i64 grid = 8 * groups * P.stateOff[l].second;
                        evs.emplace_back(new RtEvent());
                        evs.back()->record();
                        RT_LAUNCH(k_exec_state_ops, grid, state_block(), c, P.dLists + P.stateOff[l].first, P.stateOff[l].second);
                        evs.emplace_back(new RtEvent());
                        evs.back()->record();
                        stats.kernel_launches++;
                        stats.state_launches++;
                        stats.state_tokens_per_round += tokSum;
                        stats.state_bytes_per_round += uSum;
                    }
                    if (P.wideOff[l].second) {
                        i64 grid = 8 * groups * P.wideOff[l].second;
                        RT_LAUNCH(k_exec_state_ops_wide, grid, wide_block(), c, P.dLists + P.wideOff[l].first, P.wideOff[l].second);
                        stats.kernel_launches++;
                    }
                    keep.emplace_back(new RtEvent());
                    keep.back()->record();
                    lvlPrev = keep.back().get();
                }
                keep.emplace_back(new RtEvent());
                keep.back()->record2();
                rt_stream_wait(*keep.back());
                RT_LAUNCH(k_select, hi - lo, state_block(), c, dResults + lo);
                stats.kernel_launches++;
                laneDone.emplace_back(new RtEvent());
                laneDone.back()->record();
                stats.search_lanes = std::max<int64_t>(stats.search_lanes, G);   // most lanes any round of the batch used
            }
            rt().cur = 0;
            for (auto& ev : laneDone) rt_stream_wait(*ev);
            }
            e1.record();
            std::vector<D4GRoundResult> r(sub.size());
            rt_d2h(r.data(), dResults, sub.size() * sizeof(D4GRoundResult));
            const float roundMs = rt_elapsed_ms(e0, e1);
            msSearch += roundMs;
            if (persist && !r.empty() && r[0].improved < 0) {
                // a wait inside the persistent kernels gave up (see wg_wait_slot): nothing was selected, the blocks are
                // untouched — the same round again with the level executor, which has no cross-kernel waits
                int32_t zero[2] = {0, 0};
                rt_h2d(errors() + 1, zero, 4);
                rt_sync();
                stats.persist_fallbacks++;
                forceLevels = true;
                pass--;
                continue;
            }
            forceLevels = false;
            if (getenv("D4G_DEBUG_ROUNDS"))
                fprintf(stderr, "search round %lld (%s program, %s): %d active blocks, %.3f ms\n", (long long)stats.rounds, pass == 0 ? "dynamic" : "fixed",
                        persist ? "persistent" : "levels", nA, roundMs);
            for (size_t k = 0; k + 1 < evs.size(); k += 2) stats.ms_state_kernels += rt_elapsed_ms(*evs[k], *evs[k + 1]);
            for (size_t k = 0; k < sub.size(); k++) {
                res[subPos[k]] = r[k];
                gpuType[sub[k]] = r[k].newType;
            }
        }
        stats.rounds++;
        return res;
    }

    void check_device_errors() {
        int32_t e = 0;
        rt_d2h(&e, errors(), 4);
        if (e != 0) {
            rt_memset(errors(), 0, 4);
            rt_sync();
            throw std::runtime_error("device consistency check failed (" + std::to_string(e) + " errors)");
        }
    }

    // Winner of optimiseBlock given the device result and the stream position (stored candidate
    // = DeflateStream.java:376-383, ranked right after op 0 "optimised").  Returns true when the
    // stored candidate wins.
    static bool stored_wins(const D4GRoundResult& r, i64 uLen, i64 pos, i64* storedSize) {
        if (uLen > 65535) return false;
        i64 c = pos % 8;
        c = c == 0 ? 0 : 8 - c;
        i64 ss = (uLen + 4) * 8 + c;
        *storedSize = ss;
        if (ss < r.bestSize) return true;
        if (ss == r.bestSize && r.improved && r.bestSeq > 0) return true;
        return false;
    }

    // ---- DeflateStream.optimise, per-block part — DeflateStream.java:496-566 ----
    void phase1() {
        // blocks the reference's loop reaches: it stops right after removing the first empty block
        std::vector<int> act;
        std::vector<std::pair<int, int>> owner;  // (stream, block index in stream)
        for (size_t si = 0; si < streams.size(); si++) {
            HStream& s = streams[si];
            if (s.status != 0) continue;
            for (size_t k = 0; k < s.blocks.size(); k++) {
                HBlock& b = s.blocks[k];
                bool sole = (k == 0 && s.blocks.size() == 1);
                if (b.uLen == 0 && !sole) break;
                if (b.type != D4G_STORED) { act.push_back(b.gpu); owner.push_back({(int)si, (int)k}); }
            }
        }
        if (exec_fused()) {   // all rounds of a block inside one workgroup; blocks the fused executor does not take follow below
            std::vector<int> fa, rest;
            std::vector<std::pair<int, int>> fo, ro;
            size_t nLong = 0;
            for (int k : act) nLong += hBlocks[k].refCount > (1LL << 14);
            for (size_t i = 0; i < act.size(); i++) {
                if (hBlocks[act[i]].refCount <= fused_max_refs(nLong)) { fa.push_back(act[i]); fo.push_back(owner[i]); }
                else { rest.push_back(act[i]); ro.push_back(owner[i]); }
            }
            if (!fa.empty()) {
                std::vector<std::vector<D4GRoundResult>> ch = run_fused(fa, 1 << 20);
                for (size_t i = 0; i < fa.size(); i++) streams[fo[i].first].blocks[fo[i].second].chain = ch[i];
                stats.rounds++;
            }
            act.swap(rest);
            owner.swap(ro);
        }
        // fixpoint rounds: every block follows its own chain of strictly improving Huffman states
        while (!act.empty()) {
            std::vector<D4GRoundResult> res = run_round_legacy(act);
            std::vector<int> nact;
            std::vector<std::pair<int, int>> nowner;
            for (size_t i = 0; i < act.size(); i++) {
                HBlock& b = streams[owner[i].first].blocks[owner[i].second];
                b.chain.push_back(res[i]);
                if (res[i].improved) { nact.push_back(act[i]); nowner.push_back(owner[i]); }
            }
            act.swap(nact);
            owner.swap(nowner);
        }
        check_device_errors();
        // sequential resolution with the stream bit position (pos drift included)
        for (HStream& s : streams) {
            if (s.status != 0) continue;
            i64 pos = 0, saved = 0;
            bool first = true;
            size_t idx = 0;
            while (idx < s.blocks.size()) {
                bool finishPass = true;
                HBlock& b = s.blocks[idx];
                bool hasNext = idx + 1 < s.blocks.size();
                if (b.uLen > 0 || (first && !hasNext)) {
                    pos += 3;
                    if (b.type != D4G_STORED) {
                        size_t step = 0;
                        // chain index = number of improvements already applied to this block
                        while (step < b.chain.size() && b.chain[step].curSize != b.size) step++;
                        if (step >= b.chain.size()) throw std::runtime_error("phase1: chain lookup failed");
                        const D4GRoundResult& r = b.chain[step];
                        i64 ss = 0;
                        if (stored_wins(r, b.uLen, pos, &ss)) {
                            i64 cs = b.size - ss;
                            if (cs > 0) { saved += cs; b.type = D4G_STORED; finishPass = false; }
                        } else if (r.improved) {
                            saved += b.size - r.bestSize;
                            b.size = r.bestSize;
                            finishPass = false;
                        }
                    }
                    pos += b.size_at(pos);
                } else {
                    saved += b.size_at(pos + 3) + 3;
                    s.blocks.erase(s.blocks.begin() + idx);
                    break;
                }
                if (finishPass) { idx++; first = false; }
            }
            s.saved = saved;
            // final Huffman type per block comes from the last round that ran on it
            for (HBlock& b : s.blocks)
                if (b.type != D4G_STORED) b.type = gpuType[b.gpu];
        }
    }

    // ---- DeflateStream.mergeBlocks — DeflateStream.java:568-650, all streams in lockstep ----
    struct MergeReq { int stream; int arena; };
    static bool can_merge(const HBlock& a, const HBlock& b) {
        if (a.type == D4G_STORED) return a.uLen + b.uLen <= 65535;
        return b.type == D4G_FIXED || b.type == D4G_DYNAMIC;
    }
    // advance a stream's loop until it needs a device evaluation (returns true) or finishes
    bool merge_advance(int si, MergeReq* req) {
        HStream& s = streams[si];
        while (s.mIdx < s.blocks.size()) {
            HBlock& cur = s.blocks[s.mIdx];
            bool hasNext = s.mIdx + 1 < s.blocks.size();
            bool finishPass = true;
            if (s.mFirst && !hasNext) {
                s.mPos += cur.size_at(s.mPos + 3) + 3;
            } else if (cur.uLen > 0) {
                s.mPos += 3;
                if (hasNext && can_merge(cur, s.blocks[s.mIdx + 1])) {
                    HBlock& next = s.blocks[s.mIdx + 1];
                    if (cur.type == D4G_STORED) {  // DeflateBlockUncompressed.merge — :112-117 (host only)
                        HBlock m = cur;
                        m.uLen = cur.uLen + next.uLen;
                        m.tokCount = 0;
                        i64 curNo = cur.size_at(s.mPos);
                        i64 nextNo = next.size_at(s.mPos + curNo + 3);
                        i64 cs = (curNo + 3 + nextNo) - m.size_at(s.mPos);
                        if (cs > 0) {
                            s.mSaved += cs;
                            s.blocks[s.mIdx] = m;
                            s.blocks.erase(s.blocks.begin() + s.mIdx + 1);
                            finishPass = false;
                        }
                    } else {
                        int ar = cur.gpu == s.arena[0] ? s.arena[1] : s.arena[0];
                        D4GBlock& d = hBlocks[ar];
                        d.tokStart = cur.tokStart;
                        d.tokCount = cur.tokCount + next.tokCount;
                        d.refStart = cur.refStart;
                        d.refCount = cur.refCount + next.refCount;
                        d.uStart = cur.uStart;
                        d.uLen = cur.uLen + next.uLen;
                        d.maskWords = (d.refCount + 63) / 64;
                        if (d.refCount > cluster_min_refs()) d.maskWords = (d.maskWords + 15) & ~15LL;   // whole 128-byte lines (zero padding): see layout_blocks
                        d.type = D4G_FIXED;
                        req->stream = si;
                        req->arena = ar;
                        s.mWaiting = true;
                        return true;
                    }
                }
                s.mPos += s.blocks[s.mIdx].size_at(s.mPos);
            } else {
                s.mSaved += cur.size_at(s.mPos + 3) + 3;
                s.blocks.erase(s.blocks.begin() + s.mIdx);
                break;
            }
            if (finishPass) { commit_block(si, s.mIdx); s.mIdx++; s.mFirst = false; }
        }
        s.mDone = true;
        return false;
    }
    void merge_apply(int si, int arena, const D4GRoundResult& r) {
        HStream& s = streams[si];
        HBlock& cur = s.blocks[s.mIdx];
        HBlock& next = s.blocks[s.mIdx + 1];
        i64 uLen = cur.uLen + next.uLen;
        i64 curNo = cur.size_at(s.mPos);
        i64 nextNo = next.size_at(s.mPos + curNo + 3);
        HBlock m;
        m.tokStart = cur.tokStart;
        m.tokCount = cur.tokCount + next.tokCount;
        m.refStart = cur.refStart;
        m.refCount = cur.refCount + next.refCount;
        m.homeGpu = cur.homeGpu;
        m.ordinal = cur.ordinal;
        m.uStart = cur.uStart;
        m.uLen = uLen;
        i64 ss = 0;
        if (stored_wins(r, uLen, s.mPos, &ss)) { m.type = D4G_STORED; m.gpu = -1; m.size = 0; }
        else { m.type = r.newType; m.gpu = arena; m.size = r.bestSize; }
        i64 cs = (curNo + 3 + nextNo) - m.size_at(s.mPos);
        bool finishPass = true;
        if (cs > 0) {
            s.mSaved += cs;
            s.blocks[s.mIdx] = m;
            s.blocks.erase(s.blocks.begin() + s.mIdx + 1);
            finishPass = false;
        }
        s.mPos += s.blocks[s.mIdx].size_at(s.mPos);
        if (finishPass) { commit_block(si, s.mIdx); s.mIdx++; s.mFirst = false; }
        s.mWaiting = false;
    }
    // A merged block that the walk has finished with lives in one of the stream's two arenas, which the next chain of
    // merges will overwrite: move its descriptor, state and mask to the device block of the first parsed block it covers
    // (that block is dead now) and to the stream's commit mask area (disjoint by construction: word offset =
    // first record / 64 + position of that first block).
    std::vector<D4GMergeJob> pendingCommits;
    void commit_block(int si, size_t idx) {
        HStream& s = streams[si];
        HBlock& hb = s.blocks[idx];
        if (hb.gpu < 0 || (hb.gpu != s.arena[0] && hb.gpu != s.arena[1])) return;
        const int home = hb.homeGpu;
        D4GBlock d = hBlocks[hb.gpu];
        d.stateIdx = hBlocks[home].stateIdx;
        d.maskBase = s.commitMaskBase + ((hb.refStart - s.refBase) >> 6) + hb.ordinal;
        d.maskWords = (hb.refCount + 63) / 64;
        d.binStat = -1;
        d.passMemo = -1;
        hBlocks[home] = d;
        gpuType[home] = gpuType[hb.gpu];
        patch_block(home);
        pendingCommits.push_back({hb.gpu, 0, home, 0});
        hb.gpu = home;
    }
    // descriptor changes are collected and applied by one upload + one scatter kernel (k_patch_blocks)
    std::map<int32_t, D4GBlock> blockPatches;
    void patch_block(int idx) { blockPatches[idx] = hBlocks[idx]; }
    void flush_block_patches() {
        if (blockPatches.empty()) return;
        std::vector<int32_t> idx;
        std::vector<D4GBlock> src;
        for (auto& kv : blockPatches) { idx.push_back(kv.first); src.push_back(kv.second); }
        blockPatches.clear();
        int32_t* dIdx = (int32_t*)rt_malloc(idx.size() * 4);
        D4GBlock* dSrcB = (D4GBlock*)rt_malloc(src.size() * sizeof(D4GBlock));
        rt_h2d(dIdx, idx.data(), idx.size() * 4);
        rt_h2d(dSrcB, src.data(), src.size() * sizeof(D4GBlock));
        RT_LAUNCH(k_patch_blocks, idx.size(), 64, dBlocks, dIdx, dSrcB, (int)idx.size());
        stats.kernel_launches++;
        rt_sync();   // (the staging buffers go back to the pool)
        rt_free(dIdx); rt_free(dSrcB);
    }
    void flush_commits(D4GMergeJob* dJobs) {
        flush_block_patches();
        if (pendingCommits.empty()) return;
        rt_h2d(dJobs, pendingCommits.data(), pendingCommits.size() * sizeof(D4GMergeJob));
        D4GCtx c = make_ctx(engine().progFixed, 0);
        RT_LAUNCH(k_commit_merged, pendingCommits.size(), 256, c, dJobs);
        stats.kernel_launches++;
        rt_sync();   // (the job list is re-used right away)
        pendingCommits.clear();
    }
    void phase_merge() {
        Engine& E = engine();
        D4GMergeJob* dJobs = (D4GMergeJob*)rt_malloc(2 * streams.size() * sizeof(D4GMergeJob) + 64);
        while (true) {
            std::vector<MergeReq> reqs;
            std::vector<D4GMergeJob> jobs;
            for (size_t si = 0; si < streams.size(); si++) {
                HStream& s = streams[si];
                if (s.status != 0 || s.mDone) continue;
                MergeReq rq;
                if (merge_advance((int)si, &rq)) {
                    reqs.push_back(rq);
                    D4GMergeJob j;
                    j.blkA = s.blocks[s.mIdx].gpu;
                    j.blkB = s.blocks[s.mIdx + 1].gpu;
                    j.blkM = rq.arena;
                    j.pad = 0;
                    jobs.push_back(j);
                    patch_block(rq.arena);
                }
            }
            flush_commits(dJobs);   // before any arena is overwritten
            if (reqs.empty()) break;
            rt_h2d(dJobs, jobs.data(), jobs.size() * sizeof(D4GMergeJob));
            D4GCtx c = make_ctx(E.progFixed, 0);
            RT_LAUNCH(k_make_merged, jobs.size(), state_block(), c, dJobs);
            stats.kernel_launches++;
            std::vector<int> act;
            for (auto& rq : reqs) { act.push_back(rq.arena); gpuType[rq.arena] = D4G_FIXED; }
            std::vector<D4GRoundResult> res = run_round(act);
            for (size_t i = 0; i < reqs.size(); i++) merge_apply(reqs[i].stream, reqs[i].arena, res[i]);
        }
        rt_free(dJobs);
        flush_block_patches();
        check_device_errors();
        for (HStream& s : streams)
            if (s.status == 0) s.saved += s.mSaved;
    }

    // ---- DeflateStream.write — :128-145 ----
    void phase_write() {
        std::vector<D4GWriteJob> jobs;
        i64 words = 0;
        for (HStream& s : streams) {
            s.outWordBase = words;
            if (s.status != 0) continue;
            i64 pos = 0;
            for (size_t k = 0; k < s.blocks.size(); k++) {
                const HBlock& b = s.blocks[k];
                D4GWriteJob j;
                memset(&j, 0, sizeof(j));
                j.blk = b.gpu;
                j.type = b.type;
                j.isFinal = k + 1 == s.blocks.size();
                j.bitStart = words * 32 + pos;
                j.uAbs = s.uBase + b.uStart;
                j.uLen = b.uLen;
                jobs.push_back(j);
                pos += 3;
                pos += b.size_at(pos);
            }
            s.outBits = pos;
            words += (pos + 31) / 32 + 2;
        }
        outWords = words;
        dOut = (uint32_t*)rt_malloc((size_t)words * 4 + 64);
        rt_memset(dOut, 0, (size_t)words * 4 + 64);
        if (!jobs.empty()) {
            D4GWriteJob* dJobs = (D4GWriteJob*)rt_malloc(jobs.size() * sizeof(D4GWriteJob));
            rt_h2d(dJobs, jobs.data(), jobs.size() * sizeof(D4GWriteJob));
            D4GCtx c = make_ctx(engine().progDyn, 0);
#ifdef D4G_HOSTSIM
            const int writeBlock = state_block();
#else
            const int writeBlock = 1024;   // one workgroup per block walks its tokens in order: wide steps, few of them
#endif
            RT_LAUNCH(k_write, jobs.size(), writeBlock, c, dJobs, dOut);
            stats.kernel_launches++;
            rt_sync();
            rt_free(dJobs);
        }
        check_device_errors();
        for (HStream& s : streams)
            if (s.status == 0) stats.bytes_out += (s.outBits + 7) / 8;
    }

    // ---- trailer checksums of the decoded bytes (gzip CRC-32 + ISIZE, zlib Adler-32) ----
    std::vector<D4GCsumOut> csums;
    void checksums() {
        if (!csums.empty() || streams.empty()) return;
        if (!dU) throw std::runtime_error("checksums: the batch has not been parsed");
        size_t n = streams.size();
        std::vector<long long> base(n + 1, 0);
        for (size_t i = 0; i < n; i++) base[i + 1] = base[i] + (streams[i].status == 0 ? (streams[i].nU + D4G_CSUM_TILE - 1) / D4G_CSUM_TILE : 0);
        long long nTiles = base[n];
        long long* dBase = (long long*)rt_malloc((n + 1) * 8);
        rt_h2d(dBase, base.data(), (n + 1) * 8);
        D4GCsumRec* dCh = (D4GCsumRec*)rt_malloc((size_t)nTiles * sizeof(D4GCsumRec) + 16);
        D4GCsumOut* dOutC = (D4GCsumOut*)rt_malloc(n * sizeof(D4GCsumOut));
        RtEvent e0, e1;
        e0.record();
        if (nTiles) {
            RT_LAUNCH(k_csum_tiles, nTiles, 256, dStreams, dBase, (int)n, dU, engine().dCrcTab, dCh);
            stats.kernel_launches++;
        }
        RT_LAUNCH(k_csum_combine, n, 256, dStreams, dBase, dCh, engine().dCrcTab + 1024, dOutC);
        stats.kernel_launches++;
        e1.record();
        csums.resize(n);
        rt_d2h(csums.data(), dOutC, n * sizeof(D4GCsumOut));
        stats.ms_checksum_kernels = rt_elapsed_ms(e0, e1);
        rt_free(dBase); rt_free(dCh); rt_free(dOutC);
    }

    void run(bool merge) {
        run_parse(merge);
        run_rest(merge);
    }
    // run() in two steps, for callers that start other work on the decoded bytes between them (the recompress modes)
    double tRun0 = 0, tRun1 = 0;
    void run_parse(bool merge) {
        if (ran) throw std::runtime_error("batch already ran");
        ran = true;
        engine().init();
        tRun0 = now_ms();
        parse_probe();
        build_blocks(merge, true);
        tRun1 = now_ms();
    }
    void run_rest(bool merge) {
        const double t0 = tRun0, t1 = tRun1, t1b = now_ms();   // (other work may have run between the two steps)
        phase1();
        double t2 = now_ms();
        if (merge) phase_merge();
        double t3 = now_ms();
        phase_write();
        double t4 = now_ms();
        stats.ms_parse = t1 - t0;
        stats.ms_optimise = t2 - t1b;
        stats.ms_merge = t3 - t2;
        stats.ms_write = t4 - t3;
        stats.ms_total = (t1 - t0) + (t4 - t1b);
        stats.ms_search_kernels = msSearch;
        stats.ms_parse_kernels = msParseKernels;
        stats.search_bytes_algorithmic = stats.bytes_in + stats.bytes_decoded + stats.bytes_out;
        release_scratch();
    }
    // After the write phase only the results are needed (output words, decoded bytes, stream table): the search's
    // working set goes back to the memory pool, where the next batch finds it.
    void release_scratch() {
        rt_sync_all();
        rt_free(dTok); dTok = nullptr;
        rt_free(dRefs); dRefs = nullptr;
        rt_free(dTokRef); dTokRef = nullptr;
        rt_free(dBinStat); dBinStat = nullptr;
        rt_free(dBinMask); dBinMask = nullptr;
        rt_free(dHsMemo); dHsMemo = nullptr;
        rt_free(dRcMemo); dRcMemo = nullptr;
        rt_free(dPassMemo); dPassMemo = nullptr;
        rt_free(dBlocks); dBlocks = nullptr;
        rt_free(dStates); dStates = nullptr;
        rt_free(dMasks); dMasks = nullptr;
        rt_free(dKeys); dKeys = nullptr;
        rt_free(dActive); dActive = nullptr;
        rt_free(dResults); dResults = nullptr;
        rt_free(dReady); dReady = nullptr;
        rt_free(dHeads); dHeads = nullptr;
    }
};

}  // namespace d4g
