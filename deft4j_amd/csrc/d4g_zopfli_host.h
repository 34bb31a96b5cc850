// d4g_zopfli_host.h — host sequencing of the Zopfli encoder kernels (d4g_zopfli.h).
//
// One ZfFront serves a group of inputs that are resident in HBM: it builds the keys and the shared match table once per
// input, then encodes every (input, options) pair asked for — the options being what deft4j passes to CafeUndZopfli
// (C/MultiCafeUndZopfliCompressor.java:19-25: splitting FIRST / LAST / NONE, `iter`, master block 8 MiB) and to jzopfli
// (C/MultiJZopfliCompressor.java:18-60: blocksplittingmax 15 / 0, first / last / none, master block 1 000 000).  All master
// blocks of all outputs move through the stages together, so the block-level parallelism of the whole group is on the
// device at once:
//   A  (FIRST)  greedy parse of each master block + split search          -> byte cut points
//   B           squeeze: `iter` shortest-path runs per block               -> best store per block
//   C           concatenate; (FIRST with > 1 cut, LAST) split search on the final parse; keep the cheaper cut set
//   D           stored / fixed / dynamic cost per final block; fixed-tree re-parse where Zopfli tries it; bit layout; emit
// The result of every output is a complete raw deflate stream in device memory.
#pragma once
#include <tuple>
#include "d4g_lz77_host.h"
#include "d4g_zopfli.h"

namespace d4g {

struct ZfSpec { int32_t input, iterations, splitting, maxblocks; long long master; };

struct ZfFront {
    LzScratch own;
    std::vector<ZfInput> hIn;
    ZfInput* dIn = nullptr;
    ZfPool pool{};
    std::vector<uint32_t*> outWords;     // per spec: device words of the stream
    std::vector<i64> outBits;
    double msTable = 0, msSplit = 0, msSqueeze = 0, msEmit = 0;
    i64 squeezeBlocks = 0, squeezePositions = 0;

    template <typename T> T* dalloc(size_t n, bool zero = false) {
        T* p = own.own((T*)rt_malloc((n ? n : 1) * sizeof(T)));
        if (zero) rt_memset(p, 0, (n ? n : 1) * sizeof(T));
        return p;
    }
    template <typename T> T* upload(const std::vector<T>& v) {
        T* p = dalloc<T>(v.size());
        rt_h2d(p, v.data(), v.size() * sizeof(T));
        return p;
    }

    // inputs: device pointers, 16-byte aligned, with >= 320 readable zero bytes after the end
    void create(size_t n, const uint8_t* const* dData, const i64* len) {
        double t0 = now_ms();
        hIn.resize(n);
        i64 total = 0;
        std::vector<ZfKeyJob> kj;
        for (size_t i = 0; i < n; i++) {
            if (len[i] >= (1LL << 31) - 65536) throw std::runtime_error("input of 2 GiB or more: split it");
            ZfInput& in = hIn[i];
            in.data = dData[i];
            in.n = len[i];
            const i64 tiles = (len[i] + ZF_KEY_TILE - 1) / ZF_KEY_TILE;
            in.val = dalloc<uint16_t>((size_t)len[i] + 8);
            in.same = dalloc<uint16_t>((size_t)len[i] + 8);
            in.lead = dalloc<uint16_t>((size_t)tiles + 1);
            in.val2 = dalloc<uint16_t>((size_t)len[i] + 8);
            const i64 sblocks = (len[i] + ZF_SORT_BLOCK - 1) / ZF_SORT_BLOCK;
            for (int kind = 0; kind < 2; kind++) {
                in.sorted[kind] = dalloc<uint16_t>((size_t)sblocks * ZF_SORT_BLOCK + 8);
                in.rank[kind] = dalloc<uint16_t>((size_t)sblocks * ZF_SORT_BLOCK + 8);
                in.bstart[kind] = dalloc<uint16_t>((size_t)sblocks * ZF_SORT_BLOCK + 8);
            }
            in.table = dalloc<uint32_t>((size_t)len[i] * 8 + 8);
            in.best = dalloc<uint32_t>((size_t)len[i] + 8);
            for (i64 t = 0; t < tiles; t++) kj.push_back({(int32_t)i, (int32_t)t});
            total += len[i];
        }
        dIn = upload(hIn);
        pool.cap = (uint32_t)std::min<i64>(total / 2 + (1 << 16), 0x7fff0000LL);
        if (const char* pw = getenv("D4G_ZF_POOL_WORDS")) pool.cap = (uint32_t)std::max(64, atoi(pw));   // tests: start small, exercise the growth
        pool.words = dalloc<uint32_t>(pool.cap);
        pool.used = dalloc<uint32_t>(4, true);
        pool.error = (int32_t*)(pool.used + 1);
        if (!kj.empty()) {
            ZfKeyJob* dK = upload(kj);
            RT_LAUNCH(k_zf_keys_a, kj.size(), 256, dIn, dK);
            RT_LAUNCH(k_zf_keys_b, kj.size(), 256, dIn, dK);
        }
        // the shared table: chains enumerated from sorted buckets (D4G_ZF_TABLE=scan: the window-scan kernel the tails use)
        static const bool scanTable = getenv("D4G_ZF_TABLE") && !strcmp(getenv("D4G_ZF_TABLE"), "scan");
        std::vector<ZfMatchJob> mj;
        if (scanTable) {
            for (size_t i = 0; i < n; i++)
                for (i64 p = 0; p < len[i]; p += ZF_TILE)
                    mj.push_back({(int32_t)i, (int32_t)std::min<i64>(ZF_TILE, len[i] - p), p, len[i], hIn[i].table + p * 8, hIn[i].best + p});
            run_match(mj);
        } else {
            std::vector<ZfSortJob> sj;
            for (size_t i = 0; i < n; i++)
                for (i64 b = 0; b * ZF_SORT_BLOCK < len[i]; b++)
                    for (int kind = 0; kind < 2; kind++) sj.push_back({(int32_t)i, (int32_t)b, kind, 0});
            if (!sj.empty()) {
                ZfSortJob* dS = upload(sj);
                RT_LAUNCH(k_zf_sort, sj.size(), ZF_SORT_THREADS, dIn, dS);
            }
            const i64 per = 1024;
            for (size_t i = 0; i < n; i++)
                for (i64 p = 0; p < len[i]; p += per)
                    mj.push_back({(int32_t)i, (int32_t)std::min<i64>(per, len[i] - p), p, len[i], hIn[i].table + p * 8, hIn[i].best + p});
            if (!mj.empty()) {
                ZfMatchJob* dM = upload(mj);
                with_pool_retry([&]() { RT_LAUNCH(k_zf_match_sorted, mj.size(), ZF_MS_THREADS, dIn, dM, pool); });
            }
        }
        msTable += now_ms() - t0;
    }
    void run_match(const std::vector<ZfMatchJob>& mj) {
        if (mj.empty()) return;
        ZfMatchJob* dM = upload(mj);
        with_pool_retry([&]() { RT_LAUNCH(k_zf_match, mj.size(), ZF_MATCH_THREADS, dIn, dM, pool); });
    }
    // The change-point pool is sized for ordinary data (half a word per input byte); inputs whose positions have many
    // record-setting matches need more.  A launch that ran out of room is run again with twice the pool — what earlier
    // launches stored is kept, the failed launch's entries are all rewritten — instead of failing the whole batch
    // (the reference never fails on valid input, C/CompressionUtil.java:151-173).
    uint32_t poolUsedHost = 0;
    template <typename Launch>
    void with_pool_retry(Launch launch) {
        for (int attempt = 0;; attempt++) {
            launch();
            uint32_t st[2] = {0, 0};
            rt_d2h(st, pool.used, 8);   // {used, error}
            if (!st[1]) { poolUsedHost = st[0]; return; }
            if (attempt >= 16 || pool.cap >= 0x7fff0000u) throw std::runtime_error("zopfli match table: change-point pool exhausted");
            const uint32_t newCap = (uint32_t)std::min<i64>((i64)pool.cap * 2, 0x7fff0000LL);
            uint32_t* nw = dalloc<uint32_t>(newCap);
            if (poolUsedHost) rt_d2d(nw, pool.words, (size_t)poolUsedHost * 4);
            pool.words = nw;
            pool.cap = newCap;
            const uint32_t z[2] = {poolUsedHost, 0};
            rt_h2d(pool.used, z, 8);
            rt_sync();
            poolGrowths++;
        }
    }
    int poolGrowths = 0;

    // ---- tails ----
    struct Tail { i64 start; uint32_t* table; uint32_t* best; };
    std::map<std::pair<int, i64>, Tail> tails;
    void ensure_tails(const std::vector<std::pair<int, i64>>& ends) {
        std::vector<ZfTailQuery> qs;
        std::vector<std::pair<int, i64>> keys;
        for (auto& e : ends)
            if (!tails.count(e) && e.second > 0) { tails[e] = Tail{0, nullptr, nullptr}; qs.push_back({e.first, 0, e.second}); keys.push_back(e); }
        if (qs.empty()) return;
        ZfTailQuery* dQ = upload(qs);
        uint32_t* dT = dalloc<uint32_t>(qs.size());
        RT_LAUNCH(k_zf_tail_len, qs.size(), 64, dIn, dQ, dT);
        std::vector<uint32_t> T(qs.size());
        rt_d2h(T.data(), dT, qs.size() * 4);
        std::vector<ZfMatchJob> mj;
        for (size_t k = 0; k < qs.size(); k++) {
            const i64 end = qs[k].end, start = std::max<i64>(0, end - (i64)T[k]);
            Tail& t = tails[keys[k]];
            t.start = start;
            if (end == hIn[qs[k].input].n) {     // the shared table was made for this end
                t.table = hIn[qs[k].input].table + start * 8;
                t.best = hIn[qs[k].input].best + start;
                continue;
            }
            t.table = dalloc<uint32_t>((size_t)(end - start) * 8);
            t.best = dalloc<uint32_t>((size_t)(end - start));
            for (i64 p = start; p < end; p += ZF_TILE)
                mj.push_back({qs[k].input, (int32_t)std::min<i64>(ZF_TILE, end - p), p, end, t.table + (p - start) * 8, t.best + (p - start)});
        }
        run_match(mj);
    }
    ZfView view(int input, i64 start, i64 end) {
        const ZfInput& in = hIn[input];
        ZfView v{};
        v.data = in.data; v.same = in.same; v.table = in.table; v.best = in.best; v.pool = pool.words;
        v.start = start; v.end = end;
        if (end > 0) {
            const Tail& t = tails.at({input, end});
            v.tailStart = t.start; v.tailTable = t.table; v.tailBest = t.best;
        } else v.tailStart = 0;
        return v;
    }

    // ---- stores ----
    struct Store { uint16_t* lit = nullptr; uint16_t* dist = nullptr; uint32_t* pos = nullptr; uint32_t size = 0; };
    Store alloc_store(size_t cap) {
        Store s;
        s.lit = dalloc<uint16_t>(cap + 8); s.dist = dalloc<uint16_t>(cap + 8); s.pos = dalloc<uint32_t>(cap + 8);
        return s;
    }
    static ZfStore dev(const Store& s) { return ZfStore{s.lit, s.dist, s.pos, s.size}; }

    struct Split { std::vector<uint32_t> points, bytePos; };
    // block_split_lz77 on a list of stores
    std::vector<Split> run_split(const std::vector<Store>& stores, const std::vector<uint32_t>& maxblocks) {
        const size_t n = stores.size();
        std::vector<Split> res(n);
        if (!n) return res;
        std::vector<ZfSplitJob> jobs(n);
        std::vector<uint32_t*> dPts(n), dBp(n);
        uint32_t* dNp = dalloc<uint32_t>(n + 1, true);
        int32_t* dErr = (int32_t*)dalloc<uint32_t>(1, true);
        for (size_t k = 0; k < n; k++) {
            // unlimited splitting (blocksplittingmax 0): split points are at least ten symbols apart
            const uint32_t cap = maxblocks[k] ? maxblocks[k] : (uint32_t)(stores[k].size / 10 + 2);
            dPts[k] = dalloc<uint32_t>(cap + 1);
            dBp[k] = dalloc<uint32_t>(cap + 1);
            jobs[k] = {dev(stores[k]), maxblocks[k], cap, dalloc<uint8_t>(stores[k].size + 8, true), dPts[k], dBp[k], dNp + k, dErr};
        }
        ZfSplitJob* dJ = upload(jobs);
        RT_LAUNCH(k_zf_split, n, ZF_SPLIT_WAVES * 64, dJ);
        std::vector<uint32_t> np(n);
        rt_d2h(np.data(), dNp, n * 4);
        int32_t err = 0;
        rt_d2h(&err, dErr, 4);
        if (err) throw std::runtime_error("zopfli block splitting: more split points than the buffer holds");
        for (size_t k = 0; k < n; k++) {
            res[k].points.resize(np[k]);
            res[k].bytePos.resize(np[k]);
            if (np[k]) { rt_d2h(res[k].points.data(), dPts[k], np[k] * 4); rt_d2h(res[k].bytePos.data(), dBp[k], np[k] * 4); }
        }
        return res;
    }
    std::vector<ZfRangeOut> run_ranges(const std::vector<ZfRangeJob>& jobs) {
        std::vector<ZfRangeOut> out(jobs.size());
        if (jobs.empty()) return out;
        ZfRangeJob* dJ = upload(jobs);
        ZfRangeOut* dO = dalloc<ZfRangeOut>(jobs.size());
        RT_LAUNCH(k_zf_range_cost, jobs.size(), 64, dJ, dO);
        rt_d2h(out.data(), dO, jobs.size() * sizeof(ZfRangeOut));
        return out;
    }

    struct Block { int mb; i64 start, end; Store buf[2]; uint16_t* la; uint32_t* path; ZfSqOut out; };
    void encode(const std::vector<ZfSpec>& specs);
};

inline void ZfFront::encode(const std::vector<ZfSpec>& specs) {
    const size_t nS = specs.size();
    outWords.assign(nS, nullptr);
    outBits.assign(nS, 0);
    struct MB { int spec; i64 start, end; std::vector<i64> cuts; std::vector<int> blocks; Store lz; std::vector<uint32_t> sp; };
    std::vector<MB> mbs;
    std::vector<std::vector<int>> mbOf(nS);
    for (size_t s = 0; s < nS; s++) {
        const ZfSpec& sp = specs[s];
        if (sp.input < 0 || (size_t)sp.input >= hIn.size() || sp.iterations < 1 || sp.splitting < 0 || sp.splitting > 2 || sp.maxblocks < 0 || sp.master < 0)
            throw std::runtime_error("bad zopfli spec");
        const i64 n = hIn[sp.input].n;
        const i64 master = sp.master ? sp.master : std::max<i64>(n, 1);
        if (master > (8LL << 20)) throw std::runtime_error("zopfli master block larger than 8 MiB");
        i64 i = 0;
        do {
            const i64 size = i + master >= n ? n - i : master;
            mbOf[s].push_back((int)mbs.size());
            mbs.push_back(MB{(int)s, i, i + size, {}, {}, Store{}, {}});
            i += size;
        } while (i < n);
    }
    // ---- A: FIRST — greedy parse + split search per master block ----
    double t0 = now_ms();
    {
        std::vector<std::pair<int, i64>> ends;
        for (MB& m : mbs) ends.push_back({specs[m.spec].input, m.end});
        ensure_tails(ends);
        std::vector<int> who;
        std::vector<ZfGreedyJob> gj;
        std::vector<Store> gs;
        std::vector<uint32_t> maxb;
        for (size_t k = 0; k < mbs.size(); k++) {
            MB& m = mbs[k];
            if (specs[m.spec].splitting != ZF_SPLIT_FIRST || m.end == m.start) continue;
            Store st = alloc_store((size_t)(m.end - m.start));
            who.push_back((int)k);
            gs.push_back(st);
            maxb.push_back((uint32_t)specs[m.spec].maxblocks);
        }
        if (!who.empty()) {
            uint32_t* dCnt = dalloc<uint32_t>(who.size());
            for (size_t q = 0; q < who.size(); q++) {
                MB& m = mbs[who[q]];
                gj.push_back({view(specs[m.spec].input, m.start, m.end), gs[q].lit, gs[q].dist, gs[q].pos, dCnt + q});
            }
            ZfGreedyJob* dG = upload(gj);
            RT_LAUNCH(k_zf_greedy, gj.size(), 64, dG);
            std::vector<uint32_t> cnt(who.size());
            rt_d2h(cnt.data(), dCnt, who.size() * 4);
            for (size_t q = 0; q < who.size(); q++) gs[q].size = cnt[q];
            std::vector<Split> sp = run_split(gs, maxb);
            for (size_t q = 0; q < who.size(); q++)
                for (uint32_t b : sp[q].bytePos) mbs[who[q]].cuts.push_back((i64)b);
        }
    }
    msSplit += now_ms() - t0;
    // ---- B: squeeze every block ----
    t0 = now_ms();
    std::vector<Block> blocks;
    for (size_t k = 0; k < mbs.size(); k++) {
        MB& m = mbs[k];
        if (m.end == m.start) continue;
        i64 s = m.start;
        for (size_t c = 0; c <= m.cuts.size(); c++) {
            const i64 e = c == m.cuts.size() ? m.end : m.cuts[c];
            m.blocks.push_back((int)blocks.size());
            blocks.push_back(Block{(int)k, s, e, {}, nullptr, nullptr, {}});
            s = e;
        }
    }
    {
        std::vector<std::pair<int, i64>> ends;
        for (Block& b : blocks) ends.push_back({specs[mbs[b.mb].spec].input, b.end});
        ensure_tails(ends);
        // identical squeezes (same bytes, same iteration count: e.g. LAST and NONE of one master block) run once
        std::map<std::tuple<int, i64, i64, int>, size_t> first;
        std::vector<size_t> rep(blocks.size()), jobOf(blocks.size(), (size_t)-1);
        std::vector<ZfSqJob> jobs;
        for (size_t q = 0; q < blocks.size(); q++) {
            Block& b = blocks[q];
            const int input = specs[mbs[b.mb].spec].input, iters = specs[mbs[b.mb].spec].iterations;
            auto key = std::make_tuple(input, b.start, b.end, iters);
            auto it = first.find(key);
            if (it != first.end()) { rep[q] = it->second; continue; }
            first[key] = q;
            rep[q] = q;
            const size_t cap = (size_t)(b.end - b.start);
            b.buf[0] = alloc_store(cap); b.buf[1] = alloc_store(cap);
            b.la = dalloc<uint16_t>(cap + 8); b.path = dalloc<uint32_t>(cap + 8);
            ZfSqJob j{};
            j.v = view(input, b.start, b.end);
            for (int x = 0; x < 2; x++) { j.lit[x] = b.buf[x].lit; j.dist[x] = b.buf[x].dist; j.pos[x] = b.buf[x].pos; }
            j.lengthArray = b.la; j.path = b.path; j.iterations = iters; j.fixedModel = 0;
            jobOf[q] = jobs.size();
            jobs.push_back(j);
            squeezeBlocks++;
            squeezePositions += (b.end - b.start) * j.iterations;
        }
        if (!jobs.empty()) {
            ZfSqJob* dJ = upload(jobs);
            ZfSqOut* dO = dalloc<ZfSqOut>(jobs.size());
            RT_LAUNCH(k_zf_squeeze, jobs.size(), 64, dJ, dO);
            std::vector<ZfSqOut> outs(jobs.size());
            rt_d2h(outs.data(), dO, jobs.size() * sizeof(ZfSqOut));
            for (size_t q = 0; q < blocks.size(); q++) {
                if (rep[q] != q) continue;
                const ZfSqOut& o = outs[jobOf[q]];
                blocks[q].out = o;
                blocks[q].buf[o.bestBuf].size = o.bestSize;
            }
            for (size_t q = 0; q < blocks.size(); q++)
                if (rep[q] != q) { blocks[q].out = blocks[rep[q]].out; blocks[q].buf[0] = blocks[rep[q]].buf[0]; blocks[q].buf[1] = blocks[rep[q]].buf[1]; }
            if (env_int("D4G_DEBUG_ZOPFLI", 0) > 1)
                for (size_t q = 0; q < jobs.size() && q < 8; q++)
                    fprintf(stderr, "[zopfli] squeeze job %zu (%lld bytes): ticks greedy %lld, DP %lld, trace+follow %lld, cost %lld, statistics %lld\n", q,
                            (long long)(jobs[q].v.end - jobs[q].v.start), outs[q].cyc[0], outs[q].cyc[1], outs[q].cyc[2], outs[q].cyc[3], outs[q].cyc[4]);
        }
    }
    msSqueeze += now_ms() - t0;
    // ---- C: concatenate, second split ----
    t0 = now_ms();
    {
        std::vector<ZfRangeJob> rj;          // cost of every FIRST block's own store
        std::vector<int> rjBlock;
        for (size_t k = 0; k < mbs.size(); k++) {
            MB& m = mbs[k];
            if (m.blocks.empty()) continue;
            if (m.blocks.size() == 1) { Block& b = blocks[m.blocks[0]]; m.lz = b.buf[b.out.bestBuf]; }
            else {
                size_t tot = 0;
                for (int bi : m.blocks) tot += blocks[bi].out.bestSize;
                m.lz = alloc_store(tot);
                size_t at = 0;
                for (size_t c = 0; c < m.blocks.size(); c++) {
                    Block& b = blocks[m.blocks[c]];
                    const Store& s = b.buf[b.out.bestBuf];
                    rt_d2d(m.lz.lit + at, s.lit, (size_t)s.size * 2);
                    rt_d2d(m.lz.dist + at, s.dist, (size_t)s.size * 2);
                    rt_d2d(m.lz.pos + at, s.pos, (size_t)s.size * 4);
                    at += s.size;
                    if (c + 1 < m.blocks.size()) m.sp.push_back((uint32_t)at);
                }
                m.lz.size = (uint32_t)tot;
            }
            if (specs[m.spec].splitting == ZF_SPLIT_FIRST && m.cuts.size() > 1)
                for (int bi : m.blocks) { Block& b = blocks[bi]; const Store& s = b.buf[b.out.bestBuf]; rj.push_back({dev(s), 0, s.size}); rjBlock.push_back((int)k); }
        }
        std::vector<ZfRangeOut> c1 = run_ranges(rj);
        std::vector<long long> total1(mbs.size(), 0);
        for (size_t q = 0; q < rj.size(); q++) total1[rjBlock[q]] += c1[q].autoCost;
        std::vector<int> who;
        std::vector<Store> ss;
        std::vector<uint32_t> maxb;
        for (size_t k = 0; k < mbs.size(); k++) {
            MB& m = mbs[k];
            if (m.blocks.empty()) continue;
            const int sp = specs[m.spec].splitting;
            if ((sp == ZF_SPLIT_FIRST && m.cuts.size() > 1) || sp == ZF_SPLIT_LAST) { who.push_back((int)k); ss.push_back(m.lz); maxb.push_back((uint32_t)specs[m.spec].maxblocks); }
        }
        std::vector<Split> s2 = run_split(ss, maxb);
        std::vector<ZfRangeJob> rj2;
        std::vector<int> rj2Who;
        for (size_t q = 0; q < who.size(); q++) {
            MB& m = mbs[who[q]];
            if (specs[m.spec].splitting == ZF_SPLIT_LAST) { m.sp = s2[q].points; continue; }
            for (size_t c = 0; c <= s2[q].points.size(); c++) {
                const uint32_t a = c == 0 ? 0 : s2[q].points[c - 1], b = c == s2[q].points.size() ? m.lz.size : s2[q].points[c];
                rj2.push_back({dev(m.lz), a, b});
                rj2Who.push_back((int)q);
            }
        }
        std::vector<ZfRangeOut> c2 = run_ranges(rj2);
        std::vector<long long> total2(who.size(), 0);
        for (size_t q = 0; q < rj2.size(); q++) total2[rj2Who[q]] += c2[q].autoCost;
        for (size_t q = 0; q < who.size(); q++) {
            MB& m = mbs[who[q]];
            if (specs[m.spec].splitting == ZF_SPLIT_FIRST && total2[q] < total1[who[q]]) m.sp = s2[q].points;
        }
    }
    msSplit += now_ms() - t0;
    // ---- D: final blocks ----
    t0 = now_ms();
    struct Fin { int mb; uint32_t a, b; ZfRangeOut c; bool expensive; int fixedBlock; ZfRangeOut cf; int btype; };
    std::vector<Fin> fins;
    {
        std::vector<ZfRangeJob> rj;
        for (size_t k = 0; k < mbs.size(); k++) {
            MB& m = mbs[k];
            if (m.blocks.empty()) continue;
            for (size_t c = 0; c <= m.sp.size(); c++) {
                const uint32_t a = c == 0 ? 0 : m.sp[c - 1], b = c == m.sp.size() ? m.lz.size : m.sp[c];
                fins.push_back(Fin{(int)k, a, b, {}, false, -1, {}, 2});
                rj.push_back({dev(m.lz), a, b});
            }
        }
        std::vector<ZfRangeOut> rc = run_ranges(rj);
        std::vector<i64> firstPos(fins.size()), blen(fins.size());
        for (size_t q = 0; q < fins.size(); q++) {
            fins[q].c = rc[q];
            fins[q].expensive = mbs[fins[q].mb].lz.size < 1000 || (double)rc[q].fixedc <= (double)rc[q].dyn * 1.1;
            firstPos[q] = rc[q].firstPos;
            blen[q] = rc[q].byteLen;
        }
        std::vector<Block> fblocks;
        {
            std::vector<std::pair<int, i64>> ends;
            for (size_t q = 0; q < fins.size(); q++)
                if (fins[q].expensive) ends.push_back({specs[mbs[fins[q].mb].spec].input, (i64)firstPos[q] + blen[q]});
            ensure_tails(ends);
            std::vector<ZfSqJob> jobs;
            for (size_t q = 0; q < fins.size(); q++) {
                Fin& f = fins[q];
                if (!f.expensive) continue;
                Block b{f.mb, (i64)firstPos[q], (i64)firstPos[q] + blen[q], {}, nullptr, nullptr, {}};
                const size_t cap = (size_t)blen[q];
                b.buf[0] = alloc_store(cap);
                b.la = dalloc<uint16_t>(cap + 8); b.path = dalloc<uint32_t>(cap + 8);
                ZfSqJob j{};
                j.v = view(specs[mbs[f.mb].spec].input, b.start, b.end);
                j.lit[0] = b.buf[0].lit; j.dist[0] = b.buf[0].dist; j.pos[0] = b.buf[0].pos;
                j.lit[1] = j.lit[0]; j.dist[1] = j.dist[0]; j.pos[1] = j.pos[0];
                j.lengthArray = b.la; j.path = b.path; j.iterations = 1; j.fixedModel = 1;
                f.fixedBlock = (int)fblocks.size();
                fblocks.push_back(b);
                jobs.push_back(j);
            }
            if (!jobs.empty()) {
                ZfSqJob* dJ = upload(jobs);
                ZfSqOut* dO = dalloc<ZfSqOut>(jobs.size());
                RT_LAUNCH(k_zf_squeeze, jobs.size(), 64, dJ, dO);
                std::vector<ZfSqOut> outs(jobs.size());
                rt_d2h(outs.data(), dO, jobs.size() * sizeof(ZfSqOut));
                std::vector<ZfRangeJob> fr;
                for (size_t q = 0; q < fblocks.size(); q++) { fblocks[q].buf[0].size = outs[q].bestSize; fr.push_back({dev(fblocks[q].buf[0]), 0, outs[q].bestSize}); }
                std::vector<ZfRangeOut> fc = run_ranges(fr);
                for (Fin& f : fins) if (f.fixedBlock >= 0) f.cf = fc[f.fixedBlock];
            }
        }
        // choice of block type (AddLZ77BlockAutoType), bit layout, emission
        std::vector<ZfEmitJob> ej;
        std::vector<long long> bitPos(nS, 0);
        std::vector<size_t> capWords(nS, 0);
        for (size_t s = 0; s < nS; s++) {
            const i64 n = hIn[specs[s].input].n;
            capWords[s] = (size_t)((n + n / 8 + 1024) / 4 + 64);
            outWords[s] = dalloc<uint32_t>(capWords[s], true);
        }
        size_t q = 0;
        for (size_t s = 0; s < nS; s++) {
            long long bp = 0;
            for (size_t mi = 0; mi < mbOf[s].size(); mi++) {
                const MB& m = mbs[mbOf[s][mi]];
                const bool lastMb = mi + 1 == mbOf[s].size();
                if (m.blocks.empty()) {      // empty input: one empty fixed block
                    ZfEmitJob e{};
                    e.s = ZfStore{nullptr, nullptr, nullptr, 0}; e.a = 0; e.b = 0; e.btype = 1; e.final = lastMb ? 1 : 0; e.bitPos = bp; e.out = outWords[s];
                    ej.push_back(e);
                    bp += 10;
                    continue;
                }
                for (size_t c = 0; c <= m.sp.size(); c++, q++) {
                    Fin& f = fins[q];
                    const bool fin = lastMb && c == m.sp.size();
                    const long long unc = f.c.unc, dyn = f.c.dyn;
                    const long long fixedc = f.expensive ? f.cf.fixedc : f.c.fixedc;
                    ZfEmitJob e{};
                    e.out = outWords[s];
                    if (unc < fixedc && unc < dyn) {
                        i64 pos = firstPos[q];
                        const i64 end = pos + blen[q];
                        const uint8_t* data = hIn[specs[s].input].data;
                        for (;;) {
                            i64 bs = std::min<i64>(65535, end - pos);
                            const bool cf = pos + bs >= end;
                            ZfEmitJob z{};
                            z.btype = 0; z.final = (fin && cf) ? 1 : 0; z.bitPos = bp; z.src = data + pos; z.srcLen = (uint32_t)bs; z.out = outWords[s];
                            ej.push_back(z);
                            bp = (((bp + 3 + 7) >> 3) + 4 + bs) * 8;
                            if (cf) break;
                            pos += bs;
                        }
                        continue;
                    }
                    if (fixedc < dyn) {
                        e.btype = 1;
                        if (f.expensive) { const Store& fs = fblocks[f.fixedBlock].buf[0]; e.s = dev(fs); e.a = 0; e.b = fs.size; }
                        else { e.s = dev(m.lz); e.a = f.a; e.b = f.b; }
                        e.final = fin; e.bitPos = bp;
                        ej.push_back(e);
                        bp += fixedc;
                    } else {
                        e.btype = 2; e.s = dev(m.lz); e.a = f.a; e.b = f.b; e.final = fin; e.bitPos = bp;
                        ej.push_back(e);
                        bp += dyn;
                    }
                }
            }
            bitPos[s] = bp;
            if ((size_t)((bp + 31) / 32) + 2 > capWords[s]) throw std::runtime_error("zopfli output larger than its buffer");
            outBits[s] = bp;
        }
        if (!ej.empty()) {
            ZfEmitJob* dE = upload(ej);
            RT_LAUNCH(k_zf_emit, ej.size(), 256, dE);
        }
        rt_sync();
    }
    msEmit += now_ms() - t0;
}

}  // namespace d4g
