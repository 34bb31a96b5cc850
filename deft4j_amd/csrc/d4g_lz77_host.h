// d4g_lz77_host.h — host sequencing of the LZ77 front end (d4g_lz77.h): a batch whose streams are produced by the
// zlib-level-9-compatible encoder kernels instead of parsed from compressed input.  The encoder leaves tokens,
// back-reference records, decoded bytes (= the raw input) and one D4GState per block in exactly the layout the parser
// produces, so DeflateStream.optimise / mergeBlocks / write (d4g_host.h) run on them unchanged — the reference's
// `compressor.compress(data)` followed by `Deft.optimiseDeflateStream(out)` (C/CompressorTask.java:29-35) without
// serialising and re-parsing the intermediate stream.
#pragma once
#include "d4g_host.h"
#include "d4g_lz77.h"

namespace d4g {

struct LzSpec { int32_t input, encoder, strategy; };   // mirrors d4g_encoder_spec (include/deft4g.h)

// device buffers of the front end that must not outlive it: released before the batch's own phases start, and by the
// destructor when an exception unwinds the front end
struct LzScratch {
    std::vector<void*> v;
    template <typename T> T* own(T* p) { v.push_back((void*)p); return p; }
    void release() { for (void* p : v) rt_free(p); v.clear(); }
    ~LzScratch() { release(); }
};

struct LzFront {
    Batch& B;
    std::vector<LzSpec> specs;
    std::vector<i64> rawLen, rawU;
    explicit LzFront(Batch& b) : B(b) {}

    void create(size_t nIn, const uint8_t* const* raw, const size_t* len, size_t nOut, const LzSpec* sp, bool fromDevice = false) {
        memset(&B.stats, 0, sizeof(B.stats));
        double t0 = now_ms();
        specs.assign(sp, sp + nOut);
        rawLen.resize(nIn);
        rawU.resize(nIn);
        i64 off = 0;
        for (size_t i = 0; i < nIn; i++) {
            if ((i64)len[i] >= 0x7fff0000LL) throw std::runtime_error("input of 2 GiB or more: split it");
            rawLen[i] = (i64)len[i];
            rawU[i] = off;
            off += (((i64)len[i] + 15) & ~15LL) + 512;   // zero padding: the kernels read (never use) a few words past the end
            B.stats.bytes_decoded += (i64)len[i];
        }
        for (const LzSpec& s : specs)
            if (s.input < 0 || (size_t)s.input >= nIn || s.encoder < 0 || s.encoder > 1 || s.strategy < 0 || s.strategy > 2)
                throw std::runtime_error("bad encoder spec");
        B.dU = (uint8_t*)rt_malloc((size_t)off + 1024);
        rt_memset(B.dU, 0, (size_t)off + 1024);
        for (size_t i = 0; i < nIn; i++) {
            if (fromDevice) rt_d2d(B.dU + rawU[i], raw[i], len[i]);
            else rt_h2d(B.dU + rawU[i], raw[i], len[i]);
        }
        rt_sync();
        B.streams.resize(nOut);
        B.stats.n_streams = (i64)nOut;
        B.stats.ms_upload = now_ms() - t0;
    }

    // The whole front end: fills B.ps / B.streams / the device arrays, then runs the batch's own phases.
    void run(bool optimise, bool merge) {
        if (B.ran) throw std::runtime_error("batch already ran");
        B.ran = true;
        Engine& E = engine();
        E.init();
        double t0 = now_ms();
        const size_t nIn = rawLen.size(), nOut = specs.size();
        // ---- 1. streams, sort blocks, chunks ----
        std::vector<LzStream> hs(nIn);
        std::vector<LzSortJob> sortJobs;
        i64 posTot = 0;
        for (size_t i = 0; i < nIn; i++) {
            hs[i].data = B.dU + rawU[i];
            hs[i].len = rawLen[i];
            hs[i].posBase = posTot;
            hs[i].sortBlock0 = (int32_t)(posTot / LZ_SORT_BLOCK);
            hs[i].nChunks = (int32_t)((rawLen[i] + LZ_CHUNK - 1) / LZ_CHUNK);
            i64 nsb = (rawLen[i] + LZ_SORT_BLOCK - 1) / LZ_SORT_BLOCK;
            posTot += nsb * LZ_SORT_BLOCK;
        }
        // parses needed: (input, strategy) for DEFAULT / FILTERED; HUFFMAN_ONLY needs neither sort nor parse
        std::map<std::pair<int, int>, int> parseOf;
        struct Parse { int input, strategy; i64 metaBase; };
        std::vector<Parse> parses;
        std::vector<char> needSort(nIn, 0);
        i64 metaTot = 0;
        for (const LzSpec& s : specs) {
            if (s.strategy == LZ_HUFFMAN_ONLY) continue;
            auto key = std::make_pair((int)s.input, (int)s.strategy);
            if (parseOf.count(key)) continue;
            parseOf[key] = (int)parses.size();
            parses.push_back({s.input, s.strategy, metaTot});
            metaTot += hs[s.input].nChunks;
            needSort[s.input] = 1;
        }
        for (size_t i = 0; i < nIn; i++)
            if (needSort[i])
                for (i64 b = 0; b * LZ_SORT_BLOCK < rawLen[i]; b++) sortJobs.push_back({(int32_t)i, (int32_t)b});
        LzScratch tmp;
        LzStream* dStreamsLz = tmp.own((LzStream*)rt_malloc(nIn * sizeof(LzStream) + 16));
        rt_h2d(dStreamsLz, hs.data(), nIn * sizeof(LzStream));
        uint16_t *dS16 = nullptr, *dRank = nullptr, *dBstart = nullptr;
        LzChunkMeta* dMeta = nullptr;
        uint32_t* dChunkTok = nullptr;
        std::vector<LzChunkMeta> meta((size_t)metaTot);
        RtEvent e0, e1, e2;
        e0.record();
        if (!sortJobs.empty()) {
            dS16 = tmp.own((uint16_t*)rt_malloc((size_t)posTot * 2 + 64));
            dRank = tmp.own((uint16_t*)rt_malloc((size_t)posTot * 2 + 64));
            dBstart = tmp.own((uint16_t*)rt_malloc((size_t)posTot * 2 + 64));   // 32768 entries per sort block
            LzSortJob* dJobs = (LzSortJob*)rt_malloc(sortJobs.size() * sizeof(LzSortJob));
            rt_h2d(dJobs, sortJobs.data(), sortJobs.size() * sizeof(LzSortJob));
            RT_LAUNCH(k_lz_sort, sortJobs.size(), LZ_SORT_THREADS, dStreamsLz, dJobs, dS16, dRank, dBstart);
            B.stats.kernel_launches++;
            rt_sync();
            rt_free(dJobs);
        }
        e1.record();
        // ---- 2. parse: speculative pass over every chunk, then exact re-runs until entry == predecessor's exit ----
        LzCtx c;
        c.streams = dStreamsLz; c.S16 = dS16; c.rank16 = dRank; c.bstart = dBstart; c.errors = B.errors();
        c.meta = nullptr; c.chunkTok = nullptr;
        if (metaTot > 0) {
            dMeta = tmp.own((LzChunkMeta*)rt_malloc((size_t)metaTot * sizeof(LzChunkMeta)));
            dChunkTok = tmp.own((uint32_t*)rt_malloc((size_t)metaTot * (LZ_CHUNK + 2) * 4 + 64));
            c.meta = dMeta; c.chunkTok = dChunkTok;
            std::vector<LzParseJob> jobs;
            std::vector<int32_t> chunkIndex((size_t)metaTot), chunkParse((size_t)metaTot);
            for (size_t pi = 0; pi < parses.size(); pi++) {
                const Parse& P = parses[pi];
                for (int fc = 0; fc < hs[P.input].nChunks; fc += LZ_PARSE_MAXWAVES) jobs.push_back({P.input, fc, (int32_t)P.metaBase, P.strategy});
                for (int k = 0; k < hs[P.input].nChunks; k++) { chunkIndex[P.metaBase + k] = k; chunkParse[P.metaBase + k] = (int32_t)pi; }
            }
            LzParseJob* dJobs = (LzParseJob*)rt_malloc(std::max(jobs.size(), (size_t)metaTot) * sizeof(LzParseJob) + 16);
            rt_h2d(dJobs, jobs.data(), jobs.size() * sizeof(LzParseJob));
            RT_LAUNCH(k_lz_parse, jobs.size(), 64 * LZ_PARSE_MAXWAVES, c, dJobs, 0, (const uint8_t*)nullptr);
            B.stats.kernel_launches++;
            B.stats.lz_parse_passes = 1;
            int32_t* dIdx = tmp.own((int32_t*)rt_malloc((size_t)metaTot * 4 + 16));
            int32_t* dRedo = tmp.own((int32_t*)rt_malloc((size_t)metaTot * 4 + 16));
            uint8_t* dHeads = tmp.own((uint8_t*)rt_malloc((size_t)metaTot + 16));
            unsigned* dN = tmp.own((unsigned*)rt_malloc(16));
            rt_h2d(dIdx, chunkIndex.data(), (size_t)metaTot * 4);
            std::vector<uint8_t> heads((size_t)metaTot);
            for (int pass = 0;; pass++) {
                if (pass > 100000) throw std::runtime_error("lz77 parse did not converge");
                rt_memset(dN, 0, 4);
                RT_LAUNCH(k_lz_check, (metaTot + 255) / 256, 256, dMeta, dIdx, (int)metaTot, dRedo, dN);
                B.stats.kernel_launches++;
                unsigned nr = 0;
                rt_d2h(&nr, dN, 4);
                if (nr == 0) break;
                std::vector<int32_t> redo(nr);
                rt_d2h(redo.data(), dRedo, (size_t)nr * 4);
                // A run of consecutive disagreeing chunks is one job: its first chunk (the head) is re-run from its
                // predecessor's exit and the wave carries on through the run — and beyond, while exits keep differing
                // from recorded entries — until it falls into step or reaches another job's head.
                std::fill(heads.begin(), heads.end(), 0);
                std::vector<char> bad((size_t)metaTot, 0);
                for (unsigned k = 0; k < nr; k++) bad[redo[k]] = 1;
                std::vector<LzParseJob> rj;
                for (unsigned k = 0; k < nr; k++) {
                    const int m = redo[k];
                    if (chunkIndex[m] > 0 && bad[m - 1]) continue;   // inside a run: the run's head gets there
                    const Parse& P = parses[chunkParse[m]];
                    rj.push_back({P.input, chunkIndex[m], (int32_t)P.metaBase, P.strategy});
                    heads[m] = 1;
                }
                rt_h2d(dHeads, heads.data(), (size_t)metaTot);
                rt_h2d(dJobs, rj.data(), rj.size() * sizeof(LzParseJob));
                RT_LAUNCH(k_lz_parse, rj.size(), 64, c, dJobs, 1, (const uint8_t*)dHeads);
                B.stats.kernel_launches++;
                B.stats.lz_parse_passes++;
                B.stats.lz_chunks_rerun += nr;
            }
            rt_d2h(meta.data(), dMeta, (size_t)metaTot * sizeof(LzChunkMeta));
            rt_free(dJobs);
        }
        e2.record();
        B.check_device_errors();
        // ---- 3. output streams: symbol counts, block boundaries, array bases ----
        std::vector<LzOutStream> outs(nOut);
        std::vector<LzBlockDesc> blocks;
        std::vector<LzFillJob> fillJobs;
        i64 tokTot = 0, refTot = 0;
        std::vector<i64> outSyms(nOut), outRefs(nOut);
        std::vector<std::vector<i64>> symPreOf(nOut), refPreOf(nOut);
        std::vector<char> lastIsMatchOf(nOut, 0);
        // jzlib flavour: its early-flush decisions need prefix sums at symbol granularity -> k_lz_split
        std::vector<LzSplitJob> splitJobs;
        std::vector<size_t> splitOut;
        std::vector<i64> preSym, preRef, preDc;
        i64 splitSlots = 0;
        for (size_t oi = 0; oi < nOut; oi++) {
            const LzSpec& sp = specs[oi];
            const LzStream& st = hs[sp.input];
            LzOutStream& o = outs[oi];
            o.stream = sp.input;
            o.metaBase = -1;
            std::vector<i64>&symPre = symPreOf[oi], &refPre = refPreOf[oi];
            symPre.assign(st.nChunks + 1, 0);
            refPre.assign(st.nChunks + 1, 0);
            std::vector<i64> dcPre(st.nChunks + 1, 0);
            bool lastIsMatch = false;
            if (sp.strategy != LZ_HUFFMAN_ONLY) {
                const Parse& P = parses[parseOf[std::make_pair((int)sp.input, (int)sp.strategy)]];
                o.metaBase = (int32_t)P.metaBase;
                for (int k = 0; k < st.nChunks; k++) {
                    const LzChunkMeta& m = meta[P.metaBase + k];
                    symPre[k + 1] = symPre[k] + m.ntok;
                    refPre[k + 1] = refPre[k] + m.nmatch;
                    dcPre[k + 1] = dcPre[k] + m.dcost;
                    if (m.ntok) lastIsMatch = (m.pad >> 9) != 0;
                }
            } else {
                for (int k = 0; k < st.nChunks; k++) symPre[k + 1] = std::min<i64>(st.len, (i64)(k + 1) * LZ_CHUNK);
            }
            lastIsMatchOf[oi] = lastIsMatch;
            outSyms[oi] = symPre[st.nChunks];
            outRefs[oi] = refPre[st.nChunks];
            if (sp.encoder == LZ_FLAVOR_JZLIB) {
                LzSplitJob J;
                memset(&J, 0, sizeof(J));
                J.stream = sp.input; J.metaBase = o.metaBase; J.preBase = (i64)preSym.size();
                J.nSyms = outSyms[oi]; J.nRefs = outRefs[oi]; J.dcostTotal = dcPre[st.nChunks];
                J.outBase = splitSlots; J.maxBlocks = (int32_t)(outSyms[oi] / 8192 + 3); J.lastIsMatch = lastIsMatch;
                splitSlots += J.maxBlocks;
                preSym.insert(preSym.end(), symPre.begin(), symPre.end());
                preRef.insert(preRef.end(), refPre.begin(), refPre.end());
                preDc.insert(preDc.end(), dcPre.begin(), dcPre.end());
                splitJobs.push_back(J);
                splitOut.push_back(oi);
            }
        }
        std::vector<LzSplitOut> splitRes((size_t)splitSlots);
        std::vector<int32_t> splitCnt(splitJobs.size());
        if (!splitJobs.empty()) {
            LzSplitJob* dJ = (LzSplitJob*)rt_malloc(splitJobs.size() * sizeof(LzSplitJob));
            long long *dA = (long long*)rt_malloc(preSym.size() * 8 + 16), *dBp = (long long*)rt_malloc(preSym.size() * 8 + 16),
                      *dC = (long long*)rt_malloc(preSym.size() * 8 + 16);
            LzSplitOut* dO = (LzSplitOut*)rt_malloc((size_t)splitSlots * sizeof(LzSplitOut) + 16);
            int32_t* dCnt = (int32_t*)rt_malloc(splitJobs.size() * 4 + 16);
            rt_h2d(dJ, splitJobs.data(), splitJobs.size() * sizeof(LzSplitJob));
            rt_h2d(dA, preSym.data(), preSym.size() * 8);
            rt_h2d(dBp, preRef.data(), preRef.size() * 8);
            rt_h2d(dC, preDc.data(), preDc.size() * 8);
            RT_LAUNCH(k_lz_split, splitJobs.size(), 64, c, dJ, dA, dBp, dC, dO, dCnt);
            B.stats.kernel_launches++;
            rt_d2h(splitRes.data(), dO, (size_t)splitSlots * sizeof(LzSplitOut));
            rt_d2h(splitCnt.data(), dCnt, splitJobs.size() * 4);
            rt_free(dJ); rt_free(dA); rt_free(dBp); rt_free(dC); rt_free(dO); rt_free(dCnt);
        }
        size_t splitIdx = 0;
        for (size_t oi = 0; oi < nOut; oi++) {
            const LzSpec& sp = specs[oi];
            const LzStream& st = hs[sp.input];
            LzOutStream& o = outs[oi];
            const i64 N = outSyms[oi], R = outRefs[oi];
            o.blkBase = (i64)blocks.size();
            auto push = [&](i64 symStart, i64 symCount, int isLast) {
                LzBlockDesc d;
                memset(&d, 0, sizeof(d));
                d.symStart = symStart; d.symCount = symCount; d.out = (int32_t)oi; d.isLast = isLast;
                d.uStart = d.uLen = st.len; d.refStart = d.refCount = R;   // (an empty block: overwritten for the others)
                blocks.push_back(d);
            };
            if (sp.encoder == LZ_FLAVOR_JZLIB) {
                const LzSplitJob& J = splitJobs[splitIdx];
                int nb = splitCnt[splitIdx++];
                if (nb > J.maxBlocks) throw std::runtime_error("lz77: block list overflow");
                for (int b = 0; b < nb; b++) push(splitRes[J.outBase + b].symStart, splitRes[J.outBase + b].symCount, splitRes[J.outBase + b].isLast);
            } else {
                // zlib: a block is flushed after LZ_SYMS_PER_BLOCK symbols (lit_bufsize - 1).  The symbol that fills a
                // block exactly at the end of the input flushes it as a non-last block — and an empty last block
                // follows — unless it is the literal deflate_slow's epilogue emits after its loop (its flush flag is
                // ignored); deflate_huff (zlib's HUFFMAN_ONLY) has no such epilogue.
                const bool epilogueLiteral = sp.strategy != LZ_HUFFMAN_ONLY && !lastIsMatchOf[oi];
                i64 cur = 0;
                while (true) {
                    i64 take = std::min<i64>(LZ_SYMS_PER_BLOCK, N - cur);
                    bool endsStream = cur + take == N;
                    bool last = endsStream && (take < LZ_SYMS_PER_BLOCK || (N > 0 && epilogueLiteral));
                    push(cur, take, last);
                    cur += take;
                    if (last) break;
                    if (endsStream) { push(N, 0, 1); break; }
                }
            }
            o.nBlocks = (int32_t)(blocks.size() - o.blkBase);
            o.tokBase = tokTot;
            o.refBase = refTot;
            o.pad = 0;
            tokTot += N + o.nBlocks;
            refTot += R;
            for (int k = 0; k < std::max(1, (int)st.nChunks); k++)
                fillJobs.push_back({(int32_t)oi, (int32_t)k, k < st.nChunks ? symPreOf[oi][k] : 0, k < st.nChunks ? refPreOf[oi][k] : 0});
        }
        if (refTot >= (1LL << 32)) throw std::runtime_error("batch holds 2^32 or more back-references: split it");
        // ---- 4. tokens / records in the optimiser's layout ----
        RtEvent e3, e4;
        e3.record();
        B.dTok = (uint2*)rt_malloc((size_t)tokTot * 8 + 64);
        B.dRefs = (uint4*)rt_malloc((size_t)refTot * 16 + 64);
        B.dTokRef = (uint32_t*)rt_malloc((size_t)tokTot * 4 + 64);
        const size_t nBlk = blocks.size();
        LzOutStream* dOuts = tmp.own((LzOutStream*)rt_malloc(nOut * sizeof(LzOutStream) + 16));
        LzBlockDesc* dBlk = tmp.own((LzBlockDesc*)rt_malloc(nBlk * sizeof(LzBlockDesc) + 16));
        LzFillJob* dFill = tmp.own((LzFillJob*)rt_malloc(fillJobs.size() * sizeof(LzFillJob) + 16));
        rt_h2d(dOuts, outs.data(), nOut * sizeof(LzOutStream));
        rt_h2d(dBlk, blocks.data(), nBlk * sizeof(LzBlockDesc));
        rt_h2d(dFill, fillJobs.data(), fillJobs.size() * sizeof(LzFillJob));
        if (!fillJobs.empty()) {
            RT_LAUNCH(k_lz_fill, fillJobs.size(), 64, c, dOuts, dFill, dBlk, B.dTok, B.dRefs, B.dTokRef);
            B.stats.kernel_launches++;
        }
        rt_d2h(blocks.data(), dBlk, nBlk * sizeof(LzBlockDesc));
        for (LzBlockDesc& d : blocks)
            if (d.symCount > 0) { d.uLen -= d.uStart; d.refCount -= d.refStart; }
            else { d.uLen = 0; d.refCount = 0; }
        rt_h2d(dBlk, blocks.data(), nBlk * sizeof(LzBlockDesc));
        // ---- 5. per block: zlib's trees and block type, and the block's state ----
        D4GState* dTmpStates = tmp.own((D4GState*)rt_malloc(nBlk * sizeof(D4GState) + 16));
        LzBlockOut* dBo = tmp.own((LzBlockOut*)rt_malloc(nBlk * sizeof(LzBlockOut) + 16));
        std::vector<LzBlockOut> bo(nBlk);
        if (nBlk) {
            RT_LAUNCH(k_lz_blocks, nBlk, 64, dStreamsLz, dOuts, dBlk, (int)nBlk, B.dTok, dTmpStates, dBo);
            B.stats.kernel_launches++;
            rt_d2h(bo.data(), dBo, nBlk * sizeof(LzBlockOut));
        }
        e4.record();
        // ---- 6. the batch's own view: parsed-stream descriptions, layout, states in slot 0 ----
        B.ps.assign(nOut, Batch::PStream());
        for (size_t oi = 0; oi < nOut; oi++) {
            Batch::PStream& P = B.ps[oi];
            const LzOutStream& o = outs[oi];
            P.uBaseFixed = rawU[specs[oi].input];
            P.nU = rawLen[specs[oi].input];
            i64 spos = 0;
            for (int b = 0; b < o.nBlocks; b++) {
                const LzBlockDesc& d = blocks[o.blkBase + b];
                const LzBlockOut& r = bo[o.blkBase + b];
                Batch::PBlock pb;
                pb.type = r.type; pb.bfinal = d.isLast; pb.bitPos = 0; pb.endBit = 0;
                pb.nTok = d.symCount + 1; pb.uLen = d.uLen; pb.sizeBits = r.sizeBits; pb.nRef = d.refCount; pb.firstBatch = -1;
                pb.refSpan = d.refCount;
                P.blocks.push_back(pb);
                P.nTok += pb.nTok;
                spos += 3;
                if (r.type == D4G_STORED) {
                    i64 al = spos % 8;
                    al = al == 0 ? 0 : 8 - al;
                    spos += (d.uLen + 4) * 8 + al;
                } else spos += r.sizeBits;
            }
            P.sizeBits = spos;
            P.consumed = (spos + 7) / 8;
        }
        Batch::Layout LY;
        B.layout_blocks(merge, optimise, LY);
        if (nBlk) {
            std::vector<long long> dst(nBlk, -1);
            for (size_t oi = 0; oi < nOut; oi++)
                for (int b = 0; b < outs[oi].nBlocks; b++) {
                    const HBlock& hb = B.streams[oi].blocks[b];
                    if (hb.gpu >= 0) dst[outs[oi].blkBase + b] = B.hBlocks[hb.gpu].stateIdx;
                }
            long long* dDst = (long long*)rt_malloc(nBlk * 8 + 16);
            rt_h2d(dDst, dst.data(), nBlk * 8);
            RT_LAUNCH(k_lz_place_states, nBlk, 256, dTmpStates, dDst, (int)nBlk, B.dStates);
            B.stats.kernel_launches++;
            rt_sync();
            rt_free(dDst);
        }
        void* binList = B.block_bins(LY.realBlocks, optimise);
        rt_sync();
        rt_free(binList);
        B.stats.ms_lz_sort = rt_elapsed_ms(e0, e1);
        B.stats.ms_lz_parse = rt_elapsed_ms(e1, e2);
        B.stats.ms_lz_emit = rt_elapsed_ms(e3, e4);
        B.stats.lz_symbols = 0;
        for (size_t oi = 0; oi < nOut; oi++) B.stats.lz_symbols += outSyms[oi];
        tmp.release();   // the sort arrays, chunk tokens and block tables are done with: the candidate search needs the memory
        B.check_device_errors();
        double t1 = now_ms();
        if (optimise) B.phase1();
        double t2 = now_ms();
        if (optimise && merge) B.phase_merge();
        double t3 = now_ms();
        B.phase_write();
        double t4 = now_ms();
        B.stats.ms_parse = t1 - t0;     // here: the encoder front end
        B.stats.ms_optimise = t2 - t1;
        B.stats.ms_merge = t3 - t2;
        B.stats.ms_write = t4 - t3;
        B.stats.ms_total = t4 - t0;
        B.stats.ms_search_kernels = B.msSearch;
        B.stats.search_bytes_algorithmic = B.stats.bytes_decoded + B.stats.bytes_out;
        B.release_scratch();
    }
};

}  // namespace d4g
