// hipsim.cpp — TEST-ONLY fiber scheduler of the SIMT emulator (see hipsim.h) plus the
// translation unit that compiles the library's kernels and host code for the CPU.
#include "hipsim.h"

SimDim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipsim {

static const size_t STACK = 256 * 1024;

Sim& sim() {
    static Sim s;
    return s;
}

void yield_to_scheduler() {
    Sim& s = sim();
    Fiber* f = s.cur;
    swapcontext(&f->ctx, &s.sched);
}

long long collective(long long v) {
    Sim& s = sim();
    Fiber* f = s.cur;
    int lane = f->tid & 63;
    WaveBuf& w = s.waves[f->tid >> 6];
    int p = f->parity;
    w.val[p][lane] = v;
    w.arrived[p][lane] = true;
    f->state = AT_COLLECTIVE;
    yield_to_scheduler();
    f->parity ^= 1;
    return p;
}

static void fiber_entry() {
    Sim& s = sim();
    s.body();
    s.cur->state = DONE;
    swapcontext(&s.cur->ctx, &s.sched);
}

void run_grid(unsigned grid, unsigned block, const std::function<void()>& body) {
    Sim& s = sim();
    s.body = body;
    s.launches++;
    if (s.fibers.size() < block) {
        size_t old = s.fibers.size();
        s.fibers.resize(block);
        for (size_t i = old; i < block; i++) s.fibers[i].stack = (char*)malloc(STACK);
    }
    unsigned nWaves = (block + 63) / 64;
    s.waves.resize(nWaves);
    gridDim = {grid, 1, 1};
    blockDim = {block, 1, 1};
    for (unsigned b = 0; b < grid; b++) {
        blockIdx = {b, 0, 0};
        for (unsigned t = 0; t < block; t++) {
            Fiber& f = s.fibers[t];
            f.state = READY;
            f.tid = (int)t;
            f.parity = 0;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, fiber_entry, 0);
        }
        for (auto& w : s.waves) memset(&w, 0, sizeof(w));
        while (true) {
            bool progressed = false, anyLive = false;
            for (unsigned t = 0; t < block; t++) {
                Fiber& f = s.fibers[t];
                if (f.state != READY) continue;
                progressed = true;
                s.cur = &f;
                threadIdx = {t, 0, 0};
                swapcontext(&s.sched, &f.ctx);
            }
            // release waves whose live lanes all reached the same collective
            for (unsigned w = 0; w < nWaves; w++) {
                unsigned lo = w * 64, hi = std::min(block, lo + 64);
                int nColl = 0, nBar = 0, nReady = 0;
                for (unsigned t = lo; t < hi; t++) {
                    int st = s.fibers[t].state;
                    nColl += st == AT_COLLECTIVE;
                    nBar += st == AT_BARRIER;
                    nReady += st == READY;
                }
                if (nColl && !nReady) {
                    if (nBar) { fprintf(stderr, "hipsim: %s block %u wave %u diverged (collective vs __syncthreads)\n", s.kname, b, w); abort(); }
                    int p = -1;
                    for (unsigned t = lo; t < hi; t++)
                        if (s.fibers[t].state == AT_COLLECTIVE) {
                            if (p == -1) p = s.fibers[t].parity;
                            else if (p != s.fibers[t].parity) { fprintf(stderr, "hipsim: collective parity mismatch\n"); abort(); }
                        }
                    for (int l = 0; l < 64; l++) s.waves[w].arrived[p ^ 1][l] = false;
                    for (unsigned t = lo; t < hi; t++)
                        if (s.fibers[t].state == AT_COLLECTIVE) s.fibers[t].state = READY;
                    progressed = true;
                }
            }
            int nBar = 0, nOther = 0;
            for (unsigned t = 0; t < block; t++) {
                int st = s.fibers[t].state;
                if (st != DONE) anyLive = true;
                if (st == AT_BARRIER) nBar++;
                else if (st != DONE) nOther++;
            }
            if (!anyLive) break;
            if (nBar && !nOther) {
                for (unsigned t = 0; t < block; t++)
                    if (s.fibers[t].state == AT_BARRIER) s.fibers[t].state = READY;
                progressed = true;
            }
            if (!progressed) { fprintf(stderr, "hipsim: deadlock in block %u\n", b); abort(); }
        }
    }
}

}  // namespace hipsim

#include "../../deft4j_amd/csrc/libdeft4g.hip"
