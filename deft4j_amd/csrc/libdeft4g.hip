// libdeft4g.hip — the single translation unit of libdeft4g.so (kernels + host + C ABI).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -o libdeft4g.so libdeft4g.hip
#include <functional>
#include <mutex>
#include <thread>

#include "d4g_host.h"
#include "d4g_lz77_host.h"
#include "d4g_zopfli_host.h"

using namespace d4g;

struct d4g_batch {
    int ctx = 0;   // the context (device) the batch lives on: every call with the batch works there
    Batch impl;
    std::unique_ptr<LzFront> lz;   // set for batches made by d4g_batch_create_encode
    // d4g_batch_run_recompress: the re-optimised winners of the recompression (its own copy of their bytes) and which streams took them
    std::unique_ptr<Batch> reopt;
    std::vector<int> reoptIndex;       // stream -> index in reopt (-1: none)
    std::vector<char> graft;
    std::vector<int64_t> recompSaved;
    // where stream i's final bytes live
    const HStream& final_stream(size_t i, const Batch** owner) const {
        if (i < graft.size() && graft[i]) { *owner = reopt.get(); return reopt->streams[reoptIndex[i]]; }
        *owner = &impl;
        return impl.streams[i];
    }
};

namespace {
// CompressionUtil calls in from a thread pool (C/CompressionUtil.java:111-117).  d4g_init / d4g_shutdown are exclusive;
// everything else runs concurrently: a batch belongs to the thread that is calling with it, every host thread has its own
// HIP streams (d4g_rt.h) and the shared pieces (memory pool, engine set-up) take their own locks.  The test-only CPU
// emulator is single-threaded, so its build keeps one library-wide lock.
std::mutex g_mu;
#ifdef D4G_HOSTSIM
#define D4G_API_LOCK() std::lock_guard<std::mutex> lk(g_mu)
#else
#define D4G_API_LOCK() do { } while (0)
#endif
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
bool ready() { return rt().ready; }
// The context of the calling thread for this call: the batch's, or — for calls that create batches / one-shot calls — the
// thread's choice (d4g_set_device; context 0 unless it chose).
thread_local int g_tlsCtx = 0;
int g_nCtx = 0;   // contexts initialised (d4g_init: 1)
void enter_ctx(const d4g_batch* b) { rt_ctx() = b ? b->ctx : g_tlsCtx; }
bool rtp_ctx_ready(int k) {
#ifdef D4G_HOSTSIM
    return k < g_nCtx;
#else
    return rtp().ctx[k].ready;
#endif
}
// HIP's current device is per thread: CompressionUtil's pool threads (and any rank with device != 0) must bind
// the library's device before allocating or launching.  Called at the top of every entry point (d4g_init / d4g_shutdown hold g_mu;
// the others run concurrently, each on its calling thread).
void bind_device() {
#ifndef D4G_HOSTSIM
    if (rt().ready && rt().device >= 0) RT_CHECK(hipSetDevice(rt().device));
#endif
}
}  // namespace

extern "C" {

const char* d4g_last_error(void) { return g_err.c_str(); }

// one context: device + memory pool + programs (caller holds g_mu)
static int init_ctx(int ctx, int device_index) {
    rt_ctx() = ctx;
#ifndef D4G_HOSTSIM
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(D4G_ERR_NODEVICE, "no HIP device available (libdeft4g has no CPU fallback)");
    if (device_index < 0 || device_index >= n) return fail(D4G_ERR_ARG, "device index out of range");
    if (rt().ready && rt().device != device_index)
        return fail(D4G_ERR_ARG, "already initialised on device " + std::to_string(rt().device) + ": call d4g_shutdown() before selecting another device");
    RT_CHECK(hipSetDevice(device_index));
    rt().device = device_index;
    if (const char* mb = getenv("D4G_POOL_MAX_MB")) rt_pool().maxHeldBytes = (size_t)atoll(mb) << 20;
#endif
    rt().ready = true;
    (void)rt();           // this thread's streams
    engine().init();
    return D4G_OK;
}

int d4g_init(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    try {
        int rc = init_ctx(0, device_index);
        if (rc == D4G_OK && g_nCtx < 1) g_nCtx = 1;
        rt_ctx() = g_tlsCtx;
        return rc;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_init_devices(int n, const int* device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n < 1 || n > RT_MAX_CTX || !device_index) return fail(D4G_ERR_ARG, "1 to 16 contexts");
    try {
        for (int k = 0; k < n; k++) {
            int rc = init_ctx(k, device_index[k]);
            if (rc != D4G_OK) { rt_ctx() = g_tlsCtx; return rc; }
        }
        g_nCtx = std::max(g_nCtx, n);
        rt_ctx() = g_tlsCtx;
        return D4G_OK;
    } catch (const std::exception& ex) {
        rt_ctx() = g_tlsCtx;
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_device_count(void) { return g_nCtx; }

int d4g_set_device(int context) {
    if (context < 0 || context >= RT_MAX_CTX || !rtp_ctx_ready(context)) return fail(D4G_ERR_ARG, "no such context (d4g_init_devices)");
    g_tlsCtx = context;
    rt_ctx() = context;
    return D4G_OK;
}

void d4g_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int k = 0; k < RT_MAX_CTX; k++) {
        rt_ctx() = k;
        if (!rt().ready) continue;
        try {
            bind_device();
            rt_sync_all();
            engine().release();
#ifndef D4G_HOSTSIM
            rt_pool_release();
            {   // every host thread's streams of this context (no batch may be running: shutdown is exclusive by contract)
                std::vector<RtGlobals*> all;
                { std::lock_guard<std::mutex> lk2(rtp().mu); all = rtp().threads; }
                for (RtGlobals* g : all)
                    if (g->ctxIdx == k) g->destroy_streams();
            }
            rt().device = -1;
#endif
        } catch (const std::exception&) {
        }
        rt().ready = false;
    }
    g_nCtx = 0;
    g_tlsCtx = 0;
    rt_ctx() = 0;
}

d4g_batch* d4g_batch_create(size_t n, const uint8_t* const* in, const size_t* in_len) {
    enter_ctx(nullptr);
    D4G_API_LOCK();
    if (!ready()) { fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded"); return nullptr; }
    try {
        bind_device();
        std::unique_ptr<d4g_batch> b(new d4g_batch());
        b->ctx = rt_ctx();
        b->impl.create(n, in, in_len);
        return b.release();
    } catch (const std::exception& ex) {
        fail(D4G_ERR_RUNTIME, ex.what());
        return nullptr;
    }
}

int d4g_batch_run(d4g_batch* b, int merge_blocks) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    if (!b) return fail(D4G_ERR_ARG, "null batch");
    if (b->lz) return fail(D4G_ERR_ARG, "encoder batch: use d4g_batch_run_encode");
    try {
        bind_device();
        b->impl.run(merge_blocks != 0);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

d4g_batch* d4g_batch_create_encode(size_t n_in, const uint8_t* const* raw, const size_t* raw_len, size_t n_out,
                                   const d4g_encoder_spec* spec) {
    enter_ctx(nullptr);
    D4G_API_LOCK();
    if (!ready()) { fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded"); return nullptr; }
    if ((n_in && (!raw || !raw_len)) || (n_out && !spec)) { fail(D4G_ERR_ARG, "null argument"); return nullptr; }
    try {
        bind_device();
        static_assert(sizeof(LzSpec) == sizeof(d4g_encoder_spec), "spec layout");
        std::unique_ptr<d4g_batch> b(new d4g_batch());
        b->ctx = rt_ctx();
        b->lz.reset(new LzFront(b->impl));
        b->lz->create(n_in, raw, raw_len, n_out, (const LzSpec*)spec);
        return b.release();
    } catch (const std::exception& ex) {
        fail(D4G_ERR_RUNTIME, ex.what());
        return nullptr;
    }
}

int d4g_batch_run_encode(d4g_batch* b, int optimise, int merge_blocks) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    if (!b || !b->lz) return fail(D4G_ERR_ARG, "not an encoder batch");
    try {
        bind_device();
        b->lz->run(optimise != 0, merge_blocks != 0);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_deflate_streams(size_t n, const uint8_t* const* raw, const size_t* raw_len, int encoder, int strategy, uint8_t** out,
                        size_t* out_len) {
    enter_ctx(nullptr);
    if (n && (!raw || !raw_len || !out || !out_len)) return fail(D4G_ERR_ARG, "null argument");
    for (size_t i = 0; i < n; i++) { out[i] = nullptr; out_len[i] = 0; }
    std::vector<d4g_encoder_spec> sp(n);
    for (size_t i = 0; i < n; i++) sp[i] = {(int32_t)i, encoder, strategy};
    d4g_batch* b = d4g_batch_create_encode(n, raw, raw_len, n, sp.data());
    if (!b) return D4G_ERR_RUNTIME;
    int rc = d4g_batch_run_encode(b, 0, 0);
    for (size_t i = 0; i < n && rc == D4G_OK; i++) {
        size_t ol = 0;
        d4g_batch_stream_result(b, i, nullptr, nullptr, &ol, nullptr, nullptr);
        out[i] = (uint8_t*)malloc(ol ? ol : 1);
        if (!out[i]) { rc = fail(D4G_ERR_RUNTIME, "out of host memory"); break; }
        rc = d4g_batch_copy_output(b, i, out[i], ol);
        out_len[i] = ol;
    }
    if (rc != D4G_OK)
        for (size_t i = 0; i < n; i++) { free(out[i]); out[i] = nullptr; out_len[i] = 0; }
    std::string keep = g_err;
    d4g_batch_destroy(b);
    g_err = keep;
    return rc;
}

int d4g_batch_stream_result(d4g_batch* b, size_t i, int32_t* status, int64_t* saved_bits, size_t* out_len, size_t* consumed,
                            int64_t* size_bits_in) {
    enter_ctx(b);
    if (!b || i >= b->impl.streams.size()) return fail(D4G_ERR_ARG, "bad stream index");
    const HStream& s = b->impl.streams[i];
    const Batch* owner = nullptr;
    const HStream& f = b->final_stream(i, &owner);
    const bool grafted = owner != &b->impl;
    int st = s.status != 0 ? D4G_STREAM_PARSE_ERROR : ((s.saved > 0 || grafted) ? D4G_STREAM_CHANGED : D4G_STREAM_UNCHANGED);
    if (status) *status = st;
    if (saved_bits) *saved_bits = s.status == 0 ? s.saved : 0;
    if (out_len) *out_len = s.status == 0 ? (size_t)((f.outBits + 7) / 8) : 0;
    if (consumed) *consumed = (size_t)s.consumed;
    if (size_bits_in) *size_bits_in = s.status == 0 ? s.sizeBitsIn : -1;
    return D4G_OK;
}

int d4g_batch_copy_output(d4g_batch* b, size_t i, uint8_t* dst, size_t cap) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!b || i >= b->impl.streams.size()) return fail(D4G_ERR_ARG, "bad stream index");
    if (b->impl.streams[i].status != 0) return fail(D4G_ERR_ARG, "stream did not parse");
    const Batch* owner = nullptr;
    const HStream& s = b->final_stream(i, &owner);
    size_t n = (size_t)((s.outBits + 7) / 8);
    if (cap < n) return fail(D4G_ERR_ARG, "output buffer too small");
    try {
        bind_device();
        rt_d2h(dst, (const uint8_t*)(owner->dOut + s.outWordBase), n);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_batch_copy_decoded(d4g_batch* b, size_t i, uint8_t* dst, size_t cap, size_t* len) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!b || i >= b->impl.streams.size()) return fail(D4G_ERR_ARG, "bad stream index");
    const HStream& s = b->impl.streams[i];
    if (s.status != 0) return fail(D4G_ERR_ARG, "stream did not parse");
    if (len) *len = (size_t)s.nU;
    if (!dst) return D4G_OK;
    if (cap < (size_t)s.nU) return fail(D4G_ERR_ARG, "output buffer too small");
    try {
        bind_device();
        rt_d2h(dst, b->impl.dU + s.uBase, (size_t)s.nU);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_batch_parse(d4g_batch* b) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    if (!b) return fail(D4G_ERR_ARG, "null batch");
    try {
        bind_device();
        if (b->impl.ran) return fail(D4G_ERR_ARG, "batch already ran");
        b->impl.ran = true;
        engine().init();
        b->impl.parse_probe();
        b->impl.build_blocks(false, false);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_batch_checksums(d4g_batch* b, size_t i, uint32_t* crc32, uint32_t* adler32, int64_t* isize) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!b || i >= b->impl.streams.size()) return fail(D4G_ERR_ARG, "bad stream index");
    if (b->impl.streams[i].status != 0) return fail(D4G_ERR_ARG, "stream did not parse");
    try {
        bind_device();
        b->impl.checksums();
        const D4GCsumOut& o = b->impl.csums[i];
        if (crc32) *crc32 = o.crc32;
        if (adler32) *adler32 = o.adler32;
        if (isize) *isize = o.isize;
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_batch_stats(d4g_batch* b, d4g_stats* st) {
    enter_ctx(b);
    if (!b || !st) return fail(D4G_ERR_ARG, "null argument");
    *st = b->impl.stats;
    return D4G_OK;
}

void d4g_batch_destroy(d4g_batch* b) {
    enter_ctx(b);
    D4G_API_LOCK();
    try { bind_device(); } catch (const std::exception&) {}
    delete b;
}

int d4g_optimise_streams(size_t n, const uint8_t* const* in, const size_t* in_len, int merge_blocks, uint8_t** out,
                         size_t* out_len, int64_t* saved_bits, int32_t* status) {
    enter_ctx(nullptr);
    if (n && (!in || !in_len || !out || !out_len || !status)) return fail(D4G_ERR_ARG, "null argument");
    // every result slot is defined whatever happens below: "keep the original" until a stream is known to have changed
    for (size_t i = 0; i < n; i++) {
        out[i] = nullptr;
        out_len[i] = 0;
        status[i] = D4G_STREAM_UNCHANGED;
        if (saved_bits) saved_bits[i] = 0;
    }
    d4g_batch* b = d4g_batch_create(n, in, in_len);
    if (!b) return D4G_ERR_RUNTIME;
    int rc = d4g_batch_run(b, merge_blocks);
    for (size_t i = 0; i < n && rc == D4G_OK; i++) {
        int64_t sv = 0;
        size_t ol = 0;
        int32_t st = D4G_STREAM_UNCHANGED;
        d4g_batch_stream_result(b, i, &st, &sv, &ol, nullptr, nullptr);
        if (st == D4G_STREAM_CHANGED) {
            out[i] = (uint8_t*)malloc(ol ? ol : 1);
            if (!out[i]) { rc = fail(D4G_ERR_RUNTIME, "out of host memory"); break; }
            rc = d4g_batch_copy_output(b, i, out[i], ol);
            if (rc != D4G_OK) break;
            out_len[i] = ol;
        }
        status[i] = st;
        if (saved_bits) saved_bits[i] = sv;
    }
    if (rc != D4G_OK)   // all-or-nothing: a failed call hands back no buffers
        for (size_t i = 0; i < n; i++) {
            free(out[i]);
            out[i] = nullptr;
            out_len[i] = 0;
            status[i] = D4G_STREAM_UNCHANGED;
            if (saved_bits) saved_bits[i] = 0;
        }
    std::string keep = g_err;
    d4g_batch_destroy(b);
    g_err = keep;
    return rc;
}

d4g_batch* d4g_batch_create_on(int context, size_t n, const uint8_t* const* in, const size_t* in_len) {
    if (context < 0 || context >= RT_MAX_CTX || !rtp_ctx_ready(context)) { fail(D4G_ERR_ARG, "no such context (d4g_init_devices)"); return nullptr; }
    const int keep = g_tlsCtx;
    g_tlsCtx = context;
    d4g_batch* b = d4g_batch_create(n, in, in_len);
    g_tlsCtx = keep;
    return b;
}

// DeflateFilesContainer.optimise(List<DeflateStream>, boolean) over every initialised context (K/DeflateFilesContainer.java:18-43:
// the streams are independent): longest-processing-time-first partition by compressed size, one host thread and one batch per
// context, no exchange between the devices; the outputs are gathered in the caller's arrays (host memory) in stream order.
int d4g_optimise_streams_sharded(size_t n, const uint8_t* const* in, const size_t* in_len, int merge_blocks, uint8_t** out,
                                 size_t* out_len, int64_t* saved_bits, int32_t* status) {
    enter_ctx(nullptr);
    const int nc = g_nCtx;
    if (nc <= 1) return d4g_optimise_streams(n, in, in_len, merge_blocks, out, out_len, saved_bits, status);
    if (n && (!in || !in_len || !out || !out_len || !status)) return fail(D4G_ERR_ARG, "null argument");
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return in_len[a] > in_len[b]; });
    std::vector<std::vector<size_t>> shard(nc);
    std::vector<unsigned long long> load(nc, 0);
    for (size_t i : order) {
        int r = 0;
        for (int k = 1; k < nc; k++) if (load[k] < load[r]) r = k;
        shard[r].push_back(i);
        load[r] += in_len[i];
    }
    std::vector<int> rcs(nc, D4G_OK);
    std::vector<std::string> errs(nc);
    std::vector<std::thread> th;
    for (int k = 0; k < nc; k++) {
        std::sort(shard[k].begin(), shard[k].end());
        th.emplace_back([&, k]() {
            g_tlsCtx = k;   // this thread works on context k
            const std::vector<size_t>& idx = shard[k];
            const size_t m = idx.size();
            std::vector<const uint8_t*> sin(m);
            std::vector<size_t> slen(m), solen(m);
            std::vector<uint8_t*> sout(m);
            std::vector<int64_t> ssaved(m);
            std::vector<int32_t> sst(m);
            for (size_t j = 0; j < m; j++) { sin[j] = in[idx[j]]; slen[j] = in_len[idx[j]]; }
            rcs[k] = d4g_optimise_streams(m, sin.data(), slen.data(), merge_blocks, sout.data(), solen.data(), ssaved.data(), sst.data());
            if (rcs[k] != D4G_OK) errs[k] = g_err;
            for (size_t j = 0; j < m; j++) {
                out[idx[j]] = sout[j]; out_len[idx[j]] = solen[j]; status[idx[j]] = sst[j];
                if (saved_bits) saved_bits[idx[j]] = ssaved[j];
            }
        });
    }
    for (auto& t : th) t.join();
    int rc = D4G_OK;
    for (int k = 0; k < nc; k++)
        if (rcs[k] != D4G_OK) { rc = rcs[k]; g_err = errs[k]; }
    if (rc != D4G_OK)   // all-or-nothing, like the single-device call
        for (size_t i = 0; i < n; i++) {
            free(out[i]);
            out[i] = nullptr;
            out_len[i] = 0;
            status[i] = D4G_STREAM_UNCHANGED;
            if (saved_bits) saved_bits[i] = 0;
        }
    return rc;
}

int d4g_size_bits_fallback(const uint8_t* in, size_t len, int64_t* bits) {
    enter_ctx(nullptr);
    if (!in || !bits) return fail(D4G_ERR_ARG, "null argument");
    const uint8_t* ins[1] = {in};
    size_t lens[1] = {len};
    d4g_batch* b = d4g_batch_create(1, ins, lens);
    if (!b) return D4G_ERR_RUNTIME;
    int rc;
    {
        D4G_API_LOCK();
        try {
            bind_device();
            engine().init();
            b->impl.parse_probe();
            const Batch::PStream& o = b->impl.ps[0];
            *bits = o.status == 0 ? o.sizeBits : (int64_t)len * 8;  // B/Deft.java:48-54
            rc = D4G_OK;
        } catch (const std::exception& ex) {
            rc = fail(D4G_ERR_RUNTIME, ex.what());
        }
    }
    d4g_batch_destroy(b);
    return rc;
}

int d4g_inflate(const uint8_t* in, size_t len, uint8_t** out, size_t* out_len, size_t* consumed, int32_t* status) {
    enter_ctx(nullptr);
    if (!in || !out || !out_len || !status) return fail(D4G_ERR_ARG, "null argument");
    const uint8_t* ins[1] = {in};
    size_t lens[1] = {len};
    *out = nullptr;
    *out_len = 0;
    d4g_batch* b = d4g_batch_create(1, ins, lens);
    if (!b) return D4G_ERR_RUNTIME;
    int rc;
    {
        D4G_API_LOCK();
        try {
            bind_device();
            engine().init();
            b->impl.parse_probe();
            b->impl.build_blocks(false, false);
            const Batch::PStream& o = b->impl.ps[0];
            if (consumed) *consumed = (size_t)o.consumed;
            if (o.status != 0) {
                *status = D4G_STREAM_PARSE_ERROR;
            } else {
                *status = D4G_STREAM_UNCHANGED;
                *out = (uint8_t*)malloc(o.nU ? (size_t)o.nU : 1);
                *out_len = (size_t)o.nU;
                rt_d2h(*out, b->impl.dU + b->impl.streams[0].uBase, (size_t)o.nU);
            }
            rc = D4G_OK;
        } catch (const std::exception& ex) {
            rc = fail(D4G_ERR_RUNTIME, ex.what());
        }
    }
    d4g_batch_destroy(b);
    return rc;
}

// ---- Zopfli encoder ----
namespace {
// host inputs -> one padded device buffer (16-byte aligned starts, 512 zero bytes after each end)
struct ZfUpload {
    LzScratch own;
    std::vector<const uint8_t*> ptr;
    std::vector<i64> len;
    ZfUpload(size_t n, const uint8_t* const* raw, const size_t* raw_len) {
        i64 off = 0;
        std::vector<i64> at(n);
        for (size_t i = 0; i < n; i++) { at[i] = off; off += (((i64)raw_len[i] + 15) & ~15LL) + 512; }
        uint8_t* d = own.own((uint8_t*)rt_malloc((size_t)off + 1024));
        rt_memset(d, 0, (size_t)off + 1024);
        for (size_t i = 0; i < n; i++) { rt_h2d(d + at[i], raw[i], raw_len[i]); ptr.push_back(d + at[i]); len.push_back((i64)raw_len[i]); }
        rt_sync();
    }
};
}  // namespace

int d4g_zopfli_streams(size_t n, const uint8_t* const* raw, const size_t* raw_len, int iterations, int splitting, int max_blocks,
                       size_t master_block, uint8_t** out, size_t* out_len) {
    enter_ctx(nullptr);
    if (n && (!raw || !raw_len || !out || !out_len)) return fail(D4G_ERR_ARG, "null argument");
    for (size_t i = 0; i < n; i++) { out[i] = nullptr; out_len[i] = 0; }
    if (iterations < 1 || splitting < 0 || splitting > 2 || max_blocks < 0 || master_block > ((size_t)8 << 20)) return fail(D4G_ERR_ARG, "bad zopfli options");
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    try {
        bind_device();
        ZfUpload up(n, raw, raw_len);
        ZfFront zf;
        zf.create(n, up.ptr.data(), up.len.data());
        std::vector<ZfSpec> specs;
        for (size_t i = 0; i < n; i++) {
            if (master_block == 0 && raw_len[i] > ((size_t)8 << 20)) return fail(D4G_ERR_ARG, "inputs above 8 MiB need a master block size");
            specs.push_back({(int32_t)i, iterations, splitting, max_blocks, (long long)master_block});
        }
        zf.encode(specs);
        if (env_int("D4G_DEBUG_ZOPFLI", 0))
            fprintf(stderr, "[zopfli] inputs %zu: table %.1f ms, split %.1f ms, squeeze %.1f ms (%lld blocks, %lld position-iterations), final+emit %.1f ms\n", n,
                    zf.msTable, zf.msSplit, zf.msSqueeze, (long long)zf.squeezeBlocks, (long long)zf.squeezePositions, zf.msEmit);
        for (size_t i = 0; i < n; i++) {
            const size_t nb = (size_t)((zf.outBits[i] + 7) / 8);
            out[i] = (uint8_t*)malloc(nb ? nb : 1);
            if (!out[i]) throw std::runtime_error("out of host memory");
            rt_d2h(out[i], zf.outWords[i], nb);
            out_len[i] = nb;
        }
        return D4G_OK;
    } catch (const std::exception& ex) {
        for (size_t i = 0; i < n; i++) { free(out[i]); out[i] = nullptr; out_len[i] = 0; }
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_debug_zopfli_table(const uint8_t* raw, size_t n, size_t end, uint16_t* len16, uint16_t* dist16, uint16_t* sublen) {
    enter_ctx(nullptr);
    if (!raw || !len16 || !dist16) return fail(D4G_ERR_ARG, "null argument");
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    try {
        bind_device();
        const uint8_t* rp[1] = {raw};
        size_t rl[1] = {n};
        ZfUpload up(1, rp, rl);
        ZfFront zf;
        zf.create(1, up.ptr.data(), up.len.data());
        if (end == 0 || end > n) end = n;
        zf.ensure_tails({{0, (i64)end}});
        const ZfFront::Tail& t = zf.tails.at({0, (i64)end});
        std::vector<uint32_t> table(n * 8 + 8), best(n + 8), pool(zf.pool.cap);
        rt_d2h(table.data(), zf.hIn[0].table, n * 32);
        rt_d2h(best.data(), zf.hIn[0].best, n * 4);
        rt_d2h(pool.data(), zf.pool.words, (size_t)zf.pool.cap * 4);
        if ((i64)end > t.start && t.table != zf.hIn[0].table + t.start * 8) {
            rt_d2h(table.data() + t.start * 8, t.table, (size_t)(end - t.start) * 32);
            rt_d2h(best.data() + t.start, t.best, (size_t)(end - t.start) * 4);
        }
        for (size_t i = 0; i < end; i++) {
            len16[i] = (uint16_t)(best[i] >> 16);
            dist16[i] = (uint16_t)(best[i] & 0xffff);
            if (!sublen) continue;
            int from = 3;
            auto fill = [&](uint32_t w) { for (int l = from; l <= (int)(w >> 16); l++) sublen[i * 259 + l] = (uint16_t)(w & 0xffff); from = (int)(w >> 16) + 1; };
            for (int c = 0; c < 8; c++) {
                const uint32_t w = table[i * 8 + c];
                if (!w) break;
                if (c == 7 && (w & ZF_POOL_LINK)) { for (const uint32_t* q = &pool[w & 0x7fffffffu]; *q; q++) fill(*q); break; }
                fill(w);
            }
        }
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

__global__ void __launch_bounds__(64) k_zf_debug_code_lengths(const uint32_t* freq, int n, int maxbits, uint32_t* out) {
    __shared__ ZfEvalLds E;
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < ZF_NUM_LL; i += 64) { E.llc[i] = i < n ? freq[i] : 0; E.ll[i] = 0; }
    LZ_WAVE_SYNC();
    const int m = zf_sort_leaves(E.llc, n, E.u.pm.big[0].w, E.u.pm.big[0].sym, E.u.pm.big[0].list[1]);
    LZ_WAVE_SYNC();
    {   // the wave-wide builder (what the GPU's block-size evaluation uses for the lit/len and distance trees)
        ZfPmRef r = {E.u.pm.big[0].w, E.u.pm.big[0].sym, E.u.pm.big[0].list[0], E.u.pm.big[0].list[1], &E.u.pm.big[0].bits[0][0], 18, m, E.lvl[0]};
        zf_pm_wave(r, maxbits, E.ll);
    }
    LZ_WAVE_SYNC();
    for (int i = lane; i < n; i += 64) out[i] = E.ll[i];
}

int d4g_debug_zopfli_code_lengths(const uint32_t* freq, int n, int maxbits, uint32_t* lengths) {
    enter_ctx(nullptr);
    if (!freq || !lengths || n < 1 || n > ZF_NUM_LL || maxbits < 1 || maxbits > 15) return fail(D4G_ERR_ARG, "bad argument");
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    try {
        bind_device();
        LzScratch own;
        uint32_t* dF = own.own((uint32_t*)rt_malloc(n * 4));
        uint32_t* dO = own.own((uint32_t*)rt_malloc(n * 4));
        rt_h2d(dF, freq, n * 4);
        RT_LAUNCH(k_zf_debug_code_lengths, 1, 64, dF, n, maxbits, dO);
        rt_d2h(lengths, dO, n * 4);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

void d4g_free(void* p) { free(p); }

}  // extern "C"

namespace {
// compressor list of a recompress mode, in CompressionUtil.getCompressors order (C/CompressionUtil.java:44-78 with the
// flags CMDUtil.java:44-50 derives from the mode)
#define D4G_COMP_JZOPFLI 2   // list entries beyond the two zlib flavours: `strategy` is the option set's index
#define D4G_COMP_CAFE 3
bool mode_specs(int mode, std::vector<LzSpec>& list, std::string& why) {
    list.clear();
    if (mode < D4G_MODE_CHEAP || mode > D4G_MODE_ZOPFLI_VERY_EXTENSIVE) { why = "mode out of range"; return false; }
    const bool extensive = mode >= D4G_MODE_ZOPFLI_EXTENSIVE;      // Strategy.EXTENSIVE (CMDUtil.java:46)
    for (int st : {LZ_DEFAULT, LZ_FILTERED, LZ_HUFFMAN_ONLY}) list.push_back({0, LZ_FLAVOR_ZLIB, st});       // java (JVM) first (:51-58)
    if (mode >= D4G_MODE_ZOPFLI_VERY_EXTENSIVE)                                                              // jzopfli (:60-62)
        for (int o = 0; o < (extensive ? 5 : 1); o++) list.push_back({0, D4G_COMP_JZOPFLI, o});
    if (mode >= D4G_MODE_ZOPFLI)                                                                             // CafeUndZopfli (:64-66)
        for (int o = 0; o < (extensive ? 3 : 1); o++) list.push_back({0, D4G_COMP_CAFE, o});
    for (int st : {LZ_DEFAULT, LZ_FILTERED, LZ_HUFFMAN_ONLY}) list.push_back({0, LZ_FLAVOR_JZLIB, st});      // jzlib last (:68-75)
    return true;
}
// the option set of a Zopfli list entry: MultiJZopfliCompressor.getOptions (C/MultiJZopfliCompressor.java:18-60: split15/first,
// split15/last, split0/first, split0/last, nosplit; libzopfli's 1000000-byte master block) and
// MultiCafeUndZopfliCompressor.getOptions (C/MultiCafeUndZopfliCompressor.java:19-25,33: FIRST, LAST, NONE; 8 MiB master block)
ZfSpec zopfli_options(const LzSpec& e, int input, int iter) {
    ZfSpec z{};
    z.input = input;
    z.iterations = iter < 1 ? 1 : iter;
    if (e.encoder == D4G_COMP_CAFE) {
        z.splitting = e.strategy; z.maxblocks = 15; z.master = 8LL << 20;
    } else {
        static const int split[5] = {ZF_SPLIT_FIRST, ZF_SPLIT_LAST, ZF_SPLIT_FIRST, ZF_SPLIT_LAST, ZF_SPLIT_NONE};
        static const int maxb[5] = {15, 15, 0, 0, 0};
        z.splitting = split[e.strategy]; z.maxblocks = maxb[e.strategy]; z.master = 1000000;
    }
    return z;
}

// CompressionUtil.compress for inputs that live in host or device memory.  Inputs are processed in groups sized to the
// device memory the candidate search needs (every block owns ~1.6 MB of candidate states); the winners' bytes are kept
// in one device buffer.
struct CompressRun {
    std::vector<int> winner;             // index in list order, per input
    std::vector<long long> bits;         // its parsed bit size
    std::vector<size_t> off, len;        // its bytes in dWin
    uint8_t* dWin = nullptr;
    size_t used = 0;
    size_t perInput = 0;
    d4g_stats agg;
    int iter = 20;
    double msZopfli = 0, msZfTable = 0, msZfSplit = 0, msZfSqueeze = 0, msZfEmit = 0;
    int64_t zfBlocks = 0, zfPosIter = 0;
    int64_t outputsOptimised = 0, outputsPruned = 0;
    CompressRun() { memset(&agg, 0, sizeof(agg)); }
    ~CompressRun() { rt_free(dWin); }
};
void add_stats(d4g_stats& a, const d4g_stats& o) {
    a.ms_lz_sort += o.ms_lz_sort; a.ms_lz_parse += o.ms_lz_parse; a.ms_lz_emit += o.ms_lz_emit;
    a.lz_parse_passes += o.lz_parse_passes; a.lz_chunks_rerun += o.lz_chunks_rerun; a.lz_symbols += o.lz_symbols;
    a.ms_parse += o.ms_parse; a.ms_optimise += o.ms_optimise; a.ms_merge += o.ms_merge; a.ms_write += o.ms_write;
    a.ms_state_kernels += o.ms_state_kernels; a.state_launches += o.state_launches;
    a.state_tokens_per_round += o.state_tokens_per_round; a.state_bytes_per_round += o.state_bytes_per_round;
    a.kernel_launches += o.kernel_launches; a.rounds += o.rounds; a.n_blocks += o.n_blocks; a.n_tokens += o.n_tokens;
    a.ms_search_kernels += o.ms_search_kernels; a.ms_parse_kernels += o.ms_parse_kernels;
    a.search_bytes_algorithmic += o.search_bytes_algorithmic;
}
std::unique_ptr<d4g_batch> encode_batch(size_t n, const uint8_t* const* raw, const size_t* len, bool fromDevice, const std::vector<LzSpec>& specs,
                                        bool merge) {
    std::unique_ptr<d4g_batch> e(new d4g_batch());
    e->ctx = rt_ctx();
    e->lz.reset(new LzFront(e->impl));
    e->lz->create(n, raw, len, specs.size(), specs.data(), fromDevice);
    e->lz->run(true, merge);
    return e;
}
// one group of inputs [i0, i1)
void compress_group(CompressRun& R, size_t i0, size_t i1, const uint8_t* const* raw, const size_t* len, bool fromDevice,
                    const std::vector<LzSpec>& list, bool merge, const std::function<void()>& afterStart) {
    const size_t n = i1 - i0;
    // stage 1: every compressor output that can hold back-references
    std::vector<int> lzIdx, hIdx;   // list positions
    std::vector<int> zIdx;          // the Zopfli entries
    for (size_t k = 0; k < list.size(); k++) {
        if (list[k].encoder >= D4G_COMP_JZOPFLI) zIdx.push_back((int)k);
        else (list[k].strategy == LZ_HUFFMAN_ONLY ? hIdx : lzIdx).push_back((int)k);
    }
    std::vector<LzSpec> specs;
    for (size_t i = 0; i < n; i++)
        for (int k : lzIdx) { LzSpec s = list[k]; s.input = (int32_t)i; specs.push_back(s); }
    std::unique_ptr<d4g_batch> e1(new d4g_batch());
    e1->ctx = rt_ctx();
    e1->lz.reset(new LzFront(e1->impl));
    e1->lz->create(n, raw + i0, len + i0, specs.size(), specs.data(), fromDevice);
    // stage 3 starts here, on its own host thread (its own HIP streams): the Zopfli compressors' outputs, encoded on the device
    // and parsed + optimised like any other stream.  A squeeze keeps a handful of waves busy for a long time; the zlib-family
    // stages below fill the rest of the device meanwhile.  It only reads the inputs (e1's dU, stable from create on).
    struct ZStage {
        std::unique_ptr<d4g_batch> e3;
        std::exception_ptr err;
        double ms = 0, msTable = 0, msSplit = 0, msSqueeze = 0, msEmit = 0;
        int64_t blocks = 0, posIter = 0, outputs = 0;
    } Z;
    const int parentCtx = rt_ctx();
    auto zopfli_stage = [&]() {
        try {
            rt_ctx() = parentCtx;   // (a new host thread starts on context 0)
#ifndef D4G_HOSTSIM
            rt_low_priority_thread() = true;      // this thread's streams carry kernels that run for minutes (d4g_rt.h)
#endif
            bind_device();
            double tz = now_ms();
            std::vector<const uint8_t*> dp(n);
            std::vector<i64> dl(n);
            for (size_t i = 0; i < n; i++) { dp[i] = e1->impl.dU + e1->lz->rawU[i]; dl[i] = e1->lz->rawLen[i]; }
            ZfFront zf;
            zf.create(n, dp.data(), dl.data());
            std::vector<ZfSpec> zs;
            for (size_t i = 0; i < n; i++)
                for (int k : zIdx) zs.push_back(zopfli_options(list[k], (int)i, R.iter));
            zf.encode(zs);
            Z.msTable = zf.msTable; Z.msSplit = zf.msSplit; Z.msSqueeze = zf.msSqueeze; Z.msEmit = zf.msEmit;
            Z.blocks = zf.squeezeBlocks; Z.posIter = zf.squeezePositions; Z.outputs = (int64_t)zs.size();
            std::vector<const uint8_t*> sp(zs.size());
            std::vector<size_t> sl(zs.size());
            for (size_t q = 0; q < zs.size(); q++) { sp[q] = (const uint8_t*)zf.outWords[q]; sl[q] = (size_t)((zf.outBits[q] + 7) / 8); }
            Z.e3.reset(new d4g_batch());
            Z.e3->ctx = rt_ctx();
            Z.e3->impl.create(zs.size(), sp.data(), sl.data(), true);
            Z.ms = now_ms() - tz;
            Z.e3->impl.run(merge);
        } catch (...) {
            Z.err = std::current_exception();
        }
    };
    std::thread zthread;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{zthread};   // also when a stage below throws
#ifndef D4G_HOSTSIM
    if (!zIdx.empty()) zthread = std::thread(zopfli_stage);
#endif
    const bool dbg = env_int("D4G_DEBUG_ZOPFLI", 0) > 0;
    const double tg0 = now_ms();
    if (afterStart) afterStart();      // the caller's own work that only had to wait for the Zopfli stage to be under way
    if (dbg) fprintf(stderr, "[group] +%.1f s: caller's work done\n", (now_ms() - tg0) / 1000);
    e1->lz->run(true, merge);
    if (dbg) fprintf(stderr, "[group] +%.1f s: zlib-family stage 1 done (front %.1f s, search %.1f s, merge %.1f s)\n", (now_ms() - tg0) / 1000,
                     e1->impl.stats.ms_parse / 1000, e1->impl.stats.ms_optimise / 1000, e1->impl.stats.ms_merge / 1000);
    add_stats(R.agg, e1->impl.stats);
    R.outputsOptimised += (int64_t)specs.size();
    struct Best { long long bits = 0; int listIdx = -1; const Batch* owner = nullptr; int stream = -1; };
    std::vector<Best> best(n);
    auto offer = [&](size_t i, int listIdx, const Batch* owner, int stream) {
        const HStream& s = owner->streams[stream];
        long long bits = s.sizeBitsIn - s.saved;              // Deft.getSizeBitsFallback of the optimised output
        Best& b = best[i];
        if (b.listIdx < 0 || bits < b.bits || (bits == b.bits && listIdx < b.listIdx)) { b.bits = bits; b.listIdx = listIdx; b.owner = owner; b.stream = stream; }
    };
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < lzIdx.size(); k++) offer(i, lzIdx[k], &e1->impl, (int)(i * lzIdx.size() + k));
    // stage 2: the HUFFMAN_ONLY outputs, unless their entropy bound already loses
    std::unique_ptr<d4g_batch> e2;
    if (!hIdx.empty()) {
        std::vector<LzBoundJob> jobs;
        for (size_t i = 0; i < n; i++) {
            const uint8_t* d = e1->impl.dU + e1->lz->rawU[i];
            for (long long p = 0; p < e1->lz->rawLen[i]; p += LZ_SYMS_PER_BLOCK)
                jobs.push_back({d + p, (int32_t)std::min<long long>(LZ_SYMS_PER_BLOCK, e1->lz->rawLen[i] - p), (int32_t)i});
        }
        std::vector<double> lb(n, 0.0);
        if (!jobs.empty()) {
            LzBoundJob* dJ = (LzBoundJob*)rt_malloc(jobs.size() * sizeof(LzBoundJob));
            double* dLb = (double*)rt_malloc(n * 8 + 16);
            rt_h2d(dJ, jobs.data(), jobs.size() * sizeof(LzBoundJob));
            rt_memset(dLb, 0, n * 8);
            RT_LAUNCH(k_lz_entropy_bound, jobs.size(), 256, dJ, dLb);
            rt_d2h(lb.data(), dLb, n * 8);
            rt_free(dJ); rt_free(dLb);
        }
        std::vector<size_t> need;
        for (size_t i = 0; i < n; i++) {
            double bound = lb[i] * (1.0 - 1e-9) - 1.0;       // rounding of the sum can only have raised it by less than this
            if (bound > (double)best[i].bits) R.outputsPruned += (int64_t)hIdx.size();
            else need.push_back(i);
        }
        if (!need.empty()) {
            std::vector<const uint8_t*> rp(need.size());
            std::vector<size_t> rl(need.size());
            std::vector<LzSpec> sp2;
            for (size_t q = 0; q < need.size(); q++) {
                rp[q] = e1->impl.dU + e1->lz->rawU[need[q]];
                rl[q] = (size_t)e1->lz->rawLen[need[q]];
                for (int k : hIdx) { LzSpec s = list[k]; s.input = (int32_t)q; sp2.push_back(s); }
            }
            e2 = encode_batch(need.size(), rp.data(), rl.data(), true, sp2, merge);
            add_stats(R.agg, e2->impl.stats);
            R.outputsOptimised += (int64_t)sp2.size();
            for (size_t q = 0; q < need.size(); q++)
                for (size_t k = 0; k < hIdx.size(); k++) offer(need[q], hIdx[k], &e2->impl, (int)(q * hIdx.size() + k));
        }
    }
    // stage 3's results
    if (!zIdx.empty()) {
#ifdef D4G_HOSTSIM
        zopfli_stage();            // the emulator runs one kernel at a time
#else
        if (dbg) fprintf(stderr, "[group] +%.1f s: stage 2 done, waiting for the Zopfli stage\n", (now_ms() - tg0) / 1000);
        zthread.join();
        if (dbg) fprintf(stderr, "[group] +%.1f s: Zopfli stage joined (encode %.1f s, its outputs' search %.1f s + merge %.1f s)\n", (now_ms() - tg0) / 1000,
                         Z.ms / 1000, Z.e3 ? Z.e3->impl.stats.ms_optimise / 1000 : 0.0, Z.e3 ? Z.e3->impl.stats.ms_merge / 1000 : 0.0);
#endif
        if (Z.err) std::rethrow_exception(Z.err);
        R.msZfTable += Z.msTable; R.msZfSplit += Z.msSplit; R.msZfSqueeze += Z.msSqueeze; R.msZfEmit += Z.msEmit;
        R.zfBlocks += Z.blocks; R.zfPosIter += Z.posIter;
        R.msZopfli += Z.ms;
        add_stats(R.agg, Z.e3->impl.stats);
        R.outputsOptimised += Z.outputs;
        for (size_t i = 0; i < n; i++)
            for (size_t k = 0; k < zIdx.size(); k++) {
                const int st = (int)(i * zIdx.size() + k);
                if (Z.e3->impl.streams[st].status != 0) throw std::runtime_error("zopfli output does not parse");
                offer(i, zIdx[k], &Z.e3->impl, st);
            }
    }
    for (size_t i = 0; i < n; i++) {
        const Best& b = best[i];
        const HStream& w = b.owner->streams[b.stream];
        size_t nb = (size_t)((w.outBits + 7) / 8);
        R.winner[i0 + i] = b.listIdx;
        R.bits[i0 + i] = b.bits;
        R.off[i0 + i] = R.used;
        R.len[i0 + i] = nb;
        rt_d2d(R.dWin + R.used, (const uint8_t*)(b.owner->dOut + w.outWordBase), nb);
        R.used += (nb + 15) & ~(size_t)15;
    }
    rt_sync();
}
void compress_run(CompressRun& R, size_t n, const uint8_t* const* raw, const size_t* len, bool fromDevice, int mode, int iter, bool merge,
                  const std::function<void()>& between = nullptr) {
    R.iter = iter;
    int betweenState = 0;      // 0 not run, 1 running, 2 done
    std::function<void()> once = [&]() { if (between && betweenState == 0) { betweenState = 1; between(); betweenState = 2; } };
    std::vector<LzSpec> list;
    std::string why;
    if (!mode_specs(mode, list, why)) throw std::runtime_error(why);
    R.perInput = list.size();
    R.winner.assign(n, -1); R.bits.assign(n, 0); R.off.assign(n, 0); R.len.assign(n, 0);
    size_t cap = 64;
    for (size_t i = 0; i < n; i++) cap += len[i] + len[i] / 512 + 96;   // a deflate stream never exceeds its input by more than this
    R.dWin = (uint8_t*)rt_malloc(cap);
    // group size: the candidate search keeps ~1.6 MB of states per block; estimate >= 3 bytes per symbol and let an
    // out-of-memory failure halve the group
    const long long budget = env_int("D4G_GROUP_BLOCKS", 12000);
    size_t i0 = 0;
    long long shrink = 1;
    while (i0 < n) {
        size_t i1 = i0;
        long long est = 0;
        while (i1 < n) {
            long long e = 4 * ((long long)len[i1] / (3LL * LZ_SYMS_PER_BLOCK) + 2);
            if (i1 > i0 && (est + e) * shrink > budget) break;
            est += e;
            i1++;
        }
        try {
            compress_group(R, i0, i1, raw, len, fromDevice, list, merge, once);
            i0 = i1;
        } catch (const std::runtime_error& ex) {
            if (betweenState != 1 && strstr(ex.what(), "hipMalloc") && i1 - i0 > 1) { shrink *= 2; continue; }   // group too large for the device: retry smaller
            throw;
        }
    }
    once();
}
uint8_t* copy_stream_out(Batch& b, size_t i, size_t* len) {
    const HStream& s = b.streams[i];
    size_t nb = (size_t)((s.outBits + 7) / 8);
    uint8_t* p = (uint8_t*)malloc(nb ? nb : 1);
    if (!p) throw std::runtime_error("out of host memory");
    rt_d2h(p, (const uint8_t*)(b.dOut + s.outWordBase), nb);
    *len = nb;
    return p;
}
}  // namespace

extern "C" {

int d4g_compress(size_t n, const uint8_t* const* raw, const size_t* raw_len, int mode, int iter, int merge_blocks, uint8_t** out,
                 size_t* out_len, int32_t* winner) {
    enter_ctx(nullptr);
    if (n && (!raw || !raw_len || !out || !out_len)) return fail(D4G_ERR_ARG, "null argument");
    for (size_t i = 0; i < n; i++) { out[i] = nullptr; out_len[i] = 0; if (winner) winner[i] = -1; }
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    try {
        bind_device();
        std::vector<LzSpec> probe;
        std::string why;
        if (!mode_specs(mode, probe, why)) return fail(D4G_ERR_ARG, why);
        CompressRun R;
        compress_run(R, n, raw, raw_len, false, mode, iter, merge_blocks != 0);
        for (size_t i = 0; i < n; i++) {
            out[i] = (uint8_t*)malloc(R.len[i] ? R.len[i] : 1);
            if (!out[i]) throw std::runtime_error("out of host memory");
            rt_d2h(out[i], R.dWin + R.off[i], R.len[i]);
            out_len[i] = R.len[i];
            if (winner) winner[i] = (int32_t)R.winner[i];
        }
        return D4G_OK;
    } catch (const std::exception& ex) {
        for (size_t i = 0; i < n; i++) { free(out[i]); out[i] = nullptr; out_len[i] = 0; }
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

// CMDUtil.optimise's per-stream loop on a batch made by d4g_batch_create (locked by the caller)
static void run_recompress_locked(d4g_batch* b, int mode, int iter, bool merge) {
    Batch& A = b->impl;
    const size_t n = A.streams.size();
    double t0 = now_ms();
    // container.optimise(mergeBlocks) — CMDUtil.java:70.  In a Zopfli mode only the parse happens here: the search of the originals
    // runs once the Zopfli stage of the recompression (its own host thread, a few long-running waves) is under way.
    const bool deferSearch = mode >= D4G_MODE_ZOPFLI;
    A.run_parse(merge);
    if (!deferSearch) A.run_rest(merge);
    b->graft.assign(n, 0);
    b->recompSaved.assign(n, 0);
    b->reoptIndex.assign(n, -1);
    std::vector<size_t> ok;
    for (size_t i = 0; i < n; i++)
        if (A.streams[i].status == 0) ok.push_back(i);
    if (mode == D4G_MODE_NONE || ok.empty()) { if (deferSearch) A.run_rest(merge); return; }
    const double tc0 = now_ms();
    // stream.getUncompressedData() -> compUtil.compress(uncompressed, true) — :83-84; the decoded bytes stay in HBM
    std::vector<const uint8_t*> rp(ok.size());
    std::vector<size_t> rl(ok.size());
    for (size_t k = 0; k < ok.size(); k++) { rp[k] = A.dU + A.streams[ok[k]].uBase; rl[k] = (size_t)A.streams[ok[k]].nU; }
    CompressRun R;
    compress_run(R, ok.size(), rp.data(), rl.data(), true, mode, iter, merge, deferSearch ? std::function<void()>([&]() { A.run_rest(merge); }) : std::function<void()>());
    double t1 = now_ms();
    // new DeflateStream().parse(recompressed); recompStream.optimise(mergeBlocks) — :85-89
    std::vector<const uint8_t*> wp(ok.size());
    std::vector<size_t> wl(ok.size());
    for (size_t k = 0; k < ok.size(); k++) { wp[k] = R.dWin + R.off[k]; wl[k] = R.len[k]; }
    b->reopt.reset(new Batch());
    b->reopt->create(ok.size(), wp.data(), wl.data(), true);
    b->reopt->run(merge);
    for (size_t k = 0; k < ok.size(); k++) {
        const HStream& a = A.streams[ok[k]];
        const HStream& c2 = b->reopt->streams[k];
        b->reoptIndex[ok[k]] = (int)k;
        if (c2.status != 0) continue;                  // recompStream.parse failed: the original stays (:88)
        long long recompSize = c2.sizeBitsIn - c2.saved, originalSize = a.sizeBitsIn - a.saved;
        if (recompSize < originalSize) {               // :94-98
            b->graft[ok[k]] = 1;
            b->recompSaved[ok[k]] = originalSize - recompSize;
        }
    }
    const d4g_stats& es = R.agg;
    A.stats.ms_lz_sort = es.ms_lz_sort; A.stats.ms_lz_parse = es.ms_lz_parse; A.stats.ms_lz_emit = es.ms_lz_emit;
    A.stats.lz_parse_passes = es.lz_parse_passes; A.stats.lz_chunks_rerun = es.lz_chunks_rerun; A.stats.lz_symbols = es.lz_symbols;
    A.stats.ms_recompress_encode = t1 - tc0 - (deferSearch ? A.stats.ms_total - A.stats.ms_parse : 0.0);
    A.stats.ms_recompress_encode_front = es.ms_parse;
    A.stats.ms_recompress_encode_search = es.ms_optimise + es.ms_merge;
    A.stats.ms_recompress_reoptimise = now_ms() - t1;
    A.stats.recompress_outputs = R.outputsOptimised;
    A.stats.recompress_outputs_pruned = R.outputsPruned;
    A.stats.ms_zopfli_table = R.msZfTable; A.stats.ms_zopfli_split = R.msZfSplit; A.stats.ms_zopfli_squeeze = R.msZfSqueeze; A.stats.ms_zopfli_emit = R.msZfEmit;
    A.stats.zopfli_blocks = R.zfBlocks; A.stats.zopfli_position_iterations = R.zfPosIter;
    const d4g_stats* chained[2] = {&es, &b->reopt->stats};
    for (const d4g_stats* o : chained) {   // the same kernels ran in the chained batches: one set of counters
        A.stats.ms_state_kernels += o->ms_state_kernels; A.stats.state_launches += o->state_launches;
        A.stats.state_tokens_per_round += o->state_tokens_per_round; A.stats.state_bytes_per_round += o->state_bytes_per_round;
        A.stats.kernel_launches += o->kernel_launches; A.stats.rounds += o->rounds;
        A.stats.ms_search_kernels += o->ms_search_kernels; A.stats.ms_parse_kernels += o->ms_parse_kernels;
    }
    A.stats.search_bytes_algorithmic += es.search_bytes_algorithmic + b->reopt->stats.search_bytes_algorithmic;
}

int d4g_batch_run_recompress(d4g_batch* b, int mode, int iter, int merge_blocks) {
    enter_ctx(b);
    D4G_API_LOCK();
    if (!ready()) return fail(D4G_ERR_NODEVICE, "d4g_init has not succeeded");
    if (!b || b->lz) return fail(D4G_ERR_ARG, "not a batch of deflate streams");
    try {
        bind_device();
        std::vector<LzSpec> probe;
        std::string why;
        if (mode != D4G_MODE_NONE && !mode_specs(mode, probe, why)) return fail(D4G_ERR_ARG, why);
        run_recompress_locked(b, mode, iter, merge_blocks != 0);
        return D4G_OK;
    } catch (const std::exception& ex) {
        return fail(D4G_ERR_RUNTIME, ex.what());
    }
}

int d4g_batch_recompress_result(d4g_batch* b, size_t i, int32_t* grafted, int64_t* recompress_saved) {
    enter_ctx(b);
    if (!b || i >= b->impl.streams.size()) return fail(D4G_ERR_ARG, "bad stream index");
    if (grafted) *grafted = i < b->graft.size() ? b->graft[i] : 0;
    if (recompress_saved) *recompress_saved = i < b->recompSaved.size() ? b->recompSaved[i] : 0;
    return D4G_OK;
}

int d4g_recompress_streams(size_t n, const uint8_t* const* in, const size_t* in_len, int mode, int iter, int merge_blocks,
                           uint8_t** out, size_t* out_len, int64_t* saved_bits, int64_t* recompress_saved, int32_t* status) {
    enter_ctx(nullptr);
    if (n && (!in || !in_len || !out || !out_len || !status)) return fail(D4G_ERR_ARG, "null argument");
    for (size_t i = 0; i < n; i++) {
        out[i] = nullptr; out_len[i] = 0; status[i] = D4G_STREAM_UNCHANGED;
        if (saved_bits) saved_bits[i] = 0;
        if (recompress_saved) recompress_saved[i] = 0;
    }
    d4g_batch* b = d4g_batch_create(n, in, in_len);
    if (!b) return D4G_ERR_RUNTIME;
    int rc = d4g_batch_run_recompress(b, mode, iter, merge_blocks);
    for (size_t i = 0; i < n && rc == D4G_OK; i++) {
        int64_t sv = 0, rs = 0;
        size_t ol = 0;
        int32_t st = D4G_STREAM_UNCHANGED;
        d4g_batch_stream_result(b, i, &st, &sv, &ol, nullptr, nullptr);
        d4g_batch_recompress_result(b, i, nullptr, &rs);
        if (st == D4G_STREAM_CHANGED) {
            out[i] = (uint8_t*)malloc(ol ? ol : 1);
            if (!out[i]) { rc = fail(D4G_ERR_RUNTIME, "out of host memory"); break; }
            rc = d4g_batch_copy_output(b, i, out[i], ol);
            if (rc != D4G_OK) break;
            out_len[i] = ol;
        }
        status[i] = st;
        if (saved_bits) saved_bits[i] = sv;
        if (recompress_saved) recompress_saved[i] = rs;
    }
    if (rc != D4G_OK)
        for (size_t i = 0; i < n; i++) {
            free(out[i]); out[i] = nullptr; out_len[i] = 0; status[i] = D4G_STREAM_UNCHANGED;
            if (saved_bits) saved_bits[i] = 0;
            if (recompress_saved) recompress_saved[i] = 0;
        }
    std::string keep = g_err;
    d4g_batch_destroy(b);
    g_err = keep;
    return rc;
}

}  // extern "C"

extern "C" {

#ifdef D4G_HOSTSIM
// emulator builds only: the closed-form pack summary against the reference-shaped loop, every flag set and run length
long long d4g_test_pack_kinds(void) {
    long long bad = 0;
    for (int flags = 0; flags < 256; flags++)
        for (int v = 0; v <= 7; v += 7)
            for (int r = 1; r <= 420; r++) {
                int a[19 * 160] = {0}, b[19 * 160] = {0};
                d4g_pack_run(v, r, flags, [&](int sym, int run, int) { a[sym * 160 + run]++; });
                d4g_pack_kinds(v, r, flags, [&](int sym, int run, int, int cnt) { b[sym * 160 + run] += cnt; }, [&](int cnt) { b[v * 160] += cnt; });
                for (int i = 0; i < 19 * 160; i++) bad += a[i] != b[i];
            }
    return bad;
}
#endif

// dev tool: the fused executor's accounting (collected while D4G_FUSED_STATS is set; see k_search_fused); read and cleared
int d4g_debug_fused_stats(long long* out64) {
    enter_ctx(nullptr);
    D4G_API_LOCK();
    d4g::engine().init();
    rt_sync_all();
    rt_d2h(out64, d4g::engine().dOpStats, 64 * 8);
    rt_memset(d4g::engine().dOpStats, 0, 64 * 8);
    rt_sync();
    return 0;
}

#ifdef D4G_PROFILE_OPS
// profiling builds only (scripts/build_profile_lib.sh): cycles and counts per op kind
int d4g_debug_set_experiment(long long mode) {
    enter_ctx(nullptr);
    D4G_API_LOCK();
    rt_h2d(engine().dOpStats + 63, &mode, 8);
    rt_sync();
    return 0;
}
int d4g_debug_opstats(long long* out64) {
    enter_ctx(nullptr);
    D4G_API_LOCK();
    rt_d2h(out64, engine().dOpStats, 64 * 8);
    (void)hipMemcpyFromSymbol(out64 + 56, HIP_SYMBOL(d4g_dbg_counters), 7 * 8);
    (void)hipMemcpyFromSymbol(out64 + 28, HIP_SYMBOL(d4g_dbg_hdr), 4 * 8);
    (void)hipMemcpyFromSymbol(out64 + 16, HIP_SYMBOL(d4g_dbg_tree), 3 * 8);
    (void)hipMemcpyFromSymbol(out64 + 34, HIP_SYMBOL(d4g_dbg_pass), 4 * 8);
    return 0;
}
#endif

}  // extern "C"
