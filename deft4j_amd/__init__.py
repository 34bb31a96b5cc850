"""deft4j_amd — host-side mirror of deft4j's optimiser API over libdeft4g.so (HIP, gfx950).

The Java reference (NeRdTheNed/deft4j) exposes
    Deft.optimiseDeflateStream(byte[], boolean)            B/Deft.java:21-34
    Deft.getSizeBitsFallback(byte[])                       B/Deft.java:48-54
    DeflateStream.parse / optimise / getSizeBits / getUncompressedData / asBytes
                                                           B/deflate/DeflateStream.java:68-182,496,652
    DeflateFilesContainer.optimise(List<DeflateStream>, boolean)   K/DeflateFilesContainer.java:18-43
(B/ = deft4j-base/src/main/java/com/github/NeRdTheNed/deft4j/, K/ = deft4j-container/...).
This module keeps those names, argument meanings and the "fall back to the original bytes"
error behaviour, and forwards every call to the C ABI declared in include/deft4g.h.  There
is no CPU path here: importing works anywhere, but any call raises RuntimeError unless the
HIP library is built and a GPU is present.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdeft4g.so")

_lib = None
_ready = False


class d4g_stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("ms_upload", "ms_parse", "ms_optimise", "ms_merge", "ms_write", "ms_total")] + \
               [(n, ctypes.c_int64) for n in ("n_streams", "n_blocks", "n_tokens", "bytes_in", "bytes_decoded", "bytes_out",
                                              "rounds", "kernel_launches", "search_bytes_algorithmic")] + \
               [("ms_search_kernels", ctypes.c_double), ("ms_parse_kernels", ctypes.c_double)] + \
               [(n, ctypes.c_int64) for n in ("scan_candidates", "scan_confirmed", "exact_probes", "jump_rounds")] + \
               [("ms_state_kernels", ctypes.c_double), ("state_launches", ctypes.c_int64), ("state_tokens_per_round", ctypes.c_int64),
                ("state_bytes_per_round", ctypes.c_int64), ("search_lanes", ctypes.c_int64), ("ms_checksum_kernels", ctypes.c_double),
                ("ms_lz_sort", ctypes.c_double), ("ms_lz_parse", ctypes.c_double), ("ms_lz_emit", ctypes.c_double),
                ("lz_parse_passes", ctypes.c_int64), ("lz_chunks_rerun", ctypes.c_int64), ("lz_symbols", ctypes.c_int64),
                ("ms_recompress_encode", ctypes.c_double), ("ms_recompress_encode_front", ctypes.c_double),
                ("ms_recompress_encode_search", ctypes.c_double), ("ms_recompress_reoptimise", ctypes.c_double),
                ("recompress_outputs", ctypes.c_int64), ("recompress_outputs_pruned", ctypes.c_int64),
                ("ms_zopfli_table", ctypes.c_double), ("ms_zopfli_split", ctypes.c_double), ("ms_zopfli_squeeze", ctypes.c_double),
                ("ms_zopfli_emit", ctypes.c_double), ("zopfli_blocks", ctypes.c_int64), ("zopfli_position_iterations", ctypes.c_int64),
                ("rounds_fused", ctypes.c_int64), ("fused_fallbacks", ctypes.c_int64), ("persist_fallbacks", ctypes.c_int64), ("rounds_cluster", ctypes.c_int64)]


class d4g_encoder_spec(ctypes.Structure):
    _fields_ = [("input", ctypes.c_int32), ("encoder", ctypes.c_int32), ("strategy", ctypes.c_int32)]


ENC_JVM, ENC_JZLIB = 0, 1                                   # D4G_ENC_*: JavaCompressor / JZLibCompressor
STRATEGY_DEFAULT, STRATEGY_FILTERED, STRATEGY_HUFFMAN_ONLY = 0, 1, 2

EXPORTS = ["d4g_init", "d4g_shutdown", "d4g_last_error", "d4g_batch_create", "d4g_batch_run", "d4g_batch_stream_result",
           "d4g_batch_copy_output", "d4g_batch_copy_decoded", "d4g_batch_checksums", "d4g_batch_parse", "d4g_batch_stats", "d4g_batch_destroy", "d4g_optimise_streams",
           "d4g_size_bits_fallback", "d4g_inflate", "d4g_free", "d4g_batch_create_encode", "d4g_batch_run_encode", "d4g_deflate_streams",
           "d4g_compress", "d4g_recompress_streams", "d4g_batch_run_recompress", "d4g_batch_recompress_result", "d4g_zopfli_streams",
           "d4g_debug_zopfli_table", "d4g_debug_zopfli_code_lengths", "d4g_init_devices", "d4g_device_count", "d4g_set_device",
           "d4g_batch_create_on", "d4g_optimise_streams_sharded"]


def load_library(path=None):
    """Load libdeft4g.so and declare the prototypes of include/deft4g.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("D4G_LIB") or LIB_PATH   # D4G_LIB: development builds of the same library
    if not os.path.exists(p):
        raise RuntimeError("libdeft4g.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback" % p)
    L = ctypes.CDLL(p)
    L.d4g_init.restype = ctypes.c_int
    L.d4g_init.argtypes = [ctypes.c_int]
    L.d4g_shutdown.restype = None
    L.d4g_last_error.restype = ctypes.c_char_p
    L.d4g_batch_create.restype = ctypes.c_void_p
    L.d4g_batch_create.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t)]
    L.d4g_batch_run.restype = ctypes.c_int
    L.d4g_batch_run.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.d4g_batch_stream_result.restype = ctypes.c_int
    L.d4g_batch_stream_result.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_size_t),
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int64)]
    L.d4g_batch_copy_output.restype = ctypes.c_int
    L.d4g_batch_copy_output.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.d4g_batch_copy_decoded.restype = ctypes.c_int
    L.d4g_batch_copy_decoded.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t)]
    L.d4g_batch_checksums.restype = ctypes.c_int
    L.d4g_batch_checksums.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_int64)]
    L.d4g_batch_parse.restype = ctypes.c_int
    L.d4g_batch_parse.argtypes = [ctypes.c_void_p]
    L.d4g_batch_stats.restype = ctypes.c_int
    L.d4g_batch_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(d4g_stats)]
    L.d4g_batch_destroy.restype = None
    L.d4g_batch_destroy.argtypes = [ctypes.c_void_p]
    L.d4g_free.restype = None
    L.d4g_free.argtypes = [ctypes.c_void_p]
    L.d4g_optimise_streams.restype = ctypes.c_int
    L.d4g_optimise_streams.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]
    L.d4g_size_bits_fallback.restype = ctypes.c_int
    L.d4g_size_bits_fallback.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int64)]
    L.d4g_inflate.restype = ctypes.c_int
    L.d4g_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                              ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                              ctypes.POINTER(ctypes.c_int32)]
    L.d4g_batch_create_encode.restype = ctypes.c_void_p
    L.d4g_batch_create_encode.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t,
                                          ctypes.POINTER(d4g_encoder_spec)]
    L.d4g_batch_run_encode.restype = ctypes.c_int
    L.d4g_batch_run_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.d4g_deflate_streams.restype = ctypes.c_int
    L.d4g_deflate_streams.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.d4g_zopfli_streams.restype = ctypes.c_int
    L.d4g_zopfli_streams.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.d4g_debug_zopfli_table.restype = ctypes.c_int
    L.d4g_debug_zopfli_table.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.d4g_debug_zopfli_code_lengths.restype = ctypes.c_int
    L.d4g_debug_zopfli_code_lengths.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    L.d4g_compress.restype = ctypes.c_int
    L.d4g_compress.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int32)]
    L.d4g_batch_run_recompress.restype = ctypes.c_int
    L.d4g_batch_run_recompress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.d4g_batch_recompress_result.restype = ctypes.c_int
    L.d4g_batch_recompress_result.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    L.d4g_recompress_streams.restype = ctypes.c_int
    L.d4g_recompress_streams.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]
    L.d4g_init_devices.restype = ctypes.c_int
    L.d4g_init_devices.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.d4g_device_count.restype = ctypes.c_int
    L.d4g_set_device.restype = ctypes.c_int
    L.d4g_set_device.argtypes = [ctypes.c_int]
    L.d4g_batch_create_on.restype = ctypes.c_void_p
    L.d4g_batch_create_on.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t)]
    L.d4g_optimise_streams_sharded.restype = ctypes.c_int
    L.d4g_optimise_streams_sharded.argtypes = L.d4g_optimise_streams.argtypes
    if path is None:
        _lib = L
    return L


def init(device_index=None, lib=None):
    """d4g_init: one process per GPU.  Defaults to LOCAL_RANK (torch.distributed launch) or 0."""
    global _ready
    L = lib or load_library()
    if device_index is None:
        device_index = int(os.environ.get("LOCAL_RANK", "0"))
    rc = L.d4g_init(device_index)
    if rc != 0:
        raise RuntimeError("d4g_init(%d) failed: %s" % (device_index, L.d4g_last_error().decode()))
    if lib is None:
        _ready = True
    return L


def init_devices(devices, lib=None):
    """d4g_init_devices: one process driving several GPUs — context k = devices[k] (a device may appear twice)."""
    global _ready
    L = lib or load_library()
    arr = (ctypes.c_int * len(devices))(*devices)
    rc = L.d4g_init_devices(len(devices), arr)
    if rc != 0:
        raise RuntimeError("d4g_init_devices(%r) failed: %s" % (list(devices), L.d4g_last_error().decode()))
    if lib is None:
        _ready = True
    return L


def optimise_streams_sharded(streams, merge_blocks=True, lib=None):
    """d4g_optimise_streams_sharded: the list over every initialised context.  -> (outputs, saved_bits, status) with
    outputs[i] = None when stream i keeps its original bytes."""
    L = lib or _need()
    n = len(streams)
    keep = [bytes(s) for s in streams]
    arr = (ctypes.c_char_p * n)(*keep)
    lens = (ctypes.c_size_t * n)(*[len(s) for s in keep])
    out = (ctypes.c_void_p * n)()
    olen = (ctypes.c_size_t * n)()
    saved = (ctypes.c_int64 * n)()
    status = (ctypes.c_int32 * n)()
    rc = L.d4g_optimise_streams_sharded(n, arr, lens, 1 if merge_blocks else 0, out, olen, saved, status)
    if rc != 0:
        raise RuntimeError("d4g_optimise_streams_sharded: " + L.d4g_last_error().decode())
    res = []
    for i in range(n):
        if out[i]:
            res.append(ctypes.string_at(out[i], olen[i]))
            L.d4g_free(out[i])
        else:
            res.append(None)
    return res, list(saved), list(status)


def _need():
    if not _ready:
        init()
    return _lib


class Batch:
    """A list of independent raw DEFLATE streams resident in HBM (d4g_batch_*)."""

    def __init__(self, streams, lib=None, context=None):
        self.L = lib or _need()
        self.n = len(streams)
        self._keep = [bytes(s) for s in streams]
        arr = (ctypes.c_char_p * self.n)(*self._keep)
        lens = (ctypes.c_size_t * self.n)(*[len(s) for s in self._keep])
        if context is None:
            self.h = self.L.d4g_batch_create(self.n, arr, lens)
        else:   # d4g_batch_create_on: the batch lives on that context (device) whichever thread calls with it
            self.h = self.L.d4g_batch_create_on(context, self.n, arr, lens)
        if not self.h:
            raise RuntimeError("d4g_batch_create: " + self.L.d4g_last_error().decode())

    def run(self, merge_blocks=True):
        rc = self.L.d4g_batch_run(self.h, 1 if merge_blocks else 0)
        if rc != 0:
            raise RuntimeError("d4g_batch_run: " + self.L.d4g_last_error().decode())
        return self

    def run_recompress(self, mode=1, merge_blocks=True, iter=20):   # noqa: A002
        """CMDUtil.optimise's per-stream loop on the resident streams (d4g_batch_run_recompress)."""
        rc = self.L.d4g_batch_run_recompress(self.h, mode, iter, 1 if merge_blocks else 0)
        if rc != 0:
            raise RuntimeError("d4g_batch_run_recompress: " + self.L.d4g_last_error().decode())
        return self

    def recompress_result(self, i):
        """-> (grafted, recompress_saved_bits)"""
        g = ctypes.c_int32()
        r = ctypes.c_int64()
        self.L.d4g_batch_recompress_result(self.h, i, ctypes.byref(g), ctypes.byref(r))
        return bool(g.value), r.value

    def result(self, i):
        """-> dict(status, saved_bits, out_len, consumed, size_bits_in)"""
        st = ctypes.c_int32()
        sv = ctypes.c_int64()
        ol = ctypes.c_size_t()
        co = ctypes.c_size_t()
        sb = ctypes.c_int64()
        rc = self.L.d4g_batch_stream_result(self.h, i, ctypes.byref(st), ctypes.byref(sv), ctypes.byref(ol),
                                            ctypes.byref(co), ctypes.byref(sb))
        if rc != 0:
            raise RuntimeError(self.L.d4g_last_error().decode())
        return dict(status=st.value, saved_bits=sv.value, out_len=ol.value, consumed=co.value, size_bits_in=sb.value)

    def output(self, i):
        """DeflateStream.asBytes() of stream i (the re-serialised stream, whether or not bits were saved)."""
        r = self.result(i)
        buf = ctypes.create_string_buffer(max(1, r["out_len"]))
        rc = self.L.d4g_batch_copy_output(self.h, i, buf, r["out_len"])
        if rc != 0:
            raise RuntimeError(self.L.d4g_last_error().decode())
        return buf.raw[:r["out_len"]]

    def decoded(self, i):
        n = ctypes.c_size_t()
        rc = self.L.d4g_batch_copy_decoded(self.h, i, None, 0, ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(self.L.d4g_last_error().decode())
        buf = ctypes.create_string_buffer(max(1, n.value))
        rc = self.L.d4g_batch_copy_decoded(self.h, i, buf, n.value, ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(self.L.d4g_last_error().decode())
        return buf.raw[:n.value]

    def parse(self):
        """DeflateStream.parse for every stream (no optimisation); decoded bytes and checksums become available."""
        rc = self.L.d4g_batch_parse(self.h)
        if rc != 0:
            raise RuntimeError("d4g_batch_parse: " + self.L.d4g_last_error().decode())
        return self

    def checksums(self, i):
        """-> (crc32, adler32, isize) of stream i's decoded bytes, computed on the device."""
        c = ctypes.c_uint32()
        a = ctypes.c_uint32()
        n = ctypes.c_int64()
        rc = self.L.d4g_batch_checksums(self.h, i, ctypes.byref(c), ctypes.byref(a), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError(self.L.d4g_last_error().decode())
        return c.value, a.value, n.value

    def stats(self):
        st = d4g_stats()
        self.L.d4g_batch_stats(self.h, ctypes.byref(st))
        return {k: getattr(st, k) for k, _ in d4g_stats._fields_}

    def close(self):
        if self.h:
            self.L.d4g_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EncodeBatch(Batch):
    """Streams produced by the LZ77 encoder kernels (d4g_batch_create_encode): one output per (input, encoder, strategy)
    spec — the Compressor objects of C/CompressionUtil.java:44-78.  run(optimise=False) leaves the encoder's output as it
    is (SingleCompressor.compressSingle); optimise=True also applies Deft.optimiseDeflateStream to every output."""

    def __init__(self, inputs, specs, lib=None):  # noqa: super().__init__ deliberately not called: a different constructor
        self.L = lib or _need()
        self.n = len(specs)
        self._keep = [bytes(s) for s in inputs]
        nin = len(self._keep)
        arr = (ctypes.c_char_p * max(1, nin))(*self._keep)
        lens = (ctypes.c_size_t * max(1, nin))(*[len(s) for s in self._keep])
        sp = (d4g_encoder_spec * max(1, self.n))(*[d4g_encoder_spec(*t) for t in specs])
        self.h = self.L.d4g_batch_create_encode(nin, arr, lens, self.n, sp)
        if not self.h:
            raise RuntimeError("d4g_batch_create_encode: " + self.L.d4g_last_error().decode())

    def run(self, optimise=False, merge_blocks=True):
        rc = self.L.d4g_batch_run_encode(self.h, 1 if optimise else 0, 1 if merge_blocks else 0)
        if rc != 0:
            raise RuntimeError("d4g_batch_run_encode: " + self.L.d4g_last_error().decode())
        return self


def deflate_streams(inputs, encoder=ENC_JVM, strategy=STRATEGY_DEFAULT, lib=None):
    """SingleCompressor.compressSingle for every input (d4g_deflate_streams)."""
    L = lib or _need()
    n = len(inputs)
    keep = [bytes(s) for s in inputs]
    arr = (ctypes.c_char_p * max(1, n))(*keep)
    lens = (ctypes.c_size_t * max(1, n))(*[len(s) for s in keep])
    out = (ctypes.c_void_p * max(1, n))()
    olen = (ctypes.c_size_t * max(1, n))()
    rc = L.d4g_deflate_streams(n, arr, lens, encoder, strategy, out, olen)
    if rc != 0:
        raise RuntimeError("d4g_deflate_streams: " + L.d4g_last_error().decode())
    res = []
    for i in range(n):
        res.append(ctypes.string_at(out[i], olen[i]))
        L.d4g_free(out[i])
    return res


ZOPFLI_SPLIT_FIRST, ZOPFLI_SPLIT_LAST, ZOPFLI_SPLIT_NONE = 0, 1, 2     # Options.BlockSplitting (CafeUndZopfli) / blocksplitting[last] (jzopfli)


def zopfli_streams(inputs, iterations=20, splitting=ZOPFLI_SPLIT_FIRST, max_blocks=15, master_block=8 << 20, lib=None):
    """MultiCafeUndZopfliCompressor / MultiJZopfliCompressor.compressWithOptions for every input (d4g_zopfli_streams)."""
    L = lib or _need()
    n = len(inputs)
    keep = [bytes(s) for s in inputs]
    arr = (ctypes.c_char_p * max(1, n))(*keep)
    lens = (ctypes.c_size_t * max(1, n))(*[len(s) for s in keep])
    out = (ctypes.c_void_p * max(1, n))()
    olen = (ctypes.c_size_t * max(1, n))()
    rc = L.d4g_zopfli_streams(n, arr, lens, iterations, splitting, max_blocks, master_block, out, olen)
    if rc != 0:
        raise RuntimeError("d4g_zopfli_streams: " + L.d4g_last_error().decode())
    res = []
    for i in range(n):
        res.append(ctypes.string_at(out[i], olen[i]))
        L.d4g_free(out[i])
    return res


MODE_NONE, MODE_CHEAP, MODE_ZOPFLI, MODE_ZOPFLI_EXTENSIVE, MODE_ZOPFLI_VERY_EXTENSIVE = range(5)   # M/Optimise.java RecompressMode


class CompressionUtil:
    """C/CompressionUtil.java as CMDUtil configures it (M/CMDUtil.java:44-50): the compressor list of a recompress
    mode, every output through Deft.optimiseDeflateStream, strict minimum by parsed bit size in list order."""

    def __init__(self, mode=MODE_CHEAP, iter=20, mergeBlocks=True, lib=None):   # noqa: A002 (the reference's name)
        self.mode, self.iter, self.mergeBlocks, self._lib = mode, iter, mergeBlocks, lib
        self.last_winner = None

    def compress_many(self, buffers):
        L = self._lib or _need()
        n = len(buffers)
        keep = [bytes(b) for b in buffers]
        arr = (ctypes.c_char_p * max(1, n))(*keep)
        lens = (ctypes.c_size_t * max(1, n))(*[len(b) for b in keep])
        out = (ctypes.c_void_p * max(1, n))()
        olen = (ctypes.c_size_t * max(1, n))()
        win = (ctypes.c_int32 * max(1, n))()
        rc = L.d4g_compress(n, arr, lens, self.mode, self.iter, 1 if self.mergeBlocks else 0, out, olen, win)
        if rc != 0:
            raise IOError("Unable to compress data: " + L.d4g_last_error().decode())   # CompressionUtil.java:177-179
        res = []
        for i in range(n):
            res.append(ctypes.string_at(out[i], olen[i]))
            L.d4g_free(out[i])
        self.last_winner = list(win[:n])
        return res

    def compress(self, uncompressedData, threaded=False):
        """-> the smallest optimised stream (CompressionUtil.compress :106-182).  `threaded` is accepted and ignored:
        the result is the list-order one, which the reference's threaded path only matches when no sizes tie."""
        return self.compress_many([uncompressedData])[0]


def recompress_streams(streams, mode=MODE_CHEAP, mergeBlocks=True, iter=20, lib=None):   # noqa: A002
    """CMDUtil.optimise's per-stream work (M/CMDUtil.java:70-105) for a list of raw deflate streams.
    -> list of dict(status, saved_bits, recompress_saved, out) — out is None unless status == 0 (changed)."""
    L = lib or _need()
    n = len(streams)
    keep = [bytes(b) for b in streams]
    arr = (ctypes.c_char_p * max(1, n))(*keep)
    lens = (ctypes.c_size_t * max(1, n))(*[len(b) for b in keep])
    out = (ctypes.c_void_p * max(1, n))()
    olen = (ctypes.c_size_t * max(1, n))()
    sv = (ctypes.c_int64 * max(1, n))()
    rs = (ctypes.c_int64 * max(1, n))()
    st = (ctypes.c_int32 * max(1, n))()
    rc = L.d4g_recompress_streams(n, arr, lens, mode, iter, 1 if mergeBlocks else 0, out, olen, sv, rs, st)
    if rc != 0:
        raise RuntimeError("d4g_recompress_streams: " + L.d4g_last_error().decode())
    res = []
    for i in range(n):
        data = None
        if out[i]:
            data = ctypes.string_at(out[i], olen[i])
            L.d4g_free(out[i])
        res.append(dict(status=st[i], saved_bits=sv[i], recompress_saved=rs[i], out=data))
    return res


class Deft:
    """Static façade — B/Deft.java."""

    @staticmethod
    def optimiseDeflateStream(original, mergeBlocks=True):
        """Returns new bytes iff the stream parses and bits were saved; otherwise the SAME object it was given."""
        b = Batch([original]).run(mergeBlocks)
        try:
            if b.result(0)["status"] == 0:
                return b.output(0)
            return original
        finally:
            b.close()

    @staticmethod
    def getSizeBitsFallback(deflateStream, lib=None):
        L = lib or _need()
        bits = ctypes.c_int64()
        rc = L.d4g_size_bits_fallback(bytes(deflateStream), len(deflateStream), ctypes.byref(bits))
        if rc != 0:
            raise RuntimeError(L.d4g_last_error().decode())
        return bits.value


class DeflateStream:
    """Object API — B/deflate/DeflateStream.java (parse / optimise / getSizeBits / getUncompressedData / asBytes)."""
    DEFAULT_NAME = "unnamed stream"

    def __init__(self, name=None, lib=None):
        self.name = name or self.DEFAULT_NAME
        self._lib = lib
        self._data = None
        self._batch = None
        self._parsed = None

    def getName(self):
        return self.name

    def parse(self, data):
        """-> bool.  `consumed` gives the bytes DeflateStream.parse(InputStream) reads (K/GZFile.java:84 relies on it)."""
        self._data = bytes(data)
        self.close()
        L = self._lib or _need()
        out = ctypes.c_void_p()
        ol = ctypes.c_size_t()
        co = ctypes.c_size_t()
        st = ctypes.c_int32()
        rc = L.d4g_inflate(self._data, len(self._data), ctypes.byref(out), ctypes.byref(ol), ctypes.byref(co), ctypes.byref(st))
        if rc != 0:
            raise RuntimeError(L.d4g_last_error().decode())
        self.consumed = co.value
        if st.value < 0:
            self._parsed = None
            return False
        self._parsed = ctypes.string_at(out.value, ol.value)
        L.d4g_free(out)
        return True

    def getUncompressedData(self):
        return self._parsed

    def getSizeBits(self):
        if self._batch is not None:
            r = self._batch.result(0)
            return r["size_bits_in"] - r["saved_bits"]
        return Deft.getSizeBitsFallback(self._data, self._lib)

    def optimise(self, mergeBlocks=True):
        """-> bits saved (DeflateStream.optimise(boolean), :496)."""
        self.close()
        self._batch = Batch([self._data], lib=self._lib).run(mergeBlocks)
        return self._batch.result(0)["saved_bits"]

    def asBytes(self):
        """DeflateStream.asBytes() (:652) = write(): the stream as it stands.  Before optimise() that is the parsed
        stream re-serialised unchanged: the bytes parse() consumed, with the padding bits of the last byte written as
        zeros (DeflateStream.write :143 pads with zero bits)."""
        if self._batch is not None:
            return self._batch.output(0)
        if self._parsed is None:
            raise RuntimeError("asBytes() on a stream that did not parse")
        out = bytearray(self._data[:self.consumed])
        rem = Deft.getSizeBitsFallback(self._data, self._lib) % 8
        if rem and out:
            out[-1] &= (1 << rem) - 1
        return bytes(out)

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None


class DeflateFilesContainer:
    """K/DeflateFilesContainer.java:18-43 — the batch seam: independent streams, results in index order."""

    @staticmethod
    def optimise(streams, mergeBlocks=True, printer=None):
        """streams: list of raw deflate byte strings.  Returns (total bits saved, [output bytes or the original]).
        `printer`, when given, receives the transcript lines the reference prints (:31-40)."""
        b = Batch(streams).run(mergeBlocks)
        total = 0
        outs = []
        try:
            for i, s in enumerate(streams):
                r = b.result(i)
                saved = r["saved_bits"] if r["status"] >= 0 else 0
                if printer and saved > 0:
                    printer("%d bits saved in stream %d (%s)" % (saved, i, DeflateStream.DEFAULT_NAME))
                total += saved
                outs.append(b.output(i) if r["status"] >= 0 else s)
            if printer and total > 0:
                printer("Total bits saved %d" % total)
        finally:
            b.close()
        return total, outs
