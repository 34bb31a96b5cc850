"""ctypes loader of oracle/libzopfli_oracle.so (test infrastructure only) and, where present, of the in-container
proxy libzopfli 1.0.3 used to generate / re-check the committed golden vectors."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_LIBM, LOG_PORTABLE = 0, 1
SPLIT_FIRST, SPLIT_LAST, SPLIT_NONE = 0, 1, 2
PROXY_PATH = "/opt/conda/lib/libzopfli.so.1.0.3"

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "libzopfli_oracle.so"))
        _lib.zopf_deflate.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                      C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
        _lib.zopf_deflate.restype = C.c_int
        _lib.zopf_free.argtypes = [C.c_void_p]
        _lib.zopf_length_limited.argtypes = [C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_uint)]
        _lib.zopf_portable_log.argtypes = [C.c_double]
        _lib.zopf_portable_log.restype = C.c_double
        _lib.zopf_match_table.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def deflate(data, iterations=15, splitting=SPLIT_FIRST, maxblocks=15, master=1000000, logflavor=LOG_LIBM):
    out = C.POINTER(C.c_ubyte)()
    n = C.c_size_t()
    rc = lib().zopf_deflate(bytes(data), len(data), iterations, splitting, maxblocks, master, logflavor, C.byref(out), C.byref(n))
    assert rc == 0
    r = C.string_at(out, n.value)
    lib().zopf_free(out)
    return r


def length_limited(freqs, maxbits):
    n = len(freqs)
    f = (C.c_size_t * n)(*freqs)
    o = (C.c_uint * n)()
    lib().zopf_length_limited(f, n, maxbits, o)
    return list(o)


class _Opt(C.Structure):
    _fields_ = [("verbose", C.c_int), ("verbose_more", C.c_int), ("numiterations", C.c_int), ("blocksplitting", C.c_int),
                ("blocksplittinglast", C.c_int), ("blocksplittingmax", C.c_int)]


_proxy = None


def proxy_available():
    return os.path.exists(PROXY_PATH)


def proxy_deflate(data, iterations=15, blocksplitting=1, maxblocks=15):
    """libzopfli 1.0.3's ZopfliCompress(ZOPFLI_FORMAT_DEFLATE) — generation / re-check of golden vectors only."""
    global _proxy
    if _proxy is None:
        _proxy = C.CDLL(PROXY_PATH)
        _proxy.ZopfliCompress.argtypes = [C.POINTER(_Opt), C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
        _proxy.ZopfliCompress.restype = None
    o = _Opt(0, 0, iterations, blocksplitting, 0, maxblocks)
    out = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    _proxy.ZopfliCompress(C.byref(o), 2, bytes(data), len(data), C.byref(out), C.byref(n))
    r = C.string_at(out, n.value)
    C.CDLL(None).free(out)
    return r
