"""ctypes loader for the parity oracle (oracle/deft_oracle.cpp).  Test infrastructure only."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libdeft_oracle.so")
_SRC = os.path.join(ROOT, "oracle", "deft_oracle.cpp")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        L = ctypes.CDLL(_SO)
        L.oracle_optimise.restype = ctypes.c_int
        L.oracle_optimise.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_size_t),
                                      ctypes.POINTER(ctypes.c_longlong)]
        L.oracle_size_bits.restype = ctypes.c_longlong
        L.oracle_size_bits.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.oracle_inflate.restype = ctypes.c_int
        L.oracle_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        L.oracle_block_info.restype = ctypes.c_int
        L.oracle_block_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
        L.oracle_huffman_lengths.restype = None
        L.oracle_huffman_lengths.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.oracle_pack.restype = ctypes.c_int
        L.oracle_pack.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.oracle_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def optimise(data, merge_blocks=True):
    """-> (status, serialised stream or None, saved_bits, consumed_bytes, candidates).
    status 0 changed, 1 unchanged, -1 parse failure."""
    L = lib()
    out = ctypes.c_void_p()
    ol = ctypes.c_size_t()
    sv = ctypes.c_longlong()
    cons = ctypes.c_size_t()
    nc = ctypes.c_longlong()
    rc = L.oracle_optimise(data, len(data), int(merge_blocks), ctypes.byref(out), ctypes.byref(ol),
                           ctypes.byref(sv), ctypes.byref(cons), ctypes.byref(nc))
    res = None
    if out.value:
        res = ctypes.string_at(out.value, ol.value)
        L.oracle_free(out)
    return rc, res, sv.value, cons.value, nc.value


def deft_optimise(data, merge_blocks=True):
    """Deft.optimiseDeflateStream semantics: new bytes iff parse ok and saved > 0, else the input."""
    rc, res, _, _, _ = optimise(data, merge_blocks)
    return res if rc == 0 else data


def size_bits(data):
    return lib().oracle_size_bits(data, len(data))


def inflate(data):
    L = lib()
    out = ctypes.c_void_p()
    ol = ctypes.c_size_t()
    cons = ctypes.c_size_t()
    rc = L.oracle_inflate(data, len(data), ctypes.byref(out), ctypes.byref(ol), ctypes.byref(cons))
    if rc != 0:
        return None, 0
    res = ctypes.string_at(out.value, ol.value)
    L.oracle_free(out)
    return res, cons.value


def block_info(data, max_blocks=4096):
    L = lib()
    buf = (ctypes.c_longlong * (5 * max_blocks))()
    n = L.oracle_block_info(data, len(data), buf, max_blocks)
    if n < 0:
        return None
    return [tuple(buf[i * 5:(i + 1) * 5]) for i in range(min(n, max_blocks))]


def huffman_lengths(freq, limit):
    L = lib()
    n = len(freq)
    f = (ctypes.c_int * n)(*freq)
    ol = (ctypes.c_int * n)()
    oc = (ctypes.c_int * n)()
    L.oracle_huffman_lengths(f, n, limit, ol, oc)
    return list(ol), list(oc)


def pack(codelens, ohh=True, use8=True, use7=True, alt8=False, no_rep=False, no_zrep=False, no_zrep2=False,
         no_rep_zeros=False):
    L = lib()
    flags = (ohh | use8 << 1 | use7 << 2 | alt8 << 3 | no_rep << 4 | no_zrep << 5 | no_zrep2 << 6 | no_rep_zeros << 7)
    n = len(codelens)
    c = (ctypes.c_int * n)(*codelens)
    out = (ctypes.c_int * (2 * n + 8))()
    m = L.oracle_pack(c, n, flags, out, 2 * n + 8)
    return list(out[:m])
