"""ctypes loader for the LZ77 / zlib-level-9 oracle (oracle/zlib9_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libzlib9_oracle.so")
_SRC = os.path.join(ROOT, "oracle", "zlib9_oracle.c")
_lib = None

DEFAULT, FILTERED, HUFFMAN_ONLY = 0, 1, 2   # strategy ids of the oracle and of the C ABI (d4g_lz77_deflate)
ZLIB, JZLIB = 0, 1                          # flavours


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        L = ctypes.CDLL(_SO)
        L.zl9_deflate.restype = ctypes.c_int
        L.zl9_deflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.zl9_tokens.restype = ctypes.c_int
        L.zl9_tokens.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_longlong)]
        L.zl9_match_table.restype = ctypes.c_int
        L.zl9_match_table.argtypes = [ctypes.c_char_p, ctypes.c_size_t] + [ctypes.c_void_p] * 5
        L.zl9_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def deflate(data, strategy=DEFAULT, flavor=ZLIB):
    L = lib()
    out = ctypes.c_void_p()
    n = ctypes.c_size_t()
    rc = L.zl9_deflate(data, len(data), strategy, flavor, ctypes.byref(out), ctypes.byref(n))
    if rc != 0:
        raise RuntimeError("zl9_deflate failed")
    res = ctypes.string_at(out.value, n.value)
    L.zl9_free(out)
    return res


def tokens(data, strategy=DEFAULT, flavor=ZLIB):
    """-> (tokens u32 array [literal: byte; match: len | dist << 9], block end token counts, (search calls, candidates visited))"""
    L = lib()
    t = ctypes.c_void_p()
    nt = ctypes.c_size_t()
    b = ctypes.c_void_p()
    nb = ctypes.c_size_t()
    st = (ctypes.c_longlong * 2)()
    L.zl9_tokens(data, len(data), strategy, flavor, ctypes.byref(t), ctypes.byref(nt), ctypes.byref(b), ctypes.byref(nb), st)
    toks = np.frombuffer(ctypes.string_at(t.value, nt.value * 4), dtype=np.uint32).copy() if nt.value else np.zeros(0, np.uint32)
    ends = np.frombuffer(ctypes.string_at(b.value, nb.value * 8), dtype=np.uint64).copy() if nb.value else np.zeros(0, np.uint64)
    L.zl9_free(t)
    L.zl9_free(b)
    return toks, ends, (st[0], st[1])


def match_table(data, want_visits=False):
    """Per-position search results (see zl9_match_table): (len4k, dist4k, len1k, dist1k[, visits]) as numpy arrays."""
    L = lib()
    n = len(data)
    arrs = [np.zeros(max(1, n), np.uint16) for _ in range(4)]
    vis = np.zeros(max(1, n), np.uint32) if want_visits else None
    L.zl9_match_table(data, n, *[a.ctypes.data for a in arrs], vis.ctypes.data if want_visits else None)
    res = [a[:n] for a in arrs]
    if want_visits:
        res.append(vis[:n])
    return tuple(res)
