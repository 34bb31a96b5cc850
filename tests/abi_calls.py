"""Direct ctypes calls of the one-shot C-ABI entry points (include/deft4g.h) — shared by the GPU boundary tests
and the emulator tests.  These are the functions the JNI shim (jni/deft4g_jni.c) binds."""
import ctypes


def optimise_streams(L, streams, merge):
    """d4g_optimise_streams -> (rc, [(status, saved_bits, bytes-or-None)])"""
    n = len(streams)
    keep = [bytes(s) for s in streams]
    arr = (ctypes.c_char_p * n)(*keep)
    lens = (ctypes.c_size_t * n)(*[len(s) for s in keep])
    out = (ctypes.c_void_p * n)(*([0xdead] * n))        # poisoned: the call must define every slot
    olen = (ctypes.c_size_t * n)(*([12345] * n))
    saved = (ctypes.c_int64 * n)(*([-7] * n))
    status = (ctypes.c_int32 * n)(*([99] * n))
    rc = L.d4g_optimise_streams(n, arr, lens, 1 if merge else 0, out, olen, saved, status)
    res = []
    for i in range(n):
        data = None
        if out[i]:
            data = ctypes.string_at(out[i], olen[i])
            L.d4g_free(out[i])
        res.append((status[i], saved[i], data))
    return rc, res


def size_bits_fallback(L, data):
    bits = ctypes.c_int64(-1)
    rc = L.d4g_size_bits_fallback(bytes(data), len(data), ctypes.byref(bits))
    return rc, bits.value


def inflate(L, data):
    """d4g_inflate -> (rc, status, decoded-or-None, consumed)"""
    out = ctypes.c_void_p(0xdead)
    ol = ctypes.c_size_t(777)
    co = ctypes.c_size_t(0)
    st = ctypes.c_int32(99)
    rc = L.d4g_inflate(bytes(data), len(data), ctypes.byref(out), ctypes.byref(ol), ctypes.byref(co), ctypes.byref(st))
    dec = None
    if rc == 0 and out.value:
        dec = ctypes.string_at(out.value, ol.value)
        L.d4g_free(out)
    return rc, st.value, dec, co.value


def check_one_shot_entry_points(L, O, streams, merge):
    """Every stream of one d4g_optimise_streams call against the oracle; the NULL-out contract of deft4g.h:
    out[i] is set only for D4G_STREAM_CHANGED."""
    import zlib
    rc, res = optimise_streams(L, streams, merge)
    assert rc == 0
    kinds = set()
    for a, (st, sv, data) in zip(streams, res):
        orc, want, osaved, oconsumed, _ = O.optimise(a, merge)
        assert st == orc, (st, orc)
        kinds.add(st)
        if st == 0:
            assert data == want and sv == osaved and sv > 0
        else:
            assert data is None and sv == 0          # unchanged / parse error: the caller keeps its ORIGINAL array
        rc2, bits = size_bits_fallback(L, a)
        ob = O.size_bits(a)                          # -1: DeflateStream.parse failed
        assert rc2 == 0 and bits == (ob if ob >= 0 else len(a) * 8)   # parse failure -> len * 8 (B/Deft.java:48-54)
        rc3, st3, dec, consumed = inflate(L, a)
        assert rc3 == 0
        if orc >= 0:
            assert st3 >= 0 and dec == zlib.decompress(a, -15) and consumed == oconsumed
        else:
            assert st3 < 0 and dec is None
    return kinds
