"""The drop-in boundary on the GPU: the one-shot C-ABI entry points the JNI shim binds (d4g_optimise_streams,
d4g_size_bits_fallback, d4g_inflate) called directly, and re-entrancy from several threads
(C/CompressionUtil.java:111-117 calls Deft.optimiseDeflateStream from a thread pool, one task per compressor)."""
import json
import os
import threading
import zlib

import pytest

import abi_calls
import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    import deft4j_amd
    return deft4j_amd.init(0)


def rd(name):
    return open(os.path.join(G, name), "rb").read()


def mixed_streams():
    good = rd("deflate-dynamic.parse.deflate")
    return [rd("asyoulik_asyoulik-zopfli.s00.in.deflate"),     # changed (17 bits)
            rd("text.s00.in.deflate"),                           # unchanged
            b"\x07garbage", b"", good[:len(good) // 2],          # parse errors
            synth.make_stream(40000, 0xD4F7), synth.deflate9(b"a" * 70000), synth.deflate9(b"")]


@pytest.mark.parametrize("merge", [True, False])
def test_one_shot_entry_points_vs_oracle(L, merge):
    kinds = abi_calls.check_one_shot_entry_points(L, O, mixed_streams(), merge)
    assert kinds == {0, 1, -1}          # changed, unchanged and parse-error streams in ONE call


def test_empty_call_and_null_arguments(L):
    rc, res = abi_calls.optimise_streams(L, [], True)
    assert rc == 0 and res == []
    assert L.d4g_optimise_streams(1, None, None, 1, None, None, None, None) < 0
    assert L.d4g_size_bits_fallback(None, 0, None) < 0


def test_two_threads_hammer_the_abi(L):
    """Two threads call d4g_optimise_streams concurrently (different inputs, different merge flags); every result
    must equal the single-threaded one.  The library serialises device work behind its mutex and binds its device
    in every entry point (HIP's current device is per thread)."""
    jobs = [([synth.make_stream(30000 + 1000 * k, 100 + k), rd("lz-twice-twice.s00.in.deflate"), b"\x07"], bool(k & 1)) for k in range(6)]
    want = [abi_calls.optimise_streams(L, s, m) for s, m in jobs]
    got = {}
    errs = []

    def worker(tid):
        try:
            for rep in range(3):
                for k in range(tid, len(jobs), 2):
                    got[(tid, rep, k)] = abi_calls.optimise_streams(L, jobs[k][0], jobs[k][1])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(got) == 3 * len(jobs)
    for (tid, rep, k), r in got.items():
        assert r == want[k], (tid, rep, k)


def test_object_api_before_and_after_optimise(L):
    """DeflateStream.asBytes() before optimise() re-serialises the parsed stream unchanged (B/deflate/DeflateStream.java:652)."""
    import deft4j_amd as D
    a = rd("asyoulik_asyoulik-zopfli.s00.in.deflate")
    s = D.DeflateStream()
    assert s.parse(a + b"trailer") and s.consumed == len(a)
    assert s.asBytes() == a
    assert s.getSizeBits() == 370787
    assert s.optimise() == 17
    assert s.asBytes() == rd("asyoulik_asyoulik-zopfli.s00.out.deflate")
    s.close()


def test_shutdown_and_reinit(L):
    import deft4j_amd as D
    a = rd("lz-twice-twice.s00.in.deflate")
    want = rd("lz-twice-twice.s00.out.deflate")
    L.d4g_shutdown()
    arr_rc, _ = abi_calls.optimise_streams(L, [a], True)
    assert arr_rc < 0                                  # not initialised: loud failure, no fallback
    assert L.d4g_init(0) == 0
    rc, res = abi_calls.optimise_streams(L, [a], True)
    assert rc == 0 and res[0][2] == want
    assert L.d4g_init(0) == 0                          # idempotent on the same device


def test_two_contexts_on_one_gpu_and_the_sharded_one_shot():
    """One process, several devices (d4g_init_devices): two logical contexts on the one GPU of the box — own memory pool,
    own programs, own streams each.  A batch created on context 1, batches on both contexts driven from two threads at
    once, and d4g_optimise_streams_sharded over both give byte for byte what context 0 alone gives."""
    import deft4j_amd as D
    D.init(0)
    D.init_devices([0, 0])
    lib = D.load_library()
    assert lib.d4g_device_count() == 2
    streams = [synth.make_stream(n, s) for n, s in ((60000, 1), (250000, 2), (3000, 3), (180000, 4), (90000, 5))] + [b"\x07", synth.deflate9(b"")]

    def snap(b):
        res = []
        for i in range(len(streams)):
            r = b.result(i)
            res.append((r["status"], b.output(i) if r["status"] == 0 else None, r["saved_bits"] if r["status"] == 0 else 0))
        return res
    ref = D.Batch(streams).run(True)
    want = snap(ref)
    ref.close()
    assert any(w[0] == 0 for w in want)
    b1 = D.Batch(streams, context=1).run(True)
    assert snap(b1) == want
    b1.close()
    got = {}

    def work(ctx):
        b = D.Batch(streams, context=ctx).run(True)
        got[ctx] = snap(b)
        b.close()
    ts = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got[0] == want and got[1] == want
    outs, saved, status = D.optimise_streams_sharded(streams, True)
    for i, (st, o, sv) in enumerate(want):
        assert status[i] == st and saved[i] == sv and outs[i] == o
