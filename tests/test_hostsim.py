"""Runs the HIP kernels of deft4j_amd/csrc inside the test-only CPU emulator (tests/hostsim) and checks them
against the golden fixtures and the oracle.  This is how the kernels are debugged and sanitised without a
GPU; it is not a product path (libdeft4g.so has no CPU fallback — see test_abi.py)."""
import json
import os
import subprocess
import zlib

import pytest

import oracle_lib as O
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
MAN = {p["stem"]: p for p in json.load(open(os.path.join(G, "manifest.json")))["pairs"]}


@pytest.fixture(scope="module")
def sim():
    os.environ["D4G_SIM_BLOCK"] = "64"
    so = os.path.join(ROOT, "tests", "hostsim", "libdeft4g_hostsim.so")
    subprocess.check_call([os.path.join(ROOT, "tests", "hostsim", "build.sh")])
    import deft4j_amd as D
    L = D.load_library(so)
    D.init(0, lib=L)
    return D, L


def rd(n):
    return open(os.path.join(G, n), "rb").read()


@pytest.mark.parametrize("stem", ["lz-twice-twice.s00", "deflate-store-2.s00", "text.s00", "text.s01", "text.s02", "apng_ball.s12",
                                  "apng_ball.s09"])
def test_golden_pairs_in_the_emulator(sim, stem):
    D, L = sim
    p = MAN[stem]
    a = rd(stem + ".in.deflate")
    b = D.Batch([a], lib=L).run(p["merge_blocks"])
    r = b.result(0)
    assert r["saved_bits"] == p["saved_bits"]
    assert r["consumed"] == len(a)
    assert b.output(0) == rd(stem + ".out.deflate")
    assert b.decoded(0) == zlib.decompress(a, -15)
    b.close()


def test_one_shot_abi_entry_points_in_the_emulator(sim):
    """d4g_optimise_streams / d4g_size_bits_fallback / d4g_inflate (the functions the JNI shim binds) called
    directly: changed, unchanged and malformed streams in one call, NULL-out contract, all result slots defined."""
    import abi_calls
    D, L = sim
    good = rd("deflate-dynamic.parse.deflate")
    streams = [rd("lz-twice-twice.s00.in.deflate"), rd("text.s00.in.deflate"), b"\x07garbage", b"", good[:len(good) // 2],
               synth.make_stream(1200, 8), synth.deflate9(b"")]
    for merge in (True, False):
        kinds = abi_calls.check_one_shot_entry_points(L, O, streams, merge)
        assert kinds == {0, 1, -1}
    s = D.DeflateStream(lib=L)
    a = rd("lz-twice-twice.s00.in.deflate")
    assert s.parse(a + b"xyz") and s.consumed == len(a) and s.asBytes() == a
    assert s.optimise() == MAN["lz-twice-twice.s00"]["saved_bits"] and s.asBytes() == rd("lz-twice-twice.s00.out.deflate")
    s.close()


@pytest.mark.parametrize("mode", ["levels", "persistent", "fused"])
def test_batch_of_synthetic_streams_matches_oracle(sim, monkeypatch, mode):
    monkeypatch.setenv("D4G_EXEC", mode)
    D, L = sim
    t = synth.reptext(6000, 11)
    c0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    ins = [synth.make_stream(1500, 3), synth.deflate9(t, zlib.Z_HUFFMAN_ONLY), c0.compress(t[:500]) + c0.flush(),
           synth.deflate9(b""), b"\x07", synth.deflate9(b"a" * 3000)]
    for merge in (False, True):
        b = D.Batch(ins, lib=L).run(merge)
        for i, a in enumerate(ins):
            rc, want, saved, consumed, _ = O.optimise(a, merge)
            r = b.result(i)
            assert r["status"] == rc, (i, merge)
            if rc >= 0:
                assert r["saved_bits"] == saved and b.output(i) == want, (i, merge)
        b.close()


@pytest.mark.parametrize("threads", ["128", "512"])
def test_block_decoder_with_several_waves(sim, monkeypatch, threads):
    """The GPU decodes a block with a whole workgroup (512 threads: a batch is 512 chunks of 512 bits, the exits travel
    between waves through LDS); the emulator's default is one wave.  Here the multi-wave path runs in the emulator:
    golden pairs, a multi-block stream whose blocks need several batches at 128 threads, stored / fixed / empty blocks."""
    monkeypatch.setenv("D4G_SIM_PARSE_THREADS", threads)
    D, L = sim
    for stem in ("lz-twice-twice.s00", "text.s01", "apng_ball.s09"):
        p = MAN[stem]
        a = rd(stem + ".in.deflate")
        b = D.Batch([a], lib=L).run(p["merge_blocks"])
        assert b.result(0)["saved_bits"] == p["saved_bits"] and b.output(0) == rd(stem + ".out.deflate")
        b.close()
    raw = synth.reptext(120000, 31)
    c0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    ins = [synth.deflate9(raw), c0.compress(raw[:3000]) + c0.flush(), c1.compress(raw[:9000]) + c1.flush(), synth.deflate9(b""),
           synth.deflate9(raw, zlib.Z_HUFFMAN_ONLY)]
    b = D.Batch(ins, lib=L).parse()
    for i, a in enumerate(ins):
        assert b.decoded(i) == zlib.decompress(a, -15), i
    b.close()
    rc, want, saved, _, _ = O.optimise(ins[0], False)
    b = D.Batch([ins[0]], lib=L).run(False)
    assert b.result(0)["saved_bits"] == saved and b.output(0) == want
    b.close()


def test_long_blocks_decode_in_checkpoint_segments(sim):
    """Blocks longer than 1024 tokens are emitted by one wave per checkpoint segment (and blocks beyond 32 Ki
    tokens thin their checkpoint set): tokens, decoded bytes and the merged histogram must still be exact."""
    D, L = sim
    raw = synth.reptext(400000, 5)
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 9)  # memLevel 9: 32 Ki symbols per block
    s9 = c.compress(raw) + c.flush()
    rc, merged, _, _, _ = O.optimise(synth.deflate9(raw), True)  # one block of > 32 Ki tokens
    assert rc == 0 and max(bi[1] for bi in O.block_info(merged)) > 32768
    b = D.Batch([s9], lib=L).parse()
    assert b.decoded(0) == raw
    b.close()
    rc, want, saved, _, _ = O.optimise(merged, False)
    b = D.Batch([merged], lib=L).run(False)
    r = b.result(0)
    assert r["status"] == rc and r["saved_bits"] == saved and b.decoded(0) == raw
    assert b.output(0) == (want if rc == 0 else merged)
    b.close()


def test_wave_wide_tree_builder_in_the_emulator():
    """The GPU builds Huffman trees with a whole wave (priority queue in a register pair + LDS); the default
    emulator build uses the one-lane builder, this variant runs the wave-wide code (slowly) on one small stream."""
    os.environ["D4G_SIM_BLOCK"] = "64"
    subprocess.check_call([os.path.join(ROOT, "tests", "hostsim", "build.sh"), "waveheap"])
    import deft4j_amd as D
    L = D.load_library(os.path.join(ROOT, "tests", "hostsim", "libdeft4g_hostsim_wh.so"))
    D.init(0, lib=L)
    a = synth.make_stream(1500, 3)
    rc, want, saved, _, _ = O.optimise(a, False)
    b = D.Batch([a], lib=L).run(False)
    r = b.result(0)
    assert r["status"] == rc and r["saved_bits"] == saved and b.output(0) == (want if rc == 0 else a)
    b.close()


def test_pack_summary_matches_the_pair_by_pair_packing(sim):
    """d4g_pack_kinds (closed form, used by the header search) == d4g_pack_run (HuffmanTable.pack's loops) as
    multisets of pairs, for all 256 flag sets, zero / non-zero values and run lengths 1..420."""
    D, L = sim
    import ctypes
    L.d4g_test_pack_kinds.restype = ctypes.c_longlong
    assert L.d4g_test_pack_kinds() == 0


def test_least_expensive_pass_both_directions(sim, monkeypatch):
    """removeDistLitLeastExpensive works from the static bin statistics minus the expanded records; masks that are
    mostly expanded are walked over the unexpanded records instead.  Force that second direction."""
    D, L = sim
    monkeypatch.setenv("D4G_SIM_LEAST_DIRECT", "1")
    for stem in ("text.s00", "lz-twice-twice.s00"):
        p = MAN[stem]
        a = rd(stem + ".in.deflate")
        b = D.Batch([a], lib=L).run(p["merge_blocks"])
        assert b.result(0)["saved_bits"] == p["saved_bits"] and b.output(0) == rd(stem + ".out.deflate")
        b.close()


def test_two_merge_chains_in_one_stream(sim):
    """mergeBlocks can finish one merged block and start merging further down the stream; the finished block must
    survive the re-use of the merge arenas (found by scripts/gpu_fuzz.py: eight blocks that merge four and four)."""
    import random
    D, L = sim
    rng = random.Random(3)
    t = synth.reptext(6000, 4)
    parts = [t[i:i + 1500] for i in range(0, 6000, 1500)] + [bytes(rng.randrange(200, 256) for _ in range(1500)) for _ in range(4)]
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
    a = b"".join(c.compress(p) + c.flush(zlib.Z_BLOCK) for p in parts) + c.flush()
    rc, want, saved, _, _ = O.optimise(a, True)
    assert rc == 0 and sum(1 for bi in O.block_info(want) if bi[1] > 3000) == 2   # two merged blocks
    b = D.Batch([a], lib=L).run(True)
    assert b.result(0)["saved_bits"] == saved and b.output(0) == want
    b.close()


def test_scan_prefilter_probe_find_every_dynamic_block(sim):
    """The header scan + lane-per-candidate pre-filter must hand every dynamic block to the probe: a block they miss
    is still parsed (the chain walk probes unseen positions one by one) but serially — so count those probes."""
    D, L = sim
    raw = synth.reptext(300000, 9)
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8)
    a = b"".join(c.compress(raw[i:i + 20000]) + c.flush(zlib.Z_BLOCK) for i in range(0, len(raw), 20000)) + c.flush()
    nblocks = len(O.block_info(a))
    b = D.Batch([a], lib=L).parse()
    st = b.stats()
    assert b.decoded(0) == raw
    assert st["n_blocks"] == nblocks and st["scan_confirmed"] >= nblocks - 2 and st["exact_probes"] <= 2, st
    b.close()


def test_lz77_encoder_in_the_emulator(sim):
    """The LZ77 front end (hash sort, wave-wide chain search, speculative lazy parse, zlib's trees) in the emulator:
    byte-identical to Python zlib level 9 for three strategies, and encode+optimise equals the oracle's optimise of
    zlib's bytes."""
    import random
    D, L = sim
    rng = random.Random(3)
    ins = [b"", b"a", b"abcabcabcabc" * 30, synth.reptext(9000, 1), bytes(rng.randrange(256) for _ in range(5000)), b"\0" * 20000]
    zs = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY)
    specs = [(i, D.ENC_JVM, st) for i in range(len(ins)) for st in range(3)]
    b = D.EncodeBatch(ins, specs, lib=L).run(False)
    for k, (i, _, st) in enumerate(specs):
        c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zs[st])
        assert b.output(k) == c.compress(ins[i]) + c.flush(), (i, st)
    b.close()
    b = D.EncodeBatch(ins, specs, lib=L).run(True, True)
    for k, (i, _, st) in enumerate(specs):
        c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zs[st])
        enc = c.compress(ins[i]) + c.flush()
        rc, want, saved, _, _ = O.optimise(enc, True)
        r = b.result(k)
        assert r["status"] == rc and r["saved_bits"] == saved, (i, st)
        assert b.output(k) == (want if rc == 0 else enc), (i, st)
    b.close()


def test_recompress_modes_in_the_emulator(sim):
    """CompressionUtil.compress (mode CHEAP: six zlib-family compressors, each output optimised, strict minimum in list
    order) and CMDUtil's recompress-compare-graft loop against the same orchestration over the oracles."""
    import oracle_compose as OC
    D, L = sim
    raws = [synth.reptext(2500, 21), bytes(range(256)) * 4, b"", b"ab" * 900]
    for merge in (True, False):
        cu = D.CompressionUtil(D.MODE_CHEAP, mergeBlocks=merge, lib=L)
        outs = cu.compress_many(raws)
        for r, o, w in zip(raws, outs, cu.last_winner):
            want, win = OC.compress(r, merge)
            assert o == want and w == win, (len(r), merge)
            assert zlib.decompress(o, -15) == r
    # streams: a weakly compressed one (recompression wins), a zlib-9 one, a malformed one
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15)
    streams = [c1.compress(raws[0]) + c1.flush(), synth.deflate9(raws[3]), b"\x07junk", synth.deflate9(b"")]
    for merge in (True, False):
        res = D.recompress_streams(streams, D.MODE_CHEAP, merge, lib=L)
        for a, r in zip(streams, res):
            assert r == OC.recompress(a, merge), (len(a), merge)
    assert res[0]["recompress_saved"] > 0
    with pytest.raises(IOError):
        D.CompressionUtil(7, lib=L).compress(b"abc")                  # no such RecompressMode: loud failure


def test_zopfli_encoder_in_the_emulator(sim):
    """The Zopfli kernels on the CPU emulation against the oracle (portable log flavour): code lengths, the match table
    with a block end inside the input, whole streams for the three splitting options, and the recompress modes that use
    them (list order JVM, [JZopfli], CafeUndZopfli, JZlib: C/CompressionUtil.java:44-78)."""
    import ctypes
    import random
    import numpy as np
    import oracle_compose as OC
    import zopf_lib as ZF
    D, L = sim
    rng = random.Random(5)
    for n, mb in ((288, 15), (32, 15), (19, 7)):
        for _ in range(25):
            f = [0] * n
            for i in rng.sample(range(n), rng.choice((1, 2, 3, n // 2, n))):
                f[i] = rng.choice((1, 1, 2, 3, rng.randrange(1, 50), rng.randrange(1, 100000)))
            out = (ctypes.c_uint32 * n)()
            assert L.d4g_debug_zopfli_code_lengths((ctypes.c_uint32 * n)(*f), n, mb, out) == 0
            assert list(out) == ZF.length_limited(f, mb)
    runs = b"".join(bytes([rng.randrange(3)]) * rng.randrange(1, 300) for _ in range(8))
    for data, end in ((synth.reptext(1000, 5), 600), (runs, len(runs) // 2)):
        n, e = len(data), end or len(data)
        l16, d16 = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        assert L.d4g_debug_zopfli_table(data, n, end, l16.ctypes.data, d16.ctypes.data, None) == 0
        ol, od = np.zeros(e, np.uint16), np.zeros(e, np.uint16)
        ZF.lib().zopf_match_table(data, 0, e, ol.ctypes.data, od.ctypes.data, None)
        assert np.array_equal(l16[:e], np.where(ol >= 3, ol, 0)) and np.array_equal(d16[:e], np.where(ol >= 3, od, 0))
    text = zlib.decompress(rd("asyoulik_asyoulik-gzip.s00.in.deflate"), -15)
    mix = text[:150] + bytes(rng.randrange(256) for _ in range(80)) + bytes(540) + text[3000:3070]
    for split in (ZF.SPLIT_FIRST, ZF.SPLIT_LAST, ZF.SPLIT_NONE):
        datas = [mix, b"", b"hello hello hello hello", synth.reptext(500, 9)] if split == ZF.SPLIT_FIRST else [mix]
        outs = D.zopfli_streams(datas, 3, split, 15, 8 << 20, lib=L)
        for d, o in zip(datas, outs):
            assert o == ZF.deflate(d, 3, split, 15, 8 << 20, ZF.LOG_PORTABLE), (len(d), split)
    assert D.zopfli_streams([mix], 2, ZF.SPLIT_FIRST, 0, 400, lib=L)[0] == ZF.deflate(mix, 2, ZF.SPLIT_FIRST, 0, 400, ZF.LOG_PORTABLE)
    data = [text[:110], b""]
    for mode in (D.MODE_ZOPFLI, D.MODE_ZOPFLI_VERY_EXTENSIVE):
        cu = D.CompressionUtil(mode, 2, True, lib=L)
        for d, o, w in zip(data, cu.compress_many(data), cu.last_winner):
            assert (o, w) == OC.compress(d, True, mode, 2), (mode, len(d))


def test_gzip_file_through_mode_cheap_in_the_emulator(sim):
    """`deft4j optimise --mode=CHEAP` on a small gzip file through the container layer (containers.optimise_files):
    the weakly compressed member is recompressed and grafted, the trailer recomputed, the transcript carries the
    reference's recompression lines (M/CMDUtil.java:95,103)."""
    import gzip
    import oracle_compose as OC
    from deft4j_amd import containers
    D, L = sim
    text = synth.reptext(2600, 77)
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15)
    payload = c1.compress(text) + c1.flush()
    gz = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\x03" + payload + (zlib.crc32(text) & 0xffffffff).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    (out, lines), = containers.optimise_files([gz], True, lib=L, mode=D.MODE_CHEAP)
    assert gzip.decompress(out) == text
    want = OC.recompress(payload, True)
    assert want["recompress_saved"] > 0 and out[10:-8] == want["out"]
    assert lines[-1] == "Saved %d bits with recompression" % want["recompress_saved"]


def test_persistent_wait_expiry_is_recovered_by_the_level_executor(sim, monkeypatch):
    """A wait of the persistent executor that gives up (D4G_SPIN_LIMIT=-1: every one does) is not an error: nothing is
    selected, the round runs again with the level executor, the output is the oracle's and the fallback is counted."""
    monkeypatch.setenv("D4G_EXEC", "persistent")
    monkeypatch.setenv("D4G_SPIN_LIMIT", "-1")
    D, L = sim
    a = synth.make_stream(9000, 4)
    for merge in (False, True):
        b = D.Batch([a], lib=L).run(merge)
        rc, want, saved, _, _ = O.optimise(a, merge)
        assert b.output(0) == want and b.result(0)["saved_bits"] == saved
        assert b.stats()["persist_fallbacks"] > 0
        b.close()


def test_zopfli_change_point_pool_grows_instead_of_failing(sim, monkeypatch):
    """Positions with many record-setting matches (nearest = shortest) overflow the change-point pool's first size:
    the match kernels are run again with a larger pool and the streams are the oracle's."""
    import zopf_lib as ZF
    monkeypatch.setenv("D4G_ZF_POOL_WORDS", "64")
    D, L = sim
    S = synth.reptext(70, 3)
    ladder = b"".join(S[:k] for k in range(60, 2, -1)) + S
    data = ladder * 2
    for split in (ZF.SPLIT_FIRST, ZF.SPLIT_NONE):
        out = D.zopfli_streams([data, S], 2, split, 15, 8 << 20, lib=L)
        assert out[0] == ZF.deflate(data, 2, split, 15, 8 << 20, ZF.LOG_PORTABLE)
        assert out[1] == ZF.deflate(S, 2, split, 15, 8 << 20, ZF.LOG_PORTABLE)
    # unlimited splitting (blocksplittingmax 0): more split points than the old fixed buffer is no longer an error either
    assert D.zopfli_streams([data], 1, ZF.SPLIT_FIRST, 0, 8 << 20, lib=L)[0] == ZF.deflate(data, 1, ZF.SPLIT_FIRST, 0, 8 << 20, ZF.LOG_PORTABLE)


def test_contexts_and_the_sharded_one_shot_in_the_emulator(sim):
    """d4g_init_devices / d4g_batch_create_on / d4g_optimise_streams_sharded (one process driving several GPUs): two
    contexts (the emulator has one device behind both): a batch created on context 1 and the sharded one-shot give what
    the single-context calls give."""
    D, L = sim
    D.init_devices([0, 0], lib=L)
    assert L.d4g_device_count() == 2
    streams = [synth.make_stream(n, s) for n, s in ((900, 1), (2500, 2), (300, 3), (1800, 4))] + [b"\x07", synth.deflate9(b"")]
    want = [O.optimise(a, True) for a in streams]
    b = D.Batch(streams, lib=L, context=1).run(True)
    for i, w in enumerate(want):
        assert b.result(i)["status"] == w[0]
        if w[0] == 0:
            assert b.output(i) == w[1]
    b.close()
    outs, saved, status = D.optimise_streams_sharded(streams, True, lib=L)
    for i, w in enumerate(want):
        assert status[i] == w[0] and outs[i] == (w[1] if w[0] == 0 else None) and saved[i] == (w[2] if w[0] == 0 else 0)
    assert D.Batch(streams[:1], lib=L, context=0).run(False).output(0) == O.optimise(streams[0], False)[1]
    with pytest.raises(RuntimeError):
        D.Batch(streams[:1], lib=L, context=7)


def test_fused_executor_mask_tasks_in_their_any_length_form(sim, monkeypatch):
    """The fused executor's mask tasks have a register form (blocks of up to 16384 back-references) and a chunked form for
    longer (merged) blocks: D4G_FUSED_REG_WORDS=0 sends every block through the chunked form."""
    monkeypatch.setenv("D4G_FUSED_REG_WORDS", "0")
    D, L = sim
    ins = [synth.make_stream(9000, 21), synth.make_stream(2500, 22), synth.deflate9(synth.pngidat(6000, 3, 64))]
    for merge in (False, True):
        b = D.Batch(ins, lib=L).run(merge)
        for i, a in enumerate(ins):
            rc, want, saved, _, _ = O.optimise(a, merge)
            assert b.result(i)["status"] == rc and b.output(i) == want and b.result(i)["saved_bits"] == saved, (i, merge)
        assert b.stats()["rounds_fused"] > 0
        b.close()


def test_cluster_mode_in_the_emulator(sim, monkeypatch):
    """k_search_cluster (many workgroups on one long merged block; the emulator runs the control workgroup first, so it works
    every command off alone, then the helpers, which find the launch over): merge rounds forced through it, output == oracle."""
    monkeypatch.setenv("D4G_FUSED_MAX_REFS", "0")
    monkeypatch.setenv("D4G_CLUSTER_MIN_REFS", "0")
    D, L = sim
    c0 = zlib.compressobj(9, zlib.DEFLATED, -15)
    t = synth.reptext(30000, 8)
    multi = c0.compress(t[:9000]) + c0.flush(zlib.Z_FULL_FLUSH) + c0.compress(t[9000:20000]) + c0.flush(zlib.Z_FULL_FLUSH) + c0.compress(t[20000:]) + c0.flush()
    ins = [multi, synth.make_stream(120000, 3)]
    b = D.Batch(ins, lib=L).run(True)
    for i, a in enumerate(ins):
        rc, want, saved, _, _ = O.optimise(a, True)
        assert b.output(i) == want and b.result(i)["saved_bits"] == saved
    assert b.stats()["rounds_cluster"] > 0
    b.close()
