"""Parity tests proper: the HIP path (through the C ABI of libdeft4g.so) against the reference's golden
fixtures and against the CPU oracle on seeded synthetic inputs.  Bit-exact (integer/byte work)."""
import json
import os
import zlib

import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(G, "manifest.json")))


@pytest.fixture(scope="module")
def D():
    import deft4j_amd
    deft4j_amd.init(0)
    return deft4j_amd


def rd(name):
    return open(os.path.join(G, name), "rb").read()


def test_all_reference_fixture_pairs_bit_exact(D):
    """All 30 stream pairs of runTestOpt.sh:3-11, batched by merge flag (streams are independent)."""
    for merge in (True, False):
        pairs = [p for p in MAN["pairs"] if p["merge_blocks"] == merge]
        ins = [rd(p["stem"] + ".in.deflate") for p in pairs]
        b = D.Batch(ins).run(merge)
        for i, p in enumerate(pairs):
            r = b.result(i)
            assert r["status"] == (0 if p["saved_bits"] > 0 else 1), p["stem"]
            assert r["saved_bits"] == p["saved_bits"], p["stem"]
            assert r["consumed"] == len(ins[i])
            assert b.output(i) == rd(p["stem"] + ".out.deflate"), p["stem"]
            assert b.decoded(i) == zlib.decompress(ins[i], -15)
        b.close()


@pytest.mark.parametrize("mode", ["levels", "persistent", "fused"])
def test_fixture_pairs_under_each_executor(D, monkeypatch, mode):
    """The candidate search has two executors (d4g_host.h exec_persistent): `persistent` work-queue kernels (default at
    <= 128 active blocks) and one launch per program level (`levels`, the one bench.py's 332-block stream runs).
    Both must reproduce all 30 reference pairs."""
    monkeypatch.setenv("D4G_EXEC", mode)
    for merge in (True, False):
        pairs = [p for p in MAN["pairs"] if p["merge_blocks"] == merge]
        ins = [rd(p["stem"] + ".in.deflate") for p in pairs]
        b = D.Batch(ins).run(merge)
        for i, p in enumerate(pairs):
            assert b.result(i)["saved_bits"] == p["saved_bits"], (mode, p["stem"])
            assert b.output(i) == rd(p["stem"] + ".out.deflate"), (mode, p["stem"])
        b.close()


def _oracle_many(ins, merge):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:   # ctypes releases the GIL
        return list(ex.map(lambda a: O.optimise(a, merge), ins))


@pytest.mark.parametrize("merge", [False, True])
def test_levels_executor_large_batch_vs_oracle(D, merge, monkeypatch):
    """More than 128 dynamic blocks in one batch, so without the fused executor the choice is `levels` with two stream lanes
    (round 2's launch sequencing): 28 x 1 MiB reptext members, every output byte-compared with the oracle."""
    monkeypatch.setenv("D4G_EXEC", "auto")   # the level / persistent executors, chosen by block count
    raws = [synth.reptext(1 << 20, 0x5EED + i) for i in range(28)]
    ins = [synth.deflate9(r) for r in raws]
    b = D.Batch(ins).run(merge)
    st = b.stats()
    assert st["n_blocks"] > 128 and st["search_lanes"] >= 2, st
    want = _oracle_many(ins, merge)
    for i, a in enumerate(ins):
        rc, w, saved, _, _ = want[i]
        r = b.result(i)
        assert r["status"] == rc and r["saved_bits"] == saved, i
        assert b.output(i) == (w if rc == 0 else a), i
    b.close()


def test_config2_full_size_vs_oracle(D):
    """BASELINE config 2 at full size: 64 MiB reptext (seed 0xD4F7), zlib-9 raw deflate, one stream, mode NONE,
    merge off — the exact input bench.py times — byte-compared with the oracle (about a minute of CPU)."""
    raw = synth.reptext(64 << 20, 0xD4F7)
    a = synth.deflate9(raw)
    b = D.Batch([a]).run(False)
    out = b.output(0)
    r = b.result(0)
    b.close()
    rc, want, saved, _, _ = O.optimise(a, False)
    assert rc == 0 and r["status"] == 0
    assert r["saved_bits"] == saved
    assert out == want
    assert zlib.decompress(out, -15) == raw


def test_deft_api_semantics(D):
    a = rd("asyoulik_asyoulik-zopfli.s00.in.deflate")
    out = D.Deft.optimiseDeflateStream(a)
    assert out == rd("asyoulik_asyoulik-zopfli.s00.out.deflate")
    same = rd("text.s00.in.deflate")
    assert D.Deft.optimiseDeflateStream(same) is same          # unchanged -> the ORIGINAL object (B/Deft.java:33)
    bad = b"\x07garbage"
    assert D.Deft.optimiseDeflateStream(bad) is bad            # parse failure -> original
    assert D.Deft.getSizeBitsFallback(a) == 370787
    assert D.Deft.getSizeBitsFallback(bad) == len(bad) * 8
    s = D.DeflateStream()
    assert s.parse(a) and s.getUncompressedData() == zlib.decompress(a, -15)
    assert s.optimise() == 17 and s.getSizeBits() == 370770
    assert not D.DeflateStream().parse(bad)


def test_parse_only_and_malformed(D):
    for v in MAN["parse_only"]:
        a = rd(v["stem"] + ".parse.deflate")
        s = D.DeflateStream()
        assert s.parse(a)
        assert s.getUncompressedData() == zlib.decompress(a, -15)
        assert s.consumed == len(a)
        assert D.Deft.getSizeBitsFallback(a) == O.size_bits(a)
    good = rd("deflate-dynamic.parse.deflate")
    for bad in (b"", b"\x07", b"\x01\x05\x00\x00\x00", good[:len(good) // 2], good[:3]):
        assert D.Deft.optimiseDeflateStream(bad) is bad
        assert O.optimise(bad)[0] == -1


@pytest.mark.parametrize("merge", [False, True])
def test_synthetic_vs_oracle(D, merge):
    """Seeded reptext streams at sizes the oracle finishes in seconds; ragged sizes, three zlib
    strategies (the JavaCompressor family), a stored-block stream and an empty-input stream."""
    ins = []
    for n, seed in ((1, 1), (300, 2), (5000, 3), (40000, 0xD4F7), (70000, 5)):
        ins.append(synth.make_stream(n, seed))
    t = synth.reptext(30000, 9)
    ins.append(synth.deflate9(t, zlib.Z_FILTERED))
    ins.append(synth.deflate9(t, zlib.Z_HUFFMAN_ONLY))
    c = zlib.compressobj(0, zlib.DEFLATED, -15)
    ins.append(c.compress(t[:3000]) + c.flush())                  # stored blocks
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    ins.append(c.compress(t[:20000]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(t[20000:]) + c.flush())  # empty stored block inside
    ins.append(synth.deflate9(b""))
    ins.append(synth.deflate9(bytes(range(256)) * 40))
    ins.append(synth.deflate9(b"a" * 100000))                     # len-258 runs, tiny alphabets (handleOne/handleZero)
    b = D.Batch(ins).run(merge)
    for i, a in enumerate(ins):
        rc, want, saved, consumed, _ = O.optimise(a, merge)
        r = b.result(i)
        assert r["status"] == rc, i
        assert r["saved_bits"] == saved, i
        assert b.output(i) == want, i
        assert zlib.decompress(b.output(i), -15) == zlib.decompress(a, -15)
    b.close()


def test_full_size_properties(D):
    """BASELINE config-2-shaped input (scaled to 4 MiB here; bench.py runs the 64 MiB one): size-independent
    properties — the output inflates to the same bytes, never grows, and is idempotent-or-smaller on a re-run."""
    raw = synth.reptext(4 << 20, 0xD4F7)
    a = synth.deflate9(raw)
    b = D.Batch([a]).run(False)
    out = b.output(0)
    r = b.result(0)
    b.close()
    assert zlib.decompress(out, -15) == raw
    assert r["saved_bits"] > 0 and len(out) <= len(a)
    assert O.size_bits(out) == r["size_bits_in"] - r["saved_bits"]
    b2 = D.Batch([out]).run(False)
    assert b2.result(0)["saved_bits"] >= 0
    assert zlib.decompress(b2.output(0), -15) == raw
    b2.close()


def test_batch_of_independent_members(D):
    """BASELINE config-3-shaped batch, scaled (64 x 256 KiB members instead of 1024 x 1 MiB): every stream
    round-trips, the batch result equals the one-by-one result (streams are independent,
    K/DeflateFilesContainer.java:22), and two members are checked bit for bit against the oracle."""
    raws = [synth.reptext(256 << 10, 0xD4F7 + i) for i in range(64)]
    ins = [synth.deflate9(r) for r in raws]
    b = D.Batch(ins).run(True)
    outs = [b.output(i) for i in range(len(ins))]
    saved = [b.result(i)["saved_bits"] for i in range(len(ins))]
    b.close()
    for r, o in zip(raws, outs):
        assert zlib.decompress(o, -15) == r
    for i in (0, 37):
        rc, want, sv, _, _ = O.optimise(ins[i], True)
        assert outs[i] == want and saved[i] == sv
    one = D.Batch([ins[5]]).run(True)
    assert one.output(0) == outs[5] and one.result(0)["saved_bits"] == saved[5]
    one.close()
    total, outs2 = D.DeflateFilesContainer.optimise(ins[:4], True)
    assert outs2 == outs[:4] and total == sum(saved[:4])


def test_memos_change_nothing(D, monkeypatch):
    """The per-block memos (header search by code-length set, Huffman rebuild by histogram, token pass by codes +
    mask — matched through two independent 64-bit hashes) only skip repeated work: with D4G_MEMO=0 every op
    computes, and the outputs must be identical, merge on and off."""
    ins = [synth.make_stream(3 << 20, 41), synth.make_stream(700000, 42), synth.deflate9(synth.reptext(2 << 20, 43))]
    for merge in (False, True):
        outs = []
        for memo in ("1", "0"):
            monkeypatch.setenv("D4G_MEMO", memo)
            b = D.Batch(ins).run(merge)
            outs.append([(b.result(i)["saved_bits"], b.output(i)) for i in range(len(ins))])
            b.close()
        assert outs[0] == outs[1]
        for i, a in enumerate(ins):
            assert zlib.decompress(outs[0][i][1], -15) == zlib.decompress(a, -15)


def test_fuzz_regressions(D):
    """Inputs the randomized run (scripts/gpu_fuzz.py) found: kept as fixtures, checked against the oracle."""
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in sorted(n for n in os.listdir(G) if n.startswith("fuzz_")):
        a = open(os.path.join(G, name), "rb").read()
        for merge in (False, True):
            rc, want, saved, _, _ = O.optimise(a, merge)
            b = D.Batch([a]).run(merge)
            r = b.result(0)
            assert r["status"] == rc and r["saved_bits"] == saved and b.output(0) == (want if rc == 0 else a), (name, merge)
            b.close()


def test_persistent_wait_expiry_is_recovered_by_the_level_executor(D, monkeypatch):
    """A wait of the persistent executor that gives up (D4G_SPIN_LIMIT=0: any wait whose slot is not there yet) is not an
    error: nothing is selected, the round runs again with the level executor; output == oracle, fallback counted."""
    monkeypatch.setenv("D4G_EXEC", "persistent")
    monkeypatch.setenv("D4G_SPIN_LIMIT", "-1")
    a = synth.make_stream(300000, 4)
    for merge in (False, True):
        b = D.Batch([a]).run(merge)
        rc, want, saved, _, _ = O.optimise(a, merge)
        assert b.output(0) == want and b.result(0)["saved_bits"] == saved
        assert b.stats()["persist_fallbacks"] > 0
        b.close()


def test_config5_mix_through_the_sharded_path(D):
    """BASELINE config 5 at reduced size on the real library: 14 streams of the PNG-IDAT-like / gzip-member mix
    (synth.mixed_spec, sizes divided by 32 -> 32-512 KiB) through shard.optimise_sharded with one rank, merge on (the CLI
    default); every output == the optimiser oracle's, the totals add up."""
    from deft4j_amd import shard
    items = []
    for i in range(14):
        kind, n = synth.mixed_spec(i)
        n = max(32 << 10, n // 32)
        raw = synth.pngidat(n, 0x1DA7 + i) if kind == "idat" else synth.reptext(n, 0xD4F7 + i)
        items.append((kind, raw, synth.deflate9(raw)))
    assert {k for k, _, _ in items} == {"idat", "gzip"}
    streams = [it[2] for it in items]
    total, outs, saved = shard.optimise_sharded([len(s) for s in streams], lambda i: streams[i], True, lambda ss: D.Batch(ss), batch_bytes=1 << 20)
    want = _oracle_many(streams, True)
    assert total == sum(w[2] for w in want if w[0] == 0)
    for i, (s, o, w) in enumerate(zip(streams, outs, want)):
        assert o == (w[1] if w[0] == 0 else None), (i, items[i][0], len(items[i][1]))
        assert zlib.decompress(o if o is not None else s, -15) == items[i][1]


def test_cluster_mode_on_a_merge_chain_vs_oracle(D):
    """A 4 MiB stream with merge on: its merge chain grows one block past the length at which a lone block gets the whole
    device (k_search_cluster: the control workgroup's commands worked off by every workgroup); output == the oracle's, and
    == the persistent executor's with D4G_CLUSTER=0."""
    a = synth.make_stream(4 << 20, 77)
    b = D.Batch([a]).run(True)
    st = b.stats()
    got, saved = b.output(0), b.result(0)["saved_bits"]
    b.close()
    assert st["rounds_cluster"] > 0
    rc, want, osaved, _, _ = O.optimise(a, True)
    assert rc == 0 and got == want and saved == osaved
