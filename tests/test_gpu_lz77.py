"""Parity of the LZ77 encoder kernels on the GPU, through the C ABI: byte-identical to the
reference-held JavaCompressor pair (asyoulik), to the committed zlib-1.2.11 golden vectors, to live Python zlib and to the
oracle (oracle/zlib9_oracle.c) on seeded inputs — and, with optimise on, identical to running the optimiser on the
encoder's serialised output (the reference's compress -> Deft.optimiseDeflateStream sequence, C/CompressorTask.java:29-35)."""
import json
import os
import random
import zlib

import pytest

import oracle_lib as O
import synth
import zl9_lib as Z

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRATS = ((0, zlib.Z_DEFAULT_STRATEGY, "default"), (1, zlib.Z_FILTERED, "filtered"), (2, zlib.Z_HUFFMAN_ONLY, "huffman"))


@pytest.fixture(scope="module")
def D():
    import deft4j_amd
    deft4j_amd.init(0)
    return deft4j_amd


def rd(n):
    return open(os.path.join(G, n), "rb").read()


def zref(d, zs):
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zs)
    return c.compress(d) + c.flush()


def test_reference_fixture_pair(D):
    fix = rd("asyoulik_asyoulik-gzip.s00.in.deflate")          # payload of test/asyoulik/asyoulik-gzip.txt.gz
    text = zlib.decompress(fix, -15)
    assert D.deflate_streams([text], D.ENC_JVM, D.STRATEGY_DEFAULT)[0] == fix


def test_golden_vectors(D):
    man = json.load(open(os.path.join(G, "lz_manifest.json")))
    datas = [rd("lz_%s.bin" % c["name"]) for c in man["cases"]]
    for st, _, sname in STRATS:
        outs = D.deflate_streams(datas, D.ENC_JVM, st)          # all cases in one device batch
        for c, o in zip(man["cases"], outs):
            assert o == rd("lz_%s.%s.deflate" % (c["name"], sname)), (c["name"], sname)


def test_one_batch_many_specs_vs_zlib_and_oracle(D):
    """Ragged inputs x three strategies in one batch (the hash sort of an input is shared by its specs)."""
    rng = random.Random(11)
    ins = [synth.reptext(n, 20 + i) for i, n in enumerate((1, 2, 3, 257, 2047, 2048, 2049, 32767, 32768, 32769, 65274, 65275, 65536, 200000, 1 << 20))]
    ins += [b"", b"z" * 300000, bytes(rng.choice(b"ACGT") for _ in range(150000)), os.urandom(100000),
            os.urandom(40000) + synth.reptext(90000, 3), bytes(rng.choice(b"ab") for _ in range(120000)),
            synth.reptext(16383 * 4, 8)[:16383 * 3 + 5]]
    specs = [(i, D.ENC_JVM, st) for i in range(len(ins)) for st, _, _ in STRATS]
    b = D.EncodeBatch(ins, specs).run(False)
    for k, (i, _, st) in enumerate(specs):
        want = zref(ins[i], STRATS[st][1])
        assert b.output(k) == want, (i, len(ins[i]), st)
        assert b.result(k)["size_bits_in"] == O.size_bits(want)
        if i % 4 == 0:
            assert Z.deflate(ins[i], st, Z.ZLIB) == want
    b.close()


def test_exact_block_fill_edges(D):
    """A block that the very last symbol fills: the literal deflate_slow's epilogue emits closes it as the last block,
    a match (or deflate_huff's last literal) flushes it as a non-last block and an empty last block follows."""
    lits = os.urandom(16383)
    two = os.urandom(16382) + b"\x00" * 300
    for d in (lits, lits + lits, two, os.urandom(16383 * 2 - 1) + b"abcabcabcabc"):
        for st, zs, _ in STRATS:
            assert D.deflate_streams([d], D.ENC_JVM, st)[0] == zref(d, zs), (len(d), st)


def test_config_sized_stream(D):
    """16 MiB of the config-2 generator: byte-identical to zlib level 9 (three strategies), and the parse converges in
    a few passes."""
    raw = synth.reptext(16 << 20, 0xD4F7)
    b = D.EncodeBatch([raw], [(0, D.ENC_JVM, st) for st, _, _ in STRATS]).run(False)
    st = b.stats()
    for k, (_, zs, _) in enumerate(STRATS):
        assert b.output(k) == zref(raw, zs), k
    assert st["lz_parse_passes"] <= 8, st
    b.close()


@pytest.mark.parametrize("merge", [False, True])
def test_encode_then_optimise_equals_optimise_of_the_encoded_stream(D, merge):
    """run_encode(optimise=1) hands the encoder's tokens and states straight to the candidate search.  The result must
    be what the reference computes: Deft.optimiseDeflateStream(compressor.compress(data)) — checked against the oracle
    run on zlib's bytes, and against the GPU's own parse-then-optimise path."""
    ins = [synth.reptext(n, 40 + i) for i, n in enumerate((5000, 70000, 300000))] + [os.urandom(30000) + synth.reptext(50000, 2), b"k" * 50000, b""]
    specs = [(i, D.ENC_JVM, st) for i in range(len(ins)) for st, _, _ in STRATS]
    b = D.EncodeBatch(ins, specs).run(True, merge)
    for k, (i, _, st) in enumerate(specs):
        enc = zref(ins[i], STRATS[st][1])
        rc, want, saved, _, _ = O.optimise(enc, merge)
        r = b.result(k)
        assert r["status"] == rc and r["saved_bits"] == saved, (i, st)
        assert b.output(k) == (want if rc == 0 else enc), (i, st)
        assert r["size_bits_in"] == O.size_bits(enc)
    b.close()


def test_jzlib_flavour_vs_oracle(D):
    """The jzlib compressor family (early block flush at 8192 symbols; PARITY UNPINNED — no jzlib here): byte-identical
    to the oracle's restatement and valid streams."""
    ins = [synth.reptext(n, 60 + i) for i, n in enumerate((100, 70000, 400000, 1 << 20))] + [os.urandom(50000), b"", synth.reptext(16383 * 2, 3)]
    specs = [(i, D.ENC_JZLIB, st) for i in range(len(ins)) for st, _, _ in STRATS]
    b = D.EncodeBatch(ins, specs).run(False)
    for k, (i, _, st) in enumerate(specs):
        out = b.output(k)
        assert out == Z.deflate(ins[i], st, Z.JZLIB), (i, st)
        assert zlib.decompress(out, -15) == ins[i]
    b.close()


@pytest.mark.parametrize("merge", [True, False])
def test_mode_cheap_on_the_asyoulik_text(D, merge):
    """CompressionUtil.compress, mode CHEAP (six compressors, each output optimised, strict minimum in list order —
    C/CompressionUtil.java:44-78,144-168) on the reference's own text, and CMDUtil's recompress-compare-graft loop
    (M/CMDUtil.java:70-105) on the reference's fixture streams, against the same orchestration over the oracles."""
    import oracle_compose as OC
    text = zlib.decompress(rd("asyoulik_asyoulik-gzip.s00.in.deflate"), -15)
    cu = D.CompressionUtil(D.MODE_CHEAP, mergeBlocks=merge)
    out = cu.compress(text)
    want, win = OC.compress(text, merge)
    assert out == want and cu.last_winner == [win]
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15)
    streams = [rd("asyoulik_asyoulik-gzip.s00.in.deflate"), rd("asyoulik_asyoulik-zopfli.s00.in.deflate"), c1.compress(text) + c1.flush(),
               b"\x07junk", rd("lz-twice-twice.s00.in.deflate")]
    res = D.recompress_streams(streams, D.MODE_CHEAP, merge)
    for a, r in zip(streams, res):
        assert r == OC.recompress(a, merge), len(a)
    assert res[2]["recompress_saved"] > 0 and res[1]["recompress_saved"] == 0      # zlib-1 loses to zlib-9; zopfli's stream stays


def test_mode_cheap_batch_of_members(D):
    """Config-3-shaped batch, scaled (24 x 256 KiB gzip members, mode CHEAP) through the resident-batch entry point:
    every final stream equals the oracle orchestration's, and round-trips."""
    import oracle_compose as OC
    from concurrent.futures import ThreadPoolExecutor
    raws = [synth.reptext(256 << 10, 0xD4F7 + i) for i in range(24)]
    c6 = [zlib.compressobj(6, zlib.DEFLATED, -15) for _ in raws]
    ins = [synth.deflate9(r) if i % 2 == 0 else c.compress(r) + c.flush() for i, (r, c) in enumerate(zip(raws, c6))]
    b = D.Batch(ins).run_recompress(D.MODE_CHEAP, True)
    with ThreadPoolExecutor(max_workers=16) as ex:
        want = list(ex.map(lambda a: OC.recompress(a, True), ins))
    ngraft = 0
    for i, a in enumerate(ins):
        r = b.result(i)
        g, rs = b.recompress_result(i)
        w = want[i]
        assert r["status"] == w["status"] and r["saved_bits"] == w["saved_bits"] and rs == w["recompress_saved"], i
        final = b.output(i) if r["status"] == 0 else a
        assert final == (w["out"] if w["status"] == 0 else a), i
        assert zlib.decompress(final, -15) == raws[i]
        ngraft += g
    assert ngraft >= 12          # every zlib-6 member is beaten by its recompression
    b.close()


def test_gzip_file_through_mode_cheap(D):
    """`deft4j optimise --mode=CHEAP` on a gzip file (M/CMDUtil.java:57-116 over K/GZFile.java): a weakly compressed
    member is recompressed, grafted, and the container re-written with a recomputed trailer; the transcript carries the
    reference's recompression lines."""
    import gzip
    import oracle_compose as OC
    from deft4j_amd import containers
    text = zlib.decompress(rd("asyoulik_asyoulik-gzip.s00.in.deflate"), -15)
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15)
    payload = c1.compress(text) + c1.flush()
    gz = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\x03" + payload + (zlib.crc32(text) & 0xffffffff).to_bytes(4, "little") + (len(text) & 0xffffffff).to_bytes(4, "little")
    (out, lines), = containers.optimise_files([gz], True, mode=D.MODE_CHEAP)
    assert gzip.decompress(out) == text
    want = OC.recompress(payload, True)
    assert want["recompress_saved"] > 0
    assert out[10:-8] == want["out"]
    assert any(l.startswith("Recompressed stream 0") for l in lines) and lines[-1] == "Saved %d bits with recompression" % want["recompress_saved"]
