"""The LZ77 / zlib-level-9 oracle (oracle/zlib9_oracle.c) against what pins it: the reference-held pair
test/asyoulik/asyoulik-gzip.txt.gz (its payload is the JavaCompressor output of its own inflated text), the committed
golden vectors (Python zlib 1.2.11, tests/golden/make_lz_golden.py) and live Python zlib on seeded inputs."""
import json
import os
import random
import zlib

import pytest

import synth
import zl9_lib as Z

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRAT = {"default": (Z.DEFAULT, zlib.Z_DEFAULT_STRATEGY), "filtered": (Z.FILTERED, zlib.Z_FILTERED), "huffman": (Z.HUFFMAN_ONLY, zlib.Z_HUFFMAN_ONLY)}


def rd(n):
    return open(os.path.join(G, n), "rb").read()


def zref(d, zs):
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zs)
    return c.compress(d) + c.flush()


def test_reference_fixture_pair():
    fix = rd("asyoulik_asyoulik-gzip.s00.in.deflate")          # payload of test/asyoulik/asyoulik-gzip.txt.gz
    text = zlib.decompress(fix, -15)
    assert Z.deflate(text, Z.DEFAULT, Z.ZLIB) == fix


def test_golden_vectors():
    man = json.load(open(os.path.join(G, "lz_manifest.json")))
    for c in man["cases"]:
        data = rd("lz_%s.bin" % c["name"])
        for sname, (st, _) in STRAT.items():
            assert Z.deflate(data, st, Z.ZLIB) == rd("lz_%s.%s.deflate" % (c["name"], sname)), (c["name"], sname)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_seeded_inputs_against_live_zlib(seed):
    rng = random.Random(seed)
    cases = [synth.reptext(rng.randint(1, 300000), seed), os.urandom(rng.randint(1, 50000)) + synth.reptext(rng.randint(1, 90000), seed + 10),
             bytes(rng.choice(b"ab") for _ in range(rng.randint(1, 100000))), b"q" * rng.randint(1, 200000),
             synth.reptext(16383 * 3, seed)[:rng.randint(16000, 49149)]]
    for d in cases:
        for st, zs in STRAT.values():
            assert Z.deflate(d, st, Z.ZLIB) == zref(d, zs)


def test_match_table_is_what_the_parse_consumes():
    """zl9_match_table (the per-position search result the GPU's wave-wide search computes) agrees with the tokens
    the sequential encoder emits: every match token's (length, distance) is the table entry of its position, unless the
    pending-match rules (lazy evaluation, TOO_FAR) replaced it — checked here only for consistency of lengths."""
    d = synth.reptext(60000, 4)
    l4, d4, l1, d1 = Z.match_table(d)
    toks, ends, _ = Z.tokens(d, Z.DEFAULT, Z.ZLIB)
    pos = 0
    for t in toks.tolist():
        dist = t >> 9
        if dist:
            ln = t & 511
            assert ln <= int(l4[pos]) and ln >= 3
            if ln == int(l4[pos]):
                assert dist == int(d4[pos]) or int(l1[pos]) == ln
            pos += ln
        else:
            pos += 1
    assert pos == len(d)
    assert int(ends[-1]) == len(toks)


def test_jzlib_flavour_round_trips_and_splits_early():
    """flavour 1 (jzlib 1.1.x: zlib's algorithm + the TRUNCATE_BLOCK early flush; PARITY UNPINNED) produces valid
    streams of the same tokens, with more blocks on text."""
    d = synth.reptext(300000, 6)
    for st in (Z.DEFAULT, Z.FILTERED, Z.HUFFMAN_ONLY):
        out = Z.deflate(d, st, Z.JZLIB)
        assert zlib.decompress(out, -15) == d
    t0, e0, _ = Z.tokens(d, Z.DEFAULT, Z.ZLIB)
    t1, e1, _ = Z.tokens(d, Z.DEFAULT, Z.JZLIB)
    assert (t0 == t1).all() and len(e1) > len(e0)
