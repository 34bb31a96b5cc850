"""Pins the CPU oracle against every fixture the reference commits for this path
(runTestOpt.sh:3-11 pairs)."""
import hashlib
import json
import os
import zlib

import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(G, "manifest.json")))


def rd(name):
    return open(os.path.join(G, name), "rb").read()


@pytest.mark.parametrize("pair", MAN["pairs"], ids=[p["stem"] for p in MAN["pairs"]])
def test_fixture_pair_bit_exact(pair):
    a = rd(pair["stem"] + ".in.deflate")
    g = rd(pair["stem"] + ".out.deflate")
    assert hashlib.sha256(a).hexdigest() == pair["in_sha256"]
    rc, res, saved, consumed, _ = O.optimise(a, pair["merge_blocks"])
    assert rc in (0, 1)
    assert consumed == len(a)
    assert saved == pair["saved_bits"]          # the transcript's "N bits saved in stream k"
    assert (rc == 0) == (saved > 0)
    assert res == g                              # byte-identical to the Java output
    assert hashlib.sha256(res).hexdigest() == pair["out_sha256"]
    # Deft.optimiseDeflateStream returns the ORIGINAL array when nothing was saved (B/Deft.java:33)
    assert O.deft_optimise(a, pair["merge_blocks"]) == (g if saved > 0 else a)


def test_file_totals_match_transcripts():
    tot = {}
    for p in MAN["pairs"]:
        tot.setdefault(p["source"], [0, p["file_total_saved"]])[0] += p["saved_bits"]
    for src, (s, t) in tot.items():
        assert s == t, src
    assert tot["apng/ball.png"][1] == 1298 and tot["nerd/nerd.png"][1] == 37303


# bit sizes and block structure of the parse-only vectors
PARSE_FACTS = {"ban": (55, [1]), "deflate-dynamic": (1047, [2]), "deflate-fixed": (530, [1]),
               "deflate-store": (664, [0]), "lz": (184, [1])}


@pytest.mark.parametrize("v", MAN["parse_only"], ids=[v["stem"] for v in MAN["parse_only"]])
def test_parse_only_vectors(v):
    a = rd(v["stem"] + ".parse.deflate")
    data, consumed = O.inflate(a)
    assert data == zlib.decompress(a, -15)
    assert consumed == len(a)
    if v["stem"] in PARSE_FACTS:
        bits, types = PARSE_FACTS[v["stem"]]
        assert O.size_bits(a) == bits
        assert [b[0] for b in O.block_info(a)] == types


def test_block_structure_facts():
    info = O.block_info(rd("asyoulik_asyoulik-gzip.s00.in.deflate"))
    assert [(b[0], b[1], b[3]) for b in info] == [(2, 16384, 575), (2, 11020, 541)]
    assert O.size_bits(rd("asyoulik_asyoulik-gzip.s00.in.deflate")) == 390170
    assert O.size_bits(rd("asyoulik_asyoulik-gzip.s00.out.deflate")) == 390003
    info = O.block_info(rd("asyoulik_asyoulik-gzip.s00.out.deflate"))
    assert [(b[0], b[1], b[3]) for b in info] == [(2, 29585, 615)]
    assert [b[1] for b in O.block_info(rd("asyoulik_asyoulik-zopfli.s00.out.deflate"))] == [553, 850, 2970, 2296, 20832]
    assert O.size_bits(rd("nerd_nerd.s00.in.deflate")) == 3483115
    assert O.size_bits(rd("nerd_nerd.s00.out.deflate")) == 3445812
    assert O.size_bits(rd("284-edge-case_284.s00.in.deflate")) == 651652


def _freq(n, d):
    f = [0] * n
    for k, v in d.items():
        f[k] = v
    return f


def test_huffman_tree_kats():
    """JDK PriorityQueue tie-breaking (FIFO-on-ties heaps fail KAT 1)."""
    f1 = {32: 18, 46: 2, 50: 1, 84: 2, 97: 5, 98: 1, 99: 3, 100: 2, 101: 8, 103: 1, 104: 4, 105: 7, 107: 2, 108: 3,
          109: 1, 110: 5, 111: 5, 112: 2, 114: 3, 115: 9, 116: 9, 117: 2, 120: 2, 121: 1, 256: 1}
    e1 = {32: 3, 46: 6, 50: 6, 84: 6, 97: 4, 98: 6, 99: 5, 100: 6, 101: 4, 103: 6, 104: 5, 105: 4, 107: 6, 108: 5,
          109: 7, 110: 4, 111: 4, 112: 6, 114: 5, 115: 3, 116: 3, 117: 6, 120: 6, 121: 6, 256: 7}
    lens, _ = O.huffman_lengths(_freq(257, f1), 15)
    assert lens == _freq(257, e1)
    lens, _ = O.huffman_lengths(_freq(19, {0: 9, 3: 3, 4: 5, 5: 4, 6: 11, 7: 2, 18: 5}), 7)
    assert lens == _freq(19, {0: 2, 3: 4, 4: 3, 5: 3, 6: 2, 7: 4, 18: 3})
    lens, _ = O.huffman_lengths(_freq(19, {0: 15, 2: 3, 3: 3, 4: 6, 5: 11, 6: 10, 18: 5}), 7)
    assert lens == _freq(19, {0: 2, 2: 4, 3: 4, 4: 3, 5: 2, 6: 3, 18: 3})


def test_huffman_tree_dummy_leaves_and_limit():
    # one used symbol -> dummy leaf at the first zero-frequency index (HuffmanTree.java:50-58)
    lens, codes = O.huffman_lengths([0, 0, 5], 15)
    assert lens == [1, 0, 1]
    lens, _ = O.huffman_lengths([7, 0, 0], 15)
    assert lens == [1, 1, 0]
    # Fibonacci weights force depth > limit; the limiter must return a complete code within the limit
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    for limit in (7, 15):
        lens, _ = O.huffman_lengths(fib[:12] if limit == 7 else fib, limit)
        assert max(lens) <= limit
        assert sum(2 ** (limit - l) for l in lens if l) == 2 ** limit


def test_pack_rle():
    """HuffmanTable.pack (B/huffman/HuffmanTable.java:70-159), SURVEY Appendix A.4."""
    assert O.pack([0] * 138 + [5]) == [18, 127, 5]
    assert O.pack([0] * 150 + [5]) == [18, 127, 18, 1, 5]
    assert O.pack([3] * 9 + [0]) == [3, 16, 1, 16, 1, 0]                      # 1 + (4+4)
    assert O.pack([3] * 9 + [0], use8=False) == [3, 16, 3, 3, 3, 0]           # 1 + 6 + 2 literals
    assert O.pack([3] * 8 + [0]) == [3, 16, 1, 16, 0, 0]                      # 1 + (4+3)
    assert O.pack([3] * 8 + [0], ohh=False) == [3, 16, 3, 3, 0]
    assert O.pack([3] * 15 + [0]) == [3, 16, 3, 16, 1, 16, 1, 0]              # 1 + 6 + (4+4)
    assert O.pack([0] * 5 + [2], no_zrep=True) == [0, 16, 1, 2]               # zeros via 16 when 17 is off
    assert O.pack([0] * 5 + [2], no_zrep=True, no_rep_zeros=True) == [0, 0, 0, 0, 0, 2]
    assert O.pack([4] * 5, no_rep=True) == [4, 4, 4, 4, 4]
    assert O.pack([0] * 12 + [1], no_zrep2=True) == [17, 7, 0, 0, 1]


def test_malformed_streams_fail_like_the_reference():
    assert O.optimise(b"")[0] == -1
    assert O.optimise(b"\x07")[0] == -1                      # BTYPE 3 (DeflateStream.java:103-105)
    assert O.optimise(b"\x01\x05\x00\x00\x00")[0] == -1       # stored NLEN mismatch
    good = rd("deflate-dynamic.parse.deflate")
    assert O.optimise(good[:len(good) // 2])[0] == -1         # truncated dynamic block
    a = rd("lz.parse.deflate")
    assert O.deft_optimise(a[:3]) == a[:3]                    # failure -> original returned (B/Deft.java:33)
