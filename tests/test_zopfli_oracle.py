"""The Zopfli oracle (oracle/zopfli_oracle.c) against what pins it: the reference's own fixture
test/asyoulik/asyoulik-zopfli.txt.gz (5 iterations) and the committed vectors generated from the in-container proxy
libzopfli 1.0.3 (tests/golden/make_zopfli_golden.py).  Both log flavours: libm (as Zopfli) and the portable routine
the GPU kernels restate."""
import json
import math
import os
import zlib

import pytest

import zopf_lib as Z

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(G, "zopfli_manifest.json")))["cases"]


@pytest.mark.parametrize("case", MAN, ids=[c["name"] for c in MAN])
def test_oracle_matches_proxy_vectors(case):
    data = open(os.path.join(G, "zopfli_%s.bin" % case["name"]), "rb").read()
    want = open(os.path.join(G, "zopfli_%s.deflate" % case["name"]), "rb").read()
    split = Z.SPLIT_FIRST if case["blocksplitting"] else Z.SPLIT_NONE
    for flavor in (Z.LOG_LIBM, Z.LOG_PORTABLE):
        got = Z.deflate(data, case["iterations"], split, case["blocksplittingmax"], 1000000, flavor)
        assert got == want, "flavour %d" % flavor
    assert zlib.decompress(want, -15) == data


def test_oracle_matches_reference_fixture():
    """test/asyoulik/asyoulik-zopfli.txt.gz: payload == zopfli --i5 of the inflated text."""
    fix = open(os.path.join(G, "asyoulik_asyoulik-zopfli.s00.in.deflate"), "rb").read()
    text = zlib.decompress(fix, -15)
    assert Z.deflate(text, 5) == fix
    assert Z.deflate(text, 5, logflavor=Z.LOG_PORTABLE) == fix


def test_splitting_last_and_master_blocks_round_trip():
    """LAST (1.0.0 flow, unpinned) and deft4j's 8 MiB master block: valid streams, no larger than stored."""
    data = open(os.path.join(G, "zopfli_reptext40k.bin"), "rb").read()
    for split in (Z.SPLIT_FIRST, Z.SPLIT_LAST, Z.SPLIT_NONE):
        out = Z.deflate(data, 5, split, 15, 8 << 20)
        assert zlib.decompress(out, -15) == data
    out = Z.deflate(data, 3, Z.SPLIT_FIRST, 15, 15000)     # three master blocks, window across them
    assert zlib.decompress(out, -15) == data


def test_portable_log_accuracy():
    L = Z.lib()
    for x in list(range(1, 5000)) + [10 ** k for k in range(4, 10)] + [2 ** k + 1 for k in range(31)]:
        a, b = L.zopf_portable_log(float(x)), math.log(x)
        assert abs(a - b) <= 4e-16 * max(1.0, abs(b))


@pytest.mark.skipif(not Z.proxy_available(), reason="libzopfli proxy not in this environment")
def test_length_limited_codes_match_proxy():
    import ctypes as C
    import random
    P = C.CDLL(Z.PROXY_PATH)
    rng = random.Random(3)
    for n, maxbits in ((288, 15), (32, 15), (19, 7)):
        for _ in range(200):
            k = rng.choice((1, 2, 3, n // 2, n))
            f = [0] * n
            for i in rng.sample(range(n), k):
                f[i] = rng.choice((1, 1, 2, 3, rng.randrange(1, 50), rng.randrange(1, 100000)))
            arr = (C.c_size_t * n)(*f)
            out = (C.c_uint * n)()
            P.ZopfliLengthLimitedCodeLengths(arr, n, maxbits, out)
            assert list(out) == Z.length_limited(f, maxbits)
