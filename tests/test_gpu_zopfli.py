"""Zopfli encoder kernels (d4g_zopfli.h) through the C ABI on the GPU: byte-identical to the committed vectors of the
in-container proxy libzopfli 1.0.3, to the reference's own asyoulik-zopfli fixture, and to the oracle
(oracle/zopfli_oracle.c, portable log flavour) for the options deft4j uses that the proxy does not have."""
import json
import os
import zlib

import pytest

import deft4j_amd as D
import synth
import zopf_lib as Z

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(G, "zopfli_manifest.json")))["cases"]


@pytest.fixture(scope="module", autouse=True)
def _init():
    D.init(0)


def test_proxy_vectors_in_one_batch_per_option_set():
    """every committed libzopfli vector, grouped by option set so that several inputs share a launch"""
    groups = {}
    for c in MAN:
        groups.setdefault((c["iterations"], c["blocksplitting"], c["blocksplittingmax"]), []).append(c)
    for (it, bs, mb), cases in groups.items():
        datas = [open(os.path.join(G, "zopfli_%s.bin" % c["name"]), "rb").read() for c in cases]
        outs = D.zopfli_streams(datas, it, D.ZOPFLI_SPLIT_FIRST if bs else D.ZOPFLI_SPLIT_NONE, mb, 1000000)
        for c, d, o in zip(cases, datas, outs):
            want = open(os.path.join(G, "zopfli_%s.deflate" % c["name"]), "rb").read()
            assert o == want, c["name"]
            assert zlib.decompress(o, -15) == d


def test_reference_fixture_asyoulik_zopfli():
    """test/asyoulik/asyoulik-zopfli.txt.gz: its payload is what 5 iterations make of the inflated text"""
    fix = open(os.path.join(G, "asyoulik_asyoulik-zopfli.s00.in.deflate"), "rb").read()
    text = zlib.decompress(fix, -15)
    assert D.zopfli_streams([text], 5, D.ZOPFLI_SPLIT_FIRST, 15, 1000000)[0] == fix


@pytest.mark.parametrize("split", [D.ZOPFLI_SPLIT_FIRST, D.ZOPFLI_SPLIT_LAST, D.ZOPFLI_SPLIT_NONE])
def test_deft4j_option_sets_against_the_oracle(split):
    """CafeUndZopfli as deft4j calls it (8 MiB master block, FIRST / LAST / NONE, iter 20) and jzopfli's unlimited split"""
    datas = [synth.reptext(60000, 11), synth.pngidat(50000, 3, 200), bytes(40000), synth.reptext(900, 9), b"", b"a"]
    outs = D.zopfli_streams(datas, 20, split, 15, 8 << 20)
    for d, o in zip(datas, outs):
        assert o == Z.deflate(d, 20, split, 15, 8 << 20, Z.LOG_PORTABLE)
        assert zlib.decompress(o, -15) == d
    outs = D.zopfli_streams(datas[:2], 6, split, 0, 8 << 20)
    for d, o in zip(datas[:2], outs):
        assert o == Z.deflate(d, 6, split, 0, 8 << 20, Z.LOG_PORTABLE)


def test_master_blocks_share_the_window():
    data = synth.reptext(250000, 21)
    out = D.zopfli_streams([data], 5, D.ZOPFLI_SPLIT_FIRST, 15, 100000)[0]      # three master blocks
    assert out == Z.deflate(data, 5, Z.SPLIT_FIRST, 15, 100000, Z.LOG_PORTABLE)
    assert zlib.decompress(out, -15) == data


def test_long_runs_and_block_ends_inside_runs():
    import random
    rng = random.Random(77)
    data = b"".join(bytes([rng.randrange(4)]) * rng.randrange(1, 3000) for _ in range(120))
    for master in (8 << 20, 30000):
        out = D.zopfli_streams([data], 8, D.ZOPFLI_SPLIT_FIRST, 15, master)[0]
        assert out == Z.deflate(data, 8, Z.SPLIT_FIRST, 15, master, Z.LOG_PORTABLE)


@pytest.mark.parametrize("mode", [D.MODE_ZOPFLI, D.MODE_ZOPFLI_EXTENSIVE, D.MODE_ZOPFLI_VERY_EXTENSIVE])
def test_recompress_modes_with_zopfli_against_the_composed_oracles(mode):
    """CompressionUtil.compress for the modes that add the Zopfli compressors (C/CompressionUtil.java:44-78 list order,
    strict minimum by parsed bit size :144-168): same bytes and the same winning list index as the oracles composed."""
    import oracle_compose as OC
    fix = open(os.path.join(G, "asyoulik_asyoulik-gzip.s00.in.deflate"), "rb").read()
    text = zlib.decompress(fix, -15)
    datas = [text[:30000], synth.reptext(20000, 31), synth.pngidat(20000, 3, 200), b""]
    for merge in (True, False):
        cu = D.CompressionUtil(mode, 5, merge)
        outs = cu.compress_many(datas)
        for d, o, w in zip(datas, outs, cu.last_winner):
            assert (o, w) == OC.compress(d, merge, mode, 5), (mode, merge, len(d))
            assert zlib.decompress(o, -15) == d


def test_recompress_graft_loop_in_mode_zopfli():
    """CMDUtil.optimise's recompress-compare-graft loop (M/CMDUtil.java:76-105) with mode ZOPFLI"""
    import oracle_compose as OC
    raw = synth.reptext(40000, 77)
    c1 = zlib.compressobj(1, zlib.DEFLATED, -15)
    streams = [c1.compress(raw) + c1.flush(), synth.deflate9(raw[:15000]), b"\x07junk"]
    res = D.recompress_streams(streams, D.MODE_ZOPFLI, True, 5)
    for a, r in zip(streams, res):
        assert r == OC.recompress(a, True, D.MODE_ZOPFLI, 5), len(a)
    assert res[0]["recompress_saved"] > 0


def test_batch_of_members_through_mode_zopfli_extensive():
    """eight members in one d4g_compress call: the Zopfli stage (its own host thread) and the zlib-family stages run side by
    side on the device; every member's winner and bytes equal the composed oracles'"""
    import oracle_compose as OC
    datas = [synth.reptext(48000, 500 + i) if i % 2 == 0 else synth.pngidat(40000, 700 + i, 200) for i in range(8)]
    cu = D.CompressionUtil(D.MODE_ZOPFLI_EXTENSIVE, 4, True)
    outs = cu.compress_many(datas)
    for d, o, w in zip(datas, outs, cu.last_winner):
        assert (o, w) == OC.compress(d, True, D.MODE_ZOPFLI_EXTENSIVE, 4), len(d)


def test_change_point_pool_grows_instead_of_failing(monkeypatch):
    """a valid input whose positions have many record-setting matches (groups S[:200], S[:199], ... S[:3], S)
    ran the change-point pool out and failed the whole batch.  The match kernels now run again with a larger pool: the
    246 KB input of the report round-trips at the pool's natural size, a small one equals the oracle with the pool forced tiny."""
    S = synth.reptext(200, 1)
    group = b"".join(S[:k] for k in range(200, 2, -1)) + S
    big = group * 12
    assert len(big) > 240000
    out = D.zopfli_streams([big, S], 2, Z.SPLIT_FIRST, 15, 8 << 20)
    assert zlib.decompress(out[0], -15) == big and zlib.decompress(out[1], -15) == S
    monkeypatch.setenv("D4G_ZF_POOL_WORDS", "256")
    small = group[:6000]
    for split in (Z.SPLIT_FIRST, Z.SPLIT_LAST):
        assert D.zopfli_streams([small], 2, split, 15, 8 << 20)[0] == Z.deflate(small, 2, split, 15, 8 << 20, Z.LOG_PORTABLE)
    # a whole recompress call survives as well (the candidate list is complete)
    cu = D.CompressionUtil(D.MODE_ZOPFLI, 1, True)
    o = cu.compress_many([small])[0]
    assert zlib.decompress(o, -15) == small
