"""The reference's recompress orchestration restated over the two oracles (test infrastructure only):
CompressionUtil.compress (C/CompressionUtil.java:106-182, non-threaded order) and CMDUtil.optimise's per-stream loop
(M/CMDUtil.java:70-105)."""
import oracle_lib as O
import zl9_lib as Z
import zopf_lib as ZF

CHEAP_LIST = [(Z.ZLIB, Z.DEFAULT), (Z.ZLIB, Z.FILTERED), (Z.ZLIB, Z.HUFFMAN_ONLY),
              (Z.JZLIB, Z.DEFAULT), (Z.JZLIB, Z.FILTERED), (Z.JZLIB, Z.HUFFMAN_ONLY)]   # getCompressors order, mode CHEAP
MODE_CHEAP, MODE_ZOPFLI, MODE_ZOPFLI_EXTENSIVE, MODE_ZOPFLI_VERY_EXTENSIVE = 1, 2, 3, 4
# (splitting, blocksplittingmax, master block): MultiJZopfliCompressor.getOptions :18-60 / MultiCafeUndZopfliCompressor.getOptions :19-25,33
JZOPFLI_OPTIONS = [(ZF.SPLIT_FIRST, 15, 1000000), (ZF.SPLIT_LAST, 15, 1000000), (ZF.SPLIT_FIRST, 0, 1000000), (ZF.SPLIT_LAST, 0, 1000000),
                   (ZF.SPLIT_NONE, 0, 1000000)]
CAFE_OPTIONS = [(ZF.SPLIT_FIRST, 15, 8 << 20), (ZF.SPLIT_LAST, 15, 8 << 20), (ZF.SPLIT_NONE, 15, 8 << 20)]


def mode_list(mode, iterations=20):
    """getCompressors order (C/CompressionUtil.java:44-78): callables data -> raw deflate stream"""
    def zl(flavor, strat):
        return lambda d: Z.deflate(d, strat, flavor)

    def zo(opt):
        return lambda d: ZF.deflate(d, max(1, iterations), opt[0], opt[1], opt[2], ZF.LOG_PORTABLE)
    extensive = mode >= MODE_ZOPFLI_EXTENSIVE
    lst = [zl(f, s) for f, s in CHEAP_LIST[:3]]
    if mode >= MODE_ZOPFLI_VERY_EXTENSIVE:
        lst += [zo(o) for o in (JZOPFLI_OPTIONS if extensive else JZOPFLI_OPTIONS[:1])]
    if mode >= MODE_ZOPFLI:
        lst += [zo(o) for o in (CAFE_OPTIONS if extensive else CAFE_OPTIONS[:1])]
    lst += [zl(f, s) for f, s in CHEAP_LIST[3:]]
    return lst


def compress(data, merge, mode=MODE_CHEAP, iterations=20):
    """-> (best bytes, winner index)"""
    best, best_bits, win = None, None, -1
    for k, enc in enumerate(mode_list(mode, iterations)):
        cand = O.deft_optimise(enc(data), merge)     # CompressorTask.call / :154
        bits = O.size_bits(cand)                                           # Deft.getSizeBitsFallback
        if best is None or bits < best_bits:
            best, best_bits, win = cand, bits, k
    return best, win


def recompress(stream, merge, mode=MODE_CHEAP, iterations=20):
    """-> dict(status, saved_bits, recompress_saved, out) as d4g_recompress_streams reports them"""
    rc, out, saved, _, _ = O.optimise(stream, merge)
    if rc < 0:
        return dict(status=-1, saved_bits=0, recompress_saved=0, out=None)
    cur = out if rc == 0 else stream
    original_size = O.size_bits(cur)
    raw, _ = O.inflate(stream)
    comp, _ = compress(raw, merge, mode, iterations)
    rc2, out2, _, _, _ = O.optimise(comp, merge)
    if rc2 >= 0:
        cur2 = out2 if rc2 == 0 else comp
        recomp_size = O.size_bits(cur2)
        if recomp_size < original_size:
            return dict(status=0, saved_bits=saved, recompress_saved=original_size - recomp_size, out=cur2)
    return dict(status=0 if rc == 0 else 1, saved_bits=saved if rc == 0 else 0, recompress_saved=0, out=out if rc == 0 else None)
