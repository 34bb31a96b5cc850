"""The reference's recompress orchestration restated over the two oracles (test infrastructure only):
CompressionUtil.compress (C/CompressionUtil.java:106-182, non-threaded order) and CMDUtil.optimise's per-stream loop
(M/CMDUtil.java:70-105)."""
import oracle_lib as O
import zl9_lib as Z

CHEAP_LIST = [(Z.ZLIB, Z.DEFAULT), (Z.ZLIB, Z.FILTERED), (Z.ZLIB, Z.HUFFMAN_ONLY),
              (Z.JZLIB, Z.DEFAULT), (Z.JZLIB, Z.FILTERED), (Z.JZLIB, Z.HUFFMAN_ONLY)]   # getCompressors order, mode CHEAP


def compress(data, merge):
    """-> (best bytes, winner index)"""
    best, best_bits, win = None, None, -1
    for k, (flavor, strat) in enumerate(CHEAP_LIST):
        cand = O.deft_optimise(Z.deflate(data, strat, flavor), merge)     # CompressorTask.call / :154
        bits = O.size_bits(cand)                                           # Deft.getSizeBitsFallback
        if best is None or bits < best_bits:
            best, best_bits, win = cand, bits, k
    return best, win


def recompress(stream, merge):
    """-> dict(status, saved_bits, recompress_saved, out) as d4g_recompress_streams reports them"""
    rc, out, saved, _, _ = O.optimise(stream, merge)
    if rc < 0:
        return dict(status=-1, saved_bits=0, recompress_saved=0, out=None)
    cur = out if rc == 0 else stream
    original_size = O.size_bits(cur)
    raw, _ = O.inflate(stream)
    comp, _ = compress(raw, merge)
    rc2, out2, _, _, _ = O.optimise(comp, merge)
    if rc2 >= 0:
        cur2 = out2 if rc2 == 0 else comp
        recomp_size = O.size_bits(cur2)
        if recomp_size < original_size:
            return dict(status=0, saved_bits=saved, recompress_saved=original_size - recomp_size, out=cur2)
    return dict(status=0 if rc == 0 else 1, saved_bits=saved if rc == 0 else 0, recompress_saved=0, out=out if rc == 0 else None)
