"""Stage times of the Zopfli encoder on the GPU (D4G_DEBUG_ZOPFLI=1 prints them per call)."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("D4G_DEBUG_ZOPFLI", "1")
import deft4j_amd as D
import synth

D.init(0)
cases = sys.argv[1:] or ["rep1m", "png1m", "rep8m_none", "batch64"]
for c in cases:
    if c == "rep1m":
        datas, args = [synth.reptext(1 << 20, 3)], (20, 0, 15, 8 << 20)
    elif c == "png1m":
        datas, args = [synth.pngidat(1 << 20, 3)], (20, 0, 15, 8 << 20)
    elif c == "rep8m_none":
        datas, args = [synth.reptext(8 << 20, 3)], (20, 2, 15, 8 << 20)
    elif c == "rep8m_first":
        datas, args = [synth.reptext(8 << 20, 3)], (20, 0, 15, 8 << 20)
    elif c == "batch64":
        datas, args = [synth.reptext(1 << 20, 100 + i) for i in range(64)], (20, 0, 15, 8 << 20)
    else:
        continue
    t = time.time()
    outs = D.zopfli_streams(datas, *args)
    dt = time.time() - t
    ok = all(zlib.decompress(o, -15) == d for o, d in zip(outs, datas))
    print(c, "bytes", sum(map(len, datas)), "->", sum(map(len, outs)), "%.2f s" % dt, "%.2f MB/s" % (sum(map(len, datas)) / dt / 1e6), "roundtrip", ok, flush=True)
