import os, sys, time, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["D4G_DEBUG_ZOPFLI"] = "1"
import deft4j_amd as D, synth
D.init(0)
datas = [synth.reptext(1500 + 13 * i, 1000 + i) if i % 2 else synth.pngidat(2000 + 7 * i, 50 + i, 100) for i in range(1000)]
t = time.time()
outs = D.zopfli_streams(datas, 20, 0, 15, 8 << 20)
dt = time.time() - t
print("1000 small inputs", sum(map(len, datas)), "->", sum(map(len, outs)), "%.2f s" % dt, all(zlib.decompress(o, -15) == d for o, d in zip(outs, datas)))
