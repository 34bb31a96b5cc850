#!/usr/bin/env python3
"""Dev tool: BASELINE config-3-shaped batch (N x 1 MiB independent members) timing + round-trip check."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
merge = int(sys.argv[2]) if len(sys.argv) > 2 else 0
D.init(0)
t = time.time()
raws = [synth.reptext(1 << 20, 0xD4F7 + i) for i in range(n)]
ins = [synth.deflate9(r) for r in raws]
print("generated %d streams in %.1fs" % (n, time.time() - t), flush=True)
for it in range(2):
    b = D.Batch(ins)
    t = time.time()
    b.run(bool(merge))
    dt = time.time() - t
    st = b.stats()
    print("run %d: %.3f s -> %.1f MB/s; blocks %d tokens %d rounds %d launches %d; parse %.0f opt %.0f merge %.0f write %.0f ms" % (
        it, dt, n * (1 << 20) / 1e6 / dt, st["n_blocks"], st["n_tokens"], st["rounds"], st["kernel_launches"],
        st["ms_parse"], st["ms_optimise"], st["ms_merge"], st["ms_write"]), flush=True)
    if it == 1:
        bad = sum(zlib.decompress(b.output(i), -15) != raws[i] for i in range(0, n, max(1, n // 64)))
        print("roundtrip mismatches (sampled):", bad, "saved bits total:", sum(b.result(i)["saved_bits"] for i in range(n)))
    b.close()
