#!/usr/bin/env python3
"""Dev tool: run one raw deflate file through the library (GPU, or the emulator with --sim) against the oracle."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, oracle_lib as O
a = open(sys.argv[1], "rb").read()
L = None
if "--sim" in sys.argv:
    os.environ.setdefault("D4G_SIM_BLOCK", "64")
    L = D.load_library(os.path.join(ROOT, "tests", "hostsim", "libdeft4g_hostsim.so"))
D.init(0, lib=L)
for merge in (False, True):
    rc, want, saved, _, _ = O.optimise(a, merge)
    for memo in ("1", "0"):
        os.environ["D4G_MEMO"] = memo
        b = D.Batch([a], lib=L).run(merge)
        r = b.result(0); out = b.output(0); b.close()
        same = out == (want if rc == 0 else a)
        print("merge %s memo %s: status %d/%d saved %d/%d len %d/%d %s" % (merge, memo, r["status"], rc, r["saved_bits"], saved, len(out), len(want), "OK" if same else "DIFF"))
        if not same:
            bi_g, bi_o = O.block_info(out), O.block_info(want)
            if bi_g is None or bi_o is None:
                print("  output does not parse:", bi_g is None, bi_o is None)
                try:
                    print("  inflates to the same bytes:", zlib.decompress(out, -15) == zlib.decompress(a, -15))
                except Exception as e:
                    print("  zlib:", e)
                continue
            print("  blocks gpu/oracle:", len(bi_g), len(bi_o))
            for k, (x, y) in enumerate(zip(bi_g, bi_o)):
                if x != y:
                    print("  first differing block", k, "gpu", x, "oracle", y)
                    break
