#!/usr/bin/env python3
"""Run the reference's golden fixture pairs one by one on the GPU and print timing per phase (dev tool)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D  # noqa: E402

t = time.time()
D.init(0)
print("init %.2fs" % (time.time() - t), flush=True)
G = os.path.join(ROOT, "tests", "golden")
man = {p["stem"]: p for p in json.load(open(os.path.join(G, "manifest.json")))["pairs"]}
order = sys.argv[1:] or ["lz-twice-twice.s00", "deflate-store-2.s00", "text.s00", "text.s01", "text.s02", "apng_ball.s09",
                         "asyoulik_asyoulik-zopfli.s00", "asyoulik_asyoulik-gzip.s00", "284-edge-case_284.s00",
                         "nerd_nerd.s00", "nerd_nerd-extopt.s00"]
for nm in order:
    p = man[nm]
    a = open(os.path.join(G, nm + ".in.deflate"), "rb").read()
    g = open(os.path.join(G, nm + ".out.deflate"), "rb").read()
    t = time.time()
    try:
        b = D.Batch([a]).run(p["merge_blocks"])
        r = b.result(0)
        out = b.output(0)
        st = b.stats()
        print(nm, "OK" if out == g else "MISMATCH", r["saved_bits"], p["saved_bits"], "%.2fs" % (time.time() - t),
              "parse %.1f opt %.1f merge %.1f write %.1f ms; rounds %d launches %d search_ms %.1f" % (
                  st["ms_parse"], st["ms_optimise"], st["ms_merge"], st["ms_write"], st["rounds"], st["kernel_launches"],
                  st["ms_search_kernels"]), flush=True)
        b.close()
    except Exception as e:  # noqa: BLE001
        print(nm, "EXC", e, flush=True)
