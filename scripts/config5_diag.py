#!/usr/bin/env python3
"""Dev tool (GPU box): where the time of a config-5-like batch (mixed PNG-IDAT-like and text streams, merge on) goes.
usage: config5_diag.py [streams=24]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, synth
D.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gen = [synth.mixed_stream(i) for i in range(n)]
streams = [g[2] for g in gen]
print("streams", n, "idat", sum(1 for g in gen if g[0] == "idat"), "decoded MiB %.1f" % (sum(len(g[1]) for g in gen) / 2**20), flush=True)
for it in range(2):
    b = D.Batch(streams); t0 = time.time(); b.run(True); dt = time.time() - t0; st = b.stats(); b.close()
keys = ["n_blocks", "ms_parse", "ms_optimise", "ms_merge", "ms_write", "ms_total", "rounds_fused", "rounds_cluster", "fused_fallbacks", "persist_fallbacks", "kernel_launches", "state_launches", "ms_state_kernels"]
print("%.2f s;" % dt, {k: st[k] for k in keys if k in st}, flush=True)
