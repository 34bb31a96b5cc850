#!/usr/bin/env python3
"""Dev tool (GPU box): where the time of a config-5-like batch (mixed PNG-IDAT-like and text streams, merge on) goes.
usage: config5_diag.py [streams=24]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["D4G_FUSED_STATS"] = "1"
import ctypes
import deft4j_amd as D, synth
D.init(0)
L = D.load_library()
buf = (ctypes.c_longlong * 64)()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gen = [synth.mixed_stream(i) for i in range(n)]
streams = [g[2] for g in gen]
print("streams", n, "idat", sum(1 for g in gen if g[0] == "idat"), "decoded MiB %.1f" % (sum(len(g[1]) for g in gen) / 2**20), flush=True)
for it in range(2):
    L.d4g_debug_fused_stats(buf)
    b = D.Batch(streams); t0 = time.time(); b.run(True); dt = time.time() - t0; st = b.stats(); b.close()
keys = ["n_blocks", "ms_parse", "ms_optimise", "ms_merge", "ms_write", "ms_total", "rounds_fused", "rounds_cluster", "fused_fallbacks", "persist_fallbacks", "kernel_launches", "state_launches", "ms_state_kernels"]
print("%.2f s;" % dt, {k: st[k] for k in keys if k in st}, flush=True)
L.d4g_debug_fused_stats(buf)
names = ["sweep", "apply", "binbase", "least", "tree", "hdr", "hs", "fixdot"]
tot = buf[29]; rounds = max(1, buf[27])
print("block-rounds %d, cycles per block-round %.0f" % (rounds, tot / rounds))
for k, nm in enumerate(names):
    if buf[8 + k]:
        print("%-8s tasks/round %6.1f  phases/round %5.1f  cycles/phase %8.0f  share %5.1f %%" % (nm, buf[k] / rounds, buf[8 + k] / rounds, buf[16 + k] / buf[8 + k], 100.0 * buf[16 + k] / tot))
print("advance %.1f %%  set-up %.1f %%  selection %.1f %%" % (100.0 * buf[24] / tot, 100.0 * buf[25] / tot, 100.0 * buf[26] / tot))
if buf[37]: print("tree batch: lit %.0f wait %.0f header %.0f publish %.0f; %.1f batches per round" % (buf[33] / buf[37], buf[34] / buf[37], buf[35] / buf[37], buf[36] / buf[37], buf[37] / rounds))
