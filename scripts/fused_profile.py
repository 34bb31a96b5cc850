#!/usr/bin/env python3
"""Dev tool (GPU box): the fused executor's per-phase accounting on the config-2 generator.
usage: fused_profile.py [MiB=64] [merge=0]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["D4G_FUSED_STATS"] = "1"
import deft4j_amd as D, synth
D.init(0)
L = D.load_library()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
merge = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
s = synth.make_stream(mib << 20)
buf = (ctypes.c_longlong * 64)()
for it in range(2):
    L.d4g_debug_fused_stats(buf)
    b = D.Batch([s]).run(merge); st = b.stats(); b.close()
L.d4g_debug_fused_stats(buf)
names = ["sweep", "apply", "binbase", "least", "tree", "hdr", "hs", "fixdot"]
rounds = max(1, buf[27])
print("ms_optimise %.2f search_kernels %.2f; %d block-rounds, %d blocks, %.1f steps per round" % (st["ms_optimise"], st["ms_search_kernels"], buf[27], st["n_blocks"], buf[28] / rounds))
tot = buf[29]
print("cycles per block-round %.0f (%.1f us @2.4GHz); per block %.0f" % (tot / rounds, tot / rounds / 2400.0, tot / max(1, st["n_blocks"])))
for k, nm in enumerate(names):
    if buf[8 + k]:
        print("%-8s tasks/round %6.1f  phases/round %5.1f  cycles/phase %8.0f  share %5.1f %%" % (nm, buf[k] / rounds, buf[8 + k] / rounds, buf[16 + k] / buf[8 + k], 100.0 * buf[16 + k] / tot))
print("advance  %5.1f %%   set-up %5.1f %%   selection %5.1f %%" % (100.0 * buf[24] / tot, 100.0 * buf[25] / tot, 100.0 * buf[26] / tot))
if buf[37]:
    print("tree batch (mean cycles): own literal tree %.0f  wait for the others %.0f  header %.0f  publish %.0f  (%.1f batches per round, %.1f rebuilds per round taken from an equal histogram)" % (buf[33] / buf[37], buf[34] / buf[37], buf[35] / buf[37], buf[36] / buf[37], buf[37] / rounds, buf[38] / rounds))
print("ids per round: masks %.1f codes %.1f headers %.1f" % (buf[30] / rounds, buf[31] / rounds, buf[32] / rounds))
print("slowest block %.0f cycles (%.2f ms); blocks by duration (0.21 ms classes): %s" % (buf[39], buf[39] / 2.4e6, " ".join(str(buf[40 + k]) for k in range(23))))
print("first rounds: %.1f %% of the cycles" % (100.0 * buf[63] / tot))
