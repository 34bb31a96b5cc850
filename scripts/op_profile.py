#!/usr/bin/env python3
"""Dev tool: build libdeft4g with -DD4G_PROFILE_OPS and print mean cycles per search-program op kind."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = os.path.join(ROOT, "gpurun_out", "libdeft4g_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DD4G_PROFILE_OPS", "-shared", "-fPIC",
                       "-o", so, os.path.join(ROOT, "deft4j_amd", "csrc", "libdeft4g.hip")])
import deft4j_amd as D, synth
L = D.load_library(so); D.init(0, lib=L)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
exp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L.d4g_debug_set_experiment.argtypes = [ctypes.c_longlong]
L.d4g_debug_set_experiment(exp)
s = synth.make_stream(mib << 20)
for it in range(2):
    try:
        b = D.Batch([s], lib=L).run(False); st = b.stats(); b.close()
    except RuntimeError as e:
        print('run error (expected in experiments):', str(e)[:80])
buf = (ctypes.c_longlong * 64)()
L.d4g_debug_opstats(buf)
names = {1: "OPT", 2: "RECODE", 3: "FULL", 4: "LEAST", 5: "POST", 6: "PRUNEHDR", 7: "TOFIXED"}
print("ms_optimise %.1f state_ms %.1f search_ms %.1f" % (st["ms_optimise"], st["ms_state_kernels"], st["ms_search_kernels"]))
if buf[23]:
    print("OPT sections (mean cycles): load %.0f  token pass %.0f  optimise_header %.0f" % (buf[20] / buf[23], buf[21] / buf[23], buf[22] / buf[23]))
if buf[27]:
    print("recode_huffman sections (mean cycles): lit tree %.0f  dist-tree wait %.0f  sizes+header %.0f  (n=%d)"
          % (buf[24] / buf[27], buf[25] / buf[27], buf[26] / buf[27], buf[27]))
if buf[54]:
    print("hdr search (mean cycles): load+runs %.0f  candidates %.0f  (n=%d); lane 0: count pass %.0f  tree %.0f  rest %.0f"
          % (buf[52] / buf[54], buf[53] / buf[54], buf[54], buf[48] / max(1, buf[51]), buf[49] / max(1, buf[51]), buf[50] / max(1, buf[51])))
if buf[54]:
    print("hdr search: mean flag-dependent runs per op %.1f" % (buf[55] / buf[54]))
if buf[31]:
    print("header rewrite (mean cycles): runs+pairs %.0f  code-length tree %.0f  tail %.0f  (n=%d)" % (buf[28] / buf[31], buf[29] / buf[31], buf[30] / buf[31], buf[31]))
if buf[56]:
    print("lit tree (mean cycles): leaves %.0f  merges %.0f  depths %.0f" % (buf[16] / buf[56], buf[17] / buf[56], buf[18] / buf[56]))
print("wave trees: lit %d (limiter %d, mean leaves %.1f)  dist %d (limiter %d)  code-length %d (limiter %d)"
      % (buf[56], buf[57], buf[58] / max(1, buf[56]), buf[59], buf[60], buf[61], buf[62]))
if buf[36]:
    print("computed replace passes (mean cycles): memo lookup %.0f  record loop %.0f  (n=%d, mean records %.0f)" % (buf[34] / buf[36], buf[35] / buf[36], buf[36], buf[37] / buf[36]))
print("token passes: %d computed, %d served from the memo (mean wait %.0f cycles)" % (buf[19], buf[33], buf[32] / max(1, buf[33])))
for arg in (0, 1):
    for k, nm in names.items():
        cyc, n = buf[k * 2 + arg * 32], buf[k * 2 + 1 + arg * 32]
        if n:
            print("%-9s arg&1=%d  n=%8d  mean %9.0f cycles (%.1f us @2.4GHz)  total %.1f Gcyc" % (nm, arg, n, cyc / n, cyc / n / 2400.0, cyc / 1e9))

