#!/usr/bin/env python3
"""Dev tool: time the CRC-32/Adler-32 kernels on the 64 MiB config-2 stream (HBM-bound reduction)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, synth
D.init(0)
raw = synth.reptext(64 << 20)
s = synth.deflate9(raw)
for it in range(3):
    b = D.Batch([s]).parse()
    ok = b.checksums(0) == (zlib.crc32(raw), zlib.adler32(raw), len(raw))
    ms = b.stats()["ms_checksum_kernels"]
    print("run %d: ok=%s checksum kernels %.3f ms -> %.1f GB/s of decoded bytes (%.2f%% of 8 TB/s)" % (it, ok, ms, len(raw) / ms / 1e6, len(raw) / ms / 1e6 / 80))
    b.close()
