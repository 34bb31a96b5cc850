#!/usr/bin/env python3
"""Dev tool (GPU box): randomized parity run of the LZ77 encoder path.  Differently shaped inputs go through
d4g_batch_create_encode in batches; every DEFAULT / FILTERED / HUFFMAN_ONLY output of the zlib flavour is compared byte for
byte with Python zlib level 9, every jzlib-flavour output with the oracle (oracle/zlib9_oracle.c), and the encode+optimise
path with the library's own parse+optimise of the encoder's bytes (merge on and off).

usage: gpu_fuzz_lz.py [seconds] [seed] [--big]"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, synth, zl9_lib as Z

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
BIG = "--big" in sys.argv
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
D.init(0)
ZS = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY)


def zref(d, st):
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, ZS[st])
    return c.compress(d) + c.flush()


def gen():
    kind = rng.randrange(12)
    sizes = [0, 1, 2, 3, 4, 257, 258, 259, 2047, 2048, 2049, 4000, 16383, 16384, 20000, 32767, 32768, 32769, 65273, 65274, 65275, 65276, 65535, 65536, 70000, 131072, 200000]
    n = rng.choice(sizes + ([1 << 19, (1 << 20) + rng.randrange(5000), 3 << 20] if BIG else [])) + (rng.randrange(3) if rng.random() < 0.2 else 0)
    if kind == 0: raw = synth.reptext(n, rng.randrange(1 << 30))
    elif kind == 1: raw = os.urandom(min(n, 120000))
    elif kind == 2: raw = bytes([rng.randrange(256)]) * n                                   # one long run: 258-byte matches
    elif kind == 3: raw = (os.urandom(rng.choice([2, 3, 5, 7, 97, 259, 1000])) * (n + 1))[:n]   # periodic
    elif kind == 4: raw = bytes(rng.choice(b"ACGT") for _ in range(min(n, 150000)))        # 4-letter alphabet: chains hit the 4096 cap
    elif kind == 5: raw = bytes(rng.choice(b"ab") for _ in range(min(n, 120000)))
    elif kind == 6: raw = synth.pngidat(n, rng.randrange(1 << 30), rng.choice([16, 100, 640]))
    elif kind == 7: raw = os.urandom(min(n // 2, 40000)) + synth.reptext(n // 2, rng.randrange(1 << 20))
    elif kind == 8:                                                                            # far repeats: distances near 32506 / TOO_FAR
        blk = os.urandom(rng.choice([3, 4, 5, 40]))
        gap = rng.choice([4090, 4096, 4097, 32500, 32505, 32506, 32507, 32768])
        raw = b"".join(blk + os.urandom(max(0, gap - len(blk))) for _ in range(max(1, min(n, 400000) // gap)))
    elif kind == 9: raw = b"".join(bytes([rng.randrange(256)]) * rng.randrange(1, 700) for _ in range(max(1, n // 350)))
    elif kind == 10: raw = synth.reptext(16383 * rng.randrange(1, 4), rng.randrange(1 << 20))[:n] if n else b""
    else: raw = bytes(min(255, int(rng.expovariate(0.03))) for _ in range(min(n, 100000)))
    return raw


t0 = time.time()
nin = nout = 0
while time.time() - t0 < budget:
    ins = [gen() for _ in range(rng.randrange(1, 10))]
    specs = [(i, D.ENC_JVM, st) for i in range(len(ins)) for st in range(3)] + [(i, D.ENC_JZLIB, st) for i in range(len(ins)) for st in range(3)]
    b = D.EncodeBatch(ins, specs).run(False)
    encs = []
    for k, (i, enc, st) in enumerate(specs):
        got = b.output(k)
        want = zref(ins[i], st) if enc == D.ENC_JVM else Z.deflate(ins[i], st, Z.JZLIB)
        if got != want:
            fn = os.path.join(ROOT, "gpurun_out", "lzfuzz_fail_%d_%d_%d.bin" % (len(ins[i]), enc, st))
            open(fn, "wb").write(ins[i])
            print("MISMATCH len %d enc %d strategy %d -> %s (got %d bytes, want %d)" % (len(ins[i]), enc, st, fn, len(got), len(want)))
            sys.exit(1)
        encs.append(got)
    b.close()
    merge = rng.random() < 0.5
    small = [k for k, (i, _, _) in enumerate(specs) if len(ins[i]) <= 300000]
    if small:
        b = D.EncodeBatch(ins, [specs[k] for k in small]).run(True, merge)
        p = D.Batch([encs[k] for k in small]).run(merge)
        for q, k in enumerate(small):
            ra, rb = b.result(q), p.result(q)
            if ra["saved_bits"] != rb["saved_bits"] or b.output(q) != p.output(q):
                i, enc, st = specs[k]
                fn = os.path.join(ROOT, "gpurun_out", "lzfuzz_optfail_%d_%d_%d.bin" % (len(ins[i]), enc, st))
                open(fn, "wb").write(ins[i])
                print("ENCODE+OPTIMISE != PARSE+OPTIMISE len %d enc %d strategy %d merge %s -> %s" % (len(ins[i]), enc, st, merge, fn))
                sys.exit(1)
        b.close(); p.close()
    nin += len(ins); nout += len(specs)
    if nin % 50 < 10:
        print("%.0fs: %d inputs, %d encoder outputs ok" % (time.time() - t0, nin, nout), flush=True)
print("LZFUZZ_OK %d inputs %d outputs in %.0fs" % (nin, nout, time.time() - t0))
