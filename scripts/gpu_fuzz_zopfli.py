"""Randomised parity run of the Zopfli encoder kernels against the oracle (portable log flavour) on the GPU.
Data shapes are chosen to hit the special paths: long byte runs (the 258-shortcut, the run-length second hash, `same` capped at
65535), block ends inside runs (tail tables), chain-hit caps, stored / fixed-tree blocks, tiny inputs, several master blocks.
    python scripts/gpu_fuzz_zopfli.py [seconds] [seed] [--big]     (--big: inputs up to 400 KB, several master blocks, few iterations)"""
import os
import random
import sys
import time
import zlib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D
import synth
import zopf_lib as Z


def shape(rng, n):
    k = rng.randrange(9)
    if k == 0:
        return synth.reptext(n, rng.randrange(1 << 30))
    if k == 1:
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    if k == 2:
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(4)]) * rng.randrange(1, 3000)
        return bytes(out[:n])
    if k == 3:
        return bytes(rng.randrange(256) for _ in range(n))
    if k == 4:
        return synth.pngidat(n, rng.randrange(1 << 30), rng.choice((50, 200, 333)))
    if k == 5:
        base = synth.reptext(max(64, n // 7), rng.randrange(1 << 30))
        return (base * 8)[:n]
    if k == 6:
        return bytes(n)
    if k == 7:
        out = bytearray()
        while len(out) < n:
            out += synth.reptext(rng.randrange(50, 4000), rng.randrange(1 << 30)) + bytes([rng.randrange(256)]) * rng.randrange(0, 1200)
            out += bytes(rng.randrange(256) for _ in range(rng.randrange(0, 300)))
        return bytes(out[:n])
    return bytes(rng.randrange(16) * 16 for _ in range(n))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    D.init(0)
    t0 = time.time()
    cases = outs = 0
    while time.time() - t0 < budget:
        it = rng.choice((1, 2, 3, 5, 8))
        split = rng.randrange(3)
        maxb = rng.choice((15, 15, 0, 3))
        nmax = rng.choice((300, 2000, 12000, 40000))
        if "--big" in sys.argv:
            nmax = rng.choice((70000, 150000, 400000))
            it = rng.choice((1, 2, 3))
        datas = [shape(rng, rng.randrange(0, nmax)) for _ in range(rng.randrange(1, 7 if nmax < 50000 else 3))]
        longest = max(len(d) for d in datas)
        master = rng.choice((8 << 20, 1000000, max(1, longest // 3 + 1), max(1, rng.randrange(1, longest + 2))))
        got = D.zopfli_streams(datas, it, split, maxb, master)
        for d, g in zip(datas, got):
            want = Z.deflate(d, it, split, maxb, master, Z.LOG_PORTABLE)
            if g != want or zlib.decompress(g, -15) != d:
                name = "/tmp/zopfli_fuzz_fail_%d.bin" % cases
                open(name, "wb").write(d)
                print("MISMATCH: len %d it %d split %d maxb %d master %d -> %s" % (len(d), it, split, maxb, master, name), flush=True)
                sys.exit(1)
            outs += 1
        cases += 1
        if cases % 20 == 0:
            print("%d calls, %d streams identical, %.0f s" % (cases, outs, time.time() - t0), flush=True)
    print("done: %d calls, %d streams, no mismatch (seed %d)" % (cases, outs, seed), flush=True)


if __name__ == "__main__":
    main()
