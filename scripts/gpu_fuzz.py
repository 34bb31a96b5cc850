#!/usr/bin/env python3
"""Dev tool (GPU box): randomized parity run — many differently shaped raw deflate streams through the library,
merge on and off, every output compared bit for bit with the CPU oracle (tests/oracle_lib.py).  The shapes aim at the
paths the benchmark workload never takes: > 64 used literal symbols (queue slots in LDS), Fibonacci-like
frequencies (trees deeper than 15 / 7 -> the reference's depth limiter), stored / fixed / empty blocks from flushes,
HUFFMAN_ONLY and RLE strategies, long runs (len-258 matches), tiny blocks.

usage: gpu_fuzz.py [seconds] [seed] [--big] [--sim] [--diff]
  --sim   the CPU emulation of the kernels, tests/hostsim — small inputs
  --diff  no oracle: every batch runs under each executor configuration (fused, fused with cluster mode for every block
          above 2000 back-references, levels, persistent) and the outputs must be identical — fast, so also 4-8 MiB inputs;
          the level executor is the one the oracle runs of rounds 1-2 pinned"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deft4j_amd as D, oracle_lib as O, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
BIG = "--big" in sys.argv   # also 1-3 MiB inputs (many blocks, several merge chains)
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 12345)
SIM = "--sim" in sys.argv
DIFF = "--diff" in sys.argv
CONFIGS = [{"D4G_EXEC": "fused"}, {"D4G_EXEC": "fused", "D4G_CLUSTER_MIN_REFS": "2000", "D4G_FUSED_MAX_REFS": "2000"},
           {"D4G_EXEC": "levels"}, {"D4G_EXEC": "persistent"}]
LIB = None
if SIM:
    os.environ.setdefault("D4G_SIM_BLOCK", "64")
    LIB = D.load_library(os.path.join(ROOT, "tests", "hostsim", os.environ.get("D4G_SIM_LIB", "libdeft4g_hostsim.so")))
D.init(0, lib=LIB)


def fib_bytes(n):
    # byte k appears ~fib(k) times: a maximally skewed histogram
    out = bytearray(); a, b, k = 1, 1, 0
    while len(out) < n and k < 40:
        out += bytes([65 + k]) * min(a, n - len(out)); a, b, k = b, a + b, k + 1
    lst = list(out); rng.shuffle(lst)
    return bytes(lst)


def gen():
    kind = rng.randrange(9)
    n = rng.choice([0, 1, 2, 50, 300, 3000, 20000] + ([] if SIM else [70000, 200000]) + ([1 << 19, 1 << 20] if BIG else []) + ([1 << 22, 1 << 23] if BIG and DIFF else []))
    if kind == 0: raw = synth.reptext(n, rng.randrange(1 << 30))
    elif kind == 1: raw = bytes(rng.randrange(256) for _ in range(min(n, 30000)))
    elif kind == 2: raw = fib_bytes(n)
    elif kind == 3: raw = bytes([rng.randrange(4)]) * n
    elif kind == 4: raw = (bytes(rng.randrange(256) for _ in range(97)) * (n // 97 + 1))[:n]
    elif kind == 5: raw = bytes(min(255, int(rng.expovariate(0.05))) for _ in range(min(n, 50000)))
    elif kind == 6: raw = synth.reptext(n, rng.randrange(1 << 30)).upper() + bytes(range(256)) * 3
    elif kind == 7: raw = b"".join(bytes([rng.randrange(256)]) * rng.randrange(1, 600) for _ in range(max(1, n // 300)))
    else: raw = synth.reptext(n // 2, 7) + bytes(rng.randrange(256) for _ in range(min(n // 2, 20000)))
    level = rng.choice([1, 6, 9, 9, 9])
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
    c = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 8, 9]), strat)
    out = bytearray(); p = 0
    while p < len(raw):
        step = rng.choice([len(raw), 5000, 700, 64])
        out += c.compress(raw[p:p + step]); p += step
        if rng.random() < 0.3: out += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK]))
    out += c.flush()
    return bytes(out), raw


t0 = time.time(); nstreams = 0; nbytes = 0; it = 0
while time.time() - t0 < budget:
    batch = [gen() for _ in range(rng.randrange(1, 6 if BIG or SIM else 24))]
    ins = [b[0] for b in batch]
    if DIFF:
        for merge in (False, True):
            ref = None
            for cfg in CONFIGS:
                for k in ("D4G_EXEC", "D4G_CLUSTER_MIN_REFS", "D4G_FUSED_MAX_REFS"): os.environ.pop(k, None)
                os.environ.update(cfg)
                bt = D.Batch(ins, lib=LIB).run(merge)
                got = [(bt.result(i)["status"], bt.result(i)["saved_bits"], bt.output(i) if bt.result(i)["status"] == 0 else None) for i in range(len(ins))]
                bt.close()
                if ref is None: ref = got
                elif got != ref:
                    i = next(j for j in range(len(ins)) if got[j] != ref[j])
                    import pickle
                    pickle.dump(ins, open("gpurun_out/fuzz_diff_fail_%d.pkl" % it, "wb"))
                    print("MISMATCH iteration %d stream %d merge %s under %s: status %s/%s saved %s/%s" % (it, i, merge, cfg, got[i][0], ref[i][0], got[i][1], ref[i][1]), flush=True)
                    sys.exit(1)
        nstreams += len(ins); nbytes += sum(len(b[1]) for b in batch); it += 1
        if it % 5 == 0: print("%d iterations, %d streams, %.1f MB decoded, %.0fs" % (it, nstreams, nbytes / 1e6, time.time() - t0), flush=True)
        continue
    for merge in (False, True):
        bt = D.Batch(ins, lib=LIB).run(merge)
        for i, (a, raw) in enumerate(batch):
            if BIG: print("  oracle: stream %d (%d bytes) merge %s" % (i, len(a), merge), flush=True)
            rc, want, saved, consumed, _ = O.optimise(a, merge)
            r = bt.result(i)
            ok = r["status"] == rc and (rc < 0 or (r["saved_bits"] == saved and bt.output(i) == want))
            if not ok:
                open("gpurun_out/fuzz_fail_%d_%d.deflate" % (it, i), "wb").write(a)
                print("MISMATCH iteration %d stream %d merge %s: status %s/%s saved %s/%s len %d" % (it, i, merge, r["status"], rc, r["saved_bits"], saved, len(a)), flush=True)
                sys.exit(1)
        bt.close()
    nstreams += len(ins); nbytes += sum(len(b[1]) for b in batch); it += 1
    if BIG or it % 10 == 0: print("%d iterations, %d streams, %.1f MB decoded, %.0fs" % (it, nstreams, nbytes / 1e6, time.time() - t0), flush=True)
print("OK: %d iterations, %d streams, %.1f MB decoded, all outputs identical %s (merge on and off)" % (it, nstreams, nbytes / 1e6, "under every executor configuration" if DIFF else "to the oracle"))
