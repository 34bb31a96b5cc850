"""Generates the committed LZ77 golden vectors (tests/golden/lz_*.bin / .deflate): seeded inputs and what Python's zlib
1.2.11 — `compressobj(9, 8, -15, 8, strategy)`, the in-container stand-in for the reference's JavaCompressor
(C/JavaCompressor.java:36-49; SURVEY.md §0.5 verified it byte-identical on the asyoulik fixture) — makes of them.
Run from the repo root: python tests/golden/make_lz_golden.py"""
import hashlib
import json
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

STRATS = {"default": zlib.Z_DEFAULT_STRATEGY, "filtered": zlib.Z_FILTERED, "huffman": zlib.Z_HUFFMAN_ONLY}


def cases():
    rng = random.Random(0x17A)
    yield "empty", b""
    yield "one", b"x"
    yield "reptext40k", synth.reptext(40000, 0xD4F7)
    yield "reptext70k", synth.reptext(70000, 5)          # crosses the 65274-byte window slide and two sort blocks
    yield "acgt30k", bytes(rng.choice(b"ACGT") for _ in range(30000))      # long hash chains, chain-length cap
    yield "zeros70k", b"\0" * 70000                       # 258-byte matches, nice_length stop
    yield "noise20k", bytes(rng.randrange(256) for _ in range(20000))       # stored blocks
    yield "lits16383", bytes(rng.randrange(256) for _ in range(16383))      # a block filled exactly by the epilogue literal
    rows = [bytes([1]) + bytes(((x * 3 + y * 2 + rng.randint(0, 3)) & 255) for x in range(200)) for y in range(120)]
    yield "pngrows24k", b"".join(rows)                    # PNG-IDAT-like filtered rows (FILTERED matters)


def main():
    man = []
    for name, data in cases():
        open(os.path.join(HERE, "lz_%s.bin" % name), "wb").write(data)
        ent = {"name": name, "len": len(data), "sha256_in": hashlib.sha256(data).hexdigest(), "out": {}}
        for sname, st in STRATS.items():
            c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, st)
            out = c.compress(data) + c.flush()
            open(os.path.join(HERE, "lz_%s.%s.deflate" % (name, sname)), "wb").write(out)
            ent["out"][sname] = {"len": len(out), "sha256": hashlib.sha256(out).hexdigest()}
        man.append(ent)
    json.dump({"zlib": zlib.ZLIB_RUNTIME_VERSION, "cases": man}, open(os.path.join(HERE, "lz_manifest.json"), "w"), indent=1)
    print("wrote", len(man), "cases with zlib", zlib.ZLIB_RUNTIME_VERSION)


if __name__ == "__main__":
    main()
