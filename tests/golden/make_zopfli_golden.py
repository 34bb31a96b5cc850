"""Generates the committed Zopfli golden vectors (tests/golden/zopfli_*.deflate + zopfli_manifest.json): seeded inputs
(re-used lz_*.bin files where they exist) and what the in-container proxy libzopfli 1.0.3 (/opt/conda/lib, SURVEY.md
§8c — it reproduces the reference's own test/asyoulik/asyoulik-zopfli.txt.gz at 5 iterations) makes of them through
ZopfliCompress(ZOPFLI_FORMAT_DEFLATE).  The proxy does not travel; only these vectors do.
Run from the repo root: python tests/golden/make_zopfli_golden.py"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
import zopf_lib as Z  # noqa: E402


def cases():
    rng = random.Random(0x20F)
    # name, data, iterations, blocksplitting, blocksplittingmax
    yield "empty", b"", 15, 1, 15
    yield "one", b"x", 15, 1, 15
    yield "tiny", b"hello hello hello hello", 15, 1, 15
    yield "reptext40k", synth.reptext(40000, 0xD4F7), 15, 1, 15
    yield "reptext40k_i20", synth.reptext(40000, 0xD4F7), 20, 1, 15
    yield "reptext40k_nosplit", synth.reptext(40000, 0xD4F7), 10, 0, 15
    yield "reptext70k_max0", synth.reptext(70000, 5), 8, 1, 0
    yield "acgt30k", bytes(rng.choice(b"ACGT") for _ in range(30000)), 10, 1, 15
    yield "zeros70k", b"\0" * 70000, 15, 1, 15                   # long-repetition shortcut, run-length second hash
    yield "runs40k", b"".join(bytes([rng.randrange(5)]) * rng.randrange(1, 900) for _ in range(90)), 15, 1, 15
    yield "noise20k", bytes(rng.randrange(256) for _ in range(20000)), 15, 1, 15   # stored block
    yield "short900", synth.reptext(900, 9), 15, 1, 15            # fixed-tree re-parse path (store < 1000 symbols)
    yield "pngrows48k", synth.pngidat(48000, 3, 200), 15, 1, 15


def main():
    man = []
    for name, data, it, bs, mb in cases():
        open(os.path.join(HERE, "zopfli_%s.bin" % name), "wb").write(data)
        out = Z.proxy_deflate(data, it, bs, mb)
        open(os.path.join(HERE, "zopfli_%s.deflate" % name), "wb").write(out)
        man.append({"name": name, "len": len(data), "iterations": it, "blocksplitting": bs, "blocksplittingmax": mb,
                    "sha256_in": hashlib.sha256(data).hexdigest(), "out_len": len(out), "sha256_out": hashlib.sha256(out).hexdigest()})
    json.dump({"proxy": "libzopfli 1.0.3 (master block 1000000)", "cases": man}, open(os.path.join(HERE, "zopfli_manifest.json"), "w"), indent=1)
    print("wrote", len(man), "cases")


if __name__ == "__main__":
    main()
