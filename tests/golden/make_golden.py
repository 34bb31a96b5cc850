#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's committed fixture pairs.

Reads the DATA files under /root/reference/test (inputs, golden outputs and stdout
transcripts written by runTestOpt.sh:3-11) and slices out the raw DEFLATE payloads the
hot path sees: gzip members lose header/trailer (K/GZFile.java:42-87), PNG IDAT/fdAT/
zTXt/iTXt chunks are concatenated per stream the way K/PNGFile.java:484-544 does and
lose the 2-byte zlib header and Adler-32 (K/ZLibFile.java:60-95).  No reference source
is copied; only fixture data.  Run once in the build container (the reference does not
exist on the GPU box); the outputs are committed.
"""
import hashlib
import json
import os
import re
import struct
import sys

REF = "/root/reference/test/"
HERE = os.path.dirname(os.path.abspath(__file__))


def gz_payload(b):
    assert b[:2] == b"\x1f\x8b" and b[2] == 8
    flg = b[3]
    p = 10
    if flg & 4:
        p += 2 + struct.unpack("<H", b[p:p + 2])[0]
    if flg & 8:
        p = b.index(b"\0", p) + 1
    if flg & 16:
        p = b.index(b"\0", p) + 1
    if flg & 2:
        p += 2
    return b[p:-8]


def png_streams(b):
    """[(name, zlib bytes)] in deft4j's stream order: IDAT, fdAT frames, then zTXt/iTXt/iCCP."""
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    p = 8
    idat = b""
    frames = []
    cur = None
    other = []
    while p < len(b):
        ln = struct.unpack(">I", b[p:p + 4])[0]
        ty = b[p + 4:p + 8]
        d = b[p + 8:p + 8 + ln]
        p += 12 + ln
        if ty == b"IDAT":
            idat += d
        elif ty == b"fcTL":
            if cur is not None:
                frames.append(cur)
            cur = b""
        elif ty == b"fdAT":
            cur += d[4:]
        elif ty in (b"zTXt", b"iCCP"):
            k = d.index(b"\0")
            other.append((ty.decode() + " chunk", d[k + 2:]))
        elif ty == b"iTXt":
            k = d.index(b"\0")
            if d[k + 1] == 1:
                q = d.index(b"\0", k + 3)
                q = d.index(b"\0", q + 1)
                other.append(("iTXt chunk", d[q + 1:]))
    if cur:
        frames.append(cur)
    frames = [f for f in frames if f]
    return [("IDAT chunk", idat)] + [("fdAT chunk %d" % (i + 1), f) for i, f in enumerate(frames)] + other


def transcript(path):
    """{stream index: bits saved} and the total from a runTestOpt.sh transcript."""
    per, total = {}, 0
    for line in open(path):
        m = re.match(r"(\d+) bits saved in stream (\d+) ", line)
        if m:
            per[int(m.group(2))] = int(m.group(1))
        m = re.match(r"Total bits saved (\d+)", line)
        if m:
            total = int(m.group(1))
    return per, total


CASES = [  # (input, golden output, merge_blocks)
    ("deflate-store-2.txt.gz", "deflate-store-2-opt.txt.gz", True),
    ("lz-twice-twice.txt.gz", "lz-twice-twice-opt.txt.gz", True),
    ("asyoulik/asyoulik-gzip.txt.gz", "asyoulik/asyoulik-gzip-opt.txt.gz", True),
    ("asyoulik/asyoulik-zopfli.txt.gz", "asyoulik/asyoulik-zopfli-opt.txt.gz", True),
    ("text.png", "text-opt.png", True),
    ("apng/ball.png", "apng/ball-opt.png", True),
    ("284-edge-case/284.png", "284-edge-case/284-opt.png", True),
    ("nerd/nerd.png", "nerd/nerd-opt.png", False),
    ("nerd/nerd-extopt.png", "nerd/nerd-fullopt.png", False),
]
PARSE_ONLY = ["ban.txt.gz", "deflate-dynamic.txt.gz", "deflate-fixed.txt.gz", "deflate-store.txt.gz", "lz.txt.gz"]


def main():
    manifest = {"pairs": [], "parse_only": []}
    for inp, outp, merge in CASES:
        bi = open(REF + inp, "rb").read()
        bo = open(REF + outp, "rb").read()
        per, total = transcript(REF + outp + ".txt")
        if inp.endswith(".png"):
            si = [(n, z[2:-4]) for n, z in png_streams(bi)]
            so = [(n, z[2:-4]) for n, z in png_streams(bo)]
        else:
            si = [("unnamed stream", gz_payload(bi))]
            so = [("unnamed stream", gz_payload(bo))]
        assert len(si) == len(so)
        base = inp.replace("/", "_").rsplit(".", 1)[0].replace(".txt", "")
        for k, ((name, a), (_, g)) in enumerate(zip(si, so)):
            stem = "%s.s%02d" % (base, k)
            open(os.path.join(HERE, stem + ".in.deflate"), "wb").write(a)
            open(os.path.join(HERE, stem + ".out.deflate"), "wb").write(g)
            manifest["pairs"].append({
                "stem": stem, "source": inp, "golden": outp, "stream": k, "name": name,
                "merge_blocks": merge, "saved_bits": per.get(k, 0), "file_total_saved": total,
                "in_sha256": hashlib.sha256(a).hexdigest(), "out_sha256": hashlib.sha256(g).hexdigest(),
                "in_len": len(a), "out_len": len(g)})
    for inp in PARSE_ONLY:
        a = gz_payload(open(REF + inp, "rb").read())
        stem = inp.rsplit(".", 1)[0].replace(".txt", "")
        open(os.path.join(HERE, stem + ".parse.deflate"), "wb").write(a)
        manifest["parse_only"].append({"stem": stem, "source": inp, "in_len": len(a)})
    z = open(REF + "deflate-fixed.txt.zz", "rb").read()
    open(os.path.join(HERE, "deflate-fixed-zz.parse.deflate"), "wb").write(z[2:-4])
    manifest["parse_only"].append({"stem": "deflate-fixed-zz", "source": "deflate-fixed.txt.zz", "in_len": len(z) - 6})
    # whole container files (gzip, PNG) with their transcripts, for the container-level tests
    manifest["files"] = []
    for inp, outp, merge in CASES:
        stem = inp.replace("/", "_")
        open(os.path.join(HERE, stem + ".file.in"), "wb").write(open(REF + inp, "rb").read())
        open(os.path.join(HERE, stem + ".file.out"), "wb").write(open(REF + outp, "rb").read())
        manifest["files"].append({"stem": stem, "source": inp, "golden": outp, "merge_blocks": merge,
                                  "transcript": open(REF + outp + ".txt").read().splitlines()})
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("wrote", len(manifest["pairs"]), "pairs,", len(manifest["parse_only"]), "parse-only vectors")


if __name__ == "__main__":
    sys.exit(main())
