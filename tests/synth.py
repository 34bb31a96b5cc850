"""Synthetic workloads of SURVEY.md §8d (BASELINE.json configs): `reptext` repetitive text and its
zlib level-9 raw deflate (== the reference's JavaCompressor, C/JavaCompressor.java:36-49)."""
import random
import zlib


def reptext(n, seed=0xD4F7):
    """Vocabulary of 4096 pseudo-words drawn Zipf(1.1), 72-column lines, and with p=0.05 after a word a
    copy of an earlier 32-512 byte span from the last 32 KiB."""
    rng = random.Random(seed)
    vocab = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 10))) for _ in range(4096)]
    weights = [1.0 / (i + 1) ** 1.1 for i in range(4096)]
    out = bytearray()
    col = 0
    # draw words in chunks (random.choices does the inverse-CDF walk on the same RNG)
    while len(out) < n:
        for w in rng.choices(vocab, weights, k=4096):
            out += w.encode()
            col += len(w)
            if col >= 72:
                out += b"\n"
                col = 0
            else:
                out += b" "
                col += 1
            if rng.random() < 0.05 and len(out) > 64:
                ln = rng.randint(32, 512)
                lo = max(0, len(out) - 32768)
                st = rng.randint(lo, max(lo, len(out) - ln))
                out += out[st:st + ln]
            if len(out) >= n:
                break
    return bytes(out[:n])


def deflate9(data, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def make_stream(n, seed=0xD4F7):
    return deflate9(reptext(n, seed))


def pngidat(n, seed=0x1DA7, width=1024):
    """PNG-IDAT-like bytes: rows of `width` RGB pixels, each row = one filter byte (1 = Sub)
    followed by the Sub-filtered samples of smooth gradients plus a little noise.  Returns n bytes (whole rows, truncated)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    row = 1 + 3 * width
    rows = (n + row - 1) // row
    y = np.arange(rows, dtype=np.int64)[:, None, None]
    x = np.arange(width, dtype=np.int64)[None, :, None]
    ch = np.arange(3, dtype=np.int64)[None, None, :]
    base = (x * (2 + ch) // 3 + y * (1 + ch) + 40 * np.sin((x + 3 * y) / 97.0).astype(np.int64))
    noise = rng.integers(0, 3, size=(rows, width, 3))
    img = ((base + noise) & 255).astype(np.uint8).reshape(rows, width * 3)
    sub = img.copy()
    sub[:, 3:] = img[:, 3:] - img[:, :-3]          # PNG filter type 1 (Sub), bpp = 3
    out = np.empty((rows, row), dtype=np.uint8)
    out[:, 0] = 1
    out[:, 1:] = sub
    return out.tobytes()[:n]


def mixed_spec(i, seed=5):
    """Stream i of the config-5 mix (PNG-IDAT-like streams of 1-16 MiB and 1 MiB text members, interleaved by a seeded
    choice) without generating it: -> (kind, uncompressed length)."""
    rng = random.Random((seed << 20) ^ i)
    if rng.random() < 0.25:
        return "idat", rng.randint(1, 16) << 20
    return "gzip", 1 << 20


def mixed_stream(i, seed=5):
    """-> (kind, raw bytes, zlib-9 raw deflate stream) of stream i of the config-5 mix."""
    kind, n = mixed_spec(i, seed)
    raw = pngidat(n, 0x1DA7 + i) if kind == "idat" else reptext(n, 0xD4F7 + i)
    return kind, raw, deflate9(raw)
