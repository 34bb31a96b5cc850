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
