"""Container wrappers around the device hot path: raw deflate, gzip, zlib.

Host-side mirrors of K/RawDeflateFile.java, K/GZFile.java and K/ZLibFile.java (K/ = deft4j-container/
src/main/java/com/github/NeRdTheNed/deft4j/container/): same fields, same read/write order, same quirks
(a gzip FCOMMENT is re-written without its terminating NUL, K/GZFile.java:117-119).  The deflate payload
is parsed / optimised / re-serialised by libdeft4g, and the trailers the reference recomputes on write
(`RECALC`, K/DeflateFilesContainer.java:15) — CRC-32 + ISIZE, Adler-32 — come from the device kernels.
`optimise_files` is the `deft4j optimise` flow of M/CMDUtil.java:57-116 for mode NONE, batched: all files'
streams go to the GPU in one batch.
"""
import struct

from . import Batch

FTEXT, FHCRC, FEXTRA, FNAME, FCOMMENT = 1, 2, 4, 8, 16
DEFAULT_NAME = "unnamed stream"


class RawDeflateFile:
    file_type = "Raw deflate stream"          # K/RawDeflateFile.java:34

    def read(self, data):
        self.payload = bytes(data)
        self.name = DEFAULT_NAME
        return True

    def payload_offset(self):
        return 0

    def write(self, deflate, crc32, adler32, isize):
        return deflate


class GZFile:
    file_type = "GZip"                        # K/GZFile.java:223

    def read(self, data):                     # K/GZFile.java:42-87
        d = bytes(data)
        if len(d) < 10 or d[0] != 0x1f or d[1] != 0x8b or d[2] != 8 or (d[3] & 0xe0):
            return False
        self.cm, self.flags = d[2], d[3]
        self.time = struct.unpack("<I", d[4:8])[0]
        self.xfl, self.os = d[8], d[9]
        p = 10
        self.extra = None
        if self.flags & FEXTRA:
            xlen = struct.unpack("<H", d[p:p + 2])[0]
            self.extra = d[p + 2:p + 2 + xlen]
            p += 2 + xlen
        self.filename = None
        if self.flags & FNAME:
            e = d.index(b"\0", p)
            self.filename = d[p:e]
            p = e + 1
        self.comment = None
        if self.flags & FCOMMENT:
            e = d.index(b"\0", p)
            self.comment = d[p:e]
            p = e + 1
        self.crc16 = None
        if self.flags & FHCRC:
            self.crc16 = d[p:p + 2]
            p += 2
        self._off = p
        self.payload = d[p:]                  # the parser reports how many bytes the deflate stream used
        self.name = self.filename.decode("latin-1") if self.filename else DEFAULT_NAME
        if self.filename is not None and len(self.filename) == 0:   # setFilename(""): FNAME is cleared (:155-166)
            self.flags &= ~FNAME
        return True

    def payload_offset(self):
        return self._off

    def write(self, deflate, crc32, adler32, isize):   # K/GZFile.java:93-152
        out = bytearray([0x1f, 0x8b, self.cm, self.flags])
        out += struct.pack("<I", self.time & 0xffffffff)
        out += bytes([self.xfl, self.os])
        if self.flags & FEXTRA:
            out += struct.pack("<H", len(self.extra)) + self.extra
        if self.flags & FNAME:
            out += self.filename + b"\0"
        if self.flags & FCOMMENT:
            out += self.comment               # no terminating NUL: the reference omits it
        if self.flags & FHCRC:
            out += self.crc16
        out += deflate
        out += struct.pack("<II", crc32 & 0xffffffff, isize & 0xffffffff)
        return bytes(out)


class ZLibFile:
    file_type = "ZLib"                        # K/ZLibFile.java:98

    def read(self, data):                     # K/ZLibFile.java:60-95
        d = bytes(data)
        if len(d) < 2:
            return False
        self.cmf, self.flg = d[0], d[1]
        if (self.cmf & 0xf) != 8 or ((self.cmf << 8) + self.flg) % 31 != 0 or (self.flg & 0x20):
            return False
        self.payload = d[2:]
        self.name = DEFAULT_NAME
        return True

    def payload_offset(self):
        return 2

    def write(self, deflate, crc32, adler32, isize):   # K/ZLibFile.java:33-57 (Adler-32 big-endian)
        return bytes([self.cmf, self.flg]) + deflate + struct.pack(">I", adler32 & 0xffffffff)


def detect(data):
    """K/ContainerUtil.java:64-86 by magic bytes (gzip, zlib); anything else must be given explicitly."""
    d = bytes(data[:2])
    if d == b"\x1f\x8b":
        return GZFile()
    if len(d) == 2 and (d[0] & 0xf) == 8 and ((d[0] << 8) + d[1]) % 31 == 0:
        return ZLibFile()
    return None


def optimise_files(files, merge_blocks=True, formats=None, lib=None):
    """files: list of bytes.  formats: optional list of container instances / None (auto-detect) / "raw".
    Returns [(output bytes or None when unreadable, transcript lines)] — the lines M/CMDUtil.java:64-74 and
    K/DeflateFilesContainer.java:31-40 print."""
    conts = []
    for i, f in enumerate(files):
        c = formats[i] if formats and formats[i] is not None else None
        if c == "raw":
            c = RawDeflateFile()
        if c is None:
            c = detect(f)
        ok = c is not None and c.read(f)
        conts.append(c if ok else None)
    idx = [i for i, c in enumerate(conts) if c is not None]
    batch = Batch([conts[i].payload for i in idx], lib=lib).run(merge_blocks) if idx else None
    results = [(None, ["Failed to read file"]) for _ in files]
    for k, i in enumerate(idx):
        c = conts[i]
        r = batch.result(k)
        if r["status"] < 0:
            continue
        lines = ["File type recognised as " + c.file_type]
        saved = r["saved_bits"]
        if saved > 0:
            lines.append("%d bits saved in stream 0 (%s)" % (saved, c.name))
            lines.append("Total bits saved %d" % saved)
            lines.append("Saved %d bits with optimisation" % saved)
        crc, adler, isize = batch.checksums(k)
        results[i] = (c.write(batch.output(k), crc, adler, isize), lines)
    if batch is not None:
        batch.close()
    return results
