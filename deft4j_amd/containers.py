"""Container wrappers around the device hot path: raw deflate, gzip, zlib, PNG, zip.

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


class PNGFile:
    """K/PNGFile.java: chunk walk (:162-215), stream collection state machine (:413-544), re-chunking on write
    (syncStreams :262-369).  One zlib stream for the concatenated IDAT chunks, one per APNG frame (fdAT chunks
    between fcTLs), one per zTXt / iTXt / iCCP chunk."""
    file_type = "PNG"
    SIG = b"\x89PNG\r\n\x1a\n"

    @staticmethod
    def _zlib_offset(ty, d):   # PNGChunk.getZLibCompressedNonIdat — :84-113 (None: not deflate-compressed)
        off = d.index(b"\0") + 2
        if ty == b"iTXt":
            if d[off - 1] != 1:
                return None
            off += 1
        if d[off - 1] != 0:
            return None
        if ty == b"iTXt":
            off = d.index(b"\0", off) + 1
            off = d.index(b"\0", off) + 1
        return off

    def read(self, data):
        import zlib as _z
        d = bytes(data)
        if d[:8] != self.SIG:
            return False
        self.chunks = []          # [type, data]
        self.streams = []         # (kind, name, ZLibFile, chunk index for non-IDAT)
        p = 8
        reading_idat = reading_fdat = seen_actl = seen_idat = seen_iend = False
        seq = 0
        acc = bytearray()
        nfd = 0

        def flush():
            nonlocal reading_idat, reading_fdat, seen_idat, nfd, acc
            z = ZLibFile()
            if not z.read(bytes(acc)):
                return False
            if reading_idat:
                self.streams.append(("IDAT", "IDAT chunk", z, None))
                seen_idat = True
                reading_idat = False
            else:
                nfd += 1
                self.streams.append(("fdAT", "fdAT chunk %d" % nfd, z, None))
                reading_fdat = False
            acc = bytearray()
            return True

        while True:
            if p + 12 > len(d):
                return False
            ln = struct.unpack(">I", d[p:p + 4])[0]
            ty = d[p + 4:p + 8]
            cd = d[p + 8:p + 8 + ln]
            crc = struct.unpack(">I", d[p + 8 + ln:p + 12 + ln])[0]
            p += 12 + ln
            if _z.crc32(ty + cd) != crc:          # PNGChunk.read returns calcCrc == CRC32
                return False
            if seen_iend:
                return False
            if ty in (b"fcTL", b"fdAT"):
                s = struct.unpack(">I", cd[:4])[0]
                if (seen_idat and not seen_actl) or s != seq:
                    return False
                seq += 1
            if (reading_idat and ty != b"IDAT") or (reading_fdat and ty in (b"fcTL", b"IEND")):
                if not flush():
                    return False
            self.chunks.append([ty, cd])
            if ty == b"IEND":
                seen_iend = True
                break
            if ty == b"acTL":
                if seen_idat or seen_actl:
                    return False
                seen_actl = True
                continue
            if ty == b"fdAT":
                if not seen_idat or reading_idat:
                    return False
                reading_fdat = True
            elif ty == b"IDAT":
                if seen_idat or reading_fdat:
                    return False
                reading_idat = True
            if len(cd) > 0:
                if reading_idat:
                    acc += cd
                elif reading_fdat and ty == b"fdAT":
                    acc += cd[4:]
                elif ty in (b"zTXt", b"iTXt", b"iCCP"):
                    off = self._zlib_offset(ty, cd)
                    if off is not None:
                        z = ZLibFile()
                        if z.read(cd[off:]):
                            self.streams.append((ty.decode(), ty.decode() + " chunk", z, len(self.chunks) - 1))
        if not (seen_idat and not reading_idat and not reading_fdat and (nfd == 0 or seen_actl)):
            return False
        # deft4j's stream order: IDAT, fdAT frames, then the other chunks (K/PNGFile.java:377-389)
        self.streams.sort(key=lambda s: {"IDAT": 0, "fdAT": 1}.get(s[0], 2))
        return True

    def stream_payloads(self):
        return [(name, z.payload) for _, name, z, _ in self.streams]

    def write(self, zlib_outputs):
        """zlib_outputs[i] = re-serialised zlib bytes of stream i (same order as stream_payloads)."""
        import zlib as _z
        chunks = [list(c) for c in self.chunks]
        outs = {i: zlib_outputs[i] for i in range(len(self.streams))}
        idat_i = [i for i, s in enumerate(self.streams) if s[0] == "IDAT"][0]
        fd_is = [i for i, s in enumerate(self.streams) if s[0] == "fdAT"]
        # non-IDAT chunks first (indices refer to the original chunk list)
        for i, (kind, _, _, ci) in enumerate(self.streams):
            if ci is not None:
                ty, cd = chunks[ci]
                off = self._zlib_offset(ty, cd)
                chunks[ci][1] = cd[:off] + outs[i]
        # IDAT: one chunk at the position of the first
        first = next(k for k, c in enumerate(chunks) if c[0] == b"IDAT")
        chunks = [c for c in chunks if c[0] != b"IDAT"]
        chunks.insert(first, [b"IDAT", outs[idat_i]])
        # fdAT frames: the first fdAT of each frame is replaced, the frame's other fdAT chunks are dropped
        if fd_is:
            res = []
            k = 0
            it = iter(fd_is)
            cur = next(it, None)
            state = "seek"
            for c in chunks:
                if cur is None:
                    res.append(c)
                    continue
                if state == "seek":
                    if c[0] == b"fdAT":
                        res.append([b"fdAT", b"\0\0\0\0" + outs[cur]])
                        state = "drop"
                    else:
                        res.append(c)
                else:
                    if c[0] == b"fdAT":
                        continue
                    res.append(c)
                    if c[0] == b"fcTL":
                        cur = next(it, None)
                        state = "seek"
            chunks = res
            seq = 0
            for c in chunks:
                if c[0] in (b"fcTL", b"fdAT"):
                    c[1] = struct.pack(">I", seq) + c[1][4:]
                    seq += 1
        out = bytearray(self.SIG)
        for ty, cd in chunks:
            out += struct.pack(">I", len(cd)) + ty + cd + struct.pack(">I", _z.crc32(ty + cd) & 0xffffffff)
        return bytes(out)


class ZipFile:
    """K/ZipFile.java:83-128 + K/lljzip/RecalculatingZipWriter.java:23-136.  The reference reads the
    archive with the third-party lljzip `ZipIO.readStandard` (un-vendored; end-of-central-directory record ->
    central directory -> local headers) — restated here for plain archives (no Zip64, no split archives), so this
    row's parity is unpinned: no reference fixture exercises it.  Writing follows RecalculatingZipWriter line by
    line: local headers take CRC and sizes from their central-directory entry, data descriptors are not re-emitted,
    central-directory offsets are remapped, the end record is rebuilt."""
    file_type = "Zip"                         # K/ZipFile.java:131
    multi = True

    def read(self, data):
        d = bytes(data)
        e = d.rfind(b"PK\x05\x06")
        if e < 0 or e + 22 > len(d):
            return False
        (self.disk, self.start_disk, _n_here, n_total, cd_size, cd_off, clen) = struct.unpack("<HHHHIIH", d[e + 4:e + 22])
        if n_total == 0xffff or cd_off == 0xffffffff:
            return False                       # Zip64: "may be due to use of ... Zip64 features" (K/ZipFile.java:90-93)
        self.comment = d[e + 22:e + 22 + clen]
        self.central = []
        p = cd_off
        for _ in range(n_total):
            if d[p:p + 4] != b"PK\x01\x02":
                return False
            f = struct.unpack("<HHHHHHIIIHHHHHII", d[p + 4:p + 46])
            ent = dict(made_by=f[0], needed=f[1], flags=f[2], method=f[3], time=f[4], date=f[5], crc=f[6], csize=f[7],
                       usize=f[8], disk_start=f[12], int_attr=f[13], ext_attr=f[14], offset=f[15])
            q = p + 46
            ent["name"] = d[q:q + f[9]]
            ent["extra"] = d[q + f[9]:q + f[9] + f[10]]
            ent["fcomment"] = d[q + f[9] + f[10]:q + f[9] + f[10] + f[11]]
            self.central.append(ent)
            p = q + f[9] + f[10] + f[11]
        self.locals = []
        for ent in sorted(self.central, key=lambda x: x["offset"]):
            o = ent["offset"]
            if d[o:o + 4] != b"PK\x03\x04":
                return False
            f = struct.unpack("<HHHHHIIIHH", d[o + 4:o + 30])
            lf = dict(needed=f[0], flags=f[1], method=f[2], time=f[3], date=f[4], crc=f[5], csize=f[6], usize=f[7], cen=ent)
            q = o + 30
            lf["name"] = d[q:q + f[8]]
            lf["extra"] = d[q + f[8]:q + f[8] + f[9]]
            if lf["csize"] == 0 and (ent["csize"], ent["usize"], ent["crc"]) != (lf["csize"], lf["usize"], lf["crc"]):
                lf["csize"], lf["usize"], lf["crc"] = ent["csize"], ent["usize"], ent["crc"]   # data descriptors (K/ZipFile.java:103-106)
            q += f[8] + f[9]
            lf["data"] = d[q:q + lf["csize"]]
            self.locals.append(lf)
        if not self.locals:
            return False
        self.members = [lf for lf in self.locals if lf["method"] == 8]   # DEFLATED only (K/ZipFile.java:98-100)
        return True

    def stream_payloads(self):
        return [(lf["name"].decode("utf-8", "replace") if lf["name"] else DEFAULT_NAME, lf["data"]) for lf in self.members]

    def write(self, deflate_outputs):
        for lf, out in zip(self.members, deflate_outputs):   # syncStreams — K/ZipFile.java:46-68
            lf["data"] = out
            lf["csize"] = len(out)
            lf["cen"]["csize"] = len(out)
        out = bytearray()
        new_off = {}
        for lf in self.locals:
            c = lf["cen"]
            new_off[c["offset"]] = len(out)
            out += struct.pack("<IHHHHHIIIHH", 0x04034b50, lf["needed"], lf["flags"], lf["method"], lf["time"], lf["date"],
                               c["crc"], c["csize"] & 0xffffffff, c["usize"] & 0xffffffff, len(lf["name"]), len(lf["extra"]))
            out += lf["name"] + lf["extra"] + lf["data"]
        start_central = len(out)
        for c in self.central:
            out += struct.pack("<IHHHHHHIIIHHHHHII", 0x02014b50, c["made_by"], c["needed"], c["flags"], c["method"], c["time"],
                               c["date"], c["crc"], c["csize"] & 0xffffffff, c["usize"] & 0xffffffff, len(c["name"]),
                               len(c["extra"]), len(c["fcomment"]), c["disk_start"], c["int_attr"], c["ext_attr"],
                               new_off[c["offset"]])
            out += c["name"] + c["extra"] + c["fcomment"]
        out += struct.pack("<IHHHHIIH", 0x06054b50, self.disk, self.start_disk, len(self.central), len(self.central),
                           len(out) - start_central, start_central, len(self.comment))
        out += self.comment
        return bytes(out)


def detect(data):
    """K/ContainerUtil.java:64-86 by magic bytes (PNG, zip, gzip, zlib); anything else must be given explicitly."""
    if bytes(data[:8]) == PNGFile.SIG:
        return PNGFile()
    if bytes(data[:4]) == b"PK\x03\x04":
        return ZipFile()
    d = bytes(data[:2])
    if d == b"\x1f\x8b":
        return GZFile()
    if len(d) == 2 and (d[0] & 0xf) == 8 and ((d[0] << 8) + d[1]) % 31 == 0:
        return ZLibFile()
    return None


def optimise_files(files, merge_blocks=True, formats=None, lib=None, mode=0):
    """files: list of bytes.  formats: optional list of container instances / None (auto-detect) / "raw".
    mode: RecompressMode ordinal of `deft4j optimise --mode` (0 NONE, 1 CHEAP, 2 ZOPFLI, 3 ZOPFLI_EXTENSIVE, 4 ZOPFLI_VERY_EXTENSIVE; M/CMDUtil.java:44-50,76-105): above NONE every
    stream is also recompressed and the recompression grafted in where it is smaller.
    Returns [(output bytes or None when unreadable, transcript lines)] — the lines M/CMDUtil.java:64-74 and
    K/DeflateFilesContainer.java:31-40 print.  Every deflate stream of every file goes to the GPU in one batch."""
    conts = []
    for i, f in enumerate(files):
        c = formats[i] if formats and formats[i] is not None else None
        if c == "raw":
            c = RawDeflateFile()
        if c is None:
            c = detect(f)
        ok = c is not None and c.read(f)
        conts.append(c if ok else None)
    payloads, owner = [], []
    for i, c in enumerate(conts):
        if c is None:
            continue
        sp = c.stream_payloads() if isinstance(c, (PNGFile, ZipFile)) else [(c.name, c.payload)]
        for name, pl in sp:
            payloads.append(pl)
            owner.append((i, name))
    batch = None
    if payloads:
        batch = Batch(payloads, lib=lib)
        batch.run_recompress(mode, merge_blocks) if mode > 0 else batch.run(merge_blocks)
    results = [(None, ["Failed to read file"]) for _ in files]
    k = 0
    for i, c in enumerate(conts):
        if c is None:
            continue
        n = len(c.stream_payloads()) if isinstance(c, (PNGFile, ZipFile)) else 1
        lines = ["File type recognised as " + c.file_type]
        total = 0
        ok = True
        pieces = []
        rlines, rtotal = [], 0
        for j in range(n):
            r = batch.result(k + j)
            if r["status"] < 0:
                ok = False
                break
            saved = r["saved_bits"]
            if saved > 0:
                lines.append("%d bits saved in stream %d (%s)" % (saved, j, owner[k + j][1]))
            total += saved
            if mode > 0:
                grafted, rs = batch.recompress_result(k + j)
                if grafted:                      # M/CMDUtil.java:94-98
                    orig = r["size_bits_in"] - saved
                    rlines.append("Recompressed stream %d (%s) from %d bits to %d bits, saved %d bits" % (j, owner[k + j][1], orig, orig - rs, rs))
                    rtotal += rs
            pieces.append((batch.output(k + j), None if isinstance(c, ZipFile) else batch.checksums(k + j)))
        if ok:
            if total > 0:
                lines.append("Total bits saved %d" % total)
                lines.append("Saved %d bits with optimisation" % total)
            lines += rlines
            if rtotal > 0:
                lines.append("Saved %d bits with recompression" % rtotal)
            if isinstance(c, ZipFile):   # members keep their CRC-32 (the decoded bytes do not change)
                results[i] = (c.write([pc[0] for pc in pieces]), lines)
            elif isinstance(c, PNGFile):
                zl = [c.streams[j][2].write(pieces[j][0], *pieces[j][1]) for j in range(n)]
                results[i] = (c.write(zl), lines)
            else:
                results[i] = (c.write(pieces[0][0], *pieces[0][1]), lines)
        k += n
    if batch is not None:
        batch.close()
    return results
