"""Container level: whole gzip and PNG/APNG files through deft4j_amd.containers must equal the
reference's golden output FILES byte for byte (gzip: header kept, deflate re-serialised, CRC-32/ISIZE recomputed on
the device; PNG: IDAT / fdAT / zTXt / iTXt streams re-chunked as K/PNGFile.java does, Adler-32 from the device)
and print the reference's transcript lines — this is runTestOpt.sh:3-11 end to end.  CPU tests run the kernels in the emulator; the GPU test uses
the real library."""
import json
import os
import subprocess
import zlib

import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
FILES = json.load(open(os.path.join(G, "manifest.json")))["files"]


def rd(n):
    return open(os.path.join(G, n), "rb").read()


def check_files(lib, cases):
    from deft4j_amd import containers as C
    for merge in (True, False):
        sel = [f for f in cases if f["merge_blocks"] == merge]
        if not sel:
            continue
        res = C.optimise_files([rd(f["stem"] + ".file.in") for f in sel], merge, lib=lib)
        for f, (out, lines) in zip(sel, res):
            assert out == rd(f["stem"] + ".file.out"), f["stem"]
            assert lines == f["transcript"], f["stem"]


def check_zlib_and_raw(lib):
    from deft4j_amd import containers as C
    raw = synth.reptext(3000, 21)
    z = zlib.compress(raw, 9)
    (out, lines), (rout, _), (bad, _) = C.optimise_files([z, synth.deflate9(raw), b"\x00\x01garbage"], True,
                                                       formats=[None, "raw", None], lib=lib)
    assert lines[0] == "File type recognised as ZLib"
    assert zlib.decompress(out) == raw and out[:2] == z[:2] and out[-4:] == z[-4:]   # Adler-32 recomputed on the device
    assert zlib.decompress(rout, -15) == raw
    assert bad is None


def check_zip(lib):
    """§8f-4 (parity unpinned: the reference reads archives with the un-vendored lljzip; no fixture): the output must be
    a valid archive with the same members and contents, every deflated member must be the oracle's optimise() of the
    original member stream, stored members and metadata must be untouched."""
    import io
    import zipfile

    import oracle_lib as O
    from deft4j_amd import containers as C
    members = {"a/text.txt": synth.reptext(4000, 5), "b.bin": bytes(range(256)) * 4, "stored.txt": b"kept as is", "empty": b""}
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w") as z:
        for name, data in members.items():
            z.writestr(zipfile.ZipInfo(name, (2020, 1, 2, 3, 4, 6)), data,
                       zipfile.ZIP_STORED if name == "stored.txt" else zipfile.ZIP_DEFLATED, 9)
        z.comment = b"archive comment"
    src = bio.getvalue()
    (out, lines), = C.optimise_files([src], True, lib=lib)
    assert lines[0] == "File type recognised as Zip"
    zi, zo = zipfile.ZipFile(io.BytesIO(src)), zipfile.ZipFile(io.BytesIO(out))
    assert zo.testzip() is None and zo.comment == b"archive comment"
    assert [i.filename for i in zo.infolist()] == [i.filename for i in zi.infolist()]
    total = 0
    for a, b in zip(zi.infolist(), zo.infolist()):
        assert zo.read(b) == members[a.filename] and (b.CRC, b.file_size, b.date_time, b.compress_type) == (a.CRC, a.file_size, a.date_time, a.compress_type)
        ca = src[a.header_offset + 30 + len(a.filename.encode()) + len(a.extra):][:a.compress_size]
        cb = out[b.header_offset + 30 + len(b.filename.encode()) + len(b.extra):][:b.compress_size]
        if a.compress_type == zipfile.ZIP_DEFLATED:
            rc, want, saved, _, _ = O.optimise(ca, True)
            assert rc >= 0 and cb == (want if rc == 0 else ca), a.filename
            total += saved
        else:
            assert cb == ca
    assert ("Total bits saved %d" % total in lines) == (total > 0)


@pytest.fixture(scope="module")
def sim():
    os.environ["D4G_SIM_BLOCK"] = "64"
    subprocess.check_call([os.path.join(ROOT, "tests", "hostsim", "build.sh")])
    import deft4j_amd as D
    L = D.load_library(os.path.join(ROOT, "tests", "hostsim", "libdeft4g_hostsim.so"))
    D.init(0, lib=L)
    return L


def test_small_gzip_files_in_the_emulator(sim):
    check_files(sim, [f for f in FILES if f["stem"] in ("deflate-store-2.txt.gz", "lz-twice-twice.txt.gz", "text.png")])
    check_zlib_and_raw(sim)


def test_zip_archive_in_the_emulator(sim):
    check_zip(sim)


def test_checksum_kernels_in_the_emulator(sim):
    import deft4j_amd as D
    raws = [synth.reptext(n, s) for n, s in ((5000, 1), (1, 2), (70000, 3), (2048, 4), (4097, 5))] + [b""]
    b = D.Batch([synth.deflate9(r) for r in raws], lib=sim).parse()
    for i, r in enumerate(raws):
        assert b.checksums(i) == (zlib.crc32(r), zlib.adler32(r), len(r))
    b.close()


@pytest.mark.gpu
def test_all_fixture_files_on_the_gpu():
    import deft4j_amd as D
    D.init(0)
    check_files(None, FILES)
    check_zlib_and_raw(None)
    check_zip(None)
    raws = [synth.reptext(n, s) for n, s in ((1 << 20, 7), (12345, 8), (3 << 20, 9))]
    b = D.Batch([synth.deflate9(r) for r in raws]).parse()
    for i, r in enumerate(raws):
        assert b.checksums(i) == (zlib.crc32(r), zlib.adler32(r), len(r))
    b.close()
