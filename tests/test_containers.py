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
    raws = [synth.reptext(n, s) for n, s in ((1 << 20, 7), (12345, 8), (3 << 20, 9))]
    b = D.Batch([synth.deflate9(r) for r in raws]).parse()
    for i, r in enumerate(raws):
        assert b.checksums(i) == (zlib.crc32(r), zlib.adler32(r), len(r))
    b.close()
