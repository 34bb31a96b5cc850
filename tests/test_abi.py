"""CPU-side checks of the drop-in boundary: libdeft4g.so loads, exports every symbol include/deft4g.h
declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import deft4j_amd
    return deft4j_amd.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "deft4g.h")).read()
    names = sorted(set(re.findall(r"\b(d4g_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 14
    for n in names:
        assert getattr(lib, n) is not None, n
    import deft4j_amd
    assert sorted(deft4j_amd.EXPORTS) == names


def test_no_cpu_fallback_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import deft4j_amd
    rc = lib.d4g_init(0)
    assert rc < 0
    assert b"no CPU fallback" in lib.d4g_last_error() or b"device" in lib.d4g_last_error()
    with pytest.raises(RuntimeError):
        deft4j_amd.Deft.optimiseDeflateStream(b"\x03\x00")
    with pytest.raises(RuntimeError):
        deft4j_amd.zopfli_streams([b"abc"], 3)           # the Zopfli encoder has no CPU path either
    # the raw entry points also refuse
    arr = (ctypes.c_char_p * 1)(b"\x03\x00")
    lens = (ctypes.c_size_t * 1)(2)
    assert not lib.d4g_batch_create(1, arr, lens)


def test_product_library_does_not_link_the_oracle():
    out = os.popen("nm -D --defined-only %s" % os.path.join(ROOT, "deft4j_amd", "libdeft4g.so")).read()
    assert "oracle_" not in out
    src = "".join(open(os.path.join(ROOT, "deft4j_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "deft4j_amd", "csrc")))
    assert "oracle" not in src.lower() or "deft_oracle" not in src
    py = open(os.path.join(ROOT, "deft4j_amd", "__init__.py")).read() + open(os.path.join(ROOT, "deft4j_amd", "shard.py")).read()
    assert "oracle" not in py


def test_jni_shim_and_java_binding_match_the_header():
    """jni/deft4g_jni.c and java/.../NativeDeft.java cannot be compiled here (no JDK), so keep them honest textually:
    every d4g_ function the shim calls is declared in include/deft4g.h, and every native method of NativeDeft has its
    Java_..._NativeDeft_<name> definition in the shim."""
    hdr = open(os.path.join(ROOT, "include", "deft4g.h")).read()
    shim = open(os.path.join(ROOT, "jni", "deft4g_jni.c")).read()
    java = open(os.path.join(ROOT, "java", "com", "github", "NeRdTheNed", "deft4j", "NativeDeft.java")).read()
    declared = set(re.findall(r"\b(d4g_[a-z_0-9]+)\s*\(", hdr))
    for f in set(re.findall(r"\b(d4g_[a-z_0-9]+)\s*\(", shim)):
        assert f in declared, f
    natives = re.findall(r"native\s+[\w\[\]]+\s+(\w+)\s*\(", java)
    assert len(natives) >= 6
    for m in natives:
        assert "JFN(%s)" % m in shim, m
