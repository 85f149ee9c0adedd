"""The C-ABI library loads and exports every symbol include/fpx.h declares (no compute without a GPU)."""
import os
import re

from fpx_testlib import ROOT, fpx


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fpx_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = fpx.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    from acoustid_index_amd import _lib
    for name in declared:
        assert hasattr(lib, name), f"libfpx.so does not export {name}"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_error_strings():
    lib = fpx.lib()
    assert lib.fpx_version() == 2
    assert lib.fpx_strerror(0) == b"ok"
    assert lib.fpx_strerror(-2) == b"search timeout"


def test_no_gpu_means_loud_failure():
    """Without a gfx950 device the context cannot be created: there is no CPU path to fall back to."""
    import ctypes as C
    import pytest
    h = C.c_void_p()
    rc = fpx.lib().fpx_ctx_create(0, C.byref(h))
    if rc == 0:
        fpx.lib().fpx_ctx_destroy(h)
        pytest.skip("a GPU is visible here")
    assert rc == -5 and b"no CPU fallback" in fpx.lib().fpx_last_error()
    with pytest.raises(fpx.FpxError):
        fpx.Context(0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under acoustid-index_amd/ may reference it."""
    pkg = os.path.join(ROOT, "acoustid-index_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libfpx_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
                if f.endswith((".hip", ".h", ".hpp", ".cpp")):
                    assert "fpx_oracle.h" not in text and "orc_" not in text.replace("orc_synth_hash", ""), f
