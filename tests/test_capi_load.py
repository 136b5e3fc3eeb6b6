"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/exp_amd.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "exp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(exp_amd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from exp_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/exp_amd.h but not exported"
    # and the Python binding types every one of them
    assert set(syms) == set(_lib.SIGNATURES.keys())
    assert lib.exp_amd_abi_version() == 1


def test_no_cpu_fallback_without_device():
    import ctypes
    from exp_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.exp_amd_ctx_create(0, None, ctypes.byref(h))
    if rc == 0:                      # a GPU is present (GPU box): nothing to assert here
        lib.exp_amd_ctx_destroy(h)
        pytest.skip("HIP device present")
    assert rc == 4                   # EXP_AMD_ERR_NODEVICE
    msg = lib.exp_amd_last_global_error().decode()
    assert "no CPU fallback" in msg
    from exp_amd.runtime import Context
    with pytest.raises(_lib.ExpAmdError):
        Context(0)


def test_product_package_does_not_touch_the_oracle():
    """exp_amd/ must never import, link or call anything under oracle/ (or tests/)."""
    pkg = os.path.join(ROOT, "exp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower(), (dirpath, f)
