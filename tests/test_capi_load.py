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


def test_host_only_entry_points_and_their_argument_checks():
    """The two host loops of the C-ABI need no device: the float bin sums (bins outside [0, nbins) skipped, float
    accumulator + double addend in the order given) and the PSP record unpacking; bad arguments return EXP_AMD_ERR_ARG (1)."""
    import ctypes
    import numpy as np
    from exp_amd import _lib
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    bins = np.array([0, 2, 2, -1, 5, 2, 0], dtype=np.int32)
    vals = np.array([1.0, 1e-9, 1.0, 7.0, 7.0, 2.0 ** -30, 0.5])
    out = np.zeros(3, dtype=np.float32)
    assert lib.exp_amd_host_binsum_f32(len(bins), vp(bins), vp(vals), 3, vp(out)) == 0
    want2 = np.float32(np.float64(np.float32(np.float64(np.float32(1e-9)) + 1.0)) + 2.0 ** -30)
    assert out.tolist() == [1.5, 0.0, float(want2)]
    assert lib.exp_amd_host_binsum_f32(-1, vp(bins), vp(vals), 3, vp(out)) == 1
    assert lib.exp_amd_host_binsum_f32(3, None, vp(vals), 3, vp(out)) == 1
    assert lib.exp_amd_host_binsum_f32(0, None, None, 0, None) == 0
    # two float records with an index, one integer and one real attribute, 44 bytes each, read with a stride of one record
    rec = np.zeros(2, dtype=[("i", "<u8"), ("r", "<f4", (8,)), ("ia", "<i4"), ("da", "<f4")])
    rec["i"], rec["ia"], rec["da"] = [7, 9], [-3, 4], [0.5, 0.25]
    rec["r"] = np.arange(16, dtype=np.float32).reshape(2, 8)
    indx, mass, pos, vel, pot = np.zeros(2, np.uint64), np.zeros(2), np.zeros((2, 3)), np.zeros((2, 3)), np.zeros(2)
    ia, da = np.zeros((2, 1), np.int32), np.zeros((2, 1))
    args = (vp(indx), vp(mass), vp(pos), vp(vel), vp(pot), vp(ia), vp(da))
    assert lib.exp_amd_host_psp_unpack(2, vp(rec), rec.dtype.itemsize, 4, 1, 1, 1, *args) == 0
    assert indx.tolist() == [7, 9] and mass.tolist() == [0.0, 8.0] and pos[1].tolist() == [9.0, 10.0, 11.0]
    assert vel[0].tolist() == [4.0, 5.0, 6.0] and pot.tolist() == [7.0, 15.0] and ia[:, 0].tolist() == [-3, 4] and da[:, 0].tolist() == [0.5, 0.25]
    assert lib.exp_amd_host_psp_unpack(2, vp(rec), rec.dtype.itemsize, 6, 1, 1, 1, *args) == 1      # reals are 4 or 8 bytes
    assert lib.exp_amd_host_psp_unpack(2, vp(rec), 40, 4, 1, 1, 1, *args) == 1                       # a record cannot be shorter than its fields
    assert lib.exp_amd_host_psp_unpack(2, None, 44, 4, 1, 1, 1, *args) == 1
