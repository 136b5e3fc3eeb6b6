"""EXP's SLGridSph HDF5 cache format (exputil/SLGridMP2.cc:490-696) through the HDF5 C library:
write -> inspect with the HDF5 tools -> read back; the older [numr][nmax] matrix layout is
transposed on reading.  CPU only; skipped where the shim could not be built (no hdf5.h)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import make_grid


@pytest.fixture(scope="module")
def h5():
    from exp_amd import h5cache
    if not h5cache.available():
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run(["make", "-s", "h5"], cwd=root, check=False)
    if not h5cache.available():
        pytest.skip("HDF5 C headers/library not available")
    return h5cache


def test_slgrid_cache_roundtrip_and_layout(h5, tmp_path):
    model, g = make_grid("plummer", 4, 8, 400)
    path = str(tmp_path / "SLGridSph.cache.run0")
    h5.write_slgrid_cache(path, g, "SLGridSph.model")
    hdr = h5.read_slgrid_header(path)
    assert hdr["geometry"] == "sphere" and hdr["forceID"] == "SLGridSph" and hdr["version"] == "1.0"
    assert hdr["model"] == "SLGridSph.model"
    assert (hdr["lmax"], hdr["nmax"], hdr["numr"], hdr["cmap"], hdr["diverge"]) == (4, 8, 400, 1, 0)
    assert hdr["rmin"] == g.rmin and hdr["rmax"] == g.rmax and hdr["rmapping"] == g.rmap
    back = h5.read_slgrid_cache(path, model, check={"lmax": 4, "nmax": 8, "numr": 400,
                                                    "rmapping": g.rmap, "model": "SLGridSph.model"})
    for k in ("ev", "ef", "xi", "r", "p0", "d0"):
        assert np.array_equal(getattr(back, k), getattr(g, k)), k
    assert back.dxi == g.dxi and back.xmin == g.xmin
    with pytest.raises(RuntimeError):
        h5.read_slgrid_cache(path, model, check={"nmax": 9})
    # structure as the reference writes it: Harmonic/<l>/{ev [nmax], ef [nmax, numr]}
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'GROUP "Harmonic"' in txt and 'GROUP "4"' in txt
        assert "DATASPACE  SIMPLE { ( 8, 400 ) / ( 8, 400 ) }" in txt
        assert 'ATTRIBUTE "rmapping"' in txt and "H5T_VARIABLE" in txt and "H5T_CSET_UTF8" in txt


def test_old_matrix_layout_is_transposed(h5, tmp_path):
    """Caches written through the older HighFive Eigen path hold ef as [numr][nmax] (what the
    "Version" attribute guards against, exputil/SLGridMP2.cc:550-560): the reader accepts both,
    and refuses shapes that are neither."""
    model, g = make_grid("plummer", 4, 8, 400)
    path = str(tmp_path / "old.cache")
    h5.write_slgrid_cache(path, g, "m", old_layout=True)
    back = h5.read_slgrid_cache(path, model)
    assert np.array_equal(back.ef, g.ef) and np.array_equal(back.ev, g.ev)
    import ctypes
    ev = np.zeros((g.lmax + 1, 7)); ef = np.zeros((g.lmax + 1, 7, 400))
    rc = h5._load().exp_h5_slgrid_read_tables(path.encode(), g.lmax, 7, 400,
                                              ev.ctypes.data_as(ctypes.c_void_p),
                                              ef.ctypes.data_as(ctypes.c_void_p))
    assert rc != 0
