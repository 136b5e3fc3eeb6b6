import os
import sys

# some GPU boxes expose hundreds of cores under a small CPU quota: unbounded BLAS/OpenMP pools then
# spin against each other and the (numpy-side) table builds run 20x slower
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_lib import Oracle
    return Oracle()


_GRID_CACHE = {}


def make_grid(kind: str, lmax: int, nmax: int, numr: int = 800):
    """Session-cached SL grids (built by exp_amd.slgrid, a second or two each)."""
    from exp_amd.models import NFWModel, PlummerModel
    from exp_amd.slgrid import build_slgrid
    key = (kind, lmax, nmax, numr)
    if key not in _GRID_CACHE:
        if kind == "plummer":
            model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
            g = build_slgrid(model, lmax, nmax, numr=numr, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0,
                             nel=32, P=8)
        elif kind == "nfw":
            model = NFWModel(1.0, 20.0, 6.0, 1e-3, 50.0)
            g = build_slgrid(model, lmax, nmax, numr=numr, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0,
                             nel=32, P=8)
        elif kind == "plummer_log":
            model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
            g = build_slgrid(model, lmax, nmax, numr=numr, rmin=1e-3, rmax=49.5, cmap=2, rmap=1.0,
                             nel=32, P=8)
        else:
            raise KeyError(kind)
        _GRID_CACHE[key] = (model, g)
    return _GRID_CACHE[key]


@pytest.fixture(scope="session")
def plummer_s6():
    return make_grid("plummer", 6, 18, 800)


@pytest.fixture(scope="session")
def plummer_small():
    return make_grid("plummer", 4, 8, 400)
