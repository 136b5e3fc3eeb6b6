"""Two self-contained headers of the REFERENCE that sit on the path, compiled where they lie into
oracle/_ref/libref_headers.so (oracle/ref/Makefile, oracle/ref/headers_driver.cc):

* include/QuadLS.H -- the quadratic least squares behind Orient's pseudo-acceleration estimate
  (include/PseudoAccel.H:45-91): pins oracle/bfe_oracle.c:orc_quadls bit for bit, and through it the
  device estimator (tests/test_orient_gpu.py compares the device with the oracle);
* include/coef.H -- SphCoefHeader / CylCoefHeader, the headers of the legacy native coefficient streams:
  pins the struct formats exp_amd/coefs.py reads and writes.

CPU only; skipped where neither the reference tree nor a prebuilt library exists."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_headers.so")
REF = "/root/reference/include/QuadLS.H"


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(LIB):
        if not os.path.exists(REF):
            pytest.skip("no reference tree and no prebuilt oracle/_ref/libref_headers.so")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref")], check=True)
    return ctypes.CDLL(LIB)


def _ref_quadls(lib, x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(3)
    lib.ref_quadls(ctypes.c_int(len(x)), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                   out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_oracle_quadls_is_the_references(ref, oracle):
    rng = np.random.default_rng(12)
    for n in (3, 4, 5, 8, 16, 64):
        for _ in range(20):
            t = np.sort(rng.uniform(0.0, 3.0, n)) + rng.uniform(-1.0, 50.0)
            y = rng.normal() * t * t + rng.normal() * t + rng.normal() + 1e-3 * rng.normal(size=n)
            assert np.array_equal(oracle.quadls(t, y), _ref_quadls(ref, t, y)), n
    # exact on a quadratic, zero for a degenerate abscissa (the reference's fabs(denom) > 0 guard), n = 0
    t = np.array([0.0, 1.0, 2.0, 3.0])
    assert np.allclose(_ref_quadls(ref, t, 2 * t * t - 3 * t + 0.5), [2.0, -3.0, 0.5], atol=1e-12)
    same = np.full(5, 1.25)
    assert np.array_equal(oracle.quadls(same, np.arange(5.0)), _ref_quadls(ref, same, np.arange(5.0)))
    assert np.array_equal(_ref_quadls(ref, same, np.arange(5.0)), [0.0, 0.0, 0.0])


def test_pseudo_accel_fit_from_the_references_quadls(ref, oracle):
    """PseudoAccel::operator() (include/PseudoAccel.H:45-91) restated on the REFERENCE's QuadLS: accel =
    2a of the centre fits, omega = n x dn/dt, domega/dt = n x d2n/dt2 of the axis fits at the last time."""
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(0.01, 0.03, 7))
    rows = np.column_stack([t, rng.normal(size=(7, 3)) * 0.01 + np.outer(t * t, [0.5, -0.2, 0.1]),
                            rng.normal(size=(7, 3)) * 0.01 + [0.0, 0.0, 1.0]])
    acc, om, dom = oracle.pseudo_accel_fit(rows)
    fits = [_ref_quadls(ref, t, rows[:, 1 + k]) for k in range(6)]
    assert np.array_equal(acc, [2.0 * fits[k][0] for k in range(3)])
    T = t[-1]
    n = np.array([f[0] * T * T + f[1] * T + f[2] for f in fits[3:]])
    dn = np.array([2.0 * f[0] * T + f[1] for f in fits[3:]])
    d2n = np.array([2.0 * f[0] for f in fits[3:]])
    assert np.allclose(om, np.cross(n, dn), rtol=0, atol=1e-15 * np.abs(dn).max())
    assert np.allclose(dom, np.cross(n, d2n), rtol=0, atol=1e-15 * np.abs(d2n).max())


def test_native_coefficient_headers_have_the_references_layout(ref):
    from exp_amd import coefs
    out = (ctypes.c_long * 10)()
    ref.ref_coef_layout(out)
    sph, cyl = list(out[:6]), list(out[6:])
    # SphCoefHeader {char id[64]; double tnow, scale; int nmax, Lmax;}  ->  "<64sddii"
    assert sph == [coefs._LEGACY.size, 0, 64, 72, 80, 84]
    # CylCoefHeader {double time; int mmax, nmax;}  ->  "<dii"
    assert cyl == [coefs._LEGACY_CYL.size, 0, 8, 12]
