"""The oracle's restatement of pyEXP's sub-sample coefficient covariance (Spherical::accumulate with
pcavar, expui/BiorthBasis.cc:583-665) against an independent numpy statement built from the
oracle's coefficient accumulation.  CPU only."""
import numpy as np
import pytest

from tests.conftest import make_grid


def test_subsample_means_and_covariances(oracle):
    model, g = make_grid("plummer", 3, 6, 400)
    rng = np.random.default_rng(12)
    n, sampT = 2000, 7
    pos = rng.standard_normal((n, 3)) * 2.0
    pos[:40] *= 60.0                                   # some particles beyond rmax: not counted
    m = rng.uniform(0.5, 1.5, n) / n
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    cv = oracle.pyexp_sph_covariance(g, prm, pos, m, sampT)
    r = np.linalg.norm(pos, axis=1)
    ok = (r >= g.rmin) & (r <= g.rmax)
    assert cv["used"] == ok.sum() and cv["counts"].sum() == ok.sum()
    # sub-sample index: running count of accepted particles, incremented BEFORE the modulus
    rank = np.cumsum(ok)
    T = rank % sampT
    for t in range(sampT):
        sel = ok & (T == t)
        assert cv["counts"][t] == sel.sum()
        assert cv["masses"][t] == pytest.approx(m[sel].sum(), rel=1e-13)
        # the sub-sample mean is the sub-sample's coefficient set in complex (l, m>=0) packing
        coef, _ = oracle.sph_accumulate(g, prm, pos[sel], m[sel])
        L = off = 0
        for l in range(g.lmax + 1):
            for mm in range(l + 1):
                want = coef[off] + (1j * coef[off + 1] if mm else 0.0)
                off += 2 if mm else 1
                assert np.abs(cv["mean"][t, L] - want).max() <= 1e-12 * np.abs(coef).max()
                L += 1
    # covariance: sum of mass * v v^T per particle, symmetric positive semi-definite
    c0 = cv["covr"][3, 0]
    assert np.allclose(c0, c0.T, rtol=0, atol=1e-18) and np.linalg.eigvalsh(c0).min() > -1e-12 * np.abs(c0).max()
    # one particle: covr = outer(g, conj g) * mass with g = mean / mass
    one = oracle.pyexp_sph_covariance(g, prm, pos[100:101], m[100:101], sampT)
    t = int(np.flatnonzero(one["counts"])[0])
    assert t == 1                                       # used = 1 -> T = 1 % sampT
    for L in range(one["mean"].shape[1]):
        gvec = one["mean"][t, L] / m[100]
        assert np.allclose(one["covr"][t, L], np.real(np.outer(gvec, np.conj(gvec))) * m[100],
                           rtol=1e-12, atol=1e-300)
    # continuing an accumulation keeps the running count
    a = oracle.pyexp_sph_covariance(g, prm, pos[:1000], m[:1000], sampT)
    a = oracle.pyexp_sph_covariance(g, prm, pos[1000:], m[1000:], sampT, acc=a)
    assert np.array_equal(a["counts"], cv["counts"]) and np.allclose(a["covr"], cv["covr"], rtol=1e-13, atol=0)
