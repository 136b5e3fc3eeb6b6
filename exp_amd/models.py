"""Spherical / disk background models and synthetic initial conditions (host, init only).

* ``SphericalModel`` plays the role of EXP's ``AxiSymModel``/``SphericalModelTable``
  (``exputil/massmodel.cc:266-349``): it supplies rho0(r), M(r), Phi0(r), Phi0'(r)
  to the Sturm-Liouville table builder.  ``TableModel`` reads the reference's
  4-column model-file format (``tests/Halo/SLGridSph.model:1-4``: ``!``/``#``
  comments, row count, rows of ``r rho M Phi``) and can write it.
* ``sample_sphere`` / ``sample_disk`` draw seeded synthetic particle sets
  (Plummer / NFW / exponential disk) for the parity tests and the benchmark.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
from scipy.interpolate import CubicSpline
from scipy.special import erf


class SphericalModel:
    """rho0, M, Phi0 on [rmin, rmax] (G = 1)."""

    rmin: float
    rmax: float

    def dens(self, r):  # pragma: no cover - interface
        raise NotImplementedError

    def mass(self, r):  # pragma: no cover - interface
        raise NotImplementedError

    def pot(self, r):  # pragma: no cover - interface
        raise NotImplementedError

    def dpot(self, r):
        r = np.asarray(r, dtype=np.float64)
        return self.mass(r) / (r * r)

    # -- model-file I/O (reference format) -----------------------------------------
    def write_table(self, path: str, num: int = 1000, header: str = "") -> None:
        r = np.exp(np.linspace(math.log(self.rmin), math.log(self.rmax), num))
        with open(path, "w") as f:
            f.write(f"! {header}\n! 1) = r   2) = rho   3) = M(r)   4) U(r)\n")
            f.write(f"{num:10d}\n")
            for a, b, c, d in zip(r, self.dens(r), self.mass(r), self.pot(r)):
                f.write(f"{a:20.12e}{b:20.12e}{c:20.12e}{d:20.12e}\n")


class PlummerModel(SphericalModel):
    """Plummer sphere, total mass ``M``, scale ``b``."""

    def __init__(self, b: float = 1.0, M: float = 1.0, rmin: float = 1e-3, rmax: float = 50.0):
        self.b, self.M, self.rmin, self.rmax = b, M, rmin, rmax

    def dens(self, r):
        r = np.asarray(r, dtype=np.float64)
        return 3.0 * self.M / (4.0 * math.pi * self.b ** 3) * (1.0 + (r / self.b) ** 2) ** -2.5

    def mass(self, r):
        r = np.asarray(r, dtype=np.float64)
        return self.M * r ** 3 / (r * r + self.b * self.b) ** 1.5

    def pot(self, r):
        r = np.asarray(r, dtype=np.float64)
        return -self.M / np.sqrt(r * r + self.b * self.b)


class NumericModel(SphericalModel):
    """Model defined by a density profile; M and Phi by quadrature on a log grid."""

    def __init__(self, dens_fn, rmin: float, rmax: float, ngrid: int = 4001,
                 total_mass: Optional[float] = 1.0):
        self.rmin, self.rmax = rmin, rmax
        # integrate from well inside rmin so that M(rmin) is right
        t = np.linspace(math.log(rmin * 1e-4), math.log(rmax), ngrid)
        r = np.exp(t)
        rho = dens_fn(r)
        # dM = 4 pi rho r^3 dt ;  dPsi = 4 pi rho r^2 dt  (Psi = int_r^inf 4 pi rho r' dr')
        fm = CubicSpline(t, 4.0 * math.pi * rho * r ** 3)
        fp = CubicSpline(t, 4.0 * math.pi * rho * r ** 2)
        Mt = fm.antiderivative()(t)
        Mt += 4.0 * math.pi * rho[0] * r[0] ** 3 / 3.0      # inner constant-density cap
        Pt = fp.antiderivative()(t)
        Psi = Pt[-1] - Pt                                   # outer part (rho = 0 beyond rmax)
        norm = 1.0 if total_mass is None else total_mass / Mt[-1]
        self._norm = norm
        self._dens_fn = dens_fn
        self._M = CubicSpline(t, Mt * norm)
        self._Phi = CubicSpline(t, (-Mt / r - Psi) * norm)
        self._t, self._r, self._Mt = t, r, Mt * norm

    def dens(self, r):
        return self._norm * self._dens_fn(np.asarray(r, dtype=np.float64))

    def mass(self, r):
        return self._M(np.log(np.asarray(r, dtype=np.float64)))

    def pot(self, r):
        return self._Phi(np.log(np.asarray(r, dtype=np.float64)))


def NFWModel(rs: float = 1.0, rtrunc: float = 20.0, wtrunc: float = 6.0,
             rmin: float = 1e-3, rmax: float = 50.0, total_mass: float = 1.0) -> NumericModel:
    """NFW profile with an error-function truncation (the form used by EXP's gensph
    model headers, cf. ``tests/Halo/SLGridSph.model:1``: rtrunc / wtrunc)."""

    def rho(r):
        x = r / rs
        return 1.0 / (x * (1.0 + x) ** 2) * 0.5 * (1.0 - erf((r - rtrunc) / wtrunc))

    return NumericModel(rho, rmin, rmax, total_mass=total_mass)


class TableModel(SphericalModel):
    """The reference's 4-column model file, spline-interpolated in log r."""

    def __init__(self, path: str):
        rows = []
        n = None
        with open(path) as f:
            for line in f:
                s = line.strip()
                if not s or s[0] in "!#":
                    continue
                if n is None:
                    n = int(s.split()[0])
                    continue
                rows.append([float(v) for v in s.split()[:4]])
        a = np.asarray(rows[:n], dtype=np.float64)
        self.r_tab, self.d_tab, self.m_tab, self.p_tab = a.T
        self.rmin, self.rmax = float(a[0, 0]), float(a[-1, 0])
        t = np.log(self.r_tab)
        self._d = CubicSpline(t, np.log(np.maximum(self.d_tab, 1e-300)))
        self._m = CubicSpline(t, self.m_tab)
        self._p = CubicSpline(t, self.p_tab)

    def dens(self, r):
        return np.exp(self._d(np.log(np.asarray(r, dtype=np.float64))))

    def mass(self, r):
        return self._m(np.log(np.asarray(r, dtype=np.float64)))

    def pot(self, r):
        return self._p(np.log(np.asarray(r, dtype=np.float64)))

    def dpot(self, r):
        r = np.asarray(r, dtype=np.float64)
        return self._p(np.log(r), 1) / r


# ----------------------------------------------------------------------------------
# synthetic particle sets
# ----------------------------------------------------------------------------------

def sphere_sampling_tables(model: SphericalModel, rlim: Optional[float] = None, ntab: int = 8192):
    """(u, r(u), sigma(r)) tables: inverse mass CDF and isotropic Jeans dispersion."""
    rlim = float(rlim or model.rmax)
    r = np.exp(np.linspace(math.log(model.rmin), math.log(rlim), ntab))
    M = model.mass(r)
    M = np.maximum.accumulate(M)
    u = (M - M[0]) / (M[-1] - M[0])
    # sigma_r^2 = (1/rho) int_r^inf rho M / r'^2 dr'
    rho = model.dens(r)
    integrand = rho * M / r            # d(log r) measure: rho M / r^2 * r
    t = np.log(r)
    F = CubicSpline(t, integrand).antiderivative()(t)
    sig2 = (F[-1] - F) / np.maximum(rho, 1e-300)
    sig2[-1] = sig2[-2]
    return u, r, np.sqrt(np.maximum(sig2, 0.0))


def sample_sphere(model: SphericalModel, n: int, seed: int, rlim: Optional[float] = None,
                  velocities: bool = True) -> Tuple[np.ndarray, ...]:
    """Equal-mass isotropic sphere drawn from ``model``: returns (m, pos[n,3], vel[n,3])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u_tab, r_tab, s_tab = sphere_sampling_tables(model, rlim)
    u = rng.random(n)
    r = np.interp(u, u_tab, r_tab)
    ct = rng.uniform(-1.0, 1.0, n)
    ph = rng.uniform(0.0, 2.0 * math.pi, n)
    st = np.sqrt(1.0 - ct * ct)
    pos = np.stack([r * st * np.cos(ph), r * st * np.sin(ph), r * ct], axis=1)
    if velocities:
        sig = np.interp(r, r_tab, s_tab)
        vel = rng.standard_normal((n, 3)) * sig[:, None]
    else:
        vel = np.zeros((n, 3))
    mtot = float(model.mass(r_tab[-1]) - model.mass(r_tab[0]))
    m = np.full(n, mtot / n)
    return m, pos, vel


def sample_disk(n: int, seed: int, a: float = 0.01, h: float = 0.001, mass: float = 1.0,
                rmax_over_a: float = 12.0, vcirc=None) -> Tuple[np.ndarray, ...]:
    """Exponential disk Sigma ~ exp(-R/a), sech^2(z/h) vertical profile
    (the reference's conditioning density, ``src/Cylinder.cc:315-322``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    # inverse CDF of 1-(1+x)exp(-x) by table
    x = np.linspace(0.0, rmax_over_a, 16384)
    cdf = 1.0 - (1.0 + x) * np.exp(-x)
    cdf /= cdf[-1]
    R = a * np.interp(rng.random(n), cdf, x)
    ph = rng.uniform(0.0, 2.0 * math.pi, n)
    uz = rng.uniform(1e-12, 1.0 - 1e-12, n)
    z = h * np.arctanh(2.0 * uz - 1.0)
    pos = np.stack([R * np.cos(ph), R * np.sin(ph), z], axis=1)
    vel = np.zeros((n, 3))
    if vcirc is not None:
        vc = vcirc(np.maximum(R, 1e-12))
        vel[:, 0] = -vc * np.sin(ph)
        vel[:, 1] = vc * np.cos(ph)
        vel += rng.standard_normal((n, 3)) * (0.1 * vc)[:, None]
    m = np.full(n, mass / n)
    return m, pos, vel
