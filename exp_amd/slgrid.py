"""Sturm-Liouville radial basis tables for the spherical BFE (init-time, host only).

Builds what EXP's ``SLGridSph`` holds (``include/SLGridMP2.H:28``,
``include/sltableMP2.H:16-24``): for every harmonic order ``l`` the first ``nmax``
eigenpairs of

    -(p u')' + q u = lambda w u,   p = r^2 Phi0^2,  q = (l(l+1) Phi0 - rho r^2) Phi0,
                                   w = -rho r^2 Phi0,   rho = 4 pi rho0

(``exputil/SLGridMP2.cc:3647-3654``) with the boundary conditions of
``exputil/SLGridMP2.cc:1153-1164`` tabulated on the xi-uniform grid of
``SLGridSph::init_table`` (``exputil/SLGridMP2.cc:1355-1382``), eigenfunction sign
fixed as in ``:1329-1340`` (``nevsign = 4``, ``exputil/libvars.cc:38``) and
normalised so that ``orthoCheck`` (``:1775-1824``) returns the identity.

The reference solves the ODE with the Fortran code SLEDGE, which cannot be built in
this image.  This module is an independent solver: an hp finite-element Galerkin
discretisation in ``log r`` (integrated-Legendre shape functions, Robin boundary
terms in the weak form, symmetric generalised eigenproblem via LAPACK).  The hot path
only consumes the resulting tables; it never depends on how they were made.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np
from numpy.polynomial import legendre as npleg
import scipy.linalg as sla
from scipy.interpolate import CubicSpline

from .models import SphericalModel

try:                                   # keep LAPACK/BLAS from oversubscribing big hosts
    from threadpoolctl import threadpool_limits
except Exception:                      # pragma: no cover
    import contextlib

    def threadpool_limits(limits=None):
        return contextlib.nullcontext()



def blas_limit(cap: int = 2):
    """Context manager capping the BLAS/LAPACK pools for the (small) table-build eigenproblems.
    Only ever LOWERS the thread count: raising an OpenBLAS pool above the size it was initialised
    with can crash, and hosts that expose hundreds of cores under a small CPU quota make
    unbounded pools spin against each other."""
    try:
        import threadpoolctl
        cur = [int(i.get("num_threads", 1)) for i in threadpoolctl.threadpool_info()]
        if cur and max(cur) > cap:
            return threadpool_limits(limits=cap)
    except Exception:   # pragma: no cover
        pass
    import contextlib
    return contextlib.nullcontext()


NEVSIGN = 4  # exputil/libvars.cc:38


@dataclasses.dataclass
class SLGridSph:
    """Host-side copy of the SLGridSph tables (all float64, C-contiguous)."""

    lmax: int
    nmax: int
    numr: int
    cmap: int
    rmin: float
    rmax: float
    rmap: float
    xmin: float
    xmax: float
    dxi: float
    xi: np.ndarray   # [numr]
    r: np.ndarray    # [numr]
    p0: np.ndarray   # [numr]
    d0: np.ndarray   # [numr]  (4 pi rho0)
    ev: np.ndarray   # [lmax+1, nmax]
    ef: np.ndarray   # [lmax+1, nmax, numr]

    # -- coordinate maps: exputil/SLGridMP2.cc:711-765 ------------------------------
    def r_to_xi(self, r):
        r = np.asarray(r, dtype=np.float64)
        if self.cmap == 1:
            return (r / self.rmap - 1.0) / (r / self.rmap + 1.0)
        if self.cmap == 2:
            return np.log(r)
        return r

    def xi_to_r(self, xi):
        xi = np.asarray(xi, dtype=np.float64)
        if self.cmap == 1:
            return (1.0 + xi) / (1.0 - xi) * self.rmap
        if self.cmap == 2:
            return np.exp(xi)
        return xi

    def save(self, path: str) -> None:
        np.savez_compressed(path, **{f.name: getattr(self, f.name)
                                     for f in dataclasses.fields(self)})

    @staticmethod
    def load(path: str) -> "SLGridSph":
        z = np.load(path)
        kw = {}
        for f in dataclasses.fields(SLGridSph):
            v = z[f.name]
            kw[f.name] = v.item() if v.ndim == 0 else np.ascontiguousarray(v, dtype=np.float64)
        for k in ("lmax", "nmax", "numr", "cmap"):
            kw[k] = int(kw[k])
        return SLGridSph(**kw)


def _xi_grid(cmap: int, rmin: float, rmax: float, rmap: float, numr: int):
    """exputil/SLGridMP2.cc:1355-1382"""
    if cmap == 1:
        xmin = (rmin / rmap - 1.0) / (rmin / rmap + 1.0)
        xmax = (rmax / rmap - 1.0) / (rmax / rmap + 1.0)
    elif cmap == 2:
        xmin, xmax = math.log(rmin), math.log(rmax)
    else:
        xmin, xmax = rmin, rmax
    dxi = (xmax - xmin) / (numr - 1)
    xi = xmin + dxi * np.arange(numr, dtype=np.float64)
    if cmap == 1:
        r = (1.0 + xi) / (1.0 - xi) * rmap
    elif cmap == 2:
        r = np.exp(xi)
    else:
        r = xi.copy()
    return xmin, xmax, dxi, xi, r


# ----------------------------------------------------------------------------------
# hp-FEM machinery (integrated Legendre shape functions on [-1, 1])
# ----------------------------------------------------------------------------------

def _shape(s: np.ndarray, P: int):
    """Values and s-derivatives of the P+1 shape functions at points s.

    N_0 = (1-s)/2, N_1 = (1+s)/2, N_k = (P_k - P_{k-2}) / sqrt(2(2k-1)) for k>=2
    (vanish at both ends), N_k' = sqrt((2k-1)/2) P_{k-1}.
    """
    s = np.asarray(s, dtype=np.float64)
    V = npleg.legvander(s, P)            # [npts, P+1] Legendre P_0..P_P
    N = np.empty((s.size, P + 1))
    dN = np.empty((s.size, P + 1))
    N[:, 0] = 0.5 * (1.0 - s)
    N[:, 1] = 0.5 * (1.0 + s)
    dN[:, 0] = -0.5
    dN[:, 1] = 0.5
    for k in range(2, P + 1):
        N[:, k] = (V[:, k] - V[:, k - 2]) / math.sqrt(2.0 * (2 * k - 1))
        dN[:, k] = math.sqrt((2 * k - 1) / 2.0) * V[:, k - 1]
    return N, dN


class _FEMesh:
    def __init__(self, ta: float, tb: float, nel: int, P: int, nq: int):
        self.ta, self.tb, self.nel, self.P = ta, tb, nel, P
        self.edges = np.linspace(ta, tb, nel + 1)
        self.h = np.diff(self.edges)
        sq, wq = npleg.leggauss(nq)
        self.sq, self.wq = sq, wq
        self.Nq, self.dNq = _shape(sq, P)
        # global dof map: vertex v -> v ; bubble k (2..P) of element e -> nel+1 + e*(P-1) + (k-2)
        self.ndof = nel + 1 + nel * (P - 1)
        self.map = np.empty((nel, P + 1), dtype=np.int64)
        for e in range(nel):
            self.map[e, 0] = e
            self.map[e, 1] = e + 1
            self.map[e, 2:] = nel + 1 + e * (P - 1) + np.arange(P - 1)
        # quadrature points in t for all elements: [nel, nq]
        self.tq = self.edges[:-1, None] + 0.5 * (sq[None, :] + 1.0) * self.h[:, None]

    def assemble(self, cK: np.ndarray, cQ: np.ndarray, cW: np.ndarray):
        """Assemble  K = int cK u_t v_t + cQ u v dt,  M = int cW u v dt.

        cK, cQ, cW are the t-space coefficient functions sampled at self.tq."""
        n = self.ndof
        K = np.zeros((n, n))
        M = np.zeros((n, n))
        for e in range(self.nel):
            jac = 0.5 * self.h[e]
            w = self.wq
            # derivatives wrt t: dN/ds / jac
            A = (self.dNq.T * (w * cK[e] / jac)) @ self.dNq      # int cK N_t N_t dt
            B = (self.Nq.T * (w * cQ[e] * jac)) @ self.Nq
            C = (self.Nq.T * (w * cW[e] * jac)) @ self.Nq
            idx = self.map[e]
            K[np.ix_(idx, idx)] += A + B
            M[np.ix_(idx, idx)] += C
        return K, M

    def evaluate(self, coef: np.ndarray, t: np.ndarray) -> np.ndarray:
        """Evaluate FE functions (columns of coef, [ndof, k]) at points t -> [npts, k]."""
        t = np.asarray(t, dtype=np.float64)
        e = np.clip(np.searchsorted(self.edges, t, side="right") - 1, 0, self.nel - 1)
        s = 2.0 * (t - self.edges[e]) / self.h[e] - 1.0
        s = np.clip(s, -1.0, 1.0)
        N, _ = _shape(s, self.P)                       # [npts, P+1]
        out = np.zeros((t.size, coef.shape[1]))
        for j in range(self.P + 1):
            out += N[:, j:j + 1] * coef[self.map[e, j], :]
        return out


def solve_sl_order(model: SphericalModel, l: int, nmax: int, ra: float, rb: float,
                   nel: int = 48, P: int = 10, nq: Optional[int] = None):
    """First ``nmax`` eigenpairs of the order-``l`` problem on [ra, rb].

    Returns (ev[nmax], mesh, coef[ndof, nmax]); eigenfunctions are normalised to
    int w u^2 dr = 1."""
    nq = nq or (P + 8)
    mesh = _FEMesh(math.log(ra), math.log(rb), nel, P, nq)
    r = np.exp(mesh.tq)
    f = model.pot(r)
    rho = 4.0 * math.pi * model.dens(r)
    L2 = float(l * (l + 1))
    # exputil/SLGridMP2.cc:3647-3654 ; dr = r dt, u_r = u_t / r
    px = r * r * f * f
    qx = (L2 * f - rho * r * r) * f
    wx = -rho * r * r * f
    if not np.all(wx > 0.0):
        raise ValueError("SL weight w = -rho r^2 Phi0 is not positive everywhere: the model density "
                         "vanishes (or Phi0 >= 0) inside [rmin, rmax]")
    K, M = mesh.assemble(px / r, qx * r, wx * r)

    # Boundary terms (exputil/SLGridMP2.cc:1153-1164).  Weak form:
    #   int p u'v' + q u v dr + (pu')(a) v(a) - (pu')(b) v(b) = lambda int w u v dr
    #   inner l=0 : A1 u - A2 (pu') = 0  ->  (pu')(a) =  (A1/A2) u(a)
    #   outer     : B1 u + B2 (pu') = 0  ->  (pu')(b) = -(B1/B2) u(b)
    fa, fb = float(model.pot(ra)), float(model.pot(rb))
    dfa, dfb = float(model.dpot(ra)), float(model.dpot(rb))
    ia, ib = 0, mesh.nel                      # vertex dofs at the two ends
    B1 = (1.0 + l) / rb + dfb / fb
    B2 = 1.0 / (rb * rb * fb * fb)
    K[ib, ib] += B1 / B2
    free = np.ones(mesh.ndof, dtype=bool)
    if l == 0:
        A1 = dfa / fa
        A2 = 1.0 / (ra * ra * fa * fa)
        K[ia, ia] += A1 / A2
    else:
        free[ia] = False                      # u(a) = 0

    Kf = K[np.ix_(free, free)]
    Mf = M[np.ix_(free, free)]
    # Solve the INVERSE pencil  M u = mu K u  (mu = 1/lambda, wanted: the nmax largest).  The weight
    # w = -rho r^2 Phi0 spans 14+ decades for a truncated halo, so the pencil's spectrum runs from
    # O(1) to O(1e20): reducing K u = lambda M u through a Cholesky factor of M leaves absolute
    # errors ~ eps * 1e20 on the O(1) eigenvalues we want (observed: the l = 0 ground state, exactly
    # 1 in theory, came out anywhere in 0.92..1.40 depending on LAPACK driver / BLAS threads).
    # With K (SPD: p, q > 0) on the right the spectrum lies in (0, 1] and the wanted end is the
    # accurately computed one.  Jacobi scaling by diag(K) tames the dynamic range of the shape
    # functions over the decades in r.
    d = 1.0 / np.sqrt(np.diag(Kf))
    Ks = Kf * d[:, None] * d[None, :]
    Ms = Mf * d[:, None] * d[None, :]
    n = Ks.shape[0]
    mu, vec = sla.eigh(Ms, Ks, subset_by_index=[n - nmax, n - 1])
    mu, vec = mu[::-1], vec[:, ::-1]
    ev = 1.0 / mu
    vec = vec * d[:, None] / np.sqrt(mu)[None, :]        # v^T K v = 1  ->  int w u^2 dr = 1
    coef = np.zeros((mesh.ndof, nmax))
    coef[free, :] = vec
    return ev, mesh, coef


def build_slgrid(model: SphericalModel, lmax: int, nmax: int, numr: int = 2000,
                 rmin: Optional[float] = None, rmax: Optional[float] = None,
                 cmap: int = 1, rmap: float = 1.0, nel: int = 48, P: int = 10,
                 Lswitch: int = 32, Lalpha: float = 100.0) -> SLGridSph:
    """Tables with the conventions of ``SLGridSph`` (see module docstring)."""
    rmin = float(model.rmin if rmin is None else rmin)
    rmax = float(model.rmax if rmax is None else rmax)
    xmin, xmax, dxi, xi, r = _xi_grid(cmap, rmin, rmax, rmap, numr)
    p0 = np.asarray(model.pot(r), dtype=np.float64)
    d0 = 4.0 * math.pi * np.asarray(model.dens(r), dtype=np.float64)

    ev = np.zeros((lmax + 1, nmax))
    ef = np.zeros((lmax + 1, nmax, numr))
    _limit = blas_limit()
    _limit.__enter__()
    for l in range(lmax + 1):
        # exputil/SLGridMP2.cc:1119-1135: narrowed radial domain for large l
        Nlo, Nhi = 0, numr
        if l > Lswitch:
            Rfac = 10.0 ** (Lalpha / l)
            Rmin = max(rmin, rmap / Rfac)
            Rmax = min(rmax, rmap * Rfac)
            Nlo = int(np.searchsorted(r, Rmin, side="left"))
            Nhi = int(np.searchsorted(r, Rmax, side="right"))
        ra, rb = r[Nlo], r[Nhi - 1]
        lam, mesh, coef = solve_sl_order(model, l, nmax, ra, rb, nel=nel, P=P)
        u = mesh.evaluate(coef, np.log(r[Nlo:Nhi]))          # [NUM, nmax]
        # sign convention exputil/SLGridMP2.cc:1329-1340
        nfid = min(NEVSIGN, Nhi - Nlo) - 1
        sgn = np.where(u[nfid, :] < 0.0, -1.0, 1.0)
        ev[l] = lam
        ef[l, :, Nlo:Nhi] = (u * sgn[None, :]).T
    _limit.__exit__(None, None, None)

    return SLGridSph(lmax=lmax, nmax=nmax, numr=numr, cmap=cmap, rmin=rmin, rmax=rmax,
                     rmap=rmap, xmin=xmin, xmax=xmax, dxi=dxi, xi=xi, r=r, p0=p0, d0=d0,
                     ev=np.ascontiguousarray(ev), ef=np.ascontiguousarray(ef))


def orthocheck(g: "SLGridSph", num: int = 200):
    """``SLGridSph::orthoCheck`` (exputil/SLGridMP2.cc:1775-1824) on the host tables: one nmax x nmax matrix per l,
    the identity for a biorthonormal basis."""
    from numpy.polynomial import legendre as npleg
    x, w = npleg.leggauss(num)
    knots, weights = 0.5 * (x + 1.0), 0.5 * w
    ximin, ximax = float(g.r_to_xi(g.rmin)), float(g.r_to_xi(g.rmax))
    xs = ximin + (ximax - ximin) * knots
    if g.cmap == 1:
        xs = np.clip(xs, -1.0, 1.0 - 1e-8)
    r = g.xi_to_r(xs)
    idx = np.clip(((xs - g.xmin) / g.dxi).astype(np.int64), 0, g.numr - 2)
    x1 = (g.xi[idx + 1] - xs) / g.dxi
    x2 = (xs - g.xi[idx]) / g.dxi
    if g.cmap == 1:
        dxr = 0.5 * (1.0 - xs) ** 2 / g.rmap
    elif g.cmap == 2:
        dxr = np.exp(-xs)
    else:
        dxr = np.ones_like(xs)
    P0 = x1 * g.p0[idx] + x2 * g.p0[idx + 1]
    D0 = x1 * g.d0[idx] + x2 * g.d0[idx + 1]
    out = []
    for L in range(g.lmax + 1):
        u = x1 * g.ef[L][:, idx] + x2 * g.ef[L][:, idx + 1]        # [nmax, num]
        pot = u / np.sqrt(g.ev[L])[:, None] * P0
        den = u * np.sqrt(g.ev[L])[:, None] * D0
        wgt = r * r / dxr * (ximax - ximin) * weights
        out.append(-(pot * wgt) @ den.T)
    return out


def orthocheck_max(g: "SLGridSph", num: int = 200) -> float:
    """worst deviation from the identity (what exputil/orthoTest.cc:19-87 holds against orthoTol = 1e-2)"""
    return float(max(np.abs(m - np.eye(g.nmax)).max() for m in orthocheck(g, num)))


def read_model_table(path: str):
    """The four columns (r, density, mass, potential) of a ``SphericalModelTable`` file (exputil/massmodel.cc:23-96): lines
    with a '#' or '!' anywhere in them are skipped, the first other line holds the number of rows."""
    with open(path) as f:
        lines = f.read().splitlines()
    k = 0
    while k < len(lines) and ("#" in lines[k] or "!" in lines[k]):
        k += 1
    num = int(lines[k].split()[0])
    rows = np.array([[float(v) for v in ln.split()[:4]] for ln in lines[k + 1:k + 1 + num]], dtype=np.float64)
    return rows[:, 0].copy(), rows[:, 1].copy(), rows[:, 2].copy(), rows[:, 3].copy()


def model_density(rtab: np.ndarray, dtab: np.ndarray, r: float) -> float:
    """``SphericalModelTable::get_density`` with the class defaults ``linear = 1``, ``even = 0``, no divergence
    (exputil/massmodel.cc:19-20, :266-291): the last table value beyond the table, else ``odd2`` -- linear interpolation on the
    interval ``Vlocate`` finds (exputil/Vodd2.cc:49-70, Vlocate.cc:50-66)."""
    if r > rtab[-1]:
        return float(dtab[-1])
    n = len(rtab)
    jl, ju = -1, n
    ascnd = rtab[-1] > rtab[0]
    while ju - jl > 1:
        jm = (ju + jl) >> 1
        if (r > rtab[jm]) == ascnd:
            jl = jm
        else:
            ju = jm
    i = min(max(jl, 0), n - 2)
    return float((dtab[i + 1] * (r - rtab[i]) - dtab[i] * (r - rtab[i + 1])) / (rtab[i + 1] - rtab[i]))


def get_pot(g: "SLGridSph", r: float) -> np.ndarray:
    """``SLGridSph::get_pot(mat, r)`` (exputil/SLGridMP2.cc:872-910): [lmax+1, nmax]"""
    x = float(g.r_to_xi(r))
    i = min(max(int((x - g.xmin) / g.dxi), 0), g.numr - 2)
    x1, x2 = (g.xi[i + 1] - x) / g.dxi, (x - g.xi[i]) / g.dxi
    return (x1 * g.ef[:, :, i] + x2 * g.ef[:, :, i + 1]) / np.sqrt(g.ev) * (x1 * g.p0[i] + x2 * g.p0[i + 1])


def compute_rms_coefs(g: "SLGridSph", model_file: str, scale: float = 1.0, numg: int = 100):
    """``SphericalBasis::compute_rms_coefs`` (src/SphericalBasis.cc:2108-2147): the mean of the l = 0 coefficients and the mean
    square of every (l, n) coefficient of ONE particle drawn from the density of ``noise_model_file``, by a 100-point
    Gauss-Legendre rule in radius over the table's range; ``sqnorm`` is 1 for the Sturm-Liouville basis (Sphere).  Returns
    (meanC[nmax], rmsC[lmax+1, nmax]) -- what ``exp_amd_sph_set_noise`` takes."""
    from numpy.polynomial import legendre as npleg
    rt, dt, _, _ = read_model_table(model_file)
    x, w = npleg.leggauss(int(numg))
    knots, weights = 0.5 * (x + 1.0), 0.5 * w               # LegeQuad: on [0, 1] (tests/test_ref_util.py pins them)
    rmin, rmax = float(rt[0]), float(rt[-1])
    dl = rmax - rmin
    meanC, rmsC = np.zeros(g.nmax), np.zeros((g.lmax + 1, g.nmax))
    for i in range(int(numg)):
        r = rmin + dl * knots[i]
        pot = get_pot(g, r / scale) / 1.0 / scale
        fac = dl * weights[i] * r * r * 4.0 * np.pi * model_density(rt, dt, r)
        meanC += fac * pot[0]
        rmsC += fac * pot * pot
    return meanC, rmsC
