"""Empirical cylindrical basis tables (EmpCylSL EOF tables), init-time, host only.

Produces what ``EmpCylSL`` holds after ``generate_eof`` (``exputil/EmpCylSL.cc:2375-2683``,
``make_eof`` ``:2866``, ``compute_eof_grid`` ``:1454-1630``): for every azimuthal order m and
radial order n the bilinear-interpolation tables potC, rforceC, zforceC (and the sine
counterparts for m >= 1) on the (NUMX+1) x (NUMY+1) grid of ``setup_table`` (``:2123-2137``).

Method (the reference's, re-implemented with numpy):
1. a spherical Sturm-Liouville helper basis (lmaxfid, nmaxfid) on the exponential-sphere model of
   ``EmpCylSL::make_sl`` / ``densR`` (``:584-700``), radii in units of ASCALE;
2. covariance of the helper-basis potentials under the conditioning disk density
   (``src/Cylinder.cc:315-322``) by Gauss-Legendre quadrature in (xi_r, cos theta) -- azimuthal
   quadrature with nump = 1, so cosine and sine blocks are identical (``:2486-2490``);
3. symmetric eigen-decomposition per m, largest-variance vectors first, sign fixed as in
   ``eigen_problem`` (``:3577-3587``);
4. tabulation of potential and cylindrical force components on the grid.

Divergences from the reference, all init-only and irrelevant to hot-path parity (which is defined
given the tables): eigenvectors are ordered by decreasing eigenvalue explicitly; the sine z-force
table holds the z-force (the reference's single-process branch copies the R-force there,
SURVEY.md section 3.7 item 5).  The split into vertically symmetric and antisymmetric functions (``ncylodd``) is the
``nodd`` argument of ``build_empcyl``.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, Optional

import numpy as np
from numpy.polynomial import legendre as npleg
from scipy.special import lpmv

from .models import NumericModel
from .slgrid import SLGridSph, build_slgrid, blas_limit

KIND = {"potC": 0, "rforceC": 1, "zforceC": 2, "potS": 3, "rforceS": 4, "zforceS": 5}


@dataclasses.dataclass
class EmpCylGrid:
    mmax: int
    norder: int
    numx: int
    numy: int
    cmapr: int
    cmapz: int
    ascale: float
    hscale: float
    rmin: float          # RMIN (units of ascale)
    rmax: float          # RMAX (units of ascale)
    rtable: float
    xmin: float
    xmax: float
    dx: float
    ymin: float
    ymax: float
    dy: float
    tab: np.ndarray      # [6, mmax+1, norder, numx+1, numy+1]
    # density tables densC / densS (exputil/EmpCylSL.cc:1507-1534), [2, mmax+1, norder, numx+1,
    # numy+1]; not on the n-body path, used by the field evaluation (pyEXP getFields)
    dens: Optional[np.ndarray] = None

    def save(self, path: str) -> None:
        np.savez_compressed(path, **{f.name: getattr(self, f.name)
                                     for f in dataclasses.fields(self)
                                     if getattr(self, f.name) is not None})

    @staticmethod
    def load(path: str) -> "EmpCylGrid":
        z = np.load(path)
        kw = {}
        for f in dataclasses.fields(EmpCylGrid):
            if f.name not in z:              # caches written before the density tables existed
                continue
            v = z[f.name]
            kw[f.name] = v.item() if v.ndim == 0 else np.ascontiguousarray(v, dtype=np.float64)
        for k in ("mmax", "norder", "numx", "numy", "cmapr", "cmapz"):
            kw[k] = int(kw[k])
        return EmpCylGrid(**kw)


# ---- coordinate maps (exputil/EmpCylSL.cc:6446-6491, :7109-7146) -------------------------------------

def r_to_xi(r, ascale, cmapr):
    r = np.asarray(r, dtype=np.float64)
    return (r / ascale - 1.0) / (r / ascale + 1.0) if cmapr > 0 else r


def xi_to_r(xi, ascale, cmapr):
    xi = np.asarray(xi, dtype=np.float64)
    return (1.0 + xi) / (1.0 - xi) * ascale if cmapr > 0 else xi


def d_xi_to_r(xi, ascale, cmapr):
    xi = np.asarray(xi, dtype=np.float64)
    return 0.5 * (1.0 - xi) ** 2 / ascale if cmapr > 0 else np.ones_like(xi)


def z_to_y(z, hscale, cmapz):
    z = np.asarray(z, dtype=np.float64)
    if cmapz == 1:
        return np.sign(z) * np.arcsinh(np.abs(z / hscale))
    if cmapz == 2:
        return z / np.sqrt(z * z + hscale * hscale)
    return z


def y_to_z(y, hscale, cmapz):
    y = np.asarray(y, dtype=np.float64)
    if cmapz == 1:
        return hscale * np.sinh(y)
    if cmapz == 2:
        return y * hscale / np.sqrt(1.0 - y * y)
    return y


# ---- vectorised SL table evaluation (exputil/SLGridMP2.cc:872-989) -----------------------------------

def sl_dens(g: SLGridSph, r: np.ndarray) -> np.ndarray:
    """dend[npts, L+1, nmax] of SLGridSph::get_dens (exputil/SLGridMP2.cc:913-950)."""
    r = np.asarray(r, dtype=np.float64)
    x = g.r_to_xi(r)
    idx = np.clip(((x - g.xmin) / g.dxi).astype(np.int64), 0, g.numr - 2)
    x1 = (g.xi[idx + 1] - x) / g.dxi
    x2 = (x - g.xi[idx]) / g.dxi
    e0 = np.moveaxis(g.ef[:, :, idx], 2, 0)
    e1 = np.moveaxis(g.ef[:, :, idx + 1], 2, 0)
    D0 = x1 * g.d0[idx] + x2 * g.d0[idx + 1]
    return (x1[:, None, None] * e0 + x2[:, None, None] * e1) * np.sqrt(g.ev)[None] * D0[:, None, None]


def sl_eval(g: SLGridSph, r: np.ndarray, want_force: bool = True):
    """potd[npts, L+1, nmax] (and dpot) of SLGridSph::get_pot / get_force at radii r."""
    r = np.asarray(r, dtype=np.float64)
    x = g.r_to_xi(r)
    idx = np.clip(((x - g.xmin) / g.dxi).astype(np.int64), 0, g.numr - 2)
    x1 = (g.xi[idx + 1] - x) / g.dxi
    x2 = (x - g.xi[idx]) / g.dxi
    sq = np.sqrt(g.ev)                                      # [L+1, nmax]
    ef = g.ef                                               # [L+1, nmax, numr]
    e0 = np.moveaxis(ef[:, :, idx], 2, 0)                   # [npts, L+1, nmax]
    e1 = np.moveaxis(ef[:, :, idx + 1], 2, 0)
    P0 = x1 * g.p0[idx] + x2 * g.p0[idx + 1]
    potd = (x1[:, None, None] * e0 + x2[:, None, None] * e1) / sq[None] * P0[:, None, None]
    if not want_force:
        return potd, None
    j = np.clip(((x - g.xmin) / g.dxi).astype(np.int64), 1, g.numr - 2)
    p = (x - g.xi[j]) / g.dxi
    if g.cmap == 1:
        dxr = 0.5 * (1.0 - x) ** 2 / g.rmap
    elif g.cmap == 2:
        dxr = np.exp(-x)
    else:
        dxr = np.ones_like(x)
    fac = dxr / g.dxi
    em = np.moveaxis(ef[:, :, j - 1], 2, 0) * g.p0[j - 1][:, None, None]
    ec = np.moveaxis(ef[:, :, j], 2, 0) * g.p0[j][:, None, None]
    ep = np.moveaxis(ef[:, :, j + 1], 2, 0) * g.p0[j + 1][:, None, None]
    dpot = fac[:, None, None] * ((p - 0.5)[:, None, None] * em - 2.0 * p[:, None, None] * ec +
                                 (p + 0.5)[:, None, None] * ep) / sq[None]
    return potd, dpot


def _legendre_all(lmax: int, m: int, x: np.ndarray):
    """Normalised associated Legendre functions of EmpCylSL::legendre_R / dlegendre_R
    (exputil/EmpCylSL.cc:6493-6612): sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_l^m(x) with the
    Condon-Shortley phase, and their x-derivatives, for l = m..lmax."""
    from scipy.special import gammaln
    ls = np.arange(m, lmax + 1)
    lognorm = 0.5 * (np.log((2.0 * ls + 1.0) / (4.0 * math.pi)) + gammaln(ls - m + 1.0) -
                     gammaln(ls + m + 1.0))
    # lpmv overflows for large l, m; build the normalised functions by the stable recurrence
    P = np.empty((x.size, ls.size))
    u = np.sqrt(np.maximum(0.0, 1.0 - x * x))
    pmm = np.full_like(x, math.sqrt(1.0 / (4.0 * math.pi)))
    for k in range(1, m + 1):
        pmm = -pmm * u * math.sqrt((2.0 * k + 1.0) / (2.0 * k))
    P[:, 0] = pmm
    if ls.size > 1:
        P[:, 1] = math.sqrt(2.0 * m + 3.0) * x * pmm
    for j in range(2, ls.size):
        l = m + j
        a = math.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
        b = math.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
        P[:, j] = a * (x * P[:, j - 1] - b * P[:, j - 2])
    del lognorm
    xc = np.clip(x, -(1.0 - 3 * np.finfo(float).eps), 1.0 - 3 * np.finfo(float).eps)
    somx2 = 1.0 / (xc * xc - 1.0)
    dP = np.empty_like(P)
    for k, l in enumerate(ls):
        if l == m:
            dP[:, k] = somx2 * xc * l * P[:, k]
        else:   # exputil/EmpCylSL.cc:6604
            dP[:, k] = somx2 * (xc * l * P[:, k] -
                                math.sqrt((l * l - m * m) * (2.0 * l + 1.0) / (2.0 * l - 1.0)) * P[:, k - 1])
    return P, dP


def default_disk_density(acyl: float, hcyl: float, sech2: bool = False) -> Callable:
    """src/Cylinder.cc:315-322"""
    h = 0.5 * hcyl if sech2 else hcyl

    def dens(R, z):
        f = np.exp(-np.abs(z) / h)
        s = 2.0 * f / (1.0 + f * f)
        return np.exp(-R / acyl) * s * s / (4.0 * math.pi * acyl * acyl * h)

    return dens


def eof_covariance_from_particles(sl: SLGridSph, m: int, ascale: float, rtable: float, rmax2: float, mass, pos,
                                  chunk: int = 16384):
    """The covariance of the helper functions over a set of particles -- ``EmpCylSL::accumulate_eof``
    (exputil/EmpCylSL.cc:2686-2862) under its caller's cut (``Cylinder::determine_coefficients_thread``,
    src/Cylinder.cc:806-820: x^2 + y^2 + z^2 < Rmax2) -- for ONE harmonic m: (SC, SS, used, cylmass) with SC, SS
    [rank, rank], rank = NMAX (LMAX - m + 1), index nn = ir + NMAX (l - m); SS is None for m = 0.  ``pos`` [n, 3] in
    the frame of the basis (centred, rotated).  The per-particle rank-one updates of the reference are one
    matrix product a chunk here: V^T diag(mass cos^2 | sin^2) V -- init-time host work in the reference too (it runs
    on the CPU there also when the step runs on a GPU)."""
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    lmax, nmax = sl.lmax, sl.nmax
    nl = lmax - m + 1
    rank = nl * nmax
    pfac = 1.0 / math.sqrt(ascale)
    SC = np.zeros((rank, rank))
    SS = np.zeros((rank, rank)) if m > 0 else None
    r2 = pos[:, 0] ** 2 + pos[:, 1] ** 2
    R2 = r2 + pos[:, 2] ** 2
    inside = R2 < rmax2
    used, cylmass = int(inside.sum()), float(mass[inside].sum())
    rr = np.sqrt(np.sqrt(r2) ** 2 + pos[:, 2] ** 2)
    sel = np.flatnonzero(inside & ~(rr / ascale > rtable))
    for b in range(0, sel.size, chunk):
        k = sel[b:b + chunk]
        potd, _ = sl_eval(sl, rr[k] / ascale, want_force=False)          # [np, L+1, nmax]
        costh = pos[k, 2] / (rr[k] + 1.0e-18)
        P, _ = _legendre_all(lmax, m, costh)                              # [np, nl]
        V = (pfac * P[:, :, None] * potd[:, m:, :]).reshape(k.size, rank)  # nn = ir + NMAX * (l - m)
        if m == 0:
            SC += (V * mass[k][:, None]).T @ V
        else:
            phi = np.arctan2(pos[k, 1], pos[k, 0])
            c, s_ = np.cos(m * phi), np.sin(m * phi)
            SC += (V * (mass[k] * c * c)[:, None]).T @ V
            SS += (V * (mass[k] * s_ * s_)[:, None]).T @ V
    return SC, SS, used, cylmass


def build_empcyl(mmax: int = 6, norder: int = 12, numx: int = 128, numy: int = 64,
                 acyl: float = 0.01, hcyl: float = 0.002, rcylmin: float = 0.001,
                 rcylmax: float = 20.0, lmaxfid: int = 32, nmaxfid: int = 24, numr: int = 2000,
                 cmapr: int = 1, cmapz: int = 1, rnum: int = 200, tnum: int = 80,
                 dens: Optional[Callable] = None, nodd: Optional[int] = None, pnum: int = 1,
                 ashift: float = 0.0, particles=None) -> EmpCylGrid:
    """``nodd`` (the reference's ``ncylodd``; EmpCylSL's constructor argument, exputil/EmpCylSL.cc:178-185): with
    0 <= nodd <= norder the functions are chosen by vertical parity -- the norder - nodd largest-variance combinations
    of the helper functions with l + m even (symmetric about the plane) first, then the nodd largest-variance ones with
    l + m odd (``lE`` / ``lO``, ``SCe`` / ``SCo``, :2250-2268, :2743-2858; tabulated even first, :1680-1760).  None or out
    of range: the norder largest-variance combinations whatever their parity.

    ``pnum`` azimuthal knots (at least one, src/Cylinder.cc:168, expui/BiorthBasis.cc:1543) and ``ashift``: with more than
    one knot the covariance of the cosine and of the sine functions of an m >= 1 are integrated separately under a
    conditioning density folded into [-pi/m, pi/m] and shifted by ashift * acyl along x (``dcond``, src/Cylinder.cc:
    325-348, expui/BiorthBasis.cc:1345-1366; ``generate_eof``, exputil/EmpCylSL.cc:2455-2500) -- the sine tables then
    differ from the cosine tables.  One knot: the azimuthal average, sine tables = cosine tables.

    ``particles`` = (mass [n], pos [n, 3] in the basis' frame): the basis is conditioned on the PARTICLES instead of a
    target density -- the reference's ``precond: false`` (``Cylinder::determine_coefficients_eof``, src/Cylinder.cc:
    1202-1249: ``accumulate_eof`` of every particle inside rcylmax * acyl, then ``make_eof``, exputil/EmpCylSL.cc:
    2866-3300): the covariance of the cosine and of the sine functions of each m are the particle sums
    (``eof_covariance_from_particles``), each decomposed on its own (``eigen_problem`` with request_id 1 / 0), and
    tabulated by the same ``compute_eof_grid``.  ``rnum`` / ``tnum`` / ``pnum`` / ``ashift`` / ``dens`` play no part then.
    The grid records ``cylmass`` and ``used`` of the conditioning pass."""
    ASCALE, HSCALE, RMIN, RMAX = acyl, hcyl, rcylmin, rcylmax
    even_odd = nodd is not None and 0 <= nodd <= norder
    nump = max(1, int(pnum))
    pfac = 1.0 / math.sqrt(ASCALE)           # exputil/EmpCylSL.cc:173-175
    ffac = pfac / ASCALE
    dens = dens or default_disk_density(acyl, hcyl)

    # 1. helper spherical basis on the exponential-sphere model (densR Exponential, :590-592)
    model = NumericModel(lambda R: np.exp(-R) / (4.0 * math.pi * R), RMIN, RMAX, total_mass=None)
    sl = build_slgrid(model, lmaxfid, nmaxfid, numr=numr, rmin=RMIN, rmax=RMAX * 0.99, cmap=1,
                      rmap=1.0, nel=48, P=10)

    # grid geometry (setup_table :2123-2137)
    rtable = math.sqrt(0.5) * RMAX
    XMIN = float(r_to_xi(RMIN * ASCALE, ASCALE, cmapr))
    XMAX = float(r_to_xi(rtable * ASCALE, ASCALE, cmapr))
    dX = (XMAX - XMIN) / numx
    YMIN = float(z_to_y(-rtable * ASCALE, HSCALE, cmapz))
    YMAX = float(z_to_y(rtable * ASCALE, HSCALE, cmapz))
    dY = (YMAX - YMIN) / numy

    # 2. quadrature nodes (generate_eof :2427-2454)
    kr, wr = npleg.leggauss(rnum)
    kr, wr = 0.5 * (kr + 1.0), 0.5 * wr
    kt, wt = npleg.leggauss(tnum)
    kt, wt = 0.5 * (kt + 1.0), 0.5 * wt
    xi_q = XMIN + (XMAX - XMIN) * kr
    rr_q = xi_to_r(xi_q, ASCALE, cmapr)                        # [rnum]
    potd_q, _ = sl_eval(sl, rr_q / ASCALE, want_force=False)   # [rnum, L+1, nmax]
    costh = -1.0 + 2.0 * kt                                    # [tnum]
    dphi = 2.0 * math.pi / nump
    RR, CT = np.meshgrid(rr_q, costh, indexing="ij")           # [rnum, tnum]
    Rq = RR * np.sqrt(1.0 - CT * CT)
    zq = RR * CT
    jfac = (dphi * 2.0 * wt[None, :] * (XMAX - XMIN) * wr[:, None] * RR * RR /
            d_xi_to_r(xi_q, ASCALE, cmapr)[:, None])
    wq = (dens(Rq, zq) * jfac).reshape(-1)                     # [rnum*tnum]: the unshifted density (m = 0, or one knot)

    def dcond(phi, M):
        """the conditioning density of harmonic M >= 1 at azimuth phi, times jfac"""
        dmult = math.pi / M
        phiS = phi + dmult * int((2.0 * math.pi - phi) / dmult) if phi > math.pi else phi - dmult * int(phi / dmult)
        x = Rq * math.cos(phiS) - ashift * acyl
        y = Rq * math.sin(phiS)
        return (dens(np.sqrt(x * x + y * y), zq) * jfac).reshape(-1)

    # grid nodes for the tabulation (compute_eof_grid :1475-1492)
    xg = XMIN + dX * np.arange(numx + 1)
    yg = YMIN + dY * np.arange(numy + 1)
    rg = xi_to_r(xg, ASCALE, cmapr)
    zg = y_to_z(yg, HSCALE, cmapz)
    Rg, Zg = np.meshgrid(rg, zg, indexing="ij")
    Rg, Zg = Rg.reshape(-1), Zg.reshape(-1)
    rrg = np.sqrt(Rg * Rg + Zg * Zg) + 1.0e-18
    potd_g, dpot_g = sl_eval(sl, rrg / ASCALE, want_force=True)  # [npts, L+1, nmax]
    dend_g = sl_dens(sl, rrg / ASCALE)
    dfac = ffac / ASCALE                                       # exputil/EmpCylSL.cc:176
    cg = Zg / rrg

    tab = np.zeros((6, mmax + 1, norder, numx + 1, numy + 1))
    dtab = np.zeros((2, mmax + 1, norder, numx + 1, numy + 1))
    _limit = blas_limit()
    _limit.__enter__()
    for m in range(mmax + 1):
        nl = lmaxfid - m + 1
        # 3. covariance: v[(l-m), ir] = pfac * P_l^m(cos) * potd(l, ir) [* 1/2 for m>0, nump = 1]
        Pq, _ = _legendre_all(lmaxfid, m, costh)                # [tnum, nl]
        V = (pfac * Pq[None, :, :, None] * potd_q[:, None, m:, :])  # [rnum, tnum, nl, nmax]
        V = V.reshape(rnum * tnum, nl * nmaxfid)                # nn = ir + NMAX*(l-m)
        SS = None
        if particles is not None:
            SC, SS, eof_used, eof_mass = eof_covariance_from_particles(sl, m, ASCALE, rtable, (RMAX * ASCALE) ** 2,
                                                                       particles[0], particles[1])
        elif m == 0:
            SC = (V * (wq * nump)[:, None]).T @ V               # (every knot adds the same)
        elif nump == 1:
            SC = 0.25 * ((V * wq[:, None]).T @ V)               # facC = facS = V / 2
        else:
            SC = np.zeros((V.shape[1], V.shape[1]))
            SS = np.zeros_like(SC)
            for qp in range(nump):
                phi = dphi * qp
                w = dcond(phi, m)
                VW = (V * w[:, None]).T @ V
                SC += math.cos(m * phi) ** 2 * VW
                SS += math.sin(m * phi) ** 2 * VW
        def top(mat, k):
            mx_ = np.abs(mat).max()
            if mx_ > 1e-5:
                mat = mat / mx_
            w, v = np.linalg.eigh(mat)
            v = v[:, np.argsort(w)[::-1][:k]]
            nf = min(4, v.shape[0]) - 1
            return v * np.where(v[nf, :] < 0.0, -1.0, 1.0)[None, :]

        def functions(mat):
            if not even_odd:
                return top(mat, norder)
            # columns nn = ir + NMAX * (l - m): the two parities do not mix under a density symmetric about the plane, and
            # each is decomposed in its own index space nn' = ir + NMAX * il (il counts the l of that parity)
            lpar = np.repeat((np.arange(m, lmaxfid + 1) + m) % 2, nmaxfid)          # 0: l + m even
            ce, co = np.flatnonzero(lpar == 0), np.flatnonzero(lpar == 1)
            out = np.zeros((nl * nmaxfid, norder))
            ne = norder - nodd
            ve = top(mat[np.ix_(ce, ce)], ne)
            out[ce, :ve.shape[1]] = ve
            if nodd and co.size:
                vo = top(mat[np.ix_(co, co)], nodd)
                out[co, ne:ne + vo.shape[1]] = vo
            return out

        ef = functions(SC)
        efS = functions(SS) if SS is not None else ef             # one knot: the sine functions ARE the cosine functions

        # 4. tabulation
        Pg, dPg = _legendre_all(lmaxfid, m, cg)                 # [npts, nl]
        fac = 1.0 if m == 0 else math.sqrt(2.0)
        potl = (fac * pfac) * Pg[:, :, None] * potd_g[:, m:, :]       # [npts, nl, nmax]
        potr = (fac * ffac) * Pg[:, :, None] * dpot_g[:, m:, :]
        pott = (fac * pfac) * dPg[:, :, None] * potd_g[:, m:, :]
        frR = -(potr * (Rg / rrg)[:, None, None] - pott * (Zg * Rg / rrg ** 3)[:, None, None])
        frZ = -(potr * (Zg / rrg)[:, None, None] + pott * (Rg * Rg / rrg ** 3)[:, None, None])
        npts = Rg.size
        dnb = ((fac * dfac * 0.25 / math.pi) * Pg[:, :, None] * dend_g[:, m:, :]).reshape(npts, -1)
        for off, vec in ((0, ef), (3, efS)):
            if off and m == 0:
                continue
            for k, arr in ((0, potl), (1, frR), (2, frZ)):
                tab[k + off, m] = (arr.reshape(npts, -1) @ vec).T.reshape(norder, numx + 1, numy + 1)
            # density (compute_eof_grid :1507, :1518, :1534): fac * P_lm * dend * dfac / (4 pi)
            dtab[off // 3, m] = (dnb @ vec).T.reshape(norder, numx + 1, numy + 1)
    _limit.__exit__(None, None, None)
    out = EmpCylGrid(mmax=mmax, norder=norder, numx=numx, numy=numy, cmapr=cmapr, cmapz=cmapz,
                     ascale=ASCALE, hscale=HSCALE, rmin=RMIN, rmax=RMAX, rtable=rtable,
                     xmin=XMIN, xmax=XMAX, dx=dX, ymin=YMIN, ymax=YMAX, dy=dY,
                     tab=np.ascontiguousarray(tab), dens=np.ascontiguousarray(dtab))
    if particles is not None:
        # "Cylinder: eof grid mass=..., number=..." (src/Cylinder.cc:1236-1237)
        out.eof_used, out.eof_cylmass = eof_used, eof_mass
    return out
