"""Exact l > 0 known answers that pin the spherical path's conventions (shared by the CPU test of
the oracle and the GPU test of the device path; neither is compared with the other here).

A density rho(r, theta, phi) = sum_k eps_k f_k(r) A_k(theta, phi), with every A_k ONE solid harmonic
P_l^m(cos theta) {cos | sin}(m phi) in scipy's un-normalised convention, has by the classical
multipole expansion the potential

    Phi_k(r, theta, phi) = -4 pi / (2l+1) * eps_k A_k(theta, phi) *
                           [ r^-(l+1) int_a^r f_k(s) s^(l+2) ds  +  r^l int_r^b f_k(s) s^(1-l) ds ]

whatever the normalisation of A_k: nothing in that statement knows factorial(l,m), the sqrt(2) of the
m > 0 rows, the Condon-Shortley sign or the -4 pi of the coefficient definition
(src/SphericalBasis.cc:328-335, :519-566, :1555-1625), which is why it pins them.  The particle set
is a deterministic product quadrature of that density (Gauss-Legendre in every xi-cell of the radial
table and in cos theta, uniform in phi; masses may have either sign), exact for the angular integrals,
so that the only errors left are the n-truncation of the radial basis and the table's linear
interpolation.  Test infrastructure."""
from __future__ import annotations

import math

import numpy as np
from scipy.integrate import quad
from scipy.special import lpmv


def harmonic(l, m, cs, costh, phi):
    """One un-normalised real solid harmonic (scipy's P_l^m, Condon-Shortley phase included)."""
    a = lpmv(m, l, costh)
    if m == 0:
        return a
    return a * (np.cos(m * phi) if cs == 0 else np.sin(m * phi))


def row_of(l, m, cs):
    """Real-row index of the reference's coefficient order (src/SphericalBasis.cc:513-590)."""
    return l * l + (2 * m - 1 + cs if m else 0)


class MultipoleCase:
    """rho = rho0(r) [1(l = 0 part, optional) + sum_k eps_k shape_l(r) A_k], a <= r <= b."""

    def __init__(self, model, g, modes, monopole=False, ncell=2):
        self.model, self.g = model, g
        self.a, self.b = float(g.rmin), float(g.rmax)
        self.modes = list(modes)
        self.monopole = monopole
        # eps: each harmonic perturbs by at most 10 % / number of modes of the unit amplitude
        ct = np.linspace(-1, 1, 2001)
        self.eps = []
        for (l, m, cs) in self.modes:
            amax = np.abs(lpmv(m, l, ct)).max()
            self.eps.append(0.5 / (len(self.modes) * amax))
        self._build_particles(ncell)

    # radial profile of the l-th perturbation: r^l at the centre, r^-(l+5) far out (so that the multipole
    # moments converge well inside rmax) on a scale length that is NOT the background model's: the
    # expansion needs several radial functions and converges geometrically in n
    CORE = 0.7

    def shape(self, l, r):
        r = np.asarray(r, dtype=np.float64)
        return r ** l * (self.CORE ** 2 + r * r) ** -(l + 2.5)

    def _xi_to_r(self, xi):
        g = self.g
        if g.cmap == 1:
            return g.rmap * (1.0 + xi) / (1.0 - xi)
        if g.cmap == 2:
            return np.exp(xi)
        return xi

    def _build_particles(self, ncell):
        g = self.g
        lmax = g.lmax
        # radial nodes: Gauss-Legendre in r inside every xi-cell of the table
        xg, wg = np.polynomial.legendre.leggauss(ncell)
        edges = self._xi_to_r(np.asarray(g.xi, dtype=np.float64))
        edges[0], edges[-1] = max(edges[0], self.a), min(edges[-1], self.b)
        lo, hi = edges[:-1], edges[1:]
        rr = (0.5 * (lo + hi)[:, None] + 0.5 * (hi - lo)[:, None] * xg[None, :]).reshape(-1)
        wr = (0.5 * (hi - lo)[:, None] * wg[None, :]).reshape(-1)
        # angular nodes: exact for products of two harmonics of degree <= lmax
        xt, wt = np.polynomial.legendre.leggauss(lmax + 2)
        nph = 2 * lmax + 3
        ph = (np.arange(nph) + 0.37) * (2 * math.pi / nph)
        wp = np.full(nph, 2 * math.pi / nph)
        R, T, P = np.meshgrid(rr, xt, ph, indexing="ij")
        W = wr[:, None, None] * wt[None, :, None] * wp[None, None, :] * R * R
        rho = np.zeros_like(R)
        if self.monopole:
            rho += self.model.dens(R)
        for (l, m, cs), e in zip(self.modes, self.eps):
            rho += e * self.shape(l, R) * harmonic(l, m, cs, T, P)
        st = np.sqrt(1.0 - T * T)
        self.pos = np.stack([(R * st * np.cos(P)).ravel(), (R * st * np.sin(P)).ravel(), (R * T).ravel()], 1)
        self.mass = (rho * W).ravel()

    # ---- the exact answer -------------------------------------------------------------------------
    def _radial(self, l, r):
        """(F, dF/dr) with F(r) = r^-(l+1) int_a^min(r,b) f s^(l+2) ds + r^l int_min(r,b)^b f s^(1-l) ds."""
        rc = min(r, self.b)
        brk = [x for x in (0.01, 0.1, 0.5, 1.0, 2.0, 5.0, 15.0) if self.a < x < self.b]
        fin = lambda s: self.shape(l, s) * s ** (l + 2)
        fout = lambda s: self.shape(l, s) * s ** (1 - l)
        Iin = quad(fin, self.a, rc, points=[x for x in brk if x < rc] or None, epsabs=0, epsrel=1e-12, limit=400)[0]
        Iout = 0.0
        if rc < self.b:
            Iout = quad(fout, rc, self.b, points=[x for x in brk if x > rc] or None, epsabs=0, epsrel=1e-12,
                        limit=400)[0]
        F = r ** -(l + 1) * Iin + r ** l * Iout
        dF = -(l + 1) * r ** -(l + 2) * Iin + l * r ** (l - 1) * Iout       # the f(r) terms cancel
        return F, dF

    def exact_mode(self, k, pts, clamp_exterior=True):
        """Potential and acceleration of mode k alone at Cartesian points [n, 3].  The acceleration is
        -grad Phi assembled as the reference assembles it (src/SphericalBasis.cc:1645-1651):
            a = -(Phi_r x/r - Phi_c x z/r^3, Phi_r y/r - Phi_c y z/r^3, Phi_r z/r + Phi_c R^2/r^3)
                + Phi_phi (y, -x, 0)/R^2,           Phi_c = dPhi/dcos(theta)
        which for r > rmax -- with `clamp_exterior` -- uses the reference's CLAMPED radius r = rmax in
        those quotients (:1555-1560: `r = rmax` before the sums, never restored) while Phi_r, Phi_c,
        Phi_phi are the exact exterior multipole's at the true radius."""
        l, m, cs = self.modes[k]
        e = self.eps[k]
        pts = np.asarray(pts, dtype=np.float64)
        pot, acc = np.zeros(len(pts)), np.zeros((len(pts), 3))
        h = 1e-6
        c = -4.0 * math.pi / (2 * l + 1) * e
        for i, p in enumerate(pts):
            x, y, z = p
            r = float(np.linalg.norm(p))
            ct, ph = z / r, math.atan2(y, x)
            F, dF = self._radial(l, r)
            A = float(harmonic(l, m, cs, ct, ph))
            pot[i] = c * F * A
            # angular derivatives of A by central differences of the closed form (smooth, O(h^2))
            dA_dct = float(harmonic(l, m, cs, ct + h, ph) - harmonic(l, m, cs, ct - h, ph)) / (2 * h)
            dA_dph = float(harmonic(l, m, cs, ct, ph + h) - harmonic(l, m, cs, ct, ph - h)) / (2 * h)
            Pr, Pc, Pp = c * dF * A, c * F * dA_dct, c * F * dA_dph
            ru = min(r, self.b) if clamp_exterior else r
            R2 = x * x + y * y
            acc[i] = [-(Pr * x / ru - Pc * x * z / ru ** 3) + Pp * y / R2,
                      -(Pr * y / ru - Pc * y * z / ru ** 3) - Pp * x / R2,
                      -(Pr * z / ru + Pc * R2 / ru ** 3)]
        return pot, acc

    def test_points(self, seed=4):
        """Directions away from the poles and the axis planes, radii inside, near the edge and
        BEYOND rmax (the multipole continuation of src/SphericalBasis.cc:1555-1560, :1605-1628)."""
        rng = np.random.default_rng(seed)
        radii = [0.2, 0.6, 1.0, 1.7, 3.0, 8.0, 0.98 * self.b, 1.5 * self.b, 4.0 * self.b]
        pts = []
        for r in radii:
            ct = rng.uniform(-0.8, 0.8)
            ph = rng.uniform(0.2, 6.0)
            st = math.sqrt(1 - ct * ct)
            pts.append([r * st * math.cos(ph), r * st * math.sin(ph), r * ct])
        return np.array(pts)
