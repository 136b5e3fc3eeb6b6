/*
 * pyexp_oracle.c -- CPU restatement of the pyEXP.basis calls beyond accumulate / getAccel / getFields:
 * getBasis, orthoCheck (cylinder), makeFromFunction, computeQuadrature.
 * TEST INFRASTRUCTURE ONLY (see bfe_oracle.h: parity unpinned; only tests/, smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/).
 *
 * Each function follows the reference statement for statement:
 *   SphericalSL::getBasis            expui/BiorthBasis.cc:960-993
 *   Spherical::makeFromFunction      expui/BiorthBasis.cc:5230-5362
 *   Spherical::computeQuadrature     expui/BiorthBasis.cc:5364-5457
 *   Cylindrical::getBasis            expui/BiorthBasis.cc:1930-1974 -> EmpCylSL::get_all, exputil/EmpCylSL.cc:5635-5800
 *   EmpCylSL::orthoCheck             exputil/EmpCylSL.cc:7199-7260
 *   Cylindrical::makeFromFunction    expui/BiorthBasis.cc:5459-5556 -> getPotSC / getDensSC, exputil/EmpCylSL.cc:7263-7375
 *   Cylindrical::computeQuadrature   expui/BiorthBasis.cc:5558-5630
 * The user's callable is evaluated by the caller: `fv` holds func(x, y, z[, time]) at the quadrature points in
 * the reference's loop order ijk = (i * knots + j) * knots + k; knot[] / weight[] are LegeQuad's (Gauss-Legendre
 * on [0, 1]).
 */
#include "bfe_oracle.h"
#include "cyl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* SphericalSL::getBasis: out[3][lmax+1][nmax][numgrid] = potential, density, rforce */
void orc_pyexp_sph_get_basis(const orc_slgrid *g, double logxmin, double logxmax, int numgrid, double *out)
{
  const int L1 = g->lmax + 1, nmax = g->nmax;
  double *tabpot = (double *)malloc(sizeof(double) * L1 * nmax);
  double *tabden = (double *)malloc(sizeof(double) * L1 * nmax);
  double *tabfrc = (double *)malloc(sizeof(double) * L1 * nmax);
  double dx = (logxmax - logxmin) / (numgrid - 1);
  const size_t plane = (size_t)L1 * nmax * numgrid;
  for (int i = 0; i < numgrid; i++) {
    orc_sl_get_pot(g, pow(10.0, logxmin + dx * i), tabpot);
    orc_sl_get_dens(g, pow(10.0, logxmin + dx * i), tabden);
    orc_sl_get_force(g, pow(10.0, logxmin + dx * i), tabfrc);
    for (int l = 0; l <= g->lmax; l++)
      for (int n = 0; n < nmax; n++) {
        const size_t o = ((size_t)l * nmax + n) * numgrid + i;
        out[o] = tabpot[l * nmax + n];
        out[plane + o] = tabden[l * nmax + n];
        out[2 * plane + o] = tabfrc[l * nmax + n] * (-1.0);
      }
  }
  free(tabpot); free(tabden); free(tabfrc);
}

/* the quadrature point (i, j, k) of Spherical::makeFromFunction / computeQuadrature and its weight without func */
static double sph_quad_point(double rmapping, double ximin, double ximax, int knots, const double *knot,
                             const double *weight, int i, int j, int k, double *x, double *y, double *z,
                             double *rr_, double *costh_, double *phi_)
{
  double xx = ximin + (ximax - ximin) * knot[i];
  double rr = (1.0 + xx) / (1.0 - xx) * rmapping;
  double costh = -1.0 + 2.0 * knot[j];
  double sinth = sqrt(fabs(1.0 - costh * costh));
  double phi = 2.0 * M_PI / knots * k;
  *x = rr * sinth * cos(phi);
  *y = rr * sinth * sin(phi);
  *z = rr * costh;
  *rr_ = rr; *costh_ = costh; *phi_ = phi;
  double dxr = 0.5 * (1.0 - xx) * (1.0 - xx) / rmapping;
  return (ximax - ximin) * rr * rr / dxr * 2.0 * weight[i] * weight[j] * 2.0 * M_PI / knots;
}

/* the points themselves, for the caller to evaluate its function at: xyz[knots^3][3] */
void orc_pyexp_sph_quad_points(double rmin, double rmax, double rmapping, int knots, const double *knot,
                               const double *weight, double *xyz)
{
  double ximin = (rmin / rmapping - 1.0) / (rmin / rmapping + 1.0);
  double ximax = (rmax / rmapping - 1.0) / (rmax / rmapping + 1.0);
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    int k = (int)(ijk - (long)i * knots * knots - (long)j * knots);
    double rr, ct, ph;
    sph_quad_point(rmapping, ximin, ximax, knots, knot, weight, i, j, k, &xyz[3 * ijk], &xyz[3 * ijk + 1],
                   &xyz[3 * ijk + 2], &rr, &ct, &ph);
  }
}

/* Spherical::makeFromFunction: mat[(L+1)(L+2)/2][nmax][2] (re, im) */
void orc_pyexp_sph_make_from_function(const orc_slgrid *g, double rmin, double rmax, double rmapping, int knots,
                                      const double *knot, const double *weight, const double *fv, int potential,
                                      double *mat)
{
  const int Lmax = g->lmax, Nmax = g->nmax;
  double *potd = (double *)malloc(sizeof(double) * (Lmax + 1) * Nmax);
  double *legs = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double ximin = (rmin / rmapping - 1.0) / (rmin / rmapping + 1.0);
  double ximax = (rmax / rmapping - 1.0) / (rmax / rmapping + 1.0);
  memset(mat, 0, sizeof(double) * (size_t)(Lmax + 1) * (Lmax + 2) / 2 * Nmax * 2);
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    int k = (int)(ijk - (long)i * knots * knots - (long)j * knots);
    double x, y, z, rr, costh, phi;
    double w = sph_quad_point(rmapping, ximin, ximax, knots, knot, weight, i, j, k, &x, &y, &z, &rr, &costh, &phi);
    if (potential) orc_sl_get_dens(g, rr, potd);
    else           orc_sl_get_pot(g, rr, potd);
    orc_legendre_R(Lmax, costh, legs);
    double fval = fv[ijk] * w;
    for (int L = 0, l = 0; L <= Lmax; L++) {
      for (int M = 0; M <= L; M++, l++) {
        double prefac = sqrt((2.0 * L + 1.0) / (4.0 * M_PI) * exp(lgamma(L - M + 1) - lgamma(L + M + 1)));
        if (M) prefac *= M_SQRT2;
        for (int n = 0; n < Nmax; n++) {
          double fac = prefac * legs[L * (Lmax + 1) + M] * potd[L * Nmax + n] * fval;
          if (M == 0) mat[((size_t)l * Nmax + n) * 2] += fac;
          else {
            mat[((size_t)l * Nmax + n) * 2] += cos(phi * M) * fac;
            mat[((size_t)l * Nmax + n) * 2 + 1] += sin(phi * M) * fac;
          }
        }
      }
    }
  }
  free(potd); free(legs);
}

double orc_pyexp_sph_compute_quadrature(double rmin, double rmax, double rmapping, int knots, const double *knot,
                                        const double *weight, const double *fv)
{
  double ximin = (rmin / rmapping - 1.0) / (rmin / rmapping + 1.0);
  double ximax = (rmax / rmapping - 1.0) / (rmax / rmapping + 1.0);
  double ret = 0.0;
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    int k = (int)(ijk - (long)i * knots * knots - (long)j * knots);
    double x, y, z, rr, ct, ph;
    ret += fv[ijk] * sph_quad_point(rmapping, ximin, ximax, knots, knot, weight, i, j, k, &x, &y, &z, &rr, &ct, &ph);
  }
  return ret;
}

/* ---- cylinder ------------------------------------------------------------------------------------------ */

#define CTAB(kind, m, n, ix, iy)                                                             \
  g->tab[((((size_t)(kind) * (g->mmax + 1) + (m)) * g->norder + (n)) * (g->numx + 1) + (ix)) * \
             (g->numy + 1) + (iy)]
/* dens[0] = densC, dens[1] = densS, each [mmax+1][norder][numx+1][numy+1] */
#define DTAB(cs, m, n, ix, iy)                                                                \
  dens[((((size_t)(cs) * (g->mmax + 1) + (m)) * g->norder + (n)) * (g->numx + 1) + (ix)) *     \
           (g->numy + 1) + (iy)]

static double cyl_xi_to_r(const orc_cylgrid *g, double xi)
{
  if (g->cmapr > 0) return (1.0 + xi) / (1.0 - xi) * g->ascale;
  return xi;
}
static double cyl_d_xi_to_r(const orc_cylgrid *g, double xi)
{
  if (g->cmapr > 0) return 0.5 * (1.0 - xi) * (1.0 - xi) / g->ascale;
  return 1.0;
}
static double cyl_y_to_z(const orc_cylgrid *g, double y)
{
  if (g->cmapz == 1) return g->hscale * sinh(y);
  else if (g->cmapz == 2) return y * g->hscale / sqrt(1.0 - y * y);
  return y;
}
static double cyl_d_y_to_z(const orc_cylgrid *g, double y)
{
  if (g->cmapz == 1) return g->hscale * cosh(y);
  else if (g->cmapz == 2) return g->hscale * pow(1.0 - y * y, -1.5);
  return 1.0;
}

/* EmpCylSL::get_all (exputil/EmpCylSL.cc:5635-5800): out = {p, d, fr, fz, fp} */
void orc_cyl_get_all(const orc_cylgrid *g, const double *dens, double cylmass, int mm, int nn, double r, double z,
                     double phi, double *out)
{
  double fr = 0.0, fz = 0.0, fp = 0.0, p = 0.0, d = 0.0;
  double rr = sqrt(r * r + z * z);
  if (rr / g->ascale > g->rtable) {
    p = -cylmass / (rr + 1.0e-16);
    fr = p * r / (rr + 1.0e-16) / (rr + 1.0e-16);
    fz = p * z / (rr + 1.0e-16) / (rr + 1.0e-16);
    out[0] = p; out[1] = d; out[2] = fr; out[3] = fz; out[4] = fp;
    return;
  }
  if (z / g->ascale > g->rtable) z = g->rtable * g->ascale;
  if (z / g->ascale < -g->rtable) z = -g->rtable * g->ascale;
  double X = (orc_cyl_r_to_xi(g, r) - g->xmin) / g->dx;
  double Y = (orc_cyl_z_to_y(g, z) - g->ymin) / g->dy;
  int ix = (int)X;
  int iy = (int)Y;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix >= g->numx) ix = g->numx - 1;
  if (iy >= g->numy) iy = g->numy - 1;
  double delx0 = (double)ix + 1.0 - X, dely0 = (double)iy + 1.0 - Y;
  double delx1 = X - (double)ix, dely1 = Y - (double)iy;
  double c00 = delx0 * dely0, c10 = delx1 * dely0, c01 = delx0 * dely1, c11 = delx1 * dely1;
  double ccos = cos(phi * mm), ssin = sin(phi * mm);
#define BL(T) (T(ix, iy) * c00 + T(ix + 1, iy) * c10 + T(ix, iy + 1) * c01 + T(ix + 1, iy + 1) * c11)
#define POTC(a, b) CTAB(0, mm, nn, a, b)
#define RFC(a, b) CTAB(1, mm, nn, a, b)
#define ZFC(a, b) CTAB(2, mm, nn, a, b)
#define POTS(a, b) CTAB(3, mm, nn, a, b)
#define RFS(a, b) CTAB(4, mm, nn, a, b)
#define ZFS(a, b) CTAB(5, mm, nn, a, b)
#define DNC(a, b) DTAB(0, mm, nn, a, b)
#define DNS(a, b) DTAB(1, mm, nn, a, b)
  p += ccos * BL(POTC);
  fr += ccos * BL(RFC);
  fz += ccos * BL(ZFC);
  fp += ssin * mm * BL(POTC);
  d += ccos * BL(DNC);
  if (mm) {
    p += ssin * BL(POTS);
    fr += ssin * BL(RFS);
    fz += ssin * BL(ZFS);
    fp += -ccos * mm * BL(POTS);
    d += ssin * BL(DNS);
  }
  out[0] = p; out[1] = d; out[2] = fr; out[3] = fz; out[4] = fp;
}

/* Cylindrical::getBasis: out[4][mmax+1][norder][numR][numZ] = potential, density, rforce, zforce */
void orc_pyexp_cyl_get_basis(const orc_cylgrid *g, const double *dens, double cylmass, double xmin, double xmax,
                             int numR, double zmin, double zmax, int numZ, int linear, double *out)
{
  double delR = (xmax - xmin) / (numR - 1 > 1 ? numR - 1 : 1);
  double delZ = (zmax - zmin) / (numZ - 1 > 1 ? numZ - 1 : 1);
  const size_t plane = (size_t)(g->mmax + 1) * g->norder * numR * numZ;
  for (int m = 0; m <= g->mmax; m++)
    for (int n = 0; n < g->norder; n++)
      for (int i = 0; i < numR; i++) {
        double R = xmin + delR * i;
        if (!linear) R = pow(10.0, R);
        for (int j = 0; j < numZ; j++) {
          double Z = zmin + delZ * j, v[5];
          orc_cyl_get_all(g, dens, cylmass, m, n, R, Z, 0.0, v);
          const size_t o = (((size_t)m * g->norder + n) * numR + i) * numZ + j;
          out[o] = v[0];
          out[plane + o] = v[1];
          out[2 * plane + o] = v[2];
          out[3 * plane + o] = v[3];
        }
      }
}

/* EmpCylSL::orthoCheck: out[mmax+1][norder][norder] */
void orc_cyl_orthocheck(const orc_cylgrid *g, const double *dens, double *out)
{
  const int NUMX = g->numx, NUMY = g->numy, NORDER = g->norder;
  for (int mm = 0; mm <= g->mmax; mm++) {
    double fac = -4.0 * M_PI * (2.0 * M_PI) * g->dx * g->dy;
    if (mm) fac *= 0.5;
    for (int nn = 0; nn < NORDER * NORDER; nn++) {
      int n1 = nn / NORDER;
      int n2 = nn - n1 * NORDER;
      double sumC = 0.0, sumS = 0.0;
      for (int ix = 0; ix <= NUMX; ix++) {
        double x = g->xmin + g->dx * ix;
        double r = cyl_xi_to_r(g, x);
        double fx = 1.0;
        if (ix == 0 || ix == NUMX) fx = 0.5;
        for (int iy = 0; iy <= NUMY; iy++) {
          double y = g->ymin + g->dy * iy;
          double fy = 1.0;
          if (iy == 0 || iy == NUMX) fy = 0.5;              /* (sic: NUMX) */
          double jac = fac * r / cyl_d_xi_to_r(g, x) * cyl_d_y_to_z(g, y) * fx * fy;
          sumC += jac * CTAB(0, mm, n1, ix, iy) * DTAB(0, mm, n2, ix, iy);
          if (mm) sumS += jac * CTAB(3, mm, n1, ix, iy) * DTAB(1, mm, n2, ix, iy);
        }
      }
      if (mm == 0) out[((size_t)mm * NORDER + n1) * NORDER + n2] = sumC;
      else out[((size_t)mm * NORDER + n1) * NORDER + n2] = sqrt(0.5 * (sumC * sumC + sumS * sumS));
    }
  }
}

/* getPotSC / getDensSC (exputil/EmpCylSL.cc:7263-7375), enforce_limits false */
static void cyl_get_sc(const orc_cylgrid *g, const double *dens, int use_dens, int mm, int n, double R, double z,
                       double *pC, double *pS)
{
  *pC = 0.0; *pS = 0.0;
  if (R / g->ascale > g->rtable || mm > g->mmax || n >= g->norder) return;
  double X = (orc_cyl_r_to_xi(g, R) - g->xmin) / g->dx;
  double Y = (orc_cyl_z_to_y(g, z) - g->ymin) / g->dy;
  int ix = (int)X;
  int iy = (int)Y;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix >= g->numx) ix = g->numx - 1;
  if (iy >= g->numy) iy = g->numy - 1;
  double delx0 = (double)ix + 1.0 - X, dely0 = (double)iy + 1.0 - Y;
  double delx1 = X - (double)ix, dely1 = Y - (double)iy;
  double c00 = delx0 * dely0, c10 = delx1 * dely0, c01 = delx0 * dely1, c11 = delx1 * dely1;
  int nn = n;
  if (use_dens) {
    *pC = BL(DNC);
    if (mm) *pS = BL(DNS);
  } else {
    *pC = BL(POTC);
    if (mm) *pS = BL(POTS);
  }
}

static void cyl_quad_box(const orc_cylgrid *g, double rmin, double *xmin, double *xmax, double *ymin, double *ymax)
{
  *xmin = orc_cyl_r_to_xi(g, rmin * g->ascale);
  *xmax = orc_cyl_r_to_xi(g, g->rtable * g->ascale);
  *ymin = orc_cyl_z_to_y(g, -g->rtable * g->ascale);
  *ymax = orc_cyl_z_to_y(g, g->rtable * g->ascale);
}

/* quadrature points of Cylindrical::makeFromFunction / computeQuadrature: xyz[knots^3][3]; RMIN = rcylmin */
void orc_pyexp_cyl_quad_points(const orc_cylgrid *g, double rmin, int knots, const double *knot, double *xyz)
{
  double xmin, xmax, ymin, ymax;
  cyl_quad_box(g, rmin, &xmin, &xmax, &ymin, &ymax);
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    int k = (int)(ijk - (long)i * knots * knots - (long)j * knots);
    double xx = xmin + (xmax - xmin) * knot[i];
    double yy = ymin + (ymax - ymin) * knot[j];
    double R = cyl_xi_to_r(g, xx);
    double z = cyl_y_to_z(g, yy);
    double phi = 2.0 * M_PI / knots * k;
    xyz[3 * ijk] = R * cos(phi);
    xyz[3 * ijk + 1] = R * sin(phi);
    xyz[3 * ijk + 2] = z;
  }
}

/* Cylindrical::makeFromFunction: mat[mmax+1][norder][2] (re, im) */
void orc_pyexp_cyl_make_from_function(const orc_cylgrid *g, const double *dens, double rmin, int knots,
                                      const double *knot, const double *weight, const double *fv, int potential,
                                      double *mat)
{
  double xmin, xmax, ymin, ymax;
  cyl_quad_box(g, rmin, &xmin, &xmax, &ymin, &ymax);
  const int Mmax = g->mmax, Nmax = g->norder;
  memset(mat, 0, sizeof(double) * (size_t)(Mmax + 1) * Nmax * 2);
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    int k = (int)(ijk - (long)i * knots * knots - (long)j * knots);
    double xx = xmin + (xmax - xmin) * knot[i];
    double yy = ymin + (ymax - ymin) * knot[j];
    double R = cyl_xi_to_r(g, xx);
    double z = cyl_y_to_z(g, yy);
    double phi = 2.0 * M_PI / knots * k;
    double fac = (xmax - xmin) * (ymax - ymin) * weight[i] * weight[j] * 2.0 * M_PI / knots * fv[ijk] * R /
                 cyl_d_xi_to_r(g, xx) * cyl_d_y_to_z(g, yy);
    for (int mm = 0; mm <= Mmax; mm++) {
      double mcos = cos(phi * mm), msin = sin(phi * mm);
      for (int nn = 0; nn < Nmax; nn++) {
        double pC, pS;
        cyl_get_sc(g, dens, potential, mm, nn, R, z, &pC, &pS);
        mat[((size_t)mm * Nmax + nn) * 2] += pC * mcos * fac;
        mat[((size_t)mm * Nmax + nn) * 2 + 1] += pS * msin * fac;
      }
    }
  }
}

double orc_pyexp_cyl_compute_quadrature(const orc_cylgrid *g, double rmin, int knots, const double *knot,
                                        const double *weight, const double *fv)
{
  double xmin, xmax, ymin, ymax;
  cyl_quad_box(g, rmin, &xmin, &xmax, &ymin, &ymax);
  double ret = 0.0;
  for (long ijk = 0; ijk < (long)knots * knots * knots; ijk++) {
    int i = (int)floor((double)ijk / ((double)knots * knots) + 1.0e-16);
    int j = (int)((ijk - (long)i * knots * knots) / knots);
    double xx = xmin + (xmax - xmin) * knot[i];
    double yy = ymin + (ymax - ymin) * knot[j];
    double R = cyl_xi_to_r(g, xx);
    ret += (xmax - xmin) * (ymax - ymin) * weight[i] * weight[j] * 2.0 * M_PI / knots * fv[ijk] * R /
           cyl_d_xi_to_r(g, xx) * cyl_d_y_to_z(g, yy);
  }
  return ret;
}
