/*
 * bfe_oracle.c -- CPU restatement of EXP's BFE hot path.  TEST INFRASTRUCTURE ONLY.
 * See bfe_oracle.h for the scope statement ("parity unpinned", who may call this).
 *
 * Every function cites the reference file:line whose arithmetic it follows.  The
 * order of floating-point operations is kept as written in the reference so that
 * this file answers "what does EXP's CPU path compute" to the last bit it can.
 */
#include "bfe_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DSMALL  1.0e-16        /* src/expand.H:130              */
#define XOFFSET 1.0e-8         /* exputil/SLGridMP2.cc:10       */
static const double MINEPS = 3.0 * DBL_EPSILON;   /* src/Basis.cc:7 */

#define P_(l, m) p[(l) * (lmax + 1) + (m)]
#define DP_(l, m) dp[(l) * (lmax + 1) + (m)]

/* src/Basis.cc:14-52 */
void orc_legendre_R(int lmax, double x, double *p)
{
  double fact, somx2, pll, pl1, pl2;

  P_(0, 0) = pll = 1.0;
  if (lmax > 0) {
    somx2 = sqrt((1.0 - x) * (1.0 + x));
    fact = 1.0;
    for (int m = 1; m <= lmax; m++) {
      pll *= -fact * somx2;
      P_(m, m) = pll;
      fact += 2.0;
    }
  }

  for (int m = 0; m < lmax; m++) {
    pl2 = P_(m, m);
    P_(m + 1, m) = pl1 = x * (2 * m + 1) * pl2;
    for (int l = m + 2; l <= lmax; l++) {
      P_(l, m) = pll = (x * (2 * l - 1) * pl1 - (l + m - 1) * pl2) / (l - m);
      pl2 = pl1;
      pl1 = pll;
    }
  }
}

/* src/Basis.cc:54-93 */
void orc_dlegendre_R(int lmax, double x, double *p, double *dp)
{
  double somx2;

  orc_legendre_R(lmax, x, p);

  if (1.0 - fabs(x) < MINEPS) {
    if (x > 0) x = 1.0 - MINEPS;
    else       x = -(1.0 - MINEPS);
  }

  somx2 = 1.0 / (x * x - 1.0);
  DP_(0, 0) = 0.0;
  for (int l = 1; l <= lmax; l++) {
    for (int m = 0; m < l; m++)
      DP_(l, m) = somx2 * (x * l * P_(l, m) - (l + m) * P_(l - 1, m));
    DP_(l, l) = somx2 * x * l * P_(l, l);
  }
}

/* src/Basis.cc:95-112 */
void orc_sinecosine_R(int mmax, double phi, double *c, double *s)
{
  c[0] = 1.0;
  s[0] = 0.0;
  if (mmax > 0) {
    c[1] = cos(phi);
    s[1] = sin(phi);
    for (int m = 2; m <= mmax; m++) {
      c[m] = 2.0 * c[1] * c[m - 1] - c[m - 2];
      s[m] = 2.0 * c[1] * s[m - 1] - s[m - 2];
    }
  }
}

/* factrl(n) = n! as a double (exputil numerical-recipes style helper) */
static double factrl(int n)
{
  double a = 1.0;
  for (int i = 2; i <= n; i++) a *= (double)i;
  return a;
}

/* src/SphericalBasis.cc:328-335 */
void orc_factorial_table(int lmax, double *f)
{
  for (int l = 0; l <= lmax; l++) {
    for (int m = 0; m <= lmax; m++) f[l * (lmax + 1) + m] = 0.0;
    for (int m = 0; m <= l; m++) {
      double v = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * factrl(l - m) / factrl(l + m));
      if (m) v *= M_SQRT2;
      f[l * (lmax + 1) + m] = v;
    }
  }
}

/* exputil/SLGridMP2.cc:711-727 */
double orc_sl_r_to_xi(const orc_slgrid *g, double r)
{
  if (g->cmap == 1)      return (r / g->rmap - 1.0) / (r / g->rmap + 1.0);
  else if (g->cmap == 2) return log(r);
  return r;
}

/* exputil/SLGridMP2.cc:729-747 */
double orc_sl_xi_to_r(const orc_slgrid *g, double xi)
{
  if (g->cmap == 1)      return (1.0 + xi) / (1.0 - xi) * g->rmap;
  else if (g->cmap == 2) return exp(xi);
  return xi;
}

/* exputil/SLGridMP2.cc:749-765 */
double orc_sl_d_xi_to_r(const orc_slgrid *g, double xi)
{
  if (g->cmap == 1)      return 0.5 * (1.0 - xi) * (1.0 - xi) / g->rmap;
  else if (g->cmap == 2) return exp(-xi);
  return 1.0;
}

#define EF_(l, n, i) g->ef[((size_t)(l) * g->nmax + (n)) * g->numr + (i)]
#define EV_(l, n)    g->ev[(l) * g->nmax + (n)]

/* exputil/SLGridMP2.cc:872-910 (which=1: argument is a radius) */
void orc_sl_get_pot(const orc_slgrid *g, double r, double *mat)
{
  double x = orc_sl_r_to_xi(g, r);

  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < 0) indx = 0;
  if (indx > g->numr - 2) indx = g->numr - 2;

  double x1 = (g->xi[indx + 1] - x) / g->dxi;
  double x2 = (x - g->xi[indx]) / g->dxi;

  for (int l = 0; l <= g->lmax; l++)
    for (int n = 0; n < g->nmax; n++)
      mat[l * g->nmax + n] =
          (x1 * EF_(l, n, indx) + x2 * EF_(l, n, indx + 1)) / sqrt(EV_(l, n)) *
          (x1 * g->p0[indx] + x2 * g->p0[indx + 1]);
}

/* exputil/SLGridMP2.cc:913-950 */
void orc_sl_get_dens(const orc_slgrid *g, double r, double *mat)
{
  double x = orc_sl_r_to_xi(g, r);

  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < 0) indx = 0;
  if (indx > g->numr - 2) indx = g->numr - 2;

  double x1 = (g->xi[indx + 1] - x) / g->dxi;
  double x2 = (x - g->xi[indx]) / g->dxi;

  for (int l = 0; l <= g->lmax; l++)
    for (int n = 0; n < g->nmax; n++)
      mat[l * g->nmax + n] =
          (x1 * EF_(l, n, indx) + x2 * EF_(l, n, indx + 1)) * sqrt(EV_(l, n)) *
          (x1 * g->d0[indx] + x2 * g->d0[indx + 1]);
}

/* exputil/SLGridMP2.cc:954-989 */
void orc_sl_get_force(const orc_slgrid *g, double r, double *mat)
{
  double x = orc_sl_r_to_xi(g, r);

  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < 1) indx = 1;
  if (indx > g->numr - 2) indx = g->numr - 2;

  double p = (x - g->xi[indx]) / g->dxi;
  double fac = orc_sl_d_xi_to_r(g, x) / g->dxi;

  for (int l = 0; l <= g->lmax; l++)
    for (int n = 0; n < g->nmax; n++)
      mat[l * g->nmax + n] =
          fac * ((p - 0.5) * EF_(l, n, indx - 1) * g->p0[indx - 1]
                 - 2.0 * p * EF_(l, n, indx) * g->p0[indx]
                 + (p + 0.5) * EF_(l, n, indx + 1) * g->p0[indx + 1]) / sqrt(EV_(l, n));
}

/* scalar get_pot / get_dens with which=0 (argument already xi), used by orthoCheck:
 * exputil/SLGridMP2.cc:767-797 and :802-830 */
static double sl_pot_xi(const orc_slgrid *g, double x, int l, int n)
{
  if (g->cmap == 1) { if (x < -1.0) x = -1.0; if (x >= 1.0) x = 1.0 - XOFFSET; }
  if (g->cmap == 2) { if (x < g->xmin) x = g->xmin; if (x > g->xmax) x = g->xmax; }
  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < 0) indx = 0;
  if (indx > g->numr - 2) indx = g->numr - 2;
  double x1 = (g->xi[indx + 1] - x) / g->dxi;
  double x2 = (x - g->xi[indx]) / g->dxi;
  return (x1 * EF_(l, n, indx) + x2 * EF_(l, n, indx + 1)) / sqrt(EV_(l, n)) *
         (x1 * g->p0[indx] + x2 * g->p0[indx + 1]);
}

static double sl_dens_xi(const orc_slgrid *g, double x, int l, int n)
{
  if (g->cmap == 1) { if (x < -1.0) x = -1.0; if (x >= 1.0) x = 1.0 - XOFFSET; }
  if (g->cmap == 2) { if (x < g->xmin) x = g->xmin; if (x > g->xmax) x = g->xmax; }
  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < 0) indx = 0;
  if (indx > g->numr - 2) indx = g->numr - 2;
  double x1 = (g->xi[indx + 1] - x) / g->dxi;
  double x2 = (x - g->xi[indx]) / g->dxi;
  return (x1 * EF_(l, n, indx) + x2 * EF_(l, n, indx + 1)) * sqrt(EV_(l, n)) *
         (x1 * g->d0[indx] + x2 * g->d0[indx + 1]);
}

/* exputil/SLGridMP2.cc:1775-1824 */
void orc_sl_orthocheck(const orc_slgrid *g, int num, const double *knots,
                       const double *weights, double *ret)
{
  double ximin = orc_sl_r_to_xi(g, g->rmin);
  double ximax = orc_sl_r_to_xi(g, g->rmax);
  int nmax = g->nmax;

  for (int L = 0; L <= g->lmax; L++) {
    for (int nn = 0; nn < nmax * nmax; nn++) {
      int n1 = nn / nmax;
      int n2 = nn - n1 * nmax;
      double ans = 0.0;
      for (int i = 0; i < num; i++) {
        double x = ximin + (ximax - ximin) * knots[i];
        double r = orc_sl_xi_to_r(g, x);
        ans += r * r * sl_pot_xi(g, x, L, n1) * sl_dens_xi(g, x, L, n2) /
               orc_sl_d_xi_to_r(g, x) * (ximax - ximin) * weights[i];
      }
      ret[(L * nmax + n1) * nmax + n2] = -ans;
    }
  }
}

/* ---- per-call options (bfe_oracle.h) ---- */
static _Thread_local orc_call_opts g_opts = {1.0, 0, 1.0e20, {0, 0, 0}, {0, 0, 0}, -1, 1.0, 1};
void orc_set_call_opts(const orc_call_opts *o)
{
  if (o) g_opts = *o;
  else { orc_call_opts d = {1.0, 0, 1.0e20, {0, 0, 0}, {0, 0, 0}, -1, 1.0, 1}; g_opts = d; }
}
/* `if (ssfrac>0.0 && ssfrac<1.0) subset = true;` (src/SphericalBasis.cc:149-152) */
int orc_opt_subset(double *ssfrac, int *nthrds)
{
  const int on = g_opts.ssfrac > 0.0 && g_opts.ssfrac < 1.0;
  if (ssfrac) *ssfrac = on ? g_opts.ssfrac : 1.0;
  if (nthrds) *nthrds = g_opts.nthrds < 1 ? 1 : g_opts.nthrds;
  return on;
}
double orc_opt_adb(void) { return g_opts.adb; }
int orc_opt_mlim(int mmax) { return (g_opts.mlim >= 0 && g_opts.mlim < mmax) ? g_opts.mlim : mmax; }
/* bool Component::freeze(unsigned indx) (src/Component.cc:4194-4202) */
int orc_opt_frozen(double x, double y, double z)
{
  if (!g_opts.frz) return 0;
  const double pos[3] = {x, y, z};
  double r2 = 0.0;
  for (int i = 0; i < 3; i++) r2 +=
                                (pos[i] - g_opts.com0[i] - g_opts.fcenter[i]) *
                                (pos[i] - g_opts.com0[i] - g_opts.fcenter[i]);
  if (r2 > g_opts.rtrunc * g_opts.rtrunc) return 1;
  else return 0;
}
/* double Component::Adiabatic() (src/Component.cc:4214-4220), `adiabatic` true */
double orc_adiabatic(double tnow, double ton, double toff, double twid)
{
  return 0.25 *
    ( 1.0 + erf((tnow - ton )/twid) ) *
    ( 1.0 + erf((toff - tnow)/twid) ) ;
}

/* Kahan-compensated add used only in arbiter mode */
static inline void kadd(double *s, double *c, double v)
{
  double y = v - *c;
  double t = *s + y;
  *c = (t - *s) - y;
  *s = t;
}

/* src/SphericalBasis.cc:429-599 (determine_coefficients_thread, one thread, one
 * level, pcavar/pcaeof/subset/mix off, sqnorm == 1 for Sphere; adb and freeze: orc_set_call_opts) */
long orc_sph_accumulate(const orc_slgrid *g, const orc_sph_params *P, long nbodies,
                        const double *X, const double *Y, const double *Z,
                        const double *M, const double *center, double *coef, int kahan)
{
  const double fac0 = -4.0 * M_PI;
  const int Lmax = g->lmax, nmax = g->nmax, lmax = g->lmax;
  const int nrows = (Lmax + 1) * (Lmax + 1);
  long use = 0;

  double *p = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *cosm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *sinm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (Lmax + 1) * nmax);
  double *factorial = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *wk = (double *)malloc(sizeof(double) * nmax);
  double *comp = kahan ? (double *)calloc((size_t)nrows * nmax, sizeof(double)) : NULL;
  orc_factorial_table(Lmax, factorial);

  memset(coef, 0, sizeof(double) * nrows * nmax);

#define ACC(row, n, v)                                               \
  do {                                                               \
    if (kahan) kadd(&coef[(row) * nmax + (n)], &comp[(row) * nmax + (n)], (v)); \
    else coef[(row) * nmax + (n)] += (v);                            \
  } while (0)

  const double adb = orc_opt_adb();                                   /* :441 */
  double ssfrac;
  int nthrds;
  const int subset = orc_opt_subset(&ssfrac, &nthrds);
  /* the threads of the reference one after the other (their partial sums are added up afterwards, :873-887; here into one
   * set: with the subset off the slices tile the list and the loop is the plain one) */
  for (int id = 0; id < nthrds; id++) {
  int nbeg = (int)((unsigned long)nbodies * (unsigned long)id / (unsigned long)nthrds);        /* :438 */
  int nend = (int)((unsigned long)nbodies * (unsigned long)(id + 1) / (unsigned long)nthrds);  /* :439 */
  if (subset) nend = (int)floor(ssfrac * nend);                       /* :460 */
  for (long i = nbeg; i < nend; i++) {
    if (orc_opt_frozen(X[i], Y[i], Z[i])) continue;                   /* :468 */
    double mass = M[i] * adb;                                         /* :471 */
    if (subset) mass /= ssfrac;                                       /* :473 */
    double xx = X[i] - center[0];
    double yy = Y[i] - center[1];
    double zz = Z[i] - center[2];

    double r2 = (xx * xx + yy * yy + zz * zz);
    double r = sqrt(r2) + DSMALL;

    if (r >= P->rmin && r <= P->rmax) {
      use++;
      double costh = zz / r;
      double phi = atan2(yy, xx);
      double rs = r / P->scale;

      orc_legendre_R(Lmax, costh, p);
      orc_sinecosine_R(Lmax, phi, cosm, sinm);
      orc_sl_get_pot(g, rs, potd);

      for (int l = 0, loffset = 0; l <= Lmax; loffset += (2 * l + 1), l++) {
        for (int m = 0, moffset = 0; m <= l; m++) {
          double facL = factorial[l * (Lmax + 1) + m] * P_(l, m);
          if (m == 0) {
            for (int n = 0; n < nmax; n++) {
              wk[n] = potd[l * nmax + n] * facL * mass * fac0 / 1.0;
              ACC(loffset + moffset, n, wk[n]);
            }
            moffset++;
          } else {
            if (!P->M0_only) {
              double fac1 = facL * cosm[m];
              double fac2 = facL * sinm[m];
              for (int n = 0; n < nmax; n++) {
                wk[n] = potd[l * nmax + n] * mass * fac0 / 1.0;
                ACC(loffset + moffset, n, wk[n] * fac1);
                ACC(loffset + moffset + 1, n, wk[n] * fac2);
              }
            }
            moffset += 2;
          }
        }
      }
    }
  }
  }
#undef ACC

  free(p); free(cosm); free(sinm); free(potd); free(factorial); free(wk);
  if (comp) free(comp);
  return use;
}

/* src/SphericalBasis.cc:1797-1813 */
static void get_pot_coefs_safe(int l, int nmax, const double *coef, double *p, double *dp,
                               const double *potd1, const double *dpot1)
{
  double pp = 0.0, dpp = 0.0;
  for (int i = 0; i < nmax; i++) {
    pp  += potd1[l * nmax + i] * coef[i];
    dpp += dpot1[l * nmax + i] * coef[i];
  }
  *p = pp;
  *dp = dpp;
}

/* src/SphericalBasis.cc:1476-1660 (determine_acceleration_and_potential_thread,
 * mix off, use_external handled by the caller supplying positions + centre)      */
/* PS (may be NULL): per-particle pseudo-acceleration [n][3] of the target component's frame,
 * Component::getPseudoAccel(pos, vel) (src/Component.cc:4407-4427).  Component::AddAcc(i, j, val) does
 * acc[j] += val - pseudo[j] on EVERY call (src/Component.H:914-921), and the thread body makes five
 * of them (:1645-1651): x, y, z, and x, y once more for the potp term when fac > DSMALL.  So with EJ
 * on, the reference subtracts the x and y components TWICE; restated as written.                */
static void sph_accel_addacc(const orc_slgrid *g, const orc_sph_params *P, long nbodies,
                             const double *X, const double *Y, const double *Z,
                             const double *center, const double *expcoef, const double *PS,
                             double *AX, double *AY, double *AZ, double *POT)
{
  const int Lmax = g->lmax, nmax = g->nmax, lmax = g->lmax;
  const double scale = P->scale, rmax = P->rmax;
  const double mfactor = 1.0;

  double *p = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *dp = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *cosm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *sinm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (Lmax + 1) * nmax);
  double *dpot = (double *)malloc(sizeof(double) * (Lmax + 1) * nmax);
  double *factorial = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  orc_factorial_table(Lmax, factorial);
#define FACT(l, m) factorial[(l) * (Lmax + 1) + (m)]
#define COEF(row)  (expcoef + (size_t)(row) * nmax)

  for (long i = 0; i < nbodies; i++) {
    double r0 = 0.0, pp, dpp, pc, dpc, ps, dps, facp, facdp;
    double potr, potl, pott, potp;

    if (orc_opt_frozen(X[i], Y[i], Z[i])) continue;                   /* :1521 (cC->freeze: the TARGET component's) */
    double xx = X[i] - center[0];
    double yy = Y[i] - center[1];
    double zz = Z[i] - center[2];

    double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
    double costh = zz / r;
    double rs = r / scale;
    double phi = atan2(yy, xx);

    orc_dlegendre_R(Lmax, costh, p, dp);
    orc_sinecosine_R(Lmax, phi, cosm, sinm);

    int ioff = 0;
    if (r > rmax) {
      ioff = 1;
      r0 = r;
      r = rmax;
      rs = r / scale;
    }

    potl = potr = pott = potp = 0.0;

    orc_sl_get_pot(g, rs, potd);
    orc_sl_get_force(g, rs, dpot);

    if (!P->NO_L0) {
      get_pot_coefs_safe(0, nmax, COEF(0), &pp, &dpp, potd, dpot);
      if (ioff) {
        pp *= rmax / r0;
        dpp = -pp / r0;
      }
      double facL = mfactor * FACT(0, 0);
      potl = facL * pp;
      potr = facL * dpp;
    }

    for (int l = 1, loffset = 1; l <= Lmax; loffset += (2 * l + 1), l++) {
      if (P->NO_L1 && l == 1) continue;
      if (P->EVEN_L && (l / 2) * 2 != l) continue;

      for (int m = 0, moffset = 0; m <= l; m++) {
        double facL = FACT(l, m) * P_(l, m) * mfactor;
        double facD = FACT(l, m) * DP_(l, m) * mfactor;

        /* NB: as in the reference these `continue`s skip the moffset update */
        if (P->EVEN_M && (m / 2) * 2 != m) continue;
        if (P->M0_only && m != 0) continue;

        if (m == 0) {
          get_pot_coefs_safe(l, nmax, COEF(loffset + moffset), &pp, &dpp, potd, dpot);
          if (ioff) {
            pp *= pow(rmax / r0, (double)(l + 1));
            dpp = -pp / r0 * (l + 1);
          }
          potl += facL * pp;
          potr += facL * dpp;
          pott += facD * pp;
          moffset++;
        } else {
          get_pot_coefs_safe(l, nmax, COEF(loffset + moffset), &pc, &dpc, potd, dpot);
          get_pot_coefs_safe(l, nmax, COEF(loffset + moffset + 1), &ps, &dps, potd, dpot);
          if (ioff) {
            facp = pow(rmax / r0, (double)(l + 1));
            facdp = -1.0 / r0 * (l + 1);
            pc *= facp;
            ps *= facp;
            dpc = pc * facdp;
            dps = ps * facdp;
          }
          potl += facL * (pc * cosm[m] + ps * sinm[m]);
          potr += facL * (dpc * cosm[m] + dps * sinm[m]);
          pott += facD * (pc * cosm[m] + ps * sinm[m]);
          potp += facL * (-pc * sinm[m] + ps * cosm[m]) * m;
          moffset += 2;
        }
      }
    }

    double fac = xx * xx + yy * yy;

    potr /= scale * scale;
    potl /= scale;
    pott /= scale;
    potp /= scale;

    /* note: for r>rmax the reference divides by the CLAMPED r (= rmax) here */
    const double ps0 = PS ? PS[3 * i] : 0.0, ps1 = PS ? PS[3 * i + 1] : 0.0, ps2 = PS ? PS[3 * i + 2] : 0.0;
    AX[i] += -(potr * xx / r - pott * xx * zz / (r * r * r)) - ps0;
    AY[i] += -(potr * yy / r - pott * yy * zz / (r * r * r)) - ps1;
    AZ[i] += -(potr * zz / r + pott * fac / (r * r * r)) - ps2;
    if (fac > DSMALL) {
      AX[i] += potp * yy / fac - ps0;
      AY[i] += -potp * xx / fac - ps1;
    }
    POT[i] += potl;
  }
#undef FACT
#undef COEF

  free(p); free(dp); free(cosm); free(sinm); free(potd); free(dpot); free(factorial);
}

void orc_sph_accel(const orc_slgrid *g, const orc_sph_params *P, long nbodies,
                   const double *X, const double *Y, const double *Z,
                   const double *center, const double *expcoef,
                   double *AX, double *AY, double *AZ, double *POT)
{
  sph_accel_addacc(g, P, nbodies, X, Y, Z, center, expcoef, NULL, AX, AY, AZ, POT);
}

void orc_sph_accel_pseudo(const orc_slgrid *g, const orc_sph_params *P, long nbodies,
                          const double *X, const double *Y, const double *Z,
                          const double *center, const double *expcoef, const double *pseudo,
                          double *AX, double *AY, double *AZ, double *POT)
{
  sph_accel_addacc(g, P, nbodies, X, Y, Z, center, expcoef, pseudo, AX, AY, AZ, POT);
}

/* src/incpos.cc:15-69 */
void orc_drift(long n, double dt, double *x, double *y, double *z,
               const double *vx, const double *vy, const double *vz)
{
  for (long i = 0; i < n; i++) {
    x[i] += vx[i] * dt;
    y[i] += vy[i] * dt;
    z[i] += vz[i] * dt;
  }
}

/* src/incvel.cc:15-88 */
void orc_kick(long n, double dt, double *vx, double *vy, double *vz,
              const double *ax, const double *ay, const double *az)
{
  for (long i = 0; i < n; i++) {
    vx[i] += ax[i] * dt;
    vy[i] += ay[i] * dt;
    vz[i] += az[i] * dt;
  }
}

/* src/step.cc:271-323 (multistep=0 block) for one self-gravitating component:
 * incr_velocity(dt/2); incr_position(dt); compute_expansion(0);
 * compute_potential() [zero acc/pot: src/ComponentContainer.cc:641-665, then self
 * force :698-716]; incr_velocity(dt/2).                                           */
void orc_sph_step(const orc_slgrid *g, const orc_sph_params *P, long n, double dt,
                  double *x, double *y, double *z, double *vx, double *vy, double *vz,
                  double *ax, double *ay, double *az, double *pot,
                  const double *mass, const double *center, double *coef)
{
  orc_kick(n, 0.5 * dt, vx, vy, vz, ax, ay, az);
  orc_drift(n, dt, x, y, z, vx, vy, vz);
  orc_sph_accumulate(g, P, n, x, y, z, mass, center, coef, 0);
  for (long i = 0; i < n; i++) ax[i] = ay[i] = az[i] = pot[i] = 0.0;
  orc_sph_accel(g, P, n, x, y, z, center, coef, ax, ay, az, pot);
  orc_kick(n, 0.5 * dt, vx, vy, vz, ax, ay, az);
}

/* src/multistep.cc:630-680 */
orc_mstep_tables *orc_mstep_create(int multistep)
{
  orc_mstep_tables *t = (orc_mstep_tables *)calloc(1, sizeof(*t));
  int Mstep = 1 << multistep;
  t->multistep = multistep;
  t->Mstep = Mstep;
  t->mintvl = (int *)calloc(multistep + 1, sizeof(int));
  t->mfirst = (int *)calloc(Mstep + 1, sizeof(int));
  t->mactive = (int *)calloc((size_t)(Mstep + 1) * (multistep + 1), sizeof(int));
  t->dstepL = (int *)calloc((size_t)(multistep + 1) * Mstep, sizeof(int));
  t->dstepN = (int *)calloc((size_t)(multistep + 1) * Mstep, sizeof(int));

  t->mintvl[0] = Mstep;
  for (int n = 1; n <= multistep; n++) t->mintvl[n] = t->mintvl[n - 1] / 2;

  for (int M = 0; M <= multistep; M++) t->mactive[0 * (multistep + 1) + M] = 1;
  for (int ms = 1; ms <= Mstep; ms++)
    for (int M = 0; M <= multistep; M++)
      if ((ms % (1 << (multistep - M))) == 0) t->mactive[ms * (multistep + 1) + M] = 1;

  for (int ms = 0; ms <= Mstep; ms++)
    for (int M = 0; M <= multistep; M++)
      if (t->mactive[ms * (multistep + 1) + M]) { t->mfirst[ms] = M; break; }

  for (int ms = 0; ms <= multistep; ms++) {
    int rev = multistep - ms;
    int d = 1 << ms;
    for (int n = 0; n < Mstep; n++) {
      t->dstepL[rev * Mstep + n] = (n / d) * d;
      t->dstepN[rev * Mstep + n] = t->dstepL[rev * Mstep + n] + d;
    }
  }
  return t;
}

void orc_mstep_free(orc_mstep_tables *t)
{
  if (!t) return;
  free(t->mintvl); free(t->mfirst); free(t->mactive); free(t->dstepL); free(t->dstepN);
  free(t);
}

void orc_mstep_export(const orc_mstep_tables *t, int *mintvl, int *mfirst,
                      int *mactive, int *dstepL, int *dstepN)
{
  int ms = t->multistep, M = t->Mstep;
  memcpy(mintvl, t->mintvl, sizeof(int) * (ms + 1));
  memcpy(mfirst, t->mfirst, sizeof(int) * (M + 1));
  memcpy(mactive, t->mactive, sizeof(int) * (size_t)(M + 1) * (ms + 1));
  memcpy(dstepL, t->dstepL, sizeof(int) * (size_t)(ms + 1) * M);
  memcpy(dstepN, t->dstepN, sizeof(int) * (size_t)(ms + 1) * M);
}

/* src/SphericalBasis.cc:1231-1333 ; src/CylEXP.cc:192-282 */
void orc_mstep_combine(const orc_mstep_tables *t, int mdrft, long ncoef,
                       const double *coefL, const double *coefN, double *coef)
{
  for (long k = 0; k < ncoef; k++) coef[k] = 0.0;

  for (int M = 0; M < t->mfirst[mdrft]; M++) {
    double numer = (double)(mdrft - t->dstepL[M * t->Mstep + mdrft]);
    double denom = (double)(t->dstepN[M * t->Mstep + mdrft] - t->dstepL[M * t->Mstep + mdrft]);
    double b = numer / denom;
    double a = 1.0 - b;
    for (long k = 0; k < ncoef; k++)
      coef[k] += a * coefL[(size_t)M * ncoef + k] + b * coefN[(size_t)M * ncoef + k];
  }
  for (int M = t->mfirst[mdrft]; M <= t->multistep; M++)
    for (long k = 0; k < ncoef; k++) coef[k] += coefN[(size_t)M * ncoef + k];
}

/* src/multistep.cc:94-130: the five criteria and the smallest of them, dt = max(eps, .) */
double orc_level_dt(const double *dynfrac, double scale, const double *v, const double *a, double pot)
{
  const double eps = 1.0e-10;
  const double dynfracD = dynfrac[0], dynfracV = dynfrac[1], dynfracS = dynfrac[2],
               dynfracA = dynfrac[3], dynfracP = dynfrac[4];
  double dtr = 0.0, vtot = 0.0, atot = 0.0;
  for (int k = 0; k < 3; k++) {
    dtr += v[k] * a[k];
    vtot += v[k] * v[k];
    atot += a[k] * a[k];
  }
  double ptot = fabs(pot);
  double dts, dtd, dtv, dta, dtA;
  if (scale > 0) dts = dynfracS * scale / fabs(sqrt(vtot) + eps);
  else           dts = 1.0 / eps;
  dtd = dynfracD * 1.0 / sqrt(vtot + eps);
  dtv = dynfracV * sqrt(vtot / (atot + eps));
  dta = dynfracA * ptot / (fabs(dtr) + eps);
  dtA = dynfracP * sqrt(ptot / (atot + eps));

  /* smallest key of the std::map (dta, dtA only if > 0) */
  double dmin = dtd;
  if (dtv < dmin) dmin = dtv;
  if (dts < dmin) dmin = dts;
  if (dta > 0.0 && dta < dmin) dmin = dta;
  if (dtA > 0.0 && dtA < dmin) dmin = dtA;

  return dmin > eps ? dmin : eps;
}

/* src/multistep.cc:160-196: the level rule on Particle::dtreq (a float, include/Particle.H:60-61) */
int orc_level_rule(double dtime, int multistep, int mfirst_mdrft, int cur_level, int shiftlevl, float dtreq_f)
{
  unsigned plev = (unsigned)cur_level;
  unsigned nlev = plev;
  if (dtreq_f > dtime) nlev = 0;
  else nlev = (unsigned)(int)floor(log(dtime / dtreq_f) / log(2.0));

  if (shiftlevl) {
    if (nlev > plev) {
      if (nlev - plev > (unsigned)shiftlevl) nlev = plev + shiftlevl;
    } else if (plev > nlev) {
      if (plev - nlev > (unsigned)shiftlevl) nlev = plev - shiftlevl;
    }
  }
  if (nlev > (unsigned)multistep) nlev = multistep;
  if ((int)nlev < mfirst_mdrft) nlev = mfirst_mdrft;
  return (int)nlev;
}

/* src/multistep.cc:94-196 with NoSwitch off (`p->dtreq = dt`, :144); FreezeLev and NoSwitch are the caller's:
 * nbody_oracle.c adjust_levels */
int orc_level_select(double dtime, int multistep, int mfirst_mdrft, int cur_level,
                     int shiftlevl, const double *dynfrac, double scale,
                     const double *v, const double *a, double pot, double *dtreq)
{
  /* Particle::dtreq is a float: the level rule sees the float-rounded value */
  float dtreq_f = (float)orc_level_dt(dynfrac, scale, v, a, pot);
  *dtreq = (double)dtreq_f;
  return orc_level_rule(dtime, multistep, mfirst_mdrft, cur_level, shiftlevl, dtreq_f);
}

/* ---- block-multistep master step for one self-gravitating spherical component ------------
 * do_step's multistep block (src/step.cc:98-269) with ComponentContainer::compute_expansion /
 * compute_potential (src/ComponentContainer.cc:1173-1226, :580-727) for a single component,
 * SphericalBasis::determine_coefficients_particles' N/L swap (src/SphericalBasis.cc:785-792),
 * compute_multistep_coefficients (:1231-1333), adjust_multistep_level (src/multistep.cc:344-627)
 * with multistep_update / _finish (src/SphericalBasis.cc:1033-1079, :1156-1228).
 * State carried by the caller: particle arrays, integer levels, coefN/coefL[(ms+1)][ncoef].
 * this_step == 0 reproduces the reference's "do all levels" rule on the first sub-step.       */

/* one particle's coefficient contribution (multistep_update window: r < rmax only) */
static void sph_one_particle(const orc_slgrid *g, const orc_sph_params *P, double xx, double yy,
                             double zz, double mass, double *val /* [(L+1)^2*nmax] */,
                             double *p, double *cosm, double *sinm, double *potd,
                             const double *factorial, int *inside)
{
  const int Lmax = g->lmax, nmax = g->nmax, lmax = g->lmax;
  const double fac0 = -4.0 * M_PI;
  double r2 = (xx * xx + yy * yy + zz * zz);
  double r = sqrt(r2) + DSMALL;
  *inside = 0;
  if (r < P->rmax) {
    *inside = 1;
    double costh = zz / r;
    double phi = atan2(yy, xx);
    double rs = r / P->scale;
    orc_legendre_R(Lmax, costh, p);
    orc_sinecosine_R(Lmax, phi, cosm, sinm);
    orc_sl_get_pot(g, rs, potd);
    for (int l = 0, loffset = 0; l <= Lmax; loffset += (2 * l + 1), l++) {
      for (int m = 0, moffset = 0; m <= l; m++) {
        double facL = factorial[l * (Lmax + 1) + m] * P_(l, m);
        if (m == 0) {
          for (int n = 0; n < nmax; n++)
            val[(loffset + moffset) * nmax + n] = potd[l * nmax + n] * facL * mass * fac0 / 1.0;
          moffset++;
        } else {
          double fac1 = facL * cosm[m];
          double fac2 = facL * sinm[m];
          for (int n = 0; n < nmax; n++) {
            val[(loffset + moffset) * nmax + n] = potd[l * nmax + n] * fac1 * mass * fac0 / 1.0;
            val[(loffset + moffset + 1) * nmax + n] = potd[l * nmax + n] * fac2 * mass * fac0 / 1.0;
          }
          moffset += 2;
        }
      }
    }
  }
}


/* public form of the above for the multi-component step loop (nbody_oracle.c): val[(L+1)^2*nmax] gets
 * the particle's contribution; returns 1 when r < rmax (SphericalBasis::multistep_update,
 * src/SphericalBasis.cc:1156-1228) */
int orc_sph_multistep_update(const orc_slgrid *g, const orc_sph_params *P, double xx, double yy,
                             double zz, double mass, double *val)
{
  const int L1 = g->lmax + 1;
  double p[L1 * L1], cosm[L1], sinm[L1], potd[L1 * g->nmax], factorial[L1 * L1];
  int inside = 0;
  orc_factorial_table(g->lmax, factorial);
  sph_one_particle(g, P, xx, yy, zz, mass, val, p, cosm, sinm, potd, factorial, &inside);
  return inside;
}

/* adjust_multistep_level (src/multistep.cc:344-627) + multistep_update / _finish
 * (src/SphericalBasis.cc:1033-1079, :1156-1228) for one spherical component              */
static long sph_adjust_levels(const orc_slgrid *g, const orc_sph_params *P, const orc_mstep_tables *T,
                              int multistep, double dtime, const double *dynfrac, int shiftlevl,
                              int mdrft, int all_levels, long n, const double *x, const double *y,
                              const double *z, const double *vx, const double *vy, const double *vz,
                              const double *ax, const double *ay, const double *az,
                              const double *pot, const double *mass, int *level,
                              const double *center, double *coefN, double *differ, double *val,
                              double *p, double *cosm, double *sinm, double *potd,
                              const double *factorial)
{
  const long ncoef = (long)(g->lmax + 1) * (g->lmax + 1) * g->nmax;
  long switched = 0;
  int first = T->mfirst[mdrft];
  if (all_levels) first = 0;
  for (int M = T->mfirst[mdrft]; M <= multistep; M++)
    memset(differ + (size_t)M * ncoef, 0, sizeof(double) * ncoef);
  for (int lev = first; lev <= multistep; lev++) {
    for (long i = 0; i < n; i++) {
      if (level[i] != lev) continue;
      double v[3] = {vx[i], vy[i], vz[i]}, a[3] = {ax[i], ay[i], az[i]};
      double dtreq;
      int nlev = orc_level_select(dtime, multistep, T->mfirst[mdrft], lev, shiftlevl, dynfrac, 0.0,
                                  v, a, pot[i], &dtreq);
      if (nlev != lev) {
        int inside;
        sph_one_particle(g, P, x[i] - center[0], y[i] - center[1], z[i] - center[2], mass[i], val,
                         p, cosm, sinm, potd, factorial, &inside);
        if (inside) {
          /* levels below mfirst[mdrft] are never cleared/added by _begin/_finish */
          for (long q = 0; q < ncoef; q++) {
            if (lev >= T->mfirst[mdrft]) differ[(size_t)lev * ncoef + q] -= val[q];
            if (nlev >= T->mfirst[mdrft]) differ[(size_t)nlev * ncoef + q] += val[q];
          }
        }
        level[i] = -(nlev + 1); /* commit after the sweep so that a particle is seen once */
        switched++;
      }
    }
  }
  for (long i = 0; i < n; i++)
    if (level[i] < 0) level[i] = -level[i] - 1;
  for (int M = T->mfirst[mdrft]; M <= multistep; M++)
    for (long q = 0; q < ncoef; q++) coefN[(size_t)M * ncoef + q] += differ[(size_t)M * ncoef + q];
  return switched;
}

void orc_sph_multistep_step(const orc_slgrid *g, const orc_sph_params *P, int multistep,
                            double dtime, const double *dynfrac, int shiftlevl, long n,
                            double *x, double *y, double *z, double *vx, double *vy, double *vz,
                            double *ax, double *ay, double *az, double *pot, const double *mass,
                            int *level, const double *center, double *coefN, double *coefL,
                            int this_step, double *coef_out, long *nswitch)
{
  const int Lmax = g->lmax, nmax = g->nmax;
  const long ncoef = (long)(Lmax + 1) * (Lmax + 1) * nmax;
  orc_mstep_tables *T = orc_mstep_create(multistep);
  const int Mstep = T->Mstep;
  const double dt = dtime / Mstep;

  double *tx = (double *)malloc(sizeof(double) * n), *ty = (double *)malloc(sizeof(double) * n),
         *tz = (double *)malloc(sizeof(double) * n), *tm = (double *)malloc(sizeof(double) * n);
  double *differ = (double *)malloc(sizeof(double) * (multistep + 1) * ncoef);
  double *val = (double *)malloc(sizeof(double) * ncoef);
  double *p = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *cosm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *sinm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (Lmax + 1) * nmax);
  double *factorial = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *tmpc = (double *)malloc(sizeof(double) * ncoef);
  orc_factorial_table(Lmax, factorial);
  long switched = 0;

  for (int mstep = 0; mstep < Mstep; mstep++) {
    int mdrft = mstep;
    for (int M = T->mfirst[mstep]; M <= multistep; M++) {
      double DT = dt * T->mintvl[M];
      /* incr_velocity(0.5*DT, M); incr_position(DT, M) */
      for (long i = 0; i < n; i++)
        if (level[i] == M) {
          vx[i] += ax[i] * (0.5 * DT); vy[i] += ay[i] * (0.5 * DT); vz[i] += az[i] * (0.5 * DT);
          x[i] += vx[i] * DT; y[i] += vy[i] * DT; z[i] += vz[i] * DT;
        }
      /* compute_expansion(M): swap N/L, accumulate level M into N[M] */
      memcpy(coefL + (size_t)M * ncoef, coefN + (size_t)M * ncoef, sizeof(double) * ncoef);
      long k = 0;
      for (long i = 0; i < n; i++)
        if (level[i] == M) { tx[k] = x[i]; ty[k] = y[i]; tz[k] = z[i]; tm[k] = mass[i]; k++; }
      orc_sph_accumulate(g, P, k, tx, ty, tz, tm, center, coefN + (size_t)M * ncoef, 0);
    }
    mdrft = mstep + 1;
    /* compute_potential(mfirst[mstep]) */
    const int mlev = T->mfirst[mstep];
    orc_mstep_combine(T, mdrft, ncoef, coefL, coefN, coef_out);
    for (long i = 0; i < n; i++)
      if (level[i] >= mlev) {
        ax[i] = ay[i] = az[i] = pot[i] = 0.0;
        orc_sph_accel(g, P, 1, x + i, y + i, z + i, center, coef_out, ax + i, ay + i, az + i, pot + i);
      }
    /* second half kick for the levels active at the next sub-step */
    for (int M = T->mfirst[mdrft]; M <= multistep; M++) {
      double DT = dt * T->mintvl[M];
      for (long i = 0; i < n; i++)
        if (level[i] == M) {
          vx[i] += ax[i] * (0.5 * DT); vy[i] += ay[i] * (0.5 * DT); vz[i] += az[i] * (0.5 * DT);
        }
    }
    /* adjust_multistep_level */
    switched += sph_adjust_levels(g, P, T, multistep, dtime, dynfrac, shiftlevl, mdrft,
                                  (this_step == 0 && mstep == 0), n, x, y, z, vx, vy, vz, ax, ay, az,
                                  pot, mass, level, center, coefN, differ, val, p, cosm, sinm, potd,
                                  factorial);
  }
  if (nswitch) *nswitch = switched;
  orc_mstep_free(T);
  free(tx); free(ty); free(tz); free(tm); free(differ); free(val); free(p); free(cosm); free(sinm);
  free(potd); free(factorial); free(tmpc);
}

/* begin_run's multistep initialisation (src/begin.cc:80-129): expansion at every level, full
 * potential, first level assignment (all levels examined), then expansion + potential again.  */
void orc_sph_multistep_init(const orc_slgrid *g, const orc_sph_params *P, int multistep,
                            double dtime, const double *dynfrac, int shiftlevl, long n,
                            const double *x, const double *y, const double *z, const double *vx,
                            const double *vy, const double *vz, double *ax, double *ay, double *az,
                            double *pot, const double *mass, int *level, const double *center,
                            double *coefN, double *coefL, double *coef_out)
{
  const int Lmax = g->lmax, nmax = g->nmax;
  const long ncoef = (long)(Lmax + 1) * (Lmax + 1) * nmax;
  orc_mstep_tables *T = orc_mstep_create(multistep);
  double *tx = (double *)malloc(sizeof(double) * n), *ty = (double *)malloc(sizeof(double) * n),
         *tz = (double *)malloc(sizeof(double) * n), *tm = (double *)malloc(sizeof(double) * n);
  double *differ = (double *)malloc(sizeof(double) * (multistep + 1) * ncoef);
  double *val = (double *)malloc(sizeof(double) * ncoef);
  double *p = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  double *cosm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *sinm = (double *)malloc(sizeof(double) * (Lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (Lmax + 1) * nmax);
  double *factorial = (double *)malloc(sizeof(double) * (Lmax + 1) * (Lmax + 1));
  orc_factorial_table(Lmax, factorial);

  for (int pass = 0; pass < 2; pass++) {
    for (int M = 0; M <= multistep; M++) {
      memcpy(coefL + (size_t)M * ncoef, coefN + (size_t)M * ncoef, sizeof(double) * ncoef);
      long k = 0;
      for (long i = 0; i < n; i++)
        if (level[i] == M) { tx[k] = x[i]; ty[k] = y[i]; tz[k] = z[i]; tm[k] = mass[i]; k++; }
      orc_sph_accumulate(g, P, k, tx, ty, tz, tm, center, coefN + (size_t)M * ncoef, 0);
    }
    orc_mstep_combine(T, 0, ncoef, coefL, coefN, coef_out);
    for (long i = 0; i < n; i++) ax[i] = ay[i] = az[i] = pot[i] = 0.0;
    orc_sph_accel(g, P, n, x, y, z, center, coef_out, ax, ay, az, pot);
    if (pass == 0)
      sph_adjust_levels(g, P, T, multistep, dtime, dynfrac, shiftlevl, 0, 1, n, x, y, z, vx, vy, vz,
                        ax, ay, az, pot, mass, level, center, coefN, differ, val, p, cosm, sinm,
                        potd, factorial);
  }
  orc_mstep_free(T);
  free(tx); free(ty); free(tz); free(tm); free(differ); free(val); free(p); free(cosm); free(sinm);
  free(potd); free(factorial);
}

/* ---- pyEXP field evaluation ------------------------------------------------------------------
 * expui/BiorthBasis.cc:711-816 (Spherical::sph_eval), :930-941 (cyl_eval), :946-958 (crt_eval);
 * normalisation table :318-330 (lgamma form of factorial(l,m)); labels :71-97.
 * coord: 0 = spherical input (r, cos(theta), phi), 1 = cylindrical (R, z, phi), 2 = Cartesian
 * (x, y, z).  out[9] = {dens m=0, dens m>0, dens, potl m=0, potl m>0, potl, three force
 * components in the input coordinate system}.  G = 1, N1..N2 = the full radial range.         */
static void pyexp_sph_eval(const orc_slgrid *g, const orc_sph_params *P, const double *coef,
                           const double *factorial, double r, double costh, double phi,
                           double *dend, double *potd, double *dpot, double *legs, double *dlegs,
                           double *out)
{
  const int lmax = g->lmax, nmax = g->nmax;
  const double scale = P->scale;
#define FC(l, m) factorial[(l) * (lmax + 1) + (m)]
#define LG(l, m) legs[(l) * (lmax + 1) + (m)]
#define DLG(l, m) dlegs[(l) * (lmax + 1) + (m)]
#define CF(row, n) coef[(size_t)(row) * nmax + (n)]
  double fac1 = FC(0, 0);

  orc_sl_get_dens(g, r / scale, dend);
  orc_sl_get_pot(g, r / scale, potd);
  orc_sl_get_force(g, r / scale, dpot);

  orc_dlegendre_R(lmax, costh, legs, dlegs);

  double den0, pot0, potr;
  double sinth = sqrt(fabs(1.0 - costh * costh));

  if (P->NO_L0) {
    den0 = 0.0;
    pot0 = 0.0;
    potr = 0.0;
  } else {
    den0 = pot0 = potr = 0.0;
    for (int n = 0; n < nmax; n++) {
      den0 += CF(0, n) * dend[n];
      pot0 += CF(0, n) * potd[n];
      potr += CF(0, n) * dpot[n];
    }
    den0 *= fac1;
    pot0 *= fac1;
    potr *= fac1;
  }

  double den1 = 0.0, pot1 = 0.0, pott = 0.0, potp = 0.0;
  /* the radial window of the l >= 1 sums (:761, :780); the l = 0 term above takes every n */
  const int nlo = P->N1 > 0 ? P->N1 : 0;
  const int nhi = (P->N2 >= 0 && P->N2 < nmax - 1) ? P->N2 : nmax - 1;

  for (int l = 1, loffset = 1; l <= lmax; loffset += (2 * l + 1), l++) {
    if (P->EVEN_L && l % 2) continue;
    if (P->NO_L1 && l == 1) continue;

    for (int m = 0, moffset = 0; m <= l; m++) {
      if (P->M0_only && m) continue;
      if (P->EVEN_M && m % 2) continue;

      fac1 = FC(l, m);
      if (m == 0) {
        double sumR = 0.0, sumP = 0.0, sumD = 0.0;
        for (int n = nlo; n <= nhi; n++) {
          sumR += CF(loffset + moffset, n) * dend[l * nmax + n];
          sumP += CF(loffset + moffset, n) * potd[l * nmax + n];
          sumD += CF(loffset + moffset, n) * dpot[l * nmax + n];
        }
        den1 += fac1 * LG(l, m) * sumR;
        pot1 += fac1 * LG(l, m) * sumP;
        potr += fac1 * LG(l, m) * sumD;
        pott += fac1 * DLG(l, m) * sumP;
        moffset++;
      } else {
        double cosm = cos(phi * m);
        double sinm = sin(phi * m);
        double sumR0 = 0.0, sumP0 = 0.0, sumD0 = 0.0;
        double sumR1 = 0.0, sumP1 = 0.0, sumD1 = 0.0;
        for (int n = nlo; n <= nhi; n++) {
          sumR0 += CF(loffset + moffset + 0, n) * dend[l * nmax + n];
          sumP0 += CF(loffset + moffset + 0, n) * potd[l * nmax + n];
          sumD0 += CF(loffset + moffset + 0, n) * dpot[l * nmax + n];
          sumR1 += CF(loffset + moffset + 1, n) * dend[l * nmax + n];
          sumP1 += CF(loffset + moffset + 1, n) * potd[l * nmax + n];
          sumD1 += CF(loffset + moffset + 1, n) * dpot[l * nmax + n];
        }
        den1 += fac1 * LG(l, m) * (sumR0 * cosm + sumR1 * sinm);
        pot1 += fac1 * LG(l, m) * (sumP0 * cosm + sumP1 * sinm);
        potr += fac1 * LG(l, m) * (sumD0 * cosm + sumD1 * sinm);
        pott += fac1 * DLG(l, m) * (sumP0 * cosm + sumP1 * sinm);
        potp += fac1 * LG(l, m) * (-sumP0 * sinm + sumP1 * cosm) * m;
        moffset += 2;
      }
    }
  }

  double densfac = 1.0 / (scale * scale * scale) * 0.25 / M_PI;
  double potlfac = 1.0 / scale;

  out[0] = den0 * densfac;
  out[1] = den1 * densfac;
  out[2] = (den0 + den1) * densfac;
  out[3] = pot0 * potlfac;
  out[4] = pot1 * potlfac;
  out[5] = (pot0 + pot1) * potlfac;
  out[6] = potr * (-potlfac) / scale;
  out[7] = pott * (-potlfac) / r;
  out[8] = potp * (-potlfac) / (r * sinth);
#undef FC
#undef LG
#undef DLG
#undef CF
}

/* expui/BiorthBasis.cc:323-329: factorial(l, m) in the lgamma form pyEXP uses */
static void pyexp_factorial(int lmax, double *factorial)
{
  for (int l = 0; l <= lmax; l++)
    for (int m = 0; m <= lmax; m++) {
      double v = 0.0;
      if (m <= l) {
        v = sqrt((0.5 * l + 0.25) / M_PI * exp(lgamma(1.0 + l - m) - lgamma(1.0 + l + m)));
        if (m != 0) v *= M_SQRT2;
      }
      factorial[l * (lmax + 1) + m] = v;
    }
}

/* Spherical::accumulate (expui/BiorthBasis.cc:583-665), the coefficient part */
long orc_pyexp_sph_accumulate(const orc_slgrid *g, const orc_sph_params *P, long nbodies, const double *X,
                              const double *Y, const double *Z, const double *M, double *expcoef)
{
  const int lmax = g->lmax, nmax = g->nmax;
  double *factorial = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *legs = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  pyexp_factorial(lmax, factorial);
  long used = 0;
  const double norm = -4.0 * M_PI;
  const double dsmall = 1.0e-20;
  for (long i = 0; i < nbodies; i++) {
    double x = X[i], y = Y[i], z = Z[i], mass = M[i];
    double fac, fac1, fac2, fac4;
    double r2 = (x * x + y * y + z * z);
    double r = sqrt(r2) + dsmall;
    double costh = z / r;
    double phi = atan2(y, x);
    double rs = r / P->scale;

    if (r < P->rmin || r > P->rmax) continue;

    used++;
    orc_sl_get_pot(g, rs, potd);
    orc_legendre_R(lmax, costh, legs);

    for (int l = 0, loffset = 0; l <= lmax; loffset += (2 * l + 1), l++) {
      for (int m = 0, moffset = 0; m <= l; m++) {
        fac = factorial[l * (lmax + 1) + m] * legs[l * (lmax + 1) + m];
        if (m == 0) {
          for (int n = 0; n < nmax; n++) {
            fac4 = potd[l * nmax + n] * fac;
            expcoef[(size_t)(loffset + moffset) * nmax + n] += fac4 * norm * mass;
          }
          moffset++;
        } else {
          fac1 = fac * cos(phi * m);
          fac2 = fac * sin(phi * m);
          for (int n = 0; n < nmax; n++) {
            fac4 = potd[l * nmax + n];
            expcoef[(size_t)(loffset + moffset) * nmax + n] += fac1 * fac4 * norm * mass;
            expcoef[(size_t)(loffset + moffset + 1) * nmax + n] += fac2 * fac4 * norm * mass;
          }
          moffset += 2;
        }
      }
    }
  }
  free(factorial); free(potd); free(legs);
  return used;
}

/* Spherical::computeAccel (expui/BiorthBasis.cc:818-926), G = 1 */
void orc_pyexp_sph_accel(const orc_slgrid *g, const orc_sph_params *P, const double *expcoef, long nbodies,
                         const double *X, const double *Y, const double *Z, double *acc)
{
  const int lmax = g->lmax, nmax = g->nmax;
  const double scale = P->scale;
  double *factorial = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *dpot = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *legs = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *dlegs = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  pyexp_factorial(lmax, factorial);
  const int nlo = P->N1 > 0 ? P->N1 : 0;
  const int nhi = (P->N2 >= 0 && P->N2 < nmax - 1) ? P->N2 : nmax - 1;
#define FC(l, m) factorial[(l) * (lmax + 1) + (m)]
#define LG(l, m) legs[(l) * (lmax + 1) + (m)]
#define DLG(l, m) dlegs[(l) * (lmax + 1) + (m)]
#define CF(row, n) expcoef[(size_t)(row) * nmax + (n)]
  for (long i = 0; i < nbodies; i++) {
    double x = X[i], y = Y[i], z = Z[i];
    double R2 = x * x + y * y;
    double r2 = R2 + z * z;
    double R = sqrt(x * x + y * y) + 1.0e-18;
    double r = sqrt(r2) + 1.0e-18;
    double costh = z / r;
    double sinth = R / r;
    double phi = atan2(y, x);

    double fac1 = FC(0, 0);
    orc_sl_get_pot(g, r / scale, potd);
    orc_sl_get_force(g, r / scale, dpot);
    orc_dlegendre_R(lmax, costh, legs, dlegs);

    double pot0, potr;
    if (P->NO_L0) {
      pot0 = 0.0;
      potr = 0.0;
    } else {
      pot0 = potr = 0.0;
      for (int n = 0; n < nmax; n++) {          /* expcoef.row(0).dot(...): every n */
        pot0 += CF(0, n) * potd[n];
        potr += CF(0, n) * dpot[n];
      }
      pot0 *= fac1;
      potr *= fac1;
    }
    (void)pot0;
    double pot1 = 0.0, pott = 0.0, potp = 0.0;
    for (int l = 1, loffset = 1; l <= lmax; loffset += (2 * l + 1), l++) {
      if (P->EVEN_L && l % 2) continue;
      if (P->NO_L1 && l == 1) continue;
      for (int m = 0, moffset = 0; m <= l; m++) {
        if (P->M0_only && m) continue;
        if (P->EVEN_M && m % 2) continue;
        fac1 = FC(l, m);
        if (m == 0) {
          double sumP = 0.0, sumD = 0.0;
          for (int n = nlo; n <= nhi; n++) {
            sumP += CF(loffset + moffset, n) * potd[l * nmax + n];
            sumD += CF(loffset + moffset, n) * dpot[l * nmax + n];
          }
          pot1 += fac1 * LG(l, m) * sumP;
          potr += fac1 * LG(l, m) * sumD;
          pott += fac1 * DLG(l, m) * sumP;
          moffset++;
        } else {
          double cosm = cos(phi * m);
          double sinm = sin(phi * m);
          double sumP0 = 0.0, sumD0 = 0.0, sumP1 = 0.0, sumD1 = 0.0;
          for (int n = nlo; n <= nhi; n++) {
            sumP0 += CF(loffset + moffset + 0, n) * potd[l * nmax + n];
            sumD0 += CF(loffset + moffset + 0, n) * dpot[l * nmax + n];
            sumP1 += CF(loffset + moffset + 1, n) * potd[l * nmax + n];
            sumD1 += CF(loffset + moffset + 1, n) * dpot[l * nmax + n];
          }
          pot1 += fac1 * LG(l, m) * (sumP0 * cosm + sumP1 * sinm);
          potr += fac1 * LG(l, m) * (sumD0 * cosm + sumD1 * sinm);
          pott += fac1 * DLG(l, m) * (sumP0 * cosm + sumP1 * sinm);
          potp += fac1 * LG(l, m) * (-sumP0 * sinm + sumP1 * cosm) * m;
          moffset += 2;
        }
      }
    }
    (void)pot1;
    double potlfac = 1.0 / scale;
    potr *= (-potlfac) / scale;
    pott *= (-potlfac);
    potp *= (-potlfac);
    /* transform to Cartesian components; R2 is the unguarded x^2 + y^2 (:917-919) */
    acc[3 * i + 0] = (potr - pott * costh / r) * x / r - potp * y / R2;
    acc[3 * i + 1] = (potr - pott * costh / r) * y / r + potp * x / R2;
    acc[3 * i + 2] = potr * costh + pott * sinth * sinth / r;
  }
#undef FC
#undef LG
#undef DLG
#undef CF
  free(factorial); free(potd); free(dpot); free(legs); free(dlegs);
}

void orc_pyexp_sph_fields(const orc_slgrid *g, const orc_sph_params *P, const double *coef, long n,
                          const double *c1, const double *c2, const double *c3, int coord,
                          double *out)
{
  const int lmax = g->lmax, nmax = g->nmax;
  double *factorial = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *dend = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *potd = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *dpot = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *legs = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *dlegs = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  /* expui/BiorthBasis.cc:323-329 */
  for (int l = 0; l <= lmax; l++)
    for (int m = 0; m <= lmax; m++) {
      double v = 0.0;
      if (m <= l) {
        v = sqrt((0.5 * l + 0.25) / M_PI * exp(lgamma(1.0 + l - m) - lgamma(1.0 + l + m)));
        if (m != 0) v *= M_SQRT2;
      }
      factorial[l * (lmax + 1) + m] = v;
    }
  for (long i = 0; i < n; i++) {
    double v[9];
    double *o = out + 9 * i;
    if (coord == 0) {
      pyexp_sph_eval(g, P, coef, factorial, c1[i], c2[i], c3[i], dend, potd, dpot, legs, dlegs, o);
      continue;
    }
    double R, z, phi, x = 0.0, y = 0.0;
    if (coord == 1) { R = c1[i]; z = c2[i]; phi = c3[i]; }
    else {                                     /* crt_eval */
      x = c1[i]; y = c2[i]; z = c3[i];
      R = sqrt(x * x + y * y) + 1.0e-18;
      phi = atan2(y, x);
    }
    /* cyl_eval */
    double r = sqrt(R * R + z * z) + 1.0e-18;
    double costh = z / r, sinth = R / r;
    pyexp_sph_eval(g, P, coef, factorial, r, costh, phi, dend, potd, dpot, legs, dlegs, v);
    double potR = v[6] * sinth - v[7] * costh * R / r;
    double potz = v[6] * costh + v[7] * sinth * R / r;
    for (int k = 0; k < 6; k++) o[k] = v[k];
    if (coord == 1) { o[6] = potR; o[7] = potz; o[8] = v[8]; }
    else {
      o[6] = potR * x / R - v[8] * y / R;
      o[7] = potR * y / R + v[8] * x / R;
      o[8] = potz;
    }
  }
  free(factorial); free(dend); free(potd); free(dpot); free(legs); free(dlegs);
}

/* ---- Component::fix_positions ---------------------------------------------------------------
 * src/Component.cc:3280-3351 (per-level sums over the levels >= mlevel; escape/tidal and frozen
 * particles not modelled), :3363-3369 (only those levels are zeroed first), :3485-3503 (levels
 * summed), :3541-3545 (division by the total mass).  lev_sums[(multistep+1)][10] carries the
 * per-level {m, m x, m v, m a} sums between calls, as com_mas / com_lev / cov_lev / coa_lev do.  */
void orc_fix_positions(long n, const double *mass, const double *x, const double *y, const double *z,
                       const double *vx, const double *vy, const double *vz, const double *ax,
                       const double *ay, const double *az, const int *level, int multistep,
                       int mlevel, double *lev_sums, double *out)
{
  for (int mm = mlevel; mm <= multistep; mm++)
    for (int k = 0; k < 10; k++) lev_sums[mm * 10 + k] = 0.0;
  for (int mm = mlevel; mm <= multistep; mm++) {
    double *s = lev_sums + mm * 10;
    for (long i = 0; i < n; i++) {
      if ((level ? level[i] : 0) != mm) continue;
      s[0] += mass[i];
      s[1] += mass[i] * x[i];  s[2] += mass[i] * y[i];  s[3] += mass[i] * z[i];
      s[4] += mass[i] * vx[i]; s[5] += mass[i] * vy[i]; s[6] += mass[i] * vz[i];
      s[7] += mass[i] * ax[i]; s[8] += mass[i] * ay[i]; s[9] += mass[i] * az[i];
    }
  }
  for (int k = 0; k < 10; k++) out[k] = 0.0;
  for (int mm = 0; mm <= multistep; mm++)
    for (int k = 0; k < 10; k++) out[k] += lev_sums[mm * 10 + k];
  if (out[0] > 0.0)
    for (int k = 1; k < 10; k++) out[k] /= out[0];
}

/* The same with the two per-particle tests of the thread body (src/Component.cc:3317-3336):
 *   consp (the component's "tidal" key, :998-1000) -- a particle beyond rcom of com0 + center (escape_com, :4204-4212) that is
 *   not yet flagged gets iattr = 1 and is left out; a flagged particle is left out from then on (iattr: iattrib[tidal],
 *   in/out, NULL = consp off).  Only the particles of the levels >= mlevel are examined.  com_system is off in this scope, so
 *   the comE / covE sums of :3322-3329 and the com0 update of :3520-3537 are not formed;
 *   freeze (:3336, :4194-4202) -- a particle beyond rtrunc of com0 + center is left out (rtrunc >= 1e20: off).
 * Products are rounded before they are added (the reference's translation unit is compiled without contraction).       */
static int orc_beyond(const double *com0, const double *center, double rad, double px, double py, double pz)
{
  const double p[3] = {px, py, pz};
  double r2 = 0.0;
  for (int i = 0; i < 3; i++) {
    volatile double t = (p[i] - com0[i] - center[i]) * (p[i] - com0[i] - center[i]);
    r2 += t;
  }
  volatile double lim = rad * rad;
  return r2 > lim;
}

void orc_fix_positions_opts(long n, const double *mass, const double *x, const double *y, const double *z,
                            const double *vx, const double *vy, const double *vz, const double *ax,
                            const double *ay, const double *az, const int *level, int multistep,
                            int mlevel, const double *com0, const double *center, double rcom, int *iattr,
                            double rtrunc, double *lev_sums, double *out)
{
  for (int mm = mlevel; mm <= multistep; mm++)
    for (int k = 0; k < 10; k++) lev_sums[mm * 10 + k] = 0.0;
  for (int mm = mlevel; mm <= multistep; mm++) {
    double *s = lev_sums + mm * 10;
    for (long i = 0; i < n; i++) {
      if ((level ? level[i] : 0) != mm) continue;
      if (iattr) {
        if (orc_beyond(com0, center, rcom, x[i], y[i], z[i]) && iattr[i] == 0) { iattr[i] = 1; continue; }
        if (iattr[i] == 1) continue;
      }
      if (rtrunc < 1.0e20 && orc_beyond(com0, center, rtrunc, x[i], y[i], z[i])) continue;
      s[0] += mass[i];
      s[1] += mass[i] * x[i];  s[2] += mass[i] * y[i];  s[3] += mass[i] * z[i];
      s[4] += mass[i] * vx[i]; s[5] += mass[i] * vy[i]; s[6] += mass[i] * vz[i];
      s[7] += mass[i] * ax[i]; s[8] += mass[i] * ay[i]; s[9] += mass[i] * az[i];
    }
  }
  for (int k = 0; k < 10; k++) out[k] = 0.0;
  for (int mm = 0; mm <= multistep; mm++)
    for (int k = 0; k < 10; k++) out[k] += lev_sums[mm * 10 + k];
  if (out[0] > 0.0)
    for (int k = 1; k < 10; k++) out[k] /= out[0];
}

/* ---- Orient ------------------------------------------------------------------------------------
 * return_euler_slater (exputil/euler_slater.cc:46-76); row-major, BODY != 0 transposes.         */
void orc_euler_slater(double phi, double theta, double psi, int body, double *o)
{
  double sph = sin(phi), cph = cos(phi), sth = sin(theta), cth = cos(theta), sps = sin(psi),
         cps = cos(psi);
  double e[3][3];
  e[0][0] = -sps * sph + cth * cph * cps; e[0][1] = sps * cph + cth * sph * cps; e[0][2] = cps * sth;
  e[1][0] = -cps * sph - cth * cph * sps; e[1][1] = cps * cph - cth * sph * sps; e[1][2] = -sps * sth;
  e[2][0] = -sth * cph;                   e[2][1] = -sth * sph;                  e[2][2] = cth;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[3 * i + j] = body ? e[j][i] : e[i][j];
}

/* Orient::Orient (src/Orient.cc:38-80) */
void orc_orient_init(orc_orient *o, int keep, int many, unsigned oflags, unsigned cflags,
                     double deltaT, double damp)
{
  memset(o, 0, sizeof(*o));
  o->keep = keep; o->many = many; o->oflags = oflags; o->cflags = cflags;
  o->deltaT = deltaT; o->damp = damp;
  o->lasttime = -DBL_MAX;
  o->axis[2] = 1.0;
  for (int k = 0; k < 3; k++) o->body[4 * k] = o->orig[4 * k] = 1.0;
}

/* one whitespace-separated number of a log row, as `line >> x` takes it: *p advances past the
 * token; *eof is set once the scan touches the end of the row (an istringstream's eofbit)        */
static double orc_row_number(const char **p, int *eof)
{
  char *e;
  while (**p == ' ' || **p == '\t') (*p)++;
  if (!**p) { *eof = 1; return 0.0; }
  double v = strtod(*p, &e);
  *p = e;
  if (!*e) *eof = 1;
  return v;
}

/* The restart block of Orient::Orient (src/Orient.cc:84-335), one process: a missing log gets its
 * two header rows (:236-284); an existing one moves to <logfile>.bak and -- with `restart` -- its
 * data rows up to tnow + 0.1*dtime/Mstep are copied into a fresh <logfile> and rebuild Ecurr, axis,
 * axis1, centre, centre0, centre1, the last `keep` entries of the two histories and body/orig.
 * queue7[naccel][7] / *nq receive what the reference hands its PseudoAccel: (time, the row's logged
 * pseudo-acceleration, axis1) (:174-186).  Returns the rows taken, -1 on a file error.            */
long orc_orient_restart(orc_orient *o, const char *logfile, int restart, double tnow, double dtime,
                        int Mstep, int naccel, double *queue7, int *nq)
{
  static const char *lab[33] = {
      "Time", "E_curr", "Used", "X-axis(reg)", "Y-axis(reg)", "Z-axis(reg)", "X-axis(cur)", "Y-axis(cur)",
      "Z-axis(cur)", "X-center(anl)", "Y-center(anl)", "Z-center(anl)", "X-center(reg)", "Y-center(reg)",
      "Z-center(reg)", "X-center(cur)", "Y-center(cur)", "Z-center(cur)", "X-com(cur)", "Y-com(cur)",
      "Z-com(cur)", "X-com(dif)", "Y-com(dif)", "Z-com(dif)", "X-accel", "Y-accel", "Z-accel", "Omega_X",
      "Omega_Y", "Omega_Z", "dOmega/dt_X", "dOmega/dt_Y", "dOmega/dt_Z"};
  if (nq) *nq = 0;
  FILE *in = fopen(logfile, "r");
  if (!in) {
    FILE *out = fopen(logfile, "w");
    if (!out) return -1;
    for (int k = 0; k < 33; k++) {
      char cell[32];
      snprintf(cell, sizeof cell, "%s%s", k ? "| " : "# ", lab[k]);
      fprintf(out, "%-15s", cell);
    }
    fputc('\n', out);
    for (int k = 0; k < 33; k++) {
      char num[16];
      int w = snprintf(num, sizeof num, "%d", k + 1);
      fputs(k ? "| " : "# ", out);
      fputs(num, out);
      for (; w < 13; w++) fputc('-', out);
    }
    fputc('\n', out);
    fclose(out);
    return 0;
  }
  fclose(in);
  size_t L = strlen(logfile);
  char *bak = (char *)malloc(L + 5);
  memcpy(bak, logfile, L);
  memcpy(bak + L, ".bak", 5);
  if (rename(logfile, bak)) { free(bak); return -1; }
  FILE *out = fopen(logfile, "w");
  in = fopen(bak, "r");
  free(bak);
  if (!out || !in) { if (out) fclose(out); if (in) fclose(in); return -1; }
  long rows = 0;
  char *row = (char *)malloc(16384);
  while (restart && fgets(row, 16384, in)) {
    size_t n = strlen(row);
    if (!n || row[n - 1] != '\n') break;           /* a last row without its newline is not taken */
    row[n - 1] = 0;
    if (row[0] == '#') continue;
    const char *p = row;
    int eof = 0;
    double time = orc_row_number(&p, &eof);
    if (tnow + 0.1 * dtime / Mstep < time) break;
    fprintf(out, "%s\n", row);
    o->Ecurr = orc_row_number(&p, &eof);
    (void)orc_row_number(&p, &eof);                /* tused */
    for (int k = 0; k < 3; k++) o->axis[k] = orc_row_number(&p, &eof);
    for (int k = 0; k < 3; k++) o->axis1[k] = orc_row_number(&p, &eof);
    for (int k = 0; k < 3; k++) o->center[k] = orc_row_number(&p, &eof);
    for (int k = 0; k < 3; k++) o->center0[k] = orc_row_number(&p, &eof);
    for (int k = 0; k < 3; k++) o->center1[k] = orc_row_number(&p, &eof);
    rows++;
    if (o->oflags & 1u) {
      if (o->nA == o->keep) {                      /* push_back, then pop_front beyond keep */
        memmove(o->tA, o->tA + 1, (size_t)(o->nA - 1) * sizeof(double));
        memmove(o->vA, o->vA + 1, (size_t)(o->nA - 1) * sizeof(o->vA[0]));
        o->nA--;
      }
      o->tA[o->nA] = time;
      memcpy(o->vA[o->nA++], o->axis1, sizeof(o->axis1));
    }
    if (o->oflags & 2u) {
      if (o->nC == o->keep) {
        memmove(o->tC, o->tC + 1, (size_t)(o->nC - 1) * sizeof(double));
        memmove(o->vC, o->vC + 1, (size_t)(o->nC - 1) * sizeof(o->vC[0]));
        o->nC--;
      }
      o->tC[o->nC] = time;
      memcpy(o->vC[o->nC++], o->center1, sizeof(o->center1));
    }
    double pseudo[3] = {0, 0, 0};
    int all = 1;
    for (int i = 0; i < 3; i++) {
      if (eof) { all = 0; break; }
      for (int k = 0; k < 3; k++) pseudo[k] = orc_row_number(&p, &eof);
    }
    if (all && naccel > 0 && queue7 && nq) {
      if (*nq == naccel) { memmove(queue7, queue7 + 7, (size_t)(naccel - 1) * 7 * sizeof(double)); (*nq)--; }
      double *q = queue7 + 7 * (*nq)++;
      q[0] = time;
      for (int k = 0; k < 3; k++) { q[1 + k] = pseudo[k]; q[4 + k] = o->axis1[k]; }
    }
  }
  free(row);
  fclose(in);
  fclose(out);
  if (o->oflags & 1u) {                            /* (:325-335) */
    double phi = atan2(o->axis[1], o->axis[0]);
    double theta = -acos(o->axis[2] / sqrt(o->axis[0] * o->axis[0] + o->axis[1] * o->axis[1] + o->axis[2] * o->axis[2]));
    orc_euler_slater(phi, theta, 0.0, 0, o->body);
    orc_euler_slater(phi, theta, 0.0, 1, o->orig);
  }
  return rows;
}

/* Orient::logEntry (src/Orient.cc:742-785): setw(15) in the stream's default format is %15.6g */
int orc_orient_log_entry(const orc_orient *o, const char *logfile, double time, const double *com,
                         const double *com0, const double *accel, const double *omega, const double *domdt)
{
  FILE *f = fopen(logfile, "a");
  if (!f) return -1;
  const double *cols[10] = {o->axis, o->axis1, o->center, o->center0, o->center1, com, com0, accel, omega, domdt};
  fprintf(f, "%15.6g%15.6g%15ld", time, o->Ecurr, o->used);
  for (int c = 0; c < 10; c++)
    for (int k = 0; k < 3; k++) fprintf(f, "%15.6g", cols[c][k]);
  fputc('\n', f);
  fclose(f);
  return 0;
}

typedef struct { double E, M, L[3], R[3]; } orc_el3;

/* the linear least-squares blocks (:576-604 axis, :620-676 centre); N as the reference passes it */
static void orc_orient_regress(int n, const double *t, double (*v)[3], int N, double damp,
                               double time, double *val, double *sig, double *sigz)
{
  double sumX = 0, sumX2 = 0, sumY[3] = {0, 0, 0}, sumXY[3] = {0, 0, 0}, slope[3], icpt[3];
  for (int j = 0; j < n; j++) {
    double x = t[j];
    sumX += x;
    sumX2 += x * x;
    for (int k = 0; k < 3; k++) { sumY[k] += v[j][k]; sumXY[k] += v[j][k] * x; }
  }
  for (int k = 0; k < 3; k++) {
    slope[k] = (sumXY[k] * N - sumX * sumY[k]) / (sumX2 * N - sumX * sumX);
    icpt[k] = (sumX2 * sumY[k] - sumX * sumXY[k]) / (sumX2 * N - sumX * sumX);
    val[k] = icpt[k] + slope[k] * (damp * time + (1.0 - damp) * t[0]);
  }
  *sig = 0.0;
  if (sigz) *sigz = 0.0;
  for (int j = 0; j < n; j++) {
    for (int k = 0; k < 3; k++) {
      double d = v[j][k] - icpt[k] - slope[k] * t[j];
      *sig += d * d;
      if (k == 2 && sigz) *sigz += d * d;
    }
  }
  *sig /= n;
  if (sigz) *sigz /= n;
}

static void orc_hist_pop(int *n, double *t, double (*v)[3])
{
  for (int j = 1; j < *n; j++) { t[j - 1] = t[j]; memcpy(v[j - 1], v[j], sizeof(v[0])); }
  (*n)--;
}

/* Orient::accumulate (src/Orient.cc:420-747) with accumulate_cpu (:325-417) inlined; numprocs = 1.
 * The std::set ordered on E alone is restated as a sorted array that, like the set, refuses a
 * second entry of equal energy and never holds more than many+1 entries.                        */
void orc_orient_accumulate(orc_orient *o, double time, double dtime, long n, const double *mass,
                           const double *x, const double *y, const double *z, const double *vx,
                           const double *vy, const double *vz, const double *pot)
{
  if (fabs(o->lasttime - time) < 1.0e-12) return;
  if (time - o->deltaT - o->lasttime < 0.0) return;
  o->lasttime = time;
  if (o->linear) {
    for (int k = 0; k < 3; k++) { o->center[k] = o->center0[k]; o->center0[k] += o->cenvel0[k] * dtime; }
    return;
  }
  long tkeep = o->many, size = 0;
  orc_el3 *angm = (orc_el3 *)malloc((size_t)(tkeep + 2) * sizeof(orc_el3));
  for (long i = 0; i < n; i++) {
    double pos[3] = {x[i], y[i], z[i]}, vel[3] = {vx[i], vy[i], vz[i]}, psa[3], v2 = 0.0;
    for (int k = 0; k < 3; k++) { psa[k] = pos[k] - o->center[k]; v2 += vel[k] * vel[k]; }
    double energy = pot[i];
    if (o->cflags & 2u) energy += 0.5 * v2;
    int test1 = size <= tkeep, test2 = 1;
    if (size) test2 = energy < angm[size - 1].E;
    if (!(test1 || test2)) continue;
    orc_el3 t;
    t.E = energy; t.M = mass[i];
    t.L[0] = mass[i] * (psa[1] * vel[2] - psa[2] * vel[1]);
    t.L[1] = mass[i] * (psa[2] * vel[0] - psa[0] * vel[2]);
    t.L[2] = mass[i] * (psa[0] * vel[1] - psa[1] * vel[0]);
    for (int k = 0; k < 3; k++) t.R[k] = mass[i] * pos[k];
    if (test2 && !test1) size--;                       /* erase the largest energy */
    long lo = 0, hi = size;                            /* lower bound on E          */
    while (lo < hi) { long mid = (lo + hi) / 2; if (angm[mid].E < energy) lo = mid + 1; else hi = mid; }
    if (lo < size && !(energy < angm[lo].E)) continue; /* equal key: set::insert is a no-op */
    memmove(angm + lo + 1, angm + lo, (size_t)(size - lo) * sizeof(orc_el3));
    angm[lo] = t;
    size++;
  }
  /* ee = the stored energies, already sorted (:452-478) */
  if (size) o->Ecurr = (size <= o->many) ? angm[size - 1].E : angm[o->many].E;
  double mtot = 0.0;
  long cnum = 0;
  for (int k = 0; k < 3; k++) o->axis1[k] = o->center1[k] = 0.0;
  for (long i = 0; i < size && angm[i].E < o->Ecurr; i++) {
    for (int k = 0; k < 3; k++) { o->axis1[k] += angm[i].L[k]; o->center1[k] += angm[i].R[k]; }
    mtot += angm[i].M;
    cnum++;
  }
  free(angm);
  o->used = cnum;
  o->mtot = mtot;
  if (mtot > 0.0) {
    for (int k = 0; k < 3; k++) { o->axis1[k] /= mtot; o->center1[k] /= mtot; }
    if ((o->oflags & 1u) && o->nA < ORC_ORIENT_HIST) {
      o->tA[o->nA] = time; memcpy(o->vA[o->nA], o->axis1, sizeof(o->axis1)); o->nA++;
    }
    if ((o->oflags & 2u) && o->nC < ORC_ORIENT_HIST) {
      o->tC[o->nC] = time; memcpy(o->vC[o->nC], o->center1, sizeof(o->center1)); o->nC++;
    }
  }
  if (o->nA > o->keep + 1) {
    orc_hist_pop(&o->nA, o->tA, o->vA);
    orc_orient_regress(o->nA, o->tA, o->vA, o->nC /* sic, :583 */, o->damp, time, o->axis, &o->sigA, 0);
    double phi = atan2(o->axis[1], o->axis[0]);
    double theta = -acos(o->axis[2] / sqrt(o->axis[0] * o->axis[0] + o->axis[1] * o->axis[1] +
                                           o->axis[2] * o->axis[2]));
    orc_euler_slater(phi, theta, 0.0, 0, o->body);
    orc_euler_slater(phi, theta, 0.0, 1, o->orig);
  }
  if (o->nC > 1) {
    if (o->nC > o->keep + 1) orc_hist_pop(&o->nC, o->tC, o->vC);
    orc_orient_regress(o->nC, o->tC, o->vC, o->nC, o->damp, time, o->center, &o->sigC, &o->sigCz);
  }
  if (o->keep > 1) {
    if (o->nC > 1) {
      double factor = (double)(o->nC - o->keep) / o->keep;
      factor = factor * factor;
      for (int k = 0; k < 3; k++) o->center[k] = o->center0[k] * factor + o->center[k] * (1.0 - factor);
    } else
      for (int k = 0; k < 3; k++) o->center[k] = o->center0[k];
  } else
    for (int k = 0; k < 3; k++) o->center[k] = o->center1[k];
  for (int k = 0; k < 3; k++) o->center0[k] += o->cenvel0[k] * dtime;
}

/* QuadLS::fit (include/QuadLS.H:17-53) */
void orc_quadls(int n, const double *x, const double *y, double *out)
{
  out[0] = out[1] = out[2] = 0.0;
  if (n <= 0) return;
  double sumx = 0, sumy = 0, sumxy = 0, sumx2y = 0, sumx2 = 0, sumx3 = 0, sumx4 = 0;
  for (int i = 0; i < n; i++) {
    sumx += x[i];
    sumy += y[i];
    sumx2 += x[i] * x[i];
    sumxy += x[i] * y[i];
    sumx2y += x[i] * x[i] * y[i];
    sumx3 += x[i] * x[i] * x[i];
    sumx4 += x[i] * x[i] * x[i] * x[i];
  }
  double Sxx = sumx2 - sumx * sumx / n, Sxy = sumxy - sumx * sumy / n;
  double Sxx2 = sumx3 - sumx * sumx2 / n, Sx2y = sumx2y - sumx2 * sumy / n;
  double Sx2x2 = sumx4 - sumx2 * sumx2 / n;
  double denom = Sxx * Sx2x2 - Sxx2 * Sxx2;
  if (fabs(denom) > 0.0) {
    out[0] = (Sx2y * Sxx - Sxy * Sxx2) / denom;
    out[1] = (Sxy * Sx2x2 - Sx2y * Sxx2) / denom;
    out[2] = (sumy - sumx2 * out[0] - sumx * out[1]) / n;
  }
}

/* PseudoAccel::operator() (include/PseudoAccel.H:45-91), queue full */
void orc_pseudo_accel_fit(int n, const double *rows, double *accel, double *omega, double *domdt)
{
  double *t = (double *)malloc(sizeof(double) * n), *v = (double *)malloc(sizeof(double) * n), q[3];
  for (int i = 0; i < n; i++) t[i] = rows[7 * i];
  for (int k = 0; k < 3; k++) {
    for (int i = 0; i < n; i++) v[i] = rows[7 * i + 1 + k];
    orc_quadls(n, t, v, q);
    accel[k] = 2.0 * q[0];
  }
  double T = t[n - 1], nn[3], dn[3], d2n[3];
  for (int k = 0; k < 3; k++) {
    for (int i = 0; i < n; i++) v[i] = rows[7 * i + 4 + k];
    orc_quadls(n, t, v, q);
    nn[k] = q[0] * T * T + q[1] * T + q[2];
    dn[k] = 2.0 * q[0] * T + q[1];
    d2n[k] = 2.0 * q[0];
  }
  omega[0] = nn[1] * dn[2] - nn[2] * dn[1];
  omega[1] = nn[2] * dn[0] - nn[0] * dn[2];
  omega[2] = nn[0] * dn[1] - nn[1] * dn[0];
  domdt[0] = nn[1] * d2n[2] - nn[2] * d2n[1];
  domdt[1] = nn[2] * d2n[0] - nn[0] * d2n[2];
  domdt[2] = nn[0] * d2n[1] - nn[1] * d2n[0];
  free(t); free(v);
}

static void orc_cross(const double *a, const double *b, double *c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Component::getPseudoAccel (src/Component.cc:4407-4427) */
void orc_get_pseudo_accel(int center, int axis, const double *accel, const double *omega,
                          const double *domdt, const double *pos, const double *vel, double *out)
{
  out[0] = out[1] = out[2] = 0.0;
  if (center) for (int k = 0; k < 3; k++) out[k] += accel[k];
  if (axis) {
    double a[3], b[3], c[3], d[3];
    orc_cross(omega, vel, a);
    orc_cross(domdt, pos, b);
    orc_cross(omega, pos, c);
    orc_cross(omega, c, d);
    for (int k = 0; k < 3; k++) out[k] += 2.0 * a[k] + b[k] + d[k];
  }
}

/* Spherical::accumulate with pcavar (expui/BiorthBasis.cc:583-665): window r < rmin or r > rmax
 * -> skip, dsmall 1e-20, used++ then T = used % sampT, g = exp(i m phi) potd.row(l) fac norm,
 * meanV[T][L] += g mass, covrV[T][L] += g g^dagger mass.                                        */
long orc_pyexp_sph_covariance(const orc_slgrid *g, const orc_sph_params *P, long n,
                              const double *X, const double *Y, const double *Z, const double *M,
                              int sampT, long used0, long *counts, double *masses, double *mean,
                              double *covr)
{
  const int lmax = g->lmax, nmax = g->nmax;
  const double norm = -4.0 * M_PI, dsmall = 1.0e-20;
  double *factorial = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *potd = (double *)malloc(sizeof(double) * (lmax + 1) * nmax);
  double *p = (double *)malloc(sizeof(double) * (lmax + 1) * (lmax + 1));
  double *gr = (double *)malloc(sizeof(double) * nmax), *gi = (double *)malloc(sizeof(double) * nmax);
  orc_factorial_table(lmax, factorial);
  long used = used0, accepted = 0;
  for (long i = 0; i < n; i++) {
    double x = X[i], y = Y[i], z = Z[i], mass = M[i];
    double r2 = x * x + y * y + z * z;
    double r = sqrt(r2) + dsmall;
    double costh = z / r, phi = atan2(y, x), rs = r / P->scale;
    if (r < P->rmin || r > P->rmax) continue;
    used++; accepted++;
    orc_sl_get_pot(g, rs, potd);
    orc_legendre_R(lmax, costh, p);
    int T = (int)(used % sampT);
    counts[T] += 1;
    masses[T] += mass;
    for (int l = 0, L = 0; l <= lmax; l++) {
      for (int m = 0; m <= l; m++, L++) {
        double fac = factorial[l * (lmax + 1) + m] * P_(l, m);
        double c = cos(m * phi), s = sin(m * phi);
        for (int k = 0; k < nmax; k++) {
          double v = potd[l * nmax + k] * fac * norm;
          gr[k] = c * v;
          gi[k] = s * v;
        }
        double *mv = mean + (((size_t)T * ((lmax + 1) * (lmax + 2) / 2) + L) * nmax) * 2;
        double *cv = covr + ((size_t)T * ((lmax + 1) * (lmax + 2) / 2) + L) * nmax * nmax;
        for (int k = 0; k < nmax; k++) {
          mv[2 * k] += gr[k] * mass;
          mv[2 * k + 1] += gi[k] * mass;
          for (int k2 = 0; k2 < nmax; k2++) cv[k * nmax + k2] += (gr[k] * gr[k2] + gi[k] * gi[k2]) * mass;
        }
      }
    }
  }
  free(factorial); free(potd); free(p); free(gr); free(gi);
  return accepted;
}
