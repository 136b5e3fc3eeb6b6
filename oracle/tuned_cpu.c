/*
 * tuned_cpu.c -- a TUNED CPU statement of the spherical hot path (TEST / BASELINE INFRASTRUCTURE ONLY).
 *
 * SURVEY section 8d asks for the CPU baseline in two forms: the reference's structure (bfe_oracle.c:
 * per-particle table look-ups with the divisions by sqrt(ev) inside the (l, n) loops,
 * src/SphericalBasis.cc:429-599, :1476-1660) and "an honest upper bound for CPUs".  This file is the
 * second: the same arithmetic with the n-dependence hoisted out of the particle loops exactly as the
 * device path does it --
 *   accumulate: per radial cell i the moments W[i][row][0|1] = sum w x1 | w x2, then ONE contraction
 *               coef[row][n] = sum_i E[i][l][n] W[i][row][0] + E[i+1][l][n] W[i][row][1],
 *               E = ef / sqrt(ev)  (get_pot is linear in the two table columns of the cell)
 *   force:      G[i][row] = sum_n E[i][l][n] coef[row][n] once per step; per particle the (l, m)
 *               sums read G (potential: linear interpolation; radial derivative: the 3-point
 *               formula of get_force on p0 G)
 * so a particle costs O(L^2) instead of O(L^2 nmax).  Results equal bfe_oracle.c's to round-off
 * (tests/test_oracle_kat.py).  Nothing under exp_amd/ uses this file.
 */
#include "bfe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DSMALL 1.0e-16

struct orc_tuned {
  int lmax, nmax, numr, nrows;
  double *E;        /* [numr][lmax+1][nmax]  ef / sqrt(ev)      */
  double *fac;      /* [(lmax+1)^2] factorial(l, m)              */
  int *row_l;       /* [nrows] l of every real row               */
};

orc_tuned *orc_tuned_create(const orc_slgrid *g)
{
  orc_tuned *t = (orc_tuned *)calloc(1, sizeof(*t));
  t->lmax = g->lmax; t->nmax = g->nmax; t->numr = g->numr;
  t->nrows = (g->lmax + 1) * (g->lmax + 1);
  t->E = (double *)malloc(sizeof(double) * (size_t)g->numr * (g->lmax + 1) * g->nmax);
  for (int i = 0; i < g->numr; i++)
    for (int l = 0; l <= g->lmax; l++)
      for (int n = 0; n < g->nmax; n++)
        t->E[((size_t)i * (g->lmax + 1) + l) * g->nmax + n] =
            g->ef[((size_t)l * g->nmax + n) * g->numr + i] / sqrt(g->ev[l * g->nmax + n]);
  t->fac = (double *)malloc(sizeof(double) * t->nrows);
  orc_factorial_table(g->lmax, t->fac);
  t->row_l = (int *)malloc(sizeof(int) * t->nrows);
  for (int l = 0, row = 0; l <= g->lmax; l++)
    for (int k = 0; k < 2 * l + 1; k++) t->row_l[row++] = l;
  return t;
}

void orc_tuned_free(orc_tuned *t)
{
  if (!t) return;
  free(t->E); free(t->fac); free(t->row_l); free(t);
}

static inline int cell_of(const orc_slgrid *g, double x, int lo)
{
  int indx = (int)((x - g->xmin) / g->dxi);
  if (indx < lo) indx = lo;
  if (indx > g->numr - 2) indx = g->numr - 2;
  return indx;
}

/* moments of one slice of particles: W[(numr-1)][nrows][2] (caller zeroes / sums over threads) */
long orc_tuned_moments(const orc_slgrid *g, const orc_tuned *t, const orc_sph_params *P, long n,
                       const double *X, const double *Y, const double *Z, const double *M,
                       const double *center, double *W)
{
  const int L = g->lmax, nrows = t->nrows;
  double *p = (double *)malloc(sizeof(double) * (L + 1) * (L + 1));
  double *cosm = (double *)malloc(sizeof(double) * (L + 1)), *sinm = (double *)malloc(sizeof(double) * (L + 1));
  long use = 0;
  for (long i = 0; i < n; i++) {
    double xx = X[i] - center[0], yy = Y[i] - center[1], zz = Z[i] - center[2];
    double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
    if (!(r >= P->rmin && r <= P->rmax)) continue;
    use++;
    orc_legendre_R(L, zz / r, p);
    orc_sinecosine_R(L, atan2(yy, xx), cosm, sinm);
    double x = orc_sl_r_to_xi(g, r / P->scale);
    int idx = cell_of(g, x, 0);
    double x1 = (g->xi[idx + 1] - x) / g->dxi, x2 = (x - g->xi[idx]) / g->dxi;
    double w0 = M[i] * (-4.0 * M_PI) * (x1 * g->p0[idx] + x2 * g->p0[idx + 1]);
    double *w = W + (size_t)idx * nrows * 2;
    for (int l = 0, row = 0; l <= L; l++)
      for (int m = 0; m <= l; m++) {
        double f = t->fac[l * (L + 1) + m] * p[l * (L + 1) + m] * w0;
        if (m == 0) { w[2 * row] += f * x1; w[2 * row + 1] += f * x2; row++; }
        else {
          if (!P->M0_only) {
            w[2 * row] += f * cosm[m] * x1;       w[2 * row + 1] += f * cosm[m] * x2;
            w[2 * row + 2] += f * sinm[m] * x1;   w[2 * row + 3] += f * sinm[m] * x2;
          }
          row += 2;
        }
      }
  }
  free(p); free(cosm); free(sinm);
  return use;
}

/* coef[row][n] = sum_i E[i][l][n] W[i][row][0] + E[i+1][l][n] W[i][row][1] */
void orc_tuned_contract(const orc_slgrid *g, const orc_tuned *t, const double *W, double *coef)
{
  const int nrows = t->nrows, nmax = g->nmax, stride = (g->lmax + 1) * nmax;
  memset(coef, 0, sizeof(double) * nrows * nmax);
  for (int i = 0; i < g->numr - 1; i++) {
    const double *w = W + (size_t)i * nrows * 2;
    for (int row = 0; row < nrows; row++) {
      double a = w[2 * row], b = w[2 * row + 1];
      if (a == 0.0 && b == 0.0) continue;
      const double *e0 = t->E + (size_t)i * stride + t->row_l[row] * nmax, *e1 = e0 + stride;
      double *c = coef + (size_t)row * nmax;
      for (int k = 0; k < nmax; k++) c[k] += a * e0[k] + b * e1[k];
    }
  }
}

/* G[i][row] = sum_n E[i][l][n] coef[row][n] */
void orc_tuned_project(const orc_slgrid *g, const orc_tuned *t, const double *coef, double *G)
{
  const int nrows = t->nrows, nmax = g->nmax, stride = (g->lmax + 1) * nmax;
  for (int i = 0; i < g->numr; i++)
    for (int row = 0; row < nrows; row++) {
      const double *e = t->E + (size_t)i * stride + t->row_l[row] * nmax, *c = coef + (size_t)row * nmax;
      double s = 0.0;
      for (int k = 0; k < nmax; k++) s += e[k] * c[k];
      G[(size_t)i * nrows + row] = s;
    }
}

/* orc_sph_accel with the n-sums replaced by look-ups in G */
void orc_tuned_accel(const orc_slgrid *g, const orc_tuned *t, const orc_sph_params *P, long n,
                     const double *X, const double *Y, const double *Z, const double *center,
                     const double *G, double *AX, double *AY, double *AZ, double *POT)
{
  const int L = g->lmax, nrows = t->nrows;
  const double scale = P->scale, rmax = P->rmax;
  double *p = (double *)malloc(sizeof(double) * (L + 1) * (L + 1));
  double *dp = (double *)malloc(sizeof(double) * (L + 1) * (L + 1));
  double *cosm = (double *)malloc(sizeof(double) * (L + 1)), *sinm = (double *)malloc(sizeof(double) * (L + 1));
  for (long i = 0; i < n; i++) {
    double xx = X[i] - center[0], yy = Y[i] - center[1], zz = Z[i] - center[2];
    double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL, r0 = 0.0;
    double costh = zz / r, phi = atan2(yy, xx);
    orc_dlegendre_R(L, costh, p, dp);
    orc_sinecosine_R(L, phi, cosm, sinm);
    int ioff = 0;
    if (r > rmax) { ioff = 1; r0 = r; r = rmax; }
    double x = orc_sl_r_to_xi(g, r / scale);
    int ip = cell_of(g, x, 0), jf = cell_of(g, x, 1);
    double x1 = (g->xi[ip + 1] - x) / g->dxi, x2 = (x - g->xi[ip]) / g->dxi;
    double P0 = x1 * g->p0[ip] + x2 * g->p0[ip + 1];
    double pf = (x - g->xi[jf]) / g->dxi, ffac = orc_sl_d_xi_to_r(g, x) / g->dxi;
    const double *g0 = G + (size_t)ip * nrows, *g1 = g0 + nrows;
    const double *h0 = G + (size_t)(jf - 1) * nrows, *h1 = h0 + nrows, *h2 = h1 + nrows;
    double q0 = (pf - 0.5) * g->p0[jf - 1], q1 = -2.0 * pf * g->p0[jf], q2 = (pf + 0.5) * g->p0[jf + 1];
    double potl = 0, potr = 0, pott = 0, potp = 0;
#define PV(row)  (P0 * (x1 * g0[row] + x2 * g1[row]))
#define DV(row)  (ffac * (q0 * h0[row] + q1 * h1[row] + q2 * h2[row]))
    for (int l = 0, loffset = 0; l <= L; loffset += (2 * l + 1), l++) {
      if (l == 0 && P->NO_L0) continue;
      if (P->NO_L1 && l == 1) continue;
      if (l > 0 && P->EVEN_L && (l / 2) * 2 != l) continue;
      double facp = ioff ? pow(rmax / r0, (double)(l + 1)) : 1.0;
      double facdp = ioff ? -1.0 / r0 * (l + 1) : 0.0;
      for (int m = 0, moffset = 0; m <= l; m++) {
        double facL = t->fac[l * (L + 1) + m] * p[l * (L + 1) + m];
        double facD = t->fac[l * (L + 1) + m] * dp[l * (L + 1) + m];
        if (l > 0 && P->EVEN_M && (m / 2) * 2 != m) continue;          /* (skips the moffset update) */
        if (l > 0 && P->M0_only && m != 0) continue;
        if (m == 0) {
          double pp = PV(loffset + moffset), dpp = DV(loffset + moffset);
          if (ioff) { pp *= facp; dpp = pp * facdp; }
          potl += facL * pp; potr += facL * dpp;
          if (l) pott += facD * pp;
          moffset++;
        } else {
          double pc = PV(loffset + moffset), dpc = DV(loffset + moffset);
          double ps = PV(loffset + moffset + 1), dps = DV(loffset + moffset + 1);
          if (ioff) { pc *= facp; ps *= facp; dpc = pc * facdp; dps = ps * facdp; }
          potl += facL * (pc * cosm[m] + ps * sinm[m]);
          potr += facL * (dpc * cosm[m] + dps * sinm[m]);
          pott += facD * (pc * cosm[m] + ps * sinm[m]);
          potp += facL * (-pc * sinm[m] + ps * cosm[m]) * m;
          moffset += 2;
        }
      }
    }
#undef PV
#undef DV
    double fac = xx * xx + yy * yy;
    potr /= scale * scale; potl /= scale; pott /= scale; potp /= scale;
    AX[i] += -(potr * xx / r - pott * xx * zz / (r * r * r));
    AY[i] += -(potr * yy / r - pott * yy * zz / (r * r * r));
    AZ[i] += -(potr * zz / r + pott * fac / (r * r * r));
    if (fac > DSMALL) { AX[i] += potp * yy / fac; AY[i] += -potp * xx / fac; }
    POT[i] += potl;
  }
  free(p); free(dp); free(cosm); free(sinm);
}
