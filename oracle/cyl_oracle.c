/*
 * cyl_oracle.c -- CPU restatement of EXP's cylindrical (EmpCylSL / Cylinder) hot path.
 * TEST INFRASTRUCTURE ONLY; see bfe_oracle.h for the scope statement (parity unpinned).
 *
 * Tables are given: potC/rforceC/zforceC[m][n] and potS/rforceS/zforceS[m>=1][n] on the
 * (NUMX+1) x (NUMY+1) grid of EmpCylSL::setup_table (exputil/EmpCylSL.cc:2123-2137), stored
 * here as tab[kind][m][n][ix][iy] with kind 0..5 = potC, rforceC, zforceC, potS, rforceS, zforceS.
 */
#include "cyl_oracle.h"
#include "bfe_oracle.h"      /* the per-call options (orc_opt_*) */

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DSMALL 1.0e-16 /* src/expand.H:130 */

#define TAB(kind, m, n, ix, iy)                                                              \
  g->tab[((((size_t)(kind) * (g->mmax + 1) + (m)) * g->norder + (n)) * (g->numx + 1) + (ix)) * \
             (g->numy + 1) + (iy)]

/* exputil/EmpCylSL.cc:6446-6463 */
double orc_cyl_r_to_xi(const orc_cylgrid *g, double r)
{
  if (g->cmapr > 0) return (r / g->ascale - 1.0) / (r / g->ascale + 1.0);
  return r;
}

/* exputil/EmpCylSL.cc:7109-7117 */
double orc_cyl_z_to_y(const orc_cylgrid *g, double z)
{
  if (g->cmapz == 1) return z / (fabs(z) + DBL_MIN) * asinh(fabs(z / g->hscale));
  else if (g->cmapz == 2) return z / sqrt(z * z + g->hscale * g->hscale);
  return z;
}

/* cell + bilinear weights shared by get_pot (:5567-5597) and accumulated_eval (:5280-5314);
 * enforce_limits == false (exputil/EmpCylSL.cc:56) */
static void cyl_weights(const orc_cylgrid *g, double r, double z, int *pix, int *piy, double c[4])
{
  double X = (orc_cyl_r_to_xi(g, r) - g->xmin) / g->dx;
  double Y = (orc_cyl_z_to_y(g, z) - g->ymin) / g->dy;
  int ix = (int)X;
  int iy = (int)Y;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix >= g->numx) ix = g->numx - 1;
  if (iy >= g->numy) iy = g->numy - 1;
  double delx0 = (double)ix + 1.0 - X;
  double dely0 = (double)iy + 1.0 - Y;
  double delx1 = X - (double)ix;
  double dely1 = Y - (double)iy;
  c[0] = delx0 * dely0; /* c00 */
  c[1] = delx1 * dely0; /* c10 */
  c[2] = delx0 * dely1; /* c01 */
  c[3] = delx1 * dely1; /* c11 */
  *pix = ix;
  *piy = iy;
}

#define INTERP(kind, m, n)                                                        \
  (TAB(kind, m, n, ix, iy) * c[0] + TAB(kind, m, n, ix + 1, iy) * c[1] +          \
   TAB(kind, m, n, ix, iy + 1) * c[2] + TAB(kind, m, n, ix + 1, iy + 1) * c[3])

/* exputil/EmpCylSL.cc:5557-5631: Vc[m][n], Vs[m][n] */
void orc_cyl_get_pot(const orc_cylgrid *g, double r, double z, double *Vc, double *Vs)
{
  if (z / g->ascale > g->rtable) z = g->rtable * g->ascale;
  if (z / g->ascale < -g->rtable) z = -g->rtable * g->ascale;
  int ix, iy;
  double c[4];
  cyl_weights(g, r, z, &ix, &iy, c);
  const double fac = 1.0;
  for (int mm = 0; mm <= orc_opt_mlim(g->mmax); mm++) {             /* :5602: min(MLIM, MMAX) */
    if (g->EVEN_M && (mm / 2) * 2 != mm) continue;
    for (int n = 0; n < g->norder; n++) {
      Vc[mm * g->norder + n] = fac * INTERP(0, mm, n);
      if (mm) Vs[mm * g->norder + n] = fac * INTERP(3, mm, n);
    }
  }
}

/* Cylinder::determine_coefficients_thread (src/Cylinder.cc:748-896, non-eof branch, one
 * thread, one level) calling EmpCylSL::accumulate (exputil/EmpCylSL.cc:4049-4146).
 * cosN[m][n], sinN[m][n] are zeroed here; returns the used count; *cylmass gets the mass
 * of the particles that pass the Rmax2 cut (src/Cylinder.cc:866).                        */
long orc_cyl_accumulate(const orc_cylgrid *g, long nbodies, const double *X, const double *Y,
                        const double *Z, const double *M, const double *center, double *cosN,
                        double *sinN, double *cylmass)
{
  const int nm = (g->mmax + 1) * g->norder;
  double *vc = (double *)calloc(nm, sizeof(double));
  double *vs = (double *)calloc(nm, sizeof(double));
  memset(cosN, 0, sizeof(double) * nm);
  memset(sinN, 0, sizeof(double) * nm);
  const double Rmax2 = g->rcylmax * g->rcylmax * g->acyl * g->acyl;
  long use = 0;
  double mass0 = 0.0;
  const double norm = -4.0 * M_PI;

  const double adb = orc_opt_adb();                                   /* src/Cylinder.cc:834 */
  for (long i = 0; i < nbodies; i++) {
    if (orc_opt_frozen(X[i], Y[i], Z[i])) continue;                   /* :842 */
    double xx = X[i] - center[0];
    double yy = Y[i] - center[1];
    double zz = Z[i] - center[2];
    double r2 = xx * xx + yy * yy;
    double r = sqrt(r2);
    double R2 = r2 + zz * zz;
    if (R2 < Rmax2) {
      double mas = M[i] * adb;
      double phi = atan2(yy, xx);
      /* EmpCylSL::accumulate */
      double rr = sqrt(r * r + zz * zz);
      if (!(rr / g->ascale > g->rtable)) {
        orc_cyl_get_pot(g, r, zz, vc, vs);
        for (int mm = 0; mm <= g->mmax; mm++) {
          double mcos = cos(phi * mm);
          double msin = sin(phi * mm);
          for (int nn = 0; nn < g->norder; nn++) {
            double hold = norm * mas * mcos * vc[mm * g->norder + nn];
            cosN[mm * g->norder + nn] += hold;
            if (mm > 0) {
              hold = norm * mas * msin * vs[mm * g->norder + nn];
              sinN[mm * g->norder + nn] += hold;
            }
          }
        }
      }
      use++;
      mass0 += mas;
    }
  }
  *cylmass = mass0;
  free(vc);
  free(vs);
  return use;
}

/* exputil/EmpCylSL.cc:5256-5410 (MMIN=0, NMIN=0, NLIM=inf; MLIM: orc_set_call_opts) */
void orc_cyl_accumulated_eval(const orc_cylgrid *g, const double *accum_cos,
                              const double *accum_sin, double r, double z, double phi,
                              double *p0, double *p, double *fr, double *fz, double *fp)
{
  *fr = 0.0;
  *fz = 0.0;
  *fp = 0.0;
  *p = 0.0;
  double rr = sqrt(r * r + z * z);
  if (rr / g->ascale > g->rtable) return;

  int ix, iy;
  double c[4];
  cyl_weights(g, r, z, &ix, &iy, c);

  double ccos, ssin = 0.0, fac;
  for (int mm = 0; mm <= orc_opt_mlim(g->mmax); mm++) {             /* :5317 */
    if (g->EVEN_M && (mm / 2) * 2 != mm) continue;
    ccos = cos(phi * mm);
    ssin = sin(phi * mm);
    for (int n = 0; n < g->norder; n++) {
      fac = accum_cos[mm * g->norder + n] * ccos;
      *p += fac * INTERP(0, mm, n);
      *fr += fac * INTERP(1, mm, n);
      *fz += fac * INTERP(2, mm, n);
      fac = accum_cos[mm * g->norder + n] * ssin;
      *fp += fac * mm * INTERP(0, mm, n);
      if (mm) {
        fac = accum_sin[mm * g->norder + n] * ssin;
        *p += fac * INTERP(3, mm, n);
        *fr += fac * INTERP(4, mm, n);
        *fz += fac * INTERP(5, mm, n);
        fac = -accum_sin[mm * g->norder + n] * ccos;
        *fp += fac * mm * INTERP(3, mm, n);
      }
    }
    if (mm == 0) *p0 = *p;
  }
}

/* Cylinder::determine_acceleration_and_potential_thread (src/Cylinder.cc:1266-1446),
 * mix off, no orientation transform: acc += frc, pot += pa.                         */
void orc_cyl_accel(const orc_cylgrid *g, long nbodies, const double *X, const double *Y,
                   const double *Z, const double *center, const double *accum_cos,
                   const double *accum_sin, double cylmass, double *AX, double *AY, double *AZ,
                   double *POT)
{
  const double ratmin = 0.75;
  const double maxerf = 3.0;
  const double midpt = ratmin + 0.5 * (1.0 - ratmin);
  const double rsmth = 0.5 * (1.0 - ratmin) / maxerf;
  double R2 = g->ascale * g->rtable;
  R2 = R2 * R2;
  const double mfactor = 1.0;

  for (long i = 0; i < nbodies; i++) {
    if (orc_opt_frozen(X[i], Y[i], Z[i])) continue;                   /* src/Cylinder.cc:1329 (cC->freeze: the TARGET's) */
    double xx = X[i] - center[0];
    double yy = Y[i] - center[1];
    double zz = Z[i] - center[2];
    double frc[3] = {0.0, 0.0, 0.0};
    double p, p0 = 0.0, fr, fz, fp, pa;

    double r2 = xx * xx + yy * yy;
    double r = sqrt(r2) + DSMALL;
    double phi = atan2(yy, xx);
    pa = 0.0;

    double ratio = sqrt((r2 + zz * zz) / R2);
    double frac, cfrac;
    if (ratio >= 1.0) {
      frac = 0.0;
      cfrac = 1.0;
    } else if (ratio > ratmin) {
      frac = 0.5 * (1.0 - erf((ratio - midpt) / rsmth));
      cfrac = 1.0 - frac;
    } else {
      cfrac = 0.0;
      frac = 1.0;
    }
    cfrac *= mfactor;
    frac *= mfactor;

    if (ratio < 1.0) {
      orc_cyl_accumulated_eval(g, accum_cos, accum_sin, r, zz, phi, &p0, &p, &fr, &fz, &fp);
      frc[0] = (fr * xx / r - fp * yy / r2) * frac;
      frc[1] = (fr * yy / r + fp * xx / r2) * frac;
      frc[2] = fz * frac;
      pa = p * frac;
    }
    if (ratio > ratmin) {
      double r3 = r2 + zz * zz;
      p = -cylmass / sqrt(r3); /* -M/r */
      fr = p / r3;             /* -M/r^3 */
      frc[0] += xx * fr * cfrac;
      frc[1] += yy * fr * cfrac;
      frc[2] += zz * fr * cfrac;
      pa += p * cfrac;
    }
    POT[i] += pa;
    AX[i] += frc[0];
    AY[i] += frc[1];
    AZ[i] += frc[2];
  }
}

/* ---- field evaluation (pyEXP getFields for the cylindrical basis) ------------------------------
 * EmpCylSL::accumulated_dens_eval (exputil/EmpCylSL.cc:5413-5502), MMIN=0, MLIM=NLIM=inf;
 * dens[2][mmax+1][norder][numx+1][numy+1] = densC, densS.                                     */
double orc_cyl_accumulated_dens_eval(const orc_cylgrid *g, const double *dens,
                                     const double *accum_cos, const double *accum_sin, double r,
                                     double z, double phi, double *d0)
{
  double ans = 0.0;
  double rr = sqrt(r * r + z * z);
  *d0 = 0.0;
  if (rr / g->ascale > g->rtable) return ans;

  int ix, iy;
  double c[4];
  cyl_weights(g, r, z, &ix, &iy, c);
  const size_t ny = (size_t)g->numy + 1, slab = ((size_t)g->numx + 1) * ny;
#define DTAB(cs, m, n, i, j) dens[(((size_t)(cs) * (g->mmax + 1) + (m)) * g->norder + (n)) * slab + (size_t)(i) * ny + (j)]
#define DINTERP(cs, m, n)                                                           \
  (DTAB(cs, m, n, ix, iy) * c[0] + DTAB(cs, m, n, ix + 1, iy) * c[1] +              \
   DTAB(cs, m, n, ix, iy + 1) * c[2] + DTAB(cs, m, n, ix + 1, iy + 1) * c[3])
  for (int mm = 0; mm <= orc_opt_mlim(g->mmax); mm++) {             /* :5465 */
    double ccos = cos(phi * mm);
    double ssin = sin(phi * mm);
    for (int n = 0; n < g->norder; n++) {
      double fac = accum_cos[mm * g->norder + n] * ccos;
      ans += fac * DINTERP(0, mm, n);
      if (mm) {
        fac = accum_sin[mm * g->norder + n] * ssin;
        ans += fac * DINTERP(1, mm, n);
      }
    }
    if (mm == 0) *d0 = ans;
  }
#undef DINTERP
#undef DTAB
  return ans;
}

/* Cylindrical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:1749-1849), G = 1, no
 * midplane column.  coord 0: (r, cos theta, phi), 1: (R, z, phi), 2: (x, y, z); out[n][9].     */
void orc_pyexp_cyl_fields(const orc_cylgrid *g, const double *dens, const double *accum_cos,
                          const double *accum_sin, long n, const double *c1, const double *c2,
                          const double *c3, int coord, double *out)
{
  for (long i = 0; i < n; i++) {
    double R, z, phi, x = 0.0, y = 0.0, r = 0.0;
    if (coord == 0) {
      r = c1[i];
      double cth = c2[i], sth = sqrt(1.0 - cth * cth);
      R = r * sth; z = r * cth; phi = c3[i];
    } else if (coord == 1) {
      R = c1[i]; z = c2[i]; phi = c3[i];
    } else {
      x = c1[i]; y = c2[i]; z = c3[i];
      R = sqrt(x * x + y * y);
      phi = atan2(y, x);
    }
    double tdens0, tdens, tpotl0 = 0.0, tpotl, tpotR, tpotz, tpotp;
    orc_cyl_accumulated_eval(g, accum_cos, accum_sin, R, z, phi, &tpotl0, &tpotl, &tpotR, &tpotz,
                             &tpotp);
    tdens = orc_cyl_accumulated_dens_eval(g, dens, accum_cos, accum_sin, R, z, phi, &tdens0);
    double *o = out + 9 * i;
    o[0] = tdens0; o[1] = tdens - tdens0; o[2] = tdens;
    o[3] = tpotl0; o[4] = tpotl - tpotl0; o[5] = tpotl;
    if (coord == 0) {
      o[6] = tpotR * R / r + tpotz * z / R;      /* as written in the reference (:1766-1767) */
      o[7] = tpotR * z / r - tpotz * R / r;
      o[8] = tpotp;
    } else if (coord == 1) {
      o[6] = tpotR; o[7] = tpotz; o[8] = tpotp;
    } else {
      o[6] = tpotR * x / R - tpotp * y / R;
      o[7] = tpotR * y / R + tpotp * x / R;
      o[8] = tpotz;
    }
  }
}

/* EmpCylSL::accumulate, `compute and covar` (exputil/EmpCylSL.cc:4049-4146): vec = (vC cos + vS sin)
 * + i (vC sin - vS cos) with vC = norm Vc(m,:), vS = norm Vs(m,:) (zero for m = 0);
 * VC[whch][m] += mass vec, MV[whch][m] += mass vec vec^dagger.                                    */
long orc_cyl_covariance(const orc_cylgrid *g, long n, const double *X, const double *Y,
                        const double *Z, const double *M, const long *seq, int sampT, long *numbT,
                        double *massT, double *VC, double *MV)
{
  const int nm = (g->mmax + 1) * g->norder, N = g->norder;
  double *vc = (double *)calloc(nm, sizeof(double)), *vs = (double *)calloc(nm, sizeof(double));
  double *re = (double *)malloc(sizeof(double) * N), *im = (double *)malloc(sizeof(double) * N);
  const double norm = -4.0 * M_PI;
  long used = 0;
  for (long i = 0; i < n; i++) {
    double r = sqrt(X[i] * X[i] + Y[i] * Y[i]), z = Z[i], phi = atan2(Y[i], X[i]), mass = M[i];
    double rr = sqrt(r * r + z * z);
    if (rr / g->ascale > g->rtable) continue;
    used++;
    long whch = (seq ? seq[i] : i) % sampT;
    numbT[whch] += 1;
    massT[whch] += mass;
    orc_cyl_get_pot(g, r, z, vc, vs);
    for (int mm = 0; mm <= g->mmax; mm++) {
      double mcos = cos(phi * mm), msin = sin(phi * mm);
      for (int k = 0; k < N; k++) {
        double vC = vc[mm * N + k] * norm, vS = mm ? vs[mm * N + k] * norm : 0.0;
        re[k] = vC * mcos + vS * msin;
        im[k] = vC * msin - vS * mcos;
      }
      double *v = VC + (((size_t)whch * (g->mmax + 1) + mm) * N) * 2;
      double *mv = MV + (((size_t)whch * (g->mmax + 1) + mm) * N * N) * 2;
      for (int k = 0; k < N; k++) {
        v[2 * k] += mass * re[k];
        v[2 * k + 1] += mass * im[k];
        for (int o = 0; o < N; o++) {          /* vec vec^dagger (k, o) = vec_k conj(vec_o) */
          mv[2 * (k * N + o)] += mass * (re[k] * re[o] + im[k] * im[o]);
          mv[2 * (k * N + o) + 1] += mass * (im[k] * re[o] - re[k] * im[o]);
        }
      }
    }
  }
  free(vc); free(vs); free(re); free(im);
  return used;
}

/* Cylindrical::accumulate (expui/BiorthBasis.cc:1851-1857) -> EmpCylSL::accumulate
 * (exputil/EmpCylSL.cc:4049-4146), compute = false */
long orc_pyexp_cyl_accumulate(const orc_cylgrid *g, long n, const double *X, const double *Y,
                              const double *Z, const double *M, double *cosN, double *sinN)
{
  const int nm = (g->mmax + 1) * g->norder;
  double *vc = (double *)calloc(nm, sizeof(double)), *vs = (double *)calloc(nm, sizeof(double));
  const double norm = -4.0 * M_PI;
  long howmany = 0;
  for (long i = 0; i < n; i++) {
    double R = sqrt(X[i] * X[i] + Y[i] * Y[i]);
    double phi = atan2(Y[i], X[i]);
    double z = Z[i], mass = M[i];
    double rr = sqrt(R * R + z * z);
    if (rr / g->ascale > g->rtable) continue;
    howmany++;
    orc_cyl_get_pot(g, R, z, vc, vs);
    for (int mm = 0; mm <= g->mmax; mm++) {
      double mcos = cos(phi * mm);
      double msin = sin(phi * mm);
      for (int nn = 0; nn < g->norder; nn++) {
        double hold = norm * mass * mcos * vc[mm * g->norder + nn];
        cosN[mm * g->norder + nn] += hold;
        if (mm > 0) {
          hold = norm * mass * msin * vs[mm * g->norder + nn];
          sinN[mm * g->norder + nn] += hold;
        }
      }
    }
  }
  free(vc); free(vs);
  return howmany;
}

/* Cylindrical::computeAccel (expui/BiorthBasis.cc:1804-1821) */
void orc_pyexp_cyl_accel(const orc_cylgrid *g, const double *accum_cos, const double *accum_sin,
                         long n, const double *X, const double *Y, const double *Z, double *acc)
{
  for (long i = 0; i < n; i++) {
    double x = X[i], y = Y[i], z = Z[i];
    double R = sqrt(x * x + y * y);
    double phi = atan2(y, x);
    double tpotl0 = 0.0, tpotl, tpotR, tpotz, tpotp;
    orc_cyl_accumulated_eval(g, accum_cos, accum_sin, R, z, phi, &tpotl0, &tpotl, &tpotR, &tpotz, &tpotp);
    acc[3 * i + 0] = tpotR * x / R - tpotp * y / R;
    acc[3 * i + 1] = tpotR * y / R + tpotp * x / R;
    acc[3 * i + 2] = tpotz;
  }
}

/* ---- conditioning the basis on the particles: the covariance of the helper functions ------------------------------------
 * EmpCylSL::legendre_R (exputil/EmpCylSL.cc:6493-6569): Holmes & Featherstone forward column recursion, normalised,
 * Condon-Shortley phase; p[l * (lmax + 1) + m]. */
void orc_emp_legendre_R(int lmax, double x, double *p)
{
  const int s = lmax + 1;
  for (int k = 0; k < s * s; k++) p[k] = 0.0;
  double u = sqrt(1 - x * x), pll;
  p[0] = 1.0;
  if (lmax >= 1) p[1 * s + 1] = pll = -sqrt(3) * u;
  if (lmax > 0) {
    for (int m = 2; m <= lmax; m++) {
      pll *= -u * sqrt((2. * m + 1) / (2. * m));
      p[m * s + m] = pll;
    }
  }
  for (int l = 1; l <= lmax; l++) {
    for (int m = 0; m < l; m++) {
      int l1 = l - 1 > 0 ? l - 1 : 0;
      int l2 = l - 2 > 0 ? l - 2 : 0;
      p[l * s + m] = sqrt((2. * l - 1) * (2 * l + 1) / (l - m) / (l + m)) * x * p[l1 * s + m]
                   - sqrt((2. * l + 1) * (l + m - 1) * (l - m - 1) / (l - m) / (l + m) / (2 * l - 3)) * p[l2 * s + m];
    }
  }
  for (int l = 0; l <= lmax; l++)
    for (int m = 0; m <= l; m++) {
      double norm = m > 0 ? sqrt(8. * M_PI) : sqrt(4. * M_PI);
      p[l * s + m] = p[l * s + m] / norm;
    }
}

/* EmpCylSL::sinecosine_R (exputil/EmpCylSL.cc:6624-6639) */
static void emp_sinecosine_R(int mmax, double phi, double *c, double *s)
{
  c[0] = 1.0; s[0] = 0.0;
  if (mmax >= 1) { c[1] = cos(phi); s[1] = sin(phi); }
  for (int m = 2; m <= mmax; m++) {
    c[m] = 2.0 * c[1] * c[m - 1] - c[m - 2];
    s[m] = 2.0 * c[1] * s[m - 1] - s[m - 2];
  }
}

/* EmpCylSL::accumulate_eof (exputil/EmpCylSL.cc:2686-2862, the branch without EvenOdd) of n particles for ONE harmonic
 * M under its caller's cut (Cylinder::determine_coefficients_thread, src/Cylinder.cc:806-820: x^2 + y^2 + z^2 <
 * Rmax2): SC, SS [rank][rank], rank = NMAX * (LMAX - M + 1), nn = ir + NMAX * (l - M).  sl: the helper SLGridSph
 * (LMAX, NMAX).  Returns the particles used (`use`), *cylmass their mass. */
long orc_cyl_accumulate_eof(const orc_slgrid *sl, int M, double ascale, double rtable, double rmax2, long n,
                            const double *X, const double *Y, const double *Z, const double *mass,
                            double *SC, double *SS, double *cylmass)
{
  const int LMAX = sl->lmax, NMAX = sl->nmax;
  const int nl = LMAX - M + 1, rank = NMAX * nl;
  const double pfac = 1.0 / sqrt(ascale);                /* exputil/EmpCylSL.cc:173 */
  double *table = (double *)malloc(sizeof(double) * (LMAX + 1) * NMAX);
  double *legs = (double *)malloc(sizeof(double) * (LMAX + 1) * (LMAX + 1));
  double *cosm = (double *)malloc(sizeof(double) * (LMAX + 1));
  double *sinm = (double *)malloc(sizeof(double) * (LMAX + 1));
  double *facC = (double *)malloc(sizeof(double) * NMAX * nl);
  double *facS = (double *)malloc(sizeof(double) * NMAX * nl);
  for (long k = 0; k < (long)rank * rank; k++) SC[k] = SS[k] = 0.0;
  long use = 0;
  *cylmass = 0.0;
  for (long i = 0; i < n; i++) {
    double xx = X[i], yy = Y[i], zz = Z[i];
    double r2 = xx * xx + yy * yy;
    double r = sqrt(r2);
    double R2 = r2 + zz * zz;
    if (!(R2 < rmax2)) continue;
    double mas = mass[i];
    double phi = atan2(yy, xx);
    use++;
    *cylmass += mas;
    /* accumulate_eof(r, zz, phi, mas, id, level) */
    double rr = sqrt(r * r + zz * zz);
    if (rr / ascale > rtable) continue;
    orc_sl_get_pot(sl, rr / ascale, table);
    double costh = zz / (rr + 1.0e-18);
    orc_emp_legendre_R(LMAX, costh, legs);
    emp_sinecosine_R(LMAX, phi, cosm, sinm);
    for (int ir = 0; ir < NMAX; ir++) {
      for (int l = M; l <= LMAX; l++) {
        double ylm = pfac * legs[l * (LMAX + 1) + M];
        if (M == 0) {
          facC[ir * nl + (l - M)] = ylm * table[l * NMAX + ir];
        } else {
          facC[ir * nl + (l - M)] = ylm * table[l * NMAX + ir] * cosm[M];
          facS[ir * nl + (l - M)] = ylm * table[l * NMAX + ir] * sinm[M];
        }
      }
    }
    for (int ir1 = 0; ir1 < NMAX; ir1++) {
      for (int l1 = M; l1 <= LMAX; l1++) {
        int nn1 = ir1 + NMAX * (l1 - M);
        for (int ir2 = 0; ir2 < NMAX; ir2++) {
          for (int l2 = M; l2 <= LMAX; l2++) {
            int nn2 = ir2 + NMAX * (l2 - M);
            SC[(long)nn1 * rank + nn2] += facC[ir1 * nl + (l1 - M)] * facC[ir2 * nl + (l2 - M)] * mas;
            if (M > 0)
              SS[(long)nn1 * rank + nn2] += facS[ir1 * nl + (l1 - M)] * facS[ir2 * nl + (l2 - M)] * mas;
          }
        }
      }
    }
  }
  free(table); free(legs); free(cosm); free(sinm); free(facC); free(facS);
  return use;
}
