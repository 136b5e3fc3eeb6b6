/* CPU oracle for the NOISE mode of SphericalBasis -- TEST INFRASTRUCTURE ONLY (see bfe_oracle.h).
 *
 * compute_rms_coefs (src/SphericalBasis.cc:2108-2147) and update_noise (:2150-2210) restated in the reference's operation
 * order.  This one translation unit is C++ because the reference draws its deviates from `std::mt19937 rgen` and
 * `std::normal_distribution<> nrand` (src/SphericalBasis.H:340-341): the objects here are the same standard-library ones,
 * so the sequence is the reference's on any host with the same libstdc++.  The model table is read by the caller and
 * interpolated as SphericalModelTable::get_density does with the class defaults linear = 1, even = 0, no divergence
 * (exputil/massmodel.cc:19-20, :266-291 -> odd2, exputil/Vodd2.cc:49-70 -> Vlocate, exputil/Vlocate.cc:50-66).           */
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
#include "bfe_oracle.h"
}

static int vlocate(double x, const double *xx, int n)          /* exputil/Vlocate.cc:50-66 */
{
  int min = 0, max = n - 1, jl = min - 1, ju = max + 1;
  int ascnd = xx[max] > xx[min];
  while (ju - jl > 1) {
    int jm = (ju + jl) >> 1;
    if ((x > xx[jm]) == ascnd) jl = jm;
    else ju = jm;
  }
  return jl;
}

static double odd2(double x, const double *xtab, const double *ftab, int n)      /* exputil/Vodd2.cc:49-70, even = 0 */
{
  int min = 0, max = n - 1;
  int index = vlocate(x, xtab, n);
  if (index < min) index = min;
  if (index >= max) index = max - 1;
  return (ftab[index + 1] * (x - xtab[index]) - ftab[index] * (x - xtab[index + 1])) / (xtab[index + 1] - xtab[index]);
}

static double get_density(double r, const double *rt, const double *dt, int n)   /* exputil/massmodel.cc:266-291 */
{
  if (r > rt[n - 1]) return dt[n - 1];
  return odd2(r, rt, dt, n);
}

static double factrl(int n) { double a = 1.0; for (int i = 2; i <= n; i++) a *= (double)i; return a; }

struct orc_noise {
  int lmax, nmax;
  std::vector<double> meanC, rmsC;
  double noiseN;
  unsigned seedN;
  bool setup_noise;
  std::mt19937 rgen;
  std::normal_distribution<> nrand;
};

extern "C" {

/* src/SphericalBasis.cc:2108-2147; knot / weight: LegeQuad(numg) on [0, 1]; sqnorm = 1 (Sphere) */
void orc_sph_compute_rms_coefs(const orc_slgrid *g, double scale, int num, const double *rtab, const double *dtab,
                               int numg, const double *knot, const double *weight, double *meanC, double *rmsC)
{
  const int Lmax = g->lmax, nmax = g->nmax;
  std::vector<double> potd((size_t)(Lmax + 1) * nmax);
  for (int n = 0; n < nmax; n++) meanC[n] = 0.0;
  for (int k = 0; k < (Lmax + 1) * nmax; k++) rmsC[k] = 0.0;
  double rmin = rtab[0], rmax = rtab[num - 1];
  double del = rmax - rmin;
  for (int i = 0; i < numg; i++) {
    double r = rmin + del * knot[i];
    double rs = r / scale;
    orc_sl_get_pot(g, rs, potd.data());
    for (int l = 0; l <= Lmax; l++) {
      for (int n = 0; n < nmax; n++) {
        double pot = potd[(size_t)l * nmax + n] / 1.0 / scale;
        if (l == 0)
          meanC[n] += del * weight[i] * r * r * pot * 4.0 * M_PI * get_density(r, rtab, dtab, num);
        rmsC[(size_t)l * nmax + n] += del * weight[i] * r * r * pot * pot * 4.0 * M_PI * get_density(r, rtab, dtab, num);
      }
    }
  }
}

void *orc_noise_create(int lmax, int nmax, const double *meanC, const double *rmsC, double noiseN, unsigned seedN)
{
  orc_noise *h = new orc_noise;
  h->lmax = lmax; h->nmax = nmax;
  h->meanC.assign(meanC, meanC + nmax);
  h->rmsC.assign(rmsC, rmsC + (size_t)(lmax + 1) * nmax);
  h->noiseN = noiseN; h->seedN = seedN; h->setup_noise = true;
  return h;
}

void orc_noise_destroy(void *p) { delete (orc_noise *)p; }

/* update_noise (src/SphericalBasis.cc:2150-2210): expcoef[(lmax+1)^2][nmax], the reference's real-row order */
void orc_noise_update(void *p, double *expcoef)
{
  orc_noise *h = (orc_noise *)p;
  const int Lmax = h->lmax, nmax = h->nmax;
  if (h->setup_noise) {
    h->setup_noise = false;
    h->rgen.seed(h->seedN);
  }
#define RMS(l, n) h->rmsC[(size_t)(l) * nmax + (n)]
  for (int l = 0, loffset = 0; l <= Lmax; loffset += (2 * l + 1), l++) {
    for (int m = 0, moffset = 0; m <= l; m++) {
      double factorial = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * factrl(l - m) / factrl(l + m));
      if (m) factorial *= M_SQRT2;
      if (m == 0) {
        for (int n = 0; n < nmax; n++) {
          expcoef[(size_t)(loffset + moffset) * nmax + n] =
              sqrt(fabs(RMS(l, n) - h->meanC[n] * h->meanC[n]) * factorial / h->noiseN) * h->nrand(h->rgen);
          if (l == 0) expcoef[(size_t)l * nmax + n] += h->meanC[n];
        }
        moffset++;
      } else {
        for (int n = 0; n < nmax; n++) {
          expcoef[(size_t)(loffset + moffset + 0) * nmax + n] =
              sqrt(fabs(RMS(l, n) - h->meanC[n] * h->meanC[n]) * factorial / h->noiseN) * h->nrand(h->rgen);
          expcoef[(size_t)(loffset + moffset + 1) * nmax + n] =
              sqrt(fabs(RMS(l, n) - h->meanC[n] * h->meanC[n]) * factorial / h->noiseN) * h->nrand(h->rgen);
        }
        moffset += 2;
      }
    }
  }
#undef RMS
}

}   /* extern "C" */
