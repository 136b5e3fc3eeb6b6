// Coefficient covariance by sub-sampling for the spherical basis: pyEXP's Spherical::accumulate with
// pcavar (expui/BiorthBasis.cc:583-665, init :342-378, accessors BiorthBasis.H:425-470).  Per
// accepted particle (window rmin <= r <= rmax) the reference bumps `used`, files the particle under
// sub-sample T = used % sampT and adds, for every (l, m >= 0),
//     g = exp(i m phi) potd.row(l) * factorial(l,m) P_lm * norm,   meanV[T][lm] += g m,
//     covrV[T][lm] += g g^dagger m                      (nmax and nmax x nmax per (T, lm)).
// potd(l,n) = P0(r) (x1 E_i[l][n] + x2 E_{i+1}[l][n]) is linear in the two table columns of the
// particle's radial cell i, so -- as in the coefficient pass -- the n-dependence is hoisted out of
// the particle loop: per (T, cell, lm) seven moments
//     sum m A c x1, m A s x1, m A c x2, m A s x2          (A = fac P_lm norm P0, c/s = cos/sin m phi)
//     sum m A^2 (c^2+s^2) {x1^2, x1 x2, x2^2}
// are accumulated (per-particle atomics: an analysis pass over <= 1e7 particles, not the n-body
// loop), and one contraction with E per (T, lm) turns them into mean (complex) and covariance
// (real: the phase cancels in g g^dagger).  The sub-sample index needs the running count of
// accepted particles in the CALLER's order: flags by caller index, exclusive scan, rank.
#include "sph_force.h"
#include "sort_kernels.h"
#include <cmath>
#include <vector>

namespace {

constexpr int COV_NMOM = 7;

struct SphCov {
  int sampT = 0;
  size_t nflag = 0;
  DevBuf<double> mom;                     // [sampT][numr-1][ltot][7]
  DevBuf<double> masses, fac, mean, covr; // [sampT]; [(L+1)^2]; [sampT][ltot][nmax][2]; [sampT][ltot][nmax][nmax]
  DevBuf<unsigned long long> counts;      // [sampT]
  DevBuf<uint32_t> flags, tot;            // [n+1] accepted flags by caller index -> exclusive ranks; [2]
};

// accepted (rmin <= r <= rmax with r = |x| + 1e-20, :596-603) by CALLER index
__global__ void __launch_bounds__(256)
k_cov_flags(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
            const double *__restrict__ Z, const uint32_t *__restrict__ id, size_t n,
            uint32_t *__restrict__ flags)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double xx = X[i] - S.cx, yy = Y[i] - S.cy, zz = Z[i] - S.cz;
  const double r = sqrt(xx * xx + yy * yy + zz * zz) + 1.0e-20;
  flags[id[i]] = (r < S.rmin || r > S.rmax) ? 0u : 1u;
}

__global__ void __launch_bounds__(256)
k_cov_accumulate(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                 const double *__restrict__ Z, const double *__restrict__ M,
                 const uint32_t *__restrict__ id, size_t n, const uint32_t *__restrict__ rank,
                 const double *__restrict__ facT, int sampT, unsigned long long used0,
                 double *__restrict__ mom, unsigned long long *__restrict__ counts,
                 double *__restrict__ masses)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double xx = X[i] - S.cx, yy = Y[i] - S.cy, zz = Z[i] - S.cz;
  const double r = sqrt(xx * xx + yy * yy + zz * zz) + 1.0e-20;
  if (r < S.rmin || r > S.rmax) return;
  const double mass = M[i];
  const unsigned long long used = used0 + (unsigned long long)rank[id[i]] + 1ull;   // used++ comes first (:603)
  const int T = (int)(used % (unsigned long long)sampT);
  atomicAdd(&counts[T], 1ull);
  unsafeAtomicAdd(&masses[T], mass);
  const double costh = zz / r, phi = atan2(yy, xx);
  // get_pot (exputil/SLGridMP2.cc:872-910): cell and weights
  const double xi = sph_r_to_xi(S, r / S.scale);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) / S.dxi, x2 = (xi - S.xi[idx]) / S.dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const double norm = -4.0 * M_PI;
  const int L = S.lmax, ltot = (L + 1) * (L + 2) / 2;
  double *base = mom + (((size_t)T * (S.numr - 1) + idx) * ltot) * COV_NMOM;
  // legendre_R (src/Basis.cc:14-52), column by column: p(m,m), p(m+1,m), then upward in l
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  double pmm = 1.0;
  for (int m = 0; m <= L; m++) {
    if (m > 0) pmm *= -(2.0 * m - 1.0) * somx2;
    double sn, cs;
    sincos((double)m * phi, &sn, &cs);
    double pl2 = 0.0, pl1 = pmm;
    for (int l = m; l <= L; l++) {
      double p;
      if (l == m) p = pmm;
      else p = (costh * (2.0 * l - 1.0) * pl1 - (double)(l + m - 1) * pl2) / (double)(l - m);
      if (l > m) { pl2 = pl1; pl1 = p; }
      const double A = facT[l * (L + 1) + m] * p * norm * P0;
      const double mA = mass * A, mA2 = mass * A * A * (cs * cs + sn * sn);
      double *w = base + (size_t)(l * (l + 1) / 2 + m) * COV_NMOM;
      unsafeAtomicAdd(w + 0, mA * cs * x1);
      unsafeAtomicAdd(w + 1, mA * sn * x1);
      unsafeAtomicAdd(w + 2, mA * cs * x2);
      unsafeAtomicAdd(w + 3, mA * sn * x2);
      unsafeAtomicAdd(w + 4, mA2 * x1 * x1);
      unsafeAtomicAdd(w + 5, mA2 * x1 * x2);
      unsafeAtomicAdd(w + 6, mA2 * x2 * x2);
    }
  }
}

// one block per (lm, T): mean[n] and covr[n][n2] from the cell moments and the table columns
__global__ void __launch_bounds__(256)
k_cov_contract(SphDev S, const double *__restrict__ mom, double *__restrict__ mean,
               double *__restrict__ covr)
{
  const int L = S.lmax, ltot = (L + 1) * (L + 2) / 2, nmax = S.nmax;
  const int lm = blockIdx.x, T = blockIdx.y;
  int l = 0;
  while ((l + 1) * (l + 2) / 2 <= lm) l++;
  const int ncell = S.numr - 1, stride = (L + 1) * nmax;
  const int npair = nmax * nmax;
  // blockIdx.z: a stretch of 4096 (n, n2) pairs, sixteen per thread (any nmax: the reference takes what the YAML gives,
  // expui/BiorthBasis.cc:342-378); the 2 nmax mean entries ride with the first stretch, up to four per thread
  const int p0 = blockIdx.z * 4096, p1 = min(npair, p0 + 4096);
  const bool do_mean = blockIdx.z == 0;
  double cacc[16], macc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < 16; k++) cacc[k] = 0.0;
  const double *w0 = mom + (((size_t)T * ncell) * ltot + lm) * COV_NMOM;
  for (int i = 0; i < ncell; i++) {
    const double *w = w0 + (size_t)i * ltot * COV_NMOM;
    const double a = w[4], b = w[5], c = w[6];
    if (a == 0.0 && b == 0.0 && c == 0.0 && w[0] == 0.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 0.0) continue;
    const double *e0 = S.E + (size_t)i * stride + l * nmax, *e1 = e0 + stride;
    if (do_mean)
      for (int j = 0, t = threadIdx.x; t < 2 * nmax && j < 4; t += 256, j++) {
        const int n = t >> 1, ri = t & 1;                           // re / im
        macc[j] += w[ri] * e0[n] + w[2 + ri] * e1[n];
      }
    for (int k = 0, p = p0 + threadIdx.x; p < p1; p += 256, k++) {
      const int n = p / nmax, n2 = p - n * nmax;
      cacc[k] += a * e0[n] * e0[n2] + b * (e0[n] * e1[n2] + e1[n] * e0[n2]) + c * e1[n] * e1[n2];
    }
  }
  if (do_mean)
    for (int j = 0, t = threadIdx.x; t < 2 * nmax && j < 4; t += 256, j++)
      mean[((size_t)T * ltot + lm) * nmax * 2 + t] = macc[j];
  for (int k = 0, p = p0 + threadIdx.x; p < p1; p += 256, k++)
    covr[((size_t)T * ltot + lm) * npair + p] = cacc[k];
}

}  // namespace

// owned by the force through an opaque pointer (kept out of sph_force.h: analysis-only state)
static SphCov *cov_of(SphForce *f) { return (SphCov *)f->cov; }

void expamd_sph_cov_release(SphForce *f)
{
  SphCov *c = cov_of(f);
  if (!c) return;
  c->mom.release(); c->masses.release(); c->fac.release(); c->mean.release(); c->covr.release();
  c->counts.release(); c->flags.release(); c->tot.release();
  delete c;
  f->cov = nullptr;
}

// enableCoefCovariance(pcavar, sampT) + init_covariance / zero_covariance (expui/BiorthBasis.cc:342-378)
extern "C" int exp_amd_sph_cov_enable(exp_amd_force *fb, int sampT)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cov_enable: not a sphereSL force");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  expamd_sph_cov_release(f);
  if (sampT <= 0) return EXP_AMD_OK;
  if (f->cfg.nmax > 512) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cov_enable: nmax > 512 not supported");
  SphCov *c = new SphCov;
  f->cov = c;
  c->sampT = sampT;
  const int L = f->cfg.lmax, ltot = (L + 1) * (L + 2) / 2, nmax = f->cfg.nmax;
  const size_t nm = (size_t)sampT * (f->cfg.numr - 1) * ltot * COV_NMOM;
  if (c->mom.alloc(nm) != hipSuccess || c->masses.alloc(sampT) != hipSuccess ||
      c->counts.alloc(sampT) != hipSuccess || c->fac.alloc((size_t)(L + 1) * (L + 1)) != hipSuccess ||
      c->mean.alloc((size_t)sampT * ltot * nmax * 2) != hipSuccess ||
      c->covr.alloc((size_t)sampT * ltot * nmax * nmax) != hipSuccess || c->tot.alloc(4) != hipSuccess) {
    expamd_sph_cov_release(f);
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cov_enable: hipMalloc failed (%zu moment doubles)", nm);
  }
  // factorial(l,m) of expui/BiorthBasis.cc:323-329
  std::vector<double> fac((size_t)(L + 1) * (L + 1), 0.0);
  for (int l = 0; l <= L; l++)
    for (int m = 0; m <= l; m++) {
      double v = sqrt((0.5 * l + 0.25) / M_PI * exp(lgamma(1.0 + l - m) - lgamma(1.0 + l + m)));
      if (m != 0) v *= M_SQRT2;
      fac[(size_t)l * (L + 1) + m] = v;
    }
  HIP_TRY(ctx, hipMemcpy(c->fac.p, fac.data(), fac.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemset(c->mom.p, 0, c->mom.bytes()));
  HIP_TRY(ctx, hipMemset(c->masses.p, 0, c->masses.bytes()));
  HIP_TRY(ctx, hipMemset(c->counts.p, 0, c->counts.bytes()));
  return EXP_AMD_OK;
}

// zero_covariance (reset_coefs, expui/BiorthBasis.cc:478)
extern "C" int exp_amd_sph_cov_reset(exp_amd_force *fb)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !cov_of(f)) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cov_reset: covariance not enabled");
  SphCov *c = cov_of(f);
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipMemsetAsync(c->mom.p, 0, c->mom.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c->masses.p, 0, c->masses.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c->counts.p, 0, c->counts.bytes(), ctx->stream));
  return EXP_AMD_OK;
}

// the pcavar block of Spherical::accumulate for every particle of `comp` in caller order;
// used_before = accepted particles of earlier calls; *accepted = those of this one
extern "C" int exp_amd_sph_cov_accumulate(exp_amd_force *fb, exp_amd_comp *comp, long long used_before,
                                          long long *accepted)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !comp || !cov_of(f))
    return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cov_accumulate: covariance not enabled");
  { int rc_ = expamd_comp_densify(comp); if (rc_) return rc_; }      // (an appended store: made an ordinary one first)
  SphCov *c = cov_of(f);
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (positions and masses only: whatever half-kick the velocities are owed or ahead by does not matter here)
  const size_t n = comp->n;
  if (accepted) *accepted = 0;
  if (n == 0) return EXP_AMD_OK;
  if (c->nflag < n + 1) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (c->flags.alloc(n + 1) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cov_accumulate: hipMalloc failed");
    c->nflag = n + 1;
  }
  SphDev S = f->dev;
  S.cx = comp->center[0]; S.cy = comp->center[1]; S.cz = comp->center[2];
  const uint32_t *id = comp->id[comp->cur].p;
  ProfScope ps(ctx, "k_sph_covariance");
  k_cov_flags<<<cdiv(n, 256), 256, 0, ctx->stream>>>(S, comp->a(A_X), comp->a(A_Y), comp->a(A_Z), id, n,
                                                    c->flags.p);
  // exclusive scan in place; one "level" of n+1 bins so that only tot[0] / tot[1] are written
  { int rc_ = expamd_launch_scan_full(ctx, ctx->stream, c->flags.p, (uint32_t)n, c->tot.p, (uint32_t)n + 1u, 1); if (rc_) return rc_; }
  k_cov_accumulate<<<cdiv(n, 256), 256, 0, ctx->stream>>>(
      S, comp->a(A_X), comp->a(A_Y), comp->a(A_Z), comp->a(A_M), id, n, c->flags.p, c->fac.p, c->sampT,
      (unsigned long long)used_before, c->mom.p, c->counts.p, c->masses.p);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t tot[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(tot, c->tot.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (accepted) *accepted = (long long)tot[1];
  return EXP_AMD_OK;
}

// getCovarSamples / getCoefCovariance (expui/BiorthBasis.H:425-470): counts[sampT], masses[sampT],
// mean[sampT][ltot][nmax][2] (re, im), covr[sampT][ltot][nmax][nmax] (real; the imaginary part of
// g g^dagger is identically zero).  Any pointer may be NULL.
extern "C" int exp_amd_sph_cov_get(exp_amd_force *fb, long long *counts, double *masses, double *mean,
                                   double *covr)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !cov_of(f)) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cov_get: covariance not enabled");
  SphCov *c = cov_of(f);
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int L = f->cfg.lmax, ltot = (L + 1) * (L + 2) / 2;
  k_cov_contract<<<dim3(ltot, c->sampT, cdiv((size_t)f->cfg.nmax * f->cfg.nmax, 4096)), 256, 0, ctx->stream>>>(
      f->dev, c->mom.p, c->mean.p, c->covr.p);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<unsigned long long> cnt(c->sampT);
  if (counts) HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), c->counts.p, c->counts.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (masses) HIP_TRY(ctx, hipMemcpyAsync(masses, c->masses.p, c->masses.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (mean) HIP_TRY(ctx, hipMemcpyAsync(mean, c->mean.p, c->mean.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (covr) HIP_TRY(ctx, hipMemcpyAsync(covr, c->covr.p, c->covr.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (counts) for (int t = 0; t < c->sampT; t++) counts[t] = (long long)cnt[t];
  return EXP_AMD_OK;
}
