// Cylindrical basis, analysis side (pyEXP): field evaluation at points, sub-sample covariance of the coefficients, the
// basis functions on an (R, z) grid and their orthogonality check.
#include "cyl_force.h"

// ---- field evaluation at points (pyEXP getFields for the cylindrical basis) -------------------------
// Cylindrical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:1749-1849) = accumulated_eval
// (exputil/EmpCylSL.cc:5256-5410) + accumulated_dens_eval (:5413-5502) at arbitrary points.  Not a
// throughput path: one lane per point, the (m, n) sums taken directly on the tables.
__global__ void __launch_bounds__(256)
k_cyl_fields(CylDev C, const double *__restrict__ tab, const double *__restrict__ dens,
             const double *__restrict__ coef, size_t n, const double *__restrict__ c1,
             const double *__restrict__ c2, const double *__restrict__ c3, int coord,
             double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double R, z, phi, x = 0.0, y = 0.0, r = 0.0;
  if (coord == 0) {
    r = c1[i];
    const double cth = c2[i], sth = sqrt(-sq_add_lit(-1.0, cth));     // 1 - cth*cth, the product rounded first (expui/BiorthBasis.cc:1753)
    R = r * sth; z = r * cth; phi = c3[i];
  } else if (coord == 1) {
    R = c1[i]; z = c2[i]; phi = c3[i];
  } else {
    x = c1[i]; y = c2[i]; z = c3[i];
    R = sqrt(sq_sum2_lit(x, y));                                       // (:1780; no fused multiply-add: sq_sum2_lit, common.h)
    phi = atan2(y, x);
  }
  double p0 = 0.0, p = 0.0, fr = 0.0, fz = 0.0, fp = 0.0, d0 = 0.0, d = 0.0;
  if (!(sqrt(R * R + z * z) > C.rtab_abs)) {
    int ix, iy;
    double c00, c10, c01, c11;
    cyl_weights(C, R, z, ix, iy, c00, c10, c01, c11);
    const size_t ny = (size_t)C.numy + 1, nnode = (size_t)(C.numx + 1) * ny;
    const size_t n00 = (size_t)ix * ny + iy;
    const size_t half = (size_t)(C.mmax + 1) * C.nmax;
    auto bl = [&](const double *T) {
      return T[n00] * c00 + T[n00 + ny] * c10 + T[n00 + 1] * c01 + T[n00 + ny + 1] * c11;
    };
    for (int mm = 0; mm <= C.mmax; mm++) {
      const double ccos = cos(phi * mm), ssin = sin(phi * mm);
      const bool on = !(C.EVEN_M && (mm & 1));                 // accumulated_eval only (:5318)
      for (int k = 0; k < C.nmax; k++) {
        const size_t mk = (size_t)mm * C.nmax + k;
        const double ac = coef[mk], as = coef[half + mk];
        const double *Tc = tab + mk * nnode;                    // kind 0 (potC); kinds are +half*nnode apart
        const size_t ks = half * nnode;
        if (on) {
          const double vp = bl(Tc), vr = bl(Tc + ks), vz = bl(Tc + 2 * ks);
          p += ac * ccos * vp; fr += ac * ccos * vr; fz += ac * ccos * vz;
          fp += ac * ssin * mm * vp;
          if (mm) {
            const double wp = bl(Tc + 3 * ks), wr = bl(Tc + 4 * ks), wz = bl(Tc + 5 * ks);
            p += as * ssin * wp; fr += as * ssin * wr; fz += as * ssin * wz;
            fp += -as * ccos * mm * wp;
          }
        }
        d += ac * ccos * bl(dens + mk * nnode);
        if (mm) d += as * ssin * bl(dens + (half + mk) * nnode);
      }
      if (mm == 0) { p0 = p; d0 = d; }
    }
  }
  double *o = out + 9 * i;
  o[0] = d0; o[1] = d - d0; o[2] = d;
  o[3] = p0; o[4] = p - p0; o[5] = p;
  if (coord == 0) { o[6] = fr * R / r + fz * z / R; o[7] = fr * z / r - fz * R / r; o[8] = fp; }
  else if (coord == 1) { o[6] = fr; o[7] = fz; o[8] = fp; }
  else { o[6] = fr * x / R - fp * y / R; o[7] = fr * y / R + fp * x / R; o[8] = fz; }
}

extern "C" int exp_amd_cyl_set_density(exp_amd_force *fb, const double *dens)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f || !dens) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cyl_set_density: not a cylinder force / NULL");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t cnt = (size_t)2 * (f->cfg.mmax + 1) * f->cfg.nmax * f->nnode;
  if (f->d_dens.alloc(cnt) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_set_density: hipMalloc failed");
  HIP_TRY(ctx, hipMemcpyAsync(f->d_dens.p, dens, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (f->mlim >= 0 && f->mlim < f->cfg.mmax) {         // accumulated_dens_eval sums up to MLIM (exputil/EmpCylSL.cc:5465)
    const size_t per_m = (size_t)f->cfg.nmax * f->nnode, per_kind = (size_t)(f->cfg.mmax + 1) * per_m;
    for (int k = 0; k < 2; k++)
      HIP_TRY(ctx, hipMemsetAsync(f->d_dens.p + k * per_kind + (size_t)(f->mlim + 1) * per_m, 0,
                                  (size_t)(f->cfg.mmax - f->mlim) * per_m * sizeof(double), ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_cyl_fields(exp_amd_force *fb, size_t n, const double *c1, const double *c2,
                                  const double *c3, int coord, double *out)
{
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cyl_fields: not a cylinder force");
  exp_amd_ctx *ctx = f->ctx;
  if (n == 0) return EXP_AMD_OK;
  if (!c1 || !c2 || !c3 || !out || coord < 0 || coord > 2)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cyl_fields: bad argument");
  if (!f->d_dens.p)
    return expamd_fail(ctx, EXP_AMD_ERR_STATE, "cyl_fields: call exp_amd_cyl_set_density first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_in, d_out;
  if (d_in.alloc(3 * n) != hipSuccess || d_out.alloc(9 * n) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_fields: hipMalloc failed");
  const double *src[3] = {c1, c2, c3};
  for (int k = 0; k < 3; k++)
    HIP_TRY(ctx, hipMemcpyAsync(d_in.p + (size_t)k * n, src[k], n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const CylDev C = f->dev;                   // fields are evaluated about the origin, as pyEXP does
  k_cyl_fields<<<cdiv(n, 256), 256, 0, ctx->stream>>>(C, f->d_tab.p, f->d_dens.p, f->d_coef.p, n, d_in.p,
                                                      d_in.p + n, d_in.p + 2 * n, coord, d_out.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, 9 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_in.release();
  d_out.release();
  return EXP_AMD_OK;
}

// ---- sub-sample covariance of the coefficients (pyEXP) ------------------------------------------------------
// The `covar` branch of EmpCylSL::accumulate (exputil/EmpCylSL.cc:4049-4146) behind
// Cylindrical::accumulate (expui/BiorthBasis.cc:1851-1857): per particle on the grid, sub-sample
// whch = seq % sampT, vec = norm [(Vc cos + Vs sin) + i (Vc sin - Vs cos)] (m = 0: Vs = 0),
// VC[whch][m] += mass vec, MV[whch][m] += mass vec vec^dagger.  Vc, Vs are bilinear in the four
// node values of the particle's cell with weights c_k, so with u_k = c_k cos, w_k = c_k sin
//   sum mass vec            = norm sum_node [U TC + W TS] + i norm sum_node [W TC - U TS]
//   sum mass vec vec^dagger = norm^2 sum_cell sum_kk' Q_kk' [TC_k TC_k' + TS_k TS_k' + i (TC_k TS_k' - TS_k TC_k')]
// (the azimuthal phase cancels: u_k u_k' + w_k w_k' = c_k c_k', u_k w_k' - w_k u_k' = 0), i.e. per
// sub-sample the node moments U, W of the coefficient pass plus TEN cell moments Q_kk' = sum mass
// c_k c_k' that do not even depend on m; two contractions with the tables finish the job.
__global__ void __launch_bounds__(256)
k_cyl_cov_accumulate(CylDev C, const double *__restrict__ X, const double *__restrict__ Y,
                     const double *__restrict__ Z, const double *__restrict__ M,
                     const uint32_t *__restrict__ id, const uint32_t *__restrict__ seq, size_t n,
                     int sampT, double *__restrict__ U, double *__restrict__ Q,
                     unsigned long long *__restrict__ cnt, double *__restrict__ msum,
                     unsigned long long *__restrict__ used)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double xx, yy, zz;
  cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
  const double r2 = xx * xx + yy * yy, r = sqrt(r2);
  if (sqrt(r * r + zz * zz) > C.rtab_abs) return;                     // EmpCylSL.cc:4062-4063
  const double mass = M[i];
  const uint32_t sq = seq ? seq[id[i]] : id[i];
  const int T = (int)(sq % (uint32_t)sampT);
  atomicAdd(&cnt[T], 1ull);
  atomicAdd(used, 1ull);
  unsafeAtomicAdd(&msum[T], mass);
  double zc = zz;                                                     // get_pot z clamp (:5563-5564)
  if (zc > C.rtab_abs) zc = C.rtab_abs;
  if (zc < -C.rtab_abs) zc = -C.rtab_abs;
  int ix, iy;
  double cw[4];
  cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
  const double phi = atan2(yy, xx);
  const int nyp = C.numy + 1, NT = C.ntrig;
  const size_t nnode = (size_t)(C.numx + 1) * nyp;
  double *u0 = U + ((size_t)T * nnode + (size_t)ix * nyp + iy) * NT;
  for (int m = 0; m <= C.mmax; m++) {
    double sn, cs;
    sincos((double)m * phi, &sn, &cs);                                // cos(phi*mm), sin(phi*mm) (:4081-4082)
    const int jc = (m == 0) ? 0 : 2 * m - 1;
    for (int k = 0; k < 4; k++) {
      double *u = u0 + (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT;
      unsafeAtomicAdd(u + jc, mass * cw[k] * cs);
      if (m) unsafeAtomicAdd(u + jc + 1, mass * cw[k] * sn);
    }
  }
  double *q = Q + ((size_t)T * C.numx * C.numy + (size_t)ix * C.numy + iy) * 10;
  int p = 0;
  for (int k = 0; k < 4; k++)
    for (int k2 = k; k2 < 4; k2++) unsafeAtomicAdd(q + p++, mass * cw[k] * cw[k2]);
}

// VC[T][m][n] (re, im): one block per (n, m, T), reduction over the nodes
__global__ void __launch_bounds__(256)
k_cyl_cov_mean(CylDev C, const double *__restrict__ tab, const double *__restrict__ U,
               double *__restrict__ vc)
{
  const int n = blockIdx.x, m = blockIdx.y, T = blockIdx.z;
  __shared__ double red[2][256];
  const size_t nnode = (size_t)(C.numx + 1) * (C.numy + 1);
  const double *TC = tab + (((size_t)0 * (C.mmax + 1) + m) * C.nmax + n) * nnode;
  const double *TS = tab + (((size_t)3 * (C.mmax + 1) + m) * C.nmax + n) * nnode;
  const double *u = U + (size_t)T * nnode * C.ntrig;
  const int jc = (m == 0) ? 0 : 2 * m - 1;
  double re = 0.0, im = 0.0;
  for (size_t k = threadIdx.x; k < nnode; k += 256) {
    const double uc = u[k * C.ntrig + jc];
    if (m == 0) { re = fma(uc, TC[k], re); continue; }
    const double us = u[k * C.ntrig + jc + 1];
    re += uc * TC[k] + us * TS[k];
    im += us * TC[k] - uc * TS[k];
  }
  red[0][threadIdx.x] = re; red[1][threadIdx.x] = im;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double norm = -4.0 * M_PI;
    double *o = vc + (((size_t)T * (C.mmax + 1) + m) * C.nmax + n) * 2;
    o[0] = norm * red[0][0];
    o[1] = norm * red[1][0];
  }
}

// MV[T][m][n][o] (re, im): one block per (m, T); the corner table values of a cell are staged in LDS
// blockIdx.z: a stretch of 1024 (n, o) pairs, four per thread (any nmax)
#define CYL_COV_MAXN 512
__global__ void __launch_bounds__(256)
k_cyl_cov_mv(CylDev C, const double *__restrict__ tab, const double *__restrict__ Q,
             double *__restrict__ mv)
{
  const int m = blockIdx.x, T = blockIdx.y, N = C.nmax;
  extern __shared__ double cov_lds[];                                // tc[4][N] | ts[4][N]
  __shared__ double qs[10];
  double *tc_ = cov_lds, *ts_ = cov_lds + 4 * N;
#define tc(k, n) tc_[(k) * N + (n)]
#define ts(k, n) ts_[(k) * N + (n)]
  const int nyp = C.numy + 1;
  const size_t nnode = (size_t)(C.numx + 1) * nyp, ncell = (size_t)C.numx * C.numy;
  const int npair = N * N;
  const int p0 = blockIdx.z * 1024, p1 = min(npair, p0 + 1024);
  double are[4] = {0, 0, 0, 0}, aim[4] = {0, 0, 0, 0};
  for (size_t cell = 0; cell < ncell; cell++) {
    const double *q = Q + ((size_t)T * ncell + cell) * 10;
    if (q[0] == 0.0 && q[4] == 0.0 && q[7] == 0.0 && q[9] == 0.0) continue;   // no mass in the cell
    __syncthreads();
    const int ix = (int)(cell / C.numy), iy = (int)(cell - (size_t)ix * C.numy);
    if (threadIdx.x < 10) qs[threadIdx.x] = q[threadIdx.x];
    for (int t = threadIdx.x; t < 8 * N; t += 256) {
      const int k = (t / N) & 3, cs = t / (4 * N), n = t % N;
      const size_t node = (size_t)(ix + (k & 1)) * nyp + iy + ((k & 2) ? 1 : 0);
      const double v = (cs && m == 0) ? 0.0
                       : tab[((((size_t)(cs ? 3 : 0)) * (C.mmax + 1) + m) * N + n) * nnode + node];
      if (cs) ts(k, n) = v; else tc(k, n) = v;
    }
    __syncthreads();
    for (int j = 0, p = p0 + threadIdx.x; p < p1; p += 256, j++) {
      const int n = p / N, o = p - n * N;
      double re = 0.0, im = 0.0;
      int qi = 0;
      for (int k = 0; k < 4; k++)
        for (int k2 = k; k2 < 4; k2++, qi++) {
          const double w = qs[qi];
          re += w * (tc(k, n) * tc(k2, o) + ts(k, n) * ts(k2, o));
          im += w * (tc(k, n) * ts(k2, o) - ts(k, n) * tc(k2, o));
          if (k2 != k) {                                             // the (k2, k) term of the double sum
            re += w * (tc(k2, n) * tc(k, o) + ts(k2, n) * ts(k, o));
            im += w * (tc(k2, n) * ts(k, o) - ts(k2, n) * tc(k, o));
          }
        }
      are[j] += re; aim[j] += im;
    }
  }
  const double norm2 = 16.0 * M_PI * M_PI;
  for (int j = 0, p = p0 + threadIdx.x; p < p1; p += 256, j++) {
    double *o = mv + ((((size_t)T * (C.mmax + 1) + m) * npair) + p) * 2;
    o[0] = norm2 * are[j];
    o[1] = norm2 * aim[j];
  }
#undef tc
#undef ts
}

static CylForce *as_cyl(exp_amd_force *fb) { return dynamic_cast<CylForce *>(fb); }

// enableCoefCovariance -> setSampT / set_covar (expui/BiorthBasis.H:1132-1145)
extern "C" int exp_amd_cyl_cov_enable(exp_amd_force *fb, int sampT)
{
  CylForce *f = as_cyl(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cyl_cov_enable: not a cylinder force");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  f->cov_U.release(); f->cov_Q.release(); f->cov_mass.release(); f->cov_vc.release(); f->cov_mv.release();
  f->cov_cnt.release(); f->cov_used.release();
  f->cov_T = 0;
  if (sampT <= 0) return EXP_AMD_OK;
  if (f->cfg.nmax > CYL_COV_MAXN) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cyl_cov_enable: nmax > %d", CYL_COV_MAXN);
  const CylDev &C = f->dev;
  const size_t nnode = (size_t)(C.numx + 1) * (C.numy + 1), ncell = (size_t)C.numx * C.numy;
  const size_t M1 = C.mmax + 1, N = C.nmax;
  if (f->cov_U.alloc((size_t)sampT * nnode * C.ntrig) != hipSuccess ||
      f->cov_Q.alloc((size_t)sampT * ncell * 10) != hipSuccess || f->cov_mass.alloc(sampT) != hipSuccess ||
      f->cov_cnt.alloc(sampT) != hipSuccess || f->cov_used.alloc(1) != hipSuccess ||
      f->cov_vc.alloc((size_t)sampT * M1 * N * 2) != hipSuccess ||
      f->cov_mv.alloc((size_t)sampT * M1 * N * N * 2) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_cov_enable: hipMalloc failed");
  f->cov_T = sampT;
  HIP_TRY(ctx, hipMemset(f->cov_U.p, 0, f->cov_U.bytes()));
  HIP_TRY(ctx, hipMemset(f->cov_Q.p, 0, f->cov_Q.bytes()));
  HIP_TRY(ctx, hipMemset(f->cov_mass.p, 0, f->cov_mass.bytes()));
  HIP_TRY(ctx, hipMemset(f->cov_cnt.p, 0, f->cov_cnt.bytes()));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_cyl_cov_reset(exp_amd_force *fb)
{
  CylForce *f = as_cyl(fb);
  if (!f || !f->cov_T) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cyl_cov_reset: covariance not enabled");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipMemsetAsync(f->cov_U.p, 0, f->cov_U.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(f->cov_Q.p, 0, f->cov_Q.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(f->cov_mass.p, 0, f->cov_mass.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(f->cov_cnt.p, 0, f->cov_cnt.bytes(), ctx->stream));
  return EXP_AMD_OK;
}

// seq[n] (caller order; NULL: 0 .. n-1) is the `seq` argument of EmpCylSL::accumulate
extern "C" int exp_amd_cyl_cov_accumulate(exp_amd_force *fb, exp_amd_comp *c, const uint32_t *seq,
                                          long long *on_grid)
{
  CylForce *f = as_cyl(fb);
  if (!f || !c || !f->cov_T) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cyl_cov_accumulate: covariance not enabled");
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }      // (an appended store: made an ordinary one first)
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (positions and masses only: whatever half-kick the velocities are owed or ahead by does not matter here)
  if (on_grid) *on_grid = 0;
  if (c->n == 0) return EXP_AMD_OK;
  if (seq) {
    if (f->cov_seq_cap < c->n) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (f->cov_seq.alloc(c->n) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_cov_accumulate: hipMalloc failed");
      f->cov_seq_cap = c->n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(f->cov_seq.p, seq, c->n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(ctx, hipMemsetAsync(f->cov_used.p, 0, sizeof(unsigned long long), ctx->stream));
  const CylDev C = cdev_for(f, c);
  {
    ProfScope ps(ctx, "k_cyl_covariance");
    k_cyl_cov_accumulate<<<cdiv(c->n, 256), 256, 0, ctx->stream>>>(
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->id[c->cur].p, seq ? f->cov_seq.p : nullptr, c->n,
        f->cov_T, f->cov_U.p, f->cov_Q.p, f->cov_cnt.p, f->cov_mass.p, f->cov_used.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  unsigned long long u = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&u, f->cov_used.p, sizeof(u), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (on_grid) *on_grid = (long long)u;
  return EXP_AMD_OK;
}

// EmpCylSL::getCovarSamples / getCoefCovariance (exputil/EmpCylSL.cc:4974-5015): counts[sampT],
// masses[sampT], VC[sampT][mmax+1][nmax][2], MV[sampT][mmax+1][nmax][nmax][2]; any may be NULL
extern "C" int exp_amd_cyl_cov_get(exp_amd_force *fb, long long *counts, double *masses, double *vc, double *mv)
{
  CylForce *f = as_cyl(fb);
  if (!f || !f->cov_T) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_STATE, "cyl_cov_get: covariance not enabled");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const CylDev &C = f->dev;
  k_cyl_cov_mean<<<dim3(C.nmax, C.mmax + 1, f->cov_T), 256, 0, ctx->stream>>>(C, f->d_tab.p, f->cov_U.p, f->cov_vc.p);
  k_cyl_cov_mv<<<dim3(C.mmax + 1, f->cov_T, cdiv((size_t)C.nmax * C.nmax, 1024)), 256, 8 * (size_t)C.nmax * sizeof(double),
                 ctx->stream>>>(C, f->d_tab.p, f->cov_Q.p, f->cov_mv.p);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<unsigned long long> cnt(f->cov_T);
  if (counts) HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), f->cov_cnt.p, f->cov_cnt.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (masses) HIP_TRY(ctx, hipMemcpyAsync(masses, f->cov_mass.p, f->cov_mass.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (vc) HIP_TRY(ctx, hipMemcpyAsync(vc, f->cov_vc.p, f->cov_vc.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (mv) HIP_TRY(ctx, hipMemcpyAsync(mv, f->cov_mv.p, f->cov_mv.bytes(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (counts) for (int t = 0; t < f->cov_T; t++) counts[t] = (long long)cnt[t];
  return EXP_AMD_OK;
}

// ---- the basis functions themselves on an (R, z) grid (pyEXP getBasis) and their orthogonality ----------------
// Cylindrical::getBasis (expui/BiorthBasis.cc:1930-1974) calls EmpCylSL::get_all(m, n, R, z, phi = 0, ...)
// (exputil/EmpCylSL.cc:5635-5800) for every (m, n): at phi = 0 only the cosine tables contribute; beyond the
// table radius the monopole -cylmass/r and its radial / vertical force.  One lane per point, blockIdx.y = m*nmax+n;
// out[4][mmax+1][nmax][npts] = potential, density, rforce, zforce.
__global__ void __launch_bounds__(256)
k_cyl_basis(CylDev C, const double *__restrict__ tab, const double *__restrict__ dens, double cylmass, size_t n,
            const double *__restrict__ Rv, const double *__restrict__ Zv, double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int mk = blockIdx.y;
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const size_t plane = half * n;
  double *o = out + (size_t)mk * n + i;
  const double r = Rv[i];
  double z = Zv[i];
  const double rr = sqrt(r * r + z * z);
  if (rr * C.inv_ascale > C.rtable) {                                  // :5653-5659
    const double p = -cylmass / (rr + 1.0e-16);
    o[0] = p; o[plane] = 0.0;
    o[2 * plane] = p * r / (rr + 1.0e-16) / (rr + 1.0e-16);
    o[3 * plane] = p * z / (rr + 1.0e-16) / (rr + 1.0e-16);
    return;
  }
  if (z * C.inv_ascale > C.rtable) z = C.rtab_abs;                     // :5661-5662
  if (z * C.inv_ascale < -C.rtable) z = -C.rtab_abs;
  int ix, iy;
  double c00, c10, c01, c11;
  cyl_weights(C, r, z, ix, iy, c00, c10, c01, c11);
  const size_t ny = (size_t)C.numy + 1, nnode = (size_t)(C.numx + 1) * ny;
  const size_t n00 = (size_t)ix * ny + iy;
  auto bl = [&](const double *T) {
    return T[n00] * c00 + T[n00 + ny] * c10 + T[n00 + 1] * c01 + T[n00 + ny + 1] * c11;
  };
  const double *Tc = tab + (size_t)mk * nnode;
  const size_t ks = half * nnode;
  o[0] = bl(Tc);
  o[plane] = bl(dens + (size_t)mk * nnode);
  o[2 * plane] = bl(Tc + ks);
  o[3 * plane] = bl(Tc + 2 * ks);
}

extern "C" int exp_amd_cyl_basis(exp_amd_force *fb, size_t n, const double *R, const double *z, double *out)
{
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cyl_basis: not a cylinder force");
  exp_amd_ctx *ctx = f->ctx;
  if (n == 0) return EXP_AMD_OK;
  if (!R || !z || !out) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cyl_basis: NULL argument");
  if (!f->d_dens.p) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "cyl_basis: call exp_amd_cyl_set_density first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t half = (size_t)(f->cfg.mmax + 1) * f->cfg.nmax;
  DevBuf<double> d_in, d_out;
  if (d_in.alloc(2 * n) != hipSuccess || d_out.alloc(4 * half * n) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_basis: hipMalloc failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_in.p, R, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_in.p + n, z, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  double cylmass = 0.0;
  HIP_TRY(ctx, hipMemcpyAsync(&cylmass, f->d_mass.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  k_cyl_basis<<<dim3(cdiv(n, 256), (unsigned)half), 256, 0, ctx->stream>>>(f->dev, f->d_tab.p, f->d_dens.p, cylmass, n,
                                                                           d_in.p, d_in.p + n, d_out.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, 4 * half * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_in.release();
  d_out.release();
  return EXP_AMD_OK;
}

// EmpCylSL::orthoCheck (exputil/EmpCylSL.cc:7199-7260) behind pyEXP's Cylindrical.orthoCheck: trapezoidal
// integral of pot x dens over the (X, Y) table grid with the gravitational-energy normalisation; cosine and
// sine parts of m > 0 combined as sqrt((C^2 + S^2)/2).  (As written the reference halves the weight of row
// iy == NUMX, not NUMY: restated.)  One block per (m, n1, n2).
__global__ void __launch_bounds__(256)
k_cyl_orthocheck(CylDev C, const double *__restrict__ tab, const double *__restrict__ dens, double *__restrict__ out)
{
  const int n2 = blockIdx.x % C.nmax, n1 = (blockIdx.x / C.nmax) % C.nmax, mm = blockIdx.x / (C.nmax * C.nmax);
  const size_t ny = (size_t)C.numy + 1, nnode = (size_t)(C.numx + 1) * ny;
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const double *pC = tab + ((size_t)mm * C.nmax + n1) * nnode;
  const double *pS = tab + (3 * half + (size_t)mm * C.nmax + n1) * nnode;
  const double *dC = dens + ((size_t)mm * C.nmax + n2) * nnode;
  const double *dS = dens + (half + (size_t)mm * C.nmax + n2) * nnode;
  double fac = -4.0 * M_PI * (2.0 * M_PI) * C.dx * C.dy;
  if (mm) fac *= 0.5;
  double sc = 0.0, ss = 0.0;
  for (size_t q = threadIdx.x; q < nnode; q += 256) {
    const int ix = (int)(q / ny), iy = (int)(q % ny);
    const double x = C.xmin + C.dx * ix, y = C.ymin + C.dy * iy;
    const double r = (C.cmapr > 0) ? (1.0 + x) / (1.0 - x) * C.ascale : x;
    const double dxr = (C.cmapr > 0) ? 0.5 * (1.0 - x) * (1.0 - x) / C.ascale : 1.0;
    double dyz = 1.0;
    if (C.cmapz == 1) dyz = C.hscale * cosh(y);
    else if (C.cmapz == 2) dyz = C.hscale * pow(1.0 - y * y, -1.5);
    const double fx = (ix == 0 || ix == C.numx) ? 0.5 : 1.0;
    const double fy = (iy == 0 || iy == C.numx) ? 0.5 : 1.0;
    const double jac = fac * r / dxr * dyz * fx * fy;
    sc += jac * pC[q] * dC[q];
    if (mm) ss += jac * pS[q] * dS[q];
  }
  __shared__ double red[2][4];
  for (int o = 32; o > 0; o >>= 1) { sc += __shfl_down(sc, o); ss += __shfl_down(ss, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sc; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out[blockIdx.x] = mm == 0 ? a : sqrt(0.5 * (a * a + b * b));
  }
}

extern "C" int exp_amd_cyl_orthocheck(exp_amd_force *fb, double *out)
{
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f || !out) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "cyl_orthocheck: not a cylinder force / NULL");
  exp_amd_ctx *ctx = f->ctx;
  if (!f->d_dens.p) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "cyl_orthocheck: call exp_amd_cyl_set_density first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t cnt = (size_t)(f->cfg.mmax + 1) * f->cfg.nmax * f->cfg.nmax;
  DevBuf<double> d_out;
  if (d_out.alloc(cnt) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_orthocheck: hipMalloc failed");
  k_cyl_orthocheck<<<(unsigned)cnt, 256, 0, ctx->stream>>>(f->dev, f->d_tab.p, f->d_dens.p, d_out.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_out.release();
  return EXP_AMD_OK;
}

