// Spherical BFE force method (sphereSL): host side of the C ABI plus the small table kernels
// (sort key, moments -> coefficients, coefficients -> projected tables).  The per-particle
// kernels live in sph_kernels.h and are instantiated per LMAX in sph_inst.hip.
#include "sph_kernels.h"
#include "sort_kernels.h"

// ---- sort key -------------------------------------------------------------------------------------

// key = level * (numr-1) + radial cell of get_pot/get_force (r clamped to rmax like the force path)
struct SphKeyFn {
  SphDev S;
  __device__ __forceinline__ uint32_t operator()(double x, double y, double z, uint8_t lev) const
  {
    const double xx = x - S.cx, yy = y - S.cy, zz = z - S.cz;
    double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
    if (r > S.rmax) r = S.rmax;               // src/SphericalBasis.cc:1555-1560
    const double xi = sph_r_to_xi(S, r / S.scale);
    return (uint32_t)lev * (uint32_t)(S.numr - 1) + (uint32_t)sph_cell(S, xi);
  }
};

// ---- moments -> coefficients ------------------------------------------------------------------------
// part[seg][row][n] = sum_{i in seg} E[i][l][n] W[i][row][0] + E[i+1][l][n] W[i][row][1]
#define CSEG 32
__global__ void __launch_bounds__(64)
k_sph_contract(SphDev S, const double *__restrict__ W, double *__restrict__ part)
{
  const int row = blockIdx.x, seg = blockIdx.y;
  int l = 0;
  while ((l + 1) * (l + 1) <= row) l++;
  const int ncell = S.numr - 1;
  const int per = (ncell + CSEG - 1) / CSEG;
  const int i0 = seg * per, i1 = min(ncell, i0 + per);
  const int stride = (S.lmax + 1) * S.nmax;
  for (int n = threadIdx.x; n < S.nmax; n += 64) {
    double s = 0.0;
    for (int i = i0; i < i1; i++) {
      const double w1 = W[((size_t)i * S.nrows + row) * 2];
      const double w2 = W[((size_t)i * S.nrows + row) * 2 + 1];
      s = fma(S.E[(size_t)i * stride + l * S.nmax + n], w1, s);
      s = fma(S.E[(size_t)(i + 1) * stride + l * S.nmax + n], w2, s);
    }
    part[((size_t)seg * S.nrows + row) * S.nmax + n] = s;
  }
}

__global__ void __launch_bounds__(256)
k_sph_sum_parts(const double *__restrict__ part, int ncoef, double *__restrict__ coef)
{
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncoef) return;
  double s = 0.0;
  for (int seg = 0; seg < CSEG; seg++) s += part[(size_t)seg * ncoef + k];
  coef[k] = s;
}

// ---- coefficients -> projected tables -----------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sph_project(SphDev S, const double *__restrict__ coef, double *__restrict__ G,
              double *__restrict__ H)
{
  const int i = blockIdx.x;
  const int stride = (S.lmax + 1) * S.nmax;
  for (int row = threadIdx.x; row < S.nrows; row += 256) {
    int l = 0;
    while ((l + 1) * (l + 1) <= row) l++;
    const double *e = S.E + (size_t)i * stride + l * S.nmax;
    const double *c = coef + (size_t)row * S.nmax;
    double s = 0.0;
    for (int n = 0; n < S.nmax; n++) s = fma(e[n], c[n], s);
    G[(size_t)i * S.nrows + row] = s;
    H[(size_t)i * S.nrows + row] = S.p0[i] * s;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------

struct exp_amd_force {
  exp_amd_ctx *ctx = nullptr;
  int kind = 0;                     // 0 = sphereSL
  exp_amd_sph_config cfg{};
  SphDev dev{};
  DevBuf<double> d_xi, d_p0, d_E, d_fact;
  DevBuf<double> d_W, d_part, d_G, d_H;
  DevBuf<double> d_coef;            // expcoef
  DevBuf<double> d_coefN, d_coefL;  // [multistep+1][ncoef]  (src/SphericalBasis.cc:785-792)
  DevBuf<unsigned long long> d_used;
  size_t ncoef = 0;
  int mlevel = 0;
  bool proj_dirty = true;
  exp_amd_comp *home = nullptr;
  std::vector<double> h_stage;
};

static double factrl(int n)
{
  double a = 1.0;
  for (int i = 2; i <= n; i++) a *= (double)i;
  return a;
}

extern "C" int exp_amd_sph_create(exp_amd_ctx *ctx, const exp_amd_sph_config *cfg, const double *xi,
                                  const double *p0, const double *ev, const double *ef,
                                  exp_amd_force **out)
{
  if (!ctx || !cfg || !xi || !p0 || !ev || !ef || !out)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_create: NULL argument");
  if (cfg->lmax < 0 || cfg->lmax > SPH_MAX_L)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_create: lmax=%d outside [0,%d]", cfg->lmax,
                       SPH_MAX_L);
  if (cfg->nmax < 1 || cfg->numr < 3 || cfg->cmap < 0 || cfg->cmap > 2 || cfg->multistep < 0 ||
      cfg->multistep > 16)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_create: bad nmax/numr/cmap/multistep");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  exp_amd_force *f = new exp_amd_force;
  f->ctx = ctx;
  f->cfg = *cfg;
  const int L = cfg->lmax, nmax = cfg->nmax, numr = cfg->numr;
  const int nrows = (L + 1) * (L + 1);
  f->ncoef = (size_t)nrows * nmax;

  // E[i][l][n] = ef_l(n,i)/sqrt(ev_l[n])
  std::vector<double> E((size_t)numr * (L + 1) * nmax);
  for (int l = 0; l <= L; l++)
    for (int n = 0; n < nmax; n++) {
      const double s = sqrt(ev[l * nmax + n]);
      const double *src = ef + ((size_t)l * nmax + n) * numr;
      for (int i = 0; i < numr; i++) E[((size_t)i * (L + 1) + l) * nmax + n] = src[i] / s;
    }
  // src/SphericalBasis.cc:328-335
  std::vector<double> fact((size_t)(L + 1) * (L + 1), 0.0);
  for (int l = 0; l <= L; l++)
    for (int m = 0; m <= l; m++) {
      double v = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * factrl(l - m) / factrl(l + m));
      if (m) v *= M_SQRT2;
      fact[l * (L + 1) + m] = v;
    }

  const int nlev = cfg->multistep + 1;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(f->d_xi.alloc(numr));
  A(f->d_p0.alloc(numr));
  A(f->d_E.alloc(E.size()));
  A(f->d_fact.alloc(fact.size()));
  A(f->d_W.alloc((size_t)(numr - 1) * nrows * 2));
  A(f->d_part.alloc((size_t)CSEG * f->ncoef));
  A(f->d_G.alloc((size_t)numr * nrows));
  A(f->d_H.alloc((size_t)numr * nrows));
  A(f->d_coef.alloc(f->ncoef));
  A(f->d_coefN.alloc((size_t)nlev * f->ncoef));
  A(f->d_coefL.alloc((size_t)nlev * f->ncoef));
  A(f->d_used.alloc(1));
  if (e != hipSuccess) {
    exp_amd_force_destroy(f);
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_create: hipMalloc failed: %s",
                       hipGetErrorString(e));
  }
  HIP_TRY(ctx, hipMemcpy(f->d_xi.p, xi, numr * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_p0.p, p0, numr * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_E.p, E.data(), E.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_fact.p, fact.data(), fact.size() * sizeof(double),
                         hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemset(f->d_coef.p, 0, f->d_coef.bytes()));
  HIP_TRY(ctx, hipMemset(f->d_coefN.p, 0, f->d_coefN.bytes()));
  HIP_TRY(ctx, hipMemset(f->d_coefL.p, 0, f->d_coefL.bytes()));
  HIP_TRY(ctx, hipMemset(f->d_used.p, 0, sizeof(unsigned long long)));

  SphDev &S = f->dev;
  S.lmax = L; S.nmax = nmax; S.numr = numr; S.cmap = cfg->cmap; S.nrows = nrows;
  S.rmap = cfg->rmap; S.scale = cfg->scale; S.rmin = cfg->rmin; S.rmax = cfg->rmax;
  S.xmin = cfg->xmin; S.dxi = cfg->dxi;
  S.cx = S.cy = S.cz = 0.0;
  S.NO_L0 = cfg->NO_L0; S.NO_L1 = cfg->NO_L1; S.EVEN_L = cfg->EVEN_L; S.EVEN_M = cfg->EVEN_M;
  S.M0_only = cfg->M0_only;
  S.xi = f->d_xi.p; S.p0 = f->d_p0.p; S.E = f->d_E.p; S.fact = f->d_fact.p;
  *out = f;
  return EXP_AMD_OK;
}

extern "C" void exp_amd_force_destroy(exp_amd_force *f)
{
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  f->d_xi.release(); f->d_p0.release(); f->d_E.release(); f->d_fact.release();
  f->d_W.release(); f->d_part.release(); f->d_G.release(); f->d_H.release();
  f->d_coef.release(); f->d_coefN.release(); f->d_coefL.release(); f->d_used.release();
  delete f;
}

extern "C" size_t exp_amd_force_ncoef(const exp_amd_force *f) { return f ? f->ncoef : 0; }

extern "C" int exp_amd_force_set_level(exp_amd_force *f, int mlevel)
{
  if (!f || mlevel < 0 || mlevel > f->cfg.multistep)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "set_level: level out of range");
  f->mlevel = mlevel;
  return EXP_AMD_OK;
}

static SphDev dev_for(const exp_amd_force *f, const double center[3])
{
  SphDev S = f->dev;
  S.cx = center[0]; S.cy = center[1]; S.cz = center[2];
  return S;
}

// (level, radial cell) order for this force's tables; with `advance` the kick dt_kick and drift
// dt_drift of the leapfrog are applied on the way (src/step.cc:279-288)
static int sph_sort(exp_amd_force *f, exp_amd_comp *c, bool move_acc, bool advance = false,
                    double dt_kick = 0.0, double dt_drift = 0.0)
{
  exp_amd_ctx *ctx = f->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  c->nlevels = f->cfg.multistep + 1;
  const uint32_t ncell = (uint32_t)(f->cfg.numr - 1);
  const uint32_t nkeys = ncell * (uint32_t)c->nlevels;
  int rc = expamd_comp_prepare_hist(c, nkeys);
  if (rc) return rc;
  {
    ProfScope ps(ctx, "k_key_hist");
    SphKeyFn kf{dev_for(f, c->center)};
    AdvanceArgs A = expamd_advance_args(c, advance, dt_kick, dt_drift);
    k_key_hist<SphKeyFn><<<cdiv(c->n, SORT_TILE), SORT_TPB, 0, ctx->stream>>>(kf, A, c->n, c->key.p,
                                                                              c->hist.p);
  }
  rc = expamd_comp_finish_sort(c, nkeys, ncell, move_acc, advance, dt_kick, dt_drift);
  if (rc) return rc;
  c->sorted_for = f;
  return EXP_AMD_OK;
}

#define DECL_L(k)                                        \
  void expamd_sph_acc_L##k(const SphAccArgs &);          \
  void expamd_sph_force_L##k(const SphForceArgs &);
DECL_L(0) DECL_L(1) DECL_L(2) DECL_L(3) DECL_L(4) DECL_L(5) DECL_L(6)
DECL_L(7) DECL_L(8) DECL_L(9) DECL_L(10) DECL_L(11) DECL_L(12)
#undef DECL_L
static const sph_acc_launcher k_acc_launch[SPH_MAX_L + 1] = {
    expamd_sph_acc_L0, expamd_sph_acc_L1, expamd_sph_acc_L2,  expamd_sph_acc_L3,  expamd_sph_acc_L4,
    expamd_sph_acc_L5, expamd_sph_acc_L6, expamd_sph_acc_L7,  expamd_sph_acc_L8,  expamd_sph_acc_L9,
    expamd_sph_acc_L10, expamd_sph_acc_L11, expamd_sph_acc_L12};
static const sph_force_launcher k_force_launch[SPH_MAX_L + 1] = {
    expamd_sph_force_L0, expamd_sph_force_L1, expamd_sph_force_L2,  expamd_sph_force_L3,
    expamd_sph_force_L4, expamd_sph_force_L5, expamd_sph_force_L6,  expamd_sph_force_L7,
    expamd_sph_force_L8, expamd_sph_force_L9, expamd_sph_force_L10, expamd_sph_force_L11,
    expamd_sph_force_L12};

static int sph_accumulate(exp_amd_force *f, exp_amd_comp *c, double *d_out)
{
  exp_amd_ctx *ctx = f->ctx;
  const int lo = f->cfg.multistep ? f->mlevel : 0;
  const int hi = f->cfg.multistep ? f->mlevel : 0;
  SphDev S = dev_for(f, c->center);
  HIP_TRY(ctx, hipMemsetAsync(f->d_W.p, 0, f->d_W.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(f->d_used.p, 0, sizeof(unsigned long long), ctx->stream));
  if (c->n) {
    ProfScope ps(ctx, "k_sph_accumulate");
    SphAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, lo, hi,
                 f->d_W.p, f->d_used.p, c->n, ctx->stream};
    k_acc_launch[f->cfg.lmax](a);
  }
  {
    ProfScope ps(ctx, "k_sph_contract");
    k_sph_contract<<<dim3(S.nrows, CSEG), 64, 0, ctx->stream>>>(S, f->d_W.p, f->d_part.p);
    k_sph_sum_parts<<<cdiv(f->ncoef, 256), 256, 0, ctx->stream>>>(f->d_part.p, (int)f->ncoef,
                                                                  d_out);
  }
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

static int sph_determine_coefficients(exp_amd_force *f, exp_amd_comp *c, bool advance,
                                      double dt_kick, double dt_drift)
{
  exp_amd_ctx *ctx = f->ctx;
  f->home = c;
  int rc = sph_sort(f, c, c->acc_live, advance, dt_kick, dt_drift);
  if (rc) return rc;
  double *dst = f->cfg.multistep ? f->d_coefN.p + (size_t)f->mlevel * f->ncoef : f->d_coef.p;
  if (f->cfg.multistep) {
    // swap N/L buffers of this level (src/SphericalBasis.cc:785-792): L <- N, then N <- new
    HIP_TRY(ctx, hipMemcpyAsync(f->d_coefL.p + (size_t)f->mlevel * f->ncoef, dst,
                                f->ncoef * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  rc = sph_accumulate(f, c, dst);
  if (rc) return rc;
  rc = expamd_allreduce(ctx, dst, f->ncoef);
  if (rc) return rc;
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_determine_coefficients(exp_amd_force *f, exp_amd_comp *c)
{
  if (!f || !c) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "determine_coefficients: NULL");
  return sph_determine_coefficients(f, c, false, 0.0, 0.0);
}

static int sph_project(exp_amd_force *f)
{
  if (!f->proj_dirty) return EXP_AMD_OK;
  exp_amd_ctx *ctx = f->ctx;
  ProfScope ps(ctx, "k_sph_project");
  k_sph_project<<<f->cfg.numr, 256, 0, ctx->stream>>>(f->dev, f->d_coef.p, f->d_G.p, f->d_H.p);
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = false;
  return EXP_AMD_OK;
}

static int sph_force(exp_amd_force *f, exp_amd_comp *t, int external, bool assign, double dt_kick)
{
  exp_amd_ctx *ctx = f->ctx;
  int rc = sph_project(f);
  if (rc) return rc;
  if (t->n == 0) return EXP_AMD_OK;
  const double *ctr = (external && f->home) ? f->home->center : t->center;
  SphDev S = dev_for(f, ctr);
  const int lo = (t->nlevels > 1) ? f->mlevel : 0;
  const int hi = t->nlevels - 1;
  {
    ProfScope ps(ctx, "k_sph_force");
    unsigned grid = cdiv(t->n, 256);   // one 64-particle chunk per wave, no loop
    SphForceArgs a{S, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_G.p, f->d_H.p,
                   t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY),
                   t->a(A_VZ), dt_kick, assign ? 1 : 0, t->n, grid, ctx->stream};
    k_force_launch[f->cfg.lmax](a);
  }
  HIP_TRY(ctx, hipGetLastError());
  t->acc_live = true;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_get_acceleration(exp_amd_force *f, exp_amd_comp *target, int external)
{
  if (!f || !target) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "get_acceleration: NULL");
  return sph_force(f, target, external, false, 0.0);
}

extern "C" int exp_amd_force_get_coefs(exp_amd_force *f, double *coef, size_t count)
{
  if (!f || !coef || count != f->ncoef)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "get_coefs: bad count");
  exp_amd_ctx *ctx = f->ctx;
  const double *src = f->d_coef.p;
  HIP_TRY(ctx, hipMemcpyAsync(coef, src, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_set_coefs(exp_amd_force *f, const double *coef, size_t count)
{
  if (!f || !coef || count != f->ncoef)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "set_coefs: bad count");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef.p, coef, count * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_used(exp_amd_force *f, long long *used)
{
  if (!f || !used) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = f->ctx;
  unsigned long long u = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&u, f->d_used.p, sizeof(u), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *used = (long long)u;
  return EXP_AMD_OK;
}

// ---- multistep coefficient bookkeeping -----------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_mstep_combine(const double *__restrict__ L, const double *__restrict__ N, int ncoef, int nlev,
                int mfirst, const double *__restrict__ ab, double *__restrict__ out)
{
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncoef) return;
  double s = 0.0;
  // src/SphericalBasis.cc:1252-1333
  for (int M = 0; M < mfirst; M++)
    s += ab[2 * M] * L[(size_t)M * ncoef + k] + ab[2 * M + 1] * N[(size_t)M * ncoef + k];
  for (int M = mfirst; M < nlev; M++) s += N[(size_t)M * ncoef + k];
  out[k] = s;
}

extern "C" int exp_amd_force_multistep_reset(exp_amd_force *f)
{
  if (!f) return EXP_AMD_ERR_ARG;
  return EXP_AMD_OK;   // src/SphericalBasis.cc multistep_reset: nothing to do per step
}

extern "C" int exp_amd_force_compute_multistep_coefficients(exp_amd_force *f, int mdrft)
{
  if (!f) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = f->ctx;
  const int ms = f->cfg.multistep;
  if (ms == 0) return EXP_AMD_OK;
  const int Mstep = 1 << ms;
  if (mdrft < 0 || mdrft > Mstep) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "mdrft out of range");
  // src/multistep.cc:630-680: mfirst[mdrft], dstepL/N[M][mdrft]
  int mfirst = 0;
  for (int M = 0; M <= ms; M++) {
    bool active = (mdrft == 0) || (mdrft % (1 << (ms - M)) == 0);
    if (active) { mfirst = M; break; }
  }
  std::vector<double> ab(2 * (ms + 1), 0.0);
  for (int M = 0; M < mfirst; M++) {
    const int d = 1 << (ms - M);
    const int dL = (mdrft / d) * d, dN = dL + d;
    const double b = (double)(mdrft - dL) / (double)(dN - dL);
    ab[2 * M] = 1.0 - b;
    ab[2 * M + 1] = b;
  }
  // tiny host->device copy on the stream (pageable memory: copied before return)
  double *d_ab = f->d_part.p;   // scratch
  HIP_TRY(ctx, hipMemcpyAsync(d_ab, ab.data(), ab.size() * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  k_mstep_combine<<<cdiv(f->ncoef, 256), 256, 0, ctx->stream>>>(
      f->d_coefL.p, f->d_coefN.p, (int)f->ncoef, ms + 1, mfirst, d_ab, f->d_coef.p);
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

// ---- fused step -----------------------------------------------------------------------------------------------

extern "C" int exp_amd_step_kdk(exp_amd_force *f, exp_amd_comp *c, double dt)
{
  if (!f || !c) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "step_kdk: NULL");
  if (f->cfg.multistep) return expamd_fail(f->ctx, EXP_AMD_ERR_STATE, "step_kdk: multistep force; drive the sub-steps explicitly");
  int rc;
  // kick dt/2 + drift dt are applied inside the sort passes (no separate HBM pass); acc/pot are
  // recomputed below, so they are not carried through the reorder
  c->acc_live = false;
  if (c->n == 0) {
    if ((rc = sph_determine_coefficients(f, c, false, 0.0, 0.0))) return rc;
  } else if ((rc = sph_determine_coefficients(f, c, true, 0.5 * dt, dt))) return rc;
  if ((rc = sph_force(f, c, 0, true, 0.5 * dt))) return rc;
  return EXP_AMD_OK;
}
