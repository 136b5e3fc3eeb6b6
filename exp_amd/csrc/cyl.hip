// Cylindrical BFE force method (cylinder / EmpCylSL): host side of the C ABI -- creation, cell sort, coefficient
// accumulation, level-change differencing, block-multistep expansion and the force pass.  The kernels are in
// cyl_kernels.h, the analysis entry points (fields, covariance, basis functions) in cyl_fields.hip.
#include "cyl_kernels.h"
#include "cyl_force.h"
#include "kick_adjust.h"

extern "C" int exp_amd_cyl_create(exp_amd_ctx *ctx, const exp_amd_cyl_config *cfg, const double *tab,
                                  exp_amd_force **out)
{
  if (!ctx || !cfg || !tab || !out) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cyl_create: NULL argument");
  if (cfg->mmax < 0 || cfg->mmax > CYL_GEN_MAX_M || cfg->nmax < 1 || cfg->numx < 1 || cfg->numy < 1 ||
      cfg->multistep < 0 || cfg->multistep > 16)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "cyl_create: bad mmax/nmax/numx/numy/multistep");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  CylForce *f = new CylForce;
  f->ctx = ctx;
  f->cfg = *cfg;
  const int M = cfg->mmax, N = cfg->nmax;
  f->nnode = (size_t)(cfg->numx + 1) * (cfg->numy + 1);
  const size_t ntab = (size_t)6 * (M + 1) * N * f->nnode;
  const int ntrig = 2 * M + 1;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(f->d_tab.alloc(ntab));
  A(f->d_Wn.alloc((size_t)(cfg->multistep + 1) * f->nnode * ntrig));   // one moment buffer per level
  A(f->d_TF.alloc(f->nnode * 3 * ntrig));
  A(f->d_cpart.alloc((size_t)(cfg->multistep + 1) * CYL_CSEG * 2 * (M + 1) * N));
  A(f->d_mass.alloc(2));
  A(f->d_tailpart.alloc(2 * CYL_TAILS));
  // coefficient buffer: cos block, sin block, then {cylmass, used} riding through the all-reduce
  if (e == hipSuccess && f->alloc_common((size_t)2 * (M + 1) * N, cfg->multistep, 2) != EXP_AMD_OK)
    e = hipErrorOutOfMemory;
  if (e != hipSuccess) {
    exp_amd_force_destroy(f);
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cyl_create: hipMalloc failed: %s", hipGetErrorString(e));
  }
  HIP_TRY(ctx, hipMemcpy(f->d_tab.p, tab, ntab * sizeof(double), hipMemcpyHostToDevice));
  {
    // tab[kind][m][n][node], kinds 0-2 cosine {potential, radial, vertical force}, 3-5 sine: compare the m >= 1 parts
    const size_t per_kind = (size_t)(M + 1) * N * f->nnode, skip = (size_t)N * f->nnode;
    f->tab_twin = M >= 1;
    if (const char *ev = getenv("EXP_AMD_CYL_TWIN")) if (atoi(ev) == 0) f->tab_twin = false;
    for (int k = 0; k < 3 && f->tab_twin; k++)
      f->tab_twin = memcmp(tab + k * per_kind + skip, tab + (k + 3) * per_kind + skip, (per_kind - skip) * sizeof(double)) == 0;
  }
  f->generic = M > CYL_MAX_M;
  if (const char *eg = getenv("EXP_AMD_CYL_GENERIC")) if (atoi(eg) != 0) f->generic = true;
  HIP_TRY(ctx, hipMemset(f->d_mass.p, 0, 2 * sizeof(double)));
  HIP_TRY(ctx, hipMemset(f->d_tailpart.p, 0, 2 * CYL_TAILS * sizeof(double)));
  CylDev &C = f->dev;
  C.mmax = M; C.nmax = N; C.numx = cfg->numx; C.numy = cfg->numy; C.cmapr = cfg->cmapr;
  C.cmapz = cfg->cmapz; C.EVEN_M = cfg->EVEN_M; C.ntrig = ntrig;
  C.ascale = cfg->ascale; C.hscale = cfg->hscale; C.rtable = cfg->rtable;
  C.inv_ascale = 1.0 / cfg->ascale; C.inv_hscale = 1.0 / cfg->hscale;
  C.rtab_abs = cfg->rtable * cfg->ascale; C.inv_rtab_abs = 1.0 / C.rtab_abs;
  C.xmin = cfg->xmin; C.dx = cfg->dx; C.ymin = cfg->ymin; C.dy = cfg->dy;
  C.inv_dx = 1.0 / cfg->dx; C.inv_dy = 1.0 / cfg->dy;
  C.umass = 0.0;
  C.mscale = 1.0;
  C.frz = nullptr;
  C.rmax2 = cfg->rcylmax * cfg->rcylmax * cfg->ascale * cfg->ascale;   // src/Cylinder.cc:752
  C.cx = C.cy = C.cz = 0.0;
  *out = f;
  return EXP_AMD_OK;
}


int CylForce::ensure_tabT()
{
  if (d_tabT.p) return EXP_AMD_OK;
  const size_t half = (size_t)(cfg.mmax + 1) * cfg.nmax;
  tabT_nk = tab_twin ? 3 : 6;
  const int per_node = (int)(tabT_nk * half);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (d_tabT.alloc(nnode * (size_t)per_node) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "cylinder: hipMalloc of the node-major table copy failed");
  k_cyl_transpose<<<cdiv(nnode * (size_t)per_node, 256), 256, 0, ctx->stream>>>(d_tab.p, d_tabT.p, nnode, per_node);
  HIP_TRY(ctx, hipGetLastError());
  // (made once; finished before anything else is issued: the step driver reads it from BOTH of its streams -- the
  // self expansion on the component's own, a cross force on the target's -- and only this stream is ordered behind it)
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

// tile sizes of the thin kernels: what fits the LDS
template <int MM>
static void cyl_thin_force_launch(hipStream_t st, size_t n, const CylDev &C, const double *X, const double *Y, const double *Z,
                                  const uint32_t *lev_off, int lo, int hi, const double *tabT, int nk, const double *coef,
                                  const double *mass, double *AX, double *AY, double *AZ, double *POT, double *VX, double *VY,
                                  double *VZ, int assign)
{
  const int tp0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TP", 4);
  const int tp = tp0 < 1 ? 1 : tp0 > 64 ? 64 : tp0;      // (small tiles: see the spherical launcher)
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const size_t lds = (2 * half + (size_t)tp * (4 * (3 * (2 * MM + 1) + 1) + 2)) * sizeof(double);
  size_t grid = cdiv(n, (size_t)tp);
  if (grid > 16384) grid = 16384;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_cyl_force_thin<MM>), "k_cyl_force_thin<MM>");
  (void)big;
  const int nt0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_NT", 0);
  const int nt = nt0 ? nt0 : 256;
  k_cyl_force_thin<MM><<<(unsigned)grid, nt, lds, st>>>(C, X, Y, Z, lev_off, lo, hi, tabT, nk, coef, mass, AX, AY, AZ, POT, VX,
                                                         VY, VZ, assign, tp);
}

template <int MM>
static void cyl_thin_acc_launch(hipStream_t st, size_t n, const CylDev &C, const double *X, const double *Y, const double *Z,
                                const double *M, const uint32_t *lev_off, int lo, int hi, const double *tabT, int nk,
                                double *part, double *tail, const ThinAdv &adv)
{
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const int nset = nk == 3 ? 1 : 2;
  const int tpa0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TPA", 8);
  int tpa = tpa0 < 1 ? 1 : tpa0 > 64 ? 64 : tpa0;
  while (tpa > 4 && (size_t)tpa * nset * half * sizeof(double) > 96 * 1024) tpa >>= 1;
  size_t grid = cdiv(n, (size_t)tpa);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_cyl_acc_thin<MM>), "k_cyl_acc_thin<MM>");
  (void)big;
  k_cyl_acc_thin<MM><<<(unsigned)grid, 256, (size_t)tpa * nset * half * sizeof(double), st>>>(C, X, Y, Z, M, lev_off, lo, hi, tabT,
                                                                                                 nk, part, tail, tpa, adv);
}

template <int MM>
static void cyl_thin_diff_launch(hipStream_t st, size_t n, const CylDev &C, const double *X, const double *Y, const double *Z,
                                 const double *M, const uint32_t *list, const uint32_t *cnt, const uint8_t *lev,
                                 const uint8_t *newlev, int mfirst, int nl, const double *tabT, int nk, double *part)
{
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const int nset = nk == 3 ? 1 : 2;
  size_t grid = cdiv(n, (size_t)8);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_cyl_diff_thin<MM>), "k_cyl_diff_thin<MM>");
  (void)big;
  k_cyl_diff_thin<MM><<<(unsigned)grid, 256, (size_t)8 * nset * half * sizeof(double), st>>>(C, X, Y, Z, M, list, cnt, lev, newlev,
                                                                                              mfirst, nl, tabT, nk, part);
}

// second formulation of the thin kernels (any order): one wave per particle / 64-particle tiles
static void cyl_wave_force_launch(hipStream_t st, size_t n, const CylDev &C, const double *X, const double *Y, const double *Z,
                                  const uint32_t *lev_off, int lo, int hi, const double *tabT, int nk, const double *coef,
                                  const double *mass, double *AX, double *AY, double *AZ, double *POT, double *VX, double *VY,
                                  double *VZ, int assign)
{
  size_t grid = cdiv(n, 4);
  if (grid > 16384) grid = 16384;
  if (grid == 0) return;
  k_cyl_force_wave<<<(unsigned)grid, 256, 0, st>>>(C, X, Y, Z, lev_off, lo, hi, tabT, nk, coef, mass, AX, AY, AZ, POT, VX, VY, VZ,
                                                   assign);
}

static void cyl_tile_acc_launch(hipStream_t st, size_t n, const CylDev &C, const double *X, const double *Y, const double *Z,
                                const double *M, const uint32_t *lev_off, int lo, int hi, const double *tabT, int nk,
                                double *part, double *tail)
{
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const int nset = nk == 3 ? 1 : 2;
  const int tile0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TILE", 64);
  int tile = tile0 < 4 ? 4 : tile0 > CYL_TILE_MAX ? CYL_TILE_MAX : tile0;
  auto need = [&](int t) { return ((((size_t)t * (C.ntrig | 1) + 1) & ~(size_t)1) + (size_t)t * nset * half) * sizeof(double); };
  while (tile > 4 && need(tile) > 120 * 1024) tile >>= 1;
  size_t grid = cdiv(n, (size_t)tile);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_cyl_acc_tile), "k_cyl_acc_tile");
  (void)big;
  k_cyl_acc_tile<<<(unsigned)grid, 256, need(tile), st>>>(C, X, Y, Z, M, lev_off, lo, hi, tabT, nk, part, tail, tile);
}

__global__ void k_cyl_mass(double *__restrict__ acc, const double *__restrict__ tail, int overwrite)
{
  if (threadIdx.x < 2) acc[threadIdx.x] = (overwrite ? 0.0 : acc[threadIdx.x]) + tail[threadIdx.x];
}

// step-driver form: {in-cut mass, count} of the sub-step are added while the master step's first
// sub-step is open, and the tail is left zero for the next accumulation either way
__global__ void k_cyl_mass_take(double *__restrict__ acc, double *__restrict__ tail, int open)
{
  if (threadIdx.x < 2) {
    if (open) acc[threadIdx.x] += tail[threadIdx.x];
    tail[threadIdx.x] = 0.0;
  }
}

int CylForce::sort(exp_amd_comp *c, bool move_acc, const AdvSpec &adv, int level, bool have_keys,
                   int level_hi)
{
  CylForce *f = this;
  if (c->n == 0) return EXP_AMD_OK;
  const CylDev C = cdev_for(f, c);
  c->nlevels = f->multistep + 1;
  const uint32_t ncell = (uint32_t)(cfg.numx * cfg.numy) + 1u;
  const uint32_t nkeys = ncell * (uint32_t)c->nlevels;
  int rc = expamd_comp_prepare_hist(c, nkeys);
  if (rc) return rc;
  if (have_keys && level < 0) {
    // c->key was written by the previous fused step's force pass for exactly this advance
    ProfScope ps(ctx, "k_hist_keys");
    // (a block-multistep run: the closing sweep's full keys; those of the levels that are not cell-sorted collapse here)
    // (a dense store: a tile's keys are a few neighbouring (x, y) cells, a row of numx apart at most -- the shorter LDS window
    // and tiles of the sort passes, sort_kernels.h: SORT_WIN_DENSE2D)
    // (one-level stores only, as the sphere's: sph.hip)
    const long long dense_mode = EXPAMD_EXPT("EXP_AMD_SORT_DENSE", 1);
    c->sort_win = (dense_mode != 0 && (c->nlevels == 1 || dense_mode >= 2) && c->n >= (size_t)SORT_DENSE_MIN * ncell &&
                   4u * (uint32_t)cfg.numx <= SORT_WIN_DENSE2D) ? SORT_WIN_DENSE2D : 0;
    k_hist_keys<<<cdiv(c->n, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(c->key.p, c->n, c->hist.p,
                                                                    f->multistep ? c->sparse_mask : 0u, ncell,
                                                                    c->sort_win ? c->sort_win : (uint32_t)SORT_WIN);
  } else {
    size_t nr = c->n;          // a level range is sized for its own population
    if (level >= 0 && (rc = expamd_comp_level_count(c, level, level_hi > level ? level_hi : level, &nr))) return rc;
    if (nr == 0) return EXP_AMD_OK;
    ProfScope ps(ctx, "k_key_hist");
    CylKeyFn kf{C, c->sparse_mask};
    AdvanceArgs A = expamd_advance_args(c, adv);
    if (nr <= HIST_SHORT_MAX)
      k_key_hist<CylKeyFn, HIST_ITEMS_SHORT><<<cdiv(nr, SORT_TPB * HIST_ITEMS_SHORT), SORT_TPB, 0, ctx->stream>>>(
          kf, A, expamd_sort_range(c, level, level_hi), c->key.p, c->hist.p);
    else
    k_key_hist<CylKeyFn><<<cdiv(nr, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(
        kf, A, expamd_sort_range(c, level, level_hi), c->key.p, c->hist.p);
  }
  rc = expamd_comp_finish_sort(c, nkeys, ncell, move_acc, adv, level, level_hi);
  if (rc) return rc;
  c->sorted_for = f;
  return EXP_AMD_OK;
}

int CylForce::resort(exp_amd_comp *c, int first)
{
  if (first > 0 && c->nlevels == multistep + 1)      // (the caller vouches for the order below `first`)
    return sort(c, true, AdvSpec(), first, false, multistep);
  return sort(c, true, AdvSpec());
}

__global__ void __launch_bounds__(256)
k_cyl_add_inplace(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) dst[k] += src[k];
}

bool CylForce::prekey_launcher(exp_amd_comp *c, ka_launch_fn *fn, void **self)
{
  if (multistep == 0 || c->n == 0 || c->n >= 0x7fffffffu) return false;
  ka_C = cdev_for(this, c);
  *fn = [](void *p, const KaLaunch &L) { ka_launch_with(L, CylKeyFn{static_cast<CylForce *>(p)->ka_C, 0u}); };
  *self = this;
  return true;
}

int CylForce::multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft)
{
  CylForce *f = this;
  const int ms = f->multistep;
  if (ms == 0) return EXP_AMD_OK;
  const size_t wl = f->nnode * dev.ntrig;
  if (f->d_Wnd.n == 0) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, f->d_Wnd.alloc(wl * (ms + 1)));
    HIP_TRY(ctx, f->d_differ.alloc(f->ncoef_dev * (ms + 1)));
  }
  // the levels that multistep_update_begin clears and _finish adds (M >= mfirst[mdrft],
  // src/CylEXP.cc:45-157); a rank without particles still takes part in the reduction
  const int nl = ms - mfirst_mdrft + 1;
  // (the contraction below leaves the moments it consumed zeroed -- proposed levels are >= mfirst[mdrft],
  // src/multistep.cc:196, so nothing else is ever written -- and overwrites every coefficient of d_differ; the
  // {mass, count} tails of d_differ are never written at all)
  if (!f->wnd_clean) {
    HIP_TRY(ctx, hipMemsetAsync(f->d_Wnd.p, 0, f->d_Wnd.bytes(), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(f->d_differ.p, 0, f->d_differ.bytes(), ctx->stream));
    f->wnd_clean = true;
  }
  const CylDev C = cdev_acc(f, c);
  size_t nr = 0;
  if (c->n) { int rc_ = expamd_comp_level_count(c, first, ms, &nr); if (rc_) return rc_; }
  // the step driver knows how many particles change level (c->mover_hint): their slots are compacted first
  // (k_mover_list, 16 slots per thread), so that the differencing launches over the movers, not over the range
  const bool listed = nr && c->mover_hint >= 0;
  if (listed && c->mover_hint > 0) { int rc_ = expamd_comp_mover_list(c, first, ms, (size_t)c->mover_hint); if (rc_) return rc_; }
  const bool thin_diff_on = EXPAMD_EXPT("EXP_AMD_THIN_DIFF", 1) != 0;
  const bool few = listed && c->mover_hint > 0 && !(ctx->mover_list_min >= 0 && c->mover_hint >= ctx->mover_list_min);
  const bool thin_diff = few && thin_diff_on && ctx->thin_max > 0 && c->mover_hint <= ctx->thin_max && !ctx->deterministic &&
                         !f->generic && (size_t)8 * 2 * (cfg.mmax + 1) * cfg.nmax * sizeof(double) <= 96 * 1024;
  if (listed && c->mover_hint == 0) {
    // nothing moved on this rank (it only takes part in the reduction)
  } else if (thin_diff) {
    // few movers, straight from the basis tables into the contraction's stage-1 sums (k_cyl_diff_thin): no node moments,
    // no pass over the nodes of every level
    ProfScope ps(ctx, "k_cyl_diff_thin");
    { int rc_ = ensure_tabT(); if (rc_) return rc_; }
    if (!f->cpart_clean) {
      HIP_TRY(ctx, hipMemsetAsync(f->d_cpart.p, 0, f->d_cpart.bytes(), ctx->stream));
      f->cpart_clean = true;
    }
#define CALL(MM)                                                                                                   \
  cyl_thin_diff_launch<MM>(ctx->stream, (size_t)c->mover_hint, C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M),      \
                           c->mover_list.p, c->mover_cnt, c->level[c->cur].p, c->newlev.p, mfirst_mdrft, nl,      \
                           f->d_tabT.p, f->tabT_nk, f->d_cpart.p)
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
  } else if (listed && ctx->mover_list_min >= 0 && c->mover_hint >= ctx->mover_list_min && !f->generic) {
    // many movers: through the accumulation kernel (CylAccList)
    ProfScope ps(ctx, "k_cyl_mstep_update");
    const size_t nm = (size_t)c->mover_hint;
    LevChunks LC;
    LC.lo = 0; LC.nlev = 1;
    size_t chunk = (nm / ((size_t)CACC_WAVES * 3072)) & ~(size_t)63;
    chunk = chunk < 64 ? 64 : chunk > CACC_CHUNK_MAX ? CACC_CHUNK_MAX : chunk;
    LC.bstart[0] = 0;
    LC.chunk[0] = (int)chunk;
    LC.bstart[1] = cdiv(nm, (size_t)CACC_WAVES * chunk);
    const int per_level = c->mover_hint >= ctx->mover_slices_min ? 1 : 0;
    const dim3 grid(LC.bstart[1], 1, per_level ? ms + 2 : 2);
    const CylAccList al{c->mover_list.p, c->level[c->cur].p, c->newlev.p, mfirst_mdrft, per_level};
#define CALL(MM)                                                                                 \
  if (C.detC != 0.0)                                                                             \
    cyl_acc_launch<MM, true, true>(grid.x, grid.z, ctx->stream,                                  \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->mover_cnt, LC, f->d_Wnd.p, nullptr, 1, al); \
  else                                                                                           \
    cyl_acc_launch<MM, false, true>(grid.x, grid.z, ctx->stream,                                 \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->mover_cnt, LC, f->d_Wnd.p, nullptr, 1, al)
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
  } else if (nr) {
    ProfScope ps(ctx, "k_cyl_mstep_update");
    const unsigned spread = listed ? expamd_mover_spread((size_t)c->mover_hint) : 1u;
    const unsigned grid = cdiv(listed ? (size_t)c->mover_hint * spread : nr, 256);
    const uint32_t *lo_ = listed ? c->mover_cnt : c->lev_off.p, *li_ = listed ? c->mover_list.p : nullptr;
    if (f->generic)
      k_cyl_moments_gen<<<cdiv(listed ? (size_t)c->mover_hint : nr, 256), 256, 0, ctx->stream>>>(
          C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, c->newlev.p, lo_, first, ms, mfirst_mdrft,
          f->d_Wnd.p, 0, 0, nullptr, li_);
    else {
#define CALL(MM)                                                                              \
  k_cyl_mstep_update<MM><<<grid, 256, 0, ctx->stream>>>(                                      \
      C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, c->newlev.p,          \
      lo_, first, ms, mfirst_mdrft, f->d_Wnd.p, 0, nullptr, li_, spread)
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
    }
  }
  // (expcoefN += differ in the summing kernel itself when this rank is alone: no all-reduce in between)
  const bool alone = ctx->nranks <= 1 && !ctx->ar_fn;
  cyl_contract(ctx->stream, C, f->d_tab.p, f->d_Wnd.p + (size_t)mfirst_mdrft * wl, f->d_cpart.p,
               f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef_dev, nl, f->ncoef_dev, nullptr, /*clear=*/1,
               alone ? f->d_coefN.p + (size_t)mfirst_mdrft * f->ncoef_dev : nullptr, nullptr, thin_diff);
  HIP_TRY(ctx, hipGetLastError());
  if (alone) return EXP_AMD_OK;
  const size_t cnt = (size_t)nl * f->ncoef_dev;
  int rc = expamd_allreduce(ctx, f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef_dev, cnt);
  if (rc) return rc;
  k_cyl_add_inplace<<<cdiv(cnt, 256), 256, 0, ctx->stream>>>(
      f->d_coefN.p + (size_t)mfirst_mdrft * f->ncoef_dev,
      f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef_dev, cnt);
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

int CylForce::determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift,
                                     bool have_keys)
{
  CylForce *f = this;
  f->home = c;
  f->home_gone = false;
  const CylDev C = cdev_acc(f, c);
  {
    const int level = (f->multistep && c->sorted_for == f && c->nlevels == f->multistep + 1)
                          ? f->mlevel : -1;
    if (level >= 0) c->sparse_mask &= ~(1u << level); else c->sparse_mask = 0;   // this call cell-sorts what it touches
    int rc = sort(c, c->acc_live, AdvSpec::step(advance, dt_kick, dt_drift), level, have_keys);
    if (rc) return rc;
  }
  // ---- accumulate ----------------------------------------------------------------------------------
  double *dst = f->multistep ? f->d_coefN.p + (size_t)f->mlevel * f->ncoef_dev : f->d_coef.p;
  if (f->multistep)   // L <- N of this level (exputil/EmpCylSL.cc:1867 setup_accumulation swap)
    HIP_TRY(ctx, hipMemcpyAsync(f->d_coefL.p + (size_t)f->mlevel * f->ncoef_dev, dst,
                                f->ncoef_dev * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(f->d_Wn.p, 0, f->nnode * dev.ntrig * sizeof(double), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(dst + f->ncoef, 0, 2 * sizeof(double), ctx->stream));
  f->tails_clean = false;
  f->wn_clean = false;
  const int lo = f->multistep ? f->mlevel : 0, hi = lo;
  size_t nrange = c->n;      // population of the accumulated level: sizes the grid and the chunks
  if (c->n && f->multistep) {
    int rc = expamd_comp_level_count(c, lo, hi, &nrange);
    if (rc) return rc;
  }
  if (nrange) {
    ProfScope ps(ctx, "k_cyl_accumulate");
    // >= ~6 rounds of blocks (as in the spherical launcher); sparse levels pay one flush per cell
    // change, serial within a wave, so they get short chunks and many waves
    size_t chunk = (nrange / ((size_t)CACC_WAVES * 3072)) & ~(size_t)63;
    chunk = chunk < 64 ? 64 : chunk > CACC_CHUNK_MAX ? CACC_CHUNK_MAX : chunk;
    if (!f->multistep) chunk = CACC_CHUNK_MAX;
    const unsigned grid = cdiv(nrange, (size_t)CACC_WAVES * chunk);
    LevChunks LC;
    LC.lo = lo; LC.nlev = 1; LC.bstart[0] = 0; LC.bstart[1] = grid; LC.chunk[0] = (int)chunk;
    if (f->generic)        // (one moment buffer, whatever the level: plain = 2)
      k_cyl_moments_gen<<<cdiv(nrange, 256), 256, 0, ctx->stream>>>(
          C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), nullptr, nullptr, c->lev_off.p, lo, hi, 0, f->d_Wn.p, 2, 0,
          f->d_tailpart.p, nullptr);
    else {
#define CALL(MM)                                                                                 \
  if (C.detC != 0.0)                                                                             \
    cyl_acc_launch<MM, true>(grid, 1, ctx->stream,                                               \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, LC, f->d_Wn.p, f->d_tailpart.p, 0); \
  else                                                                                           \
    cyl_acc_launch<MM, false>(grid, 1, ctx->stream,                                              \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, LC, f->d_Wn.p, f->d_tailpart.p, 0)
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
    }
  }
  {
    ProfScope ps(ctx, "k_cyl_contract");
    f->cpart_clean = false;
    cyl_contract(ctx->stream, C, f->d_tab.p, f->d_Wn.p, f->d_cpart.p, dst, 1, 0, nullptr, 0, nullptr, f->d_tailpart.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  int rc = expamd_allreduce(ctx, dst, f->ncoef_dev);
  if (rc) return rc;
  // used / cylmass: the reference adds the (rank-reduced) in-cut count and mass of every level
  // accumulated while tnow == resetT, i.e. during the first sub-step of a master step
  // (src/Cylinder.cc:1088-1099); without multistep that is simply the last accumulation
  if (!f->multistep)
    k_cyl_mass<<<1, 64, 0, ctx->stream>>>(f->d_mass.p, dst + f->ncoef, 1);
  else if (f->mass_open)
    k_cyl_mass<<<1, 64, 0, ctx->stream>>>(f->d_mass.p, dst + f->ncoef, 0);
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

int CylForce::substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine, int phase)
{
  CylForce *f = this;
  const int ms = f->multistep;
  if (lo < 0 || lo > ms) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "substep_expansion: level out of range");
  f->home = c;
  f->home_gone = false;
  const int nact = ms - lo + 1;
  const AdvSpec adv = dt_min > 0.0 ? AdvSpec::levels(dt_min, ms, lo) : AdvSpec();
  int rc;
  // dense levels (cell-sorted) of the active suffix end at dmax; the levels above it are sparse
  int dmax = lo - 1;
  for (int L = lo; L <= ms; L++) if (!((c->sparse_mask >> L) & 1u)) dmax = L;
  // a sweep left its level changes to this sort (exp_amd_comp::partition_stale): the whole active range is
  // re-partitioned, sparse levels included (they are advanced by the sort then, not in place)
  const bool stale_ok = c->partition_stale && c->stale_for == (const void *)f && c->nlevels == ms + 1 && c->stale_lo >= lo;
  const bool ordered = (c->sorted_for == f && c->nlevels == ms + 1) || stale_ok;
  const bool full = lo == 0 || !ordered;
  if (full || stale_ok) dmax = ms;
  if (c->n && phase != 2) {
    uint32_t keep[66];
    const bool had = c->lev_host_valid && (ordered || c->partition_stale);
    // a commit left to this sort: it must be the sort of exactly the slots the sweep examined
    if (c->commit_pending && !(stale_ok && c->stale_lo == lo && dmax >= lo) && (rc = expamd_comp_flush_commit(c))) return rc;
    c->partition_stale = false;         // (the sort below settles it: the active range, or everything)
    if (had) for (int k = 0; k <= ms + 1; k++) keep[k] = c->lev_host[k];
    // (a closing half-kick still owed rides along with the pass that advances its levels)
    if (c->pending_kick != 0.0 && (!adv.mode || dmax >= lo) &&
        (rc = adv.mode ? expamd_comp_settle_pending(c, lo, dmax, true) : expamd_comp_settle_pending(c, 0, ms, false)))
      return rc;
    if (dmax >= lo) {
      // (sub-step 0 of a master step: the sweep that closed the last one wrote these keys, k_kick_adjust / kick_adjust.h)
      const bool keys_there = full && lo == 0 && adv.mode == 2 && expamd_comp_mprekey_ok(c, f, adv.dt_min);
      rc = sort(c, /*move_acc=*/full && lo > 0, adv, full ? -1 : lo, keys_there, dmax);
      if (rc) return rc;
    }
    c->commit_pending = false;          // (the scatter stored the proposed levels)
    if (had) {
      for (int k = 0; k <= ms + 1; k++) c->lev_host[k] = keep[k];
      c->lev_host_valid = true;
    }
    // (... by the thin accumulation kernel itself when that is what follows: thin_adv.h)
    if (dmax < ms && adv.mode) {
      size_t nall = 0;
      if ((rc = expamd_comp_level_count(c, lo, ms, &nall))) return rc;
      const bool fuse_on = EXPAMD_EXPT("EXP_AMD_THIN_ADVANCE", 1) != 0;
      const bool fuse = fuse_on && !f->frozen() && dmax < lo && adv.mode == 2 && nall > 0 && ctx->thin_max > 0 && (long long)nall <= ctx->thin_max * ctx->thin_acc_scale &&
                        !ctx->deterministic && !f->generic && f->thin_lds_ok();
      if (fuse) {
        f->adv_owed = true;
        f->adv_dt_min = dt_min;
      } else if ((rc = expamd_comp_advance_levels(c, dmax + 1, ms, dt_min, ms))) return rc;
    }
    // (a half-kick owed by levels of the advanced range that no pass took along: that range was empty)
    if (c->pending_kick != 0.0 && adv.mode && !f->adv_owed && lo <= c->pending_lo) { c->pending_kick = 0.0; c->pending_lo = 0; }
  }
  if (phase == 1) return EXP_AMD_OK;
  const CylDev C = cdev_acc(f, c);
  double *dst = f->d_coefN.p + (size_t)lo * f->ncoef_dev;
  const size_t wl = f->nnode * dev.ntrig;
  // the per-level moment buffers are left clean by the contraction that consumes them (below); only the plain
  // per-level API can have dirtied one
  if (!f->wn_clean) {
    HIP_TRY(ctx, hipMemsetAsync(f->d_Wn.p, 0, f->d_Wn.bytes(), ctx->stream));
    f->wn_clean = true;
  }
  // {in-cut mass, count} of the whole launch ride in the tail of the FIRST active level's set (the
  // other tails are zero): one number per sub-step is all Cylinder keeps (src/Cylinder.cc:1081-1099).
  // k_cyl_mass_take leaves the tail zero again; only the plain per-level API can have dirtied one.
  if (!f->tails_clean) {
    for (int L = 0; L <= ms; L++)
      HIP_TRY(ctx, hipMemsetAsync(f->d_coefN.p + (size_t)L * f->ncoef_dev + f->ncoef, 0, 2 * sizeof(double), ctx->stream));
    f->tails_clean = true;
  }
  int dacc = lo - 1;                    // last level the cell-ordered kernel takes
  for (int L = lo; L <= ms; L++) if (!((c->sparse_mask >> L) & 1u)) dacc = L;
  size_t nrange = 0;
  if (c->n && dacc >= lo && (rc = expamd_comp_level_count(c, lo, dacc, &nrange))) return rc;
  if (nrange) {
    ProfScope ps(ctx, "k_cyl_accumulate");
    // per level: >= ~6 rounds of blocks; sparse levels pay one flush per cell change, serial within
    // a wave, so they get short chunks and many waves.  Thickly and thinly populated levels go in SEPARATE
    // launches (consecutive levels of one kind together): mixed in one launch they took twice the time of the
    // two apart (measured for the sphere's twin, SphForce::substep_expansion).
    for (int L0 = lo; L0 <= dacc;) {
      auto pop = [&](int L) { return (size_t)c->lev_host[L + 1] - c->lev_host[L]; };
      const bool thick = pop(L0) >= CACC_THICK_MIN;
      int L1 = L0;
      while (L1 + 1 <= dacc && (pop(L1 + 1) >= CACC_THICK_MIN) == thick) L1++;
      LevChunks LC;
      LC.lo = L0; LC.nlev = L1 - L0 + 1;
      unsigned grid = 0;
      for (int L = L0; L <= L1; L++) {
        const size_t nl = pop(L);
        size_t chunk = (nl / ((size_t)CACC_WAVES * 3072)) & ~(size_t)63;
        chunk = chunk < 64 ? 64 : chunk > CACC_CHUNK_MAX ? CACC_CHUNK_MAX : chunk;
        LC.bstart[L - L0] = grid;
        LC.chunk[L - L0] = (int)chunk;
        grid += cdiv(nl, (size_t)CACC_WAVES * chunk);
      }
      LC.bstart[LC.nlev] = grid;
      if (grid && f->generic) {
        size_t np_ = 0;
        for (int L = L0; L <= L1; L++) np_ += pop(L);
        k_cyl_moments_gen<<<cdiv(np_, 256), 256, 0, ctx->stream>>>(
            C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, nullptr, c->lev_off.p, L0, L1, 0, f->d_Wn.p, 1,
            0, f->d_tailpart.p, nullptr);
      } else if (grid) {
#define CALL(MM)                                                                                 \
  if (C.detC != 0.0)                                                                             \
    cyl_acc_launch<MM, true>(grid, 1, ctx->stream,                                               \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, LC, f->d_Wn.p, f->d_tailpart.p, 1); \
  else                                                                                           \
    cyl_acc_launch<MM, false>(grid, 1, ctx->stream,                                              \
        C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, LC, f->d_Wn.p, f->d_tailpart.p, 1)
        MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
      }
      L0 = L1 + 1;
    }
  }
  nrange = 0;
  if (c->n && dacc < ms && (rc = expamd_comp_level_count(c, dacc + 1, ms, &nrange))) return rc;
  // the whole active range is sparse and thin: straight from the basis tables into the contraction's stage-1 sums
  // (k_cyl_acc_thin), no node moments and no pass over the nodes
  const bool thin = dacc < lo && (long long)nrange <= ctx->thin_max * ctx->thin_acc_scale && !ctx->deterministic && ctx->thin_max > 0 &&
                    (f->generic || f->thin_lds_ok());
  // (the advance that kernel was to perform, should it not run after all)
  if (f->adv_owed && !(thin && nrange)) {
    f->adv_owed = false;
    if ((rc = expamd_comp_advance_levels(c, lo, ms, f->adv_dt_min, ms))) return rc;
  }
  if (thin) {
    if ((rc = ensure_tabT())) return rc;
    if (!f->cpart_clean) {
      HIP_TRY(ctx, hipMemsetAsync(f->d_cpart.p, 0, f->d_cpart.bytes(), ctx->stream));
      f->cpart_clean = true;
    }
    if (nrange) {
      ProfScope ps(ctx, "k_cyl_acc_thin");
      ThinAdv tadv{};
      if (f->adv_owed) {
        f->adv_owed = false;
        double k0 = 0.0;
        int k0lo = 0;
        if ((rc = expamd_comp_take_pending(c, lo, ms, &k0, &k0lo))) return rc;
        tadv = ThinAdv{c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ),
                       c->level[c->cur].p, f->adv_dt_min, ms, k0, k0lo, 1};
      }
      if (f->generic)
        cyl_tile_acc_launch(ctx->stream, nrange, C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, lo, ms,
                            f->d_tabT.p, f->tabT_nk, f->d_cpart.p, f->d_tailpart.p);
      else {
#define CALL(MM)                                                                                         \
  cyl_thin_acc_launch<MM>(ctx->stream, nrange, C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, lo, ms, \
                          f->d_tabT.p, f->tabT_nk, f->d_cpart.p, f->d_tailpart.p, tadv)
      MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
      }
    }
  } else if (nrange) {
    ProfScope ps(ctx, "k_cyl_accumulate_sparse");
    const unsigned grid = cdiv(nrange, 256);
    if (f->generic)
      k_cyl_moments_gen<<<grid, 256, 0, ctx->stream>>>(
          C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, nullptr, c->lev_off.p, dacc + 1, ms, 0, f->d_Wn.p,
          1, 0, f->d_tailpart.p, nullptr);
    else {
#define CALL(MM)                                                                              \
  k_cyl_mstep_update<MM><<<grid, 256, 0, ctx->stream>>>(                                      \
      C, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, nullptr,              \
      c->lev_off.p, dacc + 1, ms, 0, f->d_Wn.p, 1, f->d_tailpart.p)
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
    }
  }
  {
    ProfScope ps(ctx, "k_cyl_contract");
    // ... with setup_accumulation(M)'s swap of every active level: L <- N, N <- new
    // (exputil/EmpCylSL.cc:2010-2030)
    if (mdrft_combine >= 0) {
      int mfc = 0;
      CombineW Wc;
      expamd_combine_weights(ms, mdrft_combine, &mfc, &Wc);
      if (!thin)
        k_cyl_contract_part<<<dim3(CYL_CSEG, C.ntrig, nact), 256, 0, ctx->stream>>>(C, f->d_tab.p, f->d_Wn.p + (size_t)lo * wl,
                                                                                   f->d_cpart.p, /*clear=*/1);
      // (the stage-1 sums are left zero by the kernel that reads them, table path or thin: no memset in between)
      k_cyl_sum_combine<<<cdiv(f->ncoef, 256), 256, 0, ctx->stream>>>(
          C, f->d_cpart.p, f->d_coefN.p, f->d_coefL.p, f->ncoef_dev, lo, nact, ms + 1, mfc, Wc, f->d_coef.p,
          f->d_tailpart.p, f->d_mass.p, f->mass_open ? 1 : 0, /*clear=*/1);
      HIP_TRY(ctx, hipGetLastError());
      f->combined_mdrft = mdrft_combine;
      f->proj_dirty = true;
      return EXP_AMD_OK;
    }
    cyl_contract(ctx->stream, C, f->d_tab.p, f->d_Wn.p + (size_t)lo * wl, f->d_cpart.p, dst, nact,
                 f->ncoef_dev, f->d_coefL.p + (size_t)lo * f->ncoef_dev, /*clear=*/1, nullptr, f->d_tailpart.p, thin);
  }
  HIP_TRY(ctx, hipGetLastError());
  f->combined_mdrft = -1;
  if ((rc = expamd_allreduce(ctx, dst, (size_t)nact * f->ncoef_dev))) return rc;
  k_cyl_mass_take<<<1, 64, 0, ctx->stream>>>(f->d_mass.p, dst + f->ncoef, f->mass_open ? 1 : 0);
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

int CylForce::accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick, double nk_dtk,
                         double nk_dtd, bool *prekey_done, bool defer_kick)
{
  CylForce *f = this;
  if (prekey_done) *prekey_done = false;
  // next step's keys: single level, own (sorted) particles, fused half-kick only
  const bool prekey = prekey_done && nk_dtd != 0.0 && dt_kick != 0.0 && !external &&
                      t->nlevels == 1 && f->multistep == 0 && t->sorted_for == f;
  // closing half-kick: stored (1), deferred (0), or stored with the next opening half-kick (2); see sph.hip
  const bool deferred = defer_kick && dt_kick != 0.0;
  const int sv = !deferred ? 1 : (prekey && nk_dtk != 0.0 && ctx->prekick) ? 2 : 0;
  // a thin target range (a block-multistep sub-step's few active particles, ours or another component's) is evaluated
  // straight from the coefficient set (k_cyl_force_thin): the projected node table is not needed and stays stale
  bool thin = false;
  size_t nthin = 0;
  if (f->multistep > 0 && t->n && t->nlevels > 1 && dt_kick == 0.0 && !prekey_done && !ctx->deterministic &&
      ctx->thin_max > 0) {
    int rc_ = expamd_comp_level_count(t, f->mlevel, t->nlevels - 1, &nthin);
    if (rc_) return rc_;
    thin = (long long)nthin <= ctx->thin_max;
  }
  if (thin) {
    int rc_ = ensure_tabT();
    if (rc_) return rc_;
    f->mass_open = false;
    if (nthin) {
      ProfScope ps(ctx, "k_cyl_force_thin");
      CylDev C = !external ? cdev_for(f, t) : f->home ? cdev_for(f, f->home)
                 : f->home_gone ? cdev_frame(f, f->home_center, f->home_use_rot, f->home_rot) : cdev_for(f, t);
      C.ps = t->pseudo;
      cdev_freeze(C, t);
      if (f->generic)
        cyl_wave_force_launch(ctx->stream, nthin, C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, f->mlevel, t->nlevels - 1,
                              f->d_tabT.p, f->tabT_nk, f->d_coef.p, f->d_mass.p, t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT),
                              t->a(A_VX), t->a(A_VY), t->a(A_VZ), assign ? 1 : 0);
      else {
#define CALL(MM)                                                                                          \
  cyl_thin_force_launch<MM>(ctx->stream, nthin, C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, f->mlevel,      \
                            t->nlevels - 1, f->d_tabT.p, f->tabT_nk, f->d_coef.p, f->d_mass.p, t->a(A_AX), t->a(A_AY), \
                            t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY), t->a(A_VZ), assign ? 1 : 0)
      MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
      }
      HIP_TRY(ctx, hipGetLastError());
    }
    t->acc_live = true;
    return EXP_AMD_OK;
  }
  if (f->proj_dirty) {
    ProfScope ps(ctx, "k_cyl_project");
    k_cyl_project<<<dim3(cdiv(f->nnode, 256), cfg.mmax + 1), 256, 0, ctx->stream>>>(
        f->dev, f->d_tab.p, f->d_coef.p, f->d_TF.p, f->tab_twin ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    f->proj_dirty = false;
  }
  f->mass_open = false;          // tnow has moved past resetT once forces are evaluated
  if (t->n == 0) return EXP_AMD_OK;
  // external target: positions go into the frame of the component the expansion was built from
  CylDev C = !external ? cdev_for(f, t) : f->home ? cdev_for(f, f->home)
             : f->home_gone ? cdev_frame(f, f->home_center, f->home_use_rot, f->home_rot) : cdev_for(f, t);
  C.ps = t->pseudo;
  cdev_freeze(C, t);
  const int lo = (t->nlevels > 1) ? f->mlevel : 0;
  const int hi = t->nlevels - 1;
  size_t nr = t->n;                    // population of the level range: sizes the launch
  if (t->nlevels > 1) { int rc_ = expamd_comp_level_count(t, lo, hi, &nr); if (rc_) return rc_; }
  if (nr == 0) { t->acc_live = true; return EXP_AMD_OK; }
  {
    ProfScope ps(ctx, "k_cyl_force");
    const unsigned grid = cdiv(nr, 256);
    if (f->generic)
      k_cyl_force_gen<<<grid, 256, 0, ctx->stream>>>(
          C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_TF.p, f->d_mass.p, t->a(A_AX), t->a(A_AY), t->a(A_AZ),
          t->a(A_POT), t->a(A_VX), t->a(A_VY), t->a(A_VZ), dt_kick, assign ? 1 : 0, prekey ? t->key.p : nullptr, nk_dtk,
          nk_dtd, sv);
    else {
      // main launch + tail launch (beyond 0.75 of the table radius); the work list holds one entry per wave at most
      const size_t need = t->n / 64 + 8;
      if (f->work_cap < need && !(external && f->work_cap > 0)) {      // (an external launch uses no list)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, f->d_work.alloc(CYL_WORK_STRIDE * need + 2));
        HIP_TRY(ctx, hipMemsetAsync(f->d_work.p + CYL_WORK_STRIDE * need, 0, 2 * sizeof(uint32_t), ctx->stream));
        f->work_cap = need;
        f->work_flip = 0;
      }
      uint32_t *cnt = f->d_work.p + CYL_WORK_STRIDE * f->work_cap;
      if (external) {
#define CALL(MM)                                                                                  \
  k_cyl_force<MM, true><<<grid, 256, 0, ctx->stream>>>(                                           \
      C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_TF.p, f->d_mass.p, \
      t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY), t->a(A_VZ),        \
      dt_kick, assign ? 1 : 0, prekey ? t->key.p : nullptr, nk_dtk, nk_dtd, sv, nullptr, nullptr, nullptr)
      MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
      } else {
#define CALL(MM)                                                                                  \
  k_cyl_force<MM, false><<<grid, 256, 0, ctx->stream>>>(                                          \
      C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_TF.p, f->d_mass.p, \
      t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY), t->a(A_VZ),        \
      dt_kick, assign ? 1 : 0, prekey ? t->key.p : nullptr, nk_dtk, nk_dtd,                        \
      sv, f->d_work.p, cnt + f->work_flip, nullptr);                                               \
  k_cyl_force<MM, true><<<grid, 256, 0, ctx->stream>>>(                                           \
      C, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_TF.p, f->d_mass.p, \
      t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY), t->a(A_VZ),        \
      dt_kick, assign ? 1 : 0, prekey ? t->key.p : nullptr, nk_dtk, nk_dtd,                        \
      sv, f->d_work.p, cnt + f->work_flip, cnt + (1 - f->work_flip))
    MMAX_DISPATCH(cfg.mmax, CALL)
#undef CALL
      f->work_flip ^= 1;
      }
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  t->acc_live = true;
  if (prekey) *prekey_done = true;
  if (deferred) t->pending_kick = sv == 2 ? -nk_dtk : dt_kick;
  return EXP_AMD_OK;
}

// The "mlim" key (src/Cylinder.cc:40, :225 -> EmpCylSL::set_mlim; pyEXP: expui/BiorthBasis.cc:1466, :1620): harmonics
// m > mlim take no part -- EmpCylSL::get_pot fills Vc / Vs up to min(MLIM, MMAX) only (exputil/EmpCylSL.cc:5602) and
// accumulated_eval / accumulated_dens_eval sum up to it (:5317, :5465).  Here the TABLES of m > mlim are zeroed on the
// device: every pass -- accumulation, differencing, projection, the direct kernels, fields -- then drops those harmonics
// with no change to a kernel, and the coefficients of m > mlim come out as zero (the reference leaves the rows of Vc it
// never fills as they were allocated, :4078: unspecified there, zero here).  Can only be lowered once set.
extern "C" int exp_amd_cyl_set_mlim(exp_amd_force *fb, int mlim)
{
  expamd_mutated();
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_mlim: not a cylinder force");
  if (mlim < 0) return expamd_fail(f->ctx, EXP_AMD_ERR_ARG, "set_mlim: mlim must be >= 0 (the key's -1 means: do not call)");
  const int M = f->cfg.mmax, N = f->cfg.nmax;
  if (mlim >= M) return EXP_AMD_OK;                                  // min(MLIM, MMAX)
  if (f->mlim >= 0 && mlim > f->mlim)
    return expamd_fail(f->ctx, EXP_AMD_ERR_STATE, "set_mlim: the tables above m = %d are already dropped", f->mlim);
  exp_amd_ctx *ctx = f->ctx;
  const size_t per_m = (size_t)N * f->nnode, per_kind = (size_t)(M + 1) * per_m;
  for (int k = 0; k < 6; k++)
    HIP_TRY(ctx, hipMemsetAsync(f->d_tab.p + k * per_kind + (size_t)(mlim + 1) * per_m, 0,
                                (size_t)(M - mlim) * per_m * sizeof(double), ctx->stream));
  if (f->d_dens.n)
    for (int k = 0; k < 2; k++)
      HIP_TRY(ctx, hipMemsetAsync(f->d_dens.p + k * per_kind + (size_t)(mlim + 1) * per_m, 0,
                                  (size_t)(M - mlim) * per_m * sizeof(double), ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  f->d_tabT.release();                        // (the node-major copy of the thin path is remade from the new tables)
  f->tabT_nk = 0;
  f->mlim = mlim;
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_cyl_get_cylmass(exp_amd_force *fb, double *mass)
{
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f || !mass) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "get_cylmass: not a cylinder force");
  HIP_TRY(f->ctx, hipMemcpyAsync(mass, f->d_mass.p, sizeof(double), hipMemcpyDeviceToHost,
                                 f->ctx->stream));
  HIP_TRY(f->ctx, hipStreamSynchronize(f->ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_cyl_set_cylmass(exp_amd_force *fb, double mass)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  CylForce *f = dynamic_cast<CylForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_cylmass: not a cylinder force");
  HIP_TRY(f->ctx, hipMemcpyAsync(f->d_mass.p, &mass, sizeof(double), hipMemcpyHostToDevice,
                                 f->ctx->stream));
  HIP_TRY(f->ctx, hipStreamSynchronize(f->ctx->stream));
  return EXP_AMD_OK;
}

