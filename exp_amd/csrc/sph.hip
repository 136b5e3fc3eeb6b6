// Spherical BFE force method (sphereSL): host side of the C ABI plus the small table kernels
// (sort key, moments -> coefficients, coefficients -> projected tables).  The per-particle
// kernels live in sph_kernels.h and are instantiated per LMAX in sph_inst.hip.
#include "sph_kernels.h"
#include "sort_kernels.h"
#include "force.h"

// ---- sort key -------------------------------------------------------------------------------------

// key = level * (numr-1) + radial cell of get_pot/get_force (r clamped to rmax like the force path)
struct SphKeyFn {
  static constexpr bool on = true;     // (k_kick_adjust: writes keys, kick_adjust.h)
  SphDev S;
  uint32_t sparse_mask;      // levels that are not cell-sorted: all their particles share bin 0
  __device__ __forceinline__ uint32_t operator()(double x, double y, double z, uint8_t lev) const
  {
    const uint32_t cell = ((sparse_mask >> lev) & 1u) ? 0u : sph_key_cell(S, x, y, z);
    return (uint32_t)lev * (uint32_t)(S.numr - 1) + cell + S.key_add;
  }
};

// ---- moments -> coefficients ------------------------------------------------------------------------
// part[seg][row][n] = sum_{i in seg} E[i][l][n] W[i][row][0] + E[i+1][l][n] W[i][row][1]
// (CSEG, the number of segments: sph_kernels.h)
// clear != 0: the moments are zeroed once they have been read (each (cell, row) pair is read by exactly
// one block), so that the next accumulation finds a clean buffer without a separate memset pass
__global__ void __launch_bounds__(256)
k_sph_contract(SphDev S, double *__restrict__ W, const double *__restrict__ wscale,
               double *__restrict__ part, int clear = 0)
{
  // (segment fastest: the nrows blocks that read the same lines of W share an XCD -- CSEG is a multiple of 8)
  const int seg = blockIdx.x, row = blockIdx.y;
  int l = 0;
  while ((l + 1) * (l + 1) <= row) l++;
  const int ncell = S.numr - 1;
  // blockIdx.z: one moment buffer / partial set per level (multi-level launches)
  W += (size_t)blockIdx.z * ncell * S.nrows * 2;
  part += (size_t)blockIdx.z * CSEG * S.nrows * S.nmax;
  const int per = (ncell + CSEG - 1) / CSEG;
  const int i0 = seg * per, i1 = min(ncell, i0 + per);
  const int stride = (S.lmax + 1) * S.nmax;
  // 32 radial orders x 8 cell slots per pass: the segment's cells are walked 8 at a time (a block per
  // (row, segment) of 64 threads, 24 of them busy for 63 dependent iterations, took 29 us -- the longest
  // of the fixed per-step kernels of a strong-scaled run); the slots are added in a fixed order
  __shared__ double red[8][32];
  const int nn = threadIdx.x & 31, slot = threadIdx.x >> 5;
  for (int n0 = 0; n0 < S.nmax; n0 += 32) {
    const int n = n0 + nn;
    double s = 0.0;
    if (n < S.nmax)
      // (eight cells' loads are issued before the first is used: the kernel is a chain of load latencies otherwise)
      for (int ib = i0 + slot; ib < i1; ib += 64) {
        double w1[8], w2[8], e1[8], e2[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int i = ib + 8 * j;
          const bool in = i < i1;
          w1[j] = in ? W[((size_t)i * S.nrows + row) * 2] : 0.0;
          w2[j] = in ? W[((size_t)i * S.nrows + row) * 2 + 1] : 0.0;
          e1[j] = in ? S.E[(size_t)i * stride + l * S.nmax + n] : 0.0;
          e2[j] = in ? S.E[(size_t)(i + 1) * stride + l * S.nmax + n] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          s = fma(e1[j], w1[j], s);
          s = fma(e2[j], w2[j], s);
        }
      }
    red[slot][nn] = s;
    __syncthreads();
    if (slot == 0 && n < S.nmax) {
      const double t = ((red[0][nn] + red[1][nn]) + (red[2][nn] + red[3][nn])) +
                       ((red[4][nn] + red[5][nn]) + (red[6][nn] + red[7][nn]));
      part[((size_t)seg * S.nrows + row) * S.nmax + n] = t * wscale[row];   // 1/s(l,m), see lc_s
    }
    __syncthreads();
  }
  if (clear) {
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) {
      W[((size_t)i * S.nrows + row) * 2] = 0.0;
      W[((size_t)i * S.nrows + row) * 2 + 1] = 0.0;
    }
  }
}

// last != nullptr: the N/L swap of determine_coefficients (src/SphericalBasis.cc:785-792) on the way,
// last <- coef (the previous set of this level), coef <- new
// add_to != nullptr: add_to += the new set as well (the differencing of a single rank: expcoefN += differ with no
// all-reduce in between)
// clear != 0: the partial sums are left zero behind (what the thin accumulation, which ADDS to them, starts from)
__global__ void __launch_bounds__(256)
k_sph_sum_parts(double *__restrict__ part, int ncoef, double *__restrict__ coef,
                double *__restrict__ last = nullptr, double *__restrict__ add_to = nullptr, int clear = 0)
{
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncoef) return;
  part += (size_t)blockIdx.y * CSEG * ncoef;      // blockIdx.y: level (multi-level launches)
  coef += (size_t)blockIdx.y * ncoef;
  double s = 0.0;
  for (int seg = 0; seg < CSEG; seg++) s += part[(size_t)seg * ncoef + k];
  if (clear) for (int seg = 0; seg < CSEG; seg++) part[(size_t)seg * ncoef + k] = 0.0;
  if (last) last[(size_t)blockIdx.y * ncoef + k] = coef[k];
  coef[k] = s;
  if (add_to) add_to[(size_t)blockIdx.y * ncoef + k] += s;
}

// The block-multistep sub-step's form for a rank that is alone: every active level's segment sums with the N/L swap
// (as above), THEN the combined set of compute_multistep_coefficients (src/SphericalBasis.cc:1252-1333) in the same
// thread -- the same operations in the same order as k_sph_sum_parts + k_mstep_combine, one launch less in the chain.
__global__ void __launch_bounds__(256)
k_sph_sum_combine(double *__restrict__ part, int ncoef, double *__restrict__ N, double *__restrict__ L, int lo,
                  int nact, int nlev, int mfirst, CombineW W, double *__restrict__ out, int clear = 0)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncoef) return;
  for (int j = 0; j < nact; j++) {
    double *p = part + (size_t)j * CSEG * ncoef;
    double s = 0.0;
    for (int seg = 0; seg < CSEG; seg++) s += p[(size_t)seg * ncoef + k];
    if (clear) for (int seg = 0; seg < CSEG; seg++) p[(size_t)seg * ncoef + k] = 0.0;
    const size_t o = (size_t)(lo + j) * ncoef + k;
    L[o] = N[o];
    N[o] = s;
  }
  out[k] = expamd_combine_one(L, N, (size_t)ncoef, nlev, mfirst, W.ab, (size_t)k);
}

// ---- coefficients -> projected tables -----------------------------------------------------------------
// G[i][row] = sum_n E[i][l][n] c[row][n]   (reference row order)
__global__ void __launch_bounds__(256)
k_sph_project(SphDev S, const double *__restrict__ coef, double *__restrict__ G)
{
  const int i = blockIdx.x;
  const int stride = (S.lmax + 1) * S.nmax;
  for (int row = threadIdx.x; row < S.nrows; row += 256) {
    const int l = sph_l_of_row(row);
    const double *e = S.E + (size_t)i * stride + l * S.nmax;
    const double *c = coef + (size_t)row * S.nmax;
    G[(size_t)i * S.nrows + row] = sph_G_row(e, c, S.nmax);
  }
}

// T4[cell][q][4] = {G0, D, Bq, Aq} (see sph_kernels.h); rowmap[q] = coefficient row feeding the
// m-major slot q.  rowmap carries the reference's EVEN_M row quirk (src/SphericalBasis.cc:1590-1596:
// the skipped odd-m terms do not advance moffset, so even m >= 2 read rows l*l + m-1, l*l + m).
__global__ void __launch_bounds__(256)
k_sph_project4(SphDev S, const double *__restrict__ G, const int *__restrict__ rowmap,
               const double *__restrict__ tscale, double *__restrict__ T4)
{
  const int cell = blockIdx.x;                 // 0 .. numr-2
  const int j = cell < 1 ? 1 : cell;
  for (int q = threadIdx.x; q < S.trows; q += 256) {
    const int row = rowmap[q];
    if (row < 0) {                             // (l,m) switched off by NO_L0/NO_L1/EVEN_L/EVEN_M/M0_only
      double *t = T4 + ((size_t)cell * S.trows + q) * 4;
      t[0] = t[1] = t[2] = t[3] = 0.0;
      continue;
    }
    const double g0 = G[(size_t)cell * S.nrows + row];
    const double g1 = G[(size_t)(cell + 1) * S.nrows + row];
    const double h0 = S.p0[j - 1] * G[(size_t)(j - 1) * S.nrows + row];
    const double h1 = S.p0[j] * G[(size_t)j * S.nrows + row];
    const double h2 = S.p0[j + 1] * G[(size_t)(j + 1) * S.nrows + row];
    double *t = T4 + ((size_t)cell * S.trows + q) * 4;
    // (tscale: 1/s(l,m) of the rescaled Legendre recurrence)
    sph_t4_entry(tscale[q], g0, g1, h0, h1, h2, t[0], t[1], t[2], t[3]);
  }
}

// Both steps in one launch: the block of a cell forms G at the three nodes its T4 rows need (the cell's two and the
// force stencil's: {cell-1, cell, cell+1}, {0, 1, 2} for cell 0) into LDS -- the same sums, sph_G_row -- and builds the
// rows from there; G itself (field evaluation, sph_fields.hip) is written on the way, node `cell` by block `cell`, the
// last node by the last block.  gs: 3 nrows doubles of dynamic LDS.
__global__ void __launch_bounds__(256)
k_sph_project_both(SphDev S, const double *__restrict__ coef, const int *__restrict__ rowmap,
                   const double *__restrict__ tscale, double *__restrict__ G, double *__restrict__ T4)
{
  extern __shared__ __attribute__((aligned(16))) double gs[];
  const int cell = blockIdx.x;                 // 0 .. numr-2
  const int j = cell < 1 ? 1 : cell, b = j - 1;
  const int stride = (S.lmax + 1) * S.nmax;
  for (int it = threadIdx.x; it < 3 * S.nrows; it += 256) {
    const int k = it / S.nrows, row = it - k * S.nrows, node = b + k;
    const int l = sph_l_of_row(row);
    const double g = sph_G_row(S.E + (size_t)node * stride + l * S.nmax, coef + (size_t)row * S.nmax, S.nmax);
    gs[it] = g;
    if (node == cell || (cell == S.numr - 2 && node == S.numr - 1)) G[(size_t)node * S.nrows + row] = g;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < S.trows; q += 256) {
    const int row = rowmap[q];
    double *t = T4 + ((size_t)cell * S.trows + q) * 4;
    if (row < 0) { t[0] = t[1] = t[2] = t[3] = 0.0; continue; }
    const double g0 = gs[(cell - b) * S.nrows + row], g1 = gs[(cell + 1 - b) * S.nrows + row];
    const double h0 = S.p0[j - 1] * gs[row], h1 = S.p0[j] * gs[S.nrows + row], h2 = S.p0[j + 1] * gs[2 * S.nrows + row];
    sph_t4_entry(tscale[q], g0, g1, h0, h1, h2, t[0], t[1], t[2], t[3]);
  }
}

// ---- host side -----------------------------------------------------------------------------------------------

#include "sph_force.h"
#include "kick_adjust.h"

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
  if (cfg->lmax < 0 || cfg->lmax > SPH_GEN_MAX_L)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_create: lmax=%d outside [0,%d]", cfg->lmax,
                       SPH_GEN_MAX_L);
  if (cfg->nmax < 1 || cfg->numr < 3 || cfg->cmap < 0 || cfg->cmap > 2 || cfg->multistep < 0 ||
      cfg->multistep > 16)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_create: bad nmax/numr/cmap/multistep");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  SphForce *f = new SphForce;
  f->ctx = ctx;
  f->cfg = *cfg;
  const int L = cfg->lmax, nmax = cfg->nmax, numr = cfg->numr;
  const int nrows = (L + 1) * (L + 1);
  const size_t ncoef = (size_t)nrows * nmax;

  // E[i][l][n] = ef_l(n,i)/sqrt(ev_l[n])
  std::vector<double> E((size_t)numr * (L + 1) * nmax);
  for (int l = 0; l <= L; l++)
    for (int n = 0; n < nmax; n++) {
      const double s = sqrt(ev[l * nmax + n]);
      const double *src = ef + ((size_t)l * nmax + n) * numr;
      for (int i = 0; i < numr; i++) E[((size_t)i * (L + 1) + l) * nmax + n] = src[i] / s;
    }
  // factorial(l,m) of src/SphericalBasis.cc:328-335, folded into the Legendre recurrence
  // constants {A, B, C, e} (sph_kernels.h)
  auto fct = [](int l, int m) {
    long double v = sqrtl((2.0L * l + 1.0L) / (4.0L * M_PIl) * (long double)factrl(l - m) /
                          (long double)factrl(l + m));
    if (m) v *= sqrtl(2.0L);
    return v;
  };
  std::vector<double> lcv((size_t)(L + 1) * (L + 1) * 4, 0.0);
  for (int l = 0; l <= L; l++)
    for (int m = 0; m <= l; m++) {
      double *q = &lcv[((size_t)l * (L + 1) + m) * 4];
      if (l > m) {
        q[0] = (double)(fct(l, m) / fct(l - 1, m) * (2.0L * l - 1.0L) / (long double)(l - m));
        q[2] = (double)((long double)(l + m) * fct(l, m) / fct(l - 1, m));
      }
      if (l > m + 1)
        q[1] = (double)(fct(l, m) / fct(l - 2, m) * (long double)(l + m - 1) / (long double)(l - m));
      if (l == m)
        q[3] = (m == 0) ? (double)fct(0, 0)
                        : (double)(-(2.0L * m - 1.0L) * fct(m, m) / fct(m - 1, m - 1));
    }
  // m-major slot -> coefficient row (with the EVEN_M quirk of the reference)
  std::vector<int> rowmap(t4_rows(L), -1);     // pad row (if any) stays -1 -> zeros
  std::vector<double> tscale(t4_rows(L), 0.0), wscale(nrows, 1.0);
  for (int m = 0; m <= L; m++)
    for (int l = m; l <= L; l++) {
      const int q = t4_row(L, l, m);
      int rc = row_of(l, m, 0);
      if (cfg->EVEN_M && m > 0) rc = l * l + (m - 1);
      // terms the reference skips (src/SphericalBasis.cc:1575-1596) get zero table rows
      bool on = true;
      if (l == 0 && cfg->NO_L0) on = false;
      if (l == 1 && cfg->NO_L1) on = false;
      if (l > 0 && cfg->EVEN_L && (l & 1)) on = false;
      if (cfg->EVEN_M && (m & 1)) on = false;
      if (cfg->M0_only && m != 0) on = false;
      rowmap[q] = on ? rc : -1;
      tscale[q] = 1.0 / lc_s(l, m);
      wscale[row_of(l, m, 0)] = tscale[q];
      if (m > 0) wscale[row_of(l, m, 1)] = tscale[q];
      if (m > 0) {
        rowmap[q + 1] = on ? rc + 1 : -1;
        tscale[q + 1] = tscale[q];
      }
    }

  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(f->d_xi.alloc(numr));
  A(f->d_p0.alloc(numr));
  A(f->d_E.alloc(E.size()));
  A(f->d_lc.alloc(lcv.size()));
  A(f->d_ev.alloc((size_t)(L + 1) * nmax));
  A(f->d_litef.alloc((size_t)6 * (L + 1) * nmax));
  // the rescaled recurrence as data, for the any-order kernels (sph_gen.hip): the same constexpr functions whose
  // values the unrolled kernels fold into literals
  std::vector<double> gen_ac((size_t)(L + 1) * (L + 1) * 2, 0.0), gen_e(L + 1, 0.0);
  for (int m = 0; m <= L; m++) {
    gen_e[m] = lc_E(m);
    for (int l = m + 1; l <= L; l++) {
      gen_ac[((size_t)l * (L + 1) + m) * 2] = lc_a(l, m);
      gen_ac[((size_t)l * (L + 1) + m) * 2 + 1] = lc_c(l, m);
    }
  }
  std::vector<unsigned char> gen_slot((size_t)t4_rows(L) * 2, 0);
  for (int m = 0; m <= L; m++)
    for (int l = m; l <= L; l++) {
      const int q = t4_row(L, l, m);
      gen_slot[2 * q] = (unsigned char)l; gen_slot[2 * q + 1] = (unsigned char)m;
      if (m > 0) { gen_slot[2 * q + 2] = (unsigned char)l; gen_slot[2 * q + 3] = (unsigned char)(m | 0x80); }
    }
  A(f->d_gen_slot.alloc(gen_slot.size()));
  A(f->d_gen_ac.alloc(gen_ac.size()));
  A(f->d_gen_e.alloc(gen_e.size()));
  f->generic = L > SPH_MAX_L;
  if (const char *eg = getenv("EXP_AMD_SPH_GENERIC")) if (atoi(eg) != 0) f->generic = true;
  A(f->d_rowmap.alloc(rowmap.size()));
  A(f->d_tscale.alloc(tscale.size()));
  A(f->d_wscale.alloc(wscale.size()));
  // one moment buffer and one set of contraction partials per level (substep_expansion)
  A(f->d_W.alloc((size_t)(cfg->multistep + 1) * (numr - 1) * nrows * 2));
  A(f->d_part.alloc((size_t)(cfg->multistep + 1) * CSEG * ncoef));
  A(f->d_G.alloc((size_t)numr * nrows));
  A(f->d_T4.alloc((size_t)(numr - 1) * t4_rows(L) * 4));
  if (e == hipSuccess && f->alloc_common(ncoef, cfg->multistep) != EXP_AMD_OK) e = hipErrorOutOfMemory;
  if (e != hipSuccess) {
    exp_amd_force_destroy(f);
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_create: hipMalloc failed: %s",
                       hipGetErrorString(e));
  }
  HIP_TRY(ctx, hipMemcpy(f->d_xi.p, xi, numr * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_p0.p, p0, numr * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_E.p, E.data(), E.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_lc.p, lcv.data(), lcv.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_ev.p, ev, (size_t)(L + 1) * nmax * sizeof(double), hipMemcpyHostToDevice));
  {
    // lit_ef[edge][k][l][n] = ef_l(n, node): nodes 0, 1, 2 (edge 0) and numr-3, numr-2, numr-1 (edge 1), undivided
    std::vector<double> le((size_t)6 * (L + 1) * nmax);
    for (int edge = 0; edge < 2; edge++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l <= L; l++)
          for (int n = 0; n < nmax; n++)
            le[(((size_t)edge * 3 + k) * (L + 1) + l) * nmax + n] =
                ef[((size_t)l * nmax + n) * numr + (edge ? numr - 3 + k : k)];
    HIP_TRY(ctx, hipMemcpy(f->d_litef.p, le.data(), le.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, hipMemcpy(f->d_gen_slot.p, gen_slot.data(), gen_slot.size(), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_gen_ac.p, gen_ac.data(), gen_ac.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_gen_e.p, gen_e.data(), gen_e.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_rowmap.p, rowmap.data(), rowmap.size() * sizeof(int),
                         hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_tscale.p, tscale.data(), tscale.size() * sizeof(double),
                         hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(f->d_wscale.p, wscale.data(), wscale.size() * sizeof(double),
                         hipMemcpyHostToDevice));

  SphDev &S = f->dev;
  S.lmax = L; S.nmax = nmax; S.numr = numr; S.cmap = cfg->cmap; S.nrows = nrows; S.trows = t4_rows(L);
  S.rmap = cfg->rmap; S.scale = cfg->scale; S.rmin = cfg->rmin; S.rmax = cfg->rmax;
  S.xmin = cfg->xmin; S.dxi = cfg->dxi;
  S.inv_dxi = 1.0 / cfg->dxi; S.inv_scale = 1.0 / cfg->scale; S.inv_rmap = 1.0 / cfg->rmap;
  S.cx = S.cy = S.cz = 0.0;
  S.NO_L0 = cfg->NO_L0; S.NO_L1 = cfg->NO_L1; S.EVEN_L = cfg->EVEN_L; S.EVEN_M = cfg->EVEN_M;
  S.M0_only = cfg->M0_only;
  S.M0_acc = cfg->M0_only;
  S.no_exterior = 0;
  S.dsmall = DSMALL;
  S.xi_uniform = 1;                            // same two roundings as the device's mul_then_add
  for (int i = 0; i < numr && S.xi_uniform; i++) {
    volatile double t = cfg->dxi * (double)i;
    if (xi[i] != cfg->xmin + t) S.xi_uniform = 0;
  }
  S.xi = f->d_xi.p; S.p0 = f->d_p0.p; S.E = f->d_E.p; S.lc = f->d_lc.p;
  S.gen_ac = f->d_gen_ac.p; S.gen_e = f->d_gen_e.p; S.gen_slot = f->d_gen_slot.p;
  // the literal radial derivative (sph_dp_lit): more than four cells outside the first / last force stencil -- which
  // only the logarithmic map can reach (cmap 1 maps r -> 0 to within three cells of xmin; an unmapped grid is left alone)
  S.lit_ef = f->d_litef.p; S.lit_ev = f->d_ev.p; S.lit_coef = f->d_coef.p;
  S.lit_rowmap = f->d_rowmap.p; S.lit_tscale = f->d_tscale.p;
  const bool lit_on = cfg->cmap == 2 && !getenv("EXP_AMD_NO_LITERAL");
  f->lit_on = lit_on;
  S.lit_list = nullptr;             // (sized to the target by accelerate())
  S.lit_cap = 0;
  S.lit_lo = lit_on ? -4.0 : -1.0e300;
  S.lit_hi = lit_on ? 5.0 : 1.0e300;
  S.lit_xlo = lit_on ? xi[1] - 4.0 * cfg->dxi : -1.0e300;
  S.lit_xhi = lit_on ? xi[numr - 2] + 5.0 * cfg->dxi : 1.0e300;
  S.detC = 0.0;
  S.umass = 0.0;
  S.fac0 = -4.0 * M_PI;
  S.frz = nullptr;
  {
    // bound of a unit-mass particle's moment contribution |4 pi P0 x_k Ph(l,m)(cos theta) trig|, for the
    // rounding grid of the deterministic mode: max |p0| on the grid x max |Ph| on a fine cos(theta) grid
    double p0max = 0.0, phmax = 0.0;
    for (int i = 0; i < numr; i++) p0max = fmax(p0max, fabs(p0[i]));
    std::vector<double> ph((size_t)(L + 1) * (L + 1), 0.0);
    for (int k = 0; k <= 4000; k++) {
      const double x = -1.0 + k / 2000.0, sx = sqrt(fmax(0.0, (1.0 - x) * (1.0 + x)));
      double pmm = lc_E(0);
      for (int m = 0; m <= L; m++) {
        if (m > 0) pmm *= lc_E(m) * sx;
        double pl2 = 0.0, pl1 = 0.0;
        for (int l = m; l <= L; l++) {
          const double plm = (l == m) ? pmm : (l == m + 1) ? lc_a(l, m) * x * pl1 : lc_a(l, m) * x * pl1 - pl2;
          pl2 = pl1;
          pl1 = plm;
          phmax = fmax(phmax, fabs(plm));
        }
      }
    }
    f->term_max = 4.0 * M_PI * p0max * phmax * 2.0;
  }
  *out = f;
  return EXP_AMD_OK;
}

void SphForce::release()
{
  d_xi.release(); d_p0.release(); d_E.release(); d_lc.release(); d_litef.release(); d_litlist.release();
  d_gen_ac.release(); d_gen_e.release(); d_gen_slot.release();
  d_c0.release();
  d_rowmap.release();
  d_tscale.release();
  d_ev.release(); d_d0.release(); d_Gd.release();
  d_wscale.release();
  expamd_sph_cov_release(this);
  d_W.release(); d_coef_app.release(); d_part.release(); d_G.release(); d_T4.release(); d_work.release(); d_xwork.release();
  d_Wd.release(); d_differ.release();
  for (auto &b : d_ss) b.release();
  d_ss_prefix.release();
  if (ss_comp) { exp_amd_comp_destroy(ss_comp); ss_comp = nullptr; }
}

static SphDev dev_for(const SphForce *f, const double center[3])
{
  SphDev S = f->dev;
  S.cx = center[0]; S.cy = center[1]; S.cz = center[2];
  return S;
}

// Component::freeze of the component whose particles a launch walks (the source of an accumulation, the target of a force)
static void dev_freeze(SphDev &S, const exp_amd_comp *c)
{
  S.frz = expamd_comp_frz(c);
}

// ... for the passes that ADD particle contributions: with the deterministic mode on, the rounding
// grid that keeps every partial sum of this component exact (common.h: expamd_det_constant)
static SphDev dev_acc(const SphForce *f, const exp_amd_comp *c)
{
  SphDev S = dev_for(f, c->center);
  S.detC = expamd_det_constant(f->ctx->deterministic, c->mass_abs_sum * f->term_max);
  S.umass = c->uniform_mass ? c->mass_value : 0.0;
  S.fac0 = -4.0 * M_PI * f->mass_scale;
  dev_freeze(S, c);
  return S;
}

// (level, radial cell) order for this force's tables; with `advance` the kick dt_kick and drift
// dt_drift of the leapfrog are applied on the way (src/step.cc:279-288)
static int sph_sort(SphForce *f, exp_amd_comp *c, bool move_acc, const AdvSpec &adv = AdvSpec(),
                    int level = -1, bool have_keys = false, int level_hi = -1)
{
  exp_amd_ctx *ctx = f->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  c->nlevels = f->cfg.multistep + 1;
  const uint32_t ncell = (uint32_t)(f->cfg.numr - 1);
  const uint32_t nkeys = ncell * (uint32_t)c->nlevels;
  int rc = expamd_comp_prepare_hist(c, nkeys);
  if (rc) return rc;
  if (have_keys) {
    // c->key was filled by the previous step's force pass for exactly this advance
    // (exp_amd_step_kdk checks that): pass 1 only counts the 4-byte keys
    ProfScope ps(ctx, "k_hist_keys");
    // (a block-multistep run: the keys the closing sweep left are full (level, cell) keys, those of the levels that are
    // not cell-sorted collapse to the level's first bin here)
    // (a dense one-level store: a tile's keys are two or three neighbouring cells -- the short LDS window, sort_kernels.h)
    // (EXP_AMD_SORT_DENSE in an experimental build: 0 = the full window everywhere; 1 = one-level stores only, the default; 2 = the
    // full sort of a block-multistep store as well: measured on config 4 at +0.25 ms per master step, profiles/r05_sort_win_ab.txt)
    const long long dense_mode = EXPAMD_EXPT("EXP_AMD_SORT_DENSE", 1);
    c->sort_win = (dense_mode != 0 && (c->nlevels == 1 || dense_mode >= 2) && level < 0 && c->n >= (size_t)SORT_DENSE_MIN * ncell) ? SORT_WIN_DENSE : 0;
    k_hist_keys<<<cdiv(c->n, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(c->key.p, c->n, c->hist.p,
                                                                    f->cfg.multistep ? c->sparse_mask : 0u, ncell,
                                                                    c->sort_win ? c->sort_win : (uint32_t)SORT_WIN);
  } else {
    size_t nr = c->n;          // a level range is sized for its own population
    if (level >= 0 && (rc = expamd_comp_level_count(c, level, level_hi > level ? level_hi : level, &nr))) return rc;
    if (nr == 0) return EXP_AMD_OK;
    ProfScope ps(ctx, "k_key_hist");
    SphKeyFn kf{dev_for(f, c->center), c->sparse_mask};
    AdvanceArgs A = expamd_advance_args(c, adv);
    if (nr <= HIST_SHORT_MAX)
      k_key_hist<SphKeyFn, HIST_ITEMS_SHORT><<<cdiv(nr, SORT_TPB * HIST_ITEMS_SHORT), SORT_TPB, 0, ctx->stream>>>(
          kf, A, expamd_sort_range(c, level, level_hi), c->key.p, c->hist.p);
    else
    k_key_hist<SphKeyFn><<<cdiv(nr, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(
        kf, A, expamd_sort_range(c, level, level_hi), c->key.p, c->hist.p);
  }
  rc = expamd_comp_finish_sort(c, nkeys, ncell, move_acc, adv, level, level_hi);
  if (rc) return rc;
  c->sorted_for = f;
  return EXP_AMD_OK;
}

#define DECL_L(k)                                        \
  void expamd_sph_acc_L##k(const SphAccArgs &);          \
  void expamd_sph_force_L##k(const SphForceArgs &);  \
  void expamd_sph_upd_L##k(const SphUpdArgs &);         \
  void expamd_sph_thin_force_L##k(const SphThinForceArgs &); \
  void expamd_sph_thin_acc_L##k(const SphThinAccArgs &); \
  void expamd_sph_thin_diff_L##k(const SphThinDiffArgs &);
DECL_L(0) DECL_L(1) DECL_L(2) DECL_L(3) DECL_L(4) DECL_L(5) DECL_L(6)
DECL_L(7) DECL_L(8) DECL_L(9) DECL_L(10) DECL_L(11) DECL_L(12)
#undef DECL_L
void expamd_sph_acc_gen(const SphAccArgs &);          // any order (sph_gen.hip)
void expamd_sph_upd_gen(const SphUpdArgs &);
void expamd_sph_force_gen(const SphForceArgs &);
void expamd_sph_thin_force_gen(const SphThinForceArgs &);   // one wave per particle (k_sph_force_wave)
void expamd_sph_thin_acc_gen(const SphThinAccArgs &);        // 64-particle tiles (k_sph_acc_tile)
static const sph_acc_launcher k_acc_launch_tab[SPH_MAX_L + 1] = {
    expamd_sph_acc_L0, expamd_sph_acc_L1, expamd_sph_acc_L2,  expamd_sph_acc_L3,  expamd_sph_acc_L4,
    expamd_sph_acc_L5, expamd_sph_acc_L6, expamd_sph_acc_L7,  expamd_sph_acc_L8,  expamd_sph_acc_L9,
    expamd_sph_acc_L10, expamd_sph_acc_L11, expamd_sph_acc_L12};
static const sph_upd_launcher k_upd_launch_tab[SPH_MAX_L + 1] = {
    expamd_sph_upd_L0, expamd_sph_upd_L1, expamd_sph_upd_L2,  expamd_sph_upd_L3,  expamd_sph_upd_L4,
    expamd_sph_upd_L5, expamd_sph_upd_L6, expamd_sph_upd_L7,  expamd_sph_upd_L8,  expamd_sph_upd_L9,
    expamd_sph_upd_L10, expamd_sph_upd_L11, expamd_sph_upd_L12};
static const sph_thin_force_launcher k_thin_force_launch[SPH_MAX_L + 1] = {
    expamd_sph_thin_force_L0, expamd_sph_thin_force_L1, expamd_sph_thin_force_L2,  expamd_sph_thin_force_L3,
    expamd_sph_thin_force_L4, expamd_sph_thin_force_L5, expamd_sph_thin_force_L6,  expamd_sph_thin_force_L7,
    expamd_sph_thin_force_L8, expamd_sph_thin_force_L9, expamd_sph_thin_force_L10, expamd_sph_thin_force_L11,
    expamd_sph_thin_force_L12};
static const sph_thin_diff_launcher k_thin_diff_launch[SPH_MAX_L + 1] = {
    expamd_sph_thin_diff_L0, expamd_sph_thin_diff_L1, expamd_sph_thin_diff_L2,  expamd_sph_thin_diff_L3,
    expamd_sph_thin_diff_L4, expamd_sph_thin_diff_L5, expamd_sph_thin_diff_L6,  expamd_sph_thin_diff_L7,
    expamd_sph_thin_diff_L8, expamd_sph_thin_diff_L9, expamd_sph_thin_diff_L10, expamd_sph_thin_diff_L11,
    expamd_sph_thin_diff_L12};
static const sph_thin_acc_launcher k_thin_acc_launch[SPH_MAX_L + 1] = {
    expamd_sph_thin_acc_L0, expamd_sph_thin_acc_L1, expamd_sph_thin_acc_L2,  expamd_sph_thin_acc_L3,
    expamd_sph_thin_acc_L4, expamd_sph_thin_acc_L5, expamd_sph_thin_acc_L6,  expamd_sph_thin_acc_L7,
    expamd_sph_thin_acc_L8, expamd_sph_thin_acc_L9, expamd_sph_thin_acc_L10, expamd_sph_thin_acc_L11,
    expamd_sph_thin_acc_L12};
static const sph_force_launcher k_force_launch_tab[SPH_MAX_L + 1] = {
    expamd_sph_force_L0, expamd_sph_force_L1, expamd_sph_force_L2,  expamd_sph_force_L3,
    expamd_sph_force_L4, expamd_sph_force_L5, expamd_sph_force_L6,  expamd_sph_force_L7,
    expamd_sph_force_L8, expamd_sph_force_L9, expamd_sph_force_L10, expamd_sph_force_L11,
    expamd_sph_force_L12};

// the unrolled kernels of this order, or the any-order ones
static void sph_launch_acc(const SphForce *f, const SphAccArgs &a)
{
  if (f->generic) expamd_sph_acc_gen(a); else k_acc_launch_tab[f->cfg.lmax](a);
}
static void sph_launch_upd(const SphForce *f, const SphUpdArgs &a)
{
  if (f->generic) expamd_sph_upd_gen(a); else k_upd_launch_tab[f->cfg.lmax](a);
}
static void sph_launch_force(const SphForce *f, const SphForceArgs &a)
{
  if (f->generic) expamd_sph_force_gen(a); else k_force_launch_tab[f->cfg.lmax](a);
}

// staging buffers of the per-particle atomic path (k_sph_mstep_update<L, true> + k_mstep_apply) for up to
// ctx->stage_max particles; beyond that the launch keeps its own atomics
static int sph_stage(SphForce *f, size_t np, SphUpdArgs &a)
{
  exp_amd_ctx *ctx = f->ctx;
  if (np == 0 || (long long)np > ctx->stage_max) return EXP_AMD_OK;
  const size_t nval = (size_t)f->dev.nrows * 2;
  if (f->d_stage.n < np * nval) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    f->d_stage.release(); f->d_stage_keys.release();
    const size_t cap = np < 4096 ? 4096 : np + np / 4;
    HIP_TRY(ctx, f->d_stage.alloc(cap * nval));
    HIP_TRY(ctx, f->d_stage_keys.alloc(cap * 2));
  }
  a.stage = f->d_stage.p;
  a.keys = reinterpret_cast<int2 *>(f->d_stage_keys.p);
  return EXP_AMD_OK;
}

#define ACC_THICK_MIN 1000000u       // level population from which a multistep level is accumulated apart from thinner ones

static int sph_accumulate(SphForce *f, exp_amd_comp *c, double *d_out, const uint32_t *range = nullptr, size_t nslots = 0)
{
  exp_amd_ctx *ctx = f->ctx;
  const int lo = f->multistep ? f->mlevel : 0;
  const int hi = f->multistep ? f->mlevel : 0;
  SphDev S = dev_acc(f, c);
  HIP_TRY(ctx, hipMemsetAsync(f->d_W.p, 0, (size_t)(f->cfg.numr - 1) * S.nrows * 2 * sizeof(double), ctx->stream));
  f->w_clean = false;
  // used: multistep = 0 counts the last accumulation; a multistep force adds up the levels of the
  // first sub-step (see SphForce::used_open)
  unsigned long long *used_p = f->d_used.p + ((f->multistep && !f->used_open) ? 1 : 0);
  if (!(f->multistep && f->used_open))
    HIP_TRY(ctx, hipMemsetAsync(used_p, 0, sizeof(unsigned long long), ctx->stream));
  size_t nrange = range ? nslots : c->n;      // particles of the level(s) accumulated: sizes the grid and the chunks
  if (c->n && f->multistep) {
    int rc = expamd_comp_level_count(c, lo, hi, &nrange);
    if (rc) return rc;
  }
  if (nrange) {
    ProfScope ps(ctx, "k_sph_accumulate");
    // (range: an appended store -- its slots [range[0], range[1]), empty ones among them: particles.h)
    SphAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), range ? range : c->lev_off.p, lo, hi,
                 f->d_W.p, used_p, nrange, ctx->stream, f->multistep ? 1 : 0};
    sph_launch_acc(f, a);
  }
  {
    ProfScope ps(ctx, "k_sph_contract");
    k_sph_contract<<<dim3(CSEG, S.nrows), 256, 0, ctx->stream>>>(S, f->d_W.p, f->d_wscale.p,
                                                                f->d_part.p);
    f->part_clean = false;
    k_sph_sum_parts<<<cdiv(f->ncoef, 256), 256, 0, ctx->stream>>>(f->d_part.p, (int)f->ncoef,
                                                                  d_out);
  }
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// ---- "ssfrac": coefficients from a sub-sample of the particles (sph_force.h) -------------------------------------------
// One thread per slot: is the particle's place j in the level list (its caller index) inside its thread's shortened slice?
// Then it goes to place prefix[thread] + (j - slice begin) of the compacted set -- no atomics, the same set every time.
__global__ void __launch_bounds__(256)
k_subset_gather(const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                const double *__restrict__ M, double umass, const uint32_t *__restrict__ id, size_t n, unsigned nthrds,
                double ssfrac, const uint32_t *__restrict__ prefix, double *__restrict__ om, double *__restrict__ ox,
                double *__restrict__ oy, double *__restrict__ oz)
{
  const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= n) return;
  const unsigned long long j = id[slot], T = nthrds;
  const unsigned long long k = ((j + 1) * T + n - 1) / n - 1;        // nbeg(k) <= j < nbeg(k + 1), nbeg(k) = n k / T
  const unsigned long long nbeg = n * k / T, nend = n * (k + 1) / T;
  const long long lim = (long long)floor(ssfrac * (double)nend);     // `nend = (int)floor(ssfrac*nend)`, :460
  if ((long long)j >= lim) return;
  const size_t o = (size_t)prefix[k] + (size_t)(j - nbeg);
  om[o] = umass != 0.0 ? umass : M[slot];
  ox[o] = X[slot]; oy[o] = Y[slot]; oz[o] = Z[slot];
}

int SphForce::determine_coefficients_subset(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift)
{
  SphForce *f = this;
  int rc;
  if (f->cfg.multistep)
    return expamd_fail(ctx, EXP_AMD_ERR_STATE, "ssfrac with block multistep: the sub-sample is a slice of every LEVEL list, "
                       "whose order in the reference is the history of its level changes (src/SphericalBasis.cc:435-460)");
  if (c->n >= 0x7fffffffu) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "ssfrac: more than 2^31 particles on one rank");
  // the advance a fused step would have folded into the sort: on its own here (the store is not reordered)
  if (advance && ((rc = exp_amd_comp_kick(c, dt_kick, -1)) || (rc = exp_amd_comp_drift(c, dt_drift, -1)))) return rc;
  if ((rc = expamd_comp_touch(c))) return rc;
  const size_t n = c->n;
  const unsigned T = (unsigned)f->ss_nthrds;
  std::vector<uint32_t> prefix(T, 0u);
  size_t nsub = 0;
  for (unsigned k = 0; k < T; k++) {
    const long long nbeg = (long long)((unsigned long long)n * k / T), nend = (long long)((unsigned long long)n * (k + 1) / T);
    const long long lim = (long long)floor(f->ssfrac * (double)nend);
    prefix[k] = (uint32_t)nsub;
    if (lim > nbeg) nsub += (size_t)(lim - nbeg);
  }
  if (!f->d_ss_prefix.p || f->d_ss_prefix.n < T) HIP_TRY(ctx, f->d_ss_prefix.alloc(T));
  HIP_TRY(ctx, hipMemcpyAsync(f->d_ss_prefix.p, prefix.data(), T * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));         // (`prefix` is a local)
  if (f->ss_comp && f->ss_comp->n != nsub) { exp_amd_comp_destroy(f->ss_comp); f->ss_comp = nullptr; }
  if (!f->ss_comp && (rc = exp_amd_comp_create(ctx, nsub, &f->ss_comp))) return rc;
  exp_amd_comp *sub = f->ss_comp;
  if (nsub) {
    if (f->ss_cap < nsub) {
      for (auto &b : f->d_ss) HIP_TRY(ctx, b.alloc(nsub));
      f->ss_cap = nsub;
    }
    k_subset_gather<<<cdiv(n, 256), 256, 0, ctx->stream>>>(c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M),
                                                          c->uniform_mass ? c->mass_value : 0.0, c->id[c->cur].p, n, T,
                                                          f->ssfrac, f->d_ss_prefix.p, f->d_ss[0].p, f->d_ss[1].p,
                                                          f->d_ss[2].p, f->d_ss[3].p);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc = exp_amd_comp_upload_device(sub, f->d_ss[0].p, f->d_ss[1].p, f->d_ss[2].p, f->d_ss[3].p, nullptr, nullptr,
                                         nullptr)))
      return rc;
  }
  // the sub-sample lives in the frame of the component it was drawn from, Component::freeze included (:468 sits inside the loop)
  if ((rc = exp_amd_comp_set_center(sub, c->center))) return rc;
  if ((rc = exp_amd_comp_set_rtrunc(sub, c->freeze_on ? c->rtrunc : 1.0e20, c->com0))) return rc;
  const double ms_save = f->mass_scale;
  f->mass_scale = ms_save / f->ssfrac;                       // `mass = Mass * adb; mass /= ssfrac` (:471-473)
  f->subset_on = false;                                      // (the accumulation of the sub-sample itself is a plain one)
  rc = determine_coefficients(sub, false, 0.0, 0.0, false);
  f->subset_on = true;
  f->mass_scale = ms_save;
  f->home = c;                                               // the expansion belongs to the component, not to its sample
  f->home_gone = false;
  return rc;
}

// SphericalBasis::update_noise (src/SphericalBasis.cc:2150-2210): see sph_force.h
int SphForce::update_noise(bool self_call)
{
  SphForce *f = this;
  const int L = f->cfg.lmax, nmax = f->cfg.nmax;
  if (f->noise_setup) {
    f->noise_setup = false;                       // "Only do this initialization once"
    f->rgen.seed(f->seedN);
  }
  f->noise_calls++;
  f->n_host.assign(f->ncoef, 0.0);
  for (int l = 0, loffset = 0; l <= L; loffset += (2 * l + 1), l++) {
    for (int m = 0, moffset = 0; m <= l; m++) {
      double fac = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * factrl(l - m) / factrl(l + m));     // factorial(l, m), :328-333
      if (m) fac *= M_SQRT2;
      if (m == 0) {
        for (int n = 0; n < nmax; n++) {
          double v = sqrt(fabs(f->n_rms[(size_t)l * nmax + n] - f->n_mean[n] * f->n_mean[n]) * fac / f->noiseN) * f->nrand(f->rgen);
          if (l == 0) v += f->n_mean[n];
          f->n_host[(size_t)(loffset + moffset) * nmax + n] = v;
        }
        moffset++;
      } else {
        for (int n = 0; n < nmax; n++) {
          const double a = sqrt(fabs(f->n_rms[(size_t)l * nmax + n] - f->n_mean[n] * f->n_mean[n]) * fac / f->noiseN);
          f->n_host[(size_t)(loffset + moffset + 0) * nmax + n] = a * f->nrand(f->rgen);
          f->n_host[(size_t)(loffset + moffset + 1) * nmax + n] = a * f->nrand(f->rgen);
        }
        moffset += 2;
      }
    }
  }
  // a self call of a multistep force: compute_multistep_coefficients runs AFTER update_noise and rebuilds the set from the
  // per-level ones (:395 then :1680-1685) -- the draws are consumed, the set is not touched
  if (self_call && f->cfg.multistep && (f->self_consistent || f->initializing)) return EXP_AMD_OK;
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef.p, f->n_host.data(), f->ncoef * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // (n_host is rewritten by the next call)
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

// The NOISE keys: meanC[nmax], rmsC[(lmax+1)][nmax] of SphericalBasis::compute_rms_coefs (:2108-2147), noiseN, seedN.
// meanC == NULL switches the mode off.  Setting it (re)starts the generator: the next evaluation seeds it.
extern "C" int exp_amd_sph_set_noise(exp_amd_force *fb, const double *meanC, const double *rmsC, double noiseN, unsigned seedN)
{
  expamd_mutated();
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_noise: not a spherical force");
  if (!meanC) {
    f->noise_on = false;
    f->accel_writes_coef = f->fix_l0;
    return EXP_AMD_OK;
  }
  if (!rmsC) return expamd_fail(f->ctx, EXP_AMD_ERR_ARG, "set_noise: rmsC is NULL");
  const int L = f->cfg.lmax, nmax = f->cfg.nmax;
  f->n_mean.assign(meanC, meanC + nmax);
  f->n_rms.assign(rmsC, rmsC + (size_t)(L + 1) * nmax);
  f->noiseN = noiseN;
  f->seedN = seedN;
  f->noise_setup = true;
  f->nrand.reset();                               // (no deviate saved from an earlier run of the mode)
  f->noise_calls = 0;
  f->noise_on = true;
  f->accel_writes_coef = true;                    // the evaluation itself writes the coefficient set (host.hip: ev_self ordering)
  return EXP_AMD_OK;
}

// The "ssfrac" key (src/SphericalBasis.cc:149-152: taken when 0 < ssfrac < 1, ignored otherwise) with the thread count the
// reference's partition of the level list depends on (`nthrds`, src/SphericalBasis.cc:438-439)
extern "C" int exp_amd_sph_set_subset(exp_amd_force *fb, double ssfrac, int nthrds)
{
  expamd_mutated();
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_subset: not a spherical force");
  if (nthrds < 1 || nthrds > 4096) return expamd_fail(f->ctx, EXP_AMD_ERR_ARG, "set_subset: nthrds must be in [1, 4096]");
  if (ssfrac > 0.0 && ssfrac < 1.0 && f->cfg.multistep)
    return expamd_fail(f->ctx, EXP_AMD_ERR_STATE, "set_subset: ssfrac with block multistep is not defined here (the order of "
                       "the reference's level lists is the history of its level changes)");
  f->subset_on = ssfrac > 0.0 && ssfrac < 1.0;               // "Check for sane value" (:151)
  f->ssfrac = f->subset_on ? ssfrac : 1.0;
  f->ss_nthrds = nthrds;
  return EXP_AMD_OK;
}

int SphForce::determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift,
                                     bool have_keys)
{
  SphForce *f = this;
  if (f->subset_on) return determine_coefficients_subset(c, advance, dt_kick, dt_drift);
  f->home = c;
  f->home_gone = false;
  // multistep: only level `mlevel` has moved since the store was last put in this basis' order,
  // so only its slot range is re-sorted
  const int level = (f->multistep && c->sorted_for == f && c->nlevels == f->multistep + 1)
                        ? f->mlevel : -1;
  if (level >= 0) c->sparse_mask &= ~(1u << level); else c->sparse_mask = 0;   // this call cell-sorts what it touches
  int rc = sph_sort(f, c, c->acc_live, AdvSpec::step(advance, dt_kick, dt_drift), level, have_keys && level < 0);
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

int SphForce::substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine, int phase)
{
  SphForce *f = this;
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
  if (full || stale_ok) dmax = ms;                  // a full re-partition passes over everything anyway
  if (c->n && phase != 2) {
    // one sort of the dense part of the active slot range with the per-level kick + drift applied on
    // the way; the whole store when the level partition is not this basis' yet.  acc / pot of the
    // active levels are rewritten by the force evaluation that follows (compute_potential(mfirst[
    // mstep])), so they are not carried through the reorder -- unless a full re-partition also moves
    // inactive levels.
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
      rc = sph_sort(f, c, /*move_acc=*/full && lo > 0, adv, full ? -1 : lo, keys_there, dmax);
      if (rc) return rc;
    }
    c->commit_pending = false;          // (the scatter stored the proposed levels)
    if (had) {     // an advance changes no level population: the host mirror stays what it was
      for (int k = 0; k <= ms + 1; k++) c->lev_host[k] = keep[k];
      c->lev_host_valid = true;
    }
    // sparse levels above: advanced in place -- by the thin accumulation kernel itself when that is what follows (the
    // whole active range sparse and thin: thin_adv.h), which then needs no launch for it
    if (dmax < ms && adv.mode) {
      size_t nall = 0;
      if ((rc = expamd_comp_level_count(c, lo, ms, &nall))) return rc;
      const bool fuse_on = EXPAMD_EXPT("EXP_AMD_THIN_ADVANCE", 1) != 0;
      const bool fuse = fuse_on && !f->frozen() && dmax < lo && adv.mode == 2 && nall > 0 && ctx->thin_max > 0 && (long long)nall <= ctx->thin_max * ctx->thin_acc_scale &&
                        !ctx->deterministic && f->ncoef <= 4096 && !f->generic && f->thin_lds_ok();
      if (fuse) {
        f->adv_owed = true;
        f->adv_dt_min = dt_min;
      } else if ((rc = expamd_comp_advance_levels(c, dmax + 1, ms, dt_min, ms))) return rc;
    }
    // (a half-kick owed by levels of the advanced range that no pass took along: that range was empty)
    if (c->pending_kick != 0.0 && adv.mode && !f->adv_owed && lo <= c->pending_lo) { c->pending_kick = 0.0; c->pending_lo = 0; }
  }
  if (phase == 1) return EXP_AMD_OK;
  const SphDev S = dev_acc(f, c);
  const size_t wl = (size_t)(cfg.numr - 1) * S.nrows * 2;
  // the per-level moment buffers are left clean by the contraction that consumes them (below); only
  // a buffer the plain per-level API may have used (level 0's) is cleared here
  if (!f->w_clean) {
    HIP_TRY(ctx, hipMemsetAsync(f->d_W.p, 0, f->d_W.bytes(), ctx->stream));
    f->w_clean = true;
  }
  // counts accumulated after the first sub-step are not kept (SphForce::used_open): d_used[1] is a sink
  unsigned long long *used_p = f->d_used.p + (f->used_open ? 0 : 1);
  // the sparse levels inside a full re-partition are level-contiguous but unordered as well
  int dacc = lo - 1;                    // last level the cell-ordered kernel takes
  for (int L = lo; L <= ms; L++) if (!((c->sparse_mask >> L) & 1u)) dacc = L;
  size_t nrange = 0;
  if (c->n && dacc >= lo && (rc = expamd_comp_level_count(c, lo, dacc, &nrange))) return rc;
  if (nrange) {
    ProfScope ps(ctx, "k_sph_accumulate");
    uint32_t counts[LEVCHUNK_MAX];
    for (int L = lo; L <= dacc; L++) counts[L - lo] = c->lev_host[L + 1] - c->lev_host[L];
    // Thickly and thinly populated levels go in SEPARATE launches (consecutive levels of one kind together): measured
    // on config 4, level 0 (9.6e6 particles, 3072-particle chunks) with levels 1-2 (2.5e5 + 1.2e5, 64-particle
    // chunks) in one launch takes 970 us, level 0 alone 240 us and the thin levels together 190 us.
    for (int L0 = lo; L0 <= dacc;) {
      const bool thick = counts[L0 - lo] >= ACC_THICK_MIN;
      int L1 = L0;
      size_t nr = counts[L0 - lo];
      while (L1 + 1 <= dacc && (counts[L1 + 1 - lo] >= ACC_THICK_MIN) == thick) { L1++; nr += counts[L1 - lo]; }
      if (nr) {
        SphAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, L0, L1,
                     f->d_W.p, used_p, nr, ctx->stream, 1, counts + (L0 - lo), 1};
        sph_launch_acc(f, a);
      }
      L0 = L1 + 1;
    }
  }
  nrange = 0;
  if (c->n && dacc < ms && (rc = expamd_comp_level_count(c, dacc + 1, ms, &nrange))) return rc;
  // the whole active range is sparse and thin: straight from the basis tables into the contraction's partial sums
  // (k_sph_acc_thin), no moments and no contraction (the deterministic mode keeps the moment path: its rounding grid
  // is that of the moment terms)
  const bool thin = dacc < lo && ctx->thin_max > 0 && (long long)nrange <= ctx->thin_max * ctx->thin_acc_scale && !ctx->deterministic &&
                    f->ncoef <= 4096 && (f->generic || f->thin_lds_ok());
  // (the advance this kernel was to perform, should it not run after all)
  if (f->adv_owed && !(thin && nrange)) {
    f->adv_owed = false;
    if ((rc = expamd_comp_advance_levels(c, lo, ms, f->adv_dt_min, ms))) return rc;
  }
  if (thin) {
    if (!f->part_clean) {
      HIP_TRY(ctx, hipMemsetAsync(f->d_part.p, 0, f->d_part.bytes(), ctx->stream));
      f->part_clean = true;
    }
    if (nrange) {
      ProfScope ps(ctx, "k_sph_acc_thin");
      SphThinAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->lev_off.p, lo, ms, f->d_wscale.p, f->d_part.p,
                       used_p, nrange, ctx->stream};
      if (f->adv_owed) {
        f->adv_owed = false;
        double k0 = 0.0;
        int k0lo = 0;
        if ((rc = expamd_comp_take_pending(c, lo, ms, &k0, &k0lo))) return rc;
        a.adv = ThinAdv{c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ),
                        c->level[c->cur].p, f->adv_dt_min, ms, k0, k0lo, 1};
      }
      if (!f->generic) k_thin_acc_launch[cfg.lmax](a); else expamd_sph_thin_acc_gen(a);
    }
  } else if (nrange) {
    ProfScope ps(ctx, "k_sph_accumulate_sparse");
    SphUpdArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, nullptr,
                 c->lev_off.p, dacc + 1, ms, 0, f->d_W.p, nrange, ctx->stream, 1, used_p};
    if ((rc = sph_stage(f, nrange, a))) return rc;
    sph_launch_upd(f, a);
  }
  {
    ProfScope ps(ctx, "k_sph_contract");
    if (!thin)
      k_sph_contract<<<dim3(CSEG, S.nrows, nact), 256, 0, ctx->stream>>>(
          S, f->d_W.p + (size_t)lo * wl, f->d_wscale.p, f->d_part.p, /*clear=*/1);
    // ... with the N/L swap of every active level (src/SphericalBasis.cc:785-792): L <- N, N <- new
    // (the summing kernels leave the partial sums they read ZERO: what the thin accumulation, which adds to them, starts
    // from -- no memset between a table-path sub-step and a thin one)
    if (mdrft_combine >= 0) {
      int mfc = 0;
      CombineW Wc;
      expamd_combine_weights(ms, mdrft_combine, &mfc, &Wc);
      k_sph_sum_combine<<<cdiv(f->ncoef, 256), 256, 0, ctx->stream>>>(
          f->d_part.p, (int)f->ncoef, f->d_coefN.p, f->d_coefL.p, lo, nact, ms + 1, mfc, Wc, f->d_coef.p, /*clear=*/1);
    } else
    k_sph_sum_parts<<<dim3(cdiv(f->ncoef, 256), nact), 256, 0, ctx->stream>>>(
        f->d_part.p, (int)f->ncoef, f->d_coefN.p + (size_t)lo * f->ncoef, f->d_coefL.p + (size_t)lo * f->ncoef, nullptr,
        /*clear=*/1);
  }
  HIP_TRY(ctx, hipGetLastError());
  f->combined_mdrft = mdrft_combine;
  if (mdrft_combine < 0 &&
      (rc = expamd_allreduce(ctx, f->d_coefN.p + (size_t)lo * f->ncoef, (size_t)nact * f->ncoef))) return rc;
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

int sph_project(SphForce *f)
{
  if (!f->proj_dirty) return EXP_AMD_OK;
  exp_amd_ctx *ctx = f->ctx;
  ProfScope ps(ctx, "k_sph_project");
  const bool both = EXPAMD_EXPT("EXP_AMD_SPH_PROJECT_BOTH", 1) != 0;
  const size_t lds = (size_t)3 * f->dev.nrows * sizeof(double);
  // (one launch where a block's three nodes' rows fit one round of its threads -- lmax <= 8; above, the threefold sums
  // cost more than the launch saved: lmax 10, nmax 24: 35 us against 11 + 5)
  if (both && 3 * f->dev.nrows <= 256 && lds <= 48 * 1024 && f->cfg.numr >= 3)
    k_sph_project_both<<<f->cfg.numr - 1, 256, lds, ctx->stream>>>(f->dev, f->d_coef.p, f->d_rowmap.p, f->d_tscale.p,
                                                                   f->d_G.p, f->d_T4.p);
  else {
    k_sph_project<<<f->cfg.numr, 256, 0, ctx->stream>>>(f->dev, f->d_coef.p, f->d_G.p);
    k_sph_project4<<<f->cfg.numr - 1, 256, 0, ctx->stream>>>(f->dev, f->d_G.p, f->d_rowmap.p,
                                                            f->d_tscale.p, f->d_T4.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = false;
  return EXP_AMD_OK;
}

int SphForce::accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick, double nk_dtk,
                         double nk_dtd, bool *prekey_done, bool defer_kick)
{
  SphForce *f = this;
  if (prekey_done) *prekey_done = false;
  int rc;
  // A thin target range (a block-multistep sub-step's few active particles, ours or another component's) is evaluated
  // straight from the coefficient set (k_sph_force_thin): the projected table is not needed and stays stale.
  bool thin = false;
  size_t nthin = 0;
  if (f->cfg.multistep > 0 && t->n && t->nlevels > 1 && dt_kick == 0.0 && !prekey_done && !ctx->deterministic &&
      ctx->thin_max > 0) {
    if ((rc = expamd_comp_level_count(t, f->mlevel, t->nlevels - 1, &nthin))) return rc;
    thin = (long long)nthin <= ctx->thin_max;
  }
  // `if (NOISE) update_noise();` opens get_acceleration_and_potential, for self and external calls alike (:395)
  if (f->noise_on && (rc = f->update_noise(!external))) return rc;
  if (f->fix_l0) {
    // "Save the monopole coefficients on the first evaluation / Copy the saved coefficients to the active array"
    // (src/SphericalBasis.cc:1689-1694; outside the use_external test: self and external calls alike)
    const size_t nb = (size_t)f->cfg.nmax * sizeof(double);
    if (!f->have_c0) {
      HIP_TRY(ctx, hipMemcpyAsync(f->d_c0.p, f->d_coef.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
      f->have_c0 = true;
    } else {
      HIP_TRY(ctx, hipMemcpyAsync(f->d_coef.p, f->d_c0.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
      f->proj_dirty = true;
    }
  }
  if (!thin && (rc = sph_project(f))) return rc;
  f->used_open = false;          // tnow has moved past resetT once forces are evaluated
  if (t->n == 0) return EXP_AMD_OK;
  if (thin) {
    if (nthin) {
      ProfScope ps(ctx, "k_sph_force_thin");
      const double *ctr_ = !external ? t->center : f->home ? f->home->center : f->home_gone ? f->home_center : t->center;
      SphDev S = dev_for(f, ctr_);
      S.ps = t->pseudo;
      dev_freeze(S, t);
      SphThinForceArgs a{S, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, f->mlevel, t->nlevels - 1, f->d_coef.p,
                         f->d_rowmap.p, f->d_tscale.p, t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX),
                         t->a(A_VY), t->a(A_VZ), assign ? 1 : 0, nthin, ctx->stream};
      // (the tiled kernel keeps the coefficient set and four particles' rows in LDS: where that does not fit -- very large
      // nmax -- the one-wave-per-particle kernel, which uses none, takes over)
      const size_t tq_ = 4 * (size_t)f->dev.trows + 16, lsn_ = (size_t)(f->cfg.lmax + 1) * f->cfg.nmax;
      const bool fits = (f->ncoef + 2 + 4 * (tq_ + 3 * lsn_)) * sizeof(double) <= 120 * 1024;
      if (!f->generic && fits) k_thin_force_launch[f->cfg.lmax](a); else expamd_sph_thin_force_gen(a);
      HIP_TRY(ctx, hipGetLastError());
    }
    t->acc_live = true;
    return EXP_AMD_OK;
  }
  // next step's keys + histogram: single level, own (sorted) particles, fused half-kick only
  const bool prekey = prekey_done && nk_dtd != 0.0 && dt_kick != 0.0 && !external &&
                      t->nlevels == 1 && f->cfg.multistep == 0 && t->sorted_for == f;

  // external target: positions go into the frame of the component the expansion was built from
  const double *ctr = !external ? t->center : f->home ? f->home->center : f->home_gone ? f->home_center : t->center;
  SphDev S = dev_for(f, ctr);
  S.ps = t->pseudo;                    // Component::AddAcc of the TARGET (src/Component.H:914-921)
  dev_freeze(S, t);
  // closing half-kick: stored (1), left to the next scatter pass (0), or -- when that pass is known,
  // i.e. its keys are produced here -- stored together with its opening half-kick (2)
  const bool deferred = defer_kick && dt_kick != 0.0;
  const int sv = !deferred ? 1 : (prekey && nk_dtk != 0.0 && ctx->prekick) ? 2 : 0;
  const int lo = (t->nlevels > 1) ? f->mlevel : 0;
  const int hi = t->nlevels - 1;
  size_t nr = t->n;                    // population of the level range: sizes the launch
  if (t->nlevels > 1 && (rc = expamd_comp_level_count(t, lo, hi, &nr))) return rc;
  if (nr == 0) { t->acc_live = true; return EXP_AMD_OK; }
  // a range of sparse (not cell-sorted) levels only: no wave would pass the fast pass' uniformity test
  bool all_sparse = t->nlevels > 1 && t->sorted_for == f;
  for (int L = lo; L <= hi && all_sparse; L++) if (!((t->sparse_mask >> L) & 1u)) all_sparse = false;
  {
    unsigned grid = cdiv(nr, 256);     // one 64-particle chunk per wave, no loop
    const size_t need = t->n / 64 + 8;
    // (the work list belongs to the fast pass: the staged / gather evaluation of foreign or all-sparse targets has none --
    // and must not re-allocate it under a self force that runs on the other stream)
    const bool listless = !ctx->deterministic && (t->sorted_for != f || all_sparse);
    if (f->work_cap < need && !(listless && f->work_cap > 0)) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, f->d_work.alloc(SPH_WORK_STRIDE * need + 2));   // work list + two counters (used alternately)
      HIP_TRY(ctx, hipMemsetAsync(f->d_work.p + SPH_WORK_STRIDE * need, 0, 2 * sizeof(uint32_t), ctx->stream));
      f->work_cap = need;
      f->work_flip = 0;
    }
    // Waterfall fast pass: every lane gets the uniform-wave arithmetic whatever its neighbours are --
    // what the deterministic mode needs (the path a particle takes must not depend on the slot it
    // landed in).  (Tried for targets in ANOTHER basis' cell order too: a wave of disk particles spans
    // too many radial cells, 4.5 ms against 2.8 ms by gathers in config 4.)
    const bool foreign = t->sorted_for != f;
    const bool wfall = ctx->deterministic;
    const bool slow = !wfall && (foreign || all_sparse);
    uint32_t *cnt = f->d_work.p + SPH_WORK_STRIDE * f->work_cap;
    if (f->lit_on) {
      // room for every particle of this evaluation (a few ever land there: 4 bytes each beside the store's 186)
      if (f->d_litlist.n < nr + 1) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        f->d_litlist.release();
        if (f->d_litlist.alloc(nr + 1) != hipSuccess)
          return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph accelerate: hipMalloc of the literal-pass list failed");
      }
      S.lit_list = f->d_litlist.p;
      S.lit_cap = (uint32_t)nr;
    }
    SphForceArgs a{S, t->a(A_X), t->a(A_Y), t->a(A_Z), t->lev_off.p, lo, hi, f->d_T4.p,
                   t->a(A_AX), t->a(A_AY), t->a(A_AZ), t->a(A_POT), t->a(A_VX), t->a(A_VY),
                   t->a(A_VZ), dt_kick, assign ? 1 : 0, nr, grid, ctx->stream,
                   f->d_work.p, cnt + f->work_flip, slow ? 1 : 0, ctx,
                   prekey ? t->key.p : nullptr, nk_dtk, nk_dtd, sv,
                   cnt + (1 - f->work_flip), wfall ? 1 : 0};
    if (slow && foreign) {
      // another component's particles: local in radius though not in this basis' cell order -- the rows of each
      // block's cell range go through LDS (k_sph_force_staged), about 40 KB per block (three blocks per CU, as many as
      // its registers allow).  Measured on 1e7 disk particles in (R, z)-cell order (tools/dbg/cross_force.py,
      // profiles/r03_cross_force_staging.txt): lmax 6: global gathers 0.98 ms, 8 / 16 / 24 rows 0.57 / 0.51 / 0.48 ms;
      // lmax 10: 2.21 ms, 4 / 7 / 12 rows 1.72 / 1.32 / 1.08 ms
      const int tq = 4 * S.trows, tqs = tq + ((2 - tq % 16) + 16) % 16;
      int rows = (int)(40960 / ((size_t)tqs * sizeof(double)));
      a.stage_rows = rows > 24 ? 24 : rows < 4 ? 4 : rows;
      a.stage_rows = (int)EXPAMD_EXPT("EXP_AMD_STAGE_ROWS", a.stage_rows);
      // the lanes that kernel leaves to the general pass (polar axis, beyond rmax): a wave and its lane mask per entry
      if (f->xwork_cap < need) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, f->d_xwork.alloc(SPH_WORK_STRIDE * need + 2));
        HIP_TRY(ctx, hipMemsetAsync(f->d_xwork.p + SPH_WORK_STRIDE * need, 0, 2 * sizeof(uint32_t), ctx->stream));
        f->xwork_cap = need;
        f->xwork_flip = 0;
      }
      uint32_t *xcnt = f->d_xwork.p + SPH_WORK_STRIDE * f->xwork_cap;
      a.work = f->d_xwork.p;
      a.nwork = xcnt + f->xwork_flip;
      a.nwork_next = xcnt + (1 - f->xwork_flip);
      if (a.stage_rows > 0 && dt_kick == 0.0 && !prekey) f->xwork_flip ^= 1;      // (the launcher's own condition for that kernel)
    }
    sph_launch_force(f, a);
    if (!slow) f->work_flip ^= 1;
  }
  HIP_TRY(ctx, hipGetLastError());
  t->acc_live = true;
  if (prekey) *prekey_done = true;
  if (deferred) t->pending_kick = sv == 2 ? -nk_dtk : dt_kick;
  return EXP_AMD_OK;
}


// ---- append fused step ------------------------------------------------------------------------------------------
// One KDK step (src/step.cc:271-323) of a large single-level component WITHOUT sort passes (sph_kernels.h: AppDev).  In the
// fused step's steady state the force pass of step n already computes where every particle will be at step n + 1 (that step's
// sort key); here it also stores the particle there -- position of step n + 1 (what the next accumulation reads), the slot
// it came from (where the position of step n, what a download sees, stays until the next pass), the velocity with both half-kicks of the step boundary, acceleration, potential, id --
// in the OTHER buffer set, whose cells are regions sized from the cells' present populations plus slack.  A step is then
//   layout of the other set <- populations;  accumulate;  contract, all-reduce, project;  force + place;  mark empty slots
// and the key histogram, the scan and the scatter pass (112 B per particle, 2.2 of 10.1 ms at 1e8) are gone.  The first such
// step scatters the ordinary sorted store into a layout of this kind (the ordinary passes with the regions' offsets for
// cursors).  A pass that runs out of room (a region AND the tail full: a flag, read back after every step) is redone the
// ordinary way from its source set, which no pass writes.  Any other call on the component turns the store back into an
// ordinary one first (expamd_comp_densify).  Trajectories: those of the ordinary fused step up to the order of the sums.
// The frame of the last placing pass, put back for an evaluation on its behalf (the state's acceleration and potential the lean
// payload leaves out; the redo of a pass that ran out of room): the coefficient set KEPT at that step (d_coef_app, a 23 KB copy
// per step) in place of whatever the force holds by now, the centre of that step, none of the options under which the mode is
// not offered.  sph_app_frame_end puts everything back.
struct SphAppFrame {
  double ctr[3];
  bool frz, noise, fix0, used_open;
  PseudoDev ps;
};

static int sph_app_frame_begin(SphForce *f, exp_amd_comp *c, SphAppFrame &F)
{
  exp_amd_ctx *ctx = f->ctx;
  if (f->d_coef_app.n < 2 * f->ncoef) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "appended store: no coefficient set kept");
  const size_t nb = f->ncoef * sizeof(double);
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef_app.p + f->ncoef, f->d_coef.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef.p, f->d_coef_app.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  f->proj_dirty = true;
  for (int k = 0; k < 3; k++) { F.ctr[k] = c->center[k]; c->center[k] = c->app_center[k]; }
  F.frz = c->freeze_on; F.noise = f->noise_on; F.fix0 = f->fix_l0; F.used_open = f->used_open; F.ps = c->pseudo;
  c->freeze_on = false;
  c->pseudo.center = c->pseudo.axis = 0;
  f->noise_on = false;
  f->fix_l0 = false;
  return EXP_AMD_OK;
}

static int sph_app_frame_end(SphForce *f, exp_amd_comp *c, const SphAppFrame &F, int rc)
{
  exp_amd_ctx *ctx = f->ctx;
  c->freeze_on = F.frz;
  c->pseudo = F.ps;
  f->noise_on = F.noise;
  f->fix_l0 = F.fix0;
  f->used_open = F.used_open;
  for (int k = 0; k < 3; k++) c->center[k] = F.ctr[k];
  if (hipMemcpyAsync(f->d_coef.p, f->d_coef_app.p + f->ncoef, f->ncoef * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess && !rc)
    rc = expamd_fail(ctx, EXP_AMD_ERR_HIP, "appended store: restoring the coefficient set failed");
  f->proj_dirty = true;
  return rc;
}

// The state's acceleration and potential after append steps with the LEAN payload.  The placing pass stores neither (AppDev):
// when the store has been turned back into an ordinary one -- positions of the completed step n, velocities ahead by the next
// opening half-kick -- they are evaluated here, once, in the frame of that step.  Bits: those of the staged evaluation
// (k_sph_force_staged: the fast pass' arithmetic) -- a particle the in-step pass took through its general pass may differ
// from the value its kick used in the last place or two, which is what the order of the coefficient sums does to it anyway.
static int sph_app_reeval(void *owner, exp_amd_comp *c)
{
  SphForce *f = static_cast<SphForce *>(owner);
  if (!c->n) return EXP_AMD_OK;
  SphAppFrame F;
  int rc = sph_app_frame_begin(f, c, F);
  if (rc) return rc;
  rc = f->accelerate(c, 0, /*assign=*/true, 0.0, 0.0, 0.0, nullptr, false);
  return sph_app_frame_end(f, c, F, rc);
}

// A placing pass ran out of room (a region AND the tail full: exp_amd_comp::app_hflag): its source set is intact and holds the
// step's advanced positions and the velocities with its opening
// half-kick; the coefficient set of the step is in place, summed over the ranks and projected.  The source set is made an
// ordinary store and the force pass -- that alone -- is redone on it, with the frame of that step: the state after it is the
// completed step's, as the ordinary fused step leaves it (closing half-kick deferred).  No collective: the ranks that had room
// issue none for this either.
static int sph_app_recover(SphForce *f, exp_amd_comp *c)
{
  const double dt = c->app_dt;
  int rc;
  c->cur = 1 - c->cur;                 // (back on the source set of the pass)
  c->pending_kick = 0.0;
  c->app_acc_stale = false;
  if ((rc = expamd_comp_densify(c, /*state_positions=*/false))) return rc;
  SphAppFrame F;
  if ((rc = sph_app_frame_begin(f, c, F))) return rc;
  bool done = false;
  rc = f->accelerate(c, 0, true, 0.5 * dt, 0.5 * dt, dt, &done, /*defer_kick=*/c->n > 0);
  f->firstime_coef = false;
  return sph_app_frame_end(f, c, F, rc);
}

int SphForce::fused_step_append(exp_amd_comp *c, double dt, bool have_keys, bool *handled)
{
  SphForce *f = this;
  *handled = false;
  int rc;
  const double dt_kick = 0.5 * dt;
  const bool cont = c->appended && c->app_owner == (const void *)f && c->app_dt == dt && c->pending_kick == -dt_kick &&
                    c->app_center[0] == c->center[0] && c->app_center[1] == c->center[1] && c->app_center[2] == c->center[2];
  const long long amin = ctx->append_min < 0 ? -ctx->append_min : ctx->append_min;
  const bool offered = amin > 0 && c->n >= (size_t)amin && !f->cfg.multistep && !f->lit_on &&
                       !f->subset_on && !f->noise_on && !f->generic && !ctx->deterministic && ctx->prekick && !c->freeze_on &&
                       !(c->pseudo.center | c->pseudo.axis) && c->nlevels == 1 && c->levels_zero && !f->fix_l0 &&
                       c->n < 0x70000000u && !c->app_refused && !(c->appended && !cont);
  if (c->appended && !(cont && offered)) return expamd_comp_densify(c);     // (the ordinary step takes over)
  if (!offered) return EXP_AMD_OK;
  if (!cont && c->app_wait > 0) { c->app_wait--; return EXP_AMD_OK; }       // (hysteresis: particles.h)
  // entry: the ordinary fused step's steady state -- sorted for this force, velocities stored with this step's opening
  // half-kick, this step's keys written by the last force pass
  if (!cont && !(have_keys && c->sorted_for == (const void *)f && c->pending_kick == -dt_kick && !c->split)) return EXP_AMD_OK;
  const uint32_t ncell = (uint32_t)(f->cfg.numr - 1);
  const size_t cap = expamd_app_slots(c->n, ncell);
  f->home = c;
  f->home_gone = false;
  hipStream_t st = ctx->stream;
  if (!cont) {
    if ((rc = expamd_comp_app_reserve(c, cap))) return rc;
    if (c->app_refused) return EXP_AMD_OK;           // (no room for the layout: the ordinary step)
    c->app_ns = cap;
    if ((rc = expamd_comp_prepare_hist(c, ncell))) return rc;
    {
      ProfScope ps(ctx, "k_hist_keys");
      k_hist_keys<<<cdiv(c->n, HIST_TILE), SORT_TPB, 0, st>>>(c->key.p, c->n, c->hist.p, 0u, ncell, (uint32_t)SORT_WIN);
    }
    // both sets get a layout from these populations: the one the scatter fills now, the one the force pass fills below
    if ((rc = expamd_comp_app_layout(c, c->hist.p, ncell, c->cur, nullptr))) return rc;
    if ((rc = expamd_comp_app_layout(c, c->hist.p, ncell, 1 - c->cur, c->hist.p))) return rc;
    const AdvanceArgs A = expamd_advance_args(c, AdvSpec::step(true, dt_kick, dt));
    const ScatterSrc Ssrc{c->a(A_M), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), c->id[c->cur].p};
    const ScatterDst Sdst{c->b(A_X), c->b(A_Y), c->b(A_Z), c->b(A_VX), c->b(A_VY), c->b(A_VZ),
                          c->uniform_mass ? nullptr : c->b(A_M), c->b(A_AX), c->b(A_AY), c->b(A_AZ), c->b(A_POT),
                          c->id[1 - c->cur].p, nullptr};
    {
      ProfScope ps(ctx, "k_scatter_adv");
      k_scatter_adv<false><<<cdiv(c->n, SCAT_TILE), SORT_TPB, 0, st>>>(A, Ssrc, Sdst, expamd_sort_range(c, -1), c->key.p,
                                                                        c->hist.p, (uint32_t)SORT_WIN);
    }
    k_app_mark_launch(c, 1 - c->cur, c->hist.p);
    HIP_TRY(ctx, hipGetLastError());
    c->hist_clean = 0;
    c->pending_kick = 0.0;
    c->cur = 1 - c->cur;
    c->appended = true;
    c->app_owner = f;
    c->app_reeval = sph_app_reeval;
    ctx->appended.push_back(c);
    c->app_dt = dt;
    for (int k = 0; k < 3; k++) c->app_center[k] = c->center[k];
    c->sorted_for = nullptr;          // (no ordinary pass may take this set for a sorted dense one)
    c->acc_live = false;
  } else {
    // the other set's layout from the populations the last pass counted (its cursors; cleared on the way)
    if ((rc = expamd_comp_app_layout(c, c->app_cursor.p, ncell, 1 - c->cur, nullptr))) return rc;
    c->pending_kick = 0.0;            // (the store holds this step's positions and opening half-kick: nothing is owed)
  }
  const int src = c->cur, dst = 1 - c->cur;
  // ---- accumulate, reduce, project
  if ((rc = sph_accumulate(f, c, f->d_coef.p, c->app_range[src].p, c->app_ns))) return rc;
  if ((rc = expamd_allreduce(ctx, f->d_coef.p, f->ncoef))) return rc;
  // the lean payload (exp_amd_ctx_set_append_lean): the pass places neither acceleration nor potential, and the set they come
  // from is kept -- whatever happens to d_coef before someone asks for them (sph_app_reeval)
  const bool lean = ctx->append_lean;
  // (the set this step's pass evaluates with is kept, whatever happens to d_coef before someone acts on the pass' behalf:
  // sph_app_frame_begin)
  if (f->d_coef_app.n < 2 * f->ncoef) HIP_TRY(ctx, f->d_coef_app.alloc(2 * f->ncoef));
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef_app.p, f->d_coef.p, f->ncoef * sizeof(double), hipMemcpyDeviceToDevice, st));
  f->proj_dirty = true;
  if ((rc = sph_project(f))) return rc;
  f->used_open = false;
  // ---- force pass that places its results in the other set
  {
    const size_t need = c->app_ns / 64 + 8;
    if (f->work_cap < need) {
      HIP_TRY(ctx, hipStreamSynchronize(st));
      HIP_TRY(ctx, f->d_work.alloc(SPH_WORK_STRIDE * need + 2));
      HIP_TRY(ctx, hipMemsetAsync(f->d_work.p + SPH_WORK_STRIDE * need, 0, 2 * sizeof(uint32_t), st));
      f->work_cap = need;
      f->work_flip = 0;
    }
    SphDev S = dev_for(f, c->center);
    S.ps = c->pseudo;
    dev_freeze(S, c);
    uint32_t *cnt = f->d_work.p + SPH_WORK_STRIDE * f->work_cap;
    const AppDev app{c->arr[dst][A_X].p, c->arr[dst][A_Y].p, c->arr[dst][A_Z].p, c->app_src[dst].p,
                     c->arr[dst][A_VX].p, c->arr[dst][A_VY].p, c->arr[dst][A_VZ].p,
                     lean ? nullptr : c->arr[dst][A_AX].p, lean ? nullptr : c->arr[dst][A_AY].p,
                     lean ? nullptr : c->arr[dst][A_AZ].p, lean ? nullptr : c->arr[dst][A_POT].p,
                     c->uniform_mass ? nullptr : c->arr[dst][A_M].p, c->id[dst].p,
                     c->arr[src][A_M].p, c->id[src].p, c->app_base[dst].p, c->app_cursor.p,
                     c->app_cursor.p + ncell + 1, ncell};
    SphForceArgs a{S, c->arr[src][A_X].p, c->arr[src][A_Y].p, c->arr[src][A_Z].p, c->app_range[src].p, 0, 0, f->d_T4.p,
                   c->arr[src][A_AX].p, c->arr[src][A_AY].p, c->arr[src][A_AZ].p, c->arr[src][A_POT].p,
                   c->arr[src][A_VX].p, c->arr[src][A_VY].p, c->arr[src][A_VZ].p, dt_kick, 1, c->app_ns,
                   (unsigned)cdiv(c->app_ns, 256), st, f->d_work.p, cnt + f->work_flip, 0, ctx, nullptr, dt_kick, dt, 2,
                   cnt + (1 - f->work_flip), 0};
    a.app = &app;
    sph_launch_force(f, a);
    f->work_flip ^= 1;
  }
  HIP_TRY(ctx, hipGetLastError());
  uint32_t lost = 0;
  if ((rc = expamd_comp_app_finish(c, dst, &lost))) return rc;  // (empty slots marked; did every particle find room?)
  c->cur = dst;
  c->app_run++;
  c->acc_live = true;
  c->app_acc_stale = lean;            // (the pass placed neither acceleration nor potential: particles.h)
  c->pending_kick = -dt_kick;         // velocities stored with the next step's opening half-kick (as the ordinary step's prekick)
  c->prekey_valid = false;
  f->firstime_coef = false;
  *handled = true;
  // no room (a region and the tail full): the step is put right from its source set (complete either way: handled)
  if (lost) return sph_app_recover(f, c);
  return EXP_AMD_OK;
}

// ---- split fused step -------------------------------------------------------------------------------------------
// One KDK step (src/step.cc:271-323) of a single-level component whose slots [0, half) and
// [half, n) are kept as two independently cell-sorted halves.  Per half h the chain is
//   force_{n-1}(h) -> count keys(h) -> scan(h) -> scatter+advance(h) -> accumulate(h)
// and the two chains only meet at the coefficient sum.  The sort passes are HBM-bound, the
// accumulate and force passes fp64-VALU-bound, so they are issued on two streams in an order that
// pairs unlike kernels:        main:  F(0)   F(1)      Acc(0)    Acc(1)  contract  project | F(0) ...
//                              aux :         sort(0)   sort(1)                            |
// (sort(0) of step n+1 runs under F(1) of step n, sort(1) under Acc(0)); events carry the per-half
// dependencies.  Results differ from the unsplit step only in the order of the coefficient sums.

int SphForce::fused_step_split(exp_amd_comp *c, double dt, bool have_keys, bool *handled)
{
  SphForce *f = this;
  *handled = false;
  if (f->cfg.multistep || ctx->split_min <= 0 || c->n < (size_t)ctx->split_min || c->n >= 0x7fffffffu || f->lit_on ||
      f->subset_on)
    return EXP_AMD_OK;
  int rc = expamd_ctx_aux(ctx);
  if (rc) return rc;
  hipStream_t V = ctx->stream, H = ctx->aux;
  const uint32_t ncell = (uint32_t)(f->cfg.numr - 1);
  const uint32_t nkeys = 2u * ncell;
  const double dt_kick = 0.5 * dt;
  f->home = c;
  f->home_gone = false;
  if (!c->split) {
    // enter the mode: fix the halves (block-aligned boundary), nothing is known about the order
    HIP_TRY(ctx, hipStreamSynchronize(V));
    c->half = (c->n / 2) & ~(size_t)1023;
    if (!c->half_off.p) HIP_TRY(ctx, c->half_off.alloc(4));
    const uint32_t ho[4] = {0u, (uint32_t)c->half, (uint32_t)c->n, 0u}, lo1[2] = {0u, (uint32_t)c->n};
    HIP_TRY(ctx, hipMemcpy(c->half_off.p, ho, sizeof(ho), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(c->lev_off.p, lo1, sizeof(lo1), hipMemcpyHostToDevice));
    if (c->hist_cap < (size_t)nkeys + 1) {
      HIP_TRY(ctx, c->hist.alloc((size_t)nkeys + 1));
      c->hist_cap = (size_t)nkeys + 1;
    }
    c->hist_clean = 0;
    c->nlevels = 1;
    c->lev_host_valid = false;
    have_keys = false;
    // main-stream work issued so far (uploads, an unsplit force pass) precedes both chains
    HIP_TRY(ctx, hipEventRecord(ctx->ev_forced[0], V));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_forced[1], V));
  }
  if (c->hist_cap < (size_t)nkeys + 1) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "split step: histogram too small");
  c->hist_clean = 0;                 // (this path counts into the histogram with its own memsets)
  const size_t beg[2] = {0, c->half}, len[2] = {c->half, c->n - c->half};

  // ---- aux stream: the two sort chains (reading the live set, writing the other one)
  const AdvanceArgs A = expamd_advance_args(c, AdvSpec::step(true, dt_kick, dt));
  const ScatterSrc Ssrc{c->a(A_M), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), c->id[c->cur].p};
  const ScatterDst Sdst{c->b(A_X), c->b(A_Y), c->b(A_Z), c->b(A_VX), c->b(A_VY), c->b(A_VZ),
                        c->uniform_mass ? nullptr : c->b(A_M),
                        c->b(A_AX), c->b(A_AY), c->b(A_AZ), c->b(A_POT), c->id[1 - c->cur].p,
                        c->levels_zero ? nullptr : c->level[1 - c->cur].p};
  // (experimental builds: EXP_AMD_SPLIT_LDS = bytes of dynamic LDS asked for by the sort passes of this path on top of their
  // 16 KB -- nothing uses it; it bounds how many of their blocks a CU takes (48 KB: two), so that the small-footprint
  // HBM-bound blocks cannot crowd the fp64-bound kernel's waves out of the CU; profiles/r06_overlap_ab.txt)
  const size_t pad_lds = (size_t)EXPAMD_EXPT("EXP_AMD_SPLIT_LDS", 0);
  for (int h = 0; h < 2; h++) {
    HIP_TRY(ctx, hipStreamWaitEvent(H, ctx->ev_forced[h], 0));
    const SortRange R{c->half_off.p, h, h, c->n};
    HIP_TRY(ctx, hipMemsetAsync(c->hist.p + (size_t)h * ncell, 0, ((size_t)ncell + (h ? 1 : 0)) * sizeof(uint32_t), H));
    if (have_keys) {
      ProfScope ps(ctx, "k_hist_keys", H);
      k_hist_keys<<<cdiv(len[h], HIST_TILE), SORT_TPB, pad_lds, H>>>(c->key.p + beg[h], len[h], c->hist.p);
    } else {
      ProfScope ps(ctx, "k_key_hist", H);
      SphDev S = dev_for(f, c->center);
      S.key_add = (uint32_t)h * ncell;
      k_key_hist<SphKeyFn><<<cdiv(len[h], HIST_TILE), SORT_TPB, 0, H>>>(SphKeyFn{S, 0u}, A, R, c->key.p, c->hist.p);
    }
    {
      ProfScope ps(ctx, "k_scan", H);
      { int rc_ = expamd_launch_scan(ctx, H, c->hist.p, nkeys, c->half_off.p, ncell, 2, h); if (rc_) return rc_; }
    }
    {
      ProfScope ps(ctx, "k_scatter_adv", H);
      k_scatter_adv<false><<<cdiv(len[h], SCAT_TILE), SORT_TPB, pad_lds, H>>>(A, Ssrc, Sdst, R, c->key.p, c->hist.p);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sorted[h], H));
  }
  c->pending_kick = 0.0;
  c->cur = 1 - c->cur;
  c->split = true;
  c->sorted_for = f;
  c->acc_live = false;

  // ---- main stream: accumulate the halves as they arrive, reduce, project, force
  SphDev S = dev_acc(f, c);
  HIP_TRY(ctx, hipMemsetAsync(f->d_W.p, 0, (size_t)(f->cfg.numr - 1) * S.nrows * 2 * sizeof(double), V));
  f->w_clean = false;
  HIP_TRY(ctx, hipMemsetAsync(f->d_used.p, 0, sizeof(unsigned long long), V));
  for (int h = 0; h < 2; h++) {
    HIP_TRY(ctx, hipStreamWaitEvent(V, ctx->ev_sorted[h], 0));
    if (!len[h]) continue;
    ProfScope ps(ctx, "k_sph_accumulate");
    SphAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->half_off.p, h, h,
                 f->d_W.p, f->d_used.p, len[h], V, 0};
    sph_launch_acc(f, a);
  }
  {
    ProfScope ps(ctx, "k_sph_contract");
    k_sph_contract<<<dim3(CSEG, S.nrows), 256, 0, V>>>(S, f->d_W.p, f->d_wscale.p, f->d_part.p);
    f->part_clean = false;
    k_sph_sum_parts<<<cdiv(f->ncoef, 256), 256, 0, V>>>(f->d_part.p, (int)f->ncoef, f->d_coef.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  if ((rc = expamd_allreduce(ctx, f->d_coef.p, f->ncoef))) return rc;
  f->proj_dirty = true;
  if ((rc = sph_project(f))) return rc;
  {
    const size_t need = c->n / 64 + 8;
    if (f->work_cap < need) {
      HIP_TRY(ctx, hipStreamSynchronize(V));
      HIP_TRY(ctx, f->d_work.alloc(SPH_WORK_STRIDE * need + 2));
      HIP_TRY(ctx, hipMemsetAsync(f->d_work.p + SPH_WORK_STRIDE * need, 0, 2 * sizeof(uint32_t), V));
      f->work_cap = need;
      f->work_flip = 0;
    }
  }
  for (int h = 0; h < 2; h++) {
    if (len[h]) {
      SphDev Sh = S;
      Sh.key_add = (uint32_t)h * ncell;
      Sh.ps = c->pseudo;
      dev_freeze(Sh, c);
      SphForceArgs a{Sh, c->a(A_X), c->a(A_Y), c->a(A_Z), c->half_off.p, h, h, f->d_T4.p,
                     c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), c->a(A_VX), c->a(A_VY),
                     c->a(A_VZ), dt_kick, 1, len[h], (unsigned)cdiv(len[h], 256), V,
                     f->d_work.p, f->d_work.p + SPH_WORK_STRIDE * f->work_cap + f->work_flip, 0, ctx, c->key.p, dt_kick, dt, 0,
                     f->d_work.p + SPH_WORK_STRIDE * f->work_cap + (1 - f->work_flip), ctx->deterministic ? 1 : 0};
      sph_launch_force(f, a);
      f->work_flip ^= 1;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_forced[h], V));
  }
  HIP_TRY(ctx, hipGetLastError());
  c->acc_live = true;
  c->pending_kick = dt_kick;
  c->prekey_valid = true;
  c->prekey_split = true;
  c->prekey_owner = f;
  c->prekey_epoch = ctx->force_epoch;
  c->prekey_dtk = dt_kick;
  c->prekey_dtd = dt;
  for (int k = 0; k < 3; k++) c->prekey_center[k] = c->center[k];
  *handled = true;
  return EXP_AMD_OK;
}

// ---- multistep level changes ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_add_inplace(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) dst[k] += src[k];
}

int SphForce::resort(exp_amd_comp *c, int first)
{
  // levels below `first` were not examined (src/multistep.cc:451-453): their slots stay as they are
  if (first > 0 && c->nlevels == multistep + 1)      // (the caller vouches for the order below `first`)
    return sph_sort(this, c, true, AdvSpec(), first, false, multistep);
  return sph_sort(this, c, true);
}

bool SphForce::prekey_launcher(exp_amd_comp *c, ka_launch_fn *fn, void **self)
{
  if (cfg.multistep == 0 || c->n == 0 || c->n >= 0x7fffffffu) return false;
  ka_S = dev_for(this, c->center);
  *fn = [](void *p, const KaLaunch &L) { ka_launch_with(L, SphKeyFn{static_cast<SphForce *>(p)->ka_S, 0u}); };
  *self = this;
  return true;
}

int SphForce::multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft)
{
  SphForce *f = this;
  const int ms = f->multistep;
  if (ms == 0) return EXP_AMD_OK;
  const size_t wl = (size_t)(cfg.numr - 1) * dev.nrows * 2;
  if (f->d_Wd.n == 0) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, f->d_Wd.alloc(wl * (ms + 1)));
    HIP_TRY(ctx, f->d_differ.alloc(f->ncoef * (ms + 1)));
  }
  // the difference matrices of the levels that _begin clears and _finish adds (M >= mfirst[mdrft],
  // src/SphericalBasis.cc:1013-1079); a rank without particles still takes part in the reduction
  const int nl = ms - mfirst_mdrft + 1;
  if (!f->wd_clean) {      // (afterwards the contraction below leaves what it consumed zeroed)
    HIP_TRY(ctx, hipMemsetAsync(f->d_Wd.p, 0, f->d_Wd.bytes(), ctx->stream));
    f->wd_clean = true;
  }
  const SphDev S = dev_acc(f, c);
  size_t nr = 0;
  if (c->n) { int rc_ = expamd_comp_level_count(c, first, ms, &nr); if (rc_) return rc_; }
  // the step driver knows how many particles change level (c->mover_hint): their slots are compacted first
  // (k_mover_list, 16 slots per thread), so that the differencing launches over the movers, not over the range
  const bool listed = nr && c->mover_hint >= 0;
  if (listed && c->mover_hint > 0) { int rc_ = expamd_comp_mover_list(c, first, ms, (size_t)c->mover_hint); if (rc_) return rc_; }
  const bool thin_diff_on = EXPAMD_EXPT("EXP_AMD_THIN_DIFF", 1) != 0;
  const bool few = listed && c->mover_hint > 0 && !(ctx->mover_list_min >= 0 && c->mover_hint >= ctx->mover_list_min);
  const bool thin_diff = few && thin_diff_on && ctx->thin_max > 0 && c->mover_hint <= ctx->thin_max && !ctx->deterministic &&
                         !f->generic && f->ncoef <= 4096 && f->thin_lds_ok();
  // (no mover anywhere in the thin case: the moments were not touched, the partial sums are zero)
  if (listed && c->mover_hint == 0) {
    // nothing moved on this rank (it only takes part in the reduction)
  } else if (listed && ctx->mover_list_min >= 0 && c->mover_hint >= ctx->mover_list_min) {
    // many movers: through the accumulation kernel (AccList) -- runs of equal (level, cell) are summed in
    // registers, where the per-particle path sends 4 (L+1)^2 atomics per mover to a handful of addresses (4 % of 1e7
    // particles leaving level 0 in the first sweep: 140 ms that way, 1-3 ms this way)
    ProfScope ps(ctx, "k_sph_mstep_update");
    SphAccArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->mover_cnt, 0, 0, f->d_Wd.p, f->d_used.p + 1,
                 (size_t)c->mover_hint, ctx->stream, 1, nullptr, 1};
    a.list = c->mover_list.p;
    a.lev = c->level[c->cur].p;
    a.newlev = c->newlev.p;
    a.mfirst = mfirst_mdrft;
    a.nslices = c->mover_hint >= ctx->mover_slices_min ? ms + 2 : 2;
    sph_launch_acc(f, a);
  } else if (thin_diff) {
    // few movers, straight from the basis tables into the contraction's partial sums (k_sph_diff_thin): no staging, no
    // moments, no contraction over the cells of every level
    ProfScope ps(ctx, "k_sph_diff_thin");
    if (!f->part_clean) {
      HIP_TRY(ctx, hipMemsetAsync(f->d_part.p, 0, f->d_part.bytes(), ctx->stream));
      f->part_clean = true;
    }
    SphThinDiffArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->mover_list.p, c->mover_cnt,
                      c->level[c->cur].p, c->newlev.p, mfirst_mdrft, nl, f->d_wscale.p, f->d_part.p,
                      (size_t)c->mover_hint, ctx->stream};
    k_thin_diff_launch[cfg.lmax](a);
  } else if (nr) {
    ProfScope ps(ctx, "k_sph_mstep_update");
    SphUpdArgs a{S, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), c->level[c->cur].p, c->newlev.p,
                 listed ? c->mover_cnt : c->lev_off.p, first, ms, mfirst_mdrft, f->d_Wd.p,
                 listed ? (size_t)c->mover_hint : nr, ctx->stream};
    if (listed) {
      a.list = c->mover_list.p;
      // few movers: staged (values by plain stores, then one lane per value: see k_sph_mstep_update)
      int rc_ = sph_stage(f, (size_t)c->mover_hint, a);
      if (rc_) return rc_;
    }
    sph_launch_upd(f, a);
  }
  // moments -> coefficient differences, all levels in one launch
  if (!thin_diff)
    k_sph_contract<<<dim3(CSEG, S.nrows, nl), 256, 0, ctx->stream>>>(S, f->d_Wd.p + (size_t)mfirst_mdrft * wl,
                                                                    f->d_wscale.p, f->d_part.p, /*clear=*/1);
  // one packed all-reduce (src/SphericalBasis.cc:1063-1064), then expcoefN[M] += differ[M] -- in the summing kernel
  // itself when this rank is alone
  const bool alone = ctx->nranks <= 1 && !ctx->ar_fn;
  k_sph_sum_parts<<<dim3(cdiv(f->ncoef, 256), nl), 256, 0, ctx->stream>>>(
      f->d_part.p, (int)f->ncoef, f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef, nullptr,
      alone ? f->d_coefN.p + (size_t)mfirst_mdrft * f->ncoef : nullptr, /*clear=*/1);
  HIP_TRY(ctx, hipGetLastError());
  if (alone) return EXP_AMD_OK;
  const size_t cnt = (size_t)nl * f->ncoef;
  int rc = expamd_allreduce(ctx, f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef, cnt);
  if (rc) return rc;
  k_add_inplace<<<cdiv(cnt, 256), 256, 0, ctx->stream>>>(
      f->d_coefN.p + (size_t)mfirst_mdrft * f->ncoef, f->d_differ.p + (size_t)mfirst_mdrft * f->ncoef,
      cnt);
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// Beyond rmax the n-body force continues every (l,m) term as (rmax/r)^(l+1)
// (src/SphericalBasis.cc:1555-1560, :1605-1628); pyEXP's Spherical::computeAccel does not
// (expui/BiorthBasis.cc:818-926: the tables are simply evaluated at r/scale).  mode 1 = n-body
// (default), 0 = pyEXP.
// The small number added to r before any division: src/expand.H:130 DSMALL = 1e-16 in the n-body code
// (the default); pyEXP's Spherical::accumulate uses 1e-20 and computeAccel 1e-18 (expui/BiorthBasis.cc:
// 588, :824-825).  It only matters at the origin and ON the polar axis, where cos(theta) = z / (|z| +
// dsmall) is or is not exactly 1: sin(theta) = 2.6e-8 against 0.
extern "C" int exp_amd_sph_set_dsmall(exp_amd_force *fb, double dsmall)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !(dsmall >= 0.0)) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_dsmall: not a sphereSL force");
  f->dev.dsmall = dsmall;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sph_set_accumulate_all_m(exp_amd_force *fb, int all_m)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_accumulate_all_m: not a sphereSL force");
  f->dev.M0_acc = all_m ? 0 : f->dev.M0_only;
  return EXP_AMD_OK;
}

// FIX_L0 (src/SphericalBasis.cc:119): on -> the next force evaluation saves the l = 0 row; off -> forgotten
extern "C" int exp_amd_sph_set_fix_l0(exp_amd_force *fb, int on)
{
  expamd_mutated();
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_fix_l0: not a spherical force");
  if (on && !f->d_c0.p && f->d_c0.alloc((size_t)f->cfg.nmax) != hipSuccess)
    return expamd_fail(f->ctx, EXP_AMD_ERR_HIP, "set_fix_l0: hipMalloc failed");
  f->fix_l0 = on != 0;
  f->accel_writes_coef = f->fix_l0 || f->noise_on;
  f->have_c0 = false;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sph_set_exterior(exp_amd_force *fb, int continuation)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "set_exterior: not a sphereSL force");
  f->dev.no_exterior = continuation ? 0 : 1;
  return EXP_AMD_OK;
}
