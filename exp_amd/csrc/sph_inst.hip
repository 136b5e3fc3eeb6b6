// One translation unit per LMAX (compiled with -DSPH_L=<k>): instantiates the accumulate and
// force kernels for that harmonic order so that the orders build in parallel.
#include "sph_kernels.h"

#ifndef SPH_L
#error "compile with -DSPH_L=<lmax>"
#endif

#ifndef ACC_BLOCKS_TARGET
#define ACC_BLOCKS_TARGET 3072     // blocks a large launch is cut into (ACC_CHUNK_MAX permitting)
#endif
#ifdef EXPT_TIMING
__device__ unsigned long long g_dbg_s[16];
extern "C" int exp_amd_debug_sph_read(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_s), sizeof(g_dbg_s)); }
extern "C" int exp_amd_debug_sph_zero() { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_s), z, sizeof(z)); }
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

void CAT(expamd_sph_acc_L, SPH_L)(const SphAccArgs &a)
{
  constexpr int LMAX = SPH_L;
  constexpr int NS = acc_nsplit<LMAX>();
  constexpr int CPB = (ACC_WAVES >= NS) ? ACC_WAVES / NS : 1;
  // chunk: as large as ACC_CHUNK_MAX while >= ~6 rounds of blocks remain (2 blocks x 256 CUs).
  // Per level: a sparse multistep level (a few particles per cell: one flush per cell change,
  // serial within a wave) gets short chunks and many waves.
  LevChunks LC;
  LC.lo = a.lo;
  LC.nlev = a.hi - a.lo + 1;
  unsigned nb = 0;
  for (int j = 0; j < LC.nlev; j++) {
    const size_t n = a.counts ? a.counts[j] : a.n;
    size_t chunk = (n / ((size_t)CPB * ACC_BLOCKS_TARGET)) & ~(size_t)63;
    const size_t cmin_env = (size_t)EXPAMD_EXPT("EXP_AMD_ACC_CHUNK_MIN", 0) & ~(size_t)63;
    const size_t cmin = cmin_env ? cmin_env : ACC_CHUNK_MIN;
    chunk = chunk < cmin ? cmin : chunk > ACC_CHUNK_MAX ? ACC_CHUNK_MAX : chunk;
    LC.bstart[j] = nb;
    LC.chunk[j] = (int)chunk;
    nb += cdiv(cdiv(n, chunk), CPB);
  }
  LC.bstart[LC.nlev] = nb;
  if (nb == 0) return;
  dim3 grid(nb, (NS > ACC_WAVES) ? cdiv(NS, ACC_WAVES) : 1);
  if (a.list) {             // level-change differencing through the list of movers
    grid.z = a.nslices;
    const AccList al{a.list, a.lev, a.newlev, a.mfirst, a.S.numr - 1, a.nslices > 2 ? 1 : 0};
    if (a.S.detC != 0.0)
      k_sph_accumulate<LMAX, true, true><<<grid, ACC_WAVES * 64, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off,
                                                                              LC, a.W, a.used, a.wlevels, al);
    else
      k_sph_accumulate<LMAX, false, true><<<grid, ACC_WAVES * 64, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off,
                                                                               LC, a.W, a.used, a.wlevels, al);
    return;
  }
  if (a.S.detC != 0.0)      // deterministic mode (exp_amd_ctx_set_deterministic): order-independent sums
    k_sph_accumulate<LMAX, true><<<grid, ACC_WAVES * 64, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off,
                                                                        LC, a.W, a.used, a.wlevels);
  else if (a.S.frz)         // rtrunc set (Component::freeze): the instantiation that carries the test
    k_sph_accumulate<LMAX, false, false, true><<<grid, ACC_WAVES * 64, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off,
                                                                                     LC, a.W, a.used, a.wlevels);
  else
    k_sph_accumulate<LMAX, false, false, false><<<grid, ACC_WAVES * 64, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off,
                                                                                      LC, a.W, a.used, a.wlevels);
}

#ifndef SPH_GENERAL_GRID
#define SPH_GENERAL_GRID 4096u      // blocks of the general pass behind a fast pass (4 work items each per round)
#endif
void CAT(expamd_sph_force_L, SPH_L)(const SphForceArgs &a)
{
  constexpr int LMAX = SPH_L;
  // the logarithmic map only: the passes below leave the particles far outside the table on a list (SphDev::lit_list),
  // the literal pass at the end takes them
  struct LitPass {
    const SphForceArgs &a;
    explicit LitPass(const SphForceArgs &a_) : a(a_)
    {
      if (a.S.lit_list) (void)hipMemsetAsync(a.S.lit_list, 0, sizeof(uint32_t), a.stream);
    }
    ~LitPass()
    {
      if (!a.S.lit_list) return;
      ProfScope ps(a.ctx, "k_sph_force_literal");
      k_sph_force<LMAX, 3><<<64, 256, 0, a.stream>>>(
          a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
          a.dt_kick, a.assign, nullptr, nullptr, a.key_out, a.nk_dtk, a.nk_dtd, a.store_v, nullptr);
    }
  } lit_pass(a);
  if (a.app) {
    // the append step: fast pass and general pass that place their results in the next step's order (sph_kernels.h: AppDev)
    {
      ProfScope ps(a.ctx, "k_sph_force");
      if (a.app->AX)
        k_sph_force<LMAX, 1, 1><<<a.grid, 256, 0, a.stream>>>(
            a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
            a.dt_kick, a.assign, a.work, a.nwork, nullptr, a.nk_dtk, a.nk_dtd, a.store_v, nullptr, *a.app);
      else        // the lean payload: neither acceleration nor potential placed (AppDev)
        k_sph_force<LMAX, 1, 2><<<a.grid, 256, 0, a.stream>>>(
            a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
            a.dt_kick, a.assign, a.work, a.nwork, nullptr, a.nk_dtk, a.nk_dtd, a.store_v, nullptr, *a.app);
    }
    ProfScope ps(a.ctx, "k_sph_force_general");
    const unsigned ggrid = a.grid < SPH_GENERAL_GRID ? a.grid : SPH_GENERAL_GRID;
    k_sph_force<LMAX, 0, 1><<<ggrid, 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
        a.dt_kick, a.assign, a.work, a.nwork, nullptr, a.nk_dtk, a.nk_dtd, a.store_v, a.nwork_next, *a.app);
    return;
  }
  if (!a.all_slow) {
    // a.nwork[0..1]: two work-list counters used alternately; the general pass of launch k clears the
    // one launch k+1 will count into (no memset between the launches)
    {
      ProfScope ps(a.ctx, "k_sph_force");
      if (a.waterfall)
        k_sph_force<LMAX, 2><<<cdiv(a.grid, SPH_FORCE_CHUNKS), 256, 0, a.stream>>>(
            a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
            a.dt_kick, a.assign, a.work, a.nwork, a.key_out, a.nk_dtk, a.nk_dtd, a.store_v, nullptr);
      else
        k_sph_force<LMAX, 1><<<cdiv(a.grid, SPH_FORCE_CHUNKS), 256, 0, a.stream>>>(
            a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
            a.dt_kick, a.assign, a.work, a.nwork, a.key_out, a.nk_dtk, a.nk_dtd, a.store_v, nullptr);
    }
    // deferred waves / lanes: a fixed grid walks the work list (its length is only known on the device)
    ProfScope ps(a.ctx, "k_sph_force_general");
    const unsigned ggrid = a.grid < SPH_GENERAL_GRID ? a.grid : SPH_GENERAL_GRID;
    k_sph_force<LMAX, 0><<<ggrid, 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
        a.dt_kick, a.assign, a.work, a.nwork, a.key_out, a.nk_dtk, a.nk_dtd, a.store_v, a.nwork_next);
  } else if (a.stage_rows > 0 && a.dt_kick == 0.0 && !a.key_out) {
    // a target that is not in this basis' cell order (another component: interactions), no fused kick: the general
    // evaluation with the rows of each block's cell range staged in LDS
    ProfScope ps(a.ctx, "k_sph_force_staged");
    const int tq = 4 * a.S.trows;
    const int tqs = tq + ((2 - tq % 16) + 16) % 16;
    k_sph_force_staged<LMAX><<<a.grid, 256, (size_t)a.stage_rows * tqs * sizeof(double), a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ, a.assign,
        a.stage_rows, tqs, a.work, a.nwork);
    // ... and its special lanes (a.work: this evaluation's own list, sph.hip)
    const unsigned ggrid = a.grid < 256u ? a.grid : 256u;
    k_sph_force<LMAX, 0><<<ggrid, 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
        0.0, a.assign, a.work, a.nwork, nullptr, 0.0, 0.0, 1, a.nwork_next);
  } else {
    ProfScope ps(a.ctx, "k_sph_force_general");
    k_sph_force<LMAX, 0><<<a.grid, 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
        a.dt_kick, a.assign, nullptr, nullptr, a.key_out, a.nk_dtk, a.nk_dtd, a.store_v, nullptr);
  }
}

void CAT(expamd_sph_upd_L, SPH_L)(const SphUpdArgs &a)
{
  if (a.stage) {
    k_sph_mstep_update<SPH_L, true><<<cdiv(a.n, 256), 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.M, a.lev, a.newlev, a.lev_off, a.first, a.last, a.mfirst, a.Wd, a.plain, a.used, a.list,
        a.stage, a.keys);
    const int nval = a.S.nrows * 2;
    k_mstep_apply<><<<cdiv(a.n * (size_t)nval, 256), 256, 0, a.stream>>>(a.stage, a.keys, a.list ? a.lev_off : nullptr,
                                                                        (uint32_t)a.n, nval, a.Wd);
    return;
  }
  k_sph_mstep_update<SPH_L><<<cdiv(a.n, 256), 256, 0, a.stream>>>(
      a.S, a.X, a.Y, a.Z, a.M, a.lev, a.newlev, a.lev_off, a.first, a.last, a.mfirst, a.Wd, a.plain, a.used, a.list);
}

// thin active sets (sph_kernels.h): tile sizes from what fits the LDS next to the coefficient set / the products
void CAT(expamd_sph_thin_force_L, SPH_L)(const SphThinForceArgs &a)
{
  constexpr int LMAX = SPH_L;
  const int tq = 4 * a.S.trows;
  const int tqs = tq + ((2 - tq % 16) + 16) % 16;
  const size_t ncoef = ((size_t)a.S.nrows * a.S.nmax + 1) & ~(size_t)1;
  const size_t lsn = (size_t)(a.S.lmax + 1) * a.S.nmax;
  // (small tiles: a thin range is a few hundred to a few thousand particles and the GPU has a thousand SIMDs -- what
  // counts is the length of a block's chain of dependent phases, not the number of blocks)
  const int tp0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TP", 4);
  int tp = tp0 < 1 ? 1 : tp0 > 64 ? 64 : tp0;
  while (tp > 4 && (ncoef + (size_t)tp * (tqs + 3 * lsn)) * sizeof(double) > 100 * 1024) tp >>= 1;
  const size_t lds = (ncoef + (size_t)tp * (tqs + 3 * lsn)) * sizeof(double);
  size_t grid = cdiv(a.n, (size_t)tp);
  if (grid > 16384) grid = 16384;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_sph_force_thin<LMAX>), "k_sph_force_thin<LMAX>");
  (void)big;
  const int nt0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_NT", 0);
  const int nt = nt0 ? nt0 : 256;       // (64-thread blocks -- four times as many resident -- measured SLOWER at 2e3 and 1.3e4 particles)
  k_sph_force_thin<LMAX><<<(unsigned)grid, nt, lds, a.stream>>>(
      a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.coef, a.rowmap, a.tscale, a.AX, a.AY, a.AZ, a.POT, a.VX, a.VY, a.VZ,
      a.assign, tp, tqs);
}

void CAT(expamd_sph_thin_acc_L, SPH_L)(const SphThinAccArgs &a)
{
  constexpr int LMAX = SPH_L;
  const size_t nrows = (size_t)a.S.nrows, lsn = (size_t)(a.S.lmax + 1) * a.S.nmax;
  const int tpa0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TPA", 8);
  int tpa = tpa0 < 1 ? 1 : tpa0 > 64 ? 64 : tpa0;
  auto need = [&](int t) { return ((((size_t)t * nrows + 1) & ~(size_t)1) + (size_t)t * lsn) * sizeof(double); };
  while (tpa > 4 && need(tpa) > 96 * 1024) tpa >>= 1;
  size_t grid = cdiv(a.n, (size_t)tpa);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_sph_acc_thin<LMAX>), "k_sph_acc_thin<LMAX>");
  (void)big;
  k_sph_acc_thin<LMAX><<<(unsigned)grid, 256, need(tpa), a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off, a.lo, a.hi, a.wscale,
                                                                    a.part, a.used, tpa, a.adv);
}

void CAT(expamd_sph_thin_diff_L, SPH_L)(const SphThinDiffArgs &a)
{
  constexpr int LMAX = SPH_L;
  const size_t nrows = (size_t)a.S.nrows, lsn = (size_t)(a.S.lmax + 1) * a.S.nmax;
  int tpa = 8;
  auto need = [&](int t) { return ((((size_t)t * nrows + 1) & ~(size_t)1) + (size_t)t * lsn) * sizeof(double); };
  while (tpa > 4 && need(tpa) > 96 * 1024) tpa >>= 1;
  size_t grid = cdiv(a.n, (size_t)tpa);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_sph_diff_thin<LMAX>), "k_sph_diff_thin<LMAX>");
  (void)big;
  k_sph_diff_thin<LMAX><<<(unsigned)grid, 256, need(tpa), a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.list, a.cnt, a.lev, a.newlev,
                                                                     a.mfirst, a.nlev_out, a.wscale, a.part, tpa);
}
