// Cylindrical BFE force method (cylinder / EmpCylSL) for gfx950, from scratch.
//
// Reference per particle (CPU): EmpCylSL::accumulate -> get_pot (exputil/EmpCylSL.cc:4049-4146,
// :5557-5631) bilinearly interpolates potC/potS[m][n] at the particle's (X,Y) cell for every (m,n)
// (4 x 156 table reads at mmax 6, nmax 12); accumulated_eval (:5256-5410) interpolates six
// tables per (m,n) (4 x 468 reads).  The interpolation is LINEAR in the four corner values of the
// cell, so the n-sum commutes with the particle sum exactly as in the spherical case:
//
//   accumulate : Wn[node][j] += -4pi m trig_j(phi) c_k      node = corner k of the particle's cell,
//                                                           trig_j = cos(m phi) | sin(m phi)
//                cos[m][n]    = sum_node potC[m][n][node] Wn[node][cos m]          (once per step)
//   force      : TF[node][m]  = { sum_n cos[m][n] {potC,rforceC,zforceC}[m][n][node],
//                                 sum_n sin[m][n] {potS,rforceS,zforceS}[m][n][node] }   (once)
//                p, fr, fz, fp from the bilinear blend of 3(2 mmax+1) node values   (per particle)
//
// Particles are kept sorted by (X,Y) cell, so a wave shares its four corner rows (scalar loads) and
// its 4(2 mmax+1) moment sums stay in registers until the cell changes.
#include <cstring>
#include "sort_kernels.h"
#include "thin_adv.h"
#include "force.h"

#include <type_traits>

#define DSMALL 1.0e-16        // src/expand.H:130
#define CYL_MAX_M 12            // azimuthal orders with unrolled kernels; above: k_cyl_moments_gen / k_cyl_force_gen
#define CYL_GEN_MAX_M 64        // ... up to this order

typedef const __attribute__((address_space(4))) double *cdp;

struct CylDev {
  int mmax, nmax, numx, numy, cmapr, cmapz, EVEN_M, ntrig;
  double ascale, hscale, rtable, xmin, dx, ymin, dy, rmax2;
  // reciprocals and products of the above, so that the per-particle code multiplies where the
  // reference divides by a constant (last-ulp differences only; the bilinear blend is continuous
  // across cell edges): 1/ascale, 1/hscale, 1/dx, 1/dy, rtable*ascale
  double inv_ascale, inv_hscale, inv_dx, inv_dy, rtab_abs, inv_rtab_abs;
  double cx, cy, cz;
  // Orient::transformBody of the component the basis belongs to (src/Cylinder.cc:799, :1352);
  // forces go back through its transpose, transformOrig (:1418)
  int use_rot;
  double rot[9];
  PseudoDev ps;     // frame acceleration of the TARGET component (force pass only)
  // deterministic mode: rounding-grid constants of the moment terms / of the in-cut mass (0: off)
  double detC, detCm;
  double umass;     // != 0: every particle of the component has this mass (accumulate does not read the stream)
};

// centred, then rotated into the body frame
__device__ __forceinline__ void cyl_local(const CylDev &C, double x, double y, double z, double &xx,
                                          double &yy, double &zz)
{
  xx = x - C.cx; yy = y - C.cy; zz = z - C.cz;
  if (C.use_rot) {
    const double a = xx, b = yy, c = zz;
    xx = C.rot[0] * a + C.rot[1] * b + C.rot[2] * c;
    yy = C.rot[3] * a + C.rot[4] * b + C.rot[5] * c;
    zz = C.rot[6] * a + C.rot[7] * b + C.rot[8] * c;
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void cstatic_for(F &&f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cstatic_for<I + 1, N>(f);
  }
}

// exputil/EmpCylSL.cc:6446-6463
__device__ __forceinline__ double cyl_r_to_xi(const CylDev &C, double r)
{
  if (C.cmapr > 0) { const double u = r * C.inv_ascale; return div_fast(u - 1.0, u + 1.0); }
  return r;
}

// exputil/EmpCylSL.cc:7109-7117
__device__ __forceinline__ double cyl_z_to_y(const CylDev &C, double z)
{
  if (C.cmapz == 1) return copysign(asinh_pos(fabs(z) * C.inv_hscale), z);     // sign(z) asinh|z/h|
  if (C.cmapz == 2) {
    double g, y;
    sqrt_rsqrt(z * z + C.hscale * C.hscale, g, y);
    return z * y;
  }
  return z;
}

// cell and bilinear weights (exputil/EmpCylSL.cc:5567-5597 == :5280-5314), enforce_limits false
__device__ __forceinline__ void cyl_weights(const CylDev &C, double r, double z, int &ix, int &iy,
                                            double &c00, double &c10, double &c01, double &c11)
{
  const double X = (cyl_r_to_xi(C, r) - C.xmin) * C.inv_dx;
  const double Y = (cyl_z_to_y(C, z) - C.ymin) * C.inv_dy;
  ix = (int)X;
  iy = (int)Y;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix >= C.numx) ix = C.numx - 1;
  if (iy >= C.numy) iy = C.numy - 1;
  const double delx0 = (double)ix + 1.0 - X, dely0 = (double)iy + 1.0 - Y;
  const double delx1 = X - (double)ix, dely1 = Y - (double)iy;
  c00 = delx0 * dely0;
  c10 = delx1 * dely0;
  c01 = delx0 * dely1;
  c11 = delx1 * dely1;
}

// sort key: level * (ncell+1) + cell, cell = ix*numy + iy; off-grid particles share bin ncell
struct CylKeyFn {
  CylDev C;
  uint32_t sparse_mask;      // levels that are not cell-sorted: all their particles share bin 0
  __device__ __forceinline__ uint32_t operator()(double x, double y, double z, uint8_t lev) const
  {
    if ((sparse_mask >> lev) & 1u) return (uint32_t)lev * ((uint32_t)(C.numx * C.numy) + 1u);
    double xx, yy, zz;
    cyl_local(C, x, y, z, xx, yy, zz);
    const double r2 = xx * xx + yy * yy;
    double r, ir_, r3, ir3_;
    sqrt_rsqrt(r2, r, ir_);
    sqrt_rsqrt(r2 + zz * zz, r3, ir3_);
    const uint32_t ncell = (uint32_t)(C.numx * C.numy);
    uint32_t cell = ncell;
    if (!(r3 > C.rtab_abs)) {
      int ix, iy;
      double a, b, c, d;
      cyl_weights(C, r, zz, ix, iy, a, b, c, d);
      cell = (uint32_t)(ix * C.numy + iy);
    }
    return (uint32_t)lev * (ncell + 1u) + cell;
  }
};

// ---- accumulation ----------------------------------------------------------------------------------

// Work split of an accumulation launch over several time-step levels: the blocks [bstart[j],
// bstart[j+1]) take level lo + j in chunks of chunk[j] particles per wave (no chunk crosses a level,
// every level gets a chunk size that suits its own population).  nlev = 1: the classic launch.
#define LEVCHUNK_MAX 17
struct LevChunks {
  int lo, nlev;
  unsigned bstart[LEVCHUNK_MAX + 1];
  int chunk[LEVCHUNK_MAX];
};

#define CFLUSH_STRIDE 68
#define CACC_WAVES 4
#ifndef CACC_OCC
#define CACC_OCC 2          // waves per SIMD asked of the compiler
#endif
#ifndef CACC_CHUNK_MAX
#define CACC_CHUNK_MAX 1024   // particles per wave chunk; sparse multistep levels get shorter ones
#endif
#define CACC_THICK_MIN 1000000u   // level population from which a multistep level is accumulated apart from thinner ones
#define CYL_TAILS 128             // slot pairs the {in-cut mass, count} tallies of an accumulation launch are spread over

// Deterministic (order-independent) accumulation, as in sph_kernels.h: every term is rounded to a fixed
// absolute grid 2^e first, (w*p + C) - C with C = 1.5 * 2^(52+e), so that all later additions are exact.
template <bool DET>
__device__ __forceinline__ void cacc_add(double &a, double w, double p, double C)
{
  if constexpr (DET) {
    double t = fma(w, p, C);
    t -= C;
    a += t;
  } else {
    a = fma(w, p, a);
  }
}
__device__ __forceinline__ double cdet_round(double v, double C) { return C != 0.0 ? (v + C) - C : v; }

// reduce NV per-lane values over the wave and atomically add value j to dst[map(j)]
// (CNT < NV: only the first CNT values -- they alone are reduced, added and zeroed)
template <int NV, int CNT = NV, class MapFn>
__device__ __forceinline__ void cyl_wave_flush(double (&v)[NV], double *scratch, double *dst,
                                               MapFn map)
{
  const int lane = threadIdx.x & 63;
  const int kk = lane >> 2, q = lane & 3;
  cstatic_for<0, (CNT + 15) / 16>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    cstatic_for<0, 16>([&](auto jc) {
      constexpr int j = g * 16 + decltype(jc)::value;
      if constexpr (j < CNT) scratch[decltype(jc)::value * CFLUSH_STRIDE + lane] = v[j];
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
    if (g * 16 + kk < CNT) {
#pragma unroll
      for (int e = 0; e < 16; e++) s += scratch[kk * CFLUSH_STRIDE + q + 4 * e];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q == 0 && g * 16 + kk < CNT && s != 0.0) unsafeAtomicAdd(dst + map(g * 16 + kk), s);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  });
#pragma unroll
  for (int j = 0; j < CNT; j++) v[j] = 0.0;
}

// Profiling aid (tools/dbg/cyl_timing.py; tools/build_variant_cyl.sh timing -DEXPT_TIMING): s_memtime counters of
// the phases of a 64-particle group in k_cyl_accumulate -- waiting for outstanding memory operations at its
// top, the per-particle inputs, the moment sums, the flushes and the time until everything outstanding is
// back after one -- summed over all waves.
#ifdef EXPT_TIMING
__device__ unsigned long long g_dbg_t[8];
extern "C" int exp_amd_debug_read(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_t), sizeof(g_dbg_t)); }
extern "C" int exp_amd_debug_zero() { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_t), z, sizeof(z)); }
#define TSTAMP() __builtin_readcyclecounter()
#endif
// LIST mode: the level-change differencing of MANY movers (multistep_update, src/CylEXP.cc:159-188) through this
// kernel.  The particles are taken through a list of mover slots (k_mover_list: slot order, i.e. by (level, cell)
// where the store is cell-sorted) and blockIdx.z selects what a launch slice adds: z = 0 subtracts every mover
// from Wn[its level] (levels >= mfirst only), z = 1 + T adds the movers whose proposed level is T to Wn[T].  Runs
// of equal (level, cell) are summed in registers as in the plain accumulation; window: on the table only.
struct CylAccList {
  const uint32_t *list;
  const uint8_t *lev, *newlev;
  int mfirst;
  int per_level;                // 1: one adding slice per proposed level (z = 1 + T); 0: ONE adding slice (z = 1)
};
__device__ __forceinline__ void cyl_list_fetch(const CylAccList &al, const double *__restrict__ X,
                                               const double *__restrict__ Y, const double *__restrict__ Z,
                                               const double *__restrict__ M, double umass, size_t ip,
                                               double &x, double &y, double &z, double &m, int &lv)
{
  const uint32_t j = al.list[ip];
  const int fr = al.lev[j], to = al.newlev[j];
  const int slice = blockIdx.z;
  if (slice == 0) lv = fr >= al.mfirst ? fr : -1;
  else lv = (!al.per_level || to == slice - 1) ? to : -1;
  x = X[j]; y = Y[j]; z = Z[j];
  const double mm = umass != 0.0 ? umass : M[j];
  m = slice == 0 ? -mm : mm;
}

// Wn[node][ntrig]: trig slot 0 = m0, 2m-1 = cos m, 2m = sin m
template <int MMAX, bool DET, bool LIST = false>
__global__ void __launch_bounds__(CACC_WAVES * 64, CACC_OCC)
k_cyl_accumulate(CylDev C, const double *__restrict__ X, const double *__restrict__ Y,
                 const double *__restrict__ Z, const double *__restrict__ M,
                 const uint32_t *__restrict__ lev_off, LevChunks LC,
                 double *__restrict__ Wn, double *__restrict__ tail,
                 int multilevel /* Wn[level][node][ntrig] */, CylAccList al = CylAccList{})
{
  // which level this block works on, and with which chunk size (block-uniform: scalar loop)
  int lj = 0;
  while (lj + 1 < LC.nlev && blockIdx.x >= LC.bstart[lj + 1]) lj++;
  const int lev_lo = LC.lo + lj, lev_hi = lev_lo;
  const int CACC_CHUNK = LC.chunk[lj];
  const unsigned bx = blockIdx.x - LC.bstart[lj];
  const int lvl = multilevel ? lev_lo : 0;
  constexpr int NT = 2 * MMAX + 1;
  constexpr int NV = 4 * NT;
  __shared__ double scratch_all[CACC_WAVES][16 * CFLUSH_STRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *scratch = scratch_all[wave];
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const size_t cbeg = beg + ((size_t)bx * CACC_WAVES + wave) * CACC_CHUNK;
  if (cbeg >= end) return;
  const size_t cend = (cbeg + CACC_CHUNK < end) ? cbeg + CACC_CHUNK : end;
  const double norm = -4.0 * M_PI;
  const int nyp = C.numy + 1;

  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) acc[j] = 0.0;
  int cur = -1;
  double mass_used = 0.0, n_used = 0.0;

  const int ncellT = C.numx * C.numy;
  const size_t wlev = (size_t)(C.numx + 1) * nyp * NT;
  // next == key + 1 inside one column of cells (same ix, next iy; the particles are in cell order, so that is the usual
  // change): the two upper corners of the old cell ARE the two lower corners of the new one -- only the lower two
  // (corners 00, 10: the first 2 NT sums) are reduced and added, the upper two move down and keep accumulating.  Half
  // the work per cell change; on a thick disk (a hundred particles per cell) the flushes were 58 % of this kernel.
  auto flush = [&](int key, int next) {
    const int L = key / ncellT, cell = key - L * ncellT;       // (L = 0 in single-level launches)
    const int ix = cell / C.numy, iy = cell - ix * C.numy;
    double *base = Wn + (size_t)L * wlev + ((size_t)ix * nyp + iy) * NT;
    auto map = [&](int j) {
      const int k = j / NT, t = j - k * NT;                // corner k: 0=00, 1=10, 2=01, 3=11
      return (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT + t;
    };
    if (next == key + 1 && iy + 1 < C.numy) {
      cyl_wave_flush<NV, 2 * NT>(acc, scratch, base, map);
#pragma unroll
      for (int j = 0; j < 2 * NT; j++) { acc[j] = acc[2 * NT + j]; acc[2 * NT + j] = 0.0; }
    } else
      cyl_wave_flush<NV>(acc, scratch, base, map);
  };

  // software prefetch: the loads of group k+1 are in flight while group k is reduced (two groups ahead
  // measured the same: the waves wait on their own dependent fp64 chains, not on these loads)
  double nx = 0, ny = 0, nz = 0, nm = 0;
  int nlv = lvl;                            // LIST: per entry (< 0: not in this slice)
  const bool um = C.umass != 0.0;
  if (cbeg + lane < cend) {
    if constexpr (LIST) cyl_list_fetch(al, X, Y, Z, M, C.umass, cbeg + lane, nx, ny, nz, nm, nlv);
    else { nx = X[cbeg + lane]; ny = Y[cbeg + lane]; nz = Z[cbeg + lane]; nm = um ? C.umass : M[cbeg + lane]; }
  }
#ifdef EXPT_TIMING
  unsigned long long t_load = 0, t_prep = 0, t_red = 0, t_fl = 0, t_fld = 0, t_nfl = 0, t_all0 = TSTAMP();
#endif
  for (size_t base = cbeg; base < cend; base += 64) {
    const size_t i = base + lane;
    const bool valid = LIST ? (i < cend && nlv >= 0) : i < cend;
    const int plv = LIST ? nlv : lvl;
    double xx = 1, yy = 0, zz = 0, mass = 0;
#ifdef EXPT_TIMING
    const unsigned long long ta = TSTAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tb = TSTAMP();
    t_load += tb - ta;
#endif
    if (valid) {
      cyl_local(C, nx, ny, nz, xx, yy, zz);
      mass = nm;
    }
    if (i + 64 < cend) {
      if constexpr (LIST) cyl_list_fetch(al, X, Y, Z, M, C.umass, i + 64, nx, ny, nz, nm, nlv);
      else { nx = X[i + 64]; ny = Y[i + 64]; nz = Z[i + 64]; nm = um ? C.umass : M[i + 64]; }
    }
    // src/Cylinder.cc:853-866 (the differencing has no rcylmax cut and counts nothing: src/CylEXP.cc:159-188)
    const double r2 = xx * xx + yy * yy;
    double r, ir, rr, irr;
    sqrt_rsqrt(r2, r, ir);
    const bool incut = LIST ? valid : (valid && (r2 + zz * zz) < C.rmax2);
    if (!LIST && incut) { mass_used += cdet_round(mass, C.detCm); n_used += 1.0; }
    // EmpCylSL::accumulate (:4062-4063)
    sqrt_rsqrt(r2 + zz * zz, rr, irr);
    const bool ongrid = incut && !(rr > C.rtab_abs);
    double zc = zz;                                         // get_pot z clamp (:5563-5564)
    if (zc > C.rtab_abs) zc = C.rtab_abs;
    if (zc < -C.rtab_abs) zc = -C.rtab_abs;
    int ix, iy;
    double c00, c10, c01, c11;
    cyl_weights(C, r, zc, ix, iy, c00, c10, c01, c11);
    const int cell = ix * C.numy + iy + plv * ncellT;
    double cphi = 1.0, sphi = 0.0;                          // phi = atan2(y, x)
    if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
    const double t0 = ongrid ? norm * mass : 0.0;
#ifdef EXPT_TIMING
    asm volatile("" :: "v"(t0), "v"(c00), "v"(c11), "v"(cphi), "v"(sphi));
    const unsigned long long tc = TSTAMP();
    t_prep += tc - tb;
#endif

    unsigned long long remaining = __ballot(ongrid);
    while (remaining) {
      const int lead = __ffsll((long long)remaining) - 1;
      const int c = __builtin_amdgcn_readlane(cell, lead);
      const bool sel = ongrid && cell == c;
      if (c != cur) {
#ifdef EXPT_TIMING
        const unsigned long long tf0 = TSTAMP();
        if (cur >= 0) { flush(cur, c); t_nfl++; }
        const unsigned long long tf1 = TSTAMP();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the flush's atomics AND the prefetched loads)
        t_fl += tf1 - tf0; t_fld += TSTAMP() - tf1;
#else
        if (cur >= 0) flush(cur, c);
#endif
        cur = c;
      }
      const double t = sel ? t0 : 0.0;
      const double w0 = t * c00, w1 = t * c10, w2 = t * c01, w3 = t * c11;
      double cm = 1.0, sm = 0.0;
      cstatic_for<0, MMAX + 1>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m > 0) {
          const double cn = cm * cphi - sm * sphi;          // cos(m phi), sin(m phi)
          const double sn = sm * cphi + cm * sphi;
          cm = cn; sm = sn;
        }
        const bool on = !(C.EVEN_M && (m & 1));             // get_pot skips odd m (:5601)
        if (on) {
          constexpr int jc = (m == 0) ? 0 : 2 * m - 1;
          cacc_add<DET>(acc[0 * NT + jc], w0, cm, C.detC);
          cacc_add<DET>(acc[1 * NT + jc], w1, cm, C.detC);
          cacc_add<DET>(acc[2 * NT + jc], w2, cm, C.detC);
          cacc_add<DET>(acc[3 * NT + jc], w3, cm, C.detC);
          if constexpr (m > 0) {
            cacc_add<DET>(acc[0 * NT + jc + 1], w0, sm, C.detC);
            cacc_add<DET>(acc[1 * NT + jc + 1], w1, sm, C.detC);
            cacc_add<DET>(acc[2 * NT + jc + 1], w2, sm, C.detC);
            cacc_add<DET>(acc[3 * NT + jc + 1], w3, sm, C.detC);
          }
        }
      });
      remaining &= ~__ballot(sel);
    }
#ifdef EXPT_TIMING
    t_red += TSTAMP() - tc;
#endif
  }
#ifdef EXPT_TIMING
  if (lane == 0) {
    atomicAdd(&g_dbg_t[0], t_load); atomicAdd(&g_dbg_t[1], t_prep); atomicAdd(&g_dbg_t[2], t_red);
    atomicAdd(&g_dbg_t[3], TSTAMP() - t_all0); atomicAdd(&g_dbg_t[4], 1ull + (t_nfl << 32));
    atomicAdd(&g_dbg_t[5], (unsigned long long)((cend - cbeg + 63) / 64));
    atomicAdd(&g_dbg_t[6], t_fl); atomicAdd(&g_dbg_t[7], t_fld);
  }
#endif
  if (cur >= 0) flush(cur, -1);
  for (int off = 32; off > 0; off >>= 1) {
    mass_used += __shfl_xor(mass_used, off);
    n_used += __shfl_xor(n_used, off);
  }
  // {in-cut mass, count}: into one of CYL_TAILS slot pairs (summed and cleared by the contraction that follows).  Two
  // addresses for every wave of the launch serialise in the memory-side atomic unit at ~12 ns each: at 1e7 particles
  // the 19 532 atomics of the 9766 waves took 0.12 of this kernel's 0.32 ms -- as long as everything else it does.
  if (lane == 0 && n_used > 0.0) {
    double *tp = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
    unsafeAtomicAdd(tp + 0, mass_used);
    unsafeAtomicAdd(tp + 1, n_used);
  }
}

// ---- the same accumulation with the moment sums spread over the LANES of the wave ("slot" formulation) ---------
// k_cyl_accumulate keeps its 4 NT moment sums (52 doubles at mmax 6) in the registers of every lane and reduces them
// over the wave on each cell change; at two waves per SIMD that kernel issues in a third of its cycles and waits for
// memory in the rest.  Here a particle's contribution -- the rank-one product w_k trig_t, w_k = -4 pi m c_k (corner k),
// trig_t = cos | sin(m phi) -- is formed by the lane that OWNS the pair (cos m, sin m): lane = 8 sub + m
// (m <= MMAX <= 7 active), eight accumulators per lane (4 corners x {cos, sin}), the eight sub-groups taking every
// eighth particle of the 64-particle group.  The particle lanes leave w[4] and the trig pairs in LDS (one pass of the
// recurrences per particle), the owner lanes read them back (one 32-byte and one 16-byte broadcast read per particle and
// lane).  What that buys: no 52-register accumulator file (~100 VGPRs instead of ~230: three to four waves per SIMD and
// particle loads two groups ahead), and a cell change costs one 8-way LDS sum + ONE atomic instruction.
#ifndef CYL_SLOT_OCC
#define CYL_SLOT_OCC 3
#endif
#ifndef CSLOT_EXPT
#define CSLOT_EXPT 0
#endif
#define CSLOT_TSTRIDE 65          // trig rows [m][65] pairs: 1040 B apart, i.e. 4 banks per row for the 16-byte reads

template <int MMAX, bool DET, bool LIST = false>
__global__ void __launch_bounds__(CACC_WAVES * 64, CYL_SLOT_OCC)
k_cyl_accumulate_slot(CylDev C, const double *__restrict__ X, const double *__restrict__ Y,
                      const double *__restrict__ Z, const double *__restrict__ M,
                      const uint32_t *__restrict__ lev_off, LevChunks LC,
                      double *__restrict__ Wn, double *__restrict__ tail,
                      int multilevel /* Wn[level][node][ntrig] */, CylAccList al = CylAccList{})
{
  static_assert(MMAX <= 7, "slot formulation: one lane per azimuthal order, eight per sub-group");
  int lj = 0;
  while (lj + 1 < LC.nlev && blockIdx.x >= LC.bstart[lj + 1]) lj++;
  const int lev_lo = LC.lo + lj, lev_hi = lev_lo;
  const int CACC_CHUNK = LC.chunk[lj];
  const unsigned bx = blockIdx.x - LC.bstart[lj];
  const int lvl = multilevel ? lev_lo : 0;
  constexpr int NT = 2 * MMAX + 1;
  // per wave: w[64][4] (2 KB), trig pairs [MMAX+1][65][2], flush scratch [64][4] (2 KB)
  constexpr int WB = 64 * 4, TB = (MMAX + 1) * CSLOT_TSTRIDE * 2, SB = 64 * 4;
  __shared__ __attribute__((aligned(16))) double lds_all[CACC_WAVES][WB + TB + SB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *wbuf = lds_all[wave], *tbuf = wbuf + WB, *sbuf = tbuf + TB;
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const size_t cbeg = beg + ((size_t)bx * CACC_WAVES + wave) * CACC_CHUNK;
  if (cbeg >= end) return;
  const size_t cend = (cbeg + CACC_CHUNK < end) ? cbeg + CACC_CHUNK : end;
  const double norm = -4.0 * M_PI;
  const int nyp = C.numy + 1;
  const int sub = lane >> 3, om = lane & 7;            // owner role: sub-group, azimuthal order
  const bool owner = om <= MMAX && !(C.EVEN_M && (om & 1));
  const int orow = om <= MMAX ? om : 0;                // (idle lanes read row 0 and accumulate nothing)
  // trig row of m = 0: (1, 0) for every particle, once
  tbuf[(0 * CSLOT_TSTRIDE + lane) * 2 + 0] = 1.0;
  tbuf[(0 * CSLOT_TSTRIDE + lane) * 2 + 1] = 0.0;

  double a00 = 0, a01 = 0, a10 = 0, a11 = 0, a20 = 0, a21 = 0, a30 = 0, a31 = 0;    // a<corner><cos|sin>
  int cur = -1;
  double mass_used = 0.0, n_used = 0.0;
  const int ncellT = C.numx * C.numy;
  const size_t wlev = (size_t)(C.numx + 1) * nyp * NT;

  // sum the eight sub-groups' values of two corners (k0, k0 + 1) and add them to the node table: lanes j < 4 (MMAX+1),
  // j = 4 m + q, q = 2 (corner - k0) + cs
  auto flush_pair = [&](double *base, int k0, double v0, double v1, double v2, double v3) {
    double *mine = sbuf + lane * 4;
    mine[0] = v0; mine[1] = v1; mine[2] = v2; mine[3] = v3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 4 * (MMAX + 1)) {
      const int m = lane >> 2, q = lane & 3;
      double s = 0.0;
#pragma unroll
      for (int g = 0; g < 8; g++) s += sbuf[(g * 8 + m) * 4 + q];
      const int k = k0 + (q >> 1), cs = q & 1;
      if (s != 0.0 && !(m == 0 && cs)) {
        const int t = m == 0 ? 0 : 2 * m - 1 + cs;
        unsafeAtomicAdd(base + (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT + t, s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto flush = [&](int key, int next) {
    const int L = key / ncellT, cell = key - L * ncellT;
    const int ix = cell / C.numy, iy = cell - ix * C.numy;
    double *base = Wn + (size_t)L * wlev + ((size_t)ix * nyp + iy) * NT;
    flush_pair(base, 0, a00, a01, a10, a11);
    if (next == key + 1 && iy + 1 < C.numy) {           // next cell of the column: the upper corners move down
      a00 = a20; a01 = a21; a10 = a30; a11 = a31;
    } else {
      flush_pair(base, 2, a20, a21, a30, a31);
      a00 = a01 = a10 = a11 = 0.0;
    }
    a20 = a21 = a30 = a31 = 0.0;
  };

  // particle loads two groups ahead
  double nx[2] = {0, 0}, ny[2] = {0, 0}, nz[2] = {0, 0}, nm[2] = {0, 0};
  int nlv[2] = {lvl, lvl};
  const bool um = C.umass != 0.0;
  auto fetch = [&](size_t i, int slot) {
    if (i < cend) {
      if constexpr (LIST) cyl_list_fetch(al, X, Y, Z, M, C.umass, i, nx[slot], ny[slot], nz[slot], nm[slot], nlv[slot]);
      else { nx[slot] = X[i]; ny[slot] = Y[i]; nz[slot] = Z[i]; nm[slot] = um ? C.umass : M[i]; }
    }
  };
  fetch(cbeg + lane, 0);
  fetch(cbeg + 64 + lane, 1);
  int slot = 0;
  for (size_t base = cbeg; base < cend; base += 64, slot ^= 1) {
    const size_t i = base + lane;
    const bool valid = LIST ? (i < cend && nlv[slot] >= 0) : i < cend;
    const int plv = LIST ? nlv[slot] : lvl;
    double xx = 1, yy = 0, zz = 0, mass = 0;
    if (valid) {
      cyl_local(C, nx[slot], ny[slot], nz[slot], xx, yy, zz);
      mass = nm[slot];
    }
    fetch(i + 128, slot);
#if CSLOT_EXPT == 1 || CSLOT_EXPT == 5        // timing experiment: the particle stream alone
    mass_used += xx + yy + zz + mass; n_used += 1.0;
    continue;
#endif
    const double r2 = xx * xx + yy * yy;
    double r, ir, rr, irr;
    sqrt_rsqrt(r2, r, ir);
    const bool incut = LIST ? valid : (valid && (r2 + zz * zz) < C.rmax2);
    if (!LIST && incut) { mass_used += cdet_round(mass, C.detCm); n_used += 1.0; }
    sqrt_rsqrt(r2 + zz * zz, rr, irr);
    const bool ongrid = incut && !(rr > C.rtab_abs);
    double zc = zz;
    if (zc > C.rtab_abs) zc = C.rtab_abs;
    if (zc < -C.rtab_abs) zc = -C.rtab_abs;
    int ix, iy;
    double c00, c10, c01, c11;
    cyl_weights(C, r, zc, ix, iy, c00, c10, c01, c11);
    const int cell = ix * C.numy + iy + plv * ncellT;
    double cphi = 1.0, sphi = 0.0;
    if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
    const double t0 = ongrid ? norm * mass : 0.0;
#if CSLOT_EXPT == 2        // timing experiment: stream + per-particle inputs, no LDS, no sums
    mass_used += t0 * (c00 + c10 + c01 + c11) + cphi + sphi + (double)cell; continue;
#endif
    // this particle's row of the two LDS tables
    {
      double *w = wbuf + lane * 4;
      w[0] = t0 * c00; w[1] = t0 * c10; w[2] = t0 * c01; w[3] = t0 * c11;
      double cm = 1.0, sm = 0.0;
      cstatic_for<1, MMAX + 1>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const double cn = cm * cphi - sm * sphi;
        const double sn = sm * cphi + cm * sphi;
        cm = cn; sm = sn;
        tbuf[(m * CSLOT_TSTRIDE + lane) * 2 + 0] = cm;
        tbuf[(m * CSLOT_TSTRIDE + lane) * 2 + 1] = sm;
      });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    unsigned long long remaining = __ballot(ongrid);
    while (remaining) {
      const int lead = __ffsll((long long)remaining) - 1;
      const int c = __builtin_amdgcn_readlane(cell, lead);
      const unsigned long long mask = __ballot(ongrid && cell == c);
      if (c != cur) {
#if CSLOT_EXPT != 3        // (3: timing experiment without the flushes)
        if (cur >= 0) flush(cur, c);
#endif
        cur = c;
      }
      // owner pass over the particles of `mask`: iterations ia .. ib of eight particles each; the first and the last
      // (and every one when the run has holes: an un-sorted LIST slice) take the trig pair through the mask
      const int pa = lead, pb = 63 - __clzll((long long)mask);
      const bool holes = __popcll(mask) != pb - pa + 1;
      const unsigned long long ms = mask >> sub;                       // bit 8 i: particle 8 i + sub
      for (int it = pa >> 3; it <= (pb >> 3); it++) {
        const int p = it * 8 + sub;
        const double *w = wbuf + p * 4;
        const double w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        double tc = tbuf[(orow * CSLOT_TSTRIDE + p) * 2 + 0], tsn = tbuf[(orow * CSLOT_TSTRIDE + p) * 2 + 1];
        if (holes || it == (pa >> 3) || it == (pb >> 3)) {
          const bool in = (ms >> (8 * it)) & 1ull;
          tc = in ? tc : 0.0;
          tsn = in ? tsn : 0.0;
        }
        if (owner) {
          cacc_add<DET>(a00, w0, tc, C.detC); cacc_add<DET>(a01, w0, tsn, C.detC);
          cacc_add<DET>(a10, w1, tc, C.detC); cacc_add<DET>(a11, w1, tsn, C.detC);
          cacc_add<DET>(a20, w2, tc, C.detC); cacc_add<DET>(a21, w2, tsn, C.detC);
          cacc_add<DET>(a30, w3, tc, C.detC); cacc_add<DET>(a31, w3, tsn, C.detC);
        }
      }
      remaining &= ~mask;
    }
    // (the next group's rows are written only after every read above has been consumed: same wave, program order)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (cur >= 0) flush(cur, -1);
  for (int off = 32; off > 0; off >>= 1) {
    mass_used += __shfl_xor(mass_used, off);
    n_used += __shfl_xor(n_used, off);
  }
#if CSLOT_EXPT < 4
  if (lane == 0 && n_used > 0.0) {          // (see k_cyl_accumulate: CYL_TAILS slot pairs, not one)
    double *tp = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
    unsafeAtomicAdd(tp + 0, mass_used);
    unsafeAtomicAdd(tp + 1, n_used);
  }
#endif
}

// which formulation an accumulation launch uses: the slot kernel for mmax <= 7 (CYL_ACC_SLOT=0 builds the register
// formulation everywhere)
#ifndef CYL_ACC_SLOT
#define CYL_ACC_SLOT 1
#endif
template <int MMAX, bool DET, bool LIST = false>
static void cyl_acc_launch(unsigned gx, unsigned gz, hipStream_t st, const CylDev &C, const double *X, const double *Y,
                           const double *Z, const double *M, const uint32_t *lev_off, const LevChunks &LC, double *Wn,
                           double *tail, int multilevel, const CylAccList &al = CylAccList{})
{
  if constexpr (CYL_ACC_SLOT && MMAX <= 7)
    k_cyl_accumulate_slot<MMAX, DET, LIST><<<dim3(gx, 1, gz), CACC_WAVES * 64, 0, st>>>(C, X, Y, Z, M, lev_off, LC, Wn,
                                                                                       tail, multilevel, al);
  else
    k_cyl_accumulate<MMAX, DET, LIST><<<dim3(gx, 1, gz), CACC_WAVES * 64, 0, st>>>(C, X, Y, Z, M, lev_off, LC, Wn, tail,
                                                                                  multilevel, al);
}

// ---- multistep level change: coefficient differencing (src/CylEXP.cc:159-188) -----------------------
// Wnd[level][node][ntrig]; window: sqrt(R^2+z^2)/ASCALE <= Rtable only.
template <int MMAX>
__global__ void __launch_bounds__(256)
k_cyl_mstep_update(CylDev C, const double *__restrict__ X, const double *__restrict__ Y,
                   const double *__restrict__ Z, const double *__restrict__ M,
                   const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev,
                   const uint32_t *__restrict__ lev_off, int first, int last, int mfirst,
                   double *__restrict__ Wnd, int plain, double *__restrict__ tail,
                   const uint32_t *__restrict__ list = nullptr /* slots of the movers (k_mover_list; lev_off = {0, count}) */,
                   unsigned spread = 1)
{
  // plain != 0: every particle of the range adds its contribution to Wnd[its level] -- the accumulation
  // of SPARSE multistep levels, which are not cell-sorted (Cylinder's rcylmax cut, the in-cut mass /
  // count and EmpCylSL::accumulate's grid window, src/Cylinder.cc:853-866, exputil/EmpCylSL.cc:4062)
  constexpr int NT = 2 * MMAX + 1;
  size_t i = 0;
  bool have = false;
  if (list) {
    // few movers: one per `spread` lanes, so that their (serial, latency-bound) atomics come from more waves
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, g = t / spread;
    if (t % spread == 0 && g < lev_off[1]) { i = list[g]; have = true; }
  } else {
    i = lev_off[first] + (size_t)blockIdx.x * 256 + threadIdx.x;
    have = i < lev_off[last + 1];
  }
  bool mover = false;
  int from = 0, to = 0;
  if (have) {
    from = lev[i];
    to = plain ? from : newlev[i];
    mover = plain || from != to;
  }
  if (!__any(mover)) return;
  double xx = 1, yy = 0, zz = 0, mass = 0;
  if (mover) {
    cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
    mass = M[i];
  }
  const double r2 = xx * xx + yy * yy;
  double r, ir;
  sqrt_rsqrt(r2, r, ir);
  if (plain) {
    const bool incut = mover && (r2 + zz * zz) < C.rmax2;
    double mu = incut ? cdet_round(mass, C.detCm) : 0.0, nu = incut ? 1.0 : 0.0;
    for (int off = 32; off > 0; off >>= 1) { mu += __shfl_xor(mu, off); nu += __shfl_xor(nu, off); }
    if ((threadIdx.x & 63) == 0 && nu > 0.0) {        // (slot pairs, folded by the contraction: see k_cyl_accumulate)
      double *tp = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
      unsafeAtomicAdd(tp + 0, mu); unsafeAtomicAdd(tp + 1, nu);
    }
    mover = incut;
  }
  if (sqrt(r2 + zz * zz) > C.rtab_abs) mover = false;
  if (!__any(mover)) return;
  double zc = zz;
  if (zc > C.rtab_abs) zc = C.rtab_abs;
  if (zc < -C.rtab_abs) zc = -C.rtab_abs;
  int ix, iy;
  double cw[4];
  cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
  double cphi = 1.0, sphi = 0.0;
  if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
  const double t0 = mover ? -4.0 * M_PI * mass : 0.0;
  const int nyp = C.numy + 1;
  const size_t nnode = (size_t)(C.numx + 1) * nyp;
  // cos / sin (m phi) of this lane
  double cmv[MMAX + 1], smv[MMAX + 1];
  cmv[0] = 1.0; smv[0] = 0.0;
#pragma unroll
  for (int m = 1; m <= MMAX; m++) {
    cmv[m] = cmv[m - 1] * cphi - smv[m - 1] * sphi;
    smv[m] = smv[m - 1] * cphi + cmv[m - 1] * sphi;
  }
  // The store is ordered by (level, cell), so the movers of a wave share a handful of (cell, from,
  // to) keys: one lane per key adds the key's wave-reduced contribution.  (One atomic per mover and
  // value -- up to 64 lanes on the same word -- made the sweep that lifts a whole level 11 ms long.)
  const int lane = threadIdx.x & 63;
  const uint32_t mkey = mover ? (((uint32_t)(ix * nyp + iy) << 10) | ((uint32_t)from << 5) | (uint32_t)to)
                              : 0xffffffffu;
  unsigned long long rem = __ballot(mover);
  // Movers that do not share keys (an un-cell-sorted sparse level, or a few scattered level changes)
  // are served in parallel instead: every lane adds its own values.  The grouped loop below would
  // spend one serial round per key on them.
  {
    // (a mover is compared with the previous MOVER of the wave, non-movers in between do not count)
    const unsigned long long below = rem & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
    const int prev = below ? 63 - __clzll((long long)below) : lane;
    const uint32_t up = (uint32_t)__shfl((int)mkey, prev);
    const int npair = __popcll(__ballot(mover && below && mkey == up));
    if (2 * npair < __popcll(rem)) {
      if (mover) {
        const bool sub = !plain && from >= mfirst;
        double *wto = Wnd + ((size_t)to * nnode + (size_t)(ix * nyp + iy)) * NT;
        double *wfr = Wnd + ((size_t)from * nnode + (size_t)(ix * nyp + iy)) * NT;
        cstatic_for<0, MMAX + 1>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          if (C.EVEN_M && (m & 1)) return;
          constexpr int jc = (m == 0) ? 0 : 2 * m - 1;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const size_t off = (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT;
            const double w = t0 * cw[k];
            const double vc = cdet_round(w * cmv[m], C.detC), vs = cdet_round(w * smv[m], C.detC);
            unsafeAtomicAdd(wto + off + jc, vc);
            if (sub) unsafeAtomicAdd(wfr + off + jc, -vc);
            if constexpr (m > 0) {
              unsafeAtomicAdd(wto + off + jc + 1, vs);
              if (sub) unsafeAtomicAdd(wfr + off + jc + 1, -vs);
            }
          }
        });
      }
      return;
    }
  }
  while (rem) {
    const int lead = __ffsll((long long)rem) - 1;
    const uint32_t kk = (uint32_t)__shfl((int)mkey, lead);
    const unsigned long long mm = __ballot(mover && mkey == kk);
    rem &= ~mm;
    const bool in = (mm >> lane) & 1ull;
    const bool many = __popcll(mm) > 1;
    const int gto = (int)(kk & 31u), gfrom = (int)((kk >> 5) & 31u);
    const size_t gnode = (size_t)(kk >> 10);
    const bool sub = !plain && gfrom >= mfirst;
    double *wto = Wnd + ((size_t)gto * nnode + gnode) * NT;
    double *wfr = Wnd + ((size_t)gfrom * nnode + gnode) * NT;
    cstatic_for<0, MMAX + 1>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if (C.EVEN_M && (m & 1)) return;
      constexpr int jc = (m == 0) ? 0 : 2 * m - 1;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const size_t off = (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT;
        const double w = in ? t0 * cw[k] : 0.0;
        double vc = cdet_round(w * cmv[m], C.detC), vs = cdet_round(w * smv[m], C.detC);
        if (many) {
          for (int o = 32; o > 0; o >>= 1) {
            vc += __shfl_xor(vc, o);
            if constexpr (m > 0) vs += __shfl_xor(vs, o);
          }
        }
        if (lane == lead) {
          unsafeAtomicAdd(wto + off + jc, vc);
          if (sub) unsafeAtomicAdd(wfr + off + jc, -vc);
          if constexpr (m > 0) {
            unsafeAtomicAdd(wto + off + jc + 1, vs);
            if (sub) unsafeAtomicAdd(wfr + off + jc + 1, -vs);
          }
        }
      }
    });
  }
}

// ---- moments -> coefficients -----------------------------------------------------------------------------
// out[cs][m][n] = sum_node tab[cs ? 3 : 0][m][n][node] * Wn[node][trig(m, cs)]
// Two stages.  Stage 1: block (trig slot t, node segment, level) keeps the sums of ALL n in registers,
// so a moment Wn[node][t] -- stride ntrig, one cache line each -- is fetched once per (t, node) and
// not once per (t, n, node) as a block per coefficient would (that variant moved 330 MB through L2
// for 45 MB of tables: 70 us per contraction at 256 x 128).  Stage 2 adds the segments in a fixed
// order (same bits every run) and does setup_accumulation's swap on the way.
#ifndef CYL_CSEG
#define CYL_CSEG 48          // node segments of stage 1 (24: 312 blocks, too few to pull the 45 MB table at HBM rate: 50 -> 36 us)
#endif
#define CYL_CNB 12                 // n per register block
static_assert(CYL_CSEG % 8 == 0, "the block -> XCD mapping of k_cyl_contract_part");
__global__ void __launch_bounds__(256)
k_cyl_contract_part(CylDev C, const double *__restrict__ tab, double *__restrict__ Wn,
                    double *__restrict__ part /* [level][CYL_CSEG][ncoef] */,
                    int clear /* leave the moments zero behind (each is read by exactly one block) */)
{
  // (segment fastest: the ntrig blocks that read the same lines of Wn -- stride ntrig -- get block ids that differ by a
  // multiple of CYL_CSEG, a multiple of 8, i.e. they share an XCD and its L2)
  const int seg = blockIdx.x, t = blockIdx.y, L = blockIdx.z;
  const int m = (t + 1) >> 1, cs = t ? ((t + 1) & 1) : 0;
  const size_t nnode = (size_t)(C.numx + 1) * (C.numy + 1);
  const size_t ncoef = (size_t)2 * (C.mmax + 1) * C.nmax;
  Wn += (size_t)L * nnode * C.ntrig;
  const size_t k0 = nnode * seg / CYL_CSEG, k1 = nnode * (seg + 1) / CYL_CSEG;
  const double *T0 = tab + ((((size_t)(cs ? 3 : 0)) * (C.mmax + 1) + m) * C.nmax) * nnode;
  double *out = part + ((size_t)L * CYL_CSEG + seg) * ncoef + ((size_t)cs * (C.mmax + 1) + m) * C.nmax;
  __shared__ double red[4][CYL_CNB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int nb = 0; nb < C.nmax; nb += CYL_CNB) {
    double s[CYL_CNB];
#pragma unroll
    for (int j = 0; j < CYL_CNB; j++) s[j] = 0.0;
    const bool wipe = clear && nb + CYL_CNB >= C.nmax;
    for (size_t k = k0 + threadIdx.x; k < k1; k += 256) {
      const double w = Wn[k * C.ntrig + t];
      // a node without mass adds nothing (fma(T, 0, s) == s): its table column is not fetched -- the moments of a thinly
      // populated multistep level are almost all zero, and its contraction then reads the 3 MB of moments, not the 45 MB table
      if (w == 0.0) continue;
      if (wipe) Wn[k * C.ntrig + t] = 0.0;
#pragma unroll
      for (int j = 0; j < CYL_CNB; j++)
        if (nb + j < C.nmax) s[j] = fma(T0[(size_t)(nb + j) * nnode + k], w, s[j]);
    }
#pragma unroll
    for (int j = 0; j < CYL_CNB; j++) {
      for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_xor(s[j], off);
      if (lane == 0) red[wave][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < CYL_CNB && nb + (int)threadIdx.x < C.nmax)
      out[nb + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

// the accumulation launches' {in-cut mass, count} slot pairs, summed by the first wave of a block in a fixed order
// (two slots a lane, then a butterfly) and cleared; every lane of that wave returns the two sums.  A single thread
// walking the 128 slots took 10 us -- as long as everything else the small sub-steps' contraction does.
__device__ __forceinline__ void cyl_tail_fold(double *__restrict__ tailpart, double &t0, double &t1)
{
  static_assert(CYL_TAILS == 128, "two slot pairs a lane");
  const int lane = threadIdx.x & 63;
  double2 *tp = reinterpret_cast<double2 *>(tailpart);
  const double2 a = tp[lane], b = tp[lane + 64];
  tp[lane] = make_double2(0.0, 0.0);
  tp[lane + 64] = make_double2(0.0, 0.0);
  t0 = a.x + b.x;
  t1 = a.y + b.y;
  for (int off = 32; off > 0; off >>= 1) { t0 += __shfl_xor(t0, off); t1 += __shfl_xor(t1, off); }
}

__global__ void __launch_bounds__(256)
k_cyl_contract_sum(CylDev C, double *__restrict__ part, double *__restrict__ out, size_t ostride,
                   double *__restrict__ last, double *__restrict__ add_to /* += the new set as well, or null */,
                   double *__restrict__ tailpart /* [CYL_TAILS][2] of the accumulation launches, or null */,
                   int clear = 0 /* leave the partial sums zero behind (the thin accumulation adds to them) */)
{
  const size_t ncoef = (size_t)2 * (C.mmax + 1) * C.nmax;
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int L = blockIdx.y;                   // level of a multi-level launch
  // the accumulation's {in-cut mass, count} slots -> the tail of the FIRST set of the launch (out + ncoef), slots cleared
  if (tailpart && blockIdx.x == 0 && L == 0 && threadIdx.x < 64) {
    double t0, t1;
    cyl_tail_fold(tailpart, t0, t1);
    if (threadIdx.x < 2) out[ncoef + threadIdx.x] += threadIdx.x ? t1 : t0;
  }
  if (o >= ncoef) return;
  const bool none = o >= ncoef / 2 && o < ncoef / 2 + (size_t)C.nmax;      // (sin, m = 0): no such row
  double s = 0.0;
  if (!none)
    for (int seg = 0; seg < CYL_CSEG; seg++) s += part[((size_t)L * CYL_CSEG + seg) * ncoef + o];
  if (clear) for (int seg = 0; seg < CYL_CSEG; seg++) part[((size_t)L * CYL_CSEG + seg) * ncoef + o] = 0.0;
  out += (size_t)L * ostride;
  if (last) {                                 // setup_accumulation's swap on the way: last <- out, out <- new
    last += (size_t)L * ostride;
    last[o] = out[o];
  }
  out[o] = s;
  if (add_to) add_to[(size_t)L * ostride + o] += s;
}

// The block-multistep sub-step's form for a rank that is alone: the segment sums of every active level with
// setup_accumulation's swap, the {in-cut mass, count} slots of the accumulation folded straight into the master step's
// tally (k_cyl_mass_take: while its first sub-step is open; the sets' tails stay zero), THEN the combined set of
// CylEXP::compute_multistep_coefficients (src/CylEXP.cc:192-282) -- what k_cyl_contract_sum + k_cyl_mass_take +
// k_mstep_combine do, in one launch instead of three.
__global__ void __launch_bounds__(256)
k_cyl_sum_combine(CylDev C, double *__restrict__ part, double *__restrict__ N, double *__restrict__ Lset,
                  size_t stride, int lo, int nact, int nlev, int mfirst, CombineW W, double *__restrict__ out,
                  double *__restrict__ tailpart, double *__restrict__ mass_acc, int open, int clear = 0)
{
  const size_t ncoef = (size_t)2 * (C.mmax + 1) * C.nmax;
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double t0, t1;
    cyl_tail_fold(tailpart, t0, t1);
    if (threadIdx.x < 2) {
      if (open) mass_acc[threadIdx.x] += threadIdx.x ? t1 : t0;
      out[ncoef + threadIdx.x] = 0.0;
    }
  }
  if (o >= ncoef) return;
  const bool none = o >= ncoef / 2 && o < ncoef / 2 + (size_t)C.nmax;      // (sin, m = 0): no such row
  for (int j = 0; j < nact; j++) {
    double s = 0.0;
    if (!none)
      for (int seg = 0; seg < CYL_CSEG; seg++) s += part[((size_t)j * CYL_CSEG + seg) * ncoef + o];
    if (clear) for (int seg = 0; seg < CYL_CSEG; seg++) part[((size_t)j * CYL_CSEG + seg) * ncoef + o] = 0.0;
    const size_t q = (size_t)(lo + j) * stride + o;
    Lset[q] = N[q];
    N[q] = s;
  }
  out[o] = expamd_combine_one(Lset, N, stride, nlev, mfirst, W.ab, o);
}

// both stages; nl levels starting at Wn / out / last
// thin: stage 1 has been done by k_cyl_acc_thin (its sums ADDED to `part`, which stage 2 then leaves zero)
static void cyl_contract(hipStream_t st, const CylDev &C, const double *tab, double *Wn, double *part,
                         double *out, int nl = 1, size_t ostride = 0, double *last = nullptr, int clear = 0,
                         double *add_to = nullptr, double *tailpart = nullptr, bool thin = false)
{
  const size_t ncoef = (size_t)2 * (C.mmax + 1) * C.nmax;
  if (!thin) k_cyl_contract_part<<<dim3(CYL_CSEG, C.ntrig, nl), 256, 0, st>>>(C, tab, Wn, part, clear);
  k_cyl_contract_sum<<<dim3(cdiv(ncoef, 256), nl), 256, 0, st>>>(C, part, out, ostride, last, add_to, tailpart, (thin || clear) ? 1 : 0);
}

// ---- coefficients -> projected node table ----------------------------------------------------------------
// TF[node][3*ntrig]: for m = 0: {Pc, Rc, Zc}; for m >= 1 at 3 + 6(m-1): {Pc, Rc, Zc, Ps, Rs, Zs}
// twin != 0: the sine tables are bit for bit the cosine tables (the usual case: an EOF basis conditioned on an
// axisymmetric density has SC == SS, exputil/EmpCylSL.cc:2556-2760; CylForce checks it when the tables arrive).  The
// cosine and sine rows of a harmonic are then formed from ONE fetch of each table value -- same products, same sums,
// half the 133 MB (256 x 128, mmax 6, nmax 12) that bound this kernel.
__global__ void __launch_bounds__(256)
k_cyl_project(CylDev C, const double *__restrict__ tab, const double *__restrict__ coef,
              double *__restrict__ TF, int twin)
{
  const size_t nnode = (size_t)(C.numx + 1) * (C.numy + 1);
  const size_t node = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (node >= nnode) return;
  const int m = blockIdx.y;
  const int NF = 3 * C.ntrig;
  const int q0 = (m == 0) ? 0 : 3 + 6 * (m - 1);
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  if (twin && m > 0) {
    const int nq = C.nmax & ~3;
    for (int kind = 0; kind < 3; kind++) {
      const double *T = tab + (((size_t)kind * (C.mmax + 1) + m) * C.nmax) * nnode + node;
      const double *cc = coef + (size_t)m * C.nmax, *cs = cc + half;
      double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
      for (int nb = 0; nb < C.nmax; nb += 12) {
        double t[12];
#pragma unroll
        for (int j = 0; j < 12; j++) t[j] = nb + j < C.nmax ? T[(size_t)(nb + j) * nnode] : 0.0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
          const int n = nb + j;
          if (n < nq) { a[j & 3] = fma(t[j], cc[n], a[j & 3]); b[j & 3] = fma(t[j], cs[n], b[j & 3]); }
          else if (n < C.nmax) { a[0] = fma(t[j], cc[n], a[0]); b[0] = fma(t[j], cs[n], b[0]); }
        }
      }
      TF[node * NF + q0 + kind] = (a[0] + a[1]) + (a[2] + a[3]);
      TF[node * NF + q0 + kind + 3] = (b[0] + b[1]) + (b[2] + b[3]);
    }
    return;
  }
  for (int kind = 0; kind < (m == 0 ? 3 : 6); kind++) {
    const double *T = tab + (((size_t)kind * (C.mmax + 1) + m) * C.nmax) * nnode + node;
    const double *c = coef + (kind >= 3 ? half : 0) + (size_t)m * C.nmax;
    // four chains (orders n = j mod 4 below the last multiple of four, the rest on chain 0); the table loads of twelve
    // orders are issued before the first of them is used: the kernel is bound by the loads it keeps in flight
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const int nq = C.nmax & ~3;
    for (int nb = 0; nb < C.nmax; nb += 12) {
      double t[12];
#pragma unroll
      for (int j = 0; j < 12; j++) t[j] = nb + j < C.nmax ? T[(size_t)(nb + j) * nnode] : 0.0;
#pragma unroll
      for (int j = 0; j < 12; j++) {
        const int n = nb + j;
        if (n < nq) a[j & 3] = fma(t[j], c[n], a[j & 3]);
        else if (n < C.nmax) a[0] = fma(t[j], c[n], a[0]);
      }
    }
    TF[node * NF + q0 + kind] = (a[0] + a[1]) + (a[2] + a[3]);
  }
}

// ---- force -------------------------------------------------------------------------------------------------

struct CylOut { double p, fr, fz, fp; };

template <int MMAX, class PT>
__device__ __forceinline__ CylOut cyl_field(const CylDev &C, PT t00, PT t10, PT t01, PT t11,
                                            double c00, double c10, double c01, double c11,
                                            double cphi, double sphi)
{
  CylOut o{0.0, 0.0, 0.0, 0.0};
  double cm = 1.0, sm = 0.0;
  cstatic_for<0, MMAX + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m > 0) {
      const double cn = cm * cphi - sm * sphi;
      const double sn = sm * cphi + cm * sphi;
      cm = cn; sm = sn;
    }
    const bool on = !(C.EVEN_M && (m & 1));        // exputil/EmpCylSL.cc:5318-5319
    if (on) {
      constexpr int q = (m == 0) ? 0 : 3 + 6 * (m - 1);
      auto bl = [&](int k) {
        return c00 * t00[q + k] + c10 * t10[q + k] + c01 * t01[q + k] + c11 * t11[q + k];
      };
      const double Pc = bl(0), Rc = bl(1), Zc = bl(2);
      if constexpr (m == 0) {
        o.p += Pc;
        o.fr += Rc;
        o.fz += Zc;
      } else {
        const double Ps = bl(3), Rs = bl(4), Zs = bl(5);
        o.p += Pc * cm + Ps * sm;
        o.fr += Rc * cm + Rs * sm;
        o.fz += Zc * cm + Zs * sm;
        o.fp += (Pc * sm - Ps * cm) * m;
      }
    }
  });
  return o;
}

// TAIL == false: the launch over the slot range.  Lanes inside 0.75 of the table radius (no taper, no monopole: almost
// every particle of the component the basis belongs to) are finished here; a wave with lanes beyond leaves its first slot
// and their mask on `work` and the tail launch (TAIL == true: one wave per work item, the same body with the erf taper
// and the monopole blend of src/Cylinder.cc:1357-1408) finishes those.  The split keeps erf -- a long routine that put
// 68 bytes of scratch under every wave -- out of the kernel that does the bulk of the work; a lane's arithmetic is the
// same in either kernel (frac = 1, cfrac = 0 multiply exactly).
#define CYL_WORK_STRIDE 3
template <int MMAX, bool TAIL>
__global__ void __launch_bounds__(256)
k_cyl_force(CylDev C, const double *__restrict__ X, const double *__restrict__ Y,
            const double *__restrict__ Z, const uint32_t *__restrict__ lev_off, int lev_lo,
            int lev_hi, const double *__restrict__ TF, const double *__restrict__ cylmass_p,
            double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ,
            double *__restrict__ POT, double *__restrict__ VX, double *__restrict__ VY,
            double *__restrict__ VZ, double dt_kick, int assign, uint32_t *__restrict__ key_out,
            double nk_dtk, double nk_dtd, int store_v, uint32_t *__restrict__ work, uint32_t *__restrict__ nwork,
            uint32_t *__restrict__ nwork_clear /* the counter of the NEXT launch pair: zeroed by the tail launch */)
{
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const int lane = threadIdx.x & 63;
  size_t base;
  unsigned long long mask = ~0ull;
  if constexpr (TAIL) {
    if (work == nullptr) {
      // the whole range in this kernel alone: ANOTHER component's particles (a halo around the disk: most of them beyond
      // the table radius, every wave would go through the list)
      base = beg + ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
      if (base >= end) return;
    } else {
      if (nwork_clear && blockIdx.x == 0 && threadIdx.x == 0) *nwork_clear = 0u;
      const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
      if (w >= *nwork) return;
      base = work[CYL_WORK_STRIDE * w];
      mask = (unsigned long long)work[CYL_WORK_STRIDE * w + 1] | ((unsigned long long)work[CYL_WORK_STRIDE * w + 2] << 32);
    }
  } else {
    base = beg + ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
    if (base >= end) return;
  }
  const size_t i = base + lane;
  bool valid = i < end && ((mask >> lane) & 1ull);
  double xx = 1, yy = 0, zz = 0;
  if (valid) {
    cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
  }
  // src/Cylinder.cc:1357-1381
  const double ratmin = 0.75, maxerf = 3.0;
  const double midpt = ratmin + 0.5 * (1.0 - ratmin);
  const double rsmth = 0.5 * (1.0 - ratmin) / maxerf;
  const double r2 = xx * xx + yy * yy;
  double rp, irp, r3s, ir3s;                  // sqrt(x^2+y^2), sqrt(x^2+y^2+z^2) and their reciprocals
  sqrt_rsqrt(r2, rp, irp);
  sqrt_rsqrt(r2 + zz * zz, r3s, ir3s);
  const double r = rp + DSMALL;
  double cphi = 1.0, sphi = 0.0;
  if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
  const double ratio = r3s * C.inv_rtab_abs;              // sqrt((r^2 + z^2) / (ascale rtable)^2)
  double frac = 1.0, cfrac = 0.0;
  if constexpr (TAIL) {
    if (ratio >= 1.0) { frac = 0.0; cfrac = 1.0; }
    else if (ratio > ratmin) { frac = 0.5 * (1.0 - erf((ratio - midpt) / rsmth)); cfrac = 1.0 - frac; }
    else { cfrac = 0.0; frac = 1.0; }
  } else {
    // beyond 0.75 of the table radius: the tail launch's business
    const unsigned long long far = __ballot(valid && ratio > ratmin);
    if (far) {
      if (lane == 0) {
        const uint32_t w = atomicAdd(nwork, 1u);
        work[CYL_WORK_STRIDE * w] = (uint32_t)base;
        work[CYL_WORK_STRIDE * w + 1] = (uint32_t)far;
        work[CYL_WORK_STRIDE * w + 2] = (uint32_t)(far >> 32);
      }
      if ((far >> lane) & 1ull) valid = false;
    }
  }

  // accumulated_eval (exputil/EmpCylSL.cc:5272-5314): off grid -> zeros
  const bool ongrid = valid && ratio < 1.0 && !(r3s > C.rtab_abs);
  int ix, iy;
  double c00, c10, c01, c11;
  cyl_weights(C, r, zz, ix, iy, c00, c10, c01, c11);
  int cell = ix * C.numy + iy;
  const int cell_u = __builtin_amdgcn_readfirstlane(cell);
  if (!ongrid) cell = cell_u;
  const bool uniform = (!TAIL || work == nullptr) && __all(cell == cell_u);
  const int NF = 3 * (2 * MMAX + 1);
  const int nyp = C.numy + 1;
  CylOut o{0.0, 0.0, 0.0, 0.0};
  if (__any(ongrid)) {
    if (uniform) {
      const int ux = cell_u / C.numy, uy = cell_u - ux * C.numy;
      // (pulling node rows into L2 ahead of the sweep, as the spherical fast pass does with its table, was
      // measured neutral here: 0.27 ms with and without, at 512 - 4096 nodes of lead)
      cdp t00 = (cdp)(TF + ((size_t)ux * nyp + uy) * NF);
      cdp t01 = t00 + NF, t10 = t00 + (size_t)nyp * NF, t11 = t10 + NF;
      o = cyl_field<MMAX>(C, t00, t10, t01, t11, c00, c10, c01, c11, cphi, sphi);
    } else {
      const double *t00 = TF + ((size_t)ix * nyp + iy) * NF;
      const double *t01 = t00 + NF, *t10 = t00 + (size_t)nyp * NF, *t11 = t10 + NF;
      o = cyl_field<MMAX>(C, t00, t10, t01, t11, c00, c10, c01, c11, cphi, sphi);
    }
  }
  if (!valid) return;

  double fx = 0.0, fy = 0.0, fz = 0.0, pa = 0.0;
  if (ratio < 1.0) {
    double p = 0.0, fr = 0.0, fzz = 0.0, fp = 0.0;
    if (ongrid) { p = o.p; fr = o.fr; fzz = o.fz; fp = o.fp; }
    // (1/r2 is infinite on the axis: the reference's fp*yy/r2 is 0/0 = NaN there, and so is this)
    const double ir = rcp_refine(r, irp), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
    fx = (fr * xx * ir - fp * yy * ir2) * frac;    // src/Cylinder.cc:1387-1390
    fy = (fr * yy * ir + fp * xx * ir2) * frac;
    fz = fzz * frac;
    pa = p * frac;
  }
  if constexpr (TAIL) {
    if (ratio > ratmin) {                             // monopole blend, src/Cylinder.cc:1398-1408
      const double p = -(*cylmass_p) * ir3s;          // -M / sqrt(r^2 + z^2)
      const double fr = p * (ir3s * ir3s);
      fx += xx * fr * cfrac;
      fy += yy * fr * cfrac;
      fz += zz * fr * cfrac;
      pa += p * cfrac;
    }
  }
  if (C.use_rot) {                                  // frc = transformOrig * frc (src/Cylinder.cc:1417-1418)
    const double a = fx, b = fy, c = fz;
    fx = C.rot[0] * a + C.rot[3] * b + C.rot[6] * c;
    fy = C.rot[1] * a + C.rot[4] * b + C.rot[7] * c;
    fz = C.rot[2] * a + C.rot[5] * b + C.rot[8] * c;
  }
  if (C.ps.center | C.ps.axis) {        // acc += val - pseudo (Component::AddAcc, src/Component.H:914-921)
    double qx, qy, qz, ux = 0.0, uy = 0.0, uz = 0.0;
    if (C.ps.axis) { ux = VX[i]; uy = VY[i]; uz = VZ[i]; }
    pseudo_accel(C.ps, X[i], Y[i], Z[i], ux, uy, uz, qx, qy, qz);
    fx -= qx; fy -= qy; fz -= qz;
  }
  if (!assign) {
    fx += AX[i];
    fy += AY[i];
    fz += AZ[i];
    pa += POT[i];
  }
  AX[i] = fx;
  AY[i] = fy;
  AZ[i] = fz;
  POT[i] = pa;
  if (dt_kick != 0.0) {
    const double vx = mul_then_add(VX[i], fx, dt_kick);
    const double vy = mul_then_add(VY[i], fy, dt_kick);
    const double vz = mul_then_add(VZ[i], fz, dt_kick);
    if (store_v == 1) { VX[i] = vx; VY[i] = vy; VZ[i] = vz; }   // 0: deferred (exp_amd_comp::pending_kick)
    if (key_out) {
      // the sort key this particle will have after the NEXT fused step's kick + drift (the
      // arithmetic of advance_one on the values just stored): that step then only histograms
      // the 4-byte keys (exp_amd_step_kdk, see sph_kernels.h for the spherical twin)
      const double wx = mul_then_add(vx, fx, nk_dtk);
      const double wy = mul_then_add(vy, fy, nk_dtk);
      const double wz = mul_then_add(vz, fz, nk_dtk);
      // store_v == 2: velocities stored with the next step's opening half-kick applied (sph_kernels.h)
      if (store_v == 2) { VX[i] = wx; VY[i] = wy; VZ[i] = wz; }
      CylKeyFn kf{C, 0u};
      key_out[i] = kf(mul_then_add(X[i], wx, nk_dtd), mul_then_add(Y[i], wy, nk_dtd),
                      mul_then_add(Z[i], wz, nk_dtd), 0);
    }
  }
}

// ---- thin active sets: straight from the basis tables ---------------------------------------------------------------
// The cylinder's twin of sph_kernels.h's k_sph_acc_thin / k_sph_force_thin (see there for the why): the few active
// particles of an upper time-step level are accumulated and evaluated per particle, as EmpCylSL::accumulate and
// accumulated_eval do (exputil/EmpCylSL.cc:4049-4146, :5256-5410), with the block as the unit of parallelism -- no node
// moments, no contraction over 33 000 nodes, no projection of the 66 MB table set.  The tables are read through a
// NODE-MAJOR copy tabT[node][kind][m][n] (a particle's four corner nodes are four contiguous 2-4 KB stretches; the
// [kind][m][n][node] layout of the sweep kernels would cost one cache line per (kind, m, n, corner pair)).
//   k_cyl_acc_thin  : part[level][seg][cs][m][n] += sum_corners (-4 pi mass c_k trig_m) tab[pot cos|sin][m][n][node_k],
//                     the layout k_cyl_contract_part leaves, finished by k_cyl_contract_sum / k_cyl_sum_combine;
//   k_cyl_force_thin: TF rows of the particle's four corner nodes projected into LDS (the sums of k_cyl_project, same
//                     four chains), then cyl_field and Cylinder's taper / monopole blend exactly as in k_cyl_force.
__global__ void __launch_bounds__(256)
k_cyl_transpose(const double *__restrict__ tab, double *__restrict__ tabT, size_t nnode, int per_node /* nk (mmax+1) nmax */)
{
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;            // index into tabT
  if (o >= nnode * (size_t)per_node) return;
  const size_t node = o / per_node, k = o - node * per_node;
  tabT[o] = tab[k * nnode + node];
}

// sum_n T[n] c[n] in the four chains of k_cyl_project: orders n = j mod 4 below the last multiple of four, the rest on
// chain 0; the table values twelve at a time, all loads of a batch issued before the first is used
template <class CP>
__device__ __forceinline__ double cyl_chain4(const double *__restrict__ T, CP c, int nmax)
{
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  const int nq = nmax & ~3;
  for (int nb = 0; nb < nmax; nb += 12) {
    double t[12];
#pragma unroll
    for (int j = 0; j < 12; j++) t[j] = nb + j < nmax ? T[nb + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const int n = nb + j;
      if (n < nq) a[j & 3] = fma(t[j], c[n], a[j & 3]);
      else if (n < nmax) a[0] = fma(t[j], c[n], a[0]);
    }
  }
  return (a[0] + a[1]) + (a[2] + a[3]);
}

// ... the cosine and the sine row of a harmonic from ONE fetch of each table value (the sine tables being bit for bit the
// cosine tables: k_cyl_project's twin branch)
template <class CP>
__device__ __forceinline__ void cyl_chain4_pair(const double *__restrict__ T, CP cc, CP cs, int nmax, double &ra, double &rb)
{
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const int nq = nmax & ~3;
  for (int nb = 0; nb < nmax; nb += 12) {
    double t[12];
#pragma unroll
    for (int j = 0; j < 12; j++) t[j] = nb + j < nmax ? T[nb + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const int n = nb + j;
      if (n < nq) { a[j & 3] = fma(t[j], cc[n], a[j & 3]); b[j & 3] = fma(t[j], cs[n], b[j & 3]); }
      else if (n < nmax) { a[0] = fma(t[j], cc[n], a[0]); b[0] = fma(t[j], cs[n], b[0]); }
    }
  }
  ra = (a[0] + a[1]) + (a[2] + a[3]);
  rb = (b[0] + b[1]) + (b[2] + b[3]);
}

#define CYL_THIN_TP_MAX 64

template <int MMAX>
__global__ void __launch_bounds__(256)
k_cyl_force_thin(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                 const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi, const double *__restrict__ tabT, int nk,
                 const double *__restrict__ coef, const double *__restrict__ cylmass_p, double *__restrict__ AX,
                 double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT, double *__restrict__ VX,
                 double *__restrict__ VY, double *__restrict__ VZ, int assign, int tp)
{
  extern __shared__ __attribute__((aligned(16))) double cthin_lds[];
  __shared__ int s_node[CYL_THIN_TP_MAX];
  constexpr int NT = 2 * MMAX + 1, NF = 3 * NT, NFS = NF + 1;       // (row stride 40 doubles: lanes 16 banks apart)
  const int half = (C.mmax + 1) * C.nmax;
  double *s_coef = cthin_lds;                                       // cos block, sin block
  double *stage = cthin_lds + 2 * half;                             // [tp][PS]: four corner rows of NFS doubles each
  constexpr int PS = 4 * NFS + 2;                                   // (particle stride = 4 banks mod 64: no conflicts)
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  if (beg + (size_t)blockIdx.x * tp >= end) return;
  const int NTH = blockDim.x;                 // 256, or 64 for ranges of thousands (see k_sph_force_thin)
  for (int k = threadIdx.x; k < 2 * half; k += NTH) s_coef[k] = coef[k];
  const int t = threadIdx.x;
  const int nyp = C.numy + 1;
  const size_t per_node = (size_t)nk * half;
  for (size_t base = beg + (size_t)blockIdx.x * tp; base < end; base += (size_t)gridDim.x * tp) {
    // ---- the prologue of k_cyl_force, one particle per lane of the first wave
    const size_t i = base + t;
    const bool valid = t < tp && i < end;
    double xx = 1, yy = 0, zz = 0;
    if (valid) cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
    const double ratmin = 0.75, maxerf = 3.0;                       // src/Cylinder.cc:1357-1381
    const double midpt = ratmin + 0.5 * (1.0 - ratmin);
    const double rsmth = 0.5 * (1.0 - ratmin) / maxerf;
    const double r2 = xx * xx + yy * yy;
    double rp, irp, r3s, ir3s;
    sqrt_rsqrt(r2, rp, irp);
    sqrt_rsqrt(r2 + zz * zz, r3s, ir3s);
    const double r = rp + DSMALL;
    double cphi = 1.0, sphi = 0.0;
    if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
    const double ratio = r3s * C.inv_rtab_abs;
    double frac, cfrac;
    if (ratio >= 1.0) { frac = 0.0; cfrac = 1.0; }
    else if (ratio > ratmin) { frac = 0.5 * (1.0 - erf((ratio - midpt) / rsmth)); cfrac = 1.0 - frac; }
    else { cfrac = 0.0; frac = 1.0; }
    const bool ongrid = valid && ratio < 1.0 && !(r3s > C.rtab_abs);
    int ix, iy;
    double c00, c10, c01, c11;
    cyl_weights(C, r, zz, ix, iy, c00, c10, c01, c11);
    if (t < CYL_THIN_TP_MAX) s_node[t] = ongrid ? ix * nyp + iy : -1;
    __syncthreads();
    // ---- TF rows of the four corner nodes of every on-grid particle: item = (particle, corner, kind, m)
    const int per_p = 4 * 3 * (C.mmax + 1);
    for (int it = threadIdx.x; it < tp * per_p; it += NTH) {
      const int p = it / per_p;
      int rest = it - p * per_p;
      const int node0 = s_node[p];
      if (node0 < 0) continue;
      const int k = rest / (3 * (C.mmax + 1));
      rest -= k * 3 * (C.mmax + 1);
      const int kind = rest / (C.mmax + 1), m = rest - kind * (C.mmax + 1);
      const size_t node = (size_t)node0 + ((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0);     // 00, 10, 01, 11
      const double *T = tabT + node * per_node + ((size_t)kind * (C.mmax + 1) + m) * C.nmax;
      const double *cc = s_coef + (size_t)m * C.nmax, *cs = cc + half;
      double *o = stage + (size_t)p * PS + (size_t)k * NFS + ((m == 0) ? 0 : 3 + 6 * (m - 1));
      if (m == 0) o[kind] = cyl_chain4(T, cc, C.nmax);
      else if (nk == 3) cyl_chain4_pair(T, cc, cs, C.nmax, o[kind], o[kind + 3]);    // (sine tables == cosine tables)
      else {
        o[kind] = cyl_chain4(T, cc, C.nmax);
        o[kind + 3] = cyl_chain4(T + (size_t)3 * half, cs, C.nmax);                   // the sine tables: kinds 3-5
      }
    }
    __syncthreads();
    if (valid) {
      CylOut o{0.0, 0.0, 0.0, 0.0};
      if (ongrid) {
        const double *t00 = stage + (size_t)t * PS;
        const double *t10 = t00 + NFS, *t01 = t00 + 2 * NFS, *t11 = t00 + 3 * NFS;
        o = cyl_field<MMAX>(C, t00, t10, t01, t11, c00, c10, c01, c11, cphi, sphi);
      }
      // ---- the tail of k_cyl_force (src/Cylinder.cc:1387-1418), no fused kick
      double fx = 0.0, fy = 0.0, fz = 0.0, pa = 0.0;
      if (ratio < 1.0) {
        double p = 0.0, fr = 0.0, fzz = 0.0, fp = 0.0;
        if (ongrid) { p = o.p; fr = o.fr; fzz = o.fz; fp = o.fp; }
        const double ir = rcp_refine(r, irp), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
        fx = (fr * xx * ir - fp * yy * ir2) * frac;
        fy = (fr * yy * ir + fp * xx * ir2) * frac;
        fz = fzz * frac;
        pa = p * frac;
      }
      if (ratio > ratmin) {
        const double p = -(*cylmass_p) * ir3s;
        const double fr = p * (ir3s * ir3s);
        fx += xx * fr * cfrac;
        fy += yy * fr * cfrac;
        fz += zz * fr * cfrac;
        pa += p * cfrac;
      }
      if (C.use_rot) {
        const double a = fx, b = fy, c = fz;
        fx = C.rot[0] * a + C.rot[3] * b + C.rot[6] * c;
        fy = C.rot[1] * a + C.rot[4] * b + C.rot[7] * c;
        fz = C.rot[2] * a + C.rot[5] * b + C.rot[8] * c;
      }
      if (C.ps.center | C.ps.axis) {
        double qx, qy, qz, ux = 0.0, uy = 0.0, uz = 0.0;
        if (C.ps.axis) { ux = VX[i]; uy = VY[i]; uz = VZ[i]; }
        pseudo_accel(C.ps, X[i], Y[i], Z[i], ux, uy, uz, qx, qy, qz);
        fx -= qx; fy -= qy; fz -= qz;
      }
      if (!assign) { fx += AX[i]; fy += AY[i]; fz += AZ[i]; pa += POT[i]; }
      AX[i] = fx; AY[i] = fy; AZ[i] = fz; POT[i] = pa;
    }
    __syncthreads();
  }
}

// Accumulation of a thin, level-contiguous slot range into part[level - lo][seg][ncoef] (zero on entry; consumed and
// cleared by k_cyl_contract_sum / k_cyl_sum_combine with clear = 1).  Cuts, window and weights are those of the sparse
// accumulation (k_cyl_mstep_update with plain = 1): Cylinder's rcylmax cut with its {mass, count} tally, the grid window
// of EmpCylSL::accumulate, z clamped to the table.
template <int MMAX>
__global__ void __launch_bounds__(256)
k_cyl_acc_thin(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
               const double *__restrict__ M, const uint32_t *__restrict__ lev_off, int lo, int hi,
               const double *__restrict__ tabT, int nk, double *__restrict__ part, double *__restrict__ tail, int tpa,
               ThinAdv adv)
{
  extern __shared__ __attribute__((aligned(16))) double cthin_lds[];
  constexpr int NT = 2 * MMAX + 1;
  __shared__ int s_node[CYL_THIN_TP_MAX], s_lev[CYL_THIN_TP_MAX];
  __shared__ double s_cw[CYL_THIN_TP_MAX][4], s_trig[CYL_THIN_TP_MAX][NT + 1];
  const int half = (C.mmax + 1) * C.nmax, ncoef = 2 * half;
  const int nset = nk == 3 ? 1 : 2;                                 // potential tables: one (cos == sin) or two
  double *pe = cthin_lds;                                           // [tpa][nset][half]: sum_k c_k tab[pot][m][n][node_k]
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  const int seg = blockIdx.x % CYL_CSEG;
  const int t = threadIdx.x;
  const int nyp = C.numy + 1;
  const size_t per_node = (size_t)nk * half;
  for (size_t base = beg + (size_t)blockIdx.x * tpa; base < end; base += (size_t)gridDim.x * tpa) {
    const int np = (int)((end - base) < (size_t)tpa ? (end - base) : (size_t)tpa);
    if (t < 64) {                                                   // (the whole first wave: the tally is wave-reduced)
      const size_t i = base + t;
      const bool valid = t < tpa && i < end;
      double xx = 1, yy = 0, zz = 0, mass = 0;
      if (valid) {
        double px, py, pz;
        if (adv.on) thin_advance(adv, i, px, py, pz); else { px = X[i]; py = Y[i]; pz = Z[i]; }
        cyl_local(C, px, py, pz, xx, yy, zz);
        mass = M[i];
      }
      const double r2 = xx * xx + yy * yy;
      double r, ir;
      sqrt_rsqrt(r2, r, ir);
      const bool incut = valid && (r2 + zz * zz) < C.rmax2;
      double mu = incut ? mass : 0.0, nu = incut ? 1.0 : 0.0;
      for (int off = 32; off > 0; off >>= 1) { mu += __shfl_xor(mu, off); nu += __shfl_xor(nu, off); }
      if (t == 0 && nu > 0.0) {
        double *tp_ = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
        unsafeAtomicAdd(tp_ + 0, mu); unsafeAtomicAdd(tp_ + 1, nu);
      }
      const bool on = incut && !(sqrt(r2 + zz * zz) > C.rtab_abs);
      double zc = zz;
      if (zc > C.rtab_abs) zc = C.rtab_abs;
      if (zc < -C.rtab_abs) zc = -C.rtab_abs;
      int ix, iy;
      double cw[4];
      cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
      double cphi = 1.0, sphi = 0.0;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      const double t0 = on ? -4.0 * M_PI * mass : 0.0;
      if (t < tpa) {
        int lv = lo;
        while (lv < hi && i >= lev_off[lv + 1]) lv++;
        s_node[t] = on ? ix * nyp + iy : -1;
        s_lev[t] = lv;
#pragma unroll
        for (int k = 0; k < 4; k++) s_cw[t][k] = cw[k];
        double cm = 1.0, sm = 0.0;
        s_trig[t][0] = t0;
#pragma unroll
        for (int m = 1; m <= MMAX; m++) {
          const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi;
          cm = cn; sm = sn;
          const bool m_on = !(C.EVEN_M && (m & 1));
          s_trig[t][2 * m - 1] = m_on ? t0 * cm : 0.0;
          s_trig[t][2 * m] = m_on ? t0 * sm : 0.0;
        }
      }
    }
    __syncthreads();
    for (int it0 = threadIdx.x; it0 < np * nset * half; it0 += 3 * 256) {     // (three items = twelve loads in flight)
      double tv[3][4];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int it = it0 + u * 256;
        tv[u][0] = tv[u][1] = tv[u][2] = tv[u][3] = 0.0;
        if (it < np * nset * half) {
          const int p = it / (nset * half);
          const int rest = it - p * nset * half;
          const int set = rest / half, mn = rest - set * half;
          const int node0 = s_node[p];
          if (node0 >= 0) {
            const double *T = tabT + (size_t)node0 * per_node + (size_t)(set ? 3 : 0) * half + mn;
            tv[u][0] = T[0]; tv[u][1] = T[(size_t)nyp * per_node]; tv[u][2] = T[per_node]; tv[u][3] = T[(size_t)(nyp + 1) * per_node];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int it = it0 + u * 256;
        if (it < np * nset * half) {
          const int p = it / (nset * half);
          pe[it] = s_cw[p][0] * tv[u][0] + s_cw[p][1] * tv[u][1] + s_cw[p][2] * tv[u][2] + s_cw[p][3] * tv[u][3];
        }
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < ncoef; o += 256) {
      const int cs = o / half, mn = o - cs * half, m = mn / C.nmax;
      if (cs && m == 0) continue;                                  // (sin, m = 0): no such row
      const int jt = m == 0 ? 0 : 2 * m - 1 + cs;
      const int set = (cs && nset == 2) ? 1 : 0;
      double acc = 0.0;
      int cur = s_lev[0];
      for (int p0 = 0; p0 < np; p0 += 8) {                 // (eight particles' LDS reads issued before their fmas)
        double tt_[8], pp_[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool in = p0 + u < np;
          tt_[u] = in ? s_trig[p0 + u][jt] : 0.0;
          pp_[u] = in ? pe[((size_t)(p0 + u) * nset + set) * half + mn] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (p0 + u < np) {
            const int lv = s_lev[p0 + u];
            if (lv != cur) {
              if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(cur - lo) * CYL_CSEG + seg) * ncoef + o, acc);
              acc = 0.0;
              cur = lv;
            }
            acc = fma(tt_[u], pp_[u], acc);
          }
        }
      }
      if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(cur - lo) * CYL_CSEG + seg) * ncoef + o, acc);
    }
    __syncthreads();
  }
}

// Level-change differencing of FEW movers (multistep_update, src/CylEXP.cc:45-157), direct: tiles over the mover list;
// a mover adds its contribution to the set of its proposed level and takes it out of its level's set (levels >= mfirst
// only).  Window of k_cyl_mstep_update with plain = 0: the grid window, z clamped to the table, no rcylmax cut, no tally.
// Sums into part[level - mfirst][seg][ncoef] (zero on entry), finished by k_cyl_contract_sum (add_to = expcoefN).
template <int MMAX>
__global__ void __launch_bounds__(256)
k_cyl_diff_thin(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                const double *__restrict__ M, const uint32_t *__restrict__ list, const uint32_t *__restrict__ cnt,
                const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev, int mfirst, int nlev_out,
                const double *__restrict__ tabT, int nk, double *__restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double cthin_lds[];
  constexpr int NT = 2 * MMAX + 1;
  constexpr int TPA = 8;
  __shared__ int s_node[TPA], s_from[TPA], s_to[TPA];
  __shared__ double s_cw[TPA][4], s_trig[TPA][NT + 1];
  const int half = (C.mmax + 1) * C.nmax, ncoef = 2 * half;
  const int nset = nk == 3 ? 1 : 2;
  double *pe = cthin_lds;                                           // [TPA][nset][half]
  const size_t count = cnt[1];
  const int seg = blockIdx.x % CYL_CSEG;
  const int t = threadIdx.x;
  const int nyp = C.numy + 1;
  const size_t per_node = (size_t)nk * half;
  for (size_t base = (size_t)blockIdx.x * TPA; base < count; base += (size_t)gridDim.x * TPA) {
    const int np = (int)((count - base) < (size_t)TPA ? (count - base) : (size_t)TPA);
    if (t < TPA) {
      const bool valid = t < np;
      double xx = 1, yy = 0, zz = 0, mass = 0;
      int from = -1, to = -1;
      if (valid) {
        const uint32_t i = list[base + t];
        cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
        mass = M[i];
        from = lev[i]; to = newlev[i];
      }
      const double r2 = xx * xx + yy * yy;
      double r, ir;
      sqrt_rsqrt(r2, r, ir);
      const bool on = valid && !(sqrt(r2 + zz * zz) > C.rtab_abs);
      double zc = zz;
      if (zc > C.rtab_abs) zc = C.rtab_abs;
      if (zc < -C.rtab_abs) zc = -C.rtab_abs;
      int ix, iy;
      double cw[4];
      cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
      double cphi = 1.0, sphi = 0.0;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      const double t0 = on ? -4.0 * M_PI * mass : 0.0;
      s_node[t] = on ? ix * nyp + iy : -1;
      s_to[t] = on ? to : -1;
      s_from[t] = (on && from >= mfirst) ? from : -1;
#pragma unroll
      for (int k = 0; k < 4; k++) s_cw[t][k] = cw[k];
      double cm = 1.0, sm = 0.0;
      s_trig[t][0] = t0;
#pragma unroll
      for (int m = 1; m <= MMAX; m++) {
        const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi;
        cm = cn; sm = sn;
        const bool m_on = !(C.EVEN_M && (m & 1));
        s_trig[t][2 * m - 1] = m_on ? t0 * cm : 0.0;
        s_trig[t][2 * m] = m_on ? t0 * sm : 0.0;
      }
    }
    __syncthreads();
    for (int it0 = threadIdx.x; it0 < np * nset * half; it0 += 3 * 256) {
      double tv[3][4];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int it = it0 + u * 256;
        tv[u][0] = tv[u][1] = tv[u][2] = tv[u][3] = 0.0;
        if (it < np * nset * half) {
          const int p = it / (nset * half);
          const int rest = it - p * nset * half;
          const int set = rest / half, mn = rest - set * half;
          const int node0 = s_node[p];
          if (node0 >= 0) {
            const double *T = tabT + (size_t)node0 * per_node + (size_t)(set ? 3 : 0) * half + mn;
            tv[u][0] = T[0]; tv[u][1] = T[(size_t)nyp * per_node]; tv[u][2] = T[per_node]; tv[u][3] = T[(size_t)(nyp + 1) * per_node];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int it = it0 + u * 256;
        if (it < np * nset * half) {
          const int p = it / (nset * half);
          pe[it] = s_cw[p][0] * tv[u][0] + s_cw[p][1] * tv[u][1] + s_cw[p][2] * tv[u][2] + s_cw[p][3] * tv[u][3];
        }
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < ncoef; o += 256) {
      const int cs = o / half, mn = o - cs * half, m = mn / C.nmax;
      if (cs && m == 0) continue;                                  // (sin, m = 0): no such row
      const int jt = m == 0 ? 0 : 2 * m - 1 + cs;
      const int set = (cs && nset == 2) ? 1 : 0;
      double v[TPA];
#pragma unroll
      for (int p = 0; p < TPA; p++) v[p] = p < np ? s_trig[p][jt] * pe[((size_t)p * nset + set) * half + mn] : 0.0;
      for (int L = 0; L < nlev_out; L++) {
        const int level = mfirst + L;
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < TPA; p++)
          if (p < np) acc += (s_to[p] == level ? v[p] : 0.0) - (s_from[p] == level ? v[p] : 0.0);
        if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)L * CYL_CSEG + seg) * ncoef + o, acc);
      }
    }
    __syncthreads();
  }
}

// ---- thin active sets, second formulation (round 4; any azimuthal order) ----------------------------------------------
// k_cyl_force_wave: one WAVE per particle.  The lanes own the items (corner k, kind, m) of accumulated_eval's sums: each
// forms its node-row entries sum_n tab[kind][m][n][node_k] {cos, sin}[m][n] (the sums of k_cyl_project, from the node-major
// table copy), weights them with its corner weight and cos / sin(m phi), and the four field sums are reduced over the
// wave; lane 0 applies Cylinder's taper / monopole blend (the tail of k_cyl_force).  No LDS, no barrier.
__global__ void __launch_bounds__(256)
k_cyl_force_wave(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                 const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi, const double *__restrict__ tabT, int nk,
                 const double *__restrict__ coef, const double *__restrict__ cylmass_p, double *__restrict__ AX,
                 double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT, double *__restrict__ VX,
                 double *__restrict__ VY, double *__restrict__ VZ, int assign)
{
  const int lane = threadIdx.x & 63;
  const int M1 = C.mmax + 1, half = M1 * C.nmax, nyp = C.numy + 1;
  const size_t per_node = (size_t)nk * half;
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < end; i += (size_t)gridDim.x * 4) {
    double xx, yy, zz;
    cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
    const double ratmin = 0.75, maxerf = 3.0;                       // src/Cylinder.cc:1357-1381
    const double midpt = ratmin + 0.5 * (1.0 - ratmin);
    const double rsmth = 0.5 * (1.0 - ratmin) / maxerf;
    const double r2 = xx * xx + yy * yy;
    double rp, irp, r3s, ir3s;
    sqrt_rsqrt(r2, rp, irp);
    sqrt_rsqrt(r2 + zz * zz, r3s, ir3s);
    const double r = rp + DSMALL;
    double cphi = 1.0, sphi = 0.0;
    if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
    const double ratio = r3s * C.inv_rtab_abs;
    const bool ongrid = ratio < 1.0 && !(r3s > C.rtab_abs);
    double op = 0.0, ofr = 0.0, ofz = 0.0, ofp = 0.0;
    if (ongrid) {                                                   // (wave-uniform)
      int ix, iy;
      double cw[4];
      cyl_weights(C, r, zz, ix, iy, cw[0], cw[2], cw[1], cw[3]);    // (c00, c10, c01, c11) -> k = 0: 00, 1: 01, 2: 10, 3: 11
      const size_t node0 = (size_t)ix * nyp + iy;
      for (int it = lane; it < 12 * M1; it += 64) {
        const int k = it / (3 * M1), rest = it - k * 3 * M1;
        const int kind = rest / M1, m = rest - kind * M1;
        if (C.EVEN_M && (m & 1)) continue;
        const size_t node = node0 + ((k & 2) ? nyp : 0) + (k & 1);
        const double *T = tabT + node * per_node + ((size_t)kind * M1 + m) * C.nmax;
        const double *cc = coef + (size_t)m * C.nmax, *cs = cc + half;
        double a, b = 0.0;
        if (m == 0) a = cyl_chain4(T, cc, C.nmax);
        else if (nk == 3) cyl_chain4_pair(T, cc, cs, C.nmax, a, b);
        else { a = cyl_chain4(T, cc, C.nmax); b = cyl_chain4(T + (size_t)3 * half, cs, C.nmax); }
        double cm = 1.0, sm = 0.0;
        for (int q = 0; q < m; q++) { const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi; cm = cn; sm = sn; }
        const double w = k == 0 ? cw[0] : k == 1 ? cw[1] : k == 2 ? cw[2] : cw[3];
        const double v = w * (a * cm + b * sm);
        if (kind == 0) { op += v; ofp += w * (a * sm - b * cm) * m; }
        else if (kind == 1) ofr += v;
        else ofz += v;
      }
      for (int off = 32; off > 0; off >>= 1) {
        op += __shfl_xor(op, off); ofr += __shfl_xor(ofr, off);
        ofz += __shfl_xor(ofz, off); ofp += __shfl_xor(ofp, off);
      }
    }
    if (lane != 0) continue;
    double frac, cfrac;
    if (ratio >= 1.0) { frac = 0.0; cfrac = 1.0; }
    else if (ratio > ratmin) { frac = 0.5 * (1.0 - erf((ratio - midpt) / rsmth)); cfrac = 1.0 - frac; }
    else { cfrac = 0.0; frac = 1.0; }
    double fx = 0.0, fy = 0.0, fz = 0.0, pa = 0.0;
    if (ratio < 1.0) {
      const double ir = rcp_refine(r, irp), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
      fx = (ofr * xx * ir - ofp * yy * ir2) * frac;                 // src/Cylinder.cc:1387-1390
      fy = (ofr * yy * ir + ofp * xx * ir2) * frac;
      fz = ofz * frac;
      pa = op * frac;
    }
    if (ratio > ratmin) {                                           // monopole blend, src/Cylinder.cc:1398-1408
      const double p = -(*cylmass_p) * ir3s;
      const double fr = p * (ir3s * ir3s);
      fx += xx * fr * cfrac; fy += yy * fr * cfrac; fz += zz * fr * cfrac;
      pa += p * cfrac;
    }
    if (C.use_rot) {
      const double a = fx, b = fy, c = fz;
      fx = C.rot[0] * a + C.rot[3] * b + C.rot[6] * c;
      fy = C.rot[1] * a + C.rot[4] * b + C.rot[7] * c;
      fz = C.rot[2] * a + C.rot[5] * b + C.rot[8] * c;
    }
    if (C.ps.center | C.ps.axis) {
      double qx, qy, qz, ux = 0.0, uy = 0.0, uz = 0.0;
      if (C.ps.axis) { ux = VX[i]; uy = VY[i]; uz = VZ[i]; }
      pseudo_accel(C.ps, X[i], Y[i], Z[i], ux, uy, uz, qx, qy, qz);
      fx -= qx; fy -= qy; fz -= qz;
    }
    if (!assign) { fx += AX[i]; fy += AY[i]; fz += AZ[i]; pa += POT[i]; }
    AX[i] = fx; AY[i] = fy; AZ[i] = fz; POT[i] = pa;
  }
}

// k_cyl_acc_tile: tiles of up to 64 particles.  Lane t of the first wave prepares particle t (cuts, window, corner weights,
// -4 pi m cos / sin(m phi): k_cyl_mstep_update with plain = 1); the block blends the potential tables at the four
// corners, pe[p][set][m][n], with coalesced reads of the node-major copy; each thread owns coefficients (cs, m, n) and
// sums over the tile's runs of equal level, one atomic per run into part[level - lo][seg][ncoef].
#define CYL_TILE_MAX 64
__global__ void __launch_bounds__(256)
k_cyl_acc_tile(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
               const double *__restrict__ M, const uint32_t *__restrict__ lev_off, int lo, int hi,
               const double *__restrict__ tabT, int nk, double *__restrict__ part, double *__restrict__ tail, int tile)
{
  extern __shared__ __attribute__((aligned(16))) double ctile_lds[];
  __shared__ int s_node[CYL_TILE_MAX], s_run_beg[20], s_run_lev[20], s_nrun;
  __shared__ double s_cw[CYL_TILE_MAX][4];
  const int NT = C.ntrig, tst = NT | 1;
  const int M1 = C.mmax + 1, half = M1 * C.nmax, ncoef = 2 * half, nyp = C.numy + 1;
  const int nset = nk == 3 ? 1 : 2;
  double *trig = ctile_lds;                                         // [tile][tst]: -4 pi m {1, cos phi, sin phi, cos 2 phi, ...}
  double *pe = ctile_lds + (((size_t)tile * tst + 1) & ~(size_t)1);         // [tile][nset][half]
  const size_t per_node = (size_t)nk * half;
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  const int seg = blockIdx.x % CYL_CSEG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (size_t base = beg + (size_t)blockIdx.x * tile; base < end; base += (size_t)gridDim.x * tile) {
    const int np = (int)((end - base) < (size_t)tile ? (end - base) : (size_t)tile);
    if (wave == 0) {
      const size_t i = base + lane;
      const bool valid = lane < np;
      double xx = 1, yy = 0, zz = 0, mass = 0;
      if (valid) { cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz); mass = C.umass != 0.0 ? C.umass : M[i]; }
      const double r2 = xx * xx + yy * yy;
      double r, ir;
      sqrt_rsqrt(r2, r, ir);
      const bool incut = valid && (r2 + zz * zz) < C.rmax2;
      double mu = incut ? mass : 0.0, nu = incut ? 1.0 : 0.0;
      for (int off = 32; off > 0; off >>= 1) { mu += __shfl_xor(mu, off); nu += __shfl_xor(nu, off); }
      if (lane == 0 && nu > 0.0) {
        double *tp_ = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
        unsafeAtomicAdd(tp_ + 0, mu); unsafeAtomicAdd(tp_ + 1, nu);
      }
      const bool on = incut && !(sqrt(r2 + zz * zz) > C.rtab_abs);
      double zc = zz;
      if (zc > C.rtab_abs) zc = C.rtab_abs;
      if (zc < -C.rtab_abs) zc = -C.rtab_abs;
      int ix, iy;
      double cw[4];
      cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
      double cphi = 1.0, sphi = 0.0;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      const double t0 = on ? -4.0 * M_PI * mass : 0.0;
      int lv = lo;
      while (lv < hi && i >= lev_off[lv + 1]) lv++;
      if (!valid) lv = -1;
      const int prev = __shfl_up(lv, 1);
      const bool start = valid && (lane == 0 || lv != prev);
      const unsigned long long starts = __ballot(start);
      if (start) {
        const int rr_ = __popcll(starts & ((1ull << lane) - 1ull));
        if (rr_ < 20) { s_run_beg[rr_] = lane; s_run_lev[rr_] = lv; }
      }
      if (lane == 0) s_nrun = min(20, (int)__popcll(starts));
      if (lane < tile) {
        s_node[lane] = on ? ix * nyp + iy : -1;
#pragma unroll
        for (int k = 0; k < 4; k++) s_cw[lane][k] = cw[k];
        double *tr = trig + (size_t)lane * tst;
        double cm = 1.0, sm = 0.0;
        tr[0] = t0;
        for (int m = 1; m <= C.mmax; m++) {
          const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi;
          cm = cn; sm = sn;
          const bool m_on = !(C.EVEN_M && (m & 1));
          tr[2 * m - 1] = m_on ? t0 * cm : 0.0;
          tr[2 * m] = m_on ? t0 * sm : 0.0;
        }
      }
    }
    __syncthreads();
    // pe: wave w takes particles w, w + 4, ...; two at a time: eight table loads per lane in flight
    for (int p0 = wave * 2; p0 < np; p0 += 8) {
      for (int e0 = lane; e0 < nset * half; e0 += 64) {
        const int set = e0 / half, mn = e0 - set * half;
        double tv[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int p = p0 + u;
          tv[u][0] = tv[u][1] = tv[u][2] = tv[u][3] = 0.0;
          if (p < np) {
            const int node0 = s_node[p];
            if (node0 >= 0) {
              const double *T = tabT + (size_t)node0 * per_node + (size_t)(set ? 3 : 0) * half + mn;
              tv[u][0] = T[0]; tv[u][1] = T[(size_t)nyp * per_node]; tv[u][2] = T[per_node]; tv[u][3] = T[(size_t)(nyp + 1) * per_node];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int p = p0 + u;
          if (p < np)
            pe[((size_t)p * nset + set) * half + mn] =
                s_cw[p][0] * tv[u][0] + s_cw[p][1] * tv[u][1] + s_cw[p][2] * tv[u][2] + s_cw[p][3] * tv[u][3];
        }
      }
    }
    __syncthreads();
    const int nrun = s_nrun;
    for (int o = t; o < ncoef; o += 256) {
      const int cs = o / half, mn = o - cs * half, m = mn / C.nmax;
      if (cs && m == 0) continue;                                  // (sin, m = 0): no such row
      const int jt = m == 0 ? 0 : 2 * m - 1 + cs;
      const int set = (cs && nset == 2) ? 1 : 0;
      for (int r = 0; r < nrun; r++) {
        const int pb = s_run_beg[r], pe_ = r + 1 < nrun ? s_run_beg[r + 1] : np;
        double acc = 0.0;
        for (int p0 = pb; p0 < pe_; p0 += 8) {
          double tt_[8], pp_[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const bool in = p0 + u < pe_;
            tt_[u] = in ? trig[(size_t)(p0 + u) * tst + jt] : 0.0;
            pp_[u] = in ? pe[((size_t)(p0 + u) * nset + set) * half + mn] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) acc = fma(tt_[u], pp_[u], acc);
        }
        if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(s_run_lev[r] - lo) * CYL_CSEG + seg) * ncoef + o, acc);
      }
    }
    __syncthreads();
  }
}

// ---- any azimuthal order: run-time loops over m ------------------------------------------------------------------
// The kernels above are instantiated for mmax <= CYL_MAX_M; the reference takes any `mmax` (src/Cylinder.cc:473,
// exputil/EmpCylSL.cc:343-420).  Above CYL_MAX_M (and, for tests, at any order with EXP_AMD_CYL_GENERIC=1) every
// per-particle pass goes through these two plain kernels -- one particle per lane, node moments by atomics, node rows by
// gathers -- with the same cuts, windows and operations as k_cyl_mstep_update (differencing, plain accumulation) and
// k_cyl_force.  The contraction and projection kernels never depended on the order.
__global__ void __launch_bounds__(256)
k_cyl_moments_gen(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                  const double *__restrict__ M, const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev,
                  const uint32_t *__restrict__ lev_off, int first, int last, int mfirst, double *__restrict__ Wnd,
                  int plain /* 1: every particle into Wnd[its level]; 2: ... into Wnd[wlevel] (single-level buffers) */,
                  int wlevel, double *__restrict__ tail, const uint32_t *__restrict__ list)
{
  const int NT = C.ntrig;
  size_t i = 0;
  bool have = false;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (list) {
    if (g < lev_off[1]) { i = list[g]; have = true; }
  } else {
    i = lev_off[first] + g;
    have = i < lev_off[last + 1];
  }
  bool mover = false;
  int from = 0, to = 0;
  if (have) {
    from = plain == 2 ? wlevel : lev[i];
    to = plain ? from : newlev[i];
    mover = plain || from != to;
  }
  if (!__any(mover)) return;
  double xx = 1, yy = 0, zz = 0, mass = 0;
  if (mover) { cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz); mass = C.umass != 0.0 ? C.umass : M[i]; }
  const double r2 = xx * xx + yy * yy;
  double r, ir;
  sqrt_rsqrt(r2, r, ir);
  if (plain) {
    const bool incut = mover && (r2 + zz * zz) < C.rmax2;
    double mu = incut ? cdet_round(mass, C.detCm) : 0.0, nu = incut ? 1.0 : 0.0;
    for (int off = 32; off > 0; off >>= 1) { mu += __shfl_xor(mu, off); nu += __shfl_xor(nu, off); }
    if ((threadIdx.x & 63) == 0 && nu > 0.0 && tail) {
      double *tp = tail + 2 * (blockIdx.x & (CYL_TAILS - 1));
      unsafeAtomicAdd(tp + 0, mu); unsafeAtomicAdd(tp + 1, nu);
    }
    mover = incut;
  }
  if (sqrt(r2 + zz * zz) > C.rtab_abs) mover = false;
  if (!__any(mover)) return;
  double zc = zz;
  if (zc > C.rtab_abs) zc = C.rtab_abs;
  if (zc < -C.rtab_abs) zc = -C.rtab_abs;
  int ix, iy;
  double cw[4];
  cyl_weights(C, r, zc, ix, iy, cw[0], cw[1], cw[2], cw[3]);
  double cphi = 1.0, sphi = 0.0;
  if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
  const double t0 = mover ? -4.0 * M_PI * mass : 0.0;
  const int nyp = C.numy + 1;
  const size_t nnode = (size_t)(C.numx + 1) * nyp;
  if (!mover) return;
  const bool sub = !plain && from >= mfirst;
  double *wto = Wnd + ((size_t)to * nnode + (size_t)(ix * nyp + iy)) * NT;
  double *wfr = Wnd + ((size_t)from * nnode + (size_t)(ix * nyp + iy)) * NT;
  double cm = 1.0, sm = 0.0;
  for (int m = 0; m <= C.mmax; m++) {
    if (m > 0) {
      const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi;
      cm = cn; sm = sn;
    }
    if (C.EVEN_M && (m & 1)) continue;
    const int jc = (m == 0) ? 0 : 2 * m - 1;
    for (int k = 0; k < 4; k++) {
      const size_t off = (size_t)(((k & 1) ? nyp : 0) + ((k & 2) ? 1 : 0)) * NT;
      const double w = t0 * cw[k];
      const double vc = cdet_round(w * cm, C.detC), vs = cdet_round(w * sm, C.detC);
      unsafeAtomicAdd(wto + off + jc, vc);
      if (sub) unsafeAtomicAdd(wfr + off + jc, -vc);
      if (m > 0) {
        unsafeAtomicAdd(wto + off + jc + 1, vs);
        if (sub) unsafeAtomicAdd(wfr + off + jc + 1, -vs);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
k_cyl_force_gen(CylDev C, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi, const double *__restrict__ TF,
                const double *__restrict__ cylmass_p, double *__restrict__ AX, double *__restrict__ AY,
                double *__restrict__ AZ, double *__restrict__ POT, double *__restrict__ VX, double *__restrict__ VY,
                double *__restrict__ VZ, double dt_kick, int assign, uint32_t *__restrict__ key_out, double nk_dtk,
                double nk_dtd, int store_v)
{
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const size_t i = beg + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= end) return;
  double xx, yy, zz;
  cyl_local(C, X[i], Y[i], Z[i], xx, yy, zz);
  const double ratmin = 0.75, maxerf = 3.0;                         // src/Cylinder.cc:1357-1381
  const double midpt = ratmin + 0.5 * (1.0 - ratmin);
  const double rsmth = 0.5 * (1.0 - ratmin) / maxerf;
  const double r2 = xx * xx + yy * yy;
  double rp, irp, r3s, ir3s;
  sqrt_rsqrt(r2, rp, irp);
  sqrt_rsqrt(r2 + zz * zz, r3s, ir3s);
  const double r = rp + DSMALL;
  double cphi = 1.0, sphi = 0.0;
  if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
  const double ratio = r3s * C.inv_rtab_abs;
  double frac, cfrac;
  if (ratio >= 1.0) { frac = 0.0; cfrac = 1.0; }
  else if (ratio > ratmin) { frac = 0.5 * (1.0 - erf((ratio - midpt) / rsmth)); cfrac = 1.0 - frac; }
  else { cfrac = 0.0; frac = 1.0; }
  const bool ongrid = ratio < 1.0 && !(r3s > C.rtab_abs);
  int ix, iy;
  double c00, c10, c01, c11;
  cyl_weights(C, r, zz, ix, iy, c00, c10, c01, c11);
  const int NF = 3 * C.ntrig, nyp = C.numy + 1;
  CylOut o{0.0, 0.0, 0.0, 0.0};
  if (ongrid) {
    // cyl_field with a run-time m loop (exputil/EmpCylSL.cc:5318-5400)
    const double *t00 = TF + ((size_t)ix * nyp + iy) * NF;
    const double *t01 = t00 + NF, *t10 = t00 + (size_t)nyp * NF, *t11 = t10 + NF;
    double cm = 1.0, sm = 0.0;
    for (int m = 0; m <= C.mmax; m++) {
      if (m > 0) {
        const double cn = cm * cphi - sm * sphi, sn = sm * cphi + cm * sphi;
        cm = cn; sm = sn;
      }
      if (C.EVEN_M && (m & 1)) continue;
      const int q = (m == 0) ? 0 : 3 + 6 * (m - 1);
      auto bl = [&](int k) { return c00 * t00[q + k] + c10 * t10[q + k] + c01 * t01[q + k] + c11 * t11[q + k]; };
      const double Pc = bl(0), Rc = bl(1), Zc = bl(2);
      if (m == 0) { o.p += Pc; o.fr += Rc; o.fz += Zc; }
      else {
        const double Ps = bl(3), Rs = bl(4), Zs = bl(5);
        o.p += Pc * cm + Ps * sm;
        o.fr += Rc * cm + Rs * sm;
        o.fz += Zc * cm + Zs * sm;
        o.fp += (Pc * sm - Ps * cm) * m;
      }
    }
  }
  double fx = 0.0, fy = 0.0, fz = 0.0, pa = 0.0;
  if (ratio < 1.0) {
    double p = 0.0, fr = 0.0, fzz = 0.0, fp = 0.0;
    if (ongrid) { p = o.p; fr = o.fr; fzz = o.fz; fp = o.fp; }
    const double ir = rcp_refine(r, irp), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
    fx = (fr * xx * ir - fp * yy * ir2) * frac;                     // src/Cylinder.cc:1387-1390
    fy = (fr * yy * ir + fp * xx * ir2) * frac;
    fz = fzz * frac;
    pa = p * frac;
  }
  if (ratio > ratmin) {                                             // monopole blend, src/Cylinder.cc:1398-1408
    const double p = -(*cylmass_p) * ir3s;
    const double fr = p * (ir3s * ir3s);
    fx += xx * fr * cfrac;
    fy += yy * fr * cfrac;
    fz += zz * fr * cfrac;
    pa += p * cfrac;
  }
  if (C.use_rot) {
    const double a = fx, b = fy, c = fz;
    fx = C.rot[0] * a + C.rot[3] * b + C.rot[6] * c;
    fy = C.rot[1] * a + C.rot[4] * b + C.rot[7] * c;
    fz = C.rot[2] * a + C.rot[5] * b + C.rot[8] * c;
  }
  if (C.ps.center | C.ps.axis) {
    double qx, qy, qz, ux = 0.0, uy = 0.0, uz = 0.0;
    if (C.ps.axis) { ux = VX[i]; uy = VY[i]; uz = VZ[i]; }
    pseudo_accel(C.ps, X[i], Y[i], Z[i], ux, uy, uz, qx, qy, qz);
    fx -= qx; fy -= qy; fz -= qz;
  }
  if (!assign) { fx += AX[i]; fy += AY[i]; fz += AZ[i]; pa += POT[i]; }
  AX[i] = fx; AY[i] = fy; AZ[i] = fz; POT[i] = pa;
  if (dt_kick != 0.0) {
    const double vx = mul_then_add(VX[i], fx, dt_kick);
    const double vy = mul_then_add(VY[i], fy, dt_kick);
    const double vz = mul_then_add(VZ[i], fz, dt_kick);
    if (store_v == 1) { VX[i] = vx; VY[i] = vy; VZ[i] = vz; }
    if (key_out) {
      const double wx = mul_then_add(vx, fx, nk_dtk);
      const double wy = mul_then_add(vy, fy, nk_dtk);
      const double wz = mul_then_add(vz, fz, nk_dtk);
      if (store_v == 2) { VX[i] = wx; VY[i] = wy; VZ[i] = wz; }
      CylKeyFn kf{C, 0u};
      key_out[i] = kf(mul_then_add(X[i], wx, nk_dtd), mul_then_add(Y[i], wy, nk_dtd), mul_then_add(Z[i], wz, nk_dtd), 0);
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------

struct CylForce : exp_amd_force {
  exp_amd_cyl_config cfg{};
  CylDev dev{};
  DevBuf<double> d_tab, d_Wn, d_TF;
  bool tab_twin = false;            // the three sine tables equal the three cosine tables bit for bit (m >= 1)
  DevBuf<double> d_cpart;           // stage-1 sums of the contraction: [level][CYL_CSEG][ncoef]
  DevBuf<uint32_t> d_work;          // work list of the force pass' tail launch (lanes beyond 0.75 of the table radius)
  size_t work_cap = 0;              // ... in waves; the two counters behind it are used alternately
  int work_flip = 0;
  int step_parity() const override { return work_flip; }
  bool generic = false;             // mmax > CYL_MAX_M (or EXP_AMD_CYL_GENERIC=1): the run-time-order kernels throughout
  bool external_shares_no_scratch() const override { return !ctx->deterministic; }
  bool adv_owed = false;            // substep_expansion: the advance of the active range is left to k_cyl_acc_thin
  double adv_dt_min = 0.0;
  bool cpart_clean = false;         // ... all zero (what k_cyl_acc_thin adds to; its summing kernels keep them so)
  DevBuf<double> d_tabT;            // node-major copy tabT[node][kind][m][n] for the thin path (made on first use)
  int tabT_nk = 0;                  // kinds it holds: 3 (sine tables == cosine tables) or 6
  int ensure_tabT();
  DevBuf<double> d_Wnd, d_differ;   // multistep differencing
  DevBuf<double> d_dens;            // densC / densS tables (field evaluation only)
  // sub-sample covariance (pyEXP pcavar, analysis only): node moments U[T][node][ntrig], cell
  // moments Q[T][cell][10], counts / masses [T], results
  int cov_T = 0;
  DevBuf<double> cov_U, cov_Q, cov_mass, cov_vc, cov_mv;
  DevBuf<unsigned long long> cov_cnt, cov_used;
  DevBuf<uint32_t> cov_seq;
  size_t cov_seq_cap = 0;
  DevBuf<double> d_mass;            // {cylmass, used}: in-cut mass / count of the current master step
  DevBuf<double> d_tailpart;        // [CYL_TAILS][2]: the same tallies of ONE accumulation launch, spread over slots
  bool mass_open = true;            // still within the first sub-step (tnow == resetT)
  bool wn_clean = false;            // every per-level moment buffer of d_Wn is zero (substep_expansion's contraction keeps it so)
  bool wnd_clean = false;           // ... and d_Wnd, d_differ's tails (multistep_update)
  bool tails_clean = false;         // the {mass, count} tails of all expcoefN sets are zero (substep_expansion keeps them so)
  size_t nnode = 0;

  int determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift,
                             bool have_keys = false) override;
  int accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick, double nk_dtk = 0.0,
                 double nk_dtd = 0.0, bool *prekey_done = nullptr, bool defer_kick = false) override;
  int multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft) override;
  int substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine = -1, int phase = 0) override;
  long long sparse_threshold() const override { return 3000000LL / (4 * dev.ntrig); }
  int resort(exp_amd_comp *c, int first = 0) override;
  int multistep_reset() override
  {
    // Cylinder::multistep_reset: used = 0, cylmass = 0, resetT = tnow (src/Cylinder.cc:1209-1216)
    HIP_TRY(ctx, hipMemsetAsync(d_mass.p, 0, 2 * sizeof(double), ctx->stream));
    mass_open = true;
    return EXP_AMD_OK;
  }
  int sort(exp_amd_comp *c, bool move_acc, const AdvSpec &adv, int level = -1, bool have_keys = false,
           int level_hi = -1);
  void release() override
  {
    cov_U.release(); cov_Q.release(); cov_mass.release(); cov_vc.release(); cov_mv.release();
    cov_cnt.release(); cov_used.release(); cov_seq.release();
    d_tab.release(); d_Wn.release(); d_TF.release(); d_Wnd.release(); d_differ.release(); d_cpart.release();
    d_tabT.release(); d_work.release();
    d_mass.release();
    d_tailpart.release();
    d_dens.release();
  }
  int get_used(long long *used) override
  {
    double u = 0.0;
    HIP_TRY(ctx, hipMemcpyAsync(&u, d_mass.p + 1, sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *used = (long long)(u + 0.5);
    return EXP_AMD_OK;
  }
};

static CylDev cdev_frame(const CylForce *f, const double *center, bool use_rot, const double *rot)
{
  CylDev C = f->dev;
  C.cx = center[0]; C.cy = center[1]; C.cz = center[2];
  C.use_rot = use_rot ? 1 : 0;
  for (int k = 0; k < 9; k++) C.rot[k] = rot[k];
  return C;
}

static CylDev cdev_for(const CylForce *f, const exp_amd_comp *c)
{
  return cdev_frame(f, c->center, c->use_rot, c->rot);
}

// ... for the passes that ADD particle contributions: with the deterministic mode on, the rounding
// grids that keep every partial sum of this component exact (|-4 pi m c_k trig| <= 4 pi |m| x 2 for the
// bilinear weights; the in-cut mass itself)
static CylDev cdev_acc(const CylForce *f, const exp_amd_comp *c)
{
  CylDev C = cdev_for(f, c);
  C.detC = expamd_det_constant(f->ctx->deterministic, c->mass_abs_sum * 4.0 * M_PI * 2.0);
  C.detCm = expamd_det_constant(f->ctx->deterministic, c->mass_abs_sum);
  C.umass = c->uniform_mass ? c->mass_value : 0.0;
  return C;
}

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
  C.rmax2 = cfg->rcylmax * cfg->rcylmax * cfg->ascale * cfg->ascale;   // src/Cylinder.cc:752
  C.cx = C.cy = C.cz = 0.0;
  *out = f;
  return EXP_AMD_OK;
}

#define MMAX_DISPATCH(M, CALL)                                                        \
  switch (M) {                                                                        \
    case 0: CALL(0); break;  case 1: CALL(1); break;  case 2: CALL(2); break;        \
    case 3: CALL(3); break;  case 4: CALL(4); break;  case 5: CALL(5); break;        \
    case 6: CALL(6); break;  case 7: CALL(7); break;  case 8: CALL(8); break;        \
    case 9: CALL(9); break;  case 10: CALL(10); break; case 11: CALL(11); break;     \
    case 12: CALL(12); break;                                                        \
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
  static const int tp0 = [] { const char *e = getenv("EXP_AMD_THIN_TP"); return e ? atoi(e) : 4; }();
  const int tp = tp0 < 1 ? 1 : tp0 > 64 ? 64 : tp0;      // (small tiles: see the spherical launcher)
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  const size_t lds = (2 * half + (size_t)tp * (4 * (3 * (2 * MM + 1) + 1) + 2)) * sizeof(double);
  size_t grid = cdiv(n, (size_t)tp);
  if (grid > 16384) grid = 16384;
  if (grid == 0) return;
  static const bool big = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cyl_force_thin<MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return true;
  }();
  (void)big;
  static const int nt0 = [] { const char *e = getenv("EXP_AMD_THIN_NT"); return e ? atoi(e) : 0; }();
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
  static const int tpa0 = [] { const char *e = getenv("EXP_AMD_THIN_TPA"); return e ? atoi(e) : 8; }();
  int tpa = tpa0 < 1 ? 1 : tpa0 > 64 ? 64 : tpa0;
  while (tpa > 4 && (size_t)tpa * nset * half * sizeof(double) > 96 * 1024) tpa >>= 1;
  size_t grid = cdiv(n, (size_t)tpa);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cyl_acc_thin<MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return true;
  }();
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
  static const bool big = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cyl_diff_thin<MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return true;
  }();
  (void)big;
  k_cyl_diff_thin<MM><<<(unsigned)grid, 256, (size_t)8 * nset * half * sizeof(double), st>>>(C, X, Y, Z, M, list, cnt, lev, newlev,
                                                                                              mfirst, nl, tabT, nk, part);
}

static int cyl_thin_version()
{
  static const int v = [] { const char *e = getenv("EXP_AMD_THIN_V"); return e ? atoi(e) : 1; }();
  return v;
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
  static const int tile0 = [] { const char *e = getenv("EXP_AMD_THIN_TILE"); return e ? atoi(e) : 64; }();
  int tile = tile0 < 4 ? 4 : tile0 > CYL_TILE_MAX ? CYL_TILE_MAX : tile0;
  auto need = [&](int t) { return ((((size_t)t * (C.ntrig | 1) + 1) & ~(size_t)1) + (size_t)t * nset * half) * sizeof(double); };
  while (tile > 4 && need(tile) > 120 * 1024) tile >>= 1;
  size_t grid = cdiv(n, (size_t)tile);
  if (grid > 4096) grid = 4096;
  if (grid == 0) return;
  static const bool big = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cyl_acc_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return true;
  }();
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
    k_hist_keys<<<cdiv(c->n, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(c->key.p, c->n, c->hist.p);
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
  static const bool thin_diff_on = [] { const char *e = getenv("EXP_AMD_THIN_DIFF"); return !e || atoi(e) != 0; }();
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
      rc = sort(c, /*move_acc=*/full && lo > 0, adv, full ? -1 : lo, false, dmax);
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
      static const bool fuse_on = [] { const char *e = getenv("EXP_AMD_THIN_ADVANCE"); return !e || atoi(e) != 0; }();
      const bool fuse = fuse_on && dmax < lo && adv.mode == 2 && nall > 0 && ctx->thin_max > 0 && (long long)nall <= ctx->thin_max * ctx->thin_acc_scale &&
                        !ctx->deterministic && cyl_thin_version() == 1 && !f->generic;
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
  const bool thin = dacc < lo && (long long)nrange <= ctx->thin_max * ctx->thin_acc_scale && !ctx->deterministic && ctx->thin_max > 0;
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
      if (cyl_thin_version() != 1 || f->generic)
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
    if (!external && f->ev_tables) HIP_TRY(ctx, hipEventRecord(f->ev_tables, ctx->stream));
    f->mass_open = false;
    if (nthin) {
      ProfScope ps(ctx, "k_cyl_force_thin");
      CylDev C = !external ? cdev_for(f, t) : f->home ? cdev_for(f, f->home)
                 : f->home_gone ? cdev_frame(f, f->home_center, f->home_use_rot, f->home_rot) : cdev_for(f, t);
      C.ps = t->pseudo;
      if (cyl_thin_version() != 1 || f->generic)
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
  if (!external && f->ev_tables) HIP_TRY(ctx, hipEventRecord(f->ev_tables, ctx->stream));
  f->mass_open = false;          // tnow has moved past resetT once forces are evaluated
  if (t->n == 0) return EXP_AMD_OK;
  // external target: positions go into the frame of the component the expansion was built from
  CylDev C = !external ? cdev_for(f, t) : f->home ? cdev_for(f, f->home)
             : f->home_gone ? cdev_frame(f, f->home_center, f->home_use_rot, f->home_rot) : cdev_for(f, t);
  C.ps = t->pseudo;
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
    const double cth = c2[i], sth = sqrt(1.0 - cth * cth);
    R = r * sth; z = r * cth; phi = c3[i];
  } else if (coord == 1) {
    R = c1[i]; z = c2[i]; phi = c3[i];
  } else {
    x = c1[i]; y = c2[i]; z = c3[i];
    R = sqrt(x * x + y * y);
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
