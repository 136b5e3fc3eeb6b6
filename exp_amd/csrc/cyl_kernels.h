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
#pragma once
#include "cyl_dev.h"

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
      cyl_local_acc(C, nx, ny, nz, xx, yy, zz);
      mass = nm * C.mscale;
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
    double cphi, sphi;                          // phi = atan2(y, x)
    if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
    else atan2_trig_zero(xx, yy, cphi, sphi);
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
// Trig rows [m][68] (cos, sin) pairs, 1088 B apart: the owner lanes' 16-byte reads (ds_read_b128: four groups of 16 lanes,
// 16 slots of 16 B per 256-B bank row) then land on slot (4 m + sub) mod 16 -- the four sub-groups and the four azimuthal
// orders of a lane group all on different slots.  With the former stride of 65 pairs the slot was (m + sub) mod 16: up to
// three lanes of a group on one slot, a third of this kernel's LDS cycles (profiles/r05_cfg3_counters.txt:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32).
#define CSLOT_TSTRIDE 68

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
  // per wave: w[64][4 (+2: rows 48 B apart)] (3 KB), trig pairs [MMAX+1][68][2], flush scratch [64][4] (2 KB).  The particle
  // lanes store their w row with two 16-byte writes (eight contiguous lanes a group, banks mod 32): rows of 32 B put lanes
  // l and l + 4 on the same banks, rows of 48 B none (12 l mod 32 is a permutation of the eight 4-bank slots); the owners'
  // reads -- slot (3 p + k) mod 16 for the four particles p of a lane group -- stay conflict-free
  constexpr int WS = 6;
  constexpr int WB = 64 * WS, TB = (MMAX + 1) * CSLOT_TSTRIDE * 2, SB = 64 * 4;
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
  // (idle lanes accumulate nothing; they read the row whose slot their own row would have had -- m - 4 -- so that they do
  // not add a second address to a slot another lane of their group reads)
  const int orow = om <= MMAX ? om : (om - 4 <= MMAX ? om - 4 : 0);
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
      cyl_local_acc(C, nx[slot], ny[slot], nz[slot], xx, yy, zz);
      mass = nm[slot] * C.mscale;
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
    double cphi, sphi;
    if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
    else atan2_trig_zero(xx, yy, cphi, sphi);
    const double t0 = ongrid ? norm * mass : 0.0;
#if CSLOT_EXPT == 2        // timing experiment: stream + per-particle inputs, no LDS, no sums
    mass_used += t0 * (c00 + c10 + c01 + c11) + cphi + sphi + (double)cell; continue;
#endif
    // this particle's row of the two LDS tables
    {
      double *w = wbuf + lane * WS;
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
        const double *w = wbuf + p * WS;
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
    cyl_local_acc(C, X[i], Y[i], Z[i], xx, yy, zz);
    mass = M[i] * C.mscale;
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
  double cphi, sphi;
  if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
  else atan2_trig_zero(xx, yy, cphi, sphi);
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
  // The block's values go out through LDS: a thread's own stores would be six 8-byte writes 312 bytes apart from its
  // neighbours' (NF doubles a node) -- 64 sectors touched per store instruction; node by node, the six doubles of a
  // harmonic are contiguous (33 -> 24 us at 257 x 129 nodes, mmax 6, nmax 12.  One wave per harmonic and 64 nodes per
  // block, so that whole node rows leave contiguously, was slower: 36 us).
  __shared__ double tile[256 * 6];
  const size_t nnode = (size_t)(C.numx + 1) * (C.numy + 1);
  const size_t node0 = (size_t)blockIdx.x * 256;
  const size_t node = node0 + threadIdx.x;
  const bool in = node < nnode;
  const int m = blockIdx.y;
  const int NF = 3 * C.ntrig;
  const int q0 = (m == 0) ? 0 : 3 + 6 * (m - 1);
  const int nk = (m == 0) ? 3 : 6;
  const size_t half = (size_t)(C.mmax + 1) * C.nmax;
  double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (in && twin && m > 0) {
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
      v[kind] = (a[0] + a[1]) + (a[2] + a[3]);
      v[kind + 3] = (b[0] + b[1]) + (b[2] + b[3]);
    }
  } else if (in) {
    for (int kind = 0; kind < nk; kind++) {
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
      v[kind] = (a[0] + a[1]) + (a[2] + a[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) if (k < nk) tile[threadIdx.x * nk + k] = v[k];
  __syncthreads();
  const int nvalid = nnode - node0 < 256 ? (int)(nnode - node0) : 256;
  for (int e = threadIdx.x; e < nvalid * nk; e += 256) {
    const int nd = e / nk, k = e - nd * nk;
    TF[(node0 + nd) * NF + q0 + k] = tile[e];
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
  double cphi, sphi;
  if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
  else atan2_trig_zero(xx, yy, cphi, sphi);
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
    const double ir = cyl_inv_r(r, irp, r2), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
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
  // (a frozen target particle is skipped by the thread body, src/Cylinder.cc:1329: nothing is added)
  if (C.frz && cyl_frozen(C, X[i], Y[i], Z[i])) { fx = fy = fz = 0.0; pa = 0.0; }
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
    double cphi, sphi;
    if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
    else atan2_trig_zero(xx, yy, cphi, sphi);
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
        const double ir = cyl_inv_r(r, irp, r2), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
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
      if (C.frz && cyl_frozen(C, X[i], Y[i], Z[i])) { fx = fy = fz = 0.0; pa = 0.0; }
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
        cyl_local_acc(C, px, py, pz, xx, yy, zz);
        mass = M[i] * C.mscale;
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
      double cphi, sphi;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      else atan2_trig_zero(xx, yy, cphi, sphi);
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
        cyl_local_acc(C, X[i], Y[i], Z[i], xx, yy, zz);
        mass = M[i] * C.mscale;
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
      double cphi, sphi;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      else atan2_trig_zero(xx, yy, cphi, sphi);
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
    double cphi, sphi;
    if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
    else atan2_trig_zero(xx, yy, cphi, sphi);
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
      const double ir = cyl_inv_r(r, irp, r2), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
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
    if (C.frz && cyl_frozen(C, X[i], Y[i], Z[i])) { fx = fy = fz = 0.0; pa = 0.0; }
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
      if (valid) { cyl_local_acc(C, X[i], Y[i], Z[i], xx, yy, zz); mass = (C.umass != 0.0 ? C.umass : M[i]) * C.mscale; }
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
      double cphi, sphi;
      if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
      else atan2_trig_zero(xx, yy, cphi, sphi);
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
  if (mover) { cyl_local_acc(C, X[i], Y[i], Z[i], xx, yy, zz); mass = (C.umass != 0.0 ? C.umass : M[i]) * C.mscale; }
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
  double cphi, sphi;
  if (r2 > 0.0) { cphi = xx * ir; sphi = yy * ir; }
  else atan2_trig_zero(xx, yy, cphi, sphi);
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
  double cphi, sphi;
  if (r2 > 0.0) { cphi = xx * irp; sphi = yy * irp; }
  else atan2_trig_zero(xx, yy, cphi, sphi);
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
    const double ir = cyl_inv_r(r, irp, r2), ir2 = r2 > 0.0 ? irp * irp : __builtin_inf();
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
  if (C.frz && cyl_frozen(C, X[i], Y[i], Z[i])) { fx = fy = fz = 0.0; pa = 0.0; }
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

