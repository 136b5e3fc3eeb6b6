// Spherical BFE force method (sphereSL) for gfx950: coefficient accumulation and
// force/potential evaluation, from scratch.
//
// What the reference does per particle (CPU: src/SphericalBasis.cc:429-599 and :1476-1660
// with SLGridSph::get_pot/get_force, exputil/SLGridMP2.cc:872-989):
//   accumulate : c[lm][n] += -4pi m Y_lm(theta,phi) * lerp_i(ef_l(n,.))/sqrt(ev) * lerp_i(p0)
//   force      : p_lm = sum_n potd(l,n) c[lm][n],  dp_lm = sum_n dpot(l,n) c[lm][n], ...
// i.e. O(L^2 nmax) table work per particle.  Both radial look-ups are LINEAR in the table
// rows of the particle's radial cell i, so the n-contraction commutes with the particle sum:
//
//   accumulate : W[i][lm][0] += t x1,  W[i][lm][1] += t x2      (t = -4pi m Y_lm P0, per particle)
//                c[lm][n]     = sum_i E[i][l][n] W[i][lm][0] + E[i+1][l][n] W[i][lm][1]   (once)
//   force      : G[i][lm] = sum_n E[i][l][n] c[lm][n],  H[i][lm] = p0[i] G[i][lm]         (once)
//                p_lm  = P0 (x1 G[i][lm] + x2 G[i+1][lm])
//                dp_lm = b0 H[j-1][lm] + b1 H[j][lm] + b2 H[j+1][lm]                       (per particle)
//
// (E[i][l][n] = ef_l(n,i)/sqrt(ev_l[n]).)  Per-particle work drops from O(L^2 nmax) to
// O(L^2); the result differs from the reference only by floating-point re-association.
// Particles are kept sorted by radial cell (particles.hip), so a 64-lane wave sees ONE cell:
// its moment sums stay in registers (one LDS-transposed flush per cell change) and its G/H
// rows are wave-uniform scalar loads.  Waves that straddle cells fall back to a ballot
// waterfall (accumulate) or per-lane row gathers (force); any particle order is correct.
#pragma once
#include "particles.h"

#include <type_traits>
#include <utility>

#define DSMALL 1.0e-16                 // src/expand.H:130
#define MINEPS (3.0 * 2.220446049250313e-16)   // src/Basis.cc:7

typedef const __attribute__((address_space(4))) double *cdp;   // constant (scalar-loadable)

#define SPH_MAX_L 12

struct SphDev {
  int lmax, nmax, numr, cmap, nrows;
  double rmap, scale, rmin, rmax, xmin, dxi;
  double cx, cy, cz;
  int NO_L0, NO_L1, EVEN_L, EVEN_M, M0_only;
  const double *xi;      // [numr]
  const double *p0;      // [numr]
  const double *E;       // [numr][lmax+1][nmax]
  const double *fact;    // [(lmax+1)*(lmax+1)]  factorial(l,m), src/SphericalBasis.cc:328-335
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__host__ __device__ constexpr int rows_of_m(int L, int m) { return (L - m + 1) * (m == 0 ? 1 : 2); }
__host__ __device__ constexpr int acc_base(int L, int mlo, int m)
{
  int s = 0;
  for (int q = mlo; q < m; q++) s += rows_of_m(L, q);
  return s;
}
__host__ __device__ constexpr int row_of(int l, int m, int cs) { return l * l + (m ? 2 * m - 1 + cs : 0); }

// ---- per-particle radial/angular coordinates --------------------------------------------------

struct Coord {
  double xx, yy, zz, r, costh, cphi, sphi;
  double xi;
  int idx;            // get_pot cell: clamp [0, numr-2]   (exputil/SLGridMP2.cc:889-891)
};

__device__ __forceinline__ double sph_r_to_xi(const SphDev &S, double r)
{
  // exputil/SLGridMP2.cc:711-727
  if (S.cmap == 1) return (r / S.rmap - 1.0) / (r / S.rmap + 1.0);
  if (S.cmap == 2) return log(r);
  return r;
}

__device__ __forceinline__ double sph_d_xi_to_r(const SphDev &S, double xi)
{
  // exputil/SLGridMP2.cc:749-765
  if (S.cmap == 1) return 0.5 * (1.0 - xi) * (1.0 - xi) / S.rmap;
  if (S.cmap == 2) return exp(-xi);
  return 1.0;
}

__device__ __forceinline__ int sph_cell(const SphDev &S, double xi)
{
  int idx = (int)((xi - S.xmin) / S.dxi);
  if (idx < 0) idx = 0;
  if (idx > S.numr - 2) idx = S.numr - 2;
  return idx;
}

// cos(phi), sin(phi) for phi = atan2(y, x) without the transcendental round trip
__device__ __forceinline__ void phi_trig(double xx, double yy, double &c, double &s)
{
  double R2 = xx * xx + yy * yy;
  if (R2 > 0.0) {
    double R = sqrt(R2);
    c = xx / R;
    s = yy / R;
  } else {
    c = 1.0;
    s = 0.0;
  }
}

// ---- accumulation ----------------------------------------------------------------------------------

#define FLUSH_STRIDE 68      // doubles per scratch row: conflict-free for the 4x16 read pattern

// Reduce NV per-lane values over the wave and atomically add them to dst[map(j)].
// scratch: wave-private LDS, 16*FLUSH_STRIDE doubles.
template <int NV, class MapFn>
__device__ __forceinline__ void wave_flush(double (&v)[NV], double *scratch, double *dst, MapFn map)
{
  const int lane = threadIdx.x & 63;
  const int kk = lane >> 2, q = lane & 3;
  static_for<0, (NV + 15) / 16>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    static_for<0, 16>([&](auto jc) {
      constexpr int j = g * 16 + decltype(jc)::value;
      if constexpr (j < NV) scratch[decltype(jc)::value * FLUSH_STRIDE + lane] = v[j];
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
    if (g * 16 + kk < NV) {
#pragma unroll
      for (int e = 0; e < 16; e++) s += scratch[kk * FLUSH_STRIDE + q + 4 * e];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q == 0 && g * 16 + kk < NV && s != 0.0) unsafeAtomicAdd(dst + map(g * 16 + kk), s);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  });
#pragma unroll
  for (int j = 0; j < NV; j++) v[j] = 0.0;
}

// accumulator j = 2*k + s (s: 0 -> x1 moment, 1 -> x2 moment), k counts rows in m-major order
template <int LMAX, int MLO, int MHI>
__device__ __forceinline__ int acc_to_wrow(int j)
{
  int k = j >> 1, s = j & 1;
  // invert the m-major enumeration
  int m = MLO, base = 0;
  bool found = false;
#pragma unroll
  for (int q = MLO; q <= MHI; q++) {
    int r = rows_of_m(LMAX, q);
    if (!found) {
      if (k >= base + r) { base += r; m = q + 1; }
      else found = true;
    }
  }
  int rem = k - base;
  int l, cs;
  if (m == 0) { l = rem; cs = 0; }
  else { l = m + (rem >> 1); cs = rem & 1; }
  return row_of(l, m, cs) * 2 + s;
}

#define ACC_WAVES 4
#define ACC_CHUNK 1024        // particles per wave (contiguous, so one or two cells per wave)

// One wave accumulates the rows with m in [MLO, MHI] over the particle chunk [cbeg, cend).
template <int LMAX, int MLO, int MHI>
__device__ __forceinline__ void
sph_accumulate_wave(const SphDev &S, const double *__restrict__ X, const double *__restrict__ Y,
                    const double *__restrict__ Z, const double *__restrict__ M, size_t cbeg,
                    size_t cend, double *scratch, double *__restrict__ W,
                    unsigned long long *__restrict__ used_out)
{
  constexpr int NACC = acc_base(LMAX, MLO, MHI + 1);
  constexpr int NV = 2 * NACC;
  const int lane = threadIdx.x & 63;

  cdp fact = (cdp)S.fact;
  const double fac0 = -4.0 * M_PI;

  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) acc[j] = 0.0;
  int cur = -1;
  unsigned long long used = 0;

  for (size_t base = cbeg; base < cend; base += 64) {
    const size_t i = base + lane;
    const bool valid = i < cend;
    double xx = 0, yy = 0, zz = 1, mass = 0;
    if (valid) {
      xx = X[i] - S.cx;
      yy = Y[i] - S.cy;
      zz = Z[i] - S.cz;
      mass = M[i];
    }
    // src/SphericalBasis.cc:486-494
    const double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
    const bool inwin = valid && r >= S.rmin && r <= S.rmax;
    const double costh = zz / r;
    double cphi, sphi;
    phi_trig(xx, yy, cphi, sphi);
    const double xi = sph_r_to_xi(S, r / S.scale);
    const int idx = sph_cell(S, xi);
    // exputil/SLGridMP2.cc:894-895, :901-902
    const double x1 = (S.xi[idx + 1] - xi) / S.dxi;
    const double x2 = (xi - S.xi[idx]) / S.dxi;
    const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
    const double t0 = inwin ? mass * fac0 * P0 : 0.0;
    if (MLO == 0 && inwin) used++;

    {
      unsigned long long fp = (unsigned long long)S.fact;   // see k_sph_force: blocks LICM
      asm volatile("" : "+s"(fp));
      fact = (cdp)fp;
    }
    unsigned long long remaining = __ballot(inwin);
    while (remaining) {
      const int lead = __ffsll((long long)remaining) - 1;
      const int c = __shfl(idx, lead);
      const bool sel = inwin && idx == c;
      if (c != cur) {
        if (cur >= 0)
          wave_flush<NV>(acc, scratch, W + (size_t)cur * S.nrows * 2,
                         [](int j) { return acc_to_wrow<LMAX, MLO, MHI>(j); });
        cur = c;
      }
      const double a1 = sel ? t0 * x1 : 0.0;
      const double a2 = sel ? t0 * x2 : 0.0;

      // Legendre (src/Basis.cc:14-52) and trig (src/Basis.cc:95-112) recurrences, m-major
      const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
      double pmm = 1.0;
      double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;   // c[m], s[m], c[m-1], s[m-1]
      static_for<0, MHI + 1>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m == 1) {
          pmm *= -1.0 * somx2;
          cm1 = 1.0; sm1 = 0.0;
          cm = cphi; sm = sphi;
        } else if constexpr (m > 1) {
          pmm *= -(2.0 * m - 1.0) * somx2;
          const double cn = 2.0 * cphi * cm - cm1;
          const double sn = 2.0 * cphi * sm - sm1;
          cm1 = cm; sm1 = sm;
          cm = cn; sm = sn;
        }
        if constexpr (m >= MLO) {
          if (m == 0 || !S.M0_only) {
            double pl2 = pmm, pl1 = 0.0;
            static_for<m, LMAX + 1>([&](auto lc) {
              constexpr int l = decltype(lc)::value;
              double plm;
              if constexpr (l == m) plm = pmm;
              else if constexpr (l == m + 1) { plm = costh * (2 * m + 1) * pl2; pl1 = plm; }
              else {
                plm = (costh * (2 * l - 1) * pl1 - (l + m - 1) * pl2) * (1.0 / (l - m));
                pl2 = pl1;
                pl1 = plm;
              }
              const double Yl = fact[l * (LMAX + 1) + m] * plm;
              if constexpr (m == 0) {
                constexpr int k = acc_base(LMAX, MLO, 0) + (l - m);
                acc[2 * k] = fma(a1, Yl, acc[2 * k]);
                acc[2 * k + 1] = fma(a2, Yl, acc[2 * k + 1]);
              } else {
                constexpr int k = acc_base(LMAX, MLO, m) + 2 * (l - m);
                const double Yc = Yl * cm, Ys = Yl * sm;
                acc[2 * k] = fma(a1, Yc, acc[2 * k]);
                acc[2 * k + 1] = fma(a2, Yc, acc[2 * k + 1]);
                acc[2 * k + 2] = fma(a1, Ys, acc[2 * k + 2]);
                acc[2 * k + 3] = fma(a2, Ys, acc[2 * k + 3]);
              }
            });
          }
        }
      });
      remaining &= ~__ballot(sel);
    }
  }
  if (cur >= 0)
    wave_flush<NV>(acc, scratch, W + (size_t)cur * S.nrows * 2,
                   [](int j) { return acc_to_wrow<LMAX, MLO, MHI>(j); });
  if (MLO == 0) {
    // wave-reduce the used counter
    for (int off = 32; off > 0; off >>= 1) used += __shfl_xor(used, off);
    if (lane == 0 && used) atomicAdd(used_out, used);
  }
}

// m-range splits of the rows (keep 2 moment accumulators per real row in registers).  The waves
// of a block take the SAME particle chunk and one split each, so the chunk is fetched from HBM
// once and re-read from L1/L2 by the other splits.
template <int LMAX> __host__ __device__ constexpr int acc_nsplit()
{
  return LMAX <= 4 ? 1 : LMAX <= 7 ? 2 : LMAX <= 10 ? 4 : 6;
}

template <int LMAX>
__global__ void __launch_bounds__(ACC_WAVES * 64)
k_sph_accumulate(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                 const double *__restrict__ Z, const double *__restrict__ M,
                 const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi,
                 double *__restrict__ W, unsigned long long *__restrict__ used_out)
{
  constexpr int NS = acc_nsplit<LMAX>();
  constexpr int CPB = (ACC_WAVES >= NS) ? ACC_WAVES / NS : 1;      // chunks per block
  __shared__ double scratch_all[ACC_WAVES][16 * FLUSH_STRIDE];
  const int wave = threadIdx.x >> 6;
  double *scratch = scratch_all[wave];
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const int split = (NS <= ACC_WAVES) ? wave % NS : (int)(blockIdx.y * ACC_WAVES + wave);
  const size_t chunk = (NS <= ACC_WAVES) ? (size_t)blockIdx.x * CPB + wave / NS : blockIdx.x;
  if (split >= NS) return;
  const size_t cbeg = beg + chunk * ACC_CHUNK;
  if (cbeg >= end) return;
  const size_t cend = (cbeg + ACC_CHUNK < end) ? cbeg + ACC_CHUNK : end;
#define RUN(LO, HI) sph_accumulate_wave<LMAX, LO, HI>(S, X, Y, Z, M, cbeg, cend, scratch, W, used_out)
  if constexpr (LMAX <= 4) {
    RUN(0, LMAX);
  } else if constexpr (LMAX <= 7) {
    if (split == 0) RUN(0, 1); else RUN(2, LMAX);
  } else if constexpr (LMAX <= 10) {
    if (split == 0) RUN(0, 1); else if (split == 1) RUN(2, 3); else if (split == 2) RUN(4, 6);
    else RUN(7, LMAX);
  } else {
    if (split == 0) RUN(0, 0); else if (split == 1) RUN(1, 1); else if (split == 2) RUN(2, 3);
    else if (split == 3) RUN(4, 5); else if (split == 4) RUN(6, 8); else RUN(9, LMAX);
  }
#undef RUN
}

// ---- force ------------------------------------------------------------------------------------------------

struct ForceOut { double potl, potr, pott, potp; };

// One particle's field sums.  UNIFORM: g0.. are wave-uniform row pointers (scalar loads);
// otherwise per-lane pointers (vector gathers).  FLAGS: honour NO_L0/NO_L1/EVEN_L/EVEN_M/M0_only
// including the reference's moffset behaviour under EVEN_M (rows are not advanced for skipped m).
template <int LMAX, bool FLAGS, bool UNIFORM, class PG, class PH>
__device__ __forceinline__ ForceOut
sph_field(const SphDev &S, cdp fact, double costh, double cphi, double sphi, PG g0, PG g1, PH h0,
          PH h1, PH h2, double a1, double a2, double b0, double b1, double b2, bool ioff,
          double rmax_over_r0, double inv_r0)
{
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  // src/Basis.cc:54-93
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  double xc = costh;
  if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
  const double dfac = 1.0 / (xc * xc - 1.0);

  double pmm = 1.0;
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  static_for<0, LMAX + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m == 1) {
      pmm *= -1.0 * somx2;
      cm = cphi; sm = sphi;
    } else if constexpr (m > 1) {
      pmm *= -(2.0 * m - 1.0) * somx2;
      const double cn = 2.0 * cphi * cm - cm1;
      const double sn = 2.0 * cphi * sm - sm1;
      cm1 = cm; sm1 = sm;
      cm = cn; sm = sn;
    }
    bool m_on = true;
    if constexpr (FLAGS) {
      if (S.EVEN_M && (m & 1)) m_on = false;
      if (S.M0_only && m != 0) m_on = false;
    }
    // (rmax/r0)^(l+1) for the exterior multipole continuation, built up with l
    double rfac = rmax_over_r0;
    static_for<0, m>([&](auto) { rfac *= rmax_over_r0; });

    double pl2 = pmm, pl1 = 0.0;
    static_for<m, LMAX + 1>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      double plm, dplm;
      if constexpr (l == m) {
        plm = pmm;
        dplm = dfac * xc * l * plm;
      } else if constexpr (l == m + 1) {
        plm = costh * (2 * m + 1) * pl2;
        dplm = dfac * (xc * l * plm - (l + m) * pl2);
        pl1 = plm;
      } else {
        plm = (costh * (2 * l - 1) * pl1 - (l + m - 1) * pl2) * (1.0 / (l - m));
        dplm = dfac * (xc * l * plm - (l + m) * pl1);
        pl2 = pl1;
        pl1 = plm;
      }
      bool on = m_on;
      if constexpr (FLAGS) {
        if (l == 0 && S.NO_L0) on = false;
        if (l == 1 && S.NO_L1) on = false;
        if (l > 0 && S.EVEN_L && (l & 1)) on = false;
      }
      if (on) {
        int rc = row_of(l, m, 0);
        if constexpr (FLAGS) {
          if (S.EVEN_M && m > 0) rc = l * l + (m - 1);
        }
        const double f = fact[l * (LMAX + 1) + m];
        const double facL = f * plm, facD = f * dplm;
        if constexpr (m == 0) {
          double p = a1 * g0[rc] + a2 * g1[rc];
          double dp = b0 * h0[rc] + b1 * h1[rc] + b2 * h2[rc];
          if (ioff) {
            p *= rfac;
            dp = -p * inv_r0 * (l + 1);
          }
          o.potl += facL * p;
          o.potr += facL * dp;
          if constexpr (l > 0) o.pott += facD * p;
        } else {
          double pc = a1 * g0[rc] + a2 * g1[rc];
          double ps = a1 * g0[rc + 1] + a2 * g1[rc + 1];
          double dpc = b0 * h0[rc] + b1 * h1[rc] + b2 * h2[rc];
          double dps = b0 * h0[rc + 1] + b1 * h1[rc + 1] + b2 * h2[rc + 1];
          if (ioff) {
            pc *= rfac;
            ps *= rfac;
            const double facdp = -inv_r0 * (l + 1);
            dpc = pc * facdp;
            dps = ps * facdp;
          }
          const double pcs = pc * cm + ps * sm;
          o.potl += facL * pcs;
          o.potr += facL * (dpc * cm + dps * sm);
          o.pott += facD * pcs;
          o.potp += facL * (-pc * sm + ps * cm) * m;
        }
      }
      rfac *= rmax_over_r0;
    });
    // bound the scheduling region: one m-block of rows at a time (otherwise the whole
    // unrolled (l,m) nest is one block and register pressure explodes)
    __builtin_amdgcn_sched_barrier(0);
  });
  return o;
}

template <int LMAX, bool FLAGS>
__global__ void __launch_bounds__(256)
k_sph_force(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
            const double *__restrict__ Z, const uint32_t *__restrict__ lev_off, int lev_lo,
            int lev_hi, const double *__restrict__ G, const double *__restrict__ H,
            double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ,
            double *__restrict__ POT, double *__restrict__ VX, double *__restrict__ VY,
            double *__restrict__ VZ, double dt_kick, int assign)
{
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  cdp fact = (cdp)S.fact;
  const int lane = threadIdx.x & 63;
  // One 64-particle chunk per wave and NO particle loop: with a loop, LICM hoists the
  // hundreds of fp64 literals of the unrolled (l,m) nest out of it and spills them.
  {
    const size_t base = beg + ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
    if (base >= end) return;
    const size_t i = base + lane;
    const bool valid = i < end;
    double xx = 0, yy = 0, zz = 1;
    if (valid) {
      xx = X[i] - S.cx;
      yy = Y[i] - S.cy;
      zz = Z[i] - S.cz;
    }
    // src/SphericalBasis.cc:1545-1560
    double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
    const double costh = zz / r;
    double cphi, sphi;
    phi_trig(xx, yy, cphi, sphi);
    bool ioff = false;
    double r0 = r;
    if (r > S.rmax) {
      ioff = true;
      r = S.rmax;
    }
    const double rs = r / S.scale;
    const double xi = sph_r_to_xi(S, rs);
    int idx = sph_cell(S, xi);
    // get_pot weights (exputil/SLGridMP2.cc:894-902)
    const double x1 = (S.xi[idx + 1] - xi) / S.dxi;
    const double x2 = (xi - S.xi[idx]) / S.dxi;
    const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
    const double a1 = P0 * x1, a2 = P0 * x2;
    // get_force weights (exputil/SLGridMP2.cc:971-985)
    int jdx = idx < 1 ? 1 : idx;
    const double pf = (xi - S.xi[jdx]) / S.dxi;
    const double ffac = sph_d_xi_to_r(S, xi) / S.dxi;
    const double b0 = ffac * (pf - 0.5), b1 = ffac * (-2.0 * pf), b2 = ffac * (pf + 0.5);

    const int idx_u = __builtin_amdgcn_readfirstlane(idx);
    if (!valid) idx = idx_u;
    const bool uniform = __all(idx == idx_u);
    ForceOut o;
    const double rr = S.rmax / r0, ir0 = 1.0 / r0;
    if (uniform) {
      const int j_u = idx_u < 1 ? 1 : idx_u;
      cdp g0 = (cdp)(G + (size_t)idx_u * S.nrows), g1 = (cdp)(G + (size_t)(idx_u + 1) * S.nrows);
      cdp h0 = (cdp)(H + (size_t)(j_u - 1) * S.nrows), h1 = (cdp)(H + (size_t)j_u * S.nrows),
          h2 = (cdp)(H + (size_t)(j_u + 1) * S.nrows);
      o = sph_field<LMAX, FLAGS, true>(S, fact, costh, cphi, sphi, g0, g1, h0, h1, h2, a1, a2, b0,
                                       b1, b2, ioff, rr, ir0);
    } else {
      const double *g0 = G + (size_t)idx * S.nrows, *g1 = g0 + S.nrows;
      const double *h1 = H + (size_t)jdx * S.nrows, *h0 = h1 - S.nrows, *h2 = h1 + S.nrows;
      o = sph_field<LMAX, FLAGS, false>(S, fact, costh, cphi, sphi, g0, g1, h0, h1, h2, a1, a2, b0,
                                        b1, b2, ioff, rr, ir0);
    }
    if (!valid) return;

    // src/SphericalBasis.cc:1636-1652 (r is the clamped radius, as in the reference)
    const double fac = xx * xx + yy * yy;
    const double potr = o.potr / (S.scale * S.scale);
    const double potl = o.potl / S.scale;
    const double pott = o.pott / S.scale;
    const double potp = o.potp / S.scale;
    const double r3 = r * r * r;
    double ax = -(potr * xx / r - pott * xx * zz / r3);
    double ay = -(potr * yy / r - pott * yy * zz / r3);
    double az = -(potr * zz / r + pott * fac / r3);
    if (fac > DSMALL) {
      ax += potp * yy / fac;
      ay += -potp * xx / fac;
    }
    double pt = potl;
    if (!assign) {
      ax += AX[i];
      ay += AY[i];
      az += AZ[i];
      pt += POT[i];
    }
    AX[i] = ax;
    AY[i] = ay;
    AZ[i] = az;
    POT[i] = pt;
    if (dt_kick != 0.0) {   // fused second half-kick (src/incvel.cc:15-88), mul then add
      VX[i] = mul_then_add(VX[i], ax, dt_kick);
      VY[i] = mul_then_add(VY[i], ay, dt_kick);
      VZ[i] = mul_then_add(VZ[i], az, dt_kick);
    }
  }
}


// ---- per-LMAX launchers (one translation unit per LMAX: sph_inst.hip -DSPH_L=k) -----------------

struct SphAccArgs {
  SphDev S;
  const double *X, *Y, *Z, *M;
  const uint32_t *lev_off;
  int lo, hi;
  double *W;
  unsigned long long *used;
  size_t n;
  hipStream_t stream;
};

struct SphForceArgs {
  SphDev S;
  const double *X, *Y, *Z;
  const uint32_t *lev_off;
  int lo, hi;
  const double *G, *H;
  double *AX, *AY, *AZ, *POT, *VX, *VY, *VZ;
  double dt_kick;
  int assign;
  size_t n;
  unsigned grid;
  hipStream_t stream;
};

typedef void (*sph_acc_launcher)(const SphAccArgs &);
typedef void (*sph_force_launcher)(const SphForceArgs &);
