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
#include "thin_adv.h"

#include <type_traits>
#include <utility>

#define DSMALL 1.0e-16                 // src/expand.H:130
#define MINEPS (3.0 * 2.220446049250313e-16)   // src/Basis.cc:7

typedef const __attribute__((address_space(4))) double *cdp;   // constant (scalar-loadable)

#define SPH_MAX_L 12            // orders with unrolled kernels (sph_inst.hip); above: the run-time loops of sph_gen.hip
#define SPH_GEN_MAX_L 64        // ... up to this order

struct SphDev {
  int lmax, nmax, numr, cmap, nrows;
  int trows;             // rows per cell of the projected force table T4 (t4_rows below)
  double rmap, scale, rmin, rmax, xmin, dxi;
  double inv_rmap;
  double inv_dxi, inv_scale;   // reciprocals for the interpolation WEIGHTS (the cell index keeps the
                               // reference's exact division so that cell assignment is identical)
  double cx, cy, cz;
  int NO_L0, NO_L1, EVEN_L, EVEN_M, M0_only;
  int M0_acc;            // accumulation skips m > 0 (the n-body M0_only, src/SphericalBasis.cc:550; pyEXP's accumulate keeps every m)
  int xi_uniform;        // 1: xi[i] == xmin + dxi*i bit for bit (checked at create): no table gather
  int no_exterior;       // 1: no r>rmax multipole continuation (pyEXP computeAccel semantics)
  double detC;           // deterministic mode: 1.5 * 2^(52+e), every contribution is rounded to the grid 2^e
                         // before it is added (acc_add below); 0: off
  double umass;          // != 0: every particle of the component has this mass (the mass stream is not read)
  double fac0;           // -4 pi x Component::Adiabatic() of the basis' component at the time of the accumulation
                         // (src/SphericalBasis.cc:433, :441, :471; exp_amd_force_set_mass_scale): what a mass is multiplied by
  // Component::freeze (src/Component.cc:4194-4202) of the component whose particles the launch walks -- the source in an
  // accumulation, the TARGET in a force pass: beyond rtrunc of com0 + center a particle neither contributes nor is
  // accelerated (src/SphericalBasis.cc:468, :1159, :1521).  frz: {com0[3], center[3], rtrunc^2} in device memory (the
  // component's, exp_amd_comp::d_frz), or nullptr: rtrunc not set (the default, 1e20).  Behind a pointer because this struct
  // is a kernel argument held in scalar registers: seven more doubles there cost the accumulation 11 % in spills with the
  // option OFF (profiles/r05_freeze_ab.txt)
  const double *frz;
  double dsmall;         // added to r (src/expand.H:130: 1e-16; pyEXP: 1e-20 accumulating, 1e-18 evaluating)
  uint32_t key_add;      // added to every sort key produced (second half of a split store: +ncell)
  PseudoDev ps;          // frame acceleration of the TARGET component (force pass only)
  const double *xi;      // [numr]
  const double *p0;      // [numr]
  const double *E;       // [numr][lmax+1][nmax]
  const double *lc;      // [(lmax+1)*(lmax+1)][4] normalised-Legendre recurrence constants (below)
  const double *gen_ac;  // [(lmax+1)*(lmax+1)][2]: a(l,m), c(l,m) of the RESCALED recurrence (lc_a, lc_c) as run-time
  const double *gen_e;   // [lmax+1]: e_m (lc_E) -- data for the any-order kernels of sph_gen.hip
  const unsigned char *gen_slot;   // [trows][2]: l, m | (sine row ? 0x80 : 0) of every slot of the projected table
  // Far extrapolation beyond the table (possible with the logarithmic map only: hundreds of cells inside rmin, or outside
  // rmax in the pyEXP mode): the three-term radial derivative is then a small difference of large numbers and only the
  // reference's own operation order reproduces its value (sph_dp_lit below).  lit_lo / lit_hi: the force offset pf
  // below / above which a lane takes that evaluation (-/+ huge: never); lit_ef[edge][k][l][n]: the RAW eigenfunctions
  // at the three nodes of the first (edge 0) and of the last (edge 1) force stencil; lit_ev[l][n]; lit_coef: the
  // coefficient set the projected table was made from; lit_rowmap / lit_tscale: row and Legendre scale of a table slot.
  double lit_lo, lit_hi;
  double lit_xlo, lit_xhi;     // the same bounds in xi (the fast passes hand such lanes to the general pass)
  const double *lit_ef, *lit_ev, *lit_coef, *lit_tscale;
  const int *lit_rowmap;
  // the evaluation passes do not carry that code (its calls would put scratch under every wave they launch): they
  // append such a particle's slot to lit_list (lit_list[0] = count, entries from 1; lit_cap of them) and leave it to
  // the literal pass (k_sph_force<LMAX, 3>) that follows
  uint32_t *lit_list;
  uint32_t lit_cap;
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
// offset (in rows) of (l, m) in the m-major traversal order used by the projected force table
__host__ __device__ constexpr int mmajor_row(int L, int l, int m)
{
  return acc_base(L, 0, m) + (m == 0 ? (l - m) : 2 * (l - m));
}

// Projected force table T4: the same m-major order, plus one zero pad row after the m = 0 rows when
// their count (L+1) is odd, so that every (cos, sin) block and the cell stride are 64-byte aligned
// (a scalar 64-byte load that straddles two cache lines costs two scalar-cache requests).
__host__ __device__ constexpr int t4_rows(int L) { return (L + 1) * (L + 1) + ((L + 1) & 1); }
__host__ __device__ constexpr int t4_row(int L, int l, int m)
{
  return mmajor_row(L, l, m) + ((m > 0) ? ((L + 1) & 1) : 0);
}

// Normalised associated Legendre functions Pt(l,m) = factorial(l,m) * P_l^m(x) (the product the
// reference forms as facL, src/SphericalBasis.cc:521, :1586) by the same upward recurrences as
// Basis::legendre_R (src/Basis.cc:14-52) with the normalisation folded into the constants:
//   Pt(m,m)   = e_m * sqrt(1-x^2) * Pt(m-1,m-1)
//   Pt(l,m)   = A(l,m) * x * Pt(l-1,m) - B(l,m) * Pt(l-2,m)
//   dPt(l,m)  = (l * x * Pt(l,m) - C(l,m) * Pt(l-1,m)) / (x^2 - 1)          (src/Basis.cc:86-92)
// lc[(l*(L+1)+m)*4 + {0,1,2,3}] = {A, B, C, e (only for l == m)}
// The constants are pure functions of (l, m): they are evaluated at COMPILE time and become
// literals (s_mov pairs on the scalar ALU) instead of scalar loads that the FMAs would wait for.
//   A = sqrt((4l^2-1)/(l^2-m^2)),  B = sqrt((2l+1)(l-m-1)(l+m-1)/((2l-3)(l+m)(l-m))),
//   C = sqrt((2l+1)(l^2-m^2)/(2l-1)),  e_m = -sqrt((2m+1) k/(2m)) (k = 2 for m = 1, else 1),
//   e_0 = factorial(0,0) = sqrt(1/4pi)           [factorial(l,m): src/SphericalBasis.cc:328-335]
constexpr double lc_sqrt(double x)
{
  if (x <= 0.0) return 0.0;
  double r = x > 1.0 ? x : 1.0;
  for (int i = 0; i < 200; i++) {
    const double n = 0.5 * (r + x / r);
    if (n == r) break;
    r = n;
  }
  return r;
}
constexpr double lc_A(int l, int m) { return lc_sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m)); }
constexpr double lc_B(int l, int m)
{
  return lc_sqrt((2.0 * l + 1.0) * (l - m - 1.0) * (l + m - 1.0) / ((2.0 * l - 3.0) * (l + m) * (l - m)));
}
constexpr double lc_C(int l, int m) { return lc_sqrt((2.0 * l + 1.0) * ((double)l * l - (double)m * m) / (2.0 * l - 1.0)); }
constexpr double lc_E(int m)
{
  return m == 0 ? lc_sqrt(1.0 / (4.0 * 3.14159265358979323846))
                : -lc_sqrt((2.0 * m + 1.0) * (m == 1 ? 2.0 : 1.0) / (2.0 * m));
}
// Force evaluation runs the l-recurrence on RESCALED functions Ph(l,m) = s(l,m) Pt(l,m) with
// s(m,m) = s(m+1,m) = 1, s(l,m) = s(l-2,m) / B(l,m), which turns the three-term recurrence into
//   Ph(l,m) = a(l,m) * (x Ph(l-1,m)) - Ph(l-2,m),            a = s(l) A(l) / s(l-1)
//   (x^2-1) dPh(l,m) = l * (x Ph(l,m)) - c(l,m) * Ph(l-1,m),  c = C(l) s(l) / s(l-1)
// (one multiply fewer per term, and x*Ph(l,m) is shared by both lines); the factor 1/s(l,m) is
// folded into the rows of the projected force table when it is built (k_sph_project4).
constexpr double lc_s(int l, int m) { return (l <= m + 1) ? 1.0 : lc_s(l - 2, m) / lc_B(l, m); }
constexpr double lc_a(int l, int m) { return lc_s(l, m) * lc_A(l, m) / lc_s(l - 1, m); }
constexpr double lc_c(int l, int m) { return lc_C(l, m) * lc_s(l, m) / lc_s(l - 1, m); }
#define LC_a(l, m) (lc_const<lc_kind_a, (l), (m)>())
#define LC_c(l, m) (lc_const<lc_kind_c, (l), (m)>())
#define LC_A(l, m) (lc_const<lc_kind_A, (l), (m)>())
#define LC_B(l, m) (lc_const<lc_kind_B, (l), (m)>())
#define LC_C(l, m) (lc_const<lc_kind_C, (l), (m)>())
#define LC_E(m)    (lc_const<lc_kind_E, (m), (m)>())
enum { lc_kind_A, lc_kind_B, lc_kind_C, lc_kind_E, lc_kind_a, lc_kind_c };
template <int KIND, int L_, int M_>
__host__ __device__ constexpr double lc_const()
{
  constexpr double v = KIND == lc_kind_A ? lc_A(L_, M_)
                     : KIND == lc_kind_B ? lc_B(L_, M_)
                     : KIND == lc_kind_C ? lc_C(L_, M_)
                     : KIND == lc_kind_a ? lc_a(L_, M_)
                     : KIND == lc_kind_c ? lc_c(L_, M_)
                                         : lc_E(M_);
  return v;
}

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

// the same maps with the constant divisions replaced by multiplications (fast force pass)
__device__ __forceinline__ double sph_r_to_xi_rcp(const SphDev &S, double r)
{
  if (S.cmap == 1) { const double u = r * S.inv_rmap; return div_fast(u - 1.0, u + 1.0); }
  if (S.cmap == 2) return log(r);
  return r;
}
__device__ __forceinline__ double sph_d_xi_to_r_rcp(const SphDev &S, double xi)
{
  if (S.cmap == 1) return 0.5 * (1.0 - xi) * (1.0 - xi) * S.inv_rmap;
  if (S.cmap == 2) return exp(-xi);
  return 1.0;
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

// radial cell used as the sort key (r clamped to rmax like the force path, src/SphericalBasis.cc:1555-1560)
__device__ __forceinline__ uint32_t sph_key_cell(const SphDev &S, double x, double y, double z)
{
  const double xx = x - S.cx, yy = y - S.cy, zz = z - S.cz;
  double r = sqrt(xx * xx + yy * yy + zz * zz) + DSMALL;
  if (r > S.rmax && !S.no_exterior) r = S.rmax;
  return (uint32_t)sph_cell(S, sph_r_to_xi(S, r / S.scale));
}

// The same cell by multiplications with the stored reciprocals (one division left, in the map).
// A key only decides where a particle is PLACED; accumulation and force recompute the cell
// from the position, so a last-ulp difference at a cell edge costs nothing but a mixed wave.
__device__ __forceinline__ uint32_t sph_key_cell_rcp(const SphDev &S, double x, double y, double z)
{
  const double xx = x - S.cx, yy = y - S.cy, zz = z - S.cz;
  double r, ir_;
  sqrt_rsqrt(xx * xx + yy * yy + zz * zz, r, ir_);
  r += DSMALL;
  if (r > S.rmax && !S.no_exterior) r = S.rmax;
  const double xi = sph_r_to_xi_rcp(S, r * S.inv_scale);
  int idx = (int)((xi - S.xmin) * S.inv_dxi);
  idx = idx < 0 ? 0 : idx > S.numr - 2 ? S.numr - 2 : idx;
  return (uint32_t)idx;
}

// cos(phi), sin(phi) for phi = atan2(y, x) without the transcendental round trip
__device__ __forceinline__ void phi_trig(double xx, double yy, double &c, double &s)
{
  const double R2 = xx * xx + yy * yy;
  if (!(R2 > 0.0)) { atan2_trig_zero(xx, yy, c, s); return; }     // exactly on the axis, or x^2 + y^2 underflowed
  const double iR = 1.0 / sqrt(R2);
  c = xx * iR;
  s = yy * iR;
}

// Component::freeze, in the reference's operation order: r2 = sum_k (pos[k] - com0[k] - center[k])^2 > rtrunc^2
__device__ __forceinline__ bool sph_frozen(const SphDev &S, double px, double py, double pz)
{
#ifdef EXPAMD_NO_FREEZE      // A/B build: what the option's code costs the passes when it is off (profiles/r05_freeze_ab.txt)
  return false;
#endif
#define SPH_FRZ_ON(S) ((S).frz != nullptr)
  if (!S.frz) return false;
  const double *F = S.frz;
  const double dx = (px - F[0]) - F[3], dy = (py - F[1]) - F[4], dz = (pz - F[2]) - F[5];
  double r2 = dx * dx;
  r2 = mul_then_add(r2, dy, dy);
  r2 = mul_then_add(r2, dz, dz);
  return r2 > F[6];
}

// sin^2(theta) below which the accumulation forms cos(theta), sin(theta) as the reference does (a lane-level branch)
#ifndef SPH_POLAR_ACC
#define SPH_POLAR_ACC 1.0e-4
#endif
// The radius, in units of the reference's r = sqrt(x^2 + y^2 + z^2) + DSMALL offset (1e-16; pyEXP 1e-18), below which
// a lane takes the general pass: the fast arithmetic forms sin(theta) = R/r and 1/(x*x - 1) = -r^2/R^2 from the true R, the
// reference from x = z/r with the OFFSET r -- a relative difference of 2 DSMALL / r in every m >= 1 and every theta-derivative
// term (1.5e-8 of the acceleration at r = 1.3e-8: a particle the block-multistep campaign placed there left on a
// different trajectory).  2e-12 at this radius (r = 1e-4 with the n-body offset); a handful of particles of a realistic set
// are inside it (rmin is 1e-3 of the scale).
#define SPH_TINY_R 1.0e12

// ---- accumulation ----------------------------------------------------------------------------------

// Rows (accumulators) a flush transposes through the wave's LDS scratch at a time: 16 (64 lanes = 16 accumulators x 4
// lanes, each adding 16 of the 64 per-lane values), 8 (x 8 lanes x 8 values) or 4 (x 16 x 4).  Fewer rows = less LDS
// (4 waves x FLUSH_ROWS x FLUSH_STRIDE doubles) for more rounds.  FLUSH_STRIDE: doubles per scratch row, conflict-free
// for the read pattern of a half-wave (rows 8 / 16 / 32 banks apart).
#ifndef FLUSH_ROWS
#define FLUSH_ROWS 16
#endif
#if FLUSH_ROWS == 16
#define FLUSH_STRIDE 68
#elif FLUSH_ROWS == 8
#define FLUSH_STRIDE 72
#elif FLUSH_ROWS == 4
#define FLUSH_STRIDE 80
#else
#error "FLUSH_ROWS: 16, 8 or 4"
#endif

// Reduce NV per-lane values over the wave and atomically add them to dst[map(j)].
// scratch: wave-private LDS, FLUSH_ROWS*FLUSH_STRIDE doubles.
template <int NV, class MapFn>
__device__ __forceinline__ void wave_flush(double (&v)[NV], double *scratch, double *dst, MapFn map)
{
  constexpr int R = FLUSH_ROWS, P = 64 / R;       // P lanes per accumulator, each adds R values
  const int lane = threadIdx.x & 63;
  const int kk = lane / P, q = lane % P;
  static_for<0, (NV + R - 1) / R>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    static_for<0, R>([&](auto jc) {
      constexpr int j = g * R + decltype(jc)::value;
      if constexpr (j < NV) scratch[decltype(jc)::value * FLUSH_STRIDE + lane] = v[j];
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
    if (g * R + kk < NV) {
#pragma unroll
      for (int e = 0; e < R; e++) s += scratch[kk * FLUSH_STRIDE + q + P * e];
    }
#pragma unroll
    for (int off = 1; off < P; off <<= 1) s += __shfl_xor(s, off);
    if (q == 0 && g * R + kk < NV && s != 0.0) unsafeAtomicAdd(dst + map(g * R + kk), s);
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

// Work split of an accumulation launch over several time-step levels: the blocks [bstart[j],
// bstart[j+1]) take level lo + j in chunks of chunk[j] particles (no chunk crosses a level, and
// every level gets a chunk size that suits its own population).  nlev = 1: the classic launch.
#define LEVCHUNK_MAX 17
struct LevChunks {
  int lo, nlev;
  unsigned bstart[LEVCHUNK_MAX + 1];
  int chunk[LEVCHUNK_MAX];
};

#define ACC_WAVES 4
// Particles per block/wave chunk (contiguous, so one or two cells per wave).  Every chunk ends with
// a flush (LDS transposes + ~60 fp64 atomics per wave) and every block pays its launch: at 1024
// those fixed costs were 30% of the kernel, so the launcher picks up to ACC_CHUNK_MAX when the
// component is large enough to still fill the GPU several times over.  SMALL components (a snapshot of 1e4-1e6
// particles through pyEXP, a thin multistep level) get chunks down to one 64-particle group: what they pay is one
// serial flush per cell change within a wave, and the GPU is not even full (2e4 particles: 550 us at 1024 per
// chunk, 67 us at 64; 1e6: 150 -> 87 us).
#define ACC_CHUNK_MIN 64
#ifndef ACC_CHUNK_MAX
#define ACC_CHUNK_MAX 16384   // (4096 until the end of round 5; the launcher still keeps >= ACC_BLOCKS_TARGET blocks, so only components
                              // above 1.3e7 particles see longer chunks: 1e8 / S10 2.99 -> 2.86 ms at 16384, 2.90 at 8192, 3.03 at 32768,
                              // 3.21 at 2048 -- interleaved pairs on one box)
#endif
#ifndef ACC_P0_LDS
#define ACC_P0_LDS 2048       // p0 table entries cached in LDS by the shared-input path (numr <= this)
#endif
#ifndef ACC_P0_FILL_MIN
#define ACC_P0_FILL_MIN 512    // ... for chunks longer than this
#endif

// LIST mode of the accumulation kernel: the level-change differencing of MANY movers (multistep_update,
// src/SphericalBasis.cc:1156-1228).  The particles are taken through a list of mover slots (k_mover_list: in
// slot order, i.e. by (level, cell) where the store is cell-sorted) and blockIdx.z selects what a launch slice
// adds: z = 0 subtracts every mover from W[its level] (levels >= mfirst only), z = 1 + T adds the movers whose
// proposed level is T to W[T].  Within a slice the runs of equal (level, cell) accumulate in registers exactly as
// in the plain accumulation -- one flush per run instead of 4 (L+1)^2 same-address atomics per mover.
struct AccList {
  const uint32_t *list;
  const uint8_t *lev, *newlev;
  int mfirst, numr1;            // numr1 = numr - 1: cells per level of W
  int per_level;                // 1: one adding slice per proposed level (z = 1 + T); 0: ONE adding slice (z = 1) whose
                                // runs are the runs of equal (proposed level, cell) -- fewer passes, more flushes
};
// one list entry: position, signed mass and the level offset of its W cell (ca < 0: not in this slice)
__device__ __forceinline__ void acc_list_fetch(const AccList &al, const double *__restrict__ X,
                                               const double *__restrict__ Y, const double *__restrict__ Z,
                                               const double *__restrict__ M, double umass, size_t ip,
                                               double &x, double &y, double &z, double &m, int &ca)
{
  const uint32_t j = al.list[ip];
  const int fr = al.lev[j], to = al.newlev[j];
  const int slice = blockIdx.z;
  int lv;
  if (slice == 0) lv = fr >= al.mfirst ? fr : -1;
  else lv = (!al.per_level || to == slice - 1) ? to : -1;
  ca = lv < 0 ? -1 : lv * al.numr1;
  x = X[j]; y = Y[j]; z = Z[j];
  const double mm = umass != 0.0 ? umass : M[j];
  m = slice == 0 ? -mm : mm;
}

// Per-particle inputs of the moment accumulation: everything that does not depend on (l, m).
struct AccIn {
  double costh, sinth, cphi, sphi, a1, a2;   // a1 = -4pi m P0 x1, a2 = -4pi m P0 x2 (0 outside the window)
  int idx;                                   // radial cell, -1 when the particle does not contribute
};

// One reciprocal each of r and R = sqrt(x^2+y^2) gives cos(theta) = z/r, sin(theta) = R/r and
// cos/sin(phi); the constant divisors (scale, rmap) are multiplied by their reciprocals.  Within
// 1e-6 rad of the polar axis sin(theta) falls back to the reference's sqrt((1-x)(1+x)), whose
// cancellation error exceeds the parity tolerance there.
// p0l: the block's LDS copy of the background-potential table, or null (then S.p0 is gathered
// from global memory).  The pointer keeps its address space so that the reads are ds_read.
typedef const __attribute__((address_space(3))) double *ldp;
// cell_add: level * (numr-1) when several levels are accumulated in one launch (the moment buffer is
// then W[level][cell][row][2] and the wave's "current cell" the combined index), else 0.
// UPD: the window of the differencing, r < rmax (src/SphericalBasis.cc:1183), instead of rmin <= r <= rmax.
// MAYFRZ = false: an instantiation without the Component::freeze test (the launcher picks it when rtrunc is not set: even
// behind a scalar branch the test's code costs the dense accumulation 3 %, profiles/r05_freeze_ab.txt)
template <bool UPD = false, bool MAYFRZ = true>
__device__ __forceinline__ AccIn sph_acc_input(const SphDev &S, ldp p0l, double px,
                                               double py, double pz, double mass, bool valid,
                                               int cell_add = 0)
{
  AccIn in;
  double xx = 1, yy = 0, zz = 0;
  valid = valid && px != __builtin_inf();      // (an empty slot of an appended store, particles.h: APP_EMPTY)
  if (valid) { xx = px - S.cx; yy = py - S.cy; zz = pz - S.cz; }
  // src/SphericalBasis.cc:486-494
  const double R2 = xx * xx + yy * yy;
  double g, y;
  sqrt_rsqrt(R2 + zz * zz, g, y);
  const double r = g + S.dsmall;
  bool inwin = UPD ? (valid && r < S.rmax) : (valid && r >= S.rmin && r <= S.rmax);
  if constexpr (MAYFRZ) { if (SPH_FRZ_ON(S)) inwin = inwin && !sph_frozen(S, px, py, pz); }
  const double ir = rcp_refine(r, y);
  in.costh = zz * ir;
  if (R2 > SPH_POLAR_ACC * (r * r) && !(r < SPH_TINY_R * S.dsmall)) {
    double R, iR;
    sqrt_rsqrt(R2, R, iR);
    in.cphi = xx * iR;
    in.sphi = yy * iR;
    in.sinth = R * ir;
  } else {
    // near the axis the reference's sin(theta) = sqrt((1 - x)(1 + x)) from the rounded x = z/r carries a relative error
    // of ~1e-14 / theta^2 into the m >= 1 functions (see SPH_POLAR_FAC): its own operations, r^2 formed without fused
    // multiply-adds (src/SphericalBasis.cc:486-490, src/Basis.cc:22)
    in.costh = zz / (sqrt(sq_add_lit(sq_sum2_lit(xx, yy), zz)) + S.dsmall);
    phi_trig(xx, yy, in.cphi, in.sphi);
    in.sinth = sqrt((1.0 - in.costh) * (1.0 + in.costh));
  }
  const double xi = sph_r_to_xi_rcp(S, r * S.inv_scale);
  const int idx = sph_cell(S, xi);
  // exputil/SLGridMP2.cc:894-895, :901-902
  // the xi grid is uniform by construction (exputil/SLGridMP2.cc:1355-1382): recompute its nodes
  // rather than gather them (a dependent global load in the middle of this latency-bound chain)
  const double xlo = S.xi_uniform ? mul_then_add(S.xmin, S.dxi, (double)idx) : S.xi[idx];
  const double xhi = S.xi_uniform ? mul_then_add(S.xmin, S.dxi, (double)(idx + 1)) : S.xi[idx + 1];
  const double x1 = (xhi - xi) * S.inv_dxi;
  const double x2 = (xi - xlo) * S.inv_dxi;
  double pa, pb;
  if (p0l) { pa = p0l[idx]; pb = p0l[idx + 1]; }
  else { pa = S.p0[idx]; pb = S.p0[idx + 1]; }
  const double P0 = x1 * pa + x2 * pb;
  const double t0 = inwin ? mass * S.fac0 * P0 : 0.0;
  in.a1 = t0 * x1;
  in.a2 = t0 * x2;
  in.idx = inwin ? idx + cell_add : -1;
  return in;
}

// Deterministic (order-independent) accumulation: with DET every contribution w*p is first rounded
// to a fixed absolute grid 2^e -- (w*p + C) - C with C = 1.5 * 2^(52+e), one FMA and one subtraction,
// both exact after the FMA's rounding -- chosen by the host so that every partial sum of the launch
// stays below 2^53 * 2^e.  All later additions (registers, the LDS-transposed wave reduction, the fp64
// atomics on W) are then EXACT, hence associative: the sums no longer depend on the order the
// particles happen to sit in, and a run is bit-reproducible.  Costs two more VALU ops per term.
template <bool DET>
__device__ __forceinline__ void acc_add(double &a, double w, double p, double C)
{
  if constexpr (DET) {
    double t = fma(w, p, C);
    t -= C;
    a += t;
  } else {
    a = fma(w, p, a);
  }
}
// ... for a value about to be added by an atomic
__device__ __forceinline__ double det_round(double v, double C) { return C != 0.0 ? (v + C) - C : v; }

// Rows with m in [MLO, MHI] of one 64-particle group: ballot waterfall over the cells present,
// register accumulation, LDS-transposed flush when the wave's current cell changes.
template <int LMAX, int MLO, int MHI, int NV, bool DET>
__device__ __forceinline__ void
sph_acc_group(const SphDev &S, cdp &lc, const AccIn &in, double (&acc)[NV], int &cur,
              double *scratch, double *__restrict__ W)
{
  {
    unsigned long long fp = (unsigned long long)S.lc;   // re-derive per group: blocks LICM of the
    asm volatile("" : "+s"(fp));                        // constant loads (SGPR spills)
    lc = (cdp)fp;
  }
  const bool inwin = in.idx >= 0;
  unsigned long long remaining = __ballot(inwin);
  while (remaining) {
    const int lead = __ffsll((long long)remaining) - 1;
    const int c = __builtin_amdgcn_readlane(in.idx, lead);   // (scalar: the flush test below is a scalar branch)
    const bool sel = inwin && in.idx == c;
    if (c != cur) {
      if (cur >= 0)
        wave_flush<NV>(acc, scratch, W + (size_t)cur * S.nrows * 2,
                       [](int j) { return acc_to_wrow<LMAX, MLO, MHI>(j); });
      cur = c;
    }
    const unsigned long long selm = __ballot(sel);
    // (skipping these selects for single-cell groups behind a scalar branch measured 1 % slower)
    const double a1 = sel ? in.a1 : 0.0;
    const double a2 = sel ? in.a2 : 0.0;
    const double costh = in.costh;
    const double somx2 = in.sinth;
    double pmm = LC_E(0);                              // Pt(0,0) = factorial(0,0)
    double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;   // c[m], s[m], c[m-1], s[m-1]
    static_for<0, MHI + 1>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m == 1) {
        pmm *= LC_E(1) * somx2;
        cm1 = 1.0; sm1 = 0.0;
        cm = in.cphi; sm = in.sphi;
      } else if constexpr (m > 1) {
        pmm *= LC_E(m) * somx2;
        const double cn = 2.0 * in.cphi * cm - cm1;       // src/Basis.cc:107-110
        const double sn = 2.0 * in.cphi * sm - sm1;
        cm1 = cm; sm1 = sm;
        cm = cn; sm = sn;
      }
      if constexpr (m >= MLO) {
        if (m == 0 || !S.M0_acc) {
          // per-m weights: the four (x1|x2) x (cos|sin) moments share Pt(l,m)
          const double a1c = a1 * cm, a2c = a2 * cm, a1s = a1 * sm, a2s = a2 * sm;
          // rescaled functions Ph = s(l,m) Pt (see lc_s): W holds s(l,m) x the moments and
          // k_sph_contract divides the factor out
          double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
          static_for<m, LMAX + 1>([&](auto lc_) {
            constexpr int l = decltype(lc_)::value;
            double plm;
            if constexpr (l == m) plm = pmm;
            else if constexpr (l == m + 1) plm = LC_a(l, m) * tprev;
            else plm = fma(LC_a(l, m), tprev, -pl2);
            if constexpr (l < LMAX) tprev = costh * plm;
            pl2 = pl1;
            pl1 = plm;
            if constexpr (m == 0) {
              constexpr int k = acc_base(LMAX, MLO, 0) + (l - m);
              acc_add<DET>(acc[2 * k], a1, plm, S.detC);
              acc_add<DET>(acc[2 * k + 1], a2, plm, S.detC);
            } else {
              constexpr int k = acc_base(LMAX, MLO, m) + 2 * (l - m);
              acc_add<DET>(acc[2 * k], a1c, plm, S.detC);
              acc_add<DET>(acc[2 * k + 1], a2c, plm, S.detC);
              acc_add<DET>(acc[2 * k + 2], a1s, plm, S.detC);
              acc_add<DET>(acc[2 * k + 3], a2s, plm, S.detC);
            }
          });
        }
      }
    });
    remaining &= ~selm;
  }
}

// One wave accumulates the rows with m in [MLO, MHI] over the particle chunk [cbeg, cend), computing
// the per-particle inputs itself.
template <int LMAX, int MLO, int MHI, bool DET, bool LIST, bool MAYFRZ = true>
__device__ __forceinline__ void
sph_accumulate_wave(const SphDev &S, const double *__restrict__ X, const double *__restrict__ Y,
                    const double *__restrict__ Z, const double *__restrict__ M,
                    int cell_add, size_t cbeg,
                    size_t cend, double *scratch, double *__restrict__ W,
                    unsigned long long *__restrict__ used_out, const AccList &al)
{
  constexpr int NACC = acc_base(LMAX, MLO, MHI + 1);
  constexpr int NV = 2 * NACC;
  const int lane = threadIdx.x & 63;
  cdp lc = (cdp)S.lc;
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) acc[j] = 0.0;
  int cur = -1;
  unsigned long long used = 0;

  // software prefetch: the loads of group k+1 are in flight while group k is reduced
  double nx = 0, ny = 0, nz = 0, nm = 0;
  int nca = cell_add;                       // LIST: per entry (< 0: not in this slice)
  const bool um = S.umass != 0.0;
  if (cbeg + lane < cend) {
    if constexpr (LIST) acc_list_fetch(al, X, Y, Z, M, S.umass, cbeg + lane, nx, ny, nz, nm, nca);
    else { nx = X[cbeg + lane]; ny = Y[cbeg + lane]; nz = Z[cbeg + lane]; nm = um ? S.umass : M[cbeg + lane]; }
  }
  for (size_t base = cbeg; base < cend; base += 64) {
    const size_t i = base + lane;
    const AccIn in = sph_acc_input<LIST, MAYFRZ>(S, (ldp) nullptr, nx, ny, nz, nm, LIST ? (i < cend && nca >= 0) : i < cend,
                                         LIST ? nca : cell_add);
    if (i + 64 < cend) {
      if constexpr (LIST) acc_list_fetch(al, X, Y, Z, M, S.umass, i + 64, nx, ny, nz, nm, nca);
      else { nx = X[i + 64]; ny = Y[i + 64]; nz = Z[i + 64]; nm = um ? S.umass : M[i + 64]; }
    }
    if (MLO == 0 && in.idx >= 0) used++;
    sph_acc_group<LMAX, MLO, MHI, NV, DET>(S, lc, in, acc, cur, scratch, W);
  }
  if (cur >= 0)
    wave_flush<NV>(acc, scratch, W + (size_t)cur * S.nrows * 2,
                   [](int j) { return acc_to_wrow<LMAX, MLO, MHI>(j); });
  if (MLO == 0) {
    for (int off = 32; off > 0; off >>= 1) used += __shfl_xor(used, off);
    if (lane == 0 && used) atomicAdd(used_out, used);
  }
}

// Same rows, but the four waves of the block work on ONE chunk: each wave computes the inputs of a
// quarter of every 256-particle tile once, shares them through LDS (double-buffered, one barrier
// per tile) and then reduces its own m-range over the whole tile.
struct AccShared {
  double v[2][6][ACC_WAVES * 64];
  int idx[2][ACC_WAVES * 64];
  unsigned long long used[ACC_WAVES];      // in-window counts of the four quarters, handed over at the last tile
};

template <int LMAX, int MLO, int MHI, bool DET, bool LIST, bool MAYFRZ = true>
__device__ __forceinline__ void
sph_accumulate_shared(const SphDev &S, ldp p0t, const double *__restrict__ X,
                      const double *__restrict__ Y, const double *__restrict__ Z,
                      const double *__restrict__ M, int cell_add, size_t cbeg,
                      size_t cend, double *scratch,
                      AccShared &sh, double *__restrict__ W, unsigned long long *__restrict__ used_out,
                      const AccList &al)
{
  constexpr int NACC = acc_base(LMAX, MLO, MHI + 1);
  constexpr int NV = 2 * NACC;
  constexpr int TILE = ACC_WAVES * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  cdp lc = (cdp)S.lc;
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) acc[j] = 0.0;
  int cur = -1;
  unsigned long long used = 0;

  size_t ip = cbeg + (size_t)wave * 64 + lane;
  double nx = 0, ny = 0, nz = 0, nm = 0;
  int nca = cell_add;                       // LIST: per entry (< 0: not in this slice)
  const bool um = S.umass != 0.0;
  if (ip < cend) {
    if constexpr (LIST) acc_list_fetch(al, X, Y, Z, M, S.umass, ip, nx, ny, nz, nm, nca);
    else { nx = X[ip]; ny = Y[ip]; nz = Z[ip]; nm = um ? S.umass : M[ip]; }
  }
  int par = 0;
#ifdef EXPT_TIMING
  unsigned long long t_load = 0, t_in = 0, t_bar = 0, t_red = 0, t_all0 = __builtin_readcyclecounter();
#endif
  for (size_t tbase = cbeg; tbase < cend; tbase += TILE, par ^= 1) {
#ifdef EXPT_TIMING
    const unsigned long long ta = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tb = __builtin_readcyclecounter();
    t_load += tb - ta;
#endif
    {
      const AccIn in = sph_acc_input<LIST, MAYFRZ>(S, p0t, nx, ny, nz, nm, LIST ? (ip < cend && nca >= 0) : ip < cend,
                                           LIST ? nca : cell_add);
      if (in.idx >= 0) used++;
      const int q = wave * 64 + lane;
      sh.v[par][0][q] = in.costh; sh.v[par][1][q] = in.cphi; sh.v[par][2][q] = in.sphi;
      sh.v[par][3][q] = in.a1;    sh.v[par][4][q] = in.a2;   sh.v[par][5][q] = in.sinth;
      sh.idx[par][q] = in.idx;
    }
    // the in-window count leaves the block as ONE atomic, issued here at the last tile's barrier (four
    // same-address atomics per block right before the waves end kept their registers waiting for the acks)
    const bool last_tile = tbase + TILE >= cend;
    if (last_tile) {
      for (int off = 32; off > 0; off >>= 1) used += __shfl_xor(used, off);
      if (lane == 0) sh.used[wave] = used;
    }
#ifdef EXPT_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tc = __builtin_readcyclecounter();
    t_in += tc - tb;
#endif
    lds_barrier();
#ifdef EXPT_TIMING
    const unsigned long long td = __builtin_readcyclecounter();
    t_bar += td - tc;
#endif
    if (last_tile && wave == 0 && lane == 0) {
      const unsigned long long tot = (sh.used[0] + sh.used[1]) + (sh.used[2] + sh.used[3]);
      // (A/B builds of round 4, tools/build_variant_l10.sh: EXPT_NO_USED drops the block's one same-address atomic,
      // EXPT_USED_SPREAD sends it to one of 128 slots -- both measured NEUTRAL on the 1e8 headline, DESIGN.md section 5:
      // 24 414 atomics in 2.9 ms are one every 120 ns, a tenth of what the atomic unit serves)
#if defined(EXPT_NO_USED)
      (void)tot;
#elif defined(EXPT_USED_SPREAD)
      if (tot) atomicAdd(used_out + 8 + 8 * (blockIdx.x & 127), tot);
#else
      if (tot) atomicAdd(used_out, tot);
#endif
    }
    // next tile's particles: requested AFTER the barrier (no load is outstanding at it) and in
    // flight while this tile is reduced
    ip += TILE;
    if (ip < cend) {
      if constexpr (LIST) acc_list_fetch(al, X, Y, Z, M, S.umass, ip, nx, ny, nz, nm, nca);
      else { nx = X[ip]; ny = Y[ip]; nz = Z[ip]; nm = um ? S.umass : M[ip]; }
    }
#pragma unroll 1
    for (int sub = 0; sub < ACC_WAVES; sub++) {
      if (tbase + (size_t)sub * 64 >= cend) break;
      const int q = sub * 64 + lane;
      AccIn in;
      in.costh = sh.v[par][0][q]; in.cphi = sh.v[par][1][q]; in.sphi = sh.v[par][2][q];
      in.a1 = sh.v[par][3][q];    in.a2 = sh.v[par][4][q];   in.sinth = sh.v[par][5][q];
      in.idx = sh.idx[par][q];
      sph_acc_group<LMAX, MLO, MHI, NV, DET>(S, lc, in, acc, cur, scratch, W);
    }
#ifdef EXPT_TIMING
    t_red += __builtin_readcyclecounter() - td;
#endif
  }
#ifdef EXPT_TIMING
  if (lane == 0) {
    extern __device__ unsigned long long g_dbg_s[16];
    const int o = MLO == 0 ? 0 : 8;         // the first m-range and the others apart
    atomicAdd(&g_dbg_s[o + 0], t_load); atomicAdd(&g_dbg_s[o + 1], t_in); atomicAdd(&g_dbg_s[o + 2], t_bar);
    atomicAdd(&g_dbg_s[o + 3], t_red); atomicAdd(&g_dbg_s[o + 4], __builtin_readcyclecounter() - t_all0);
    atomicAdd(&g_dbg_s[o + 5], 1ull); atomicAdd(&g_dbg_s[o + 6], (unsigned long long)((cend - cbeg + TILE - 1) / TILE));
  }
#endif
  if (cur >= 0)
    wave_flush<NV>(acc, scratch, W + (size_t)cur * S.nrows * 2,
                   [](int j) { return acc_to_wrow<LMAX, MLO, MHI>(j); });
}

// m-range splits of the rows (keep 2 moment accumulators per real row in registers).  The waves
// of a block take the SAME particle chunk and one split each, so the chunk is fetched from HBM
// once and re-read from L1/L2 by the other splits.
// Shared-input path (3 <= LMAX <= 10): the four waves of a block take ONE chunk, compute the
// per-particle inputs of a quarter of every 256-particle tile once, share them through LDS and
// reduce one m-range each: [0,b1), [b1,b2), [b2,b3), [b3,LMAX] (rows/costs roughly balanced).
template <int LMAX> __host__ __device__ constexpr bool acc_shared() { return LMAX >= 3 && LMAX <= 10; }
template <int LMAX> __host__ __device__ constexpr int acc_bound(int k)
{
#ifndef ACC_B10
#define ACC_B10 {2, 4, 6}
#endif
  constexpr int B[11][3] = {{1, 1, 1}, {1, 1, 1}, {1, 2, 2}, {1, 2, 3}, {1, 2, 3}, {1, 2, 3},
                            {1, 2, 4}, {1, 2, 4}, {2, 4, 6}, {2, 4, 6}, ACC_B10};
  return B[LMAX <= 10 ? LMAX : 10][k];
}
template <int LMAX> __host__ __device__ constexpr int acc_nsplit()
{
  return acc_shared<LMAX>() ? 4 : LMAX <= 4 ? 1 : LMAX <= 7 ? 2 : LMAX <= 10 ? 4 : 6;
}

template <int LMAX, bool DET, bool LIST = false, bool MAYFRZ = true>
__global__ void __launch_bounds__(ACC_WAVES * 64)
k_sph_accumulate(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                 const double *__restrict__ Z, const double *__restrict__ M,
                 const uint32_t *__restrict__ lev_off, LevChunks LC,
                 double *__restrict__ W, unsigned long long *__restrict__ used_out,
                 int multilevel /* W[level][cell][row][2] */, AccList al = AccList{})
{
  constexpr int NS = acc_nsplit<LMAX>();
  // which level this block works on, and with which chunk size (block-uniform: scalar loop)
  int lj = 0;
  while (lj + 1 < LC.nlev && blockIdx.x >= LC.bstart[lj + 1]) lj++;
  const int lev_lo = LC.lo + lj, lev_hi = lev_lo;
  const int ACC_CHUNK = LC.chunk[lj];
  const unsigned bx = blockIdx.x - LC.bstart[lj];
  const int cell_add = multilevel ? lev_lo * (S.numr - 1) : 0;
  constexpr int CPB = (ACC_WAVES >= NS) ? ACC_WAVES / NS : 1;      // chunks per block
  __shared__ double scratch_all[ACC_WAVES][FLUSH_ROWS * FLUSH_STRIDE];
  const int wave = threadIdx.x >> 6;
  double *scratch = scratch_all[wave];
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  if constexpr (acc_shared<LMAX>()) {
    // one chunk per block, shared per-particle inputs, balanced m-ranges (31/34/26/30 rows at L=10)
    __shared__ AccShared sh;
    // The per-particle input chain (sqrt, divisions, cell, P0 interpolation) is latency-bound and
    // all waves of the block run it at the same time: keep the p0 table in LDS so that no global
    // gather sits in the middle of it.
    __shared__ double p0s[ACC_P0_LDS];
    const size_t cbeg = beg + (size_t)bx * ACC_CHUNK;
    if (cbeg >= end) return;
    const size_t cend = (cbeg + ACC_CHUNK < end) ? cbeg + ACC_CHUNK : end;
    // (short chunks -- LIST launches, thin multistep levels, small components -- are not worth the fill: the table
    // is numr doubles, the chunk may be 64 particles)
    const bool p0_in_lds = !LIST && S.numr <= ACC_P0_LDS && ACC_CHUNK > ACC_P0_FILL_MIN;
    if (p0_in_lds) {
      for (int k = threadIdx.x; k < S.numr; k += ACC_WAVES * 64) p0s[k] = S.p0[k];
      __syncthreads();
    }
    ldp p0t = p0_in_lds ? (ldp)p0s : (ldp) nullptr;
#define RUNS(LO, HI) sph_accumulate_shared<LMAX, LO, HI, DET, LIST, MAYFRZ>(S, p0t, X, Y, Z, M, cell_add, cbeg, cend, scratch, sh, W, used_out, al)
    constexpr int b1 = acc_bound<LMAX>(0), b2 = acc_bound<LMAX>(1), b3 = acc_bound<LMAX>(2);
    // (two A/B experiments of round 4, profiles/r04_accumulate_split_ab.txt: other m-splits -- a 36-row wave costs the
    // second wave per SIMD, 4.5-6.3 ms --, and odd blocks taking the ranges in reverse order so that a SIMD's two waves
    // would be one light and one heavy range: neutral, 3.03 against 3.01 ms)
    const int role = wave;
    if (role == 0) RUNS(0, b1 - 1); else if (role == 1) RUNS(b1, b2 - 1);
    else if (role == 2) RUNS(b2, b3 - 1); else RUNS(b3, LMAX);
#undef RUNS
    return;
  } else {
  const int split = (NS <= ACC_WAVES) ? wave % NS : (int)(blockIdx.y * ACC_WAVES + wave);
  const size_t chunk = (NS <= ACC_WAVES) ? (size_t)bx * CPB + wave / NS : bx;
  if (split >= NS) return;
  const size_t cbeg = beg + chunk * ACC_CHUNK;
  if (cbeg >= end) return;
  const size_t cend = (cbeg + ACC_CHUNK < end) ? cbeg + ACC_CHUNK : end;
#define RUN(LO, HI) sph_accumulate_wave<LMAX, LO, HI, DET, LIST, MAYFRZ>(S, X, Y, Z, M, cell_add, cbeg, cend, scratch, W, used_out, al)
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
}

// ---- per-particle atomic moments: level-change differencing and sparse levels ----------------------------
// differencing (plain == 0): for every particle whose proposed level differs from its level, add its
// moment contribution to Wd[to] and subtract it from Wd[from] (src/SphericalBasis.cc:1156-1228;
// window r < rmax only).  plain != 0: every particle of the range adds its contribution to
// Wd[its level] -- the accumulation of SPARSE multistep levels, which are not cell-sorted (window
// rmin <= r <= rmax and the used count of determine_coefficients_thread, src/SphericalBasis.cc:
// 486-494).  Each contributing lane issues its own fp64 atomics; waves without one leave at once.
// Wd[level][cell][row][2].
// STAGED (few movers, through the list): an atomic INSTRUCTION costs a wave ~65 ns whatever its lane count
// (tools/dbg/atomic_chain.hip), so a wave of this kernel with a handful of movers spends 50 us on its
// 8 (L+1)^2 of them.  Staged, the lanes write their values to stage[mover][row][2] with plain stores and their
// two W offsets to keys[mover]; k_mstep_apply then adds them with one lane per VALUE (64 values an instruction).
template <int LMAX, bool STAGED = false>
__global__ void __launch_bounds__(256)
k_sph_mstep_update(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                   const double *__restrict__ Z, const double *__restrict__ M,
                   const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev,
                   const uint32_t *__restrict__ lev_off, int first, int last, int mfirst,
                   double *__restrict__ Wd, int plain, unsigned long long *__restrict__ used_out,
                   const uint32_t *__restrict__ list /* slots of the movers (k_mover_list; lev_off = {0, count}) or null */,
                   double *__restrict__ stage = nullptr, int2 *__restrict__ keys = nullptr)
{
  size_t i = 0;
  bool have = false;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;      // list entry, or offset in the slot range
  if (list) {
    if (g < lev_off[1]) { i = list[g]; have = true; }
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
  double xx = 0, yy = 0, zz = 1, mass = 0;
  if (mover) {
    xx = X[i] - S.cx;
    yy = Y[i] - S.cy;
    zz = Z[i] - S.cz;
    mass = M[i];
    if (SPH_FRZ_ON(S) && sph_frozen(S, X[i], Y[i], Z[i])) mover = false;     // (:468, :1159: before anything else)
  }
  const double r = sqrt(sq_add_lit(sq_sum2_lit(xx, yy), zz)) + DSMALL;       // (every product rounded on its own: sq_sum2_lit)
  if (plain) {
    if (!(r >= S.rmin && r <= S.rmax)) mover = false;
    const unsigned long long in = __ballot(mover);
    if ((threadIdx.x & 63) == 0 && in) atomicAdd(used_out, (unsigned long long)__popcll(in));
  } else if (!(r < S.rmax)) mover = false;
  if constexpr (STAGED) { if (have && !mover) keys[g] = make_int2(-1, -1); }      // (g: list entry, or offset in the range)
  if (!__any(mover)) return;
  const double costh = zz / r;
  double cphi, sphi;
  phi_trig(xx, yy, cphi, sphi);
  const double xi = sph_r_to_xi(S, r / S.scale);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
  const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const double t0 = mass * S.fac0 * P0;
  const double a1 = t0 * x1, a2 = t0 * x2;
  const size_t wl = (size_t)(S.numr - 1) * S.nrows * 2;
  double *wto = Wd + (size_t)to * wl + (size_t)idx * S.nrows * 2;
  double *wfr = Wd + (size_t)from * wl + (size_t)idx * S.nrows * 2;
  const bool sub = !plain && mover && from >= mfirst;       // levels below mfirst[mdrft] are not updated
  [[maybe_unused]] double *st = nullptr;
  if constexpr (STAGED) {
    if (mover) {
      keys[g] = make_int2((int)((size_t)to * wl + (size_t)idx * S.nrows * 2),
                          sub ? (int)((size_t)from * wl + (size_t)idx * S.nrows * 2) : -1);
      st = stage + g * (size_t)(S.nrows * 2);
    }
  }
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  double pmm = LC_E(0);
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  static_for<0, LMAX + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m == 1) {
      pmm *= LC_E(1) * somx2;
      cm = cphi; sm = sphi;
    } else if constexpr (m > 1) {
      pmm *= LC_E(m) * somx2;
      const double cn = 2.0 * cphi * cm - cm1;
      const double sn = 2.0 * cphi * sm - sm1;
      cm1 = cm; sm1 = sm;
      cm = cn; sm = sn;
    }
    double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
    static_for<m, LMAX + 1>([&](auto lc_) {
      constexpr int l = decltype(lc_)::value;
      double plm;                       // rescaled Ph(l,m), as in the accumulation (same W scaling)
      if constexpr (l == m) plm = pmm;
      else if constexpr (l == m + 1) plm = LC_a(l, m) * tprev;
      else plm = fma(LC_a(l, m), tprev, -pl2);
      tprev = costh * plm;
      pl2 = pl1;
      pl1 = plm;
      constexpr int row = row_of(l, m, 0);
      if constexpr (STAGED) {
        if (mover) {
          const double yc = (m == 0) ? plm : plm * cm;
          *reinterpret_cast<double2 *>(st + 2 * row) = make_double2(det_round(a1 * yc, S.detC), det_round(a2 * yc, S.detC));
          if constexpr (m > 0) {
            const double ys = plm * sm;
            *reinterpret_cast<double2 *>(st + 2 * row + 2) = make_double2(det_round(a1 * ys, S.detC), det_round(a2 * ys, S.detC));
          }
        }
      } else
      if (mover) {
        const double yc = (m == 0) ? plm : plm * cm;
        const double v1 = det_round(a1 * yc, S.detC), v2 = det_round(a2 * yc, S.detC);
        unsafeAtomicAdd(wto + 2 * row, v1);
        unsafeAtomicAdd(wto + 2 * row + 1, v2);
        if (sub) {
          unsafeAtomicAdd(wfr + 2 * row, -v1);
          unsafeAtomicAdd(wfr + 2 * row + 1, -v2);
        }
        if constexpr (m > 0) {
          const double ys = plm * sm;
          const double u1 = det_round(a1 * ys, S.detC), u2 = det_round(a2 * ys, S.detC);
          unsafeAtomicAdd(wto + 2 * row + 2, u1);
          unsafeAtomicAdd(wto + 2 * row + 3, u2);
          if (sub) {
            unsafeAtomicAdd(wfr + 2 * row + 2, -u1);
            unsafeAtomicAdd(wfr + 2 * row + 3, -u2);
          }
        }
      }
    });
  });
}

// second half of the staged differencing: one lane per (mover, value); keys[mover] = {offset of its cell in
// W[to], in W[from] or -1}, values as stored by k_sph_mstep_update<L, true>
template <int UNUSED = 0>            // (a template only for its linkage: this header is in every sph_inst unit)
__global__ void __launch_bounds__(256)
k_mstep_apply(const double *__restrict__ stage, const int2 *__restrict__ keys, const uint32_t *__restrict__ cnt,
              uint32_t nfixed /* entries when cnt is null */, int nval, double *__restrict__ Wd)
{
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t g = t / (size_t)nval;
  if (g >= (cnt ? cnt[1] : nfixed)) return;
  const int r = (int)(t - g * (size_t)nval);
  const int2 k = keys[g];
  if (k.x < 0) return;
  const double v = stage[t];
  if (v == 0.0) return;
  unsafeAtomicAdd(Wd + (size_t)k.x + r, v);
  if (k.y >= 0) unsafeAtomicAdd(Wd + (size_t)k.y + r, -v);
}

// ---- force ------------------------------------------------------------------------------------------------
//
// Projected table T4[cell][q][4] (q = m-major row order, see mmajor_row), built once per
// coefficient set by k_sph_project4 (sph.hip) from G[i][row] = sum_n E[i][l][n] c[row][n],
// H = p0 * G and j = max(cell, 1):
//     {G0, D, Bq, Aq} = {G[cell], G[cell+1]-G[cell], (H[j+1]-H[j-1])/2, H[j-1]-2H[j]+H[j+1]}
// so that for a particle in that cell (x2, pf = its get_pot / get_force offsets):
//     p_row  = P0   * (G0 + x2 * D)          == P0 (x1 G[cell] + x2 G[cell+1])
//     dp_row = ffac * (Bq + pf * Aq)         == ffac ((pf-.5) H[j-1] - 2 pf H[j] + (pf+.5) H[j+1])
// P0, ffac (and 1/(x^2-1) of the Legendre derivative) are per-particle factors applied once at
// the end.  2 FMAs per row and field instead of 5.

struct ForceOut { double potl, potr, pott, potp; };

// sin^2(theta) below which a lane takes the general evaluation.  The fast arithmetic forms sin(theta) = R/r, good to an
// ulp; the reference forms sqrt((1 - x)(1 + x)) from the rounded x = z/r, whose relative error is ~5.5e-17 / theta^2
// -- and its m >= 1 terms carry that into the tangential force of a lane near the axis (measured against a CPU restatement of the reference:
// 1.7e-3 at theta = 1e-6, 5e-5 at 1e-5, 6e-9 at 1e-3, 8e-11 at 1e-2).  Parity is with the reference's value, rounding
// noise included, so these lanes -- 5e-6 of an isotropic set -- go where its own formula is used.
#ifndef SPH_POLAR_FAC
#define SPH_POLAR_FAC 1.0e-5
#endif
// ... and sin^2(theta) below which a lane of the FAST evaluation takes the reference's cos(theta), sin(theta) and
// 1/(x*x - 1) instead of the accurate ones (x = z/r by a true division from the literally formed r, the square root of
// (1 - x)(1 + x), the product x*x rounded before the subtraction): between the two thresholds that is all that separates
// the fast arithmetic from the reference's (measured, the accurate values against the reference's: up to 3e-9 of the
// acceleration at theta = 3e-3, 8e-11 at 1e-2; with these: 1e-13) -- the m = 0 cancellation and the clamp that the
// general pass reproduces only matter further in.  Sending these lanes (5e-5 of an isotropic set, one wave in 300) to
// the general pass instead costs the 1e8-particle step 0.13 ms (profiles/r05_polar_ab.txt); this costs nothing
// measurable: a wave-uniform branch that one wave in 300 takes.
#ifndef SPH_POLAR_FAST
#define SPH_POLAR_FAST 1.0e-4
#endif

// The radial-derivative sum of one table slot, literally as the reference forms it: dpot(l, n) of SLGridSph::get_force
// (exputil/SLGridMP2.cc:954-989: ((p - 1/2) ef[j-1] p0[j-1] - 2 p ef[j] p0[j] + (p + 1/2) ef[j+1] p0[j+1]) / sqrt(ev), each
// product rounded on its own, no fused multiply-add), contracted with the coefficient row in ascending n
// (get_pot_coefs_safe, src/SphericalBasis.cc).  The common factor d_xi_to_r / dxi stays outside (it multiplies the finished
// sum here as everywhere on the device, a rounding of the RESULT, not of the cancelling terms).  Rare lanes only.
__device__ __noinline__ double sph_dp_lit_row(const SphDev &S, int row, int l, double p)
{
  const int edge = p < 0.0 ? 0 : 1;
  const int j = edge ? S.numr - 2 : 1;
  const size_t ln = (size_t)(S.lmax + 1) * S.nmax;
  const double *e0 = S.lit_ef + ((size_t)edge * 3 + 0) * ln + (size_t)l * S.nmax;
  const double *e1 = e0 + ln, *e2 = e1 + ln;
  const double *c = S.lit_coef + (size_t)row * S.nmax;
  const double pa = S.p0[j - 1], pb = S.p0[j], pc = S.p0[j + 1];
  const double wm = p - 0.5, wp = p + 0.5, w2 = 2.0 * p;
  double dp = 0.0;
  for (int n = 0; n < S.nmax; n++) {
    double a = wm * e0[n], b = w2 * e1[n], cc = wp * e2[n];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(cc));          // (the products are rounded before the next factor joins)
    a *= pa; b *= pb; cc *= pc;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(cc));
    double x = a - b;
    asm volatile("" : "+v"(x));
    x += cc;
    const double d = x / sqrt(S.lit_ev[l * S.nmax + n]);
    dp = mul_then_add(dp, d, c[n]);
  }
  return dp;
}
// ... for a slot of the projected table (its coefficient row and the scale of the rescaled Legendre functions)
__device__ __forceinline__ double sph_dp_lit(const SphDev &S, int slot, int l, double p)
{
  const int row = S.lit_rowmap[slot];
  return row < 0 ? 0.0 : S.lit_tscale[slot] * sph_dp_lit_row(S, row, l, p);
}

// One degree of the reference's own m = 0 Legendre recurrences (src/Basis.cc:71-92: p(l,0), then x*l*p(l,0) - l*p(l-1,0)
// with the pole-clamped x), every operation rounded on its own.  For lanes near the poles (1 - |x| < SPH_POLAR_FAC: they
// are all in the general pass): there |p(l,0)| is within theta^2 of 1, the difference is O(theta^2) with a rounding
// error of an ulp of p -- 1e-16 / theta^2 of it, 1e-8 at theta = 1e-4, all of it under the clamp -- and WHICH error
// depends on the exact rounding sequence; the rescaled recurrence of sph_field has another.  Measured against the
// reference's arithmetic before: 4e-9 of such a lane's tangential force at theta = 1e-4, 1e-8 at 1e-7.
__device__ __forceinline__ void leg0_lit_step(int l, double x, double xc, double &lp1, double &lp2, double &q)
{
#pragma clang fp contract(off)
  double p;
  if (l == 0) { p = 1.0; q = 0.0; }
  else if (l == 1) { p = x * 1 * lp1; q = xc * l * p - l * lp1; }
  else { p = (x * (2 * l - 1) * lp1 - (l - 1) * lp2) / l; q = xc * l * p - l * lp1; }
  lp2 = lp1;
  lp1 = p;
}

// (SPH_POLAR_FAST) the reference's angular inputs for the near-polar lanes of a fast evaluation
__device__ __forceinline__ void sph_polar_faithful(const SphDev &S, double xx, double yy, double zz, double fac, double r,
                                                   bool regular, double &costh, double &sinth, double &dfac)
{
  const bool polarish = regular && !(fac > SPH_POLAR_FAST * (r * r));
  if (__any(polarish)) {
    const double rl = sqrt(sq_add_lit(sq_sum2_lit(xx, yy), zz)) + S.dsmall;      // src/SphericalBasis.cc:1545
    const double xr = zz / rl;
    const double sr = sqrt((1.0 - xr) * (1.0 + xr));                              // src/Basis.cc:62
    const double dr = 1.0 / sq_add_lit(-1.0, xr);                                 // src/Basis.cc:86
    if (polarish) { costh = xr; sinth = sr; dfac = dr; }
  }
}

// General (slow-path) evaluation: per-lane table gathers, exterior continuation by per-lane selects,
// run-time flags, pole-clamped x in the derivative.  Waves that the fast pass deferred come here.
template <int LMAX, class PT, bool LIT = false>
__device__ __forceinline__ ForceOut
sph_field(const SphDev &S, double costh, double xc, double cphi, double sphi, PT t4,
          double x2, double pf, bool ioff, double rr, double kappa0, double pf_lit = 0.0)
{
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  // LIT: the instantiation for lanes far outside the table (sph_field_lit below); pf_lit: their force offset by the
  // reference's own division
  [[maybe_unused]] const bool lit = LIT && !ioff && (pf_lit < S.lit_lo || pf_lit > S.lit_hi);
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  const bool clamped = 1.0 - fabs(costh) < SPH_POLAR_FAC;      // (see leg0_lit_step)
  [[maybe_unused]] double lp1 = 0.0, lp2 = 0.0;
  double pmm = LC_E(0);
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  static_for<0, LMAX + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m == 1) {
      pmm *= LC_E(1) * somx2;
      cm = cphi; sm = sphi;
    } else if constexpr (m > 1) {
      pmm *= LC_E(m) * somx2;
      const double cn = 2.0 * cphi * cm - cm1;
      const double sn = 2.0 * cphi * sm - sm1;
      cm1 = cm; sm1 = sm;
      cm = cn; sm = sn;
    }
    bool m_on = true;
    if (S.EVEN_M && (m & 1)) m_on = false;
    if (S.M0_only && m != 0) m_on = false;
    // exterior continuation (src/SphericalBasis.cc:1605-1628): (rmax/r0)^(l+1), starts at l = m
    double rl;
    {
      double t = rr;
      static_for<0, m>([&](auto) { t *= rr; });
      rl = ioff ? t : 1.0;
    }
    double Al = 0.0, Bl = 0.0, Ar = 0.0, Br = 0.0, At = 0.0, Bt = 0.0;
    double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
    static_for<m, LMAX + 1>([&](auto lc_) {
      constexpr int l = decltype(lc_)::value;
      double plm, qlm;            // Ph(l,m) and (x^2-1) dPh(l,m)
      if constexpr (l == m) plm = pmm;
      else if constexpr (l == m + 1) plm = LC_a(l, m) * tprev;
      else plm = fma(LC_a(l, m), tprev, -pl2);
      tprev = costh * plm;
      if constexpr (l == m) qlm = (xc * plm) * l;
      else qlm = fma((double)l, xc * plm, -(LC_c(l, m) * pl1));
      if constexpr (m == 0) {
        if (clamped) {                  // (plm / lp1: this function's scale of Ph(l,0) against the reference's P_l)
          double ql;
          leg0_lit_step(l, costh, xc, lp1, lp2, ql);
          qlm = ql * (plm / lp1);
        }
      }
      pl2 = pl1;
      pl1 = plm;
      constexpr int q = 4 * t4_row(LMAX, l, m);
      bool on = m_on;
      if (l == 0 && S.NO_L0) on = false;
      if (l == 1 && S.NO_L1) on = false;
      if (l > 0 && S.EVEN_L && (l & 1)) on = false;
      if (on) {
        double pc = fma(x2, t4[q + 1], t4[q + 0]);
        double dpc = fma(pf, t4[q + 3], t4[q + 2]);
        if constexpr (LIT) { if (lit) dpc = sph_dp_lit(S, q / 4, l, pf_lit); }
        pc *= rl;
        dpc = ioff ? (kappa0 * (l + 1)) * pc : dpc;
        Al = fma(plm, pc, Al);
        Ar = fma(plm, dpc, Ar);
        At = fma(qlm, pc, At);
        if constexpr (m > 0) {
          double ps = fma(x2, t4[q + 5], t4[q + 4]);
          double dps = fma(pf, t4[q + 7], t4[q + 6]);
          if constexpr (LIT) { if (lit) dps = sph_dp_lit(S, q / 4 + 1, l, pf_lit); }
          ps *= rl;
          dps = ioff ? (kappa0 * (l + 1)) * ps : dps;
          Bl = fma(plm, ps, Bl);
          Br = fma(plm, dps, Br);
          Bt = fma(qlm, ps, Bt);
        }
      }
      rl *= ioff ? rr : 1.0;
    });
    if constexpr (m == 0) {
      o.potl += Al;
      o.potr += Ar;
      o.pott += At;
    } else {
      o.potl += Al * cm + Bl * sm;
      o.potr += Ar * cm + Br * sm;
      o.pott += At * cm + Bt * sm;
      o.potp += (Bl * cm - Al * sm) * m;
    }
    // bound the scheduling region to one m-block
    __builtin_amdgcn_sched_barrier(0);
  });
  return o;
}

// ---- fast path with software-pipelined scalar table loads --------------------------------------------
// hipcc issues each (l,m) block's s_load right before its first use and waits for it at once, so
// every one of the (L+1)(L+2)/2 blocks exposes a full scalar-cache round trip.  Here the loads are
// written by hand (two SGPR buffers): block k+1 is requested before the FMAs of block k and only
// waited for after them.  The inline asm follows cdna_hip_programming.md section 5.7: the asm owns its
// waits (lgkmcnt(0), SMEM may return out of order) and every consumer is data-dependent on the
// wait statement ("+s"), so nothing can be scheduled between a load and its wait that reads it.
typedef double sd8 __attribute__((ext_vector_type(8)));

#define SLOAD8(dst, base, off_doubles) \
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(dst) : "s"(base), "n"((off_doubles) * 8))
#define SWAIT8(v) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v))

// m-major traversal: k-th (l,m) block, its table offset (doubles) -- all constexpr
template <int LMAX> __host__ __device__ constexpr int blk_m(int k)
{
  int m = 0, base = 0;
  while (k >= base + (LMAX - m + 1)) { base += LMAX - m + 1; m++; }
  return m;
}
template <int LMAX> __host__ __device__ constexpr int blk_l(int k)
{
  int m = 0, base = 0;
  while (k >= base + (LMAX - m + 1)) { base += LMAX - m + 1; m++; }
  return m + (k - base);
}

template <int LMAX>
__device__ __forceinline__ ForceOut
sph_field_fast(cdp t4, double costh, double somx2, double cphi, double sphi, double x2, double pf)
{
  constexpr int NBLK = (LMAX + 1) * (LMAX + 2) / 2;
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  const unsigned long long tb = (unsigned long long)t4;
  double pmm = LC_E(0);
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  double Ag = 0.0, Ad = 0.0, Tg = 0.0, Td = 0.0, Rb = 0.0, Ra = 0.0;       // cos rows
  double Bg = 0.0, Bd = 0.0, Ug = 0.0, Ud = 0.0, Sb = 0.0, Sa = 0.0;       // sin rows
  double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
  // m = 0 rows are 4 doubles each: fetch them two rows at a time as 8-double blocks where possible
  sd8 cur, nxt;
  asm volatile("s_load_dwordx16 %0, %1, 0" : "=s"(cur) : "s"(tb));
  // (touching all 61 lines of the cell with scalar loads up front -- a scalar-cache warm-up for the first waves
  // of a cell on a CU pair -- measured neutral once the rows are L2-resident: the prefetch in k_sph_force)
  SWAIT8(cur);
  static_for<0, NBLK>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int l = blk_l<LMAX>(k), m = blk_m<LMAX>(k);
    constexpr int q = 4 * t4_row(LMAX, l, m);
    // which 8-double window holds this block, and the next one to request
    constexpr int win = (m == 0) ? (q / 8) * 8 : q;          // m = 0: rows pair up in one window
    constexpr bool last = (k + 1 == NBLK);
    constexpr int l1 = last ? l : blk_l<LMAX>(k + 1), m1 = last ? m : blk_m<LMAX>(k + 1);
    constexpr int q1 = 4 * t4_row(LMAX, l1, m1);
    constexpr int win1 = (m1 == 0) ? (q1 / 8) * 8 : q1;
    constexpr bool need = !last && win1 != win;
    if constexpr (need) { SLOAD8(nxt, tb, win1); __builtin_amdgcn_sched_barrier(0); }

    if constexpr (l == m) {            // start of an m-block
      if constexpr (m == 1) { pmm *= LC_E(1) * somx2; cm = cphi; sm = sphi; }
      else if constexpr (m > 1) {
        pmm *= LC_E(m) * somx2;
        const double cn = 2.0 * cphi * cm - cm1, sn = 2.0 * cphi * sm - sm1;
        cm1 = cm; sm1 = sm; cm = cn; sm = sn;
      }
      Ag = Ad = Tg = Td = Rb = Ra = 0.0;
      Bg = Bd = Ug = Ud = Sb = Sa = 0.0;
      pl2 = pl1 = tprev = 0.0;
    }
    double plm, qlm;        // Ph(l,m), (x^2-1) dPh(l,m); no pole lanes here, so x == clamped x
    if constexpr (l == m) plm = pmm;
    else if constexpr (l == m + 1) plm = LC_a(l, m) * tprev;
    else plm = fma(LC_a(l, m), tprev, -pl2);
    tprev = costh * plm;
    if constexpr (l == m) qlm = tprev * l;
    else qlm = fma((double)l, tprev, -(LC_c(l, m) * pl1));
    pl2 = pl1;
    pl1 = plm;
    constexpr int o0 = q - win;        // 0 or 4 inside the window
    Ag = fma(plm, cur[o0 + 0], Ag);
    Ad = fma(plm, cur[o0 + 1], Ad);
    Rb = fma(plm, cur[o0 + 2], Rb);
    Ra = fma(plm, cur[o0 + 3], Ra);
    if constexpr (l > 0) {       // (x^2-1) dPh(0,0) = 0
      Tg = fma(qlm, cur[o0 + 0], Tg);
      Td = fma(qlm, cur[o0 + 1], Td);
    }
    if constexpr (m > 0) {
      Bg = fma(plm, cur[4], Bg);
      Bd = fma(plm, cur[5], Bd);
      Sb = fma(plm, cur[6], Sb);
      Sa = fma(plm, cur[7], Sa);
      Ug = fma(qlm, cur[4], Ug);
      Ud = fma(qlm, cur[5], Ud);
    }
    if constexpr (l == LMAX) {         // end of an m-block: apply the per-particle weights once
      const double Al = fma(x2, Ad, Ag), At = fma(x2, Td, Tg), Ar = fma(pf, Ra, Rb);
      if constexpr (m == 0) {
        o.potl += Al; o.potr += Ar; o.pott += At;
      } else {
        const double Bl = fma(x2, Bd, Bg), Bt = fma(x2, Ud, Ug), Br = fma(pf, Sa, Sb);
        o.potl += Al * cm + Bl * sm;
        o.potr += Ar * cm + Br * sm;
        o.pott += At * cm + Bt * sm;
        o.potp += (Bl * cm - Al * sm) * m;
      }
    }
    if constexpr (need) { __builtin_amdgcn_sched_barrier(0); SWAIT8(nxt); cur = nxt; }
  });
  return o;
}

// The fast pass' arithmetic with the rows behind a PER-LANE pointer (LDS-staged rows of k_sph_force_staged, or global
// memory): the same factored sums, recurrences and operation order as sph_field_fast -- results are bit-identical to
// it for the same cell -- without the scalar-load pipeline.
template <int LMAX, class PT>
__device__ __forceinline__ ForceOut
sph_field_fast_ptr(PT t4, double costh, double somx2, double cphi, double sphi, double x2, double pf)
{
  constexpr int NBLK = (LMAX + 1) * (LMAX + 2) / 2;
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  double pmm = LC_E(0);
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  double Ag = 0.0, Ad = 0.0, Tg = 0.0, Td = 0.0, Rb = 0.0, Ra = 0.0;
  double Bg = 0.0, Bd = 0.0, Ug = 0.0, Ud = 0.0, Sb = 0.0, Sa = 0.0;
  double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
  static_for<0, NBLK>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int l = blk_l<LMAX>(k), m = blk_m<LMAX>(k);
    constexpr int q = 4 * t4_row(LMAX, l, m);
    if constexpr (l == m) {
      if constexpr (m == 1) { pmm *= LC_E(1) * somx2; cm = cphi; sm = sphi; }
      else if constexpr (m > 1) {
        pmm *= LC_E(m) * somx2;
        const double cn = 2.0 * cphi * cm - cm1, sn = 2.0 * cphi * sm - sm1;
        cm1 = cm; sm1 = sm; cm = cn; sm = sn;
      }
      Ag = Ad = Tg = Td = Rb = Ra = 0.0;
      Bg = Bd = Ug = Ud = Sb = Sa = 0.0;
      pl2 = pl1 = tprev = 0.0;
    }
    double plm, qlm;
    if constexpr (l == m) plm = pmm;
    else if constexpr (l == m + 1) plm = LC_a(l, m) * tprev;
    else plm = fma(LC_a(l, m), tprev, -pl2);
    tprev = costh * plm;
    if constexpr (l == m) qlm = tprev * l;
    else qlm = fma((double)l, tprev, -(LC_c(l, m) * pl1));
    pl2 = pl1;
    pl1 = plm;
    const double g0 = t4[q + 0], g1 = t4[q + 1], g2 = t4[q + 2], g3 = t4[q + 3];
    Ag = fma(plm, g0, Ag);
    Ad = fma(plm, g1, Ad);
    Rb = fma(plm, g2, Rb);
    Ra = fma(plm, g3, Ra);
    if constexpr (l > 0) {
      Tg = fma(qlm, g0, Tg);
      Td = fma(qlm, g1, Td);
    }
    if constexpr (m > 0) {
      const double h0 = t4[q + 4], h1 = t4[q + 5], h2 = t4[q + 6], h3 = t4[q + 7];
      Bg = fma(plm, h0, Bg);
      Bd = fma(plm, h1, Bd);
      Sb = fma(plm, h2, Sb);
      Sa = fma(plm, h3, Sa);
      Ug = fma(qlm, h0, Ug);
      Ud = fma(qlm, h1, Ud);
    }
    if constexpr (l == LMAX) {
      const double Al = fma(x2, Ad, Ag), At = fma(x2, Td, Tg), Ar = fma(pf, Ra, Rb);
      if constexpr (m == 0) {
        o.potl += Al; o.potr += Ar; o.pott += At;
      } else {
        const double Bl = fma(x2, Bd, Bg), Bt = fma(x2, Ud, Ug), Br = fma(pf, Sa, Sb);
        o.potl += Al * cm + Bl * sm;
        o.potr += Ar * cm + Br * sm;
        o.pott += At * cm + Bt * sm;
        o.potp += (Bl * cm - Al * sm) * m;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  return o;
}

// out-of-line copy for the waterfall loop of MODE 2: inlined into a loop, the (l, m) recurrence literals
// are hoisted out of it as loop invariants -- several hundred SGPRs, i.e. a kernel that lives in scratch
template <int LMAX>
__device__ __noinline__ ForceOut
sph_field_fast_call(cdp t4, double costh, double somx2, double cphi, double sphi, double x2, double pf)
{
  // (arguments of a real call arrive in vector registers: the wave-uniform table address goes back
  // to scalar registers for the s_load pipeline)
  const unsigned long long a = (unsigned long long)t4;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return sph_field_fast<LMAX>((cdp)(((unsigned long long)hi << 32) | lo), costh, somx2, cphi, sphi, x2, pf);
}

// Two launches share this body.  FAST: every wave whose lanes sit in one radial cell with no
// exterior particle is done here; the rest push their first slot on `work` and leave.
// !FAST: one wave per work item (or per 64-slot chunk when work == nullptr, i.e. "all waves").
#ifndef SPH_FORCE_WAVES
#define SPH_FORCE_WAVES 5      // min waves/SIMD for the fast pass: 5 beats 4 and 6 on MI355X (A/B, profiles/)
#endif
#ifndef SPH_FORCE_CHUNKS
#define SPH_FORCE_CHUNKS 1
#endif
#ifndef SPH_T4_PREFETCH
#define SPH_T4_PREFETCH 2     // cells ahead whose table rows a fast-pass wave pulls into L2 (0: off)
#endif
// One 64-particle chunk of one wave (slots base .. base+63 of [.., end)).
// MODE 0: general evaluation of the lanes in `lanemask`.  MODE 1: the fast pass (a wave either is
// cell-uniform and done here, or is deferred whole).  MODE 2: the fast pass as a WATERFALL -- the
// wave's distinct radial cells are served one after the other by the same scalar-table code, so
// every non-special lane gets exactly the arithmetic it would get in a uniform wave whatever its
// neighbours are (deterministic mode), and targets that are not in this basis' cell order but
// still local in radius (another component's particles) avoid the gather path; only the special
// lanes (polar axis, exterior) go to the general pass, by lane mask.
// Work-list entries are three words: first slot, lane mask low / high.
#define SPH_WORK_STRIDE 3

// ---- the "append" fused step (sph.hip: SphForce::fused_step_append) ---------------------------------------------------------
// The force pass of a fused step knows where every particle will be at the NEXT step (it computes that step's sort key);
// here it also puts the particle there: each cell of the next step's order owns a region [base[c], base[c + 1]) of the OTHER
// buffer set, sized from the cell's current population plus slack, and a cursor; a block counts its particles per destination
// cell in LDS, reserves with ONE atomic per (block, cell) -- one per (wave, cell) serialises on the few cells the resident
// waves feed, one per (block, cell) is free: tools/dbg/append_cursor.hip -- and every lane stores its particle at its reserved
// slot.  The sort passes of the next step (key histogram, scan, scatter: 2.2 of the 10.1 ms step at 1e8, 112 B per
// particle) disappear.  Slots a region does not fill hold x = +inf (k_app_finish): the accumulation and force passes treat
// such a slot as empty; a region that overflows spills into a tail region, the tail into `flag` (the step is then redone
// the ordinary way from the source buffer, which is never written).
#ifndef APP_EMPTY
#define APP_EMPTY __builtin_inf()    // (particles.h) x of a slot that holds no particle
#endif
struct AppDev {
  double *X, *Y, *Z;                 // destination: the position the NEXT accumulation and force pass read
  uint32_t *SRC;                     // ... and the slot of the SOURCE set the particle came from: the position of THIS step (the
                                     // state a download or any other call sees) stays there until the next pass -- 4 bytes
                                     // instead of 24, and this pass is sensitive to what it stores (profiles/r06_append_ab.txt)
  double *VX, *VY, *VZ;
  double *AX, *AY, *AZ, *POT;        // nullptr (all four): the LEAN payload -- acceleration and potential are not placed, 32 of
                                     // the 88 bytes, which no pass of the next step reads; whoever asks for the state of the
                                     // completed step has them re-evaluated (sph.hip: sph_app_reeval; exp_amd_ctx_set_append_lean)
  double *M;                         // nullptr: uniform mass (both buffer sets hold the constant)
  uint32_t *ID;
  const double *Msrc;
  const uint32_t *IDsrc;
  const uint32_t *base;              // [ncell + 2]: start of every cell's region, of the tail, end of the tail
  uint32_t *cursor;                  // [ncell + 1]: arrivals per cell (all of them, spilled ones included) and in the tail
  uint32_t *flag;                    // particles that found no room at all
  uint32_t ncell;
};
struct AppOut {                      // one lane's particle, ready to be placed
  double nx, ny, nz, vx, vy, vz, ax, ay, az, pot;
  uint32_t cell;
};
#define APP_TAB 32
struct AppShared { uint32_t cell[APP_TAB], cnt[APP_TAB], base[APP_TAB]; };

// LEAN: 1 -- the lean payload (compile time: the fast pass), 0 -- the full one, -1 -- whichever A says (the general pass)
template <int LEAN = -1>
__device__ __forceinline__ void app_store_at(const AppDev &A, size_t slot, const AppOut &o, size_t isrc);
template <int LEAN = -1>
__device__ __forceinline__ void app_store(const AppDev &A, uint32_t cell, uint32_t r, const AppOut &o, size_t isrc)
{
  size_t slot;
  const uint32_t cap = A.base[cell + 1] - A.base[cell];
  if (r < cap) slot = (size_t)A.base[cell] + r;
  else {
    // the region is full: the tail (still a valid place: the passes recompute every particle's cell from its position)
    const uint32_t t = atomicAdd(&A.cursor[A.ncell], 1u);
    if (t >= A.base[A.ncell + 1] - A.base[A.ncell]) { atomicAdd(A.flag, 1u); return; }
    slot = (size_t)A.base[A.ncell] + t;
  }
  app_store_at<LEAN>(A, slot, o, isrc);
}
template <int LEAN>
__device__ __forceinline__ void app_store_at(const AppDev &A, size_t slot, const AppOut &o, size_t isrc)
{
  A.X[slot] = o.nx; A.Y[slot] = o.ny; A.Z[slot] = o.nz;
  A.SRC[slot] = (uint32_t)isrc;
  A.VX[slot] = o.vx; A.VY[slot] = o.vy; A.VZ[slot] = o.vz;
  if (LEAN == 0 || (LEAN < 0 && A.AX)) {
    A.AX[slot] = o.ax; A.AY[slot] = o.ay; A.AZ[slot] = o.az;
    A.POT[slot] = o.pot;
  }
  if (A.M) A.M[slot] = A.Msrc[isrc];
  A.ID[slot] = A.IDsrc[isrc];
}

// a wave enters its particles in the block's table (one entry per destination cell): h = the entry (-1: the table is full:
// the lane reserves for itself), rank = its place among the block's particles of that cell
__device__ __forceinline__ void app_block_register(AppShared &sh, bool have, uint32_t cell, int &h, uint32_t &rank)
{
  const int lane = threadIdx.x & 63;
  h = -1;
  rank = 0;
  unsigned long long todo = __ballot(have);
  while (todo) {
    const int lead = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(cell, lead);
    const unsigned long long m = __ballot(have && cell == c);
    int hh = -1;
    uint32_t woff = 0;
    if (lane == lead) {
      uint32_t k = c & (APP_TAB - 1);
      for (int probe = 0; probe < APP_TAB; probe++, k = (k + 1) & (APP_TAB - 1)) {
        const uint32_t old = atomicCAS(&sh.cell[k], 0xffffffffu, c);
        if (old == 0xffffffffu || old == c) { hh = (int)k; break; }
      }
      if (hh >= 0) woff = atomicAdd(&sh.cnt[hh], (uint32_t)__popcll(m));
    }
    hh = __shfl(hh, lead);
    woff = __shfl(woff, lead);
    if (have && cell == c) { h = hh; rank = woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); }
    todo &= ~m;
  }
}
// ... after a barrier one thread per entry reserves in the cell's cursor; after another, the lanes store
__device__ __forceinline__ void app_block_reserve(AppShared &sh, const AppDev &A)
{
  if (threadIdx.x < APP_TAB && sh.cell[threadIdx.x] != 0xffffffffu)
    sh.base[threadIdx.x] = atomicAdd(&A.cursor[sh.cell[threadIdx.x]], sh.cnt[threadIdx.x]);
}
// the general pass' few waves: one atomic per (wave, cell)
__device__ __forceinline__ void app_wave_store(const AppDev &A, bool have, const AppOut &o, size_t isrc)
{
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(have);
  while (todo) {
    const int lead = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(o.cell, lead);
    const unsigned long long m = __ballot(have && o.cell == c);
    uint32_t b = 0;
    if (lane == lead) b = atomicAdd(&A.cursor[c], (uint32_t)__popcll(m));
    b = __shfl(b, lead);
    if (have && o.cell == c) app_store(A, c, b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), o, isrc);
    todo &= ~m;
  }
}

// The Cartesian projection and the stores of one particle's field (src/SphericalBasis.cc:1636-1652), with the fused
// half-kick and the next step's sort key of the fused step: shared by every evaluation path.
template <bool FAST, int APP = 0>
__device__ __forceinline__ void
sph_force_finish(const SphDev &S, const ForceOut &o, size_t i, double xx, double yy, double zz, double px, double py,
                 double pz, double fac, double ir, double iR2, double P0, double ffac, double dfac,
                 double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT,
                 double *__restrict__ VX, double *__restrict__ VY, double *__restrict__ VZ, double dt_kick, int assign,
                 uint32_t *__restrict__ key_out, double nk_dtk, double nk_dtd, int store_v, AppOut *ao = nullptr)
{
  // src/SphericalBasis.cc:1636-1652 (r is the clamped radius, as in the reference)
  const double potr = o.potr * ffac * (S.inv_scale * S.inv_scale);
  const double potl = o.potl * P0 * S.inv_scale;
  const double pott = o.pott * (P0 * dfac) * S.inv_scale;
  const double potp = o.potp * P0 * S.inv_scale;
  const double ir3 = ir * ir * ir;
  const double pr = potr * ir, pt3 = pott * ir3;
  double ax = -(pr * xx - pt3 * xx * zz);
  double ay = -(pr * yy - pt3 * yy * zz);
  double az = -(pr * zz + pt3 * fac);
  // (the n-body thread body adds the azimuthal term `if (fac > DSMALL)`, src/SphericalBasis.cc:1647; pyEXP's
  // Spherical::computeAccel -- the no_exterior mode -- adds potp*y/R2 always, expui/BiorthBasis.cc:918-919, and within
  // 1e-8 of the axis that term is of order one: P_l^1 / sin(theta).  On the axis itself it is 0/0 there; not added here.)
  const double fac_floor = S.no_exterior ? 0.0 : DSMALL;
  if (fac > fac_floor) {
    const double pf2 = FAST ? potp * iR2 : potp / fac;
    ax += pf2 * yy;
    ay += -pf2 * xx;
  }
  if (S.ps.center | S.ps.axis) {
    // Component::AddAcc(i, j, val) is acc[j] += val - pseudo[j] on EVERY call (src/Component.H:914-921)
    // and the reference's thread body calls it for x, y, z and, when fac > DSMALL, for x and y once
    // more (the potp term, src/SphericalBasis.cc:1645-1651): x and y lose the frame term twice.
    double qx, qy, qz, ux = 0.0, uy = 0.0, uz = 0.0;
    if (S.ps.axis) { ux = VX[i]; uy = VY[i]; uz = VZ[i]; }
    pseudo_accel(S.ps, px, py, pz, ux, uy, uz, qx, qy, qz);
    ax -= qx; ay -= qy; az -= qz;
    if (fac > DSMALL) { ax -= qx; ay -= qy; }
  }
  double pt = potl;
  if constexpr (APP) {
    // the append step (fused, self force, assign, velocities stored with the next opening half-kick: store_v == 2): the
    // arithmetic of the stores below, the results left in registers for the caller to place
    const double vx = mul_then_add(VX[i], ax, dt_kick);
    const double vy = mul_then_add(VY[i], ay, dt_kick);
    const double vz = mul_then_add(VZ[i], az, dt_kick);
    const double wx = mul_then_add(vx, ax, nk_dtk);
    const double wy = mul_then_add(vy, ay, nk_dtk);
    const double wz = mul_then_add(vz, az, nk_dtk);
    ao->nx = mul_then_add(px, wx, nk_dtd);
    ao->ny = mul_then_add(py, wy, nk_dtd);
    ao->nz = mul_then_add(pz, wz, nk_dtd);
    ao->vx = wx; ao->vy = wy; ao->vz = wz;
    if constexpr (APP != 2) { ao->ax = ax; ao->ay = ay; ao->az = az; ao->pot = pt; }
    ao->cell = sph_key_cell_rcp(S, ao->nx, ao->ny, ao->nz);
    return;
  }
  // a frozen target particle is skipped by the thread body (src/SphericalBasis.cc:1521): nothing is added, the frame
  // term neither; the fused half-kick below still applies whatever other forces left in acc
  if (SPH_FRZ_ON(S) && sph_frozen(S, px, py, pz)) { ax = ay = az = 0.0; pt = 0.0; }
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
    const double vx = mul_then_add(VX[i], ax, dt_kick);
    const double vy = mul_then_add(VY[i], ay, dt_kick);
    const double vz = mul_then_add(VZ[i], az, dt_kick);
    if (store_v == 1) { VX[i] = vx; VY[i] = vy; VZ[i] = vz; }   // 0: deferred (exp_amd_comp::pending_kick)
    if (key_out) {
      // Where this particle will be after the NEXT step's kick + drift (the arithmetic of
      // advance_one, sort_kernels.h, on the values just stored): its sort key.  The next step
      // then histograms the 4-byte keys (k_hist_keys) instead of re-reading x, v, a (72 B).
      // (Counting the keys here as well, one atomic per distinct key per wave, doubled this
      // kernel's time: 5e6 atomics on ~2000 hot addresses.)
      const double wx = mul_then_add(vx, ax, nk_dtk);
      const double wy = mul_then_add(vy, ay, nk_dtk);
      const double wz = mul_then_add(vz, az, nk_dtk);
      // store_v == 2: the velocities go out WITH the next step's opening half-kick (the same two
      // rounding steps that step's scatter pass would take), so that pass only drifts and never
      // reads the accelerations (24 B/particle); pending_kick = -nk_dtk tells everyone else
      if (store_v == 2) { VX[i] = wx; VY[i] = wy; VZ[i] = wz; }
      const uint32_t key = sph_key_cell_rcp(S, mul_then_add(px, wx, nk_dtd),
                                            mul_then_add(py, wy, nk_dtd), mul_then_add(pz, wz, nk_dtd));
      key_out[i] = key + S.key_add;
    }
  }
}

// APP (the append step, MODE 0 and 1): the lane's results are left in *ao instead of being stored (the return value says
// whether the lane has any); the store it reads may hold empty slots (x = +inf).
template <int LMAX, int MODE, int APP = 0 /* 1: the append step; 2: ... with the lean payload (fast pass) */>
__device__ __forceinline__ bool
sph_force_chunk(const SphDev &S, const double *__restrict__ X, const double *__restrict__ Y,
                const double *__restrict__ Z, size_t base, size_t end, const double *__restrict__ T4,
                double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ,
                double *__restrict__ POT, double *__restrict__ VX, double *__restrict__ VY,
                double *__restrict__ VZ, double dt_kick, int assign, uint32_t *__restrict__ work,
                uint32_t *__restrict__ nwork, uint32_t *__restrict__ key_out, double nk_dtk,
                double nk_dtd, int store_v, unsigned long long lanemask = ~0ull, AppOut *ao = nullptr)
{
  constexpr bool FAST = MODE != 0 && MODE != 3;
  const int lane = threadIdx.x & 63;
  // MODE 3 (the literal pass): base / end index the list of slots the evaluation passes left behind
  size_t i = base + lane;
  bool valid = i < end && ((lanemask >> lane) & 1ull);
  if constexpr (MODE == 3) i = valid ? S.lit_list[1 + i] : 0;
  double xx = 1, yy = 0, zz = 0;          // idle lanes: a harmless off-axis point
  double px = 0, py = 0, pz = 0;          // kept for the next-step key (a reload at the end of the
  if (valid) {                            // wave would expose a full memory round trip)
    px = X[i]; py = Y[i]; pz = Z[i];
    if constexpr (APP) valid = px != APP_EMPTY;
  }
  if constexpr (APP) { if (!__any(valid)) return false; }        // (a wave of empty slots: the end of a cell's region)
  if (valid) {
    xx = px - S.cx;
    yy = py - S.cy;
    zz = pz - S.cz;
  }
  double fac = xx * xx + yy * yy;
  const size_t tq = (size_t)4 * S.trows;
  double r, ir, iR2, P0, ffac, dfac;
  [[maybe_unused]] double t4_sink = 0.0;      // destination of the table prefetch of the fast pass
  ForceOut o;
  if constexpr (FAST) {
    // Same quantities as the general path below with the divisions shared: one reciprocal each of
    // r and R = sqrt(x^2+y^2) gives cos(theta), sin(theta) = R/r, cos/sin(phi), 1/(x^2-1) =
    // -(r/R)^2 and the 1/r, 1/R^2 of the Cartesian projection (10 fp64 divisions -> 4; results
    // agree with the reference's formulas to a few ulp).  Lanes on the polar axis, where the
    // reference clamps x (src/Basis.cc:81-84), outside rmax, or waves spanning several radial
    // cells are left to the general pass.
    double g, y, R, iR;
    sqrt_rsqrt(fac + zz * zz, g, y);
    r = g + S.dsmall;                                        // src/SphericalBasis.cc:1545-1560
    ir = rcp_refine(r, y);
    double costh = zz * ir;
    sqrt_rsqrt(fac, R, iR);
    iR2 = iR * iR;
    const double cphi = xx * iR, sphi = yy * iR;
    double sinth = R * ir;
    // sin^2(theta) < SPH_POLAR_FAC: sin(theta) = R/r and the reference's sqrt((1-x)(1+x)) differ by more than the
    // parity tolerance there (cancellation in 1-x), so those lanes take the reference's formula too
    const double xi = sph_r_to_xi_rcp(S, r * S.inv_scale);
    // (far outside the table -- the logarithmic map only -- the radial derivative takes the reference's literal
    // evaluation in the general pass: sph_dp_lit)
    const bool special = (r > S.rmax && !S.no_exterior) || !(fac > SPH_POLAR_FAC * (r * r)) || !(fac > DSMALL) || r < SPH_TINY_R * S.dsmall ||
                         xi < S.lit_xlo || xi > S.lit_xhi;
    int idx = sph_cell(S, xi);
    ffac = sph_d_xi_to_r_rcp(S, xi) * S.inv_dxi;
    dfac = -(r * r) * iR2;
    sph_polar_faithful(S, xx, yy, zz, fac, r, valid && !special, costh, sinth, dfac);
    if constexpr (MODE == 1) {
      const int idx_u = __builtin_amdgcn_readfirstlane(idx);
      if (!valid) idx = idx_u;
      const bool fast = __all(idx == idx_u) && !__any(special && valid);
      if (!fast) {
        if (lane == 0) {
          const uint32_t w = atomicAdd(nwork, 1u);
          work[SPH_WORK_STRIDE * w] = (uint32_t)base;
          work[SPH_WORK_STRIDE * w + 1] = 0xffffffffu;
          work[SPH_WORK_STRIDE * w + 2] = 0xffffffffu;
        }
        return false;
      }
#if SPH_T4_PREFETCH
      {
        // Warm this XCD's L2 with the table rows of the cell SPH_T4_PREFETCH ahead.  The whole table (7.7 MB
        // at S10) does not fit the 4 MB L2 of an XCD, so the first waves to reach a new cell would take each
        // of their 61 scalar 64-byte loads from HBM, one behind the other (the s_load pipeline is one block
        // deep); at 5e8 particles, where those front waves are five times rarer, the same kernel runs 9 %
        // faster per particle.  One vector load per wave, a cache line per lane, result never used.
        int pc = idx_u + SPH_T4_PREFETCH;
        pc = pc > S.numr - 2 ? S.numr - 2 : pc;
        const double *pp = T4 + (size_t)pc * tq + (size_t)lane * 8;
        // (the load is asynchronous and the compiler does not know it: the destination registers stay
        // reserved until the end of the wave -- see the matching fake use below -- or the data would land in
        // whatever had been given those registers in the meantime)
        if ((size_t)lane * 8 < tq)
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t4_sink) : "v"(pp) : "memory");
      }
#endif
      // get_pot / get_force weights (exputil/SLGridMP2.cc:894-902, :971-985)
      const double x1 = (S.xi[idx_u + 1] - xi) * S.inv_dxi;
      const double x2 = (xi - S.xi[idx_u]) * S.inv_dxi;
      P0 = x1 * S.p0[idx_u] + x2 * S.p0[idx_u + 1];
      const int jdx = idx_u < 1 ? 1 : idx_u;
      const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
      cdp t4 = (cdp)(T4 + (size_t)idx_u * tq);
      o = sph_field_fast<LMAX>(t4, costh, sinth, cphi, sphi, x2, pf);
    } else {
      // the same weights from the lane's own cell (gathers of the small xi / p0 tables: same operands,
      // same operations as above)
      const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
      const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
      P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
      const int jdx = idx < 1 ? 1 : idx;
      const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
      const unsigned long long sp = __ballot(valid && special);
      if (sp && lane == 0) {              // the special lanes of this chunk: general pass, by lane mask
        const uint32_t w = atomicAdd(nwork, 1u);
        work[SPH_WORK_STRIDE * w] = (uint32_t)base;
        work[SPH_WORK_STRIDE * w + 1] = (uint32_t)sp;
        work[SPH_WORK_STRIDE * w + 2] = (uint32_t)(sp >> 32);
      }
      if (special) valid = false;
#if SPH_T4_PREFETCH
      {                       // the same L2 warm-up as in the fast pass, from the first lane's cell
        int pc = __builtin_amdgcn_readfirstlane(idx) + SPH_T4_PREFETCH;
        pc = pc > S.numr - 2 ? S.numr - 2 : pc;
        const double *pp = T4 + (size_t)pc * tq + (size_t)lane * 8;
        if ((size_t)lane * 8 < tq)
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t4_sink) : "v"(pp) : "memory");
      }
#endif
      unsigned long long todo = __ballot(valid);
      o = ForceOut{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
      while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int idx_u = __builtin_amdgcn_readfirstlane(__shfl(idx, lead));
        const bool sel = valid && idx == idx_u;
        cdp t4 = (cdp)(T4 + (size_t)idx_u * tq);
        const ForceOut q = sph_field_fast_call<LMAX>(t4, costh, sinth, cphi, sphi, x2, pf);
        if (sel) o = q;
        todo &= ~__ballot(sel);
      }
    }
  } else {
    // src/SphericalBasis.cc:1545-1560
    fac = sq_sum2_lit(xx, yy);
    r = sqrt(sq_add_lit(fac, zz)) + S.dsmall;
    const double costh = zz / r;
    double cphi, sphi;
    phi_trig(xx, yy, cphi, sphi);
    bool ioff = false;
    const double r0 = r;
    if (r > S.rmax && !S.no_exterior) {
      ioff = true;
      r = S.rmax;
    }
    const double rs = r / S.scale;
    const double xi = sph_r_to_xi(S, rs);
    const int idx = sph_cell(S, xi);
    // get_pot weights (exputil/SLGridMP2.cc:894-902)
    const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
    const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
    P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
    // get_force weights (exputil/SLGridMP2.cc:971-985)
    const int jdx = idx < 1 ? 1 : idx;
    const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
    ffac = sph_d_xi_to_r(S, xi) * S.inv_dxi;
    // Legendre derivative pole clamp (src/Basis.cc:81-84)
    double xc = costh;
    if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
    dfac = 1.0 / sq_add_lit(-1.0, xc);
    const double rr = S.rmax / r0;
    const double kappa0 = -P0 / (r0 * ffac);     // dp = -(l+1)/r0 * p, in units of ffac
    const double *t4 = T4 + (size_t)idx * tq;
    if constexpr (MODE == 3) {
      o = sph_field<LMAX, const double *, true>(S, costh, xc, cphi, sphi, t4, x2, pf, ioff, rr, kappa0,
                                                (xi - S.xi[jdx]) / S.dxi);
    } else {
      o = sph_field<LMAX>(S, costh, xc, cphi, sphi, t4, x2, pf, ioff, rr, kappa0);
      // far outside the table: the literal pass takes the particle (sph_dp_lit)
      if (valid && !ioff && (pf < S.lit_lo || pf > S.lit_hi)) {
        const uint32_t k = atomicAdd(S.lit_list, 1u);
        if (k < S.lit_cap) { S.lit_list[1 + k] = (uint32_t)i; valid = false; }
      }
    }
    ir = 1.0 / r;
    iR2 = 1.0 / fac;
  }
  if (valid)
    sph_force_finish<FAST, APP>(S, o, i, xx, yy, zz, px, py, pz, fac, ir, iR2, P0, ffac, dfac, AX, AY, AZ, POT, VX, VY, VZ,
                                dt_kick, assign, key_out, nk_dtk, nk_dtd, store_v, ao);
#if SPH_T4_PREFETCH
  if constexpr (MODE != 0) asm volatile("" : : "v"(t4_sink));     // keeps the prefetch's registers out of circulation
#endif
  return valid;
}

#ifndef SPH_APP_WAVES
#define SPH_APP_WAVES SPH_FORCE_WAVES
#endif
template <int LMAX, int MODE, int APP = 0>
__global__ void __launch_bounds__(256, MODE == 1 ? (APP ? SPH_APP_WAVES : SPH_FORCE_WAVES) : MODE == 2 ? 2 : 1)
k_sph_force(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
            const double *__restrict__ Z, const uint32_t *__restrict__ lev_off, int lev_lo,
            int lev_hi, const double *__restrict__ T4, double *__restrict__ AX,
            double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT,
            double *__restrict__ VX, double *__restrict__ VY, double *__restrict__ VZ,
            double dt_kick, int assign, uint32_t *__restrict__ work, uint32_t *__restrict__ nwork,
            uint32_t *__restrict__ key_out, double nk_dtk, double nk_dtd, int store_v,
            uint32_t *__restrict__ nwork_clear /* counter of the NEXT launch pair: zeroed here */,
            AppDev app = AppDev{})
{
  if (nwork_clear && blockIdx.x == 0 && threadIdx.x == 0) *nwork_clear = 0u;
  if constexpr (APP && MODE == 1) {
    // the append step's fast pass: every wave of the block reaches the two barriers (no early exit); a deferred wave has
    // nothing to place here -- the general pass behind this launch places its particles
    __shared__ AppShared sh;
    if (threadIdx.x < APP_TAB) { sh.cell[threadIdx.x] = 0xffffffffu; sh.cnt[threadIdx.x] = 0u; }
    __syncthreads();
    const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
    const size_t base = beg + ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
    AppOut ao;
    ao.cell = 0u;
    bool have = false;
    if (base < end)
      have = sph_force_chunk<LMAX, 1, APP>(S, X, Y, Z, base, end, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick, assign, work,
                                            nwork, nullptr, nk_dtk, nk_dtd, store_v, ~0ull, &ao);
    // (reserving BEFORE the evaluation, by a next cell predicted from the last step's acceleration -- so that the atomics'
    // round trip would pass under the evaluation -- was built and measured: 7.0 against 5.9 ms for this pass at 1e8; the
    // early barrier and the registers held across the evaluation cost more than the round trip, profiles/r06_append_ab.txt)
#if defined(APP_EXPT) && APP_EXPT == 1
    // timing experiment: the wider stores alone -- own slot, no table, no barrier, no atomic (results are NOT usable)
    if (have) { AppDev q = app; uint32_t zb[2] = {0u, 0xffffffffu}; (void)zb; app_store_at(q, base + (threadIdx.x & 63), ao, base + (threadIdx.x & 63)); }
    return;
#endif
    int h;
    uint32_t rank;
    app_block_register(sh, have, ao.cell, h, rank);
    __syncthreads();
    app_block_reserve(sh, app);
    __syncthreads();
    if (have) {
      const uint32_t r = h >= 0 ? sh.base[h] + rank : atomicAdd(&app.cursor[ao.cell], 1u);
#if defined(APP_EXPT) && APP_EXPT == 2
      // timing experiment: table, barriers and atomics as they are, the stores at the lane's own slot
      (void)r; app_store_at(app, base + (threadIdx.x & 63), ao, base + (threadIdx.x & 63));
#else
      app_store<APP == 2 ? 1 : 0>(app, ao.cell, r, ao, base + (threadIdx.x & 63));
#endif
    }
    return;
  }
  if constexpr (MODE == 3) {
    // the literal pass: a fixed small grid walks the list (its length is only known on the device)
    const size_t n = S.lit_list[0] < S.lit_cap ? S.lit_list[0] : S.lit_cap;
    for (size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < n; base += (size_t)gridDim.x * 256)
      sph_force_chunk<LMAX, 3>(S, X, Y, Z, base, n, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick, assign, nullptr, nullptr,
                               key_out, nk_dtk, nk_dtd, store_v);
    return;
  }
  if constexpr (MODE == 0) {
    // the general pass behind a fast pass: how many work items there are is only known on the device, so a FIXED grid
    // walks the list (a grid over every possible item is ~4e5 blocks at 1e8 particles that find nothing and leave: 0.1 ms)
    if (work != nullptr) {
      const size_t nw = *nwork;
      const size_t end_ = lev_off[lev_hi + 1];
      for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (size_t)gridDim.x * 4) {
        const size_t base = work[SPH_WORK_STRIDE * w];
        const unsigned long long mask = (unsigned long long)work[SPH_WORK_STRIDE * w + 1] |
                                        ((unsigned long long)work[SPH_WORK_STRIDE * w + 2] << 32);
        if constexpr (APP) {
          AppOut ao;
          ao.cell = 0u;
          const bool have = sph_force_chunk<LMAX, 0, 1>(S, X, Y, Z, base, end_, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick,
                                                           assign, work, nwork, nullptr, nk_dtk, nk_dtd, store_v, mask, &ao);
          app_wave_store(app, have, ao, base + (threadIdx.x & 63));
        } else
        sph_force_chunk<LMAX, 0>(S, X, Y, Z, base, end_, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick, assign, work, nwork,
                                 key_out, nk_dtk, nk_dtd, store_v, mask);
      }
      return;
    }
  }
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  if constexpr (MODE != 0) {
    // SPH_FORCE_CHUNKS consecutive chunks per wave (rolled loop: the literal recurrence constants
    // are rematerialised, not hoisted) amortise the wave launch and its first-load latency.
#pragma unroll 1
    for (int c = 0; c < SPH_FORCE_CHUNKS; c++) {
      const size_t base = beg + (((size_t)blockIdx.x * SPH_FORCE_CHUNKS + c) * 256 + (threadIdx.x & ~63));
      if (base >= end) return;
      sph_force_chunk<LMAX, MODE>(S, X, Y, Z, base, end, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick,
                                  assign, work, nwork, key_out, nk_dtk, nk_dtd, store_v);
    }
  } else {
    const size_t base = beg + ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
    if (base >= end) return;
    sph_force_chunk<LMAX, 0>(S, X, Y, Z, base, end, T4, AX, AY, AZ, POT, VX, VY, VZ, dt_kick,
                             assign, work, nwork, key_out, nk_dtk, nk_dtd, store_v, ~0ull);
  }
}

// ---- general evaluation with the table rows of the block's cells staged in LDS ----------------------------------
// The forces of this basis on ANOTHER component's particles (interactions, src/ComponentContainer.cc:785-853) cannot
// take the fast pass: the target is in its own basis' cell order, a wave spans several radial cells of this one, and
// the general pass above pays ~100 dependent per-lane gathers from global memory per wave for its table rows (the
// largest single launch of a two-component master step).  But a block's 256 consecutive target particles are still
// LOCAL in radius (a disk in (R, z)-cell order: ~8 cells of the halo's radial grid), so the block copies the rows of
// its cell range [cmin, cmax] into LDS once -- coalesced -- and every lane reads its own cell's rows from there, with
// the FAST pass' arithmetic (shared reciprocals, factored weights: half the instructions of the general evaluation;
// the same operations as a cell-uniform wave of the fast pass performs, so a particle gets the bits it would get
// there).  Lanes on the polar axis or beyond rmax are left on a work list for the general pass; blocks whose range
// does not fit (nstage rows) read their rows from global memory with the same arithmetic.  tqs: LDS row stride in doubles (tq padded to 2 mod 16: consecutive cells
// start four banks apart).
typedef const __attribute__((address_space(3))) double *ldsp;

#ifndef STAGED_MINB
#define STAGED_MINB 2          // (A/B: tools/build_variant_tu.sh <suffix> sph_inst_L6 "-DSTAGED_MINB=5")
#endif
template <int LMAX>
__global__ void __launch_bounds__(256, STAGED_MINB)
k_sph_force_staged(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                   const double *__restrict__ Z, const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi,
                   const double *__restrict__ T4, double *__restrict__ AX, double *__restrict__ AY,
                   double *__restrict__ AZ, double *__restrict__ POT, double *__restrict__ VX,
                   double *__restrict__ VY, double *__restrict__ VZ, int assign, int nstage, int tqs,
                   uint32_t *__restrict__ work, uint32_t *__restrict__ nwork)
{
  extern __shared__ __attribute__((aligned(16))) double stage[];
  __shared__ int s_min, s_max;
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const size_t base0 = beg + (size_t)blockIdx.x * 256;
  if (base0 >= end) return;
  if (threadIdx.x == 0) { s_min = 0x7fffffff; s_max = -1; }
  const int lane = threadIdx.x & 63;
  const size_t i = base0 + threadIdx.x;
  const bool valid = i < end;
  double xx = 1, yy = 0, zz = 0, px = 0, py = 0, pz = 0;
  if (valid) {
    px = X[i]; py = Y[i]; pz = Z[i];
    xx = px - S.cx; yy = py - S.cy; zz = pz - S.cz;
  }
  const double fac = xx * xx + yy * yy;
  const size_t tq = (size_t)4 * S.trows;
  // the fast pass' prologue (sph_force_chunk<LMAX, 1>): shared reciprocals, cell, weights
  double g, y, R, iR;
  sqrt_rsqrt(fac + zz * zz, g, y);
  const double r = g + S.dsmall;
  const double ir = rcp_refine(r, y);
  double costh = zz * ir;
  sqrt_rsqrt(fac, R, iR);
  const double iR2 = iR * iR;
  const double cphi = xx * iR, sphi = yy * iR;
  double sinth = R * ir;
  const double xi = sph_r_to_xi_rcp(S, r * S.inv_scale);
  const bool special = (r > S.rmax && !S.no_exterior) || !(fac > SPH_POLAR_FAC * (r * r)) || !(fac > DSMALL) || r < SPH_TINY_R * S.dsmall ||
                       xi < S.lit_xlo || xi > S.lit_xhi;
  const int idx = sph_cell(S, xi);
  const double ffac = sph_d_xi_to_r_rcp(S, xi) * S.inv_dxi;
  double dfac = -(r * r) * iR2;
  const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
  const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const int jdx = idx < 1 ? 1 : idx;
  const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
  const bool regular = valid && !special;
  sph_polar_faithful(S, xx, yy, zz, fac, r, regular, costh, sinth, dfac);
  // the block's range of cells (regular lanes only: the special ones -- polar axis, beyond rmax -- take the general
  // evaluation on global rows below)
  int lo = regular ? idx : 0x7fffffff, hi = regular ? idx : -1;
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
  }
  __syncthreads();
  if (lane == 0 && hi >= 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
  __syncthreads();
  const int cmin = s_min, span = s_max - cmin + 1;
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  if (span >= 1 && span <= nstage) {            // block-uniform
    const double *src = T4 + (size_t)cmin * tq;
    const int total = span * (int)tq;
    for (int t = threadIdx.x; t < total; t += 256) {
      const int c = t / (int)tq, k = t - c * (int)tq;
      stage[c * tqs + k] = src[t];
    }
    __syncthreads();
    ldsp t4 = (ldsp)stage + (size_t)((regular ? idx : cmin) - cmin) * tqs;
    o = sph_field_fast_ptr<LMAX>(t4, costh, sinth, cphi, sphi, x2, pf);
  } else if (span >= 1) {
    const double *t4 = T4 + (size_t)(regular ? idx : cmin) * tq;
    o = sph_field_fast_ptr<LMAX>(t4, costh, sinth, cphi, sphi, x2, pf);
  }
  if (regular)
    sph_force_finish<true>(S, o, i, xx, yy, zz, px, py, pz, fac, ir, iR2, P0, ffac, dfac, AX, AY, AZ, POT, VX, VY, VZ,
                           0.0, assign, nullptr, 0.0, 0.0, 1);
  // the special lanes of this wave: the general pass that follows (k_sph_force<LMAX, 0> on the work list), by lane mask
  // -- kept inline, that evaluation cost this kernel a third of its registers (159 against 126 at lmax 6, 217 at lmax 10)
  const unsigned long long sp = __ballot(valid && special);
  if (sp && lane == 0) {
    const uint32_t w = atomicAdd(nwork, 1u);
    work[SPH_WORK_STRIDE * w] = (uint32_t)(base0 + (threadIdx.x & ~63u));
    work[SPH_WORK_STRIDE * w + 1] = (uint32_t)sp;
    work[SPH_WORK_STRIDE * w + 2] = (uint32_t)(sp >> 32);
  }
}

// ---- thin active sets: straight from the basis tables ---------------------------------------------------------------
// The upper time-step levels of a block-multistep run hold a few hundred to a few thousand particles and are stepped
// 2^level times per master step.  For them the table formulation above is all fixed cost: the contraction reads every
// cell's moments, the projection rebuilds all numr rows of T4, and the evaluation of 2000 particles by 32 waves is one
// long chain of gathers -- ~25 launches of 5-35 us for particles that need microseconds of arithmetic.  These two
// kernels do what the reference's thread bodies do (src/SphericalBasis.cc:429-599, :1476-1660): per particle, straight
// from E[i][l][n] and the coefficient set, with the block as the unit of parallelism:
//   k_sph_acc_thin  : part[level][seg][row][n] += s(l,m)^-1 Yh_row(p) (a1 E[i][l][n] + a2 E[i+1][l][n])   (a1, a2 as in
//                     sph_acc_input: -4 pi m P0 x1|x2), in the layout k_sph_contract leaves its partial sums in, so the
//                     same k_sph_sum_parts / k_sph_sum_combine finish the job (N/L swap, combined set);
//   k_sph_force_thin: the rows of T4 for THE CELLS ITS PARTICLES SIT IN are projected into LDS by the whole block (the
//                     arithmetic of k_sph_project + k_sph_project4 through the shared sph_G_row / sph_t4_entry: the same
//                     bits as the global table would hold), then evaluated by the staged kernel's code.
// No moments, no contraction, no projection launch, no global T4: exp_amd_force::proj_dirty simply stays set.
#ifndef CSEG
#define CSEG 32                // segments of the contraction's partial sums (sph.hip)
#endif

__host__ __device__ __forceinline__ int sph_l_of_row(int row)
{
  int l = 0;
  while ((l + 1) * (l + 1) <= row) l++;
  return l;
}

// G[i][row] = sum_n E[i][l][n] c[row][n], ascending n, one fma per term (k_sph_project and k_sph_force_thin)
template <class CP>
__device__ __forceinline__ double sph_G_row(const double *__restrict__ e, CP c, int nmax)
{
  double s = 0.0;
  for (int n = 0; n < nmax; n++) s = fma(e[n], c[n], s);
  return s;
}

// one slot of T4 from G at the cell's two nodes and H = p0 G at the three nodes of the force stencil
// (k_sph_project4 and k_sph_force_thin)
__device__ __forceinline__ void sph_t4_entry(double sc, double g0, double g1, double h0, double h1, double h2,
                                             double &t0, double &t1, double &t2, double &t3)
{
  t0 = sc * g0;
  t1 = sc * (g1 - g0);
  t2 = sc * (0.5 * (h2 - h0));
  t3 = sc * ((h0 - 2.0 * h1) + h2);
}

#define SPH_THIN_TP_MAX 64      // particles per tile: the first wave of the block owns them

template <int LMAX>
__global__ void __launch_bounds__(256)
k_sph_force_thin(SphDev S, const double *__restrict__ X, const double *__restrict__ Y,
                 const double *__restrict__ Z, const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi,
                 const double *__restrict__ coef, const int *__restrict__ rowmap, const double *__restrict__ tscale,
                 double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT,
                 double *__restrict__ VX, double *__restrict__ VY, double *__restrict__ VZ, int assign, int tp, int tqs)
{
  extern __shared__ __attribute__((aligned(16))) double thin_lds[];
  __shared__ int s_cell[SPH_THIN_TP_MAX];
  const int ncoef = S.nrows * S.nmax;
  const int lsn = (S.lmax + 1) * S.nmax;
  double *s_coef = thin_lds;                                  // [nrows][nmax]
  double *stage = thin_lds + ((ncoef + 1) & ~1);              // [tp][tqs]: the T4 rows of each particle's cell
  double *s_E = stage + (size_t)tp * tqs;                     // [tp][3][lsn]: E at the three nodes of its force stencil
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  if (beg + (size_t)blockIdx.x * tp >= end) return;
  // (blockDim.x threads, 256 by default; EXP_AMD_THIN_NT=64 -- the 232 VGPRs
  // allow two waves per SIMD, i.e. two 256-thread blocks per CU but eight 64-thread ones)
  const int NT = blockDim.x;
  for (int k = threadIdx.x; k < ncoef; k += NT) s_coef[k] = coef[k];
  const int t = threadIdx.x;
  for (size_t base = beg + (size_t)blockIdx.x * tp; base < end; base += (size_t)gridDim.x * tp) {
    // ---- the prologue of k_sph_force_staged, one particle per lane of the first wave
    const size_t i = base + t;
    const bool valid = t < tp && i < end;
    double xx = 1, yy = 0, zz = 0, px = 0, py = 0, pz = 0;
    if (valid) {
      px = X[i]; py = Y[i]; pz = Z[i];
      xx = px - S.cx; yy = py - S.cy; zz = pz - S.cz;
    }
    const double fac = xx * xx + yy * yy;
    double g, y, R, iR;
    sqrt_rsqrt(fac + zz * zz, g, y);
    const double r = g + S.dsmall;
    const double ir = rcp_refine(r, y);
    double costh = zz * ir;
    sqrt_rsqrt(fac, R, iR);
    const double iR2 = iR * iR;
    const double cphi = xx * iR, sphi = yy * iR;
    double sinth = R * ir;
    const double xi = sph_r_to_xi_rcp(S, r * S.inv_scale);
    const bool special = (r > S.rmax && !S.no_exterior) || !(fac > SPH_POLAR_FAC * (r * r)) || !(fac > DSMALL) || r < SPH_TINY_R * S.dsmall;
    const int idx = sph_cell(S, xi);
    const double ffac = sph_d_xi_to_r_rcp(S, xi) * S.inv_dxi;
    double dfac = -(r * r) * iR2;
    const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
    const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
    const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
    const int jdx = idx < 1 ? 1 : idx;
    const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
    const bool regular = valid && !special;
    sph_polar_faithful(S, xx, yy, zz, fac, r, regular, costh, sinth, dfac);
    // the general evaluation's own radius and cell for the special lanes (sph_force_chunk<LMAX, 0>)
    const double facg = sq_sum2_lit(xx, yy);
    double rg = sqrt(sq_add_lit(facg, zz)) + S.dsmall;
    const double r0 = rg;
    bool ioff = false;
    if (rg > S.rmax && !S.no_exterior) { ioff = true; rg = S.rmax; }
    const double xig = sph_r_to_xi(S, rg / S.scale);
    const int idg = sph_cell(S, xig);
    if (t < SPH_THIN_TP_MAX) s_cell[t] = !valid ? -1 : regular ? idx : idg;
    __syncthreads();
    // ---- E[j-1 .. j+1][l][n] of every particle's stencil: one contiguous stretch of 3 lsn doubles each, copied with
    // four loads in flight per thread (a dependent load per term of the sums below is what this kernel would
    // otherwise consist of)
    {
      const int e3 = 3 * lsn, total = tp * e3;
      for (int it0 = threadIdx.x; it0 < total; it0 += 4 * NT) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int it = it0 + u * NT;
          v[u] = 0.0;
          if (it < total) {
            const int p = it / e3, cell = s_cell[p];
            if (cell >= 0) v[u] = S.E[(size_t)((cell < 1 ? 1 : cell) - 1) * lsn + (it - p * e3)];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (it0 + u * NT < total) s_E[it0 + u * NT] = v[u];
      }
    }
    __syncthreads();
    // ---- the T4 rows of those cells, projected by the whole block: slot q of particle p
    for (int it = threadIdx.x; it < tp * S.trows; it += NT) {
      const int p = it / S.trows, q = it - p * S.trows;
      const int cell = s_cell[p];
      if (cell < 0) continue;
      double *o = stage + (size_t)p * tqs + 4 * q;
      const int row = rowmap[q];
      if (row < 0) { o[0] = o[1] = o[2] = o[3] = 0.0; continue; }
      const int l = sph_l_of_row(row);
      const int j = cell < 1 ? 1 : cell;                   // force stencil j-1, j, j+1 (k_sph_project4)
      const double *e = s_E + (size_t)p * 3 * lsn + l * S.nmax;
      const double *c = s_coef + (size_t)row * S.nmax;
      // (sph_G_row for the three nodes at once, six orders' LDS reads issued before their fmas: same terms, same order)
      double ga = 0.0, gb = 0.0, gc = 0.0;
      for (int n0 = 0; n0 < S.nmax; n0 += 6) {
        double cv[6], ea[6], eb[6], ec[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const bool in = n0 + u < S.nmax;
          cv[u] = in ? c[n0 + u] : 0.0;
          ea[u] = in ? e[n0 + u] : 0.0;
          eb[u] = in ? e[lsn + n0 + u] : 0.0;
          ec[u] = in ? e[2 * lsn + n0 + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
          if (n0 + u < S.nmax) { ga = fma(ea[u], cv[u], ga); gb = fma(eb[u], cv[u], gb); gc = fma(ec[u], cv[u], gc); }
      }
      const double g0 = cell < 1 ? ga : gb, g1 = cell < 1 ? gb : gc;      // G[cell], G[cell + 1]
      const double h0 = S.p0[j - 1] * ga, h1 = S.p0[j] * gb, h2 = S.p0[j + 1] * gc;
      sph_t4_entry(tscale[q], g0, g1, h0, h1, h2, o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // ---- evaluation: the fast pass' arithmetic on the lane's own rows; polar axis / beyond rmax: the general one
    if (t < tp) {
      if (regular) {
        const ForceOut o = sph_field_fast_ptr<LMAX>((ldsp)(stage + (size_t)t * tqs), costh, sinth, cphi, sphi, x2, pf);
        sph_force_finish<true>(S, o, i, xx, yy, zz, px, py, pz, fac, ir, iR2, P0, ffac, dfac, AX, AY, AZ, POT, VX, VY, VZ,
                               0.0, assign, nullptr, 0.0, 0.0, 1);
      }
      if (__any(valid && special)) {
        const double costh_g = zz / r0;
        double cphi_g, sphi_g;
        phi_trig(xx, yy, cphi_g, sphi_g);
        const double y1 = (S.xi[idg + 1] - xig) * S.inv_dxi;
        const double y2 = (xig - S.xi[idg]) * S.inv_dxi;
        const double P0g = y1 * S.p0[idg] + y2 * S.p0[idg + 1];
        const int jdg = idg < 1 ? 1 : idg;
        const double pfg = (xig - S.xi[jdg]) * S.inv_dxi;
        const double ffacg = sph_d_xi_to_r(S, xig) * S.inv_dxi;
        double xc = costh_g;
        if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
        const double dfacg = 1.0 / sq_add_lit(-1.0, xc);
        const double rr = S.rmax / r0;
        const double kappa0 = -P0g / (r0 * ffacg);
        const double *t4 = stage + (size_t)t * tqs;
        const ForceOut og = sph_field<LMAX>(S, costh_g, xc, cphi_g, sphi_g, t4, y2, pfg, ioff, rr, kappa0);
        if (valid && special)
          sph_force_finish<false>(S, og, i, xx, yy, zz, px, py, pz, facg, 1.0 / rg, 1.0 / facg, P0g, ffacg, dfacg, AX, AY,
                                  AZ, POT, VX, VY, VZ, 0.0, assign, nullptr, 0.0, 0.0, 1);
      }
    }
    __syncthreads();
  }
}

// Accumulation of a thin slot range [lev_off[lo], lev_off[hi + 1]) -- level-contiguous, in no cell order -- into
// part[level - lo][seg][row][n] (seg = blockIdx.x mod CSEG; the buffer must be zero on entry and is consumed AND cleared
// by k_sph_sum_parts / k_sph_sum_combine with clear = 1).  Window, inputs and the rescaled Legendre recurrence are those
// of the dense kernel (sph_acc_input, sph_acc_group); tpa particles per tile.
template <int LMAX>
__global__ void __launch_bounds__(256)
k_sph_acc_thin(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
               const double *__restrict__ M, const uint32_t *__restrict__ lev_off, int lo, int hi,
               const double *__restrict__ wscale, double *__restrict__ part, unsigned long long *__restrict__ used_out,
               int tpa, ThinAdv adv)
{
  extern __shared__ __attribute__((aligned(16))) double thin_lds[];
  __shared__ int s_idx[SPH_THIN_TP_MAX], s_lev[SPH_THIN_TP_MAX];
  __shared__ double s_a1[SPH_THIN_TP_MAX], s_a2[SPH_THIN_TP_MAX];
  const int nrows = S.nrows, lsn = (S.lmax + 1) * S.nmax, ncoef = nrows * S.nmax;
  double *yv = thin_lds;                                      // [tpa][nrows]: Yh_row of each particle
  double *pe = thin_lds + (((size_t)tpa * nrows + 1) & ~(size_t)1);   // [tpa][lsn]: a1 E[i] + a2 E[i+1]
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  const int seg = blockIdx.x % CSEG;
  const int t = threadIdx.x;
  const bool um = S.umass != 0.0;
  for (size_t base = beg + (size_t)blockIdx.x * tpa; base < end; base += (size_t)gridDim.x * tpa) {
    const int np = (int)((end - base) < (size_t)tpa ? (end - base) : (size_t)tpa);
    if (t < tpa) {
      const size_t i = base + t;
      const bool valid = i < end;
      double x = 0, y = 0, z = 0, m = 0;
      if (valid) {
        if (adv.on) thin_advance(adv, i, x, y, z); else { x = X[i]; y = Y[i]; z = Z[i]; }
        m = um ? S.umass : M[i];
      }
      const AccIn in = sph_acc_input<false>(S, (ldp) nullptr, x, y, z, m, valid);
      int lv = lo;
      while (lv < hi && i >= lev_off[lv + 1]) lv++;
      s_idx[t] = in.idx; s_lev[t] = lv; s_a1[t] = in.a1; s_a2[t] = in.a2;
      double *yr = yv + (size_t)t * nrows;
      const bool on = in.idx >= 0;
      double pmm = LC_E(0);
      double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
      static_for<0, LMAX + 1>([&](auto mc) {
        constexpr int m_ = decltype(mc)::value;
        if constexpr (m_ == 1) {
          pmm *= LC_E(1) * in.sinth;
          cm = in.cphi; sm = in.sphi;
        } else if constexpr (m_ > 1) {
          pmm *= LC_E(m_) * in.sinth;
          const double cn = 2.0 * in.cphi * cm - cm1;
          const double sn = 2.0 * in.cphi * sm - sm1;
          cm1 = cm; sm1 = sm;
          cm = cn; sm = sn;
        }
        const bool m_on = on && (m_ == 0 || !S.M0_acc);
        double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
        static_for<m_, LMAX + 1>([&](auto lc_) {
          constexpr int l = decltype(lc_)::value;
          double plm;
          if constexpr (l == m_) plm = pmm;
          else if constexpr (l == m_ + 1) plm = LC_a(l, m_) * tprev;
          else plm = fma(LC_a(l, m_), tprev, -pl2);
          tprev = in.costh * plm;
          pl2 = pl1;
          pl1 = plm;
          constexpr int row = row_of(l, m_, 0);
          if constexpr (m_ == 0) yr[row] = m_on ? plm : 0.0;
          else { yr[row] = m_on ? plm * cm : 0.0; yr[row + 1] = m_on ? plm * sm : 0.0; }
        });
      });
    }
    __syncthreads();
    for (int it0 = threadIdx.x; it0 < np * lsn; it0 += 4 * 256) {         // (four pairs of loads in flight per thread)
      double ea[4], eb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int it = it0 + u * 256;
        ea[u] = eb[u] = 0.0;
        if (it < np * lsn) {
          const int p = it / lsn, idx = s_idx[p];
          if (idx >= 0) { ea[u] = S.E[(size_t)idx * lsn + (it - p * lsn)]; eb[u] = S.E[(size_t)(idx + 1) * lsn + (it - p * lsn)]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int it = it0 + u * 256;
        if (it < np * lsn) { const int p = it / lsn; pe[it] = fma(s_a2[p], eb[u], s_a1[p] * ea[u]); }
      }
    }
    if (threadIdx.x == 0) {
      unsigned long long u = 0;
      for (int p = 0; p < np; p++) u += s_idx[p] >= 0 ? 1u : 0u;
      if (u) atomicAdd(used_out, u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ncoef; k += 256) {
      const int row = k / S.nmax, n = k - row * S.nmax;
      const int ln = sph_l_of_row(row) * S.nmax + n;
      const double ws = wscale[row];
      double acc = 0.0;
      int cur = s_lev[0];
      for (int p0 = 0; p0 < np; p0 += 8) {                 // (eight particles' LDS reads issued before their fmas)
        double yy_[8], pp_[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool in = p0 + u < np;
          yy_[u] = in ? yv[(size_t)(p0 + u) * nrows + row] : 0.0;
          pp_[u] = in ? pe[(size_t)(p0 + u) * lsn + ln] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (p0 + u < np) {
            const int lv = s_lev[p0 + u];
            if (lv != cur) {
              if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(cur - lo) * CSEG + seg) * ncoef + k, acc * ws);
              acc = 0.0;
              cur = lv;
            }
            acc = fma(yy_[u], pp_[u], acc);
          }
        }
      }
      if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(cur - lo) * CSEG + seg) * ncoef + k, acc * ws);
    }
    __syncthreads();
  }
}

// Level-change differencing of FEW movers (multistep_update, src/SphericalBasis.cc:1156-1228), direct: the tiles run
// over the list of movers (k_mover_list); a mover adds its contribution to the set of its proposed level and takes it
// out of its level's set (levels >= mfirst only).  The sums go to part[level - mfirst][seg][row][n] -- one atomic per
// (tile, level, coefficient) -- and k_sph_sum_parts adds them to expcoefN (clear = 1).  Replaces the staged moment path
// (k_sph_mstep_update<L, true> + k_mstep_apply + a contraction over every cell of every level: three launches, 30 us).
template <int LMAX>
__global__ void __launch_bounds__(256)
k_sph_diff_thin(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                const double *__restrict__ M, const uint32_t *__restrict__ list, const uint32_t *__restrict__ cnt,
                const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev, int mfirst, int nlev_out,
                const double *__restrict__ wscale, double *__restrict__ part, int tpa)
{
  extern __shared__ __attribute__((aligned(16))) double thin_lds[];
  __shared__ int s_idx[SPH_THIN_TP_MAX], s_from[SPH_THIN_TP_MAX], s_to[SPH_THIN_TP_MAX];
  __shared__ double s_a1[SPH_THIN_TP_MAX], s_a2[SPH_THIN_TP_MAX];
  const int nrows = S.nrows, lsn = (S.lmax + 1) * S.nmax, ncoef = nrows * S.nmax;
  double *yv = thin_lds;
  double *pe = thin_lds + (((size_t)tpa * nrows + 1) & ~(size_t)1);
  const size_t count = cnt[1];
  const int seg = blockIdx.x % CSEG;
  const int t = threadIdx.x;
  const bool um = S.umass != 0.0;
  for (size_t base = (size_t)blockIdx.x * tpa; base < count; base += (size_t)gridDim.x * tpa) {
    const int np = (int)((count - base) < (size_t)tpa ? (count - base) : (size_t)tpa);
    if (t < tpa) {
      const bool valid = t < np;
      double x = 0, y = 0, z = 0, m = 0;
      int from = -1, to = -1;
      if (valid) {
        const uint32_t i = list[base + t];
        x = X[i]; y = Y[i]; z = Z[i]; m = um ? S.umass : M[i];
        from = lev[i]; to = newlev[i];
      }
      const AccIn in = sph_acc_input<true>(S, (ldp) nullptr, x, y, z, m, valid);      // (window r < rmax, :1183)
      const bool on = in.idx >= 0;
      s_idx[t] = in.idx; s_a1[t] = in.a1; s_a2[t] = in.a2;
      s_to[t] = on ? to : -1;
      s_from[t] = (on && from >= mfirst) ? from : -1;          // levels below mfirst[mdrft] are not updated
      double *yr = yv + (size_t)t * nrows;
      double pmm = LC_E(0);
      double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
      static_for<0, LMAX + 1>([&](auto mc) {
        constexpr int m_ = decltype(mc)::value;
        if constexpr (m_ == 1) {
          pmm *= LC_E(1) * in.sinth;
          cm = in.cphi; sm = in.sphi;
        } else if constexpr (m_ > 1) {
          pmm *= LC_E(m_) * in.sinth;
          const double cn = 2.0 * in.cphi * cm - cm1;
          const double sn = 2.0 * in.cphi * sm - sm1;
          cm1 = cm; sm1 = sm;
          cm = cn; sm = sn;
        }
        double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
        static_for<m_, LMAX + 1>([&](auto lc_) {
          constexpr int l = decltype(lc_)::value;
          double plm;
          if constexpr (l == m_) plm = pmm;
          else if constexpr (l == m_ + 1) plm = LC_a(l, m_) * tprev;
          else plm = fma(LC_a(l, m_), tprev, -pl2);
          tprev = in.costh * plm;
          pl2 = pl1;
          pl1 = plm;
          constexpr int row = row_of(l, m_, 0);
          if constexpr (m_ == 0) yr[row] = on ? plm : 0.0;
          else { yr[row] = on ? plm * cm : 0.0; yr[row + 1] = on ? plm * sm : 0.0; }
        });
      });
    }
    __syncthreads();
    for (int it0 = threadIdx.x; it0 < np * lsn; it0 += 4 * 256) {
      double ea[4], eb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int it = it0 + u * 256;
        ea[u] = eb[u] = 0.0;
        if (it < np * lsn) {
          const int p = it / lsn, idx = s_idx[p];
          if (idx >= 0) { ea[u] = S.E[(size_t)idx * lsn + (it - p * lsn)]; eb[u] = S.E[(size_t)(idx + 1) * lsn + (it - p * lsn)]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int it = it0 + u * 256;
        if (it < np * lsn) { const int p = it / lsn; pe[it] = fma(s_a2[p], eb[u], s_a1[p] * ea[u]); }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ncoef; k += 256) {
      const int row = k / S.nmax, n = k - row * S.nmax;
      const int ln = sph_l_of_row(row) * S.nmax + n;
      const double ws = wscale[row];
      double v[8];                   // this coefficient's term of each mover (tpa <= 8)
#pragma unroll
      for (int p = 0; p < 8; p++) v[p] = p < np ? yv[(size_t)p * nrows + row] * pe[(size_t)p * lsn + ln] : 0.0;
      for (int L = 0; L < nlev_out; L++) {
        const int level = mfirst + L;
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < 8; p++)
          if (p < np) acc += (s_to[p] == level ? v[p] : 0.0) - (s_from[p] == level ? v[p] : 0.0);
        if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)L * CSEG + seg) * ncoef + k, acc * ws);
      }
    }
    __syncthreads();
  }
}

// ---- per-LMAX launchers (one translation unit per LMAX: sph_inst.hip -DSPH_L=k) -----------------

struct SphAccArgs {
  SphDev S;
  const double *X, *Y, *Z, *M;
  const uint32_t *lev_off;
  int lo, hi;               // levels lo..hi, each with its own chunking (hi == lo: the classic launch)
  double *W;
  unsigned long long *used;
  size_t n;                 // population of level lo when hi == lo ...
  hipStream_t stream;
  int multilevel;           // the range is one level of many (short chunks allowed)
  const uint32_t *counts = nullptr;   // ... else counts[j] = population of level lo + j
  int wlevels = 0;          // 1: W[level][cell][row][2] (one moment buffer per level)
  // LIST mode (level-change differencing of many movers): lev_off -> {0, list length}, n = the host's count
  const uint32_t *list = nullptr;
  const uint8_t *lev = nullptr, *newlev = nullptr;
  int mfirst = 0, nslices = 1;        // nslices = multistep + 2 (AccList::per_level) or 2
};

struct SphForceArgs {
  SphDev S;
  const double *X, *Y, *Z;
  const uint32_t *lev_off;
  int lo, hi;
  const double *T4;
  double *AX, *AY, *AZ, *POT, *VX, *VY, *VZ;
  double dt_kick;
  int assign;
  size_t n;
  unsigned grid;
  hipStream_t stream;
  uint32_t *work, *nwork;   // slow-path work list (first slot of each deferred wave) + count
  int all_slow;             // skip the fast pass (every wave takes the general evaluation)
  exp_amd_ctx *ctx;         // for the per-launch profiling scopes
  uint32_t *key_out;        // next step's sort keys (nullptr: not wanted)
  double nk_dtk, nk_dtd;    // ... for that step's kick and drift
  int store_v;              // 0: the half-kick is deferred, v is left as it is; 1: v + a dt_kick; 2: that plus the
                            // next step's opening half-kick (needs key_out)
  uint32_t *nwork_next = nullptr;   // the counter the next launch will use (cleared by this one's general pass)
  int waterfall = 0;        // fast pass as a waterfall over each wave's radial cells (MODE 2)
  int stage_rows = 0;       // all_slow launches: cells whose table rows a block stages in LDS (0: global gathers)
  const AppDev *app = nullptr;      // the append step (fused_step_append): place the results in the other buffer set
};

struct SphUpdArgs {
  SphDev S;
  const double *X, *Y, *Z, *M;
  const uint8_t *lev, *newlev;
  const uint32_t *lev_off;
  int first, last, mfirst;
  double *Wd;
  size_t n;
  hipStream_t stream;
  int plain = 0;                        // 1: accumulate every particle of the range into Wd[its level]
  unsigned long long *used = nullptr;   // ... and count those inside the window
  const uint32_t *list = nullptr;       // slots of the movers: lev_off = {0, count}, n = the expected count
  double *stage = nullptr;              // ... staged: values [mover][nrows][2] and W offsets, applied by k_mstep_apply
  int2 *keys = nullptr;
};

struct SphThinForceArgs {
  SphDev S;
  const double *X, *Y, *Z;
  const uint32_t *lev_off;
  int lo, hi;
  const double *coef;       // the coefficient set evaluated (expcoef), [nrows][nmax]
  const int *rowmap;
  const double *tscale;
  double *AX, *AY, *AZ, *POT, *VX, *VY, *VZ;
  int assign;
  size_t n;                 // population of the range (sizes the grid; the kernel strides over whatever is there)
  hipStream_t stream;
};
struct SphThinAccArgs {
  SphDev S;
  const double *X, *Y, *Z, *M;
  const uint32_t *lev_off;
  int lo, hi;
  const double *wscale;
  double *part;
  unsigned long long *used;
  size_t n;
  hipStream_t stream;
  ThinAdv adv;                      // (k_sph_acc_thin only: the advance of the range folded in)
};
struct SphThinDiffArgs {
  SphDev S;
  const double *X, *Y, *Z, *M;
  const uint32_t *list, *cnt;       // slots of the movers, {0, count}
  const uint8_t *lev, *newlev;
  int mfirst, nlev_out;
  const double *wscale;
  double *part;
  size_t n;                         // the host's count of movers (sizes the grid)
  hipStream_t stream;
};
typedef void (*sph_thin_diff_launcher)(const SphThinDiffArgs &);
typedef void (*sph_thin_force_launcher)(const SphThinForceArgs &);
typedef void (*sph_thin_acc_launcher)(const SphThinAccArgs &);
typedef void (*sph_upd_launcher)(const SphUpdArgs &);
typedef void (*sph_acc_launcher)(const SphAccArgs &);
typedef void (*sph_force_launcher)(const SphForceArgs &);
