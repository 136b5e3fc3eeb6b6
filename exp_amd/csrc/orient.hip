// Orientation / expansion-centre estimator: replaces class Orient (src/Orient.cc:38-790, Orient.H)
// with its O(N) part on the device.  What the reference does per call (accumulate, :420-747):
//   1. energy of every particle, E = pot [+ v^2/2 with KE]             (accumulate_cpu :325-417)
//   2. keep the `many`+1 most bound; Ecurr = the (many+1)-th lowest energy (or the highest when the
//      component holds no more than `many` particles)                     (:452-483)
//   3. over the particles with E < Ecurr: sum m (x - centre) x v, m x, m  (:485-555)
//   4. push (time, L/M) and (time, R/M) on two short histories and take a damped linear
//      least-squares extrapolation of each as the new axis / centre      (:557-700)
// Steps 1-3 are a selection problem over N doubles.  The reference sorts 200000-particle bunches of
// 64-byte records with thrust and walks a std::set on the host (src/cudaOrient.cu:109-199); here
// the energies become order-preserving 64-bit keys (8 B/particle, written once), the k-th smallest
// is found EXACTLY by a most-significant-digit radix select (12-bit digits, LDS histograms with
// wave-aggregated updates; the first digit is counted while the keys are formed, after the second
// the few keys that still share the threshold's prefix are copied out and finish the select, one small all-reduce of the 4096 bins per pass when the
// component is sharded over ranks -- so the threshold is the global one, where the reference's
// per-rank many/numprocs trimming is only approximately that), and one streaming pass forms the
// sums.  Step 4 is a few dozen flops on the host, restated statement for statement including the
// reference's use of the CENTRE history length in the AXIS regression (:583).
// The PseudoAccel helper (include/PseudoAccel.H: quadratic least squares over the last Naccel
// (time, centre, axis) triples -> frame acceleration, angular velocity and its rate) rides along.
// The log file and the restart from it (:84-335, :742-785) are at the end of this file.
// The EXTERNAL flag (`energy += p->potext`, :377) is accepted and adds nothing: Particle::potext only ever receives the
// potential of the External force plug-ins (Component::AddPotExt: src/HaloBulge.cc, externalShock.cc, tidalField.cc), which
// are outside this build; the cross forces between components add to `pot` in the reference as here
// (src/SphericalBasis.cc:1652, src/Cylinder.cc:1416).  Not carried over: keep == 0, whose code path in the reference
// indexes its 3-vectors out of range (:741-744).
#include "particles.h"
#include <cmath>
#include <cstring>
#include <deque>
#include <vector>
#include <array>
#include <new>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>

namespace {

constexpr int ORI_BITS = 12, ORI_BINS = 1 << ORI_BITS, ORI_TPB = 256, ORI_ITEMS = 8;
constexpr int ORI_TILE = ORI_TPB * ORI_ITEMS;
enum { ORI_AXIS = 1, ORI_CENTER = 2 };          // Orient::OrientFlags  (src/Orient.H:129)
enum { ORI_DIAG = 1, ORI_KE = 2, ORI_EXTERNAL = 4 };   // Orient::ControlFlags (:132)

// device-resident selection state
struct OriState {
  unsigned long long prefix;     // digits chosen so far (most significant first)
  unsigned long long krem;       // rank still to descend inside the chosen prefix
  unsigned long long total;      // particles over all ranks
  unsigned long long pad;
};

// order-preserving map of a double onto an unsigned 64-bit integer (-0 is folded onto +0 first)
__device__ __forceinline__ unsigned long long ord_key(double e)
{
  e = e + 0.0;
  const unsigned long long u = (unsigned long long)__double_as_longlong(e);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// energies cluster: group the lanes of a wave by digit, one LDS add per distinct digit
__device__ __forceinline__ void ori_hist_add(uint32_t *lh, unsigned digit, bool valid)
{
  const int lane = threadIdx.x & 63;
  unsigned long long rem = __ballot(valid);
  while (rem) {
    const int lead = __ffsll((long long)rem) - 1;
    const unsigned dd = (unsigned)__shfl((int)digit, lead);
    const unsigned long long mm = __ballot(valid && digit == dd);
    if (lane == lead) atomicAdd(&lh[dd], (uint32_t)__popcll(mm));
    rem &= ~mm;
  }
}

// energies -> keys, and the histogram of their most significant digit in the same pass
__global__ void __launch_bounds__(ORI_TPB)
k_orient_keys(const double *__restrict__ POT, const double *__restrict__ VX,
              const double *__restrict__ VY, const double *__restrict__ VZ, const double *__restrict__ AX,
              const double *__restrict__ AY, const double *__restrict__ AZ, double back, size_t n, int ke,
              unsigned long long *__restrict__ key, double *__restrict__ hist)
{
  // back != 0: the stored velocities carry the next step's opening half-kick (exp_amd_comp::pending_kick < 0);
  // the velocity of the step boundary is v + a * back, formed here and never stored (expamd_comp_velocity_view)
  __shared__ uint32_t lh[ORI_BINS];
  for (int b = threadIdx.x; b < ORI_BINS; b += ORI_TPB) lh[b] = 0;
  __syncthreads();
  for (size_t base = (size_t)blockIdx.x * ORI_TILE; base < n; base += (size_t)gridDim.x * ORI_TILE) {
#pragma unroll
    for (int j = 0; j < ORI_ITEMS; j++) {
      const size_t i = base + (size_t)j * ORI_TPB + threadIdx.x;
      const bool valid = i < n;
      unsigned long long k = 0;
      if (valid) {
        double e = POT[i];
        if (ke) {                  // v2 += vel[k]*vel[k]; energy += 0.5*v2 -- each product rounded (:352-359)
          double vx = VX[i], vy = VY[i], vz = VZ[i];
          if (back != 0.0) {
            vx = mul_then_add(vx, AX[i], back); vy = mul_then_add(vy, AY[i], back); vz = mul_then_add(vz, AZ[i], back);
          }
          double v2 = mul_then_add(0.0, vx, vx);
          v2 = mul_then_add(v2, vy, vy);
          v2 = mul_then_add(v2, vz, vz);
          e = mul_then_add(e, 0.5, v2);
        }
        k = ord_key(e);
        key[i] = k;
      }
      ori_hist_add(lh, (unsigned)(k >> 52), valid);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < ORI_BINS; b += ORI_TPB) {
    const uint32_t c = lh[b];
    if (c) unsafeAtomicAdd(&hist[b], (double)c);
  }
}

// after two digits (24 bits) the bin that holds the threshold is small: copy its keys out, the
// remaining 40 bits are resolved on that short list
__global__ void __launch_bounds__(ORI_TPB)
k_orient_compact(const unsigned long long *__restrict__ key, size_t n, const OriState *__restrict__ st,
                 unsigned long long *__restrict__ cand, unsigned long long *__restrict__ ncand)
{
  const unsigned long long prefix = st->prefix;         // 24 bits
  const int lane = threadIdx.x & 63;
  for (size_t base = (size_t)blockIdx.x * ORI_TILE; base < n; base += (size_t)gridDim.x * ORI_TILE) {
#pragma unroll
    for (int j = 0; j < ORI_ITEMS; j++) {
      const size_t i = base + (size_t)j * ORI_TPB + threadIdx.x;
      unsigned long long k = 0;
      bool hit = false;
      if (i < n) { k = key[i]; hit = (k >> 40) == prefix; }
      const unsigned long long mm = __ballot(hit);
      if (!mm) continue;
      unsigned long long base_slot = 0;
      const int lead = __ffsll((long long)mm) - 1;
      if (lane == lead) base_slot = atomicAdd(ncand, (unsigned long long)__popcll(mm));
      base_slot = (unsigned long long)__shfl((long long)base_slot, lead);
      if (hit) cand[base_slot + (unsigned long long)__popcll(mm & ((1ull << lane) - 1ull))] = k;
    }
  }
}

// one radix-select pass: histogram of the digit at `shift` over the keys whose higher bits equal
// the prefix chosen so far.  The bins are doubles so that the context's all-reduce can sum them.
// (n_dev != nullptr: the keys are the compacted candidates, their number lives on the device)
__global__ void __launch_bounds__(ORI_TPB)
k_orient_hist(const unsigned long long *__restrict__ key, size_t n, const unsigned long long *__restrict__ n_dev,
              const OriState *__restrict__ st, int pass, int shift, int bits, double *__restrict__ hist)
{
  __shared__ uint32_t lh[ORI_BINS];
  for (int b = threadIdx.x; b < ORI_BINS; b += ORI_TPB) lh[b] = 0;
  __syncthreads();
  if (n_dev) { const size_t nd = (size_t)*n_dev; if (nd < n) n = nd; }
  const unsigned long long prefix = st->prefix;
  const unsigned mask = (1u << bits) - 1u;
  for (size_t base = (size_t)blockIdx.x * ORI_TILE; base < n; base += (size_t)gridDim.x * ORI_TILE) {
#pragma unroll
    for (int j = 0; j < ORI_ITEMS; j++) {
      const size_t i = base + (size_t)j * ORI_TPB + threadIdx.x;
      bool valid = i < n;
      unsigned digit = 0;
      if (valid) {
        const unsigned long long k = key[i];
        valid = pass == 0 || (k >> ((shift + bits) & 63)) == prefix;
        digit = (unsigned)(k >> shift) & mask;
      }
      ori_hist_add(lh, digit, valid);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < ORI_BINS; b += ORI_TPB) {
    const uint32_t c = lh[b];
    if (c) unsafeAtomicAdd(&hist[b], (double)c);
  }
}

// descend one digit: find the bin that holds rank krem, clear the histogram for the next pass
__global__ void __launch_bounds__(64)
k_orient_pick(double *__restrict__ hist, OriState *__restrict__ st, int pass, int bits,
              unsigned long long many)
{
  __shared__ unsigned long long cum[64], sub[64], s_run, s_krem;
  __shared__ int s_blk;
  const int nb = 1 << bits, per = nb >= 64 ? nb / 64 : 1, nblk = nb / per;
  const int t = threadIdx.x;
  unsigned long long mine = 0;
  if (t < nblk)
    for (int b = t * per; b < (t + 1) * per; b++) mine += (unsigned long long)hist[b];
  cum[t] = mine;
  __syncthreads();
  if (t == 0) {
    unsigned long long krem = st->krem;
    if (pass == 0) {             // ee.size() <= many ? ee.back() : ee[many]   (:474-478)
      unsigned long long tot = 0;
      for (int k = 0; k < nblk; k++) tot += cum[k];
      st->total = tot;
      krem = tot == 0 ? 0 : (tot <= many ? tot - 1 : many);
      st->prefix = 0;
    }
    unsigned long long run = 0;
    int blk = 0;
    for (; blk < nblk - 1; blk++) {
      if (krem < run + cum[blk]) break;
      run += cum[blk];
    }
    s_blk = blk; s_run = run; s_krem = krem;
  }
  __syncthreads();
  if (t < per) sub[t] = (unsigned long long)hist[s_blk * per + t];
  __syncthreads();
  if (t == 0) {
    unsigned long long run = s_run;
    int b = 0;
    for (; b < per - 1; b++) {
      if (s_krem < run + sub[b]) break;
      run += sub[b];
    }
    st->krem = s_krem - run;
    st->prefix = (st->prefix << bits) | (unsigned long long)(s_blk * per + b);
  }
  __syncthreads();
  for (int b = t; b < ORI_BINS; b += 64) hist[b] = 0.0;
}

// sums over the particles more bound than the threshold key: {count, M, L[3], R[3]}.  No atomics:
// the expansion centre feeds back into the run, so the sums are formed in a fixed order (lane
// strides, wave shuffles, waves in order, blocks in order) and repeat bit for bit.
constexpr int ORI_SUM_BLOCKS = 1024;
__global__ void __launch_bounds__(ORI_TPB)
k_orient_sums(const unsigned long long *__restrict__ key, const double *__restrict__ M,
              const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
              const double *__restrict__ VX, const double *__restrict__ VY,
              const double *__restrict__ VZ, const double *__restrict__ AX, const double *__restrict__ AY,
              const double *__restrict__ AZ, double back, size_t n, const OriState *__restrict__ st,
              double cx, double cy, double cz, double *__restrict__ part /* [gridDim.x][8] */)
{
  __shared__ double wsum[ORI_TPB / 64][8];
  const unsigned long long thr = st->prefix;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * ORI_TPB;
  for (size_t i = (size_t)blockIdx.x * ORI_TPB + threadIdx.x; i < n; i += stride) {
    if (key[i] >= thr) continue;                     // i->E < Ecurr   (:489)
    const double m = M[i], x = X[i], y = Y[i], z = Z[i];
    double vx = VX[i], vy = VY[i], vz = VZ[i];
    if (back != 0.0) {
      vx = mul_then_add(vx, AX[i], back); vy = mul_then_add(vy, AY[i], back); vz = mul_then_add(vz, AZ[i], back);
    }
    const double px = x - cx, py = y - cy, pz = z - cz;
    v[0] += 1.0;
    v[1] += m;
    v[2] += m * (py * vz - pz * vy);                 // t.L (:378-380)
    v[3] += m * (pz * vx - px * vz);
    v[4] += m * (px * vy - py * vx);
    v[5] += m * x;                                   // t.R (:382-384)
    v[6] += m * y;
    v[7] += m * z;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    double t = v[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = 0.0;
    for (int w = 0; w < ORI_TPB / 64; w++) t += wsum[w][threadIdx.x];
    part[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
  }
}

__global__ void __launch_bounds__(64)
k_orient_final(const double *__restrict__ part, int nblk, double *__restrict__ out)
{
  // lane = (block residue, value): 8 groups of 8 values; fixed-order strided sums, then shuffles
  const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
  double t = 0.0;
  for (int b = g; b < nblk; b += 8) t += part[(size_t)b * 8 + k];
  for (int off = 8; off < 64; off <<= 1) t += __shfl_xor(t, off);
  if (g == 0) out[k] = t;
}

typedef std::array<double, 3> V3;
typedef std::pair<double, V3> DV;

// return_euler_slater (exputil/euler_slater.cc:46-76), row-major 3x3; body != 0 transposes
void euler_slater(double phi, double theta, double psi, int body, double *o)
{
  const double sph = sin(phi), cph = cos(phi), sth = sin(theta), cth = cos(theta), sps = sin(psi),
               cps = cos(psi);
  double e[9] = {-sps * sph + cth * cph * cps, sps * cph + cth * sph * cps, cps * sth,
                 -cps * sph - cth * cph * sps, cps * cph - cth * sph * sps, -sps * sth,
                 -sth * cph, -sth * sph, cth};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[3 * i + j] = body ? e[3 * j + i] : e[3 * i + j];
}

}  // namespace

// QuadLS (include/QuadLS.H:17-53): y = a x^2 + b x + c by least squares
static void quadls(const std::vector<double> &x, const std::vector<double> &y, double &a, double &b,
                   double &c)
{
  a = b = c = 0.0;
  const size_t n = x.size();
  if (n != y.size() || n == 0) return;
  double sumx = 0, sumy = 0, sumxy = 0, sumx2y = 0, sumx2 = 0, sumx3 = 0, sumx4 = 0;
  for (size_t i = 0; i < n; i++) {
    sumx += x[i];
    sumy += y[i];
    sumx2 += x[i] * x[i];
    sumxy += x[i] * y[i];
    sumx2y += x[i] * x[i] * y[i];
    sumx3 += x[i] * x[i] * x[i];
    sumx4 += x[i] * x[i] * x[i] * x[i];
  }
  const double Sxx = sumx2 - sumx * sumx / n, Sxy = sumxy - sumx * sumy / n;
  const double Sxx2 = sumx3 - sumx * sumx2 / n, Sx2y = sumx2y - sumx2 * sumy / n;
  const double Sx2x2 = sumx4 - sumx2 * sumx2 / n;
  const double denom = Sxx * Sx2x2 - Sxx2 * Sxx2;
  if (fabs(denom) > 0.0) {
    a = (Sx2y * Sxx - Sxy * Sxx2) / denom;
    b = (Sxy * Sx2x2 - Sx2y * Sxx2) / denom;
    c = (sumy - sumx2 * a - sumx * b) / n;
  }
}

struct exp_amd_orient {
  exp_amd_ctx *ctx = nullptr;
  // PseudoAccel (include/PseudoAccel.H): queue of {t, centre, axis}, the last estimates
  unsigned naccel = 0;
  std::deque<std::array<double, 7>> aq;
  double ps_accel[3] = {0, 0, 0}, ps_omega[3] = {0, 0, 0}, ps_domdt[3] = {0, 0, 0};
  int keep = 0, many = 0;
  unsigned oflags = 0, cflags = 0;
  double deltaT = 0, damp = 1;
  bool linear = false;
  V3 center{{0, 0, 0}}, center0{{0, 0, 0}}, cenvel0{{0, 0, 0}}, axis{{0, 0, 1}};
  V3 axis1{{0, 0, 0}}, center1{{0, 0, 0}};
  std::deque<DV> sumsA, sumsC;
  double lasttime = -1.7976931348623157e308;
  double body[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, orig[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  long long used = 0;
  double Ecurr = 0, sigA = 0, sigC = 0, sigCz = 0, mtot = 0;
  DevBuf<unsigned long long> keys, cand, ncand;   // all keys; those sharing the threshold's first 24 bits
  DevBuf<double> hist, sums, part;
  DevBuf<OriState> state;
  std::string logfile;                            // Orient's Logfile argument (empty: no log)
};

extern "C" int exp_amd_orient_create(exp_amd_ctx *ctx, int keep, int want, unsigned oflags,
                                     unsigned cflags, double deltaT, double damp,
                                     exp_amd_orient **out)
{
  if (!ctx || !out) return EXP_AMD_ERR_ARG;
  if (keep < 1) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "Orient: keep must be >= 1 (the keep == 0 "
                                   "branch of the reference indexes out of range, src/Orient.cc:741-744)");
  if (want < 1) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "Orient: target number must be >= 1");
  exp_amd_orient *o = new (std::nothrow) exp_amd_orient;
  if (!o) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "Orient: out of memory");
  o->ctx = ctx; o->keep = keep; o->many = want; o->oflags = oflags; o->cflags = cflags;
  o->deltaT = deltaT; o->damp = damp;
  if (hipSetDevice(ctx->device) != hipSuccess || o->hist.alloc(ORI_BINS) != hipSuccess ||
      o->sums.alloc(8) != hipSuccess || o->state.alloc(1) != hipSuccess ||
      o->part.alloc((size_t)ORI_SUM_BLOCKS * 8) != hipSuccess || o->ncand.alloc(1) != hipSuccess) {
    delete o;
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "Orient: hipMalloc failed");
  }
  *out = o;
  return EXP_AMD_OK;
}

extern "C" void exp_amd_orient_destroy(exp_amd_orient *o)
{
  if (!o) return;
  o->keys.release(); o->cand.release(); o->ncand.release(); o->hist.release(); o->sums.release();
  o->state.release(); o->part.release();
  delete o;
}

// Orient::set_center / set_cenvel / set_linear (src/Orient.H:143-160)
extern "C" int exp_amd_orient_set_center(exp_amd_orient *o, const double c[3])
{
  if (!o || !c) return EXP_AMD_ERR_ARG;
  for (int k = 0; k < 3; k++) o->center[k] = o->center0[k] = c[k];
  return EXP_AMD_OK;
}
extern "C" int exp_amd_orient_set_cenvel(exp_amd_orient *o, const double v[3])
{
  if (!o || !v) return EXP_AMD_ERR_ARG;
  for (int k = 0; k < 3; k++) o->cenvel0[k] = v[k];
  return EXP_AMD_OK;
}
extern "C" int exp_amd_orient_set_linear(exp_amd_orient *o)
{
  if (!o) return EXP_AMD_ERR_ARG;
  o->linear = true;
  return EXP_AMD_OK;
}

// steps 1-3 on the device: Ecurr, used, mtot, sum L, sum R (all ranks combined)
static int orient_select(exp_amd_orient *o, exp_amd_comp *c, double res[8], double *Ecurr)
{
  exp_amd_ctx *ctx = o->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double back = 0.0;                    // read-only view of the step-boundary velocities: the store is not touched
  { int rc_ = expamd_comp_velocity_view(c, &back); if (rc_) return rc_; }
  const size_t n = c->n;
  if (o->keys.n < n && (o->keys.alloc(n) != hipSuccess || o->cand.alloc(n) != hipSuccess))
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "Orient: hipMalloc of %zu keys failed", n);
  HIP_TRY(ctx, hipMemsetAsync(o->ncand.p, 0, sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(o->hist.p, 0, o->hist.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(o->sums.p, 0, o->sums.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(o->state.p, 0, sizeof(OriState), ctx->stream));
  const bool multi = ctx->nranks > 1 || ctx->ar_fn;
  {
    ProfScope ps(ctx, "k_orient_select");
    static const int shift[6] = {52, 40, 28, 16, 4, 0}, bits[6] = {12, 12, 12, 12, 12, 4};
    const unsigned gbig = (unsigned)(cdiv(n, ORI_TILE) < 4096 ? cdiv(n, ORI_TILE) : 4096);
    for (int p = 0; p < 6; p++) {
      if (n) {
        if (p == 0)            // keys + first digit in one pass over pot, v
          k_orient_keys<<<gbig, ORI_TPB, 0, ctx->stream>>>(c->a(A_POT), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX),
                                                         c->a(A_AY), c->a(A_AZ), back, n,
                                                         (o->cflags & ORI_KE) ? 1 : 0, o->keys.p, o->hist.p);
        else if (p == 1)
          k_orient_hist<<<gbig, ORI_TPB, 0, ctx->stream>>>(o->keys.p, n, nullptr, o->state.p, p, shift[p],
                                                         bits[p], o->hist.p);
        else {                 // digits 3..6 on the compacted candidates (this rank's share)
          if (p == 2)
            k_orient_compact<<<gbig, ORI_TPB, 0, ctx->stream>>>(o->keys.p, n, o->state.p, o->cand.p,
                                                              o->ncand.p);
          k_orient_hist<<<256, ORI_TPB, 0, ctx->stream>>>(o->cand.p, n, o->ncand.p, o->state.p, p, shift[p],
                                                         bits[p], o->hist.p);
        }
      }
      if (multi) { int rc = expamd_allreduce(ctx, o->hist.p, ORI_BINS); if (rc) return rc; }
      k_orient_pick<<<1, 64, 0, ctx->stream>>>(o->hist.p, o->state.p, p, bits[p],
                                               (unsigned long long)o->many);
    }
    if (n) {
      unsigned grid = cdiv(n, ORI_TPB * 16);
      if (grid > (unsigned)ORI_SUM_BLOCKS) grid = ORI_SUM_BLOCKS;
      k_orient_sums<<<grid, ORI_TPB, 0, ctx->stream>>>(o->keys.p, c->a(A_M), c->a(A_X), c->a(A_Y),
                                                       c->a(A_Z), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX),
                                                       c->a(A_AY), c->a(A_AZ), back, n, o->state.p, o->center[0], o->center[1],
                                                       o->center[2], o->part.p);
      k_orient_final<<<1, 64, 0, ctx->stream>>>(o->part.p, (int)grid, o->sums.p);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  if (multi) { int rc = expamd_allreduce(ctx, o->sums.p, 8); if (rc) return rc; }
  OriState st;
  HIP_TRY(ctx, hipMemcpyAsync(res, o->sums.p, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&st, o->state.p, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long u = st.prefix;                  // undo ord_key
  u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  double e;
  memcpy(&e, &u, sizeof(e));
  *Ecurr = st.total ? e : 0.0;
  return EXP_AMD_OK;
}

// damped least-squares extrapolation of a (time, vector) history (:576-604, :620-676).  N is passed
// in because the AXIS branch of the reference uses the CENTRE history's length.
static void regress(const std::deque<DV> &h, int N, double damp, double time, V3 &val, double &sig,
                    double *sigz)
{
  double sumX = 0, sumX2 = 0;
  V3 sumY{{0, 0, 0}}, sumXY{{0, 0, 0}};
  for (const DV &j : h) {
    const double x = j.first;
    sumX += x;
    sumX2 += x * x;
    for (int k = 0; k < 3; k++) { sumY[k] += j.second[k]; sumXY[k] += j.second[k] * x; }
  }
  V3 slope, icpt;
  const double den = sumX2 * N - sumX * sumX;
  for (int k = 0; k < 3; k++) {
    slope[k] = (sumXY[k] * N - sumX * sumY[k]) / den;
    icpt[k] = (sumX2 * sumY[k] - sumX * sumXY[k]) / den;
  }
  const double tt = damp * time + (1.0 - damp) * h.front().first;
  for (int k = 0; k < 3; k++) val[k] = icpt[k] + slope[k] * tt;
  sig = 0.0;
  if (sigz) *sigz = 0.0;
  int i = 0;
  for (const DV &j : h) {
    double s = 0;
    for (int k = 0; k < 3; k++) {
      const double d = j.second[k] - icpt[k] - slope[k] * j.first;
      s += d * d;
      if (k == 2 && sigz) *sigz += d * d;
    }
    sig += s;
    i++;
  }
  sig /= i;
  if (sigz) *sigz /= i;
}

// Orient::accumulate(time, c) (src/Orient.cc:420-747); dtime is the reference's global time step
// (centre drift `center0 += cenvel0*dtime`, :433, :716)
extern "C" int exp_amd_orient_accumulate(exp_amd_orient *o, double time, double dtime, exp_amd_comp *c)
{
  if (!o || !c) return EXP_AMD_ERR_ARG;
  if (fabs(o->lasttime - time) < 1.0e-12) return EXP_AMD_OK;      // no duplicate entry (:423)
  if (time - o->deltaT - o->lasttime < 0.0) return EXP_AMD_OK;    // spaced by deltaT (:425)
  o->lasttime = time;
  if (o->linear) {
    o->center = o->center0;
    for (int k = 0; k < 3; k++) o->center0[k] += o->cenvel0[k] * dtime;
    return EXP_AMD_OK;
  }
  double r[8];
  int rc = orient_select(o, c, r, &o->Ecurr);
  if (rc) return rc;
  o->used = (long long)r[0];
  o->mtot = r[1];
  for (int k = 0; k < 3; k++) { o->axis1[k] = r[2 + k]; o->center1[k] = r[5 + k]; }
  if (o->mtot > 0.0) {
    for (int k = 0; k < 3; k++) { o->axis1[k] /= o->mtot; o->center1[k] /= o->mtot; }
    if (o->oflags & ORI_AXIS) o->sumsA.push_back(DV(time, o->axis1));
    if (o->oflags & ORI_CENTER) o->sumsC.push_back(DV(time, o->center1));
  }
  if ((int)o->sumsA.size() > o->keep + 1) {
    o->sumsA.pop_front();
    regress(o->sumsA, (int)o->sumsC.size(), o->damp, time, o->axis, o->sigA, nullptr);
    const double phi = atan2(o->axis[1], o->axis[0]);
    const double theta = -acos(o->axis[2] / sqrt(o->axis[0] * o->axis[0] + o->axis[1] * o->axis[1] +
                                                 o->axis[2] * o->axis[2]));
    euler_slater(phi, theta, 0.0, 0, o->body);
    euler_slater(phi, theta, 0.0, 1, o->orig);
  }
  if (o->sumsC.size() > 1) {
    if ((int)o->sumsC.size() > o->keep + 1) o->sumsC.pop_front();
    regress(o->sumsC, (int)o->sumsC.size(), o->damp, time, o->center, o->sigC, &o->sigCz);
  }
  if (o->keep > 1) {
    if (o->sumsC.size() > 1) {
      double factor = (double)((int)o->sumsC.size() - o->keep) / o->keep;
      factor = factor * factor;
      for (int k = 0; k < 3; k++) o->center[k] = o->center0[k] * factor + o->center[k] * (1.0 - factor);
    } else
      o->center = o->center0;
  } else
    o->center = o->center1;
  // pseudo-acceleration estimator (:709-713)
  if (o->naccel) {
    o->aq.push_back({time, o->center1[0], o->center1[1], o->center1[2], o->axis1[0], o->axis1[1], o->axis1[2]});
    if (o->aq.size() > o->naccel) o->aq.pop_front();
  }
  for (int k = 0; k < 3; k++) o->center0[k] += o->cenvel0[k] * dtime;
  return EXP_AMD_OK;
}

// Orient's Naccel constructor argument: length of the PseudoAccel queue (0: none)
extern "C" int exp_amd_orient_set_naccel(exp_amd_orient *o, int naccel)
{
  if (!o || naccel < 0) return EXP_AMD_ERR_ARG;
  o->naccel = (unsigned)naccel;
  o->aq.clear();
  return EXP_AMD_OK;
}

// Orient::currentAccel() = PseudoAccel::operator() (include/PseudoAccel.H:45-91): centre
// acceleration 2a of the quadratic fits (CENTER), omega = n x dn/dt and its rate n x d2n/dt2 from
// the fits of the axis evaluated at the last time (AXIS); only once the queue is full, the last
// values persist otherwise.
extern "C" int exp_amd_orient_accel(exp_amd_orient *o, double accel[3], double omega[3], double domdt[3])
{
  if (!o) return EXP_AMD_ERR_ARG;
  const bool CEN = o->oflags & ORI_CENTER, AX = o->oflags & ORI_AXIS;
  if (o->naccel && (CEN || AX) && o->aq.size() == o->naccel) {
    std::vector<double> t, v[6];
    for (auto &e : o->aq) {
      t.push_back(e[0]);
      for (int k = 0; k < 6; k++) v[k].push_back(e[1 + k]);
    }
    double a, b, c;
    if (CEN)
      for (int k = 0; k < 3; k++) { quadls(t, v[k], a, b, c); o->ps_accel[k] = 2.0 * a; }
    if (AX) {
      const double T = t.back();
      double n[3], dn[3], d2n[3];
      for (int k = 0; k < 3; k++) {
        quadls(t, v[3 + k], a, b, c);
        n[k] = a * T * T + b * T + c;
        dn[k] = 2.0 * a * T + b;
        d2n[k] = 2.0 * a;
      }
      o->ps_omega[0] = n[1] * dn[2] - n[2] * dn[1];
      o->ps_omega[1] = n[2] * dn[0] - n[0] * dn[2];
      o->ps_omega[2] = n[0] * dn[1] - n[1] * dn[0];
      o->ps_domdt[0] = n[1] * d2n[2] - n[2] * d2n[1];
      o->ps_domdt[1] = n[2] * d2n[0] - n[0] * d2n[2];
      o->ps_domdt[2] = n[0] * d2n[1] - n[1] * d2n[0];
    }
  }
  for (int k = 0; k < 3; k++) {
    if (accel) accel[k] = o->ps_accel[k];
    if (omega) omega[k] = o->ps_omega[k];
    if (domdt) domdt[k] = o->ps_domdt[k];
  }
  return EXP_AMD_OK;
}

extern "C" unsigned exp_amd_orient_flags(const exp_amd_orient *o) { return o ? o->oflags : 0u; }

// currentCenter / currentAxis / transformBody / transformOrig and the diagnostics of logEntry
// (src/Orient.H:166-194, src/Orient.cc:749-783): stats = {Ecurr, used, sigA, sigC, sigCz, mtot,
// axis1[3], center1[3], center0[3]}
extern "C" int exp_amd_orient_get(const exp_amd_orient *o, double center[3], double axis[3],
                                  double body[9], double orig[9], double stats[15])
{
  if (!o) return EXP_AMD_ERR_ARG;
  for (int k = 0; k < 3; k++) {
    if (center) center[k] = o->center[k];
    if (axis) axis[k] = o->axis[k];
  }
  for (int k = 0; k < 9; k++) {
    if (body) body[k] = o->body[k];
    if (orig) orig[k] = o->orig[k];
  }
  if (stats) {
    stats[0] = o->Ecurr; stats[1] = (double)o->used; stats[2] = o->sigA; stats[3] = o->sigC;
    stats[4] = o->sigCz; stats[5] = o->mtot;
    for (int k = 0; k < 3; k++) {
      stats[6 + k] = o->axis1[k]; stats[9 + k] = o->center1[k]; stats[12 + k] = o->center0[k];
    }
  }
  return EXP_AMD_OK;
}

// ---- the log file and the restart from it (src/Orient.cc:84-335 constructor, :742-785 logEntry) ----
// One row per logEntry call, 33 columns of width 15 in the stream's default format (6 significant
// digits): time, Ecurr, used, axis, axis1, centre, centre0, centre1, com, com0, pseudo-acceleration,
// omega, domega/dt.  (The header labels columns 10-15 "anl" then "reg"; the rows hold the regression
// centre first, then the analytic one -- both as the reference writes them.)
static const char *const ORI_LOG_LABELS[33] = {
    "Time", "E_curr", "Used", "X-axis(reg)", "Y-axis(reg)", "Z-axis(reg)", "X-axis(cur)", "Y-axis(cur)",
    "Z-axis(cur)", "X-center(anl)", "Y-center(anl)", "Z-center(anl)", "X-center(reg)", "Y-center(reg)",
    "Z-center(reg)", "X-center(cur)", "Y-center(cur)", "Z-center(cur)", "X-com(cur)", "Y-com(cur)",
    "Z-com(cur)", "X-com(dif)", "Y-com(dif)", "Z-com(dif)", "X-accel", "Y-accel", "Z-accel", "Omega_X",
    "Omega_Y", "Omega_Z", "dOmega/dt_X", "dOmega/dt_Y", "dOmega/dt_Z"};

// What root learns from the old log and every rank then holds (the reference broadcasts Ecurr, axis,
// centre, centre0 and the two histories, :205-230; the pseudo-acceleration queue travels here too so
// that the ranks stay identical).
struct OriRestart {
  double in_ok = 0, Ecurr = 0;
  V3 axis{{0, 0, 1}}, center{{0, 0, 0}}, center0{{0, 0, 0}}, axis1{{0, 0, 0}}, center1{{0, 0, 0}};
  std::deque<DV> sumsA, sumsC;
  std::deque<std::array<double, 7>> aq;
  long long rows = 0;
};

// Root's pass over the old log (:88-199): move it to <logfile>.bak, copy every data row up to the
// current time into a fresh <logfile> (comment rows are dropped, as there), and rebuild the state
// from the rows copied.  Returns 0, or a message.
static const char *orient_read_log(exp_amd_orient *o, bool restart, double tnow, double dtime, int Mstep,
                                   bool queue_center1, OriRestart &R)
{
  {
    std::ifstream probe(o->logfile.c_str());
    if (!probe) {
      // no previous log: write the two header rows (:236-284)
      std::ofstream out(o->logfile.c_str());
      if (!out) return "Orient: error opening log file";
      out.setf(std::ios::left);
      for (int k = 0; k < 33; k++) out << std::setw(15) << (std::string(k ? "| " : "# ") + ORI_LOG_LABELS[k]);
      out << std::endl;
      out.fill('-');
      for (int k = 0; k < 33; k++) out << (k ? "| " : "# ") << std::setw(13) << k + 1;
      out << std::endl;
      return nullptr;
    }
  }
  const std::string backup = o->logfile + ".bak";
  if (std::rename(o->logfile.c_str(), backup.c_str())) return "Orient: error making backup file";
  std::ofstream out(o->logfile.c_str());
  if (!out) return "Orient: error opening new log file for writing";
  std::ifstream in(backup.c_str());
  if (!in) return "Orient: error opening original log file for reading";
  R.in_ok = 1;
  R.Ecurr = o->Ecurr; R.axis = o->axis; R.center = o->center; R.center0 = o->center0;
  R.axis1 = o->axis1; R.center1 = o->center1;
  std::string row;
  while (in && restart) {
    std::getline(in, row);
    if (in.fail() || in.eof()) break;              // a last row without its newline is not taken (:134)
    if (!row.empty() && row[0] == '#') continue;
    std::istringstream line(row);
    double time = 0;
    line >> time;
    if (tnow + 0.1 * dtime / Mstep < time) break;  // read until the current time is reached (:146)
    out << row << "\n";
    long long tused;
    line >> R.Ecurr >> tused;
    for (int k = 0; k < 3; k++) line >> R.axis[k];
    for (int k = 0; k < 3; k++) line >> R.axis1[k];
    for (int k = 0; k < 3; k++) line >> R.center[k];
    for (int k = 0; k < 3; k++) line >> R.center0[k];
    for (int k = 0; k < 3; k++) line >> R.center1[k];
    R.rows++;
    if (o->oflags & ORI_AXIS) {
      R.sumsA.push_back(DV(time, R.axis1));
      if ((int)R.sumsA.size() > o->keep) R.sumsA.pop_front();
    }
    if (o->oflags & ORI_CENTER) {
      R.sumsC.push_back(DV(time, R.center1));
      if ((int)R.sumsC.size() > o->keep) R.sumsC.pop_front();
    }
    // com, com0, pseudo-acceleration: three triples read into the same vector, the last one stays
    // (:174-186); a row that ends early feeds nothing to the queue
    double pseudo[3] = {0, 0, 0};
    bool all = true;
    for (int i = 0; i < 3; i++) {
      if (line.eof()) { all = false; break; }
      for (int k = 0; k < 3; k++) line >> pseudo[k];
    }
    if (all && o->naccel) {
      // The reference queues (time, pseudo, axis1) here -- the logged ACCELERATION in the slot that
      // accumulate() fills with centre1 (:711) -- so its fits straddle two different quantities until
      // the queue has turned over.  Restated as is unless the caller asks for centre1.
      const double *cq = queue_center1 ? R.center1.data() : pseudo;
      R.aq.push_back({time, cq[0], cq[1], cq[2], R.axis1[0], R.axis1[1], R.axis1[2]});
      if (R.aq.size() > o->naccel) R.aq.pop_front();
    }
  }
  return nullptr;
}

bool expamd_orient_has_log(const exp_amd_orient *o) { return o && !o->logfile.empty(); }

/* Orient's Logfile constructor argument and the restart block of the constructor. */
extern "C" int exp_amd_orient_open_log(exp_amd_orient *o, const char *logfile, unsigned flags, double tnow,
                                       double dtime, int Mstep, long long *rows)
{
  if (!o || !logfile || !*logfile || Mstep < 1) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = o->ctx;
  o->logfile = logfile;
  OriRestart R;
  const bool multi = ctx->nranks > 1 || ctx->ar_fn;
  const char *msg = nullptr;
  if (ctx->rank == 0) msg = orient_read_log(o, flags & 1u, tnow, dtime, Mstep, flags & 2u, R);
  if (multi) {
    // root's state to every rank: one sum over a vector the others leave at zero
    const size_t K = (size_t)o->keep, Q = o->naccel, NV = 21 + 8 * K + 7 * Q;
    std::vector<double> v(NV, 0.0);
    if (ctx->rank == 0) {
      size_t i = 0;
      v[i++] = msg ? -1.0 : R.in_ok; v[i++] = R.Ecurr;
      for (const V3 *a : {&R.axis, &R.center, &R.center0, &R.axis1, &R.center1})
        for (int k = 0; k < 3; k++) v[i++] = (*a)[k];
      v[i++] = (double)R.sumsA.size(); v[i++] = (double)R.sumsC.size(); v[i++] = (double)R.aq.size();
      v[i++] = (double)R.rows;
      for (size_t j = 0; j < R.sumsA.size(); j++) { v[21 + 4 * j] = R.sumsA[j].first; for (int k = 0; k < 3; k++) v[22 + 4 * j + k] = R.sumsA[j].second[k]; }
      for (size_t j = 0; j < R.sumsC.size(); j++) { v[21 + 4 * (K + j)] = R.sumsC[j].first; for (int k = 0; k < 3; k++) v[22 + 4 * (K + j) + k] = R.sumsC[j].second[k]; }
      for (size_t j = 0; j < R.aq.size(); j++) for (int k = 0; k < 7; k++) v[21 + 8 * K + 7 * j + k] = R.aq[j][k];
    }
    DevBuf<double> d;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (d.alloc(NV) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "Orient: hipMalloc failed");
    hipError_t e = hipMemcpyAsync(d.p, v.data(), NV * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    int rc = e == hipSuccess ? expamd_allreduce(ctx, d.p, NV) : EXP_AMD_ERR_HIP;
    if (!rc) e = hipMemcpyAsync(v.data(), d.p, NV * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (rc) return rc;
    if (e != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "Orient: restart broadcast failed");
    if (ctx->rank != 0) {
      if (v[0] < 0) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "Orient: root could not open its log file");
      size_t i = 1;
      R.in_ok = v[0]; R.Ecurr = v[i++];
      for (V3 *a : {&R.axis, &R.center, &R.center0, &R.axis1, &R.center1})
        for (int k = 0; k < 3; k++) (*a)[k] = v[i++];
      const size_t nA = (size_t)v[i++], nC = (size_t)v[i++], nQ = (size_t)v[i++];
      R.rows = (long long)v[i++];
      for (size_t j = 0; j < nA; j++) R.sumsA.push_back(DV(v[21 + 4 * j], V3{{v[22 + 4 * j], v[23 + 4 * j], v[24 + 4 * j]}}));
      for (size_t j = 0; j < nC; j++) R.sumsC.push_back(DV(v[21 + 4 * (K + j)], V3{{v[22 + 4 * (K + j)], v[23 + 4 * (K + j)], v[24 + 4 * (K + j)]}}));
      for (size_t j = 0; j < nQ; j++) {
        std::array<double, 7> q;
        for (int k = 0; k < 7; k++) q[k] = v[21 + 8 * K + 7 * j + k];
        R.aq.push_back(q);
      }
    }
  }
  if (msg) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "%s <%s>", msg, logfile);
  if (rows) *rows = R.rows;
  if (R.in_ok != 0) {
    o->Ecurr = R.Ecurr; o->axis = R.axis; o->center = R.center; o->center0 = R.center0;
    o->axis1 = R.axis1; o->center1 = R.center1;
    o->sumsA = R.sumsA; o->sumsC = R.sumsC;
    if (o->naccel) o->aq = R.aq;
    if (o->oflags & ORI_AXIS) {                   // (:325-335)
      const double phi = atan2(o->axis[1], o->axis[0]);
      const double theta = -acos(o->axis[2] / sqrt(o->axis[0] * o->axis[0] + o->axis[1] * o->axis[1] +
                                                   o->axis[2] * o->axis[2]));
      euler_slater(phi, theta, 0.0, 0, o->body);
      euler_slater(phi, theta, 0.0, 1, o->orig);
    }
  }
  return EXP_AMD_OK;
}

/* Orient::logEntry(time, c) (:742-785): root appends one row; com / com0 are the component's. */
extern "C" int exp_amd_orient_log_entry(exp_amd_orient *o, double time, const double com[3], const double com0[3])
{
  if (!o) return EXP_AMD_ERR_ARG;
  if (o->logfile.empty()) return expamd_fail(o->ctx, EXP_AMD_ERR_ARG, "Orient: no log file is open");
  double a[3], w[3], dw[3];
  int rc = exp_amd_orient_accel(o, a, w, dw);      // the queue is evaluated on every rank, written by root
  if (rc) return rc;
  if (o->ctx->rank) return EXP_AMD_OK;
  std::ofstream outl(o->logfile.c_str(), std::ios::app);
  if (!outl) return EXP_AMD_OK;                    // the reference skips the row silently
  const double zero[3] = {0, 0, 0};
  if (!com) com = zero;
  if (!com0) com0 = zero;
  outl << std::setw(15) << time << std::setw(15) << o->Ecurr << std::setw(15) << o->used;
  const double *cols[10] = {o->axis.data(), o->axis1.data(), o->center.data(), o->center0.data(),
                            o->center1.data(), com, com0, a, w, dw};
  for (const double *t : cols)
    for (int k = 0; k < 3; k++) outl << std::setw(15) << t[k];
  outl << std::endl;
  return EXP_AMD_OK;
}
