// Shared host-side plumbing of libexp_amd: context, device buffers, error handling,
// per-kernel HIP-event profiling.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <cstdlib>
#include <string>
#include <vector>
#include <atomic>

// Library-wide mutation counter: bumped by every (re)allocation or release of device memory and by every call that
// changes what a step's kernels are launched WITH (force settings, tables, uploads, frame, any outside call on a
// component).  A HIP graph of fused steps bakes those pointers and values in as kernel arguments; it is replayed only
// while the counter still has the value it had when the graph was captured (exp_amd_step_kdk_n).
inline std::atomic<unsigned long long> &expamd_mutation_counter()
{
  static std::atomic<unsigned long long> n{0};
  return n;
}
inline void expamd_mutated() { expamd_mutation_counter().fetch_add(1, std::memory_order_relaxed); }


#include "../../include/exp_amd.h"

// Run-time switches.  The default build reads only the few environment variables documented in include/exp_amd.h
// ("Environment"); the tuning and A/B switches of the development rounds are compile-time constants -- unless the library
// is built with -DEXP_AMD_EXPERIMENTAL (make EXPERIMENTAL=1), which turns each EXPAMD_EXPT(name, default) back into an
// environment variable read once.
#ifdef EXP_AMD_EXPERIMENTAL
#define EXPAMD_EXPT(name, dflt) ([&] { static const long long v_ = [&] { const char *e_ = getenv(name); return e_ ? atoll(e_) : (long long)(dflt); }(); return v_; }())
#else
#define EXPAMD_EXPT(name, dflt) ((long long)(dflt))
#endif

#define EXPAMD_WAVE 64

struct ProfileSlot {
  const char *name;
  double ms_total = 0.0;
  long long launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct exp_amd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // collectives
  void *rccl_lib = nullptr;
  void *rccl_comm = nullptr;
  void *rccl_comm2 = nullptr;        // a second communicator (ncclCommSplit of the first) for collectives issued on `aux`: the
                                     // two-stream schedule of the step driver keeps one communicator per stream
  bool rccl_comm2_tried = false;
  void *rccl_allreduce = nullptr;    // ncclAllReduce, resolved once at exp_amd_comm_init_rank
  unsigned long long ar_calls = 0;   // all-reduces issued so far (either path)
  int nranks = 1, rank = 0;
  exp_amd_allreduce_fn ar_fn = nullptr;
  void *ar_user = nullptr;
  // profiling
  bool profile = false;
  std::vector<ProfileSlot> slots;
  std::vector<hipEvent_t> event_pool;
  int num_cu = 256;
  // second stream of the split fused step: the HBM-bound sort passes of one half of a component run
  // here while the VALU-bound accumulate / force passes of the other half run on `stream`
  unsigned long long force_epoch = 0;           // bumped when a force dies: sort keys recorded for "the force
                                                // at this address" must not outlive it
  std::vector<struct exp_amd_force *> forces;   // live force objects (so that a dying component can be forgotten)
  std::vector<struct exp_amd_comp *> appended;  // components whose store is in the append step's layout (particles.h): a dying
                                                // force turns those it owns back into ordinary stores first
  long long append_min = 1 << 20;    // single-level components at least this large take the APPEND fused step: the force pass
                                     // places every particle in the next step's cell order itself, no sort passes
                                     // (sph.hip: fused_step_append; exp_amd_ctx_set_append_min; 0: never; < 0: as -nmin with
                                     // regions WITHOUT slack and a 64-slot tail, so that the run-out-of-room path is taken: tests)
  bool append_lean = false;          // ... and places neither acceleration nor potential (32 of the 88 bytes a particle, which no
                                     // pass of the next step reads: 5.7 -> 5.2 ms for the pass at 1e8): the first call that looks
                                     // at the component has them re-evaluated from the coefficient set kept at the completed step
                                     // (exp_amd_ctx_set_append_lean; EXP_AMD_APPEND_LEAN; off by default: the state a step leaves
                                     // in the store is then complete without a further pass)
  long long split_min = 0;           // components at least this large take the split step (<= 0: never;
                                     // off by default: +1.5 % at 1e8 on MI355X)
  bool prekick = true;               // the fused step stores velocities with the NEXT step's opening half-kick applied
                                     // (exp_amd_ctx_set_prekick)
  bool deterministic = false;        // order-independent (bit-reproducible) coefficient sums, exp_amd_ctx_set_deterministic
  long long mover_list_min = 8192;   // block multistep: from this many level changes in a sweep on, the differencing goes
                                     // through the accumulation kernel over a list of the movers instead of per-particle
                                     // atomics (exp_amd_ctx_set_mover_list_min; < 0: never)
  long long mover_slices_min = 65536; // ... and from this many on with one adding pass per proposed level (experimental builds: EXP_AMD_MOVER_SLICES_MIN)
  long long stage_max = 1 << 20;     // particles up to which the per-particle atomic paths are staged (values by plain stores,
                                     // one lane per value for the atomics): 8 (L+1)^2 bytes each (experimental builds: EXP_AMD_STAGE_MAX)
  long long thin_acc_scale = 4;      // ... times this for the accumulation side alone (the direct kernel replaces three launches
                                     // there; config 4, 13e3 particles in the halo's levels >= 2: 6.05 -> 5.93 ms per master step;
                                     // experimental builds: EXP_AMD_THIN_ACC_SCALE)
  long long thin_max = 8192;         // block multistep: an active slot range of at most this many particles, all of it in sparse
                                     // levels, is accumulated and evaluated straight from the basis tables (k_*_acc_thin,
                                     // k_*_force_thin: no moments, no contraction, no projected table); 0: never
                                     // (exp_amd_ctx_set_thin_max)
  long long dense_min = -1;          // block multistep: levels with fewer particles are not cell-sorted (< 0: per force method)
                                     // (exp_amd_ctx_set_dense_min)
  hipStream_t aux = nullptr;
  struct ScanSums { uint32_t *p = nullptr; size_t n = 0; hipError_t alloc(size_t c) { expamd_mutated(); if (p) (void)hipFree(p); p = nullptr; n = 0; hipError_t e = hipMalloc((void **)&p, c * sizeof(uint32_t)); if (e == hipSuccess) n = c; return e; } } scan_sums[2];   // chunk totals of multi-chunk scans, one per stream (particles.hip)
  hipEvent_t ev_sorted[2] = {nullptr, nullptr}, ev_forced[2] = {nullptr, nullptr};
};
// lazily creates ctx->aux and the four events
int expamd_ctx_aux(exp_amd_ctx *ctx);

extern thread_local std::string g_exp_amd_global_err;

int expamd_fail(exp_amd_ctx *ctx, int code, const char *fmt, ...);

#define HIP_TRY(ctx, call)                                                               \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess)                                                               \
      return expamd_fail((ctx), EXP_AMD_ERR_HIP, "%s failed: %s (%s:%d)", #call,         \
                         hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)

// RAII-less simple device buffer (freed explicitly by owners' destructors)
inline bool expamd_poison()
{
  static const bool on = getenv("EXP_AMD_POISON") && atoi(getenv("EXP_AMD_POISON")) != 0;
  return on;
}

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    expamd_mutated();
    n = count;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    // EXP_AMD_POISON=1 (tests): fresh device memory reads as NaN / 0xff..., so that anything which
    // relies on hipMalloc handing out zeros shows up
    if (e == hipSuccess && expamd_poison()) {
      e = hipMemset(p, 0xff, count * sizeof(T));       // (null stream: not ordered with ours)
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
  }
  void release() {
    if (p) { (void)hipFree(p); expamd_mutated(); }
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
};

// Scoped kernel timer: records a pair of events around a launch when profiling is on.
struct ProfScope {
  exp_amd_ctx *ctx;
  int slot = -1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  ProfScope(exp_amd_ctx *c, const char *name, hipStream_t on = nullptr);
  ~ProfScope();
};

int expamd_allreduce(exp_amd_ctx *ctx, double *dev, size_t count);
// true when collectives may be issued on ctx->aux as well as on ctx->stream: a single rank, a host callback (it is
// handed the stream), or an RCCL communicator that could be split into a second one
// Opt one kernel in to more than 64 KB of dynamic LDS (once per kernel).  A refusal is KEPT, not discarded: it becomes the
// library's global error text (exp_amd_last_global_error), and the launch that then asks for the LDS fails with an invalid-value
// status which the caller's hipGetLastError check reports.
inline bool expamd_big_lds(const void *fn, const char *name)
{
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (e != hipSuccess) {
    (void)expamd_fail(nullptr, EXP_AMD_ERR_HIP, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize, 128 KB) -> %s", name,
                      hipGetErrorString(e));
    (void)hipGetLastError();
  }
  return e == hipSuccess;
}
bool expamd_comm_two_streams(exp_amd_ctx *ctx);          // query, no side effect
bool expamd_comm_prepare_two_streams(exp_amd_ctx *ctx);  // collective: splits the RCCL communicator on first use
bool expamd_orient_has_log(const exp_amd_orient *o);     // orient.hip: a log file is open

// x + a*b rounded as a separate multiply and add (what the reference's scalar CPU code does).
// hipcc contracts a*b+c to an FMA by default and HIP's __dmul_rn/__dadd_rn are plain operators;
// the empty asm makes the product opaque so the two roundings survive.
// Component::getPseudoAccel (src/Component.cc:4407-4427): the acceleration of the component's
// non-inertial frame that Component::AddAcc subtracts from every force it is handed
// (src/Component.H:914-921): the centre's acceleration (EJ & CENTER) and the Coriolis, Euler and
// centrifugal terms of the rotating axis (EJ & AXIS), from the stored position and velocity.
struct PseudoDev {
  int center, axis;
  double a[3], om[3], dom[3];
};
#if defined(__HIPCC__)
__device__ __forceinline__ void pseudo_accel(const PseudoDev &P, double x, double y, double z, double vx,
                                             double vy, double vz, double &px, double &py, double &pz)
{
  px = py = pz = 0.0;
  if (P.center) { px += P.a[0]; py += P.a[1]; pz += P.a[2]; }
  if (P.axis) {
    // 2 omega x V + domega/dt x P + omega x (omega x P)
    const double ox = P.om[0], oy = P.om[1], oz = P.om[2];
    const double wx = oy * z - oz * y, wy = oz * x - ox * z, wz = ox * y - oy * x;     // omega x P
    px += 2.0 * (oy * vz - oz * vy) + (P.dom[1] * z - P.dom[2] * y) + (oy * wz - oz * wy);
    py += 2.0 * (oz * vx - ox * vz) + (P.dom[2] * x - P.dom[0] * z) + (oz * wx - ox * wz);
    pz += 2.0 * (ox * vy - oy * vx) + (P.dom[0] * y - P.dom[1] * x) + (ox * wy - oy * wx);
  }
}
#endif

#if defined(__HIPCC__)
// cos(phi), sin(phi) of phi = atan2(y, x) for x = y = 0 -- a particle exactly on the axis -- as the reference's libm sees
// them: atan2(+-0, +0) = +-0, atan2(+-0, -0) = +-pi (cos = -1, sin = +-1.22e-16: sin of the double next to pi).  The
// sign of a zero x is the sign of the odd-m terms there (src/SphericalBasis.cc:1548, src/Cylinder.cc:1333; they do not
// vanish on the axis: the sphere's P_l^1 sees |cos(theta)| = 1 - 1 ulp, the disk's tables are extrapolated below rmin).
// (x*x + y*y can also be zero by UNDERFLOW -- components below 1e-162 --: those are rescaled and take the ordinary route)
__device__ __forceinline__ void atan2_trig_zero(double xx, double yy, double &c, double &s)
{
  const double ax = __builtin_fabs(xx), ay = __builtin_fabs(yy), mx = ax > ay ? ax : ay;
  if (mx > 0.0) {
    const double u = xx / mx, v = yy / mx;
    const double iR = 1.0 / __builtin_sqrt(u * u + v * v);
    c = u * iR;
    s = v * iR;
    return;
  }
  const bool neg = __builtin_signbit(xx);
  c = neg ? -1.0 : 1.0;
  s = neg ? __builtin_copysign(1.2246467991473532e-16, yy) : yy;
}

// xx^2 + yy^2 and (xx^2 + yy^2) + zz^2 as the reference's compiler forms them (src/SphericalBasis.cc:1545, :1630): every
// product rounded on its own, no fused multiply-add.  For the general force pass: a lane within ~1e-7 rad of the polar
// axis has 1 - |cos(theta)| of a few ulp, its Legendre functions of order m >= 1 go with the square root of that, and one
// ulp of r^2 decides whether it is two, three or four of them (+-18 % in the tangential force of such a lane).
__device__ __forceinline__ double sq_sum2_lit(double a, double b)
{
  double p = a * a, q = b * b;
  asm volatile("" : "+v"(p), "+v"(q));
  return p + q;
}
__device__ __forceinline__ double sq_add_lit(double s, double c)
{
  double q = c * c;
  asm volatile("" : "+v"(q));
  return s + q;
}

__device__ __forceinline__ double mul_then_add(double x, double a, double b)
{
  double t = a * b;
  asm volatile("" : "+v"(t));
  return x + t;
}
#endif

#if defined(__HIPCC__)
// sqrt(t) and ~1/sqrt(t) together, for the squared radii of a model (positive, far from the
// denormal and overflow ranges: no argument scaling; t == 0 is lifted to 1e-300).  The root follows
// the compiler's own fp64 expansion -- v_rsq_f64 seed, one coupled Newton step, two residual
// corrections -- so g is the same correctly rounded value `sqrt` returns; y = 1/sqrt(t) to about an
// ulp falls out of the same iteration.  11 VALU operations where sqrt followed by an IEEE division
// costs about 34.
__device__ __forceinline__ void sqrt_rsqrt(double t, double &g, double &y)
{
#ifdef EXPAMD_EXACT_DIV        // A/B switch (tools/build_variant*.sh): the library sqrt and IEEE divisions
  g = sqrt(t); y = 1.0 / g; return;
#endif
  t = fmax(t, 1e-300);
  const double y0 = __builtin_amdgcn_rsq(t);
  double g0 = t * y0, h = 0.5 * y0;
  const double r0 = fma(-h, g0, 0.5);
  g0 = fma(g0, r0, g0);
  h = fma(h, r0, h);
  double d = fma(-g0, g0, t);
  g0 = fma(d, h, g0);
  d = fma(-g0, g0, t);
  g = fma(d, h, g0);
  y = h + h;
}
// 1/v from an estimate y of it (one Newton step: the error is squared)
__device__ __forceinline__ double rcp_refine(double v, double y)
{
#ifdef EXPAMD_EXACT_DIV
  return 1.0 / v;
#endif
  const double e = fma(-v, y, 1.0);
  return fma(y, e, y);
}
// a / b to about an ulp for b in the normal range: v_rcp_f64 seed, two Newton steps, one residual
// correction of the quotient (8 VALU operations instead of the 11-13 of the IEEE sequence)
__device__ __forceinline__ double div_fast(double a, double b)
{
#ifdef EXPAMD_EXACT_DIV
  return a / b;
#endif
  double y = __builtin_amdgcn_rcp(b);
  y = rcp_refine(b, y);
  y = rcp_refine(b, y);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
}
// asinh(u) for u >= 0 to a few ulp, as log1p(u + u^2 / (1 + sqrt(1 + u^2))) with the logarithm
// written out: 1 + t = 2^k m, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1)/(m + 1)) by its
// series (|z| <= 0.172: ten terms), the rounding error of 1 + t carried along.  About 65 VALU
// operations; the library asinh (extended-precision logarithm inside) costs several times that and
// sits on every particle of the cylinder's accumulate and force passes (EmpCylSL::z_to_y with
// cmapz = 1, exputil/EmpCylSL.cc:7109-7117).
__device__ __forceinline__ double asinh_pos(double u)
{
#ifdef EXPAMD_EXACT_DIV
  return asinh(u);
#endif
  const double u2 = u * u;
  double s, is;
  sqrt_rsqrt(1.0 + u2, s, is);
  const double t = u + div_fast(u2, 1.0 + s);
  const double f = 1.0 + t;
  const double c = t - (f - 1.0);                     // f + c == 1 + t
  int k = __builtin_amdgcn_frexp_exp(f);              // f = mant 2^k, mant in [1/2, 1)
  double m = __builtin_amdgcn_frexp_mant(f);
  const int up = m < 0.70710678118654752 ? 1 : 0;
  m = ldexp(m, up);
  k -= up;
  const double z = div_fast(m - 1.0, m + 1.0);
  const double w = z * z;
  double q = 1.0 / 19.0;
  q = fma(q, w, 1.0 / 17.0);
  q = fma(q, w, 1.0 / 15.0);
  q = fma(q, w, 1.0 / 13.0);
  q = fma(q, w, 1.0 / 11.0);
  q = fma(q, w, 1.0 / 9.0);
  q = fma(q, w, 1.0 / 7.0);
  q = fma(q, w, 1.0 / 5.0);
  q = fma(q, w, 1.0 / 3.0);
  const double z2 = z + z;
  const double kd = (double)k;
  // ln 2 in two pieces (the high one has 21 trailing zero bits: kd * hi is exact)
  const double lo = fma(kd, 1.90821492927058770002e-10, c * __builtin_amdgcn_rcp(f));
  return fma(kd, 6.93147180369123816490e-01, z2 + fma(z2 * w, q, lo));
}
#endif

#if defined(__HIPCC__)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (its fence
// covers global memory), which kills any global prefetch issued before it; the accumulation
// kernels keep next-tile loads in flight across their per-tile barrier.
__device__ __forceinline__ void lds_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

// Deterministic accumulation (acc_add<true>, sph_kernels.h): the constant C = 1.5 * 2^(52+e) whose grid
// 2^e keeps every partial sum of a launch exact, given an upper bound of the sum of |contributions|:
// bound < 2^ex, e = ex - 50 => single terms < 2^(51+e), partial sums < 2^(53+e).  0 when off.
static inline double expamd_det_constant(bool on, double bound)
{
  if (!on || !(bound > 0.0)) return 0.0;
  int ex = 0;
  (void)frexp(bound, &ex);
  return ldexp(1.5, 52 + ex - 50);
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
