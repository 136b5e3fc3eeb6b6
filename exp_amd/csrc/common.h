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

#include "../../include/exp_amd.h"

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
  long long split_min = 0;           // components at least this large take the split step (<= 0: never;
                                     // off by default: +1.5 % at 1e8 on MI355X)
  bool deterministic = false;        // order-independent (bit-reproducible) coefficient sums, exp_amd_ctx_set_deterministic
  long long dense_min = -1;          // block multistep: levels with fewer particles are not cell-sorted (< 0: per force method)
                                     // (exp_amd_ctx_set_dense_min; EXP_AMD_DENSE_MIN sets the default)
  hipStream_t aux = nullptr;
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
    if (p) (void)hipFree(p);
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
__device__ __forceinline__ double mul_then_add(double x, double a, double b)
{
  double t = a * b;
  asm volatile("" : "+v"(t));
  return x + t;
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
