// Common part of every force method (sphereSL, cylinder): coefficient buffers, multistep level
// bookkeeping and the generic half of the C ABI (force_api.hip dispatches through the virtuals).
#pragma once
#include "particles.h"

// compute_multistep_coefficients (src/SphericalBasis.cc:1252-1333, src/CylEXP.cc:192-282): the interpolation weights
// (a, b) of the levels below mfirst[mdrft] and the combination itself, shared by the stand-alone kernel of force_api.hip
// and the fused per-level-sums + combination kernels of the two force methods
struct CombineW { double ab[2 * 17]; };
inline void expamd_combine_weights(int ms, int mdrft, int *mfirst, CombineW *W)
{
  int mf = 0;                                   // src/multistep.cc:630-680: mfirst[mdrft], dstepL/N[M][mdrft]
  for (int M = 0; M <= ms; M++)
    if (mdrft == 0 || mdrft % (1 << (ms - M)) == 0) { mf = M; break; }
  for (int k = 0; k < 2 * 17; k++) W->ab[k] = 0.0;
  for (int M = 0; M < mf; M++) {
    const int d = 1 << (ms - M);
    const int dL = (mdrft / d) * d, dN = dL + d;
    const double b = (double)(mdrft - dL) / (double)(dN - dL);
    W->ab[2 * M] = 1.0 - b;
    W->ab[2 * M + 1] = b;
  }
  *mfirst = mf;
}
#if defined(__HIPCC__)
__device__ __forceinline__ double expamd_combine_one(const double *__restrict__ L, const double *__restrict__ N,
                                                     size_t stride, int nlev, int mfirst, const double *ab, size_t k)
{
  double s = 0.0;
  for (int M = 0; M < mfirst; M++) s += ab[2 * M] * L[(size_t)M * stride + k] + ab[2 * M + 1] * N[(size_t)M * stride + k];
  for (int M = mfirst; M < nlev; M++) s += N[(size_t)M * stride + k];
  return s;
}
#endif

struct KaLaunch;                                        // kick_adjust.h
typedef void (*ka_launch_fn)(void *self, const KaLaunch &L);

struct exp_amd_force {
  exp_amd_ctx *ctx = nullptr;
  int multistep = 0;
  size_t ncoef = 0;                 // coefficients visible through the ABI
  size_t ncoef_dev = 0;             // device stride of one set (ncoef + tail riding the all-reduce)
  DevBuf<double> d_coef;            // current expansion coefficients (expcoef / accum_cos|sin)
  DevBuf<double> d_coefN, d_coefL;  // per-level new / last sets  (src/SphericalBasis.cc:785-792)
  DevBuf<double> d_scratch;         // >= 64 doubles of scratch for small host->device parameters
  DevBuf<unsigned long long> d_used;
  int mlevel = 0;
  // Component::Adiabatic() of the component the basis belongs to, as the host evaluated it for the CURRENT time
  // (src/Component.cc:4214-4220; exp_amd_force_set_mass_scale): multiplies every mass the accumulation and the
  // level-change differencing read (src/SphericalBasis.cc:441, :471, :1161; src/Cylinder.cc:834, :1758)
  double mass_scale = 1.0;
  // "self_consistent: false" (src/SphericalBasis.cc:114-117, :694; src/Cylinder.cc:557, :959, :1755): once the first
  // evaluation is done and begin_run is over (`initializing`) the coefficients stay what they are: frozen()
  bool self_consistent = true, firstime_coef = true, initializing = false;
  bool frozen() const { return !self_consistent && !firstime_coef && !initializing; }
  bool proj_dirty = true;           // projected force tables are stale w.r.t. d_coef
  // accelerate() itself writes d_coef before it evaluates (the sphere's FIX_L0 copies, src/SphericalBasis.cc:1689-1694): a
  // cross force on the other stream must then stay ordered behind the self force (host.hip: ev_coef / ev_self)
  bool accel_writes_coef = false;
  exp_amd_comp *home = nullptr;     // component whose particles define the expansion centre
  // ... and what it looked like when it was destroyed while this force still pointed at it (pyEXP
  // builds its coefficients from temporary components): frame of the expansion for external targets
  bool home_gone = false;
  double home_center[3] = {0, 0, 0}, home_rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool home_use_rot = false;
  void forget_home(const exp_amd_comp *c);

  virtual ~exp_amd_force() {}
  // sort `c` into this basis' cell order (optionally applying kick+drift on the way), accumulate
  // the particles of the current level into the coefficient set, all-reduce
  virtual int determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick,
                                     double dt_drift, bool have_keys = false) = 0;
  // acc (+)= force, pot (+)= potential on the particles of levels >= mlevel of `t`
  // nk_dtk/nk_dtd != 0 (fused step only): also write the sort key each particle will have after
  // the NEXT step's kick(nk_dtk)+drift(nk_dtd) to t->key and count them into t->hist, so that
  // the next step needs no key pass.  Forces that do not support it return 0 in *prekey_done.
  // defer_kick (fused step only): do not store v += a*dt_kick; remember it in t->pending_kick (the
  // next fused step's scatter pass, or expamd_comp_touch, applies it as its own rounding step).
  virtual int accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick,
                         double nk_dtk = 0.0, double nk_dtd = 0.0, bool *prekey_done = nullptr,
                         bool defer_kick = false) = 0;
  // exp_amd_step_kdk on a large single-level component: the same step with the particle store in
  // two independently sorted halves, the sort passes of one half overlapping the accumulate /
  // force passes of the other on a second stream.  *handled = false: not supported / not worth it.
  virtual int fused_step_split(exp_amd_comp *, double, bool, bool *handled)
  {
    *handled = false;
    return EXP_AMD_OK;
  }
  // ... or without sort passes at all: the force pass places every particle in the next step's cell order (the APPEND step,
  // sph.hip).  *handled = false: not offered / not in its steady state yet.
  virtual int fused_step_append(exp_amd_comp *, double, bool, bool *handled)
  {
    *handled = false;
    return EXP_AMD_OK;
  }
  virtual void release() = 0;
  // multistep_update for every particle whose proposed level (c->newlev) differs from its
  // level: subtract its contribution from expcoefN[from], add it to expcoefN[to]
  // (src/SphericalBasis.cc:1033-1079, :1156-1228; src/CylEXP.cc:56-188), reduced over ranks
  virtual int multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft) = 0;
  // a launcher of k_kick_adjust that also writes component c's sort keys for sub-step 0 of the next master step (this
  // force's cells, full keys: the sort collapses those of sparse levels); false: not offered (any-order bases, ...)
  virtual bool prekey_launcher(exp_amd_comp *, ka_launch_fn *, void **) { return false; }
  // re-establish this basis' (level, cell) order after levels changed
  // (only levels >= first can have changed: their slot range alone is re-ordered)
  virtual int resort(exp_amd_comp *c, int first = 0) = 0;

  // First half of a block-multistep sub-step for the ACTIVE levels [lo, multistep] of `c` (the active
  // levels of a sub-step are always a suffix, src/multistep.cc:651-660) in one sweep instead of one
  // pass per level: per level M kick DT(M)/2 and drift DT(M), DT(M) = dt_min 2^(multistep-M), fused
  // into ONE cell sort of that slot range (src/step.cc:115-160: incr_velocity, incr_position), then
  // for every active level the N/L swap and the accumulation of levlist[M] into expcoefN[M]
  // (compute_expansion(M), src/ComponentContainer.cc:1173-1226) from one launch over the range with
  // per-level moment buffers, one contraction and ONE all-reduce of the contiguous level block.
  // The levels are disjoint particle sets and each level's accumulation reads only its own
  // particles, so the reference's level-by-level order gives the same sums.  dt_min <= 0: no advance
  // (begin_run's expansion of every level).
  // mdrft_combine >= 0 (a single rank only: no all-reduce sits between the two): the kernel that sums the per-level
  // sets also forms the combined set of compute_multistep_coefficients(mdrft_combine) -- one link less in the chain of
  // dependent launches of a sub-step; `combined_mdrft` then says so until the sets change again
  // phase: 0 = the whole first half; 1 = only the kick + drift (the advance sort / in-place advance), 2 = only the
  // accumulation behind it -- the step driver issues phase 1 of every component before phase 2 of any, so that the second
  // stream is not idle while the host issues the first component's chain
  virtual int substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine = -1, int phase = 0) = 0;
  int combined_mdrft = -1;

  // Level population below which a multistep level is left un-cell-sorted (see exp_amd_comp::
  // sparse_mask).  Sorting a level costs ~100 us of small launches per sub-step; accumulating one
  // particle by atomics costs `values per particle` fp64 atomics at ~3e10/s on MI355X, so the two meet
  // near 3e6 / values particles.  ctx->dense_min >= 0 overrides.
  virtual long long sparse_threshold() const { return 0; }

  // Host-side state of this force that alternates from one fused step to the next (the spherical method's two
  // work-list counters): part of the key under which exp_amd_step_kdk_n replays a captured pair of steps
  virtual int step_parity() const { return 0; }
  // false: a fused step of this force has host-side effects that a replayed graph would not repeat (the NOISE mode draws
  // its deviates on the host at every force evaluation) -- exp_amd_step_kdk_n then steps eagerly
  virtual bool step_graph_ok() const { return true; }
  // the graph of TWO consecutive fused steps of (this, one component, one dt) and the host state it was captured in
  struct StepGraph {
    hipGraphExec_t exec = nullptr;
    const exp_amd_comp *comp = nullptr;
    double dt = 0.0, center[3] = {0, 0, 0}, pending = 0.0;
    size_t n = 0;
    int cur = -1, parity = -1, prekick = -1, det = -1;
    unsigned long long epoch = 0;
    unsigned long long mutation = 0; // expamd_mutation_counter() right after the capture (common.h)
    bool refused = false;          // capture failed once for this pair: eager from then on
  } step_graph;

  virtual int get_used(long long *used);
  // PotAccel::multistep_reset (src/PotAccel.H:288): start of a master step
  virtual int multistep_reset() { return EXP_AMD_OK; }

  int alloc_common(size_t ncoef_, int multistep_, size_t tail = 0);
  void release_common();
};
