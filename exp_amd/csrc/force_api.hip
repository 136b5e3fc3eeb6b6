// Generic half of the C ABI: everything that is the same for every force method.
#include "force.h"
#include "sort_kernels.h"

#include <vector>

void exp_amd_force::forget_home(const exp_amd_comp *c)
{
  if (home != c) return;
  for (int k = 0; k < 3; k++) home_center[k] = c->center[k];
  for (int k = 0; k < 9; k++) home_rot[k] = c->rot[k];
  home_use_rot = c->use_rot;
  home_gone = true;
  home = nullptr;
}

void expamd_forget_component(exp_amd_ctx *ctx, const exp_amd_comp *c)
{
  for (exp_amd_force *f : ctx->forces) {
    f->forget_home(c);
    if (f->step_graph.comp == c) {            // a graph of steps of this component: its buffers are about to go
      if (f->step_graph.exec) (void)hipGraphExecDestroy(f->step_graph.exec);
      f->step_graph = exp_amd_force::StepGraph{};
    }
  }
}

int exp_amd_force::alloc_common(size_t ncoef_, int multistep_, size_t tail)
{
  if (ctx) ctx->forces.push_back(this);
  ncoef = ncoef_;
  ncoef_dev = ncoef_ + tail;
  multistep = multistep_;
  const int nlev = multistep + 1;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(d_coef.alloc(ncoef_dev));
  A(d_coefN.alloc((size_t)nlev * ncoef_dev));
  A(d_coefL.alloc((size_t)nlev * ncoef_dev));
  A(d_scratch.alloc(256));
  A(d_used.alloc(8 + 8 * 128));      // [0], [1]: the counts; the rest: slots of the EXPT_USED_SPREAD A/B build
  if (e != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "force: hipMalloc failed: %s", hipGetErrorString(e));
  HIP_TRY(ctx, hipMemset(d_coef.p, 0, d_coef.bytes()));
  HIP_TRY(ctx, hipMemset(d_coefN.p, 0, d_coefN.bytes()));
  HIP_TRY(ctx, hipMemset(d_coefL.p, 0, d_coefL.bytes()));
  HIP_TRY(ctx, hipMemset(d_used.p, 0, d_used.bytes()));
  return EXP_AMD_OK;
}

void exp_amd_force::release_common()
{
  d_coef.release(); d_coefN.release(); d_coefL.release(); d_scratch.release(); d_used.release();
}

extern "C" void exp_amd_force_destroy(exp_amd_force *f)
{
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  if (f->step_graph.exec) (void)hipGraphExecDestroy(f->step_graph.exec);
  f->step_graph.exec = nullptr;
  {
    // components whose store is in this force's append layout: ordinary stores again while the force can still
    // evaluate the accelerations its placing passes did not store (particles.h: app_acc_stale)
    std::vector<exp_amd_comp *> mine;
    for (exp_amd_comp *c : f->ctx->appended) if (c->app_owner == (const void *)f) mine.push_back(c);
    for (exp_amd_comp *c : mine) (void)expamd_comp_densify(c);
  }
  f->ctx->force_epoch++;
  {
    auto &v = f->ctx->forces;
    for (size_t k = 0; k < v.size(); k++)
      if (v[k] == f) { v.erase(v.begin() + k); break; }
  }
  f->release();
  f->release_common();
  delete f;
}

extern "C" size_t exp_amd_force_ncoef(const exp_amd_force *f) { return f ? f->ncoef : 0; }

extern "C" int exp_amd_force_set_level(exp_amd_force *f, int mlevel)
{
  if (!f || mlevel < 0 || mlevel > f->multistep)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "set_level: level out of range");
  f->mlevel = mlevel;
  return EXP_AMD_OK;
}

// Component::Adiabatic() of the basis' component at the current time (src/Component.cc:4214-4220), evaluated by the host
extern "C" int exp_amd_force_set_mass_scale(exp_amd_force *f, double adb)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!f || !(adb >= 0.0)) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "set_mass_scale: factor must be >= 0");
  f->mass_scale = adb;
  return EXP_AMD_OK;
}

// the "self_consistent" key (src/SphericalBasis.cc:114-117, src/Cylinder.cc:557-558) and the global `initializing`
// (src/begin.cc:80, :129) it is tested with
extern "C" int exp_amd_force_set_self_consistent(exp_amd_force *f, int on)
{
  expamd_mutated();
  if (!f) return EXP_AMD_ERR_ARG;
  f->self_consistent = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_set_initializing(exp_amd_force *f, int on)
{
  expamd_mutated();
  if (!f) return EXP_AMD_ERR_ARG;
  f->initializing = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_coefs_frozen(const exp_amd_force *f) { return f && f->frozen() ? 1 : 0; }

extern "C" int exp_amd_force_determine_coefficients(exp_amd_force *f, exp_amd_comp *c)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!f || !c) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "determine_coefficients: NULL");
  // "Return if we should leave the coefficients fixed" (src/SphericalBasis.cc:694, src/Cylinder.cc:959)
  if (f->frozen()) return EXP_AMD_OK;
  const int rc = f->determine_coefficients(c, false, 0.0, 0.0);
  if (rc == EXP_AMD_OK) f->firstime_coef = false;                  // (:1001, :1198)
  return rc;
}

extern "C" int exp_amd_force_get_acceleration(exp_amd_force *f, exp_amd_comp *target, int external)
{
  if (target) { int rc_ = expamd_comp_touch(target); if (rc_) return rc_; }
  if (!f || !target) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "get_acceleration: NULL");
  return f->accelerate(target, external, false, 0.0);
}

extern "C" int exp_amd_force_get_coefs(exp_amd_force *f, double *coef, size_t count)
{
  if (!f || !coef || count != f->ncoef)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "get_coefs: bad count");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipMemcpyAsync(coef, f->d_coef.p, count * sizeof(double), hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_set_coefs(exp_amd_force *f, const double *coef, size_t count)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!f || !coef || count != f->ncoef)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "set_coefs: bad count");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipMemcpyAsync(f->d_coef.p, coef, count * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  f->proj_dirty = true;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_get_level_coefs(exp_amd_force *f, int level, int which, double *coef,
                                             size_t count)
{
  if (!f || !coef || count != f->ncoef || level < 0 || level > f->multistep)
    return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "get_level_coefs: bad argument");
  exp_amd_ctx *ctx = f->ctx;
  const double *src = (which ? f->d_coefL.p : f->d_coefN.p) + (size_t)level * f->ncoef_dev;
  HIP_TRY(ctx, hipMemcpyAsync(coef, src, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

int exp_amd_force::get_used(long long *used)
{
  unsigned long long u = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&u, d_used.p, sizeof(u), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *used = (long long)u;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_force_used(exp_amd_force *f, long long *used)
{
  if (!f || !used) return EXP_AMD_ERR_ARG;
  return f->get_used(used);
}

// ---- multistep coefficient bookkeeping -----------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_mstep_combine(const double *__restrict__ L, const double *__restrict__ N, int ncoef, int nlev,
                int mfirst, CombineW W, double *__restrict__ out)
{
  const double *ab = W.ab;
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncoef) return;
  // src/SphericalBasis.cc:1252-1333 ; src/CylEXP.cc:192-282
  const double s = expamd_combine_one(L, N, (size_t)ncoef, nlev, mfirst, ab, (size_t)k);
  out[k] = s;
}

extern "C" int exp_amd_force_multistep_reset(exp_amd_force *f)
{
  if (!f) return EXP_AMD_ERR_ARG;
  return f->multistep_reset();
}

extern "C" int exp_amd_force_compute_multistep_coefficients(exp_amd_force *f, int mdrft)
{
  if (!f) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = f->ctx;
  const int ms = f->multistep;
  if (ms == 0) return EXP_AMD_OK;
  const int Mstep = 1 << ms;
  if (mdrft < 0 || mdrft > Mstep) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "mdrft out of range");
  // `if (multistep && (self_consistent || initializing)) compute_multistep_coefficients()`
  // (src/SphericalBasis.cc:1682, src/Cylinder.cc:1469)
  if (!f->self_consistent && !f->initializing) return EXP_AMD_OK;
  int mfirst = 0;
  CombineW W;
  expamd_combine_weights(ms, mdrft, &mfirst, &W);
  k_mstep_combine<<<cdiv(f->ncoef_dev, 256), 256, 0, ctx->stream>>>(
      f->d_coefL.p, f->d_coefN.p, (int)f->ncoef_dev, ms + 1, mfirst, W, f->d_coef.p);
  HIP_TRY(ctx, hipGetLastError());
  f->proj_dirty = true;
  f->combined_mdrft = -1;
  return EXP_AMD_OK;
}

// adjust_multistep_level (src/multistep.cc:344-627) for one component: propose levels from the
// time-step criteria, difference the coefficient sets of the movers (multistep_update / _finish),
// commit the levels and re-establish the (level, cell) order (reset_level_lists).
extern "C" int exp_amd_force_adjust_multistep_level(exp_amd_force *f, exp_amd_comp *c, double dtime,
                                                    const double dynfrac[5], int shiftlevl,
                                                    int mdrft, int first_step,
                                                    long long *nswitch)
{
  if (!f || !c || !dynfrac) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "adjust_multistep_level: NULL");
  { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  exp_amd_ctx *ctx = f->ctx;
  const int ms = f->multistep;
  if (ms == 0) { if (nswitch) *nswitch = 0; return EXP_AMD_OK; }
  const int Mstep = 1 << ms;
  if (mdrft < 0 || mdrft > Mstep) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "mdrft out of range");
  int mfirst = 0;                                   // src/multistep.cc:651-660
  for (int M = 0; M <= ms; M++)
    if (mdrft == 0 || mdrft % (1 << (ms - M)) == 0) { mfirst = M; break; }
  const int first = first_step ? 0 : mfirst;        // src/multistep.cc:451-453
  // `if (not firstCall and c->FreezeLev()) apply = false;` (src/multistep.cc:158, :534): nothing is proposed, nothing moves
  // (firstCall = this_step == 0 and mdrft == 0, begin_run's call; first_step alone is the "do all levels" rule of :453)
  if (c->freeze_levels && !(first_step && mdrft == 0)) { if (nswitch) *nswitch = 0; return EXP_AMD_OK; }
  // "noswitch" (src/multistep.cc:136-147): mstep = mdrft - 1 at do_step's call (src/step.cc:188, :221)
  c->ns_reset = ((c->dtreset && mdrft == 1) || (first_step && mdrft == 0)) ? 1 : 0;
  c->ns_apply = (mdrft == Mstep || (first_step && mdrft == 0)) ? 1 : 0;
  int rc = expamd_comp_propose_levels(c, dtime, dynfrac, shiftlevl, ms, mfirst, first);
  if (rc) return rc;
  // one 8-byte read-back decides whether anything has to be differenced / re-ordered at all
  unsigned long long u = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&u, c->nswitch.p + 64, sizeof(u), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (nswitch) *nswitch = (long long)u;
  // (not self-consistent: Cylinder::multistep_update returns at once, src/Cylinder.cc:1755; the sphere's differencing
  // only feeds level sets that nothing reads again once compute_multistep_coefficients is no longer called)
  if ((ctx->nranks > 1 || ctx->ar_fn || u) && f->self_consistent) {
    // (collective: with several ranks every rank takes part even if it has no mover)
    c->mover_hint = (long long)u;
    rc = f->multistep_update(c, first, mfirst);
    c->mover_hint = -1;
    if (rc) return rc;
  }
  if (u) {
    // only slots of levels >= first can have changed level; if the store was in this force's order
    // the rest of it still is, and only that slot range is re-ordered
    const bool ordered = c->sorted_for == (const void *)f && c->nlevels == ms + 1;
    if ((rc = expamd_comp_commit_levels(c))) return rc;
    if ((rc = f->resort(c, ordered ? first : 0))) return rc;
  }
  return EXP_AMD_OK;
}

// ---- fused step -----------------------------------------------------------------------------------------------

extern "C" int exp_amd_step_kdk(exp_amd_force *f, exp_amd_comp *c, double dt)
{
  if (!f || !c) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "step_kdk: NULL");
  if (f->multistep)
    return expamd_fail(f->ctx, EXP_AMD_ERR_STATE,
                       "step_kdk: multistep force; drive the sub-steps explicitly");
  int rc;
  if (f->frozen()) {
    // coefficients held fixed ("self_consistent: false" after the first evaluation): the step is kick, drift, force,
    // kick with the set as it is -- nothing to fuse the advance into
    if ((rc = exp_amd_comp_kick(c, 0.5 * dt, -1)) || (rc = exp_amd_comp_drift(c, dt, -1))) return rc;
    if ((rc = exp_amd_comp_zero_acc(c, 0))) return rc;
    if ((rc = f->accelerate(c, 0, false, 0.0))) return rc;
    return exp_amd_comp_kick(c, 0.5 * dt, -1);
  }
  // kick dt/2 + drift dt are applied inside the sort passes (no separate HBM pass); acc/pot are
  // recomputed below, so they are not carried through the reorder
  c->acc_live = false;
  // keys + histogram left by the previous fused step's force pass for exactly this advance?
  const bool have_keys = c->prekey_valid && c->prekey_owner == (const void *)f &&
                         c->prekey_epoch == f->ctx->force_epoch &&
                         c->prekey_dtk == 0.5 * dt && c->prekey_dtd == dt &&
                         c->prekey_center[0] == c->center[0] && c->prekey_center[1] == c->center[1] &&
                         c->prekey_center[2] == c->center[2];
  c->prekey_valid = false;
  {
    bool handled = false;
    if ((rc = f->fused_step_append(c, dt, have_keys && !c->prekey_split, &handled))) return rc;
    if (handled) return EXP_AMD_OK;
  }
  {
    bool handled = false;
    if ((rc = f->fused_step_split(c, dt, have_keys && c->prekey_split, &handled))) return rc;
    if (handled) return EXP_AMD_OK;
  }
  const bool have_keys_whole = have_keys && !c->prekey_split && c->prekey_owner == (const void *)f && c->sorted_for == (const void *)f;
  if (c->n == 0) {
    if ((rc = f->determine_coefficients(c, false, 0.0, 0.0))) return rc;
  } else if ((rc = f->determine_coefficients(c, true, 0.5 * dt, dt, have_keys_whole))) return rc;
  bool done = false;
  // the closing half-kick is deferred: the next fused step's scatter pass applies it (as its own
  // rounding step) together with its opening half-kick; any other call applies it first
  if ((rc = f->accelerate(c, 0, true, 0.5 * dt, 0.5 * dt, dt, &done, /*defer_kick=*/c->n > 0))) return rc;
  f->firstime_coef = false;
  if (done) {
    c->prekey_valid = true;
    c->prekey_split = false;
    c->prekey_owner = f;
    c->prekey_epoch = f->ctx->force_epoch;
    c->prekey_dtk = 0.5 * dt;
    c->prekey_dtd = dt;
    for (int k = 0; k < 3; k++) c->prekey_center[k] = c->center[k];
  }
  return EXP_AMD_OK;
}

// ---- several fused steps, pairs of them replayed from a HIP graph ------------------------------------------------
// A fused step is ~12 launches (histogram, scan, scatter, memsets, accumulate, contraction, all-reduce, projection,
// two force passes); at 1e8 particles they hide behind 10 ms of kernels, at an 8-GPU share (1.25e7 per GPU, 1.4 ms per
// step) the gaps between them are 3-4 % of the step.  In steady state -- same dt, same centre, the force pass of step k
// has written the keys of step k+1 -- the host-side state of a step alternates with period two (the ping-pong buffer
// set of the store, the two work-list counters of the force pass): TWO consecutive steps captured once on the context's
// stream return the host to the state they started from, and the instantiated graph is replayed for every further
// pair, the RCCL all-reduce included (a host callback cannot be captured: such contexts step eagerly).  Anything that
// does not match the captured state -- another dt, a moved centre, an intervening call -- drops the graph.
static bool step_state_matches(const exp_amd_force *f, const exp_amd_comp *c, double dt)
{
  const exp_amd_force::StepGraph &g = f->step_graph;
  return g.exec && g.mutation == expamd_mutation_counter().load(std::memory_order_relaxed) && g.comp == c && g.dt == dt && g.n == c->n && g.cur == c->cur && g.parity == f->step_parity() &&
         g.prekick == (int)f->ctx->prekick && g.det == (int)f->ctx->deterministic && g.epoch == f->ctx->force_epoch &&
         g.pending == c->pending_kick && g.center[0] == c->center[0] && g.center[1] == c->center[1] &&
         g.center[2] == c->center[2];
}

static void step_graph_drop(exp_amd_force *f)
{
  if (f->step_graph.exec) (void)hipGraphExecDestroy(f->step_graph.exec);
  f->step_graph.exec = nullptr;
}

// the next step would take the key-reusing fast path of exp_amd_step_kdk (its steady state)
static bool step_is_steady(const exp_amd_force *f, const exp_amd_comp *c, double dt)
{
  return c->n > 0 && !f->multistep && c->prekey_valid && !c->prekey_split && !c->split &&
         c->prekey_owner == (const void *)f && c->prekey_epoch == f->ctx->force_epoch && c->prekey_dtk == 0.5 * dt &&
         c->prekey_dtd == dt && c->prekey_center[0] == c->center[0] && c->prekey_center[1] == c->center[1] &&
         c->prekey_center[2] == c->center[2] && c->sorted_for == (const void *)f && c->nlevels == 1;
}

extern "C" int exp_amd_step_kdk_n(exp_amd_force *f, exp_amd_comp *c, double dt, int nsteps)
{
  if (!f || !c || nsteps < 0) return expamd_fail(f ? f->ctx : nullptr, EXP_AMD_ERR_ARG, "step_kdk_n: bad argument");
  exp_amd_ctx *ctx = f->ctx;
  static const bool graphs_on = !(getenv("EXP_AMD_STEP_GRAPH") && atoi(getenv("EXP_AMD_STEP_GRAPH")) == 0);
  int rc, done = 0;
  while (done < nsteps) {
    // (a component the append step takes is stepped eagerly: that step reads a flag back after every pass)
    const bool app = ctx->append_min != 0 && c->n >= (size_t)(ctx->append_min < 0 ? -ctx->append_min : ctx->append_min);
    const bool can = graphs_on && nsteps - done >= 2 && !ctx->profile && !ctx->ar_fn && ctx->split_min <= 0 && !app &&
                     !f->step_graph.refused && f->step_graph_ok() && step_is_steady(f, c, dt);
    if (!can) {
      if ((rc = exp_amd_step_kdk(f, c, dt))) return rc;
      done++;
      continue;
    }
    if (!step_state_matches(f, c, dt)) {
      // capture two steps from this state; the host code of the steps runs as usual, its launches are recorded
      step_graph_drop(f);
      exp_amd_force::StepGraph g;
      g.comp = c; g.dt = dt; g.n = c->n; g.cur = c->cur; g.parity = f->step_parity();
      g.prekick = (int)ctx->prekick; g.det = (int)ctx->deterministic; g.epoch = ctx->force_epoch;
      g.pending = c->pending_kick;
      for (int k = 0; k < 3; k++) g.center[k] = c->center[k];
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
      int r1 = EXP_AMD_OK, r2 = EXP_AMD_OK;
      if (e == hipSuccess) {
        r1 = exp_amd_step_kdk(f, c, dt);
        if (!r1) r2 = exp_amd_step_kdk(f, c, dt);
        e = hipStreamEndCapture(ctx->stream, &graph);
      }
      // two steps bring the host back to where it was (period two), whether or not the capture worked
      const bool back = g.cur == c->cur && g.parity == f->step_parity() && g.pending == c->pending_kick &&
                        step_is_steady(f, c, dt);
      if (e == hipSuccess && !r1 && !r2 && graph)
        e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
      else if (e == hipSuccess)
        e = hipErrorUnknown;
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess || !g.exec) {
        (void)hipGetLastError();
        f->step_graph.refused = true;        // eager from now on; nothing was executed, the two steps are still to do
        if (!back) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "step_kdk_n: the captured steps left the host state changed");
        continue;
      }
      if (!back) {
        // The pair did not start from the state a fused step leaves -- e.g. a read-only call in between had completed
        // the deferred closing half-kick (pending_kick 0 instead of dt/2 without the pre-kicked store) -- so it does
        // not return there either.  The recorded launches ARE these two steps, and the host has moved on as if they had
        // run: run them once, keep nothing; the next pair starts from a steady state and is captured afresh.
        // (found by tests/fuzz/fuzz_kdk.py; this used to be reported as an error)
        hipError_t le = hipGraphLaunch(g.exec, ctx->stream);
        if (le == hipSuccess) le = hipStreamSynchronize(ctx->stream);
        (void)hipGraphExecDestroy(g.exec);
        if (le != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "step_kdk_n: replay of the captured steps failed");
        done += 2;
        continue;
      }
      g.mutation = expamd_mutation_counter().load(std::memory_order_relaxed);
      f->step_graph = g;
      HIP_TRY(ctx, hipGraphLaunch(f->step_graph.exec, ctx->stream));      // (its all-reduces were counted while capturing)
      done += 2;
      continue;
    }
    HIP_TRY(ctx, hipGraphLaunch(f->step_graph.exec, ctx->stream));
    ctx->ar_calls += ctx->rccl_comm ? 2 : 0;
    done += 2;
  }
  return EXP_AMD_OK;
}
