// Host orchestration in C++: the step loop of EXP (do_step, src/step.cc:67-325; begin_run's
// initial expansion, src/begin.cc:80-129; ComponentContainer::compute_expansion /
// compute_potential, src/ComponentContainer.cc:1173-1226, :580-917) over a set of components,
// their self-gravity force methods and the pairwise interactions, driving the device entirely
// through the kernels of this library (no per-step host<->device particle traffic).
#include <chrono>
#include "force.h"
#include "sort_kernels.h"

#include <cmath>
#include <cstring>
#include <vector>

struct exp_amd_sim {
  exp_amd_ctx *ctx = nullptr;
  int multistep = 0, Mstep = 1;
  double dtime = 0.0, tnow = 0.0;
  double dynfrac[5] = {1000.0, 0.01, 0.01, 0.03, 0.05};   // src/global.cc:76-80 (D, V, S, A, P)
  int shiftlevl = 0;
  long long this_step = 0;
  std::vector<exp_amd_comp *> comps;
  std::vector<exp_amd_force *> forces;
  std::vector<std::pair<int, int>> inter;                   // (source, target)
  std::vector<int> mintvl, mfirst;
  long long last_switch = 0;
  long long step_switch = 0;        // level changes summed over the last exp_amd_sim_step call
  // EJ centre (Component::orient, EJdryrun; global centerlevl, src/global.cc:68)
  std::vector<exp_amd_orient *> orients;
  std::vector<int> ej_dryrun;
  std::vector<int> center_from;    // "ctr_name" (Component::c0, src/Component.cc:284-310, :3585-3587): the component whose centre
                                   // this one takes at every fix_positions, or -1
  int centerlevl = -1;
  bool gottapot = false;
  bool restart = false;            // the global `restart` (src/global.cc): the estimators take in the first state too
  bool eqmotion = true;            // the global `eqmotion` (src/global.cc:54): false = incr_position / incr_velocity return at once
                                   // (src/incpos.cc:75, src/incvel.cc:93): fields, levels and the time go on, nothing moves
  unsigned long long *pinned = nullptr;   // page-locked landing area of the per-sub-step read-back
  size_t pinned_cap = 0;                  // (components it has room for)
  unsigned long long *pinned_dev = nullptr;   // the device's address of it (k_kick_adjust's last block writes there)
  unsigned long long pub_seq = 0;         // sequence number of the counters last handed over that way
  // Two-stream sub-steps: everything that touches the particles of component k is issued on stream
  // k & 1 (the context's stream / its auxiliary stream).  The small launches of a sub-step are
  // latency-bound, so the two components' chains fill each other's gaps.  Events carry the cross
  // dependencies: ev_self[k] = force method k has projected its tables and applied its self force;
  // ev_coef[k] = its combined coefficient set is complete (all a THIN cross force reads: it then need not sit out the
  // source's self force and the latency of an event that has only just been recorded);
  // ev_used[k] = the last use of force method k's tables by a cross force on the other stream.
  // Component::Adiabatic (the ton / toff / twid keys, src/Component.cc:1040-1055, :4214-4220) per component: the driver
  // evaluates it at tnow before every accumulation and differencing (exp_amd_sim_set_adiabatic)
  struct Adiabatic { bool on = false; double ton = -1.0e20, toff = 1.0e20, twid = 0.1; };
  std::vector<Adiabatic> adb;
  bool overlap = false;
  hipStream_t main_stream = nullptr;
  std::vector<hipEvent_t> ev_self, ev_used, ev_coef;
  std::vector<char> used_pending;
  hipEvent_t ev_join = nullptr;
};

static inline double host_now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// double Component::Adiabatic() (src/Component.cc:4214-4220) at the driver's tnow -> the force's mass factor
static void set_mass_scale(exp_amd_sim *s, size_t k)
{
  const exp_amd_sim::Adiabatic &a = s->adb[k];
  const double v = !a.on ? 1.0 : 0.25 * (1.0 + erf((s->tnow - a.ton) / a.twid)) * (1.0 + erf((a.toff - s->tnow) / a.twid));
  if (v != s->forces[k]->mass_scale) expamd_mutated();       // (a captured graph of fused steps holds the old factor)
  s->forces[k]->mass_scale = v;
}

// issue on component k's stream for the lifetime of the object
struct StreamOf {
  exp_amd_ctx *ctx;
  hipStream_t saved;
  StreamOf(exp_amd_sim *s, size_t k) : ctx(s->ctx), saved(s->ctx->stream)
  {
    if (s->overlap) ctx->stream = (k & 1) ? ctx->aux : s->main_stream;
  }
  ~StreamOf() { ctx->stream = saved; }
};

static int overlap_begin(exp_amd_sim *s)
{
  exp_amd_ctx *ctx = s->ctx;
  bool any_orient = false;
  for (auto o : s->orients) any_orient = any_orient || o;
  // EXP_AMD_SIM_OVERLAP=0 (include/exp_amd.h, environment): both components on the context's one stream
  const char *env = getenv("EXP_AMD_SIM_OVERLAP");
  // (several ranks: each stream needs a transport of its own -- a host callback is handed the stream; the library's RCCL
  // communicator is split into a second one for the auxiliary stream, expamd_comm_prepare_two_streams, a collective -- and every rank issues
  // the collectives of each stream in the same order: the launch order below does not depend on the data)
  // (exactly two components: the stream of a launch is the parity of its TARGET, and a force method is
  // followed across the streams by ONE pair of events -- with a third component the cross forces of one
  // source on two targets would run on both streams at once and share the scratch of its force pass)
  s->overlap = s->multistep > 0 && s->comps.size() == 2 && !any_orient && !(env && atoi(env) == 0) &&
               expamd_comm_prepare_two_streams(ctx);
  if (!s->overlap) return EXP_AMD_OK;
  int rc = expamd_ctx_aux(ctx);
  if (rc) return rc;
  s->main_stream = ctx->stream;
  while (s->ev_self.size() < s->comps.size()) {
    hipEvent_t a, b, c;
    HIP_TRY(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&c, hipEventDisableTiming));
    s->ev_self.push_back(a);
    s->ev_used.push_back(b);
    s->ev_coef.push_back(c);
    s->used_pending.push_back(0);
  }
  if (!s->ev_join) HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  // whatever was issued on the context's stream so far precedes the auxiliary stream's work
  HIP_TRY(ctx, hipEventRecord(s->ev_join, ctx->stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, s->ev_join, 0));
  return EXP_AMD_OK;
}

static int overlap_end(exp_amd_sim *s)
{
  if (!s->overlap) return EXP_AMD_OK;
  exp_amd_ctx *ctx = s->ctx;
  HIP_TRY(ctx, hipEventRecord(s->ev_join, ctx->aux));
  HIP_TRY(ctx, hipStreamWaitEvent(s->main_stream, s->ev_join, 0));
  for (auto &u : s->used_pending) u = 0;       // (main now follows everything)
  s->overlap = false;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sim_create(exp_amd_ctx *ctx, int multistep, double dtime,
                                  const double dynfrac[5], int shiftlevl, exp_amd_sim **out)
{
  if (!ctx || !out || multistep < 0 || multistep > 16 || !(dtime > 0.0))
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sim_create: bad argument");
  exp_amd_sim *s = new exp_amd_sim;
  s->ctx = ctx;
  s->multistep = multistep;
  s->dtime = dtime;
  s->shiftlevl = shiftlevl;
  if (dynfrac) for (int k = 0; k < 5; k++) s->dynfrac[k] = dynfrac[k];
  // initialize_multistep (src/multistep.cc:630-680)
  s->Mstep = 1 << multistep;
  s->mintvl.resize(multistep + 1);
  s->mintvl[0] = s->Mstep;
  for (int n = 1; n <= multistep; n++) s->mintvl[n] = s->mintvl[n - 1] / 2;
  s->mfirst.resize(s->Mstep + 1);
  for (int ms = 0; ms <= s->Mstep; ms++) {
    s->mfirst[ms] = 0;
    for (int M = 0; M <= multistep; M++)
      if (ms == 0 || ms % (1 << (multistep - M)) == 0) { s->mfirst[ms] = M; break; }
  }
  *out = s;
  return EXP_AMD_OK;
}

extern "C" void exp_amd_sim_destroy(exp_amd_sim *s)
{
  if (!s) return;
  if (s->pinned) (void)hipHostFree(s->pinned);
  for (auto e : s->ev_self) (void)hipEventDestroy(e);
  for (auto e : s->ev_used) (void)hipEventDestroy(e);
  for (auto e : s->ev_coef) (void)hipEventDestroy(e);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  delete s;
}

extern "C" int exp_amd_sim_add_component(exp_amd_sim *s, exp_amd_comp *c, exp_amd_force *f, int *index)
{
  if (!s || !c || !f) return expamd_fail(s ? s->ctx : nullptr, EXP_AMD_ERR_ARG, "sim_add_component: NULL");
  if (f->multistep != s->multistep)
    return expamd_fail(s->ctx, EXP_AMD_ERR_ARG, "sim_add_component: force multistep (%d) != sim (%d)",
                       f->multistep, s->multistep);
  s->comps.push_back(c);
  s->forces.push_back(f);
  s->orients.push_back(nullptr);
  s->ej_dryrun.push_back(0);
  s->center_from.push_back(-1);
  s->adb.emplace_back();
  if (index) *index = (int)s->comps.size() - 1;
  return EXP_AMD_OK;
}

// the force of component `source` also acts on component `target`
// (Interaction list of ComponentContainer, src/ComponentContainer.cc:785-853)
extern "C" int exp_amd_sim_add_interaction(exp_amd_sim *s, int source, int target)
{
  if (!s || source < 0 || target < 0 || source >= (int)s->comps.size() ||
      target >= (int)s->comps.size() || source == target)
    return expamd_fail(s ? s->ctx : nullptr, EXP_AMD_ERR_ARG, "sim_add_interaction: bad index");
  s->inter.emplace_back(source, target);
  return EXP_AMD_OK;
}

// The component's adiabatic turn-on / turn-off (keys ton, toff, twid of a component, src/Component.cc:1040-1055): from now
// on the driver multiplies the masses its force method accumulates and differences by Component::Adiabatic() at tnow
extern "C" int exp_amd_sim_set_adiabatic(exp_amd_sim *s, int index, double ton, double toff, double twid)
{
  if (!s || index < 0 || index >= (int)s->comps.size() || !(twid > 0.0))
    return expamd_fail(s ? s->ctx : nullptr, EXP_AMD_ERR_ARG, "sim_set_adiabatic: bad index or width");
  s->adb[index].on = true;
  s->adb[index].ton = ton; s->adb[index].toff = toff; s->adb[index].twid = twid;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sim_set_time(exp_amd_sim *s, double tnow)
{
  if (!s) return EXP_AMD_ERR_ARG;
  s->tnow = tnow;
  return EXP_AMD_OK;
}

// Attach an Orient to a component (Component::initialize, src/Component.cc:1323-1370: the EJ keys);
// centerlevl < 0 selects multistep/2 (src/ComponentContainer.cc:42-45).
extern "C" int exp_amd_sim_set_restart(exp_amd_sim *s, int on)
{
  if (!s) return EXP_AMD_ERR_ARG;
  s->restart = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sim_set_orient(exp_amd_sim *s, int index, exp_amd_orient *o, int dryrun,
                                      int centerlevl)
{
  if (!s || index < 0 || index >= (int)s->comps.size())
    return expamd_fail(s ? s->ctx : nullptr, EXP_AMD_ERR_ARG, "sim_set_orient: bad index");
  s->orients[index] = o;
  s->ej_dryrun[index] = dryrun;
  s->centerlevl = centerlevl < 0 ? s->multistep / 2 : (centerlevl > s->multistep ? s->multistep : centerlevl);
  return EXP_AMD_OK;
}

// The centre part of ComponentContainer::compute_potential (src/ComponentContainer.cc:955-959) for
// the components that carry an Orient: Component::fix_positions zeroes the centre and adds the
// estimator's current one unless it holds a NaN (src/Component.cc:3357, :3569-3582), THEN the
// estimator takes in the present state (ComponentContainer::fix_positions :1386-1389, only once
// potentials exist) -- so the expansion centre lags the estimate by one call, as in the reference.
static int fix_centers(exp_amd_sim *s, int mstep)
{
  bool any = false;
  for (auto o : s->orients) any = any || o;
  for (int src : s->center_from) any = any || src >= 0;
  if (!any) return EXP_AMD_OK;
  const int cl = s->centerlevl < 0 ? s->multistep / 2 : s->centerlevl;
  const bool active = mstep == 0 || mstep % (1 << (s->multistep - cl)) == 0;     // mactive[mstep][centerlevl]
  if (!active) return EXP_AMD_OK;
  for (size_t k = 0; k < s->comps.size(); k++) {
    exp_amd_orient *o = s->orients[k];
    // "Alternative center" (src/Component.cc:3584-3587): the centre of the component named by ctr_name, as it stands when this
    // component's turn comes (components are visited in their order: a source further down the list still has last call's)
    if (s->center_from[k] >= 0) {
      if (!o) {
        int rc = exp_amd_comp_set_center(s->comps[k], s->comps[(size_t)s->center_from[k]]->center);
        if (rc) return rc;
        continue;
      }
    }
    if (!o) continue;
    double ctr[3], center[3] = {0.0, 0.0, 0.0};
    int rc = exp_amd_orient_get(o, ctr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (!s->ej_dryrun[k] && (exp_amd_orient_flags(o) & 2u) &&            // EJ & Orient::CENTER
        !(std::isnan(ctr[0]) || std::isnan(ctr[1]) || std::isnan(ctr[2])))
      for (int i = 0; i < 3; i++) center[i] += ctr[i];
    if (s->center_from[k] >= 0)                                            // (c0 overrides the estimator's centre, :3585)
      for (int i = 0; i < 3; i++) center[i] = s->comps[(size_t)s->center_from[k]]->center[i];
    if ((rc = exp_amd_comp_set_center(s->comps[k], center))) return rc;
    if (!s->ej_dryrun[k] && (exp_amd_orient_flags(o) & 1u)) {            // EJ & Orient::AXIS
      double body[9];
      if ((rc = exp_amd_orient_get(o, nullptr, nullptr, body, nullptr, nullptr))) return rc;
      if ((rc = exp_amd_comp_set_orientation(s->comps[k], body))) return rc;
    }
    // std::tie(accel, omega, domdt) = orient->currentAccel() sits inside `if ((EJ & CENTER) &&
    // !EJdryrun)` (src/Component.cc:3569-3571): with AXIS alone the three stay zero and
    // getPseudoAccel returns nothing; with CENTER the rotating-frame terms join when AXIS is set too
    if (!s->ej_dryrun[k] && (exp_amd_orient_flags(o) & 2u)) {
      double acc[3], om[3], dom[3];
      const unsigned fl = exp_amd_orient_flags(o);
      if ((rc = exp_amd_orient_accel(o, acc, om, dom))) return rc;
      if ((rc = exp_amd_comp_set_pseudo_accel(s->comps[k], acc, (fl & 1u) ? om : nullptr,
                                              (fl & 1u) ? dom : nullptr))) return rc;
    }
    if (s->gottapot || s->restart) {                                       // (:1386-1389)
      if ((rc = exp_amd_orient_accumulate(o, s->tnow, s->dtime, s->comps[k]))) return rc;
      if (expamd_orient_has_log(o)) {                                      // Orient::logEntry(tnow, c)
        double cm[10];
        if ((rc = exp_amd_comp_fix_positions(s->comps[k], 0, cm))) return rc;
        if ((rc = exp_amd_orient_log_entry(o, s->tnow, cm + 1, nullptr))) return rc;
      }
    }
  }
  return EXP_AMD_OK;
}

// ComponentContainer::compute_expansion(M)
static int compute_expansion(exp_amd_sim *s, int M)
{
  for (size_t k = 0; k < s->comps.size(); k++) {
    int rc = exp_amd_force_set_level(s->forces[k], M);
    if (rc) return rc;
    if (s->forces[k]->frozen()) continue;                      // "self_consistent: false" after begin_run
    set_mass_scale(s, k);
    if ((rc = s->forces[k]->determine_coefficients(s->comps[k], false, 0.0, 0.0))) return rc;
    s->forces[k]->firstime_coef = false;
  }
  return EXP_AMD_OK;
}

// ComponentContainer::compute_potential(mlevel): zero, self forces, interactions
static int compute_potential(exp_amd_sim *s, int mlevel, int mdrft, int mstep)
{
  int rc;
  if ((rc = fix_centers(s, mstep))) return rc;
  for (auto c : s->comps)
    if ((rc = exp_amd_comp_zero_acc(c, mlevel))) return rc;
  for (size_t k = 0; k < s->comps.size(); k++) {
    exp_amd_force *f = s->forces[k];
    if ((rc = exp_amd_force_set_level(f, mlevel))) return rc;
    // determine_acceleration_and_potential: use_external == false branch
    // (src/SphericalBasis.cc:1680-1685, src/Cylinder.cc:1469)
    if (s->multistep && (rc = exp_amd_force_compute_multistep_coefficients(f, mdrft))) return rc;
    if ((rc = f->accelerate(s->comps[k], 0, false, 0.0))) return rc;
  }
  for (auto &pr : s->inter) {
    exp_amd_force *f = s->forces[pr.first];
    if ((rc = f->accelerate(s->comps[pr.second], 1, false, 0.0))) return rc;
  }
  s->gottapot = true;
  return EXP_AMD_OK;
}

// ---- block multistep, level-fused ------------------------------------------------------------------
// The active levels of a sub-step are a suffix [mfirst[mstep], multistep] of the level list, i.e. ONE
// contiguous slot range of the (level, cell)-ordered store, so every phase of do_step's sub-step
// (src/step.cc:115-231) is issued once per component over that range instead of once per level, and
// the host looks at the device once per sub-step (the level changes of all components).

// force method k's coefficient sets / tables / in-cut mass are about to change on the current stream:
// the cross forces that read them on the other stream come first
static int wait_used(exp_amd_sim *s, size_t k)
{
  if (s->overlap && s->used_pending[k]) {
    HIP_TRY(s->ctx, hipStreamWaitEvent(s->ctx->stream, s->ev_used[k], 0));
    s->used_pending[k] = 0;
  }
  return EXP_AMD_OK;
}

// first half: for M = mfirst[mstep] .. multistep: incr_velocity(DT(M)/2, M); incr_position(DT(M), M);
// compute_expansion(M)  (src/step.cc:126-160)
static int substep_expansion(exp_amd_sim *s, int lo, double dt_min, int mdrft)
{
  // (alone: the combined coefficient set of the force evaluation that follows is formed by the same kernel that sums
  // the per-level sets; with several ranks the all-reduce of the level block sits between the two)
  const bool alone = s->ctx->nranks <= 1 && !s->ctx->ar_fn && !s->ctx->rccl_comm;
  // two streams: the advance of every component is issued before the accumulation of any -- the host needs ~25 us for
  // the launches of one component's first half, and the other stream would have nothing to do meanwhile
  for (int phase = s->overlap ? 1 : 0; phase <= (s->overlap ? 2 : 0); phase++)
    for (size_t k = 0; k < s->comps.size(); k++) {
      StreamOf on(s, k);
      int rc = phase == 2 ? 0 : wait_used(s, k);
      if (rc) return rc;
      // coefficients held fixed ("self_consistent: false" once begin_run is over): the advance alone
      const bool frz = s->forces[k]->frozen();
      if (frz && phase == 2) continue;
      set_mass_scale(s, k);
      rc = s->forces[k]->substep_expansion(s->comps[k], lo, dt_min, alone ? mdrft : -1, frz ? 1 : phase);
      if (rc) return rc;
      if (!frz && phase != 1) s->forces[k]->firstime_coef = false;
    }
  return EXP_AMD_OK;
}

// ComponentContainer::compute_potential(mlevel): the self force of a component is evaluated first
// and ASSIGNS acc / pot of its levels >= mlevel (the reference zeroes them and adds, src/
// ComponentContainer.cc:641-716: same values, one pass less); the interactions then add.
// join_follows: the sweep behind this force evaluation ends in a read-back that synchronises both streams; false for the
// sweeps that cannot move anything (kick_adjust_levels): the cross forces then leave their 'tables used' events, so that
// the NEXT sub-step's first half on the other stream waits for them before it rewrites those tables.
static int compute_potential_ms(exp_amd_sim *s, int mlevel, int mdrft, int mstep, bool join_follows = true)
{
  int rc;
  if ((rc = fix_centers(s, mstep))) return rc;
  for (size_t k = 0; k < s->comps.size(); k++) {
    StreamOf on(s, k);
    exp_amd_force *f = s->forces[k];
    if ((rc = exp_amd_force_set_level(f, mlevel))) return rc;
    if (f->combined_mdrft != mdrft && (rc = exp_amd_force_compute_multistep_coefficients(f, mdrft))) return rc;
    f->combined_mdrft = -1;
    if (s->overlap) HIP_TRY(s->ctx, hipEventRecord(s->ev_coef[k], s->ctx->stream));
    if ((rc = f->accelerate(s->comps[k], 0, /*assign=*/true, 0.0))) return rc;
    if (s->overlap) HIP_TRY(s->ctx, hipEventRecord(s->ev_self[k], s->ctx->stream));
  }
  for (auto &pr : s->inter) {
    StreamOf on(s, (size_t)pr.second);            // the target's particles: the target's stream
    exp_amd_force *f = s->forces[pr.first];
    const bool foreign = s->overlap && ((pr.first ^ pr.second) & 1);
    if (foreign) {
      // the source's projected tables (and the scratch of its force pass) must be ready and free -- or, for a target range
      // the force method evaluates straight from its coefficient set (the same test as in its accelerate()), that set
      exp_amd_comp *t = s->comps[pr.second];
      bool thin = f->multistep > 0 && t->n && t->nlevels > 1 && !s->ctx->deterministic && s->ctx->thin_max > 0 &&
                  !f->accel_writes_coef;
      if (thin) {
        size_t nthin = 0;
        if ((rc = expamd_comp_level_count(t, mlevel, t->nlevels - 1, &nthin))) return rc;
        thin = (long long)nthin <= s->ctx->thin_max;
      }
      // (... and a table-path evaluation projects the tables ITSELF, on this stream, when the source's self force -- issued
      // above -- left them stale, i.e. was thin: the coefficient set is all it waits for then, too)
      const bool early = thin || (f->proj_dirty && !f->accel_writes_coef);
      HIP_TRY(s->ctx, hipStreamWaitEvent(s->ctx->stream, early ? s->ev_coef[pr.first] : s->ev_self[pr.first], 0));
      if (s->used_pending[pr.first]) HIP_TRY(s->ctx, hipStreamWaitEvent(s->ctx->stream, s->ev_used[pr.first], 0));
    }
    if ((rc = f->accelerate(s->comps[pr.second], 1, false, 0.0))) return rc;
    // (the tables this cross force reads are next written by the source's stream in the NEXT sub-step's first half, and
    // every path there leads through a join of the two streams: the read-back of kick_adjust_levels synchronises both, and
    // begin_run / the end of a step call end in overlap_end.  An event per cross force for it cost each stream ~5 us a
    // sub-step, tools/dbg/launch_gap.hip: they are recorded only where no join follows)
    if (foreign && !join_follows) {
      HIP_TRY(s->ctx, hipEventRecord(s->ev_used[pr.first], s->ctx->stream));
      s->used_pending[pr.first] = 1;
    }
  }
  s->gottapot = true;
  return EXP_AMD_OK;
}

// A sweep that examines the TOP level only cannot move anything: the proposals of adjust_multistep_level are clamped to
// [mfirst[mdrft], multistep] (src/multistep.cc:188-196: `nlev = std::max<int>(nlev, mfirst[mdrft])` after the cap at multistep), and for odd
// mdrft that interval is the single level `multistep` its particles are already on.  Half the sweeps of a master step
// are of this kind: they reduce to the closing kick, with no counters, no read-back and no wait.
static bool sweep_is_noop(const exp_amd_sim *s, int mdrft, int first_step)
{
  // (a "noswitch" component: every sweep takes its particles' time steps into Particle::dtreq, whether it can move them or not)
  for (const exp_amd_comp *c : s->comps) if (c->noswitch) return false;
  return s->multistep > 0 && !first_step && s->mfirst[mdrft] == s->multistep;
}

// second half: incr_velocity(dt*mintvl[M]/2, M) for M >= mfirst[mdrft] (src/step.cc:198-203; not in
// begin_run) fused with adjust_multistep_level() (src/multistep.cc:344-627) for every component:
// time-step criteria -> proposed levels, one read-back of {changes, new level populations} for all
// components, then multistep_update / _finish, the commit and the re-ordering of the examined slots.
static int kick_adjust_levels(exp_amd_sim *s, int mdrft, int first_step, bool kick)
{
  exp_amd_ctx *ctx = s->ctx;
  const int ms = s->multistep;
  const int mf = s->mfirst[mdrft];
  const int first = first_step ? 0 : mf;            // src/multistep.cc:451-453
  const size_t nc = s->comps.size();
  const double dt_min = s->eqmotion ? s->dtime / s->Mstep : 0.0;      // (the kicks' step: zero moves nothing)
  if (s->pinned_cap < nc) {
    if (s->pinned) (void)hipHostFree(s->pinned);
    s->pinned = nullptr;
    // (40 words a component: 32 counters, the sequence number, padding; coherent = visible to the polling host while
    // the stream runs on)
    HIP_TRY(ctx, hipHostMalloc((void **)&s->pinned, nc * 40 * sizeof(unsigned long long),
                               hipHostMallocMapped | hipHostMallocCoherent));
    memset(s->pinned, 0, nc * 40 * sizeof(unsigned long long));
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&s->pinned_dev, s->pinned, 0));
    s->pinned_cap = nc;
  }
  int rc;
  // (sweep_is_noop: the closing kick alone -- the host goes straight on to the next sub-step's launches; the streams stay
  // ordered among themselves, nothing is differenced, committed or re-ordered because nothing changed)
  if (kick && sweep_is_noop(s, mdrft, first_step)) {
    for (size_t k = 0; k < nc; k++) {
      StreamOf on(s, k);
      exp_amd_comp *c = s->comps[k];
      // ... and the closing kick itself -- of the top level alone, DT(multistep)/2 = dt_min/2 -- is left to the advance
      // of the next sub-step, which applies it first, as its own rounding step (exp_amd_comp::pending_kick / pending_lo)
      if (c->nlevels == ms + 1 && c->pending_kick == 0.0) {
        // (an EMPTY top level owes nothing -- and must not: the next real sweep may move particles there, which would
        // then be kicked for a step they never took)
        size_t ntop = 0;
        if (c->n && (rc = expamd_comp_level_count(c, ms, ms, &ntop))) return rc;
        if (ntop) { c->pending_kick = 0.5 * dt_min; c->pending_lo = ms; }
        continue;
      }
      const unsigned long long *res = nullptr;
      if ((rc = expamd_comp_kick_adjust(c, s->dtime, s->dynfrac, s->shiftlevl, ms, mf, mf, /*first=*/ms + 1,
                                        dt_min, &res))) return rc;
    }
    s->last_switch = 0;
    return EXP_AMD_OK;
  }
  std::vector<unsigned long long> want(nc, 0ull);
  for (size_t k = 0; k < nc; k++) {
    StreamOf on(s, k);
    const unsigned long long *res = nullptr;
    bool launched = false;
    if (((++s->pub_seq) & 0xffffffull) == 0) ++s->pub_seq;       // (the tag of a word never written)
    const unsigned long long seq = s->pub_seq & 0xffffffull;
    // The sweep that closes a master step examines every slot and is followed by sub-step 0 of the next one, in which every
    // level is advanced: it writes that sub-step's sort keys on the way (the force method's instantiation of the kernel,
    // kick_adjust.h), and the sort counts those instead of reading x, v, a once more (73 B a slot against 4 + 28 here).
    exp_amd_comp *ck = s->comps[k];
    exp_amd_force *fk = s->forces[k];
    ka_launch_fn kfn = nullptr;
    void *kself = nullptr;
    // "freezeL" (Component::FreezeLev, src/multistep.cc:158, :534): after the first call this component's sweep examines no
    // level -- the closing kick alone, counters of zero
    // (firstCall = this_step == 0 and mdrft == 0: begin_run's assignment; the first sub-step of the run examines every level
    // too -- first_step -- but is not the first call)
    const int first_k = (ck->freeze_levels && !(first_step && mdrft == 0)) ? ms + 1 : first;
    // "noswitch" (src/multistep.cc:136-147): mstep = mdrft - 1 at do_step's call; begin_run's call is the firstCall
    ck->ns_reset = ((ck->dtreset && mdrft == 1) || (first_step && mdrft == 0)) ? 1 : 0;
    ck->ns_apply = (mdrft == s->Mstep || (first_step && mdrft == 0)) ? 1 : 0;
    const bool closing = kick && !first_step && mdrft == s->Mstep && first_k == 0 && !s->orients[k] && ck->nlevels == ms + 1 &&
                         ck->pending_kick == 0.0 && EXPAMD_EXPT("EXP_AMD_MS_PREKEY", 1) != 0;
    if (closing && !fk->prekey_launcher(ck, &kfn, &kself)) kfn = nullptr;
    if ((rc = expamd_comp_kick_adjust(ck, s->dtime, s->dynfrac, s->shiftlevl, ms, mf,
                                      kick ? mf : ms + 1, first_k, dt_min, &res, s->pinned_dev + k * 40, seq, &launched, /*build_list=*/true,
                                      kfn, kself))) return rc;
    if (kfn && launched && ck->mprekey_n == ck->n) {
      ck->mprekey_valid = true;
      ck->mprekey_owner = (const void *)fk;
      ck->mprekey_epoch = ctx->force_epoch;
      ck->mprekey_dt_min = dt_min;
      for (int q = 0; q < 3; q++) ck->mprekey_center[q] = ck->center[q];
    }
    if (launched) want[k] = seq;
    else {
      // nothing in the examined range: no counts, and no sweep on this component's stream to wait for -- but this
      // function IS the join of the two streams that compute_potential_ms(join_follows = true) relies on (the cross
      // forces before it left no 'tables used' events), so the stream itself is waited for; it is all but idle
      memset(s->pinned + k * 40, 0, 32 * sizeof(unsigned long long));
      if (s->overlap) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  // poll the tags; the streams themselves are asked every ~100 us, so that a failed launch ends the wait
  for (size_t k = 0; k < nc; k++) {
    if (!want[k]) continue;
    unsigned long long *w = s->pinned + k * 40;
    auto all_there = [&] {
      for (int q = 0; q < 32; q++) if ((__atomic_load_n(w + q, __ATOMIC_RELAXED) >> 40) != want[k]) return false;
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return true;
    };
    double t_query = host_now() + 100e-6;
    while (!all_there()) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      asm volatile("yield");
#endif
      if (host_now() < t_query) continue;
      t_query = host_now() + 100e-6;
      StreamOf on(s, k);
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) {                 // (everything ran: the words are there, or never will be)
        if (all_there()) break;
        return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sim_step: the level counters did not arrive");
      }
      if (e != hipErrorNotReady) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sim_step: %s", hipGetErrorString(e));
    }
  }
  for (size_t k = 0; k < nc; k++)
    if (want[k]) for (int q = 0; q < 32; q++) s->pinned[k * 40 + q] &= 0xffffffffffull;
  s->last_switch = 0;
  for (size_t k = 0; k < nc; k++) {
    StreamOf on(s, k);
    exp_amd_comp *c = s->comps[k];
    exp_amd_force *f = s->forces[k];
    const unsigned long long *res = s->pinned + k * 40;
    const unsigned long long u = res[0];
    s->last_switch += (long long)u;
    s->step_switch += (long long)u;
    // (collective: with several ranks every rank takes part even if it has no mover)
    c->mover_hint = (long long)u;
    // (not self-consistent: Cylinder::multistep_update returns at once, src/Cylinder.cc:1755; the sphere's only feeds
    // level sets that nothing reads again)
    if ((ctx->nranks > 1 || ctx->ar_fn || u) && f->self_consistent) {
      set_mass_scale(s, k);                                    // (tnow is the END of the sub-step here, as in the reference)
      if ((rc = f->multistep_update(c, first, mf))) return rc;
    }
    c->mover_hint = -1;
    c->mover_list_built = false;
    if (u) {
      const bool ordered = c->sorted_for == (const void *)f && c->nlevels == ms + 1;
      const bool mirror = ordered && c->lev_host_valid;
      uint32_t off[66];
      if (mirror) {
        // new level offsets: the levels below `first` were not examined, the proposals fill the rest
        for (int L = 0; L <= first; L++) off[L] = c->lev_host[L];
        for (int L = first; L <= ms; L++) off[L + 1] = off[L] + (uint32_t)res[1 + L];
      }
      // (the commit itself is left to the sort that settles the partition when that is the next thing to touch these
      // slots -- see below: it reads the proposed levels where they are, exp_amd_comp::commit_pending)
      const int next_lo_ = mdrft == s->Mstep ? 0 : s->mfirst[mdrft];
      const bool put_off = mirror && next_lo_ == first && c->n > 0;
      if (put_off) {
        c->commit_pending = true;
        c->commit_beg = (size_t)c->lev_host[first];
        c->levels_zero = false;
        c->sorted_for = nullptr;
      } else if ((rc = expamd_comp_commit_levels(c, mirror ? (size_t)c->lev_host[first] : 0))) return rc;
      if (mirror) {
        // the populations the re-ordering is about to establish decide which levels stay cell-sorted
        for (int L = 0; L <= ms + 1; L++) c->lev_host[L] = off[L];
        expamd_comp_update_sparse(c, first, ctx->dense_min >= 0 ? ctx->dense_min : f->sparse_threshold());
      } else c->sparse_mask = 0;
      // The next sub-step advances the levels >= mfirst[its mstep] = mfirst[mdrft] (0 after the last sweep of a
      // master step) -- the levels this sweep examined, unless it was the first of the run (first = 0 then): its
      // advance sort re-partitions them by the committed levels in the same pass (substep_expansion), so the
      // re-ordering is not done twice.
      const int next_lo = mdrft == s->Mstep ? 0 : s->mfirst[mdrft];
      if (mirror && next_lo == first) {
        c->partition_stale = true;
        c->stale_lo = first;
        c->stale_for = (const void *)f;
        continue;
      }
      if ((rc = f->resort(c, ordered ? first : 0))) return rc;
      if (mirror) {
        for (int L = 0; L <= ms + 1; L++) c->lev_host[L] = off[L];
        c->lev_host_valid = true;
      }
    }
  }
  return EXP_AMD_OK;
}

// begin_run (src/begin.cc:80-129)
extern "C" int exp_amd_sim_init(exp_amd_sim *s)
{
  if (s) for (exp_amd_comp *c : s->comps) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!s) return EXP_AMD_ERR_ARG;
  int rc;
  // `initializing = true` ... `= false` around begin_run's expansions (src/begin.cc:80, :129)
  struct Init {
    exp_amd_sim *s;
    Init(exp_amd_sim *s_) : s(s_) { for (auto f : s->forces) f->initializing = true; }
    ~Init() { for (auto f : s->forces) f->initializing = false; }
  } init_scope(s);
  if (s->multistep) {
    if ((rc = overlap_begin(s))) return rc;
    for (size_t k = 0; k < s->forces.size(); k++) { StreamOf on(s, k); if ((rc = wait_used(s, k)) || (rc = s->forces[k]->multistep_reset())) return rc; }
    // for (M = 0 .. multistep) compute_expansion(M): every level, nothing advanced
    if ((rc = substep_expansion(s, 0, 0.0, 0))) return rc;
    if ((rc = compute_potential_ms(s, 0, 0, 0))) return rc;
    if ((rc = kick_adjust_levels(s, 0, 1, false))) return rc;
    for (size_t k = 0; k < s->forces.size(); k++) { StreamOf on(s, k); if ((rc = wait_used(s, k)) || (rc = s->forces[k]->multistep_reset())) return rc; }
    if ((rc = substep_expansion(s, 0, 0.0, 0))) return rc;
    if ((rc = compute_potential_ms(s, 0, 0, 0))) return rc;
    return overlap_end(s);
  }
  if ((rc = compute_expansion(s, 0))) return rc;
  return compute_potential(s, 0, 0, 0);
}

// do_step (src/step.cc:67-325)
extern "C" int exp_amd_sim_step(exp_amd_sim *s, int nsteps)
{
  if (s) s->step_switch = 0;
  if (s) for (exp_amd_comp *c : s->comps) {
    // (a level partition left stale by the previous master step is this loop's own business)
    int rc_ = s->multistep ? expamd_comp_touch_keep_partition(c) : expamd_comp_touch(c);
    if (rc_) return rc_;
  }
  if (!s || nsteps < 0) return EXP_AMD_ERR_ARG;
  int rc;
  if ((rc = overlap_begin(s))) return rc;
  for (int it = 0; it < nsteps; it++) {
    if (s->multistep) {
      // comp->multistep_reset() (src/step.cc:84)
      for (size_t k = 0; k < s->forces.size(); k++) { StreamOf on(s, k); if ((rc = wait_used(s, k)) || (rc = s->forces[k]->multistep_reset())) return rc; }
      const double dt = s->dtime / s->Mstep;
      for (int mstep = 0; mstep < s->Mstep; mstep++) {
        const int mdrft = mstep + 1;
        if ((rc = substep_expansion(s, s->mfirst[mstep], s->eqmotion ? dt : 0.0, mdrft))) return rc;
        s->tnow += dt;
        const int first_step = (s->this_step == 0 && mstep == 0) ? 1 : 0;
        if ((rc = compute_potential_ms(s, s->mfirst[mstep], mdrft, mstep, !sweep_is_noop(s, mdrft, first_step)))) return rc;
        if ((rc = kick_adjust_levels(s, mdrft, first_step, true))) return rc;
      }
    } else if (s->comps.size() == 1 && s->inter.empty() && !s->orients[0] && s->eqmotion) {
      s->tnow += s->dtime;
      set_mass_scale(s, 0);            // (Component::Adiabatic at the time of this step's accumulation, as compute_expansion has it)
      if ((rc = exp_amd_step_kdk(s->forces[0], s->comps[0], s->dtime))) return rc;
    } else {
      s->tnow += s->dtime;
      for (auto c : s->comps) {
        if (!s->eqmotion) break;
        if ((rc = exp_amd_comp_kick(c, 0.5 * s->dtime, -1))) return rc;
        if ((rc = exp_amd_comp_drift(c, s->dtime, -1))) return rc;
      }
      if ((rc = compute_expansion(s, 0))) return rc;
      if ((rc = compute_potential(s, 0, 1, 0))) return rc;
      for (auto c : s->comps) {
        if (!s->eqmotion) break;
        if ((rc = exp_amd_comp_kick(c, 0.5 * s->dtime, -1))) return rc;
      }
    }
    s->this_step++;
  }
  return overlap_end(s);
}

extern "C" int exp_amd_sim_set_center_from(exp_amd_sim *s, int index, int source)
{
  if (!s || index < 0 || (size_t)index >= s->comps.size() || source >= (int)s->comps.size() || source == index)
    return expamd_fail(s ? s->ctx : nullptr, EXP_AMD_ERR_ARG, "sim_set_center_from: component index out of range");
  s->center_from[(size_t)index] = source < 0 ? -1 : source;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sim_set_eqmotion(exp_amd_sim *s, int on)
{
  if (!s) return EXP_AMD_ERR_ARG;
  s->eqmotion = on != 0;
  return EXP_AMD_OK;
}

extern "C" double exp_amd_sim_time(const exp_amd_sim *s) { return s ? s->tnow : 0.0; }
extern "C" long long exp_amd_sim_last_switches(const exp_amd_sim *s) { return s ? s->last_switch : 0; }
extern "C" long long exp_amd_sim_step_switches(const exp_amd_sim *s) { return s ? s->step_switch : 0; }
