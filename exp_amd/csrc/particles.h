// Particle store of one Component: fp64 SoA in HBM, ordered by (level, basis cell).
#pragma once
#include "common.h"

// How a sort pass advances the particles on the way (sort_kernels.h: advance_one).
//   mode 0: not at all;  1: kick dt_kick, drift dt_drift (fused single-level step);
//   2: the first half of a block-multistep sub-step, per level M: kick DT(M)/2, drift DT(M),
//      DT(M) = dt_min * 2^(multistep - M)  (src/step.cc:115-160)
struct AdvSpec {
  int mode = 0;
  double dt_kick = 0.0, dt_drift = 0.0, dt_min = 0.0;
  int multistep = 0;
  int lev_lo = 0;        // mode 2: levels below are passed through unadvanced (a full re-partition of a sub-step
                         // whose active levels start at lev_lo also moves the inactive ones)
  static AdvSpec none() { return AdvSpec{}; }
  static AdvSpec step(bool on, double dtk, double dtd) { AdvSpec a; a.mode = on ? 1 : 0; a.dt_kick = dtk; a.dt_drift = dtd; return a; }
  static AdvSpec levels(double dt_min_, int ms, int lo = 0) { AdvSpec a; a.mode = 2; a.dt_min = dt_min_; a.multistep = ms; a.lev_lo = lo; return a; }
};

enum { A_X = 0, A_Y, A_Z, A_VX, A_VY, A_VZ, A_M, A_AX, A_AY, A_AZ, A_POT, A_NARR };

struct exp_amd_comp {
  exp_amd_ctx *ctx = nullptr;
  size_t n = 0;
  int cur = 0;                       // which buffer set holds the live data
  DevBuf<double> arr[2][A_NARR];     // ping-pong sets (scatter target = 1-cur)
  DevBuf<uint32_t> id[2];            // original (caller) index of each slot
  DevBuf<uint8_t> level[2];          // multistep level of each slot
  DevBuf<uint32_t> key;              // sort key scratch
  DevBuf<uint8_t> newlev;            // level chosen by the last adjust_multistep_level sweep
  DevBuf<unsigned long long> nswitch; // counters of the level sweeps: two alternating sets [32] + one word
  int nsw_flip = 0;
  // Level-change differencing of many movers (force methods' multistep_update): the step driver leaves the number of
  // level changes of the last sweep here (< 0: unknown), expamd_comp_mover_list compacts their slots
  long long mover_hint = -1;
  DevBuf<uint32_t> mover_list, mover_cnt_buf; // slots of the movers; two {0, how many} pairs used alternately
  uint32_t *mover_cnt = nullptr;              // the pair of the last compaction: plays lev_off for the kernels
  int mover_flip = 0;
  bool mover_list_built = false;              // k_kick_adjust has compacted the last sweep's movers itself
  DevBuf<uint32_t> hist;             // histogram / cursors [nkeys+1]
  DevBuf<uint32_t> lev_off;          // [maxlev+2] start slot of every level (device)
  size_t hist_cap = 0;
  uint32_t sort_win = 0;       // LDS window of the NEXT full sort's passes (0: SORT_WIN); set by the force method's sort function
                               // for that one sort (SORT_WIN_DENSE: sort_kernels.h), taken and cleared by expamd_comp_finish_sort
  size_t hist_clean = 0;       // leading entries of `hist` known to be zero (a range sort's last kernel leaves the bins it
                               // used clean again: the next sort needs no memset)
  int nlevels = 1;                   // multistep + 1
  bool levels_zero = true;           // no slot has ever been given a level > 0 (both level arrays are 0)
  double center[3] = {0, 0, 0};
  // Component::rtrunc / com0 of Component::freeze (src/Component.cc:213, :4194-4202): a particle with
  // |pos - com0 - center| > rtrunc takes no part in any force method's accumulation, differencing or force pass
  // (exp_amd_comp_set_rtrunc; freeze_on == false: the default rtrunc of 1e20)
  bool freeze_on = false;
  double rtrunc = 1.0e20, com0[3] = {0, 0, 0};
  DevBuf<double> d_frz;                         // {com0[3], center[3], rtrunc^2} for the kernels (expamd_comp_frz)
  double frz_host[7] = {0, 0, 0, 0, 0, 0, 0};
  bool frz_valid = false;
  // Component::freezeLev (the component key "freezeL", src/Component.cc:255, :1037): levels are assigned on the first call of
  // adjust_multistep_level only (src/multistep.cc:158, :534: `if (not firstCall and c->FreezeLev()) apply = false;`).
  // "noswitch" (:253, level changes at the end of a master step only, from the smallest time step seen during it) is not
  // built: exp_amd_comp_set_level_policy refuses it
  bool freeze_levels = false;
  // "noswitch" / "dtreset" (Component::NoSwitch, DTreset; src/multistep.cc:136-147): d_dtreq is Particle::dtreq, one float per
  // particle id; ns_reset / ns_apply are the two conditions of the sweep about to be launched, set by its caller (host.hip,
  // force_api.hip) and read by expamd_comp_kick_adjust / expamd_comp_propose_levels (kick_adjust.h: NsArgs)
  bool noswitch = false, dtreset = true;
  DevBuf<float> d_dtreq;
  int ns_reset = 0, ns_apply = 1;
  // Component::consp / tidal / rcom (src/Component.cc:214-216, :998-1000, :1024): the escape bookkeeping of fix_positions
  // (:3317-3334) -- a particle beyond rcom of com0 + center is flagged once and left out of the centre-of-mass sums from
  // then on.  d_escaped: iattrib[tidal] of every particle, indexed by its id (the caller's index; ids travel with the
  // slots through every sort, the flags stay where they are).  exp_amd_comp_set_consp
  bool consp_on = false;
  double rcom = 1.0e20;
  DevBuf<uint8_t> d_escaped;
  bool use_rot = false;                    // body-frame rotation (Orient::transformBody), cylinder only
  double rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  PseudoDev pseudo = {0, 0, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // frame acceleration subtracted by the forces
  const void *sorted_for = nullptr;  // force whose cell order the store currently has
  bool acc_live = true;              // acc/pot must survive a reorder
  // host mirror of lev_off (refreshed lazily after a full re-sort: one small read-back), so that
  // launches over ONE level can be sized for that level's population
  DevBuf<double> com_lev, com_red;   // per-level sums of m, m x, m v, m a (fix_positions) / all-reduce scratch
  uint32_t lev_host[66] = {0};
  bool lev_host_valid = false;
  // Block multistep: levels with fewer particles than the force method's sparse_threshold() are kept level-contiguous but
  // NOT cell-sorted (bit L set).  A sparse level has about one particle per basis cell, so the cell
  // order buys nothing there: it is advanced in place, accumulated with per-particle atomics and its
  // forces take the gather path.  Set by the step driver from the level populations; any per-level
  // call of the plain API clears the bit of the level it sorts.
  // Every particle has the same mass (found at upload): BOTH buffer sets then hold the constant in their
  // mass arrays and the scatter passes leave the mass stream alone (16 of their 148 B per particle).
  bool uniform_mass = false;
  double mass_value = 0.0;           // ... that one value
  double mass_abs_sum = 0.0;          // sum |m| over this rank's particles (found at upload): bound of the
                                      // coefficient sums for the deterministic mode's rounding grid
  uint32_t sparse_mask = 0;
  // Levels were committed at the end of a master step but the slots not yet re-partitioned: the step
  // driver lets the next master step's first (full) advance sort do it in the same pass.  lev_host
  // already holds the offsets that sort will establish; any other entry point re-partitions first
  // (expamd_comp_touch).
  bool commit_pending = false;          // (with partition_stale) the proposed levels of the slots >= commit_beg are still
  size_t commit_beg = 0;                // in newlev: the sort that settles the partition reads them there and stores them
  bool partition_stale = false;
  // ... more generally after ANY sweep whose examined levels [stale_lo, multistep] are exactly what the next
  // sub-step advances: that sub-step's advance sort re-partitions them in the same pass (stale_for: the force
  // method whose cell order the slots below stale_lo -- and, up to the level changes, above -- still have)
  int stale_lo = 0;
  const void *stale_for = nullptr;
  // Sort keys + histogram of the NEXT fused step, produced by the force pass of the last one
  // (exp_amd_step_kdk): valid only while nothing else has touched the component since.
  bool prekey_valid = false;
  // The last half-kick of a fused step is NOT applied by its force pass (that would cost a read and
  // a write of v there); it is remembered here and applied, as its own rounding step, inside the
  // next fused step's scatter pass -- or by expamd_comp_touch() before anything else looks.
  double pending_kick = 0.0;
  int pending_lo = 0;                   // ... owed by the levels >= this only (the closing kick of a block-multistep sweep that
                                        // could not move anything: folded into the next sub-step's advance)
  const void *prekey_owner = nullptr;   // force whose cells the keys are
  unsigned long long prekey_epoch = 0;  // ctx->force_epoch when they were written
  double prekey_dtk = 0, prekey_dtd = 0, prekey_center[3] = {0, 0, 0};
  // ... and for a block-multistep run: the sweep that closed a master step (k_kick_adjust with the force's key function)
  // left in `key` the (level, cell) keys of sub-step 0 of the NEXT one -- for this force (owner, epoch), this smallest
  // step, this centre, this many particles; any call that touches the store drops them (expamd_comp_touch)
  bool mprekey_valid = false;
  const void *mprekey_owner = nullptr;
  unsigned long long mprekey_epoch = 0;
  double mprekey_dt_min = 0, mprekey_center[3] = {0, 0, 0};
  size_t mprekey_n = 0;

  // Split fused step (exp_amd_step_kdk on large single-level components): the slots [0, half) and
  // [half, n) are two independently cell-sorted halves; half_off = {0, half, n} on the device plays
  // lev_off for the kernels, keys of the second half carry +ncell.  Any full sort ends the mode.
  bool split = false, prekey_split = false;
  size_t half = 0;
  DevBuf<uint32_t> half_off;

  // The "append" fused step (sph_kernels.h: AppDev; sph.hip: SphForce::fused_step_append): the live buffer set is in the cell
  // order of the NEXT step, every cell a region of app_base[cur] with empty slots (x = +inf) behind its particles -- `n`
  // particles in up to app_ns slots.  X / Y / Z hold the positions the next accumulation reads (already drifted); the positions
  // of the step that was completed are still in the OTHER set, at slot app_src[cur][slot]: what every other call sees, once expamd_comp_densify has made the store an
  // ordinary one again (every entry point does that first: expamd_comp_apply_pending / _velocity_view / download).
  bool appended = false;
  const void *app_owner = nullptr;      // the force method whose cells the regions are
  double app_dt = 0.0;                  // ... for steps of this length
  double app_center[3] = {0, 0, 0};
  size_t app_cap = 0;                   // slots every array of both sets has room for (0: never reserved)
  size_t app_ns = 0;                    // host bound of the slots in use
  uint32_t app_ncell = 0, app_tail = 0; // cells of the layout; slots of the tail region
  uint32_t app_tail_used = 0;           // particles the last placing pass sent to the tail (their regions were full)
  DevBuf<double> xo[2][3];              // (scratch of expamd_comp_densify: the dense positions, then swapped in)
  DevBuf<uint32_t> app_src[2];          // per slot: the slot of the other set the particle came from (its state position)
  DevBuf<uint32_t> app_base[2];         // [ncell + 2] per buffer set
  DevBuf<uint32_t> app_range[2];        // {0, end of the tail}: plays lev_off for the passes over an appended set
  DevBuf<uint32_t> app_cursor;          // [ncell + 1] + flag word
  // A placing pass says at its end whether every particle found room (a flag word behind the cursors): copied to page-locked
  // words behind an event the step waits for (a pageable copy and a stream-wide wait cost an eight-GPU share of 1.25e7
  // particles 3 % of its step).  A pass that ran out of room is put right by its owner (sph.hip: sph_app_recover -- the source
  // set, which no pass writes, made an ordinary store, the force pass redone on it).
  uint32_t *app_hflag = nullptr;        // page-locked: {arrivals in the tail, particles without room} of the last placing pass
  hipEvent_t app_ev = nullptr;
  // The placing pass does not store the acceleration and the potential (32 of 88 bytes a particle that no pass of the next
  // step reads): while app_acc_stale, AX / AY / AZ / POT of the live set are not the state's.  expamd_comp_densify has the
  // owner re-evaluate them at the positions of the completed step, from the coefficient set of that step (sph.hip:
  // sph_app_reeval); the component is on its context's `appended` list meanwhile, so that a dying force can do that first.
  bool app_acc_stale = false;
  int (*app_reeval)(void *owner, struct exp_amd_comp *c) = nullptr;
  // Hysteresis: turning an appended store back into an ordinary one costs a pass over everything (and the next step a full key
  // pass), so a caller that looks at the particles every few steps must not bounce in and out of the mode -- after each such
  // exit the mode stays off for app_wait more fused steps, twice as many as the time before (8, 16, ... 1024), back to 8 once
  // it has run for 64 steps in a row
  int app_wait = 0, app_backoff = 8, app_run = 0;
  bool app_refused = false;             // no room on the device for the appended layout: the ordinary step from now on
  double *a(int k) { return arr[cur][k].p; }
  double *b(int k) { return arr[1 - cur][k].p; }
};

#ifndef APP_EMPTY
#define APP_EMPTY __builtin_inf()        // x of a slot that holds no particle
#endif
// slots an appended store of n particles in ncell cells may use (regions: population + 1/64 of it + 192, rounded up to whole
// waves; tail: max(65536, n / 128)), and the pieces of that rule the layout kernel applies
inline uint32_t expamd_app_tail(size_t n) { const size_t t = n / 128; return (uint32_t)(t < 65536 ? 65536 : t); }
inline size_t expamd_app_slots(size_t n, uint32_t ncell) { return n + n / 64 + (size_t)ncell * 256 + expamd_app_tail(n) + 64; }
// make room: both buffer sets, ids and the state positions at `cap` slots (contents of the live set are kept)
int expamd_comp_app_reserve(exp_amd_comp *c, size_t cap);
// an appended store becomes an ordinary dense one (positions: those of the completed step; order: none)
int expamd_comp_densify(exp_amd_comp *c, bool state_positions = true);
void expamd_app_unlist(exp_amd_comp *c);          // off its context's `appended` list
// the layout of the NEXT buffer set from the populations in `counts` (ncell values), cursors cleared
int expamd_comp_app_layout(exp_amd_comp *c, const uint32_t *counts, uint32_t ncell, int set, uint32_t *also_into);
// after an append pass into buffer set `set`: empty slots marked, the next layout's populations are the cursors
int expamd_comp_app_finish(exp_amd_comp *c, int set, uint32_t *lost);
// ... after the ORDINARY scatter pass filled a set's regions (its running offsets `offs` are base + population)
void k_app_mark_launch(exp_amd_comp *c, int set, const uint32_t *offs);

// exclusive scan of the key histogram (range_lo >= 0: only the bins of that level / half, starting
// at lev_off[range_lo]; lev_off is then left alone)
int expamd_launch_scan(exp_amd_ctx *ctx, hipStream_t st, uint32_t *hist, uint32_t nkeys, uint32_t *lev_off,
                       uint32_t ncell, int nlev, int range_lo);

// ... of all nkeys bins: hist[k] <- exclusive prefix, hist[nkeys] <- total, lev_off[j] <- start of
// level j (bins j*ncell ...), lev_off[nlev] <- total
int expamd_launch_scan_full(exp_amd_ctx *ctx, hipStream_t st, uint32_t *hist, uint32_t nkeys, uint32_t *lev_off,
                            uint32_t ncell, int nlev);

// a component is about to be destroyed: forces that use it as their expansion frame keep a copy
void expamd_forget_component(exp_amd_ctx *ctx, const exp_amd_comp *c);

// recompute sparse_mask for the levels >= first from the (valid) host mirror of the level offsets
void expamd_comp_update_sparse(exp_amd_comp *c, int first, long long thresh);
// kick DT(M)/2 + drift DT(M) in place for the levels [lo, hi] (no reorder): sparse levels
int expamd_comp_advance_levels(exp_amd_comp *c, int lo, int hi, double dt_min, int multistep);
int expamd_comp_settle_pending(exp_amd_comp *c, int lo, int hi, bool advancing);
int expamd_comp_flush_commit(exp_amd_comp *c);       // a commit left to the next sort (commit_pending) is done now
int expamd_comp_take_pending(exp_amd_comp *c, int lo, int hi, double *k0, int *k0lo);

// slots of the particles of levels [first, last] whose proposed level (newlev) differs from their level, in slot order
// within blocks of 256 slots: c->mover_list, c->mover_cnt = {0, count}; capacity = the expected count
int expamd_comp_mover_list(exp_amd_comp *c, int first, int last, size_t expected);
// lanes per mover of the per-mover atomics kernels: enough waves (~2048) to hide the atomics' latency
inline unsigned expamd_mover_spread(size_t movers)
{
  unsigned s = 1;
  while (s < 16 && movers * s < 131072) s <<= 1;
  return s;
}

// number of particles in levels [lo, hi] (refreshes the host mirror of lev_off when stale)
int expamd_comp_level_count(exp_amd_comp *c, int lo, int hi, size_t *count);

// An outside call is about to read or change the component: apply the deferred half-kick of the
// last fused step (if any) and drop the keys it recorded for the next one.
int expamd_comp_touch(exp_amd_comp *c);
// ... without resolving a stale level partition (the step driver's own entry)
int expamd_comp_touch_keep_partition(exp_amd_comp *c);
// ... only the first half (a read-only call: the recorded keys stay valid, they were computed from
// the kicked velocity)
int expamd_comp_apply_pending(exp_amd_comp *c);
// ... a read-only look at the step-boundary velocities that does not change the stored state: *back = 0 (velocities
// are current, a closing half-kick owed has been applied) or the (negative) dt with which v + a * dt is the value
int expamd_comp_velocity_view(exp_amd_comp *c, double *back);

// device copy of the freeze parameters of `c` ({com0, center, rtrunc^2}; nullptr when rtrunc is not set), kept current by
// exp_amd_comp_set_rtrunc / exp_amd_comp_set_center
const double *expamd_comp_frz(const exp_amd_comp *c);
