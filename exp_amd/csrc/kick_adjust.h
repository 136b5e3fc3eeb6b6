// The second half of a block-multistep sub-step (k_kick_adjust) -- a header because the sweep that closes a master step
// also writes the sort keys of the next one, with the key function of the component's force method (sph.hip, cyl.hip).
#pragma once
#include "sort_kernels.h"
#define KA_TPB 256                 // threads of a k_kick_adjust block

// v.a, v.v and a.a of the time-step criteria as the reference's compiler forms them (src/multistep.cc:100-108): every
// product rounded on its own, added in the order of k.  v.a of a near-circular orbit is what is left of terms a
// thousand to a million times larger; a fused multiply-add leaves a different residue, and the criterion built on it
// (dta) decides a level where it is the smallest.
__device__ __forceinline__ void level_sums_lit(double v0, double v1, double v2, double a0, double a1, double a2,
                                               double &dtr, double &vtot, double &atot)
{
#pragma clang fp contract(off)
  dtr = 0.0; vtot = 0.0; atot = 0.0;
  dtr += v0 * a0; vtot += v0 * v0; atot += a0 * a0;
  dtr += v1 * a1; vtot += v1 * v1; atot += a1 * a1;
  dtr += v2 * a2; vtot += v2 * v2; atot += a2 * a2;
}

struct AdjustArgs {
  double dtime, dynD, dynV, dynS, dynA, dynP;
  int multistep, shiftlevl, mfirst_mdrft;
};

// "noswitch" (Component::NoSwitch, src/multistep.cc:136-147): Particle::dtreq -- a float kept between sweeps, here one per
// particle id -- holds the smallest time step asked for since its last reset, and a sweep only assigns levels when `apply`
// (the end of a master step, or the first call).  dtreq == nullptr: the key is off, dtreq is this sweep's dt.
struct NsArgs {
  float *dtreq = nullptr;
  const uint32_t *id = nullptr;
  int reset = 0;               // (DTreset and mstep == 0) or firstCall: dtreq starts again from the largest float (:137-138)
  int apply = 1;               // mdrft == Mstep or firstCall (:147)
};
// -> the float dtreq the level rule sees, and whether it is applied
__device__ __forceinline__ float ns_dtreq(const NsArgs &N, size_t i, double dt, bool &apply)
{
  apply = true;
  if (!N.dtreq) return (float)dt;
  const uint32_t pid = N.id[i];
  float q = N.reset ? __builtin_huge_valf() : N.dtreq[pid];
  if (dt < (double)q) q = (float)dt;
  N.dtreq[pid] = q;
  apply = N.apply != 0;
  return q;
}

// key output of k_kick_adjust (KeyFn::on): positions and the key array
struct KaKeyArgs {
  const double *x = nullptr, *y = nullptr, *z = nullptr;
  uint32_t *key = nullptr;
};
struct KaNoKey {
  static constexpr bool on = false;
  __device__ __forceinline__ uint32_t operator()(double, double, double, uint8_t) const { return 0u; }
};

// The second half of a block-multistep sub-step in one pass over the slots of the levels that take
// part: incr_velocity(0.5*dt*mintvl[M], M) for M >= kick_lo (src/step.cc:198-203) and, on the kicked
// velocities, adjust_multistep_level's sweep over the levels >= first (src/multistep.cc:52-236), as
// k_adjust_levels above.  out[0] += level changes; out[1 + L] += particles proposed for level L
// among those examined (the host derives the new level offsets from them without a second
// read-back).  kick_lo > last: no kick (begin_run's first assignment).
#ifndef KA_ITEMS
#define KA_ITEMS 8       // 256-slot tiles per block at most (the counters leave a block as one atomic per value); the
                         // short ranges of the upper levels take one tile per block: a block's tiles run one after the other
#endif
template <class KeyFn>
__global__ void __launch_bounds__(KA_TPB)
k_kick_adjust(AdjustArgs A, double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
              const double *__restrict__ ax, const double *__restrict__ ay,
              const double *__restrict__ az, const double *__restrict__ pot,
              const uint8_t *__restrict__ lev, uint8_t *__restrict__ newlev,
              const uint32_t *__restrict__ lev_off, int kick_lo, int first, int last, double dt_min,
              unsigned long long *__restrict__ out, unsigned long long *__restrict__ out_next, int items,
              unsigned int *__restrict__ ticket = nullptr, unsigned long long *__restrict__ host_out = nullptr,
              unsigned long long seq = 0ull,
              uint32_t *__restrict__ list = nullptr /* the movers' slots are compacted here as well (k_mover_list's job) */,
              uint32_t *__restrict__ lcnt = nullptr, uint32_t *__restrict__ lcnt_next = nullptr,
              KeyFn kf = KeyFn(), KaKeyArgs K = KaKeyArgs(), NsArgs N = NsArgs())
{
  // two counter sets are used alternately: this launch leaves the other one clean for the next
  if (blockIdx.x == 0 && threadIdx.x < 32) out_next[threadIdx.x] = 0ull;
  if (list && blockIdx.x == 0 && threadIdx.x == 0) { lcnt_next[0] = 0u; lcnt_next[1] = 0u; }
  __shared__ unsigned int cnt[32];
  __shared__ uint32_t s_ml[KA_ITEMS * KA_TPB];      // the block's movers: one claim of the list per block
  __shared__ unsigned int s_mln, s_mlbase;
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_mln = 0;
  __syncthreads();
  const int lo = kick_lo < first ? kick_lo : first;
  const size_t beg = lev_off[lo], end = lev_off[last + 1], ebeg = lev_off[first];
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < items; it++) {
    const size_t i = beg + ((size_t)blockIdx.x * items + it) * KA_TPB + threadIdx.x;
    if (i - threadIdx.x >= end) break;          // (block-uniform)
    const bool valid = i < end;
    unsigned plev = 0, nlev = 0;
    bool examined = false;
    if (valid) {
      plev = lev[i];
      nlev = plev;
      double v0 = vx[i], v1 = vy[i], v2 = vz[i];
      const double a0 = ax[i], a1 = ay[i], a2 = az[i];
      if ((int)plev >= kick_lo) {
        const double dtk = 0.5 * level_dt(dt_min, A.multistep, (int)plev);
        v0 = mul_then_add(v0, a0, dtk);
        v1 = mul_then_add(v1, a1, dtk);
        v2 = mul_then_add(v2, a2, dtk);
        vx[i] = v0; vy[i] = v1; vz[i] = v2;
      }
      examined = i >= ebeg;
      if (examined) {
        const double eps = 1.0e-10;
        double dtr, vtot, atot;
        level_sums_lit(v0, v1, v2, a0, a1, a2, dtr, vtot, atot);
        const double ptot = fabs(pot[i]);
        const double dts = 1.0 / eps;                  // Particle::scale <= 0: criterion off
        const double dtd = A.dynD * 1.0 / sqrt(vtot + eps);
        const double dtv = A.dynV * sqrt(vtot / (atot + eps));
        const double dta = A.dynA * ptot / (fabs(dtr) + eps);
        const double dtA = A.dynP * sqrt(ptot / (atot + eps));
        double dmin = dtd;
        if (dtv < dmin) dmin = dtv;
        if (dts < dmin) dmin = dts;
        if (dta > 0.0 && dta < dmin) dmin = dta;
        if (dtA > 0.0 && dtA < dmin) dmin = dtA;
        const double dt = dmin > eps ? dmin : eps;
        bool apply;
        const float dtreq = ns_dtreq(N, i, dt, apply);
        if (apply) {
          if ((double)dtreq > A.dtime) nlev = 0;
          else nlev = (unsigned)(int)floor(log(A.dtime / (double)dtreq) / log(2.0));
          if (A.shiftlevl) {
            if (nlev > plev) { if (nlev - plev > (unsigned)A.shiftlevl) nlev = plev + A.shiftlevl; }
            else if (plev > nlev) { if (plev - nlev > (unsigned)A.shiftlevl) nlev = plev - A.shiftlevl; }
          }
          if (nlev > (unsigned)A.multistep) nlev = A.multistep;
          if ((int)nlev < A.mfirst_mdrft) nlev = A.mfirst_mdrft;
        }
        newlev[i] = (uint8_t)nlev;
      }
      if constexpr (KeyFn::on) {
        // where this slot will be after the NEXT sub-step's advance -- every level is active in sub-step 0 of a master
        // step: kick DT(level)/2, drift DT(level) with the level just proposed, the arithmetic of advance_one
        // (sort_kernels.h) on the velocities just stored -- and the sort key of that place: that sub-step's sort then
        // counts these 4-byte keys (k_hist_keys_ms) instead of reading x, v, a again (k_key_hist: 73 B a slot)
        const double dtd = level_dt(dt_min, A.multistep, (int)nlev), dtk = 0.5 * dtd;
        const double wx = mul_then_add(v0, a0, dtk), wy = mul_then_add(v1, a1, dtk), wz = mul_then_add(v2, a2, dtk);
        K.key[i] = kf(mul_then_add(K.x[i], wx, dtd), mul_then_add(K.y[i], wy, dtd), mul_then_add(K.z[i], wz, dtd), (uint8_t)nlev);
      }
    }
    // wave-aggregated counters in LDS: one add per wave and value
    const unsigned long long sw = __ballot(examined && nlev != plev);
    if (lane == 0 && sw) atomicAdd(&cnt[0], (unsigned)__popcll(sw));
    if (list && sw) {                           // (wave-uniform)
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&s_mln, (unsigned)__popcll(sw));
      base = (unsigned int)__shfl((int)base, 0);
      if ((sw >> lane) & 1ull) s_ml[base + __popcll(sw & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
    for (int L = A.mfirst_mdrft; L <= A.multistep; L++) {
      const unsigned long long mm = __ballot(examined && (int)nlev == L);
      if (lane == 0 && mm) atomicAdd(&cnt[1 + L], (unsigned)__popcll(mm));
    }
  }
  __syncthreads();
  if (list && s_mln) {                          // (block-uniform)
    if (threadIdx.x == 0) s_mlbase = atomicAdd(lcnt + 1, s_mln);
    __syncthreads();
    for (unsigned int k = threadIdx.x; k < s_mln; k += KA_TPB) list[s_mlbase + k] = s_ml[k];
  }
  // host_out: the LAST block to finish hands the 32 counters to the host itself -- page-locked, host-coherent words, each
  // tagged with the sweep's sequence number, which the step driver polls (no copy launch behind the kernel, no stream
  // wait).  The counters are only ever touched by device-scope atomics, performed at the memory side: the adds RETURN
  // (so they are done before the ticket is drawn) and the last block reads them with atomics too -- no fence, whose
  // agent-scope form writes the whole L2 back on this multi-die part.
  if (host_out) {
    __shared__ unsigned int s_last;
    unsigned long long got = 0ull;
    if (threadIdx.x < 32 && cnt[threadIdx.x]) got = atomicAdd(out + threadIdx.x, (unsigned long long)cnt[threadIdx.x]);
    // (the barrier's wait covers the returns of this wave's adds)
    asm volatile("" ::"v"(got));
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last && threadIdx.x < 32) {
      const unsigned long long v = atomicAdd(out + threadIdx.x, 0ull);
      __hip_atomic_store(host_out + threadIdx.x, (seq << 40) | (v & 0xffffffffffull), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      if (threadIdx.x == 0) atomicExch(ticket, 0u);
    }
    return;
  }
  if (threadIdx.x < 32 && cnt[threadIdx.x]) atomicAdd(out + threadIdx.x, (unsigned long long)cnt[threadIdx.x]);
}

// What a launch of k_kick_adjust takes: expamd_comp_kick_adjust (particles.hip) fills it and either launches the plain
// kernel itself or hands it to the force method's launcher, which instantiates the kernel with its key function (the sweep
// that closes a master step; sph.hip, cyl.hip: prekey_launcher).
struct KaLaunch {
  unsigned grid;
  hipStream_t stream;
  AdjustArgs A;
  double *vx, *vy, *vz;
  const double *ax, *ay, *az, *pot;
  const uint8_t *lev;
  uint8_t *newlev;
  const uint32_t *lev_off;
  int kick_lo, first, last;
  double dt_min;
  unsigned long long *out, *out_next;
  int items;
  unsigned int *ticket;
  unsigned long long *host_out, seq;
  uint32_t *list, *lcnt, *lcnt_next;
  KaKeyArgs K;
  NsArgs N;
};
template <class KeyFn>
static inline void ka_launch_with(const KaLaunch &L, const KeyFn &kf)
{
  k_kick_adjust<KeyFn><<<L.grid, KA_TPB, 0, L.stream>>>(L.A, L.vx, L.vy, L.vz, L.ax, L.ay, L.az, L.pot, L.lev, L.newlev, L.lev_off,
                                                    L.kick_lo, L.first, L.last, L.dt_min, L.out, L.out_next, L.items, L.ticket,
                                                    L.host_out, L.seq, L.list, L.lcnt, L.lcnt_next, kf, L.K, L.N);
}
