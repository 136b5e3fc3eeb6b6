// Cell sort of the particle store, fused with the leapfrog advance.
//
// Two passes per step (a counting sort needs the full histogram before it can place anything):
//   pass 1  k_key_hist    : [kick dt/2, drift dt in registers] -> key = (level, basis cell);
//                           block-private LDS histogram over a key window, then ONE global
//                           atomic per non-empty bin per block (the store is nearly sorted, so a
//                           2048-particle block touches a few dozen bins at most)
//   pass 2  k_scatter_adv : recompute the same advance (bit-identical), reserve a slot range per
//                           (block, bin) with one global atomic, rank inside the block with LDS
//                           atomics, write x', v', m, id, level to the sorted slot
// Nothing is written in place by pass 1, so the advance costs no extra HBM pass.
#pragma once
#include "particles.h"

#define SORT_TPB 256
// items per thread: the histogram passes stream 4-byte keys (or x, v, a once) and like long tiles
// (fewer block-level flushes), the scatter pass has 21 streams per item in flight and likes
// short ones (A/B on MI355X at 1e8: hist 16 -> 0.14 ms vs 0.30 at 4; scatter 4 -> 2.91 ms vs 3.13 at 8)
#ifndef HIST_ITEMS
#define HIST_ITEMS 16
#endif
#ifndef SCAT_ITEMS
#define SCAT_ITEMS 2           // (4 until the end of round 5: config 4's full sorts 5.925 -> 5.868 ms per master step at 2, 5.93 at 8 --
                               // three interleaved triples on one box; the dense one-level sorts take SCAT_ITEMS_DENSE below)
#endif
#define HIST_TILE (SORT_TPB * HIST_ITEMS)
#define SCAT_TILE (SORT_TPB * SCAT_ITEMS)
#ifndef SORT_WIN
#define SORT_WIN 4096          // LDS histogram window (bins) above the block's minimum key
#endif
// ... and the part of it a pass zeroes, counts into and walks for a DENSE ONE-LEVEL store of a force method with one-dimensional
// cells (the sphere's fused step: a 1024-slot tile of 1e8 particles over 2000 cells touches two or three bins, and walking 4096 of
// them twice is most of the tile's LDS work): k_hist_keys 0.19 -> 0.17 ms, k_scatter_adv 2.10 -> 2.05 ms at 1e8
// (profiles/r05_sort_win_ab.txt).  Keys beyond the window take the global atomics either way.  Level ranges of a multistep store
// keep the full window (their keys jump by ncell at a level boundary: config 4 +0.9 ms at 512 bins, +2 ms at 256).
#define SORT_WIN_DENSE 256
// ... and the scatter pass of such a sort takes two slots a thread instead of four: with 256 bins to zero and walk a block's
// fixed cost is small, and the shorter tile keeps more independent blocks in flight (1e8: 2.42 -> 2.29 ms on a box whose plain
// copy reaches 5.06 TB/s, one slot a thread the same, eight 2.55; profiles/r05_sort_win_ab.txt)
#define SCAT_ITEMS_DENSE 2
// the cylinder's cells are two-dimensional (key = iy * numx + ix): neighbouring rows are numx bins apart, so its dense window spans
// four rows of a 256-column grid
#define SORT_WIN_DENSE2D 1024
#define SORT_DENSE_MIN 64       // particles per cell from which a one-level store counts as dense

// A/B switch of the scatter pass (tools/build_variant_tu.sh <suffix> particles "-DSCAT_NT=n"): bit 0 = non-temporal stores of
// the scattered streams, bit 1 = non-temporal loads of the streams read once.  profiles/r05_scatter_ab.txt
#ifndef SCAT_NT
#define SCAT_NT 0
#endif
template <class T> __device__ __forceinline__ void scat_store(T *p, T v)
{
  if constexpr ((SCAT_NT & 1) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}
template <class T> __device__ __forceinline__ T scat_load(const T *p)
{
  if constexpr ((SCAT_NT & 2) != 0) return __builtin_nontemporal_load(p); else return *p;
}

struct AdvanceArgs {
  const double *x, *y, *z, *vx, *vy, *vz, *ax, *ay, *az;
  const uint8_t *lev;
  double dt_kick, dt_drift;    // 0,0 -> positions are used as they are
  double dt_kick0;             // deferred half-kick of the previous fused step, applied first (0: none)
  int kick0_lo;                // ... to the particles of levels >= this (exp_amd_comp::pending_lo; 0: all, levels not read)
  int advance;                 // 0: none; 1: dt_kick / dt_drift; 2: per-level steps of a block-multistep
                               // sub-step: drift DT(M) = dt_min * 2^(multistep - M), kick DT(M)/2
                               // (src/step.cc:115-160: dt*mintvl[M]; exact power-of-two scalings)
  int multistep;
  int lev_lo;                  // advance == 2: particles of levels below are not advanced (inactive in this sub-step)
  double dt_min;
  int nokick;                  // 1: the stored velocities already hold this step's opening half-kick (the last
                               // fused force pass stored v + a dt_close + a dt_open, exp_amd_comp::pending_kick
                               // == -dt_kick): drift only, the acceleration stream is not read
};

// the block-multistep time step of level `lev` (src/multistep.cc:640-646: mintvl[M] = Mstep >> M)
__device__ __forceinline__ double level_dt(double dt_min, int multistep, int lev)
{
  return dt_min * (double)(1u << (multistep - lev));
}

__device__ __forceinline__ void advance_one(const AdvanceArgs &A, size_t i, double &x, double &y,
                                            double &z, double &vx, double &vy, double &vz)
{
  x = scat_load(A.x + i); y = scat_load(A.y + i); z = scat_load(A.z + i);
  if (A.advance == 2 && (int)A.lev[i] < A.lev_lo) {       // inactive level: carried through as it is
    vx = A.vx[i]; vy = A.vy[i]; vz = A.vz[i];
    return;
  }
  if (A.advance) {
    // src/incvel.cc:15-88 then src/incpos.cc:15-69, same roundings as k_kick / k_drift
    vx = scat_load(A.vx + i); vy = scat_load(A.vy + i); vz = scat_load(A.vz + i);
    double dtk = A.dt_kick, dtd = A.dt_drift;
    if (A.advance == 2) {
      dtd = level_dt(A.dt_min, A.multistep, A.lev[i]);
      dtk = 0.5 * dtd;
    }
    if (!A.nokick) {
      const double ax = A.ax[i], ay = A.ay[i], az = A.az[i];
      if (A.dt_kick0 != 0.0 && (A.kick0_lo == 0 || (int)A.lev[i] >= A.kick0_lo)) {   // its own rounding step, as the separate kick
        vx = mul_then_add(vx, ax, A.dt_kick0);
        vy = mul_then_add(vy, ay, A.dt_kick0);
        vz = mul_then_add(vz, az, A.dt_kick0);
      }
      vx = mul_then_add(vx, ax, dtk);
      vy = mul_then_add(vy, ay, dtk);
      vz = mul_then_add(vz, az, dtk);
    }
    x = mul_then_add(x, vx, dtd);
    y = mul_then_add(y, vy, dtd);
    z = mul_then_add(z, vz, dtd);
  }
}

__device__ __forceinline__ uint32_t block_min_u32(uint32_t v, uint32_t *slot)
{
  for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off));
  if (threadIdx.x == 0) *slot = 0xffffffffu;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMin(slot, v);
  __syncthreads();
  return *slot;
}

// Wave-aggregated LDS histogram updates.  The store is nearly sorted, so the 64 keys of a wave
// are mostly equal: 64 lanes adding to ONE LDS word serialise (a 64-way same-address conflict per
// instruction).  Instead the lanes are grouped by key with ballots and one lane per distinct key
// adds the group's population.
__device__ __forceinline__ void wave_hist_add(uint32_t key, bool valid, uint32_t kmin,
                                              uint32_t *lh, uint32_t *__restrict__ hist, uint32_t win = SORT_WIN)
{
  const int lane = threadIdx.x & 63;
  unsigned long long rem = __ballot(valid);
  while (rem) {
    const int lead = __ffsll((long long)rem) - 1;
    const uint32_t kk = (uint32_t)__shfl((int)key, lead);
    const unsigned long long mm = __ballot(valid && key == kk);
    // A thinly populated multistep level has a different key in almost every lane: a group that is a small part of
    // what is left means ~64 rounds of this loop -- the lanes left then add for themselves, few of them to one word.
    if (__popcll(mm) * 8 < __popcll(rem)) {
      if ((rem >> lane) & 1ull) {
        const uint32_t d = key - kmin;
        if (d < win) atomicAdd(&lh[d], 1u);
        else atomicAdd(&hist[key], 1u);
      }
      return;
    }
    if (lane == lead) {
      const uint32_t d = kk - kmin, cnt = (uint32_t)__popcll(mm);
      if (d < win) atomicAdd(&lh[d], cnt);
      else atomicAdd(&hist[kk], cnt);
    }
    rem &= ~mm;
  }
}

// Same grouping for the scatter pass: returns the lane's rank inside its (block, bin) -- the
// group's base from one LDS atomic plus the number of lower lanes with the same key -- or
// 0xffffffff when the key lies outside the LDS window (ranked by a global atomic later).
__device__ __forceinline__ uint32_t wave_rank(uint32_t key, bool valid, uint32_t kmin, uint32_t *lh, uint32_t win = SORT_WIN)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t rk = 0xffffffffu;
  unsigned long long rem = __ballot(valid);
  while (rem) {
    const int lead = __ffsll((long long)rem) - 1;
    const uint32_t kk = (uint32_t)__shfl((int)key, lead);
    const unsigned long long mm = __ballot(valid && key == kk);
    if (__popcll(mm) * 8 < __popcll(rem)) {           // (see wave_hist_add: every lane left takes its own rank)
      if ((rem >> lane) & 1ull) {
        const uint32_t dl = key - kmin;
        if (dl < win) rk = atomicAdd(&lh[dl], 1u);
      }
      return rk;
    }
    const uint32_t d = kk - kmin;
    uint32_t base = 0;
    if (d < win) {
      if (lane == lead) base = atomicAdd(&lh[d], (uint32_t)__popcll(mm));
      base = (uint32_t)__shfl((int)base, lead);
      if (valid && key == kk) rk = base + (uint32_t)__popcll(mm & lt);
    }
    rem &= ~mm;
  }
  return rk;
}

// slot range handled by a sort pass: the whole store, or the levels [lo, hi] (device-resident
// level offsets, so that no host synchronisation is needed to sort one level)
struct SortRange {
  const uint32_t *lev_off;   // nullptr: [0, n)
  int lo, hi;
  size_t n;
};

__device__ __forceinline__ void sort_range(const SortRange &R, size_t &beg, size_t &end)
{
  if (R.lev_off) { beg = R.lev_off[R.lo]; end = R.lev_off[R.hi + 1]; }
  else { beg = 0; end = R.n; }
}

// ITEMS: slots per thread.  HIST_ITEMS for the bulk passes; the short ranges of a block-multistep sub-step (a few thousand to
// a few ten thousand slots, sorted sixteen times per master step) take HIST_ITEMS_SHORT: sixteen keys per thread one after
// the other -- each a square root and a division or more -- made a 13 000-slot pass of four blocks 35 us long
#define HIST_ITEMS_SHORT 2
#ifndef HIST_SHORT_MAX
#define HIST_SHORT_MAX 1048576     // slots up to which a range counts as short (65536 until round 5: a level range of 8e4 - 3e5 slots
                                   // then ran sixteen slots a thread on 20 - 80 blocks, 40 us where 10 do: config 4 -0.09 ms)
#endif
template <class KeyFn, int ITEMS = HIST_ITEMS>
__global__ void __launch_bounds__(SORT_TPB)
k_key_hist(KeyFn kf, AdvanceArgs A, SortRange R, uint32_t *__restrict__ key_out,
           uint32_t *__restrict__ hist)
{
  __shared__ uint32_t lh[SORT_WIN];
  __shared__ uint32_t kmin_s;
  size_t rbeg, n;
  sort_range(R, rbeg, n);
  const size_t base = rbeg + (size_t)blockIdx.x * (SORT_TPB * ITEMS);
  if (base >= n) return;
  uint32_t k[ITEMS];
  uint32_t mn = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const size_t i = base + (size_t)j * SORT_TPB + threadIdx.x;
    k[j] = 0xffffffffu;
    if (i < n) {
      double x, y, z, vx, vy, vz;
      advance_one(A, i, x, y, z, vx, vy, vz);
      k[j] = kf(x, y, z, A.lev[i]);
      key_out[i] = k[j];
      mn = min(mn, k[j]);
    }
  }
  for (int b = threadIdx.x; b < SORT_WIN; b += SORT_TPB) lh[b] = 0;
  const uint32_t kmin = block_min_u32(mn, &kmin_s);     // (also orders the zeroing)
#pragma unroll
  for (int j = 0; j < ITEMS; j++) wave_hist_add(k[j], k[j] != 0xffffffffu, kmin, lh, hist);
  __syncthreads();
  for (int b = threadIdx.x; b < SORT_WIN; b += SORT_TPB) {
    const uint32_t c = lh[b];
    if (c) atomicAdd(&hist[kmin + b], c);
  }
}

// pass 1 when the keys already exist (written by the previous step's force pass): histogram only
// sparse_mask != 0 (keys a block-multistep sweep left, kick_adjust.h: full (level, cell) keys): the keys of the levels that
// are not cell-sorted are collapsed to the level's first bin -- level = key / stride -- and stored back for the scatter pass
[[maybe_unused]] static __global__ void __launch_bounds__(SORT_TPB)
k_hist_keys(uint32_t *__restrict__ key, size_t n, uint32_t *__restrict__ hist, uint32_t sparse_mask = 0u, uint32_t stride = 1u,
            uint32_t win = SORT_WIN /* bins of the LDS window in use (<= SORT_WIN): SORT_WIN_DENSE for a dense one-level store */)
{
  __shared__ uint32_t lh[SORT_WIN];
  __shared__ uint32_t kmin_s;
  const size_t base = (size_t)blockIdx.x * HIST_TILE;
  if (base >= n) return;
  uint32_t k[HIST_ITEMS];
  uint32_t mn = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < HIST_ITEMS; j++) {
    const size_t i = base + (size_t)j * SORT_TPB + threadIdx.x;
    k[j] = (i < n) ? key[i] : 0xffffffffu;
    if (sparse_mask && i < n) {
      const uint32_t lv = k[j] / stride;
      if ((sparse_mask >> lv) & 1u) { k[j] = lv * stride; key[i] = k[j]; }
    }
    mn = min(mn, k[j]);
  }
  for (int b = threadIdx.x; b < (int)win; b += SORT_TPB) lh[b] = 0;
  const uint32_t kmin = block_min_u32(mn, &kmin_s);
#pragma unroll
  for (int j = 0; j < HIST_ITEMS; j++) wave_hist_add(k[j], k[j] != 0xffffffffu, kmin, lh, hist, win);
  __syncthreads();
  for (int b = threadIdx.x; b < (int)win; b += SORT_TPB) {
    const uint32_t c = lh[b];
    if (c) atomicAdd(&hist[kmin + b], c);
  }
}

struct ScatterDst {
  double *x, *y, *z, *vx, *vy, *vz, *m, *ax, *ay, *az, *pot;
  uint32_t *id;
  uint8_t *lev;
};

struct ScatterSrc {
  const double *m, *ax, *ay, *az, *pot;
  const uint32_t *id;
};

// (ITEMS: slots per thread; SCAT_ITEMS_SHORT for the level ranges of a block-multistep sub-step, as HIST_ITEMS_SHORT above)
#define SCAT_ITEMS_SHORT 1
#ifndef SCAT_SHORT_MAX
#define SCAT_SHORT_MAX 1048576
#endif
template <bool MOVE_ACC, int ITEMS = SCAT_ITEMS>
__global__ void __launch_bounds__(SORT_TPB)
k_scatter_adv(AdvanceArgs A, ScatterSrc S, ScatterDst D, SortRange R,
              const uint32_t *__restrict__ key, uint32_t *__restrict__ cursor, uint32_t win = SORT_WIN /* as k_hist_keys */)
{
  constexpr int SCAT_ITEMS_ = ITEMS;
  __shared__ uint32_t lh[SORT_WIN];       // count, then global base of the (block, bin) range
  __shared__ uint32_t kmin_s;
  size_t rbeg, n;
  sort_range(R, rbeg, n);
  const size_t base = rbeg + (size_t)blockIdx.x * (SORT_TPB * SCAT_ITEMS_);
  if (base >= n) return;
  uint32_t k[SCAT_ITEMS_], rk[SCAT_ITEMS_];
  uint32_t mn = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < SCAT_ITEMS_; j++) {
    const size_t i = base + (size_t)j * SORT_TPB + threadIdx.x;
    k[j] = (i < n) ? key[i] : 0xffffffffu;
    mn = min(mn, k[j]);
  }
  for (int b = threadIdx.x; b < (int)win; b += SORT_TPB) lh[b] = 0;
  const uint32_t kmin = block_min_u32(mn, &kmin_s);
  // rank inside (block, bin)
#pragma unroll
  for (int j = 0; j < SCAT_ITEMS_; j++) rk[j] = wave_rank(k[j], k[j] != 0xffffffffu, kmin, lh, win);
  __syncthreads();
  // reserve the global range of every non-empty bin: lh[b] <- base
  for (int b = threadIdx.x; b < (int)win; b += SORT_TPB) {
    const uint32_t c = lh[b];
    if (c) lh[b] = atomicAdd(&cursor[kmin + b], c);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SCAT_ITEMS_; j++) {
    if (k[j] == 0xffffffffu) continue;
    const size_t i = base + (size_t)j * SORT_TPB + threadIdx.x;
    const uint32_t d = k[j] - kmin;
    const uint32_t dest = (d < win) ? lh[d] + rk[j] : atomicAdd(&cursor[k[j]], 1u);
    double x, y, z, vx = 0, vy = 0, vz = 0;
    advance_one(A, i, x, y, z, vx, vy, vz);
    if (!A.advance) { vx = A.vx[i]; vy = A.vy[i]; vz = A.vz[i]; }
    scat_store(D.x + dest, x); scat_store(D.y + dest, y); scat_store(D.z + dest, z);
    scat_store(D.vx + dest, vx); scat_store(D.vy + dest, vy); scat_store(D.vz + dest, vz);
    if (D.m) D.m[dest] = S.m[i];            // nullptr: uniform mass, both buffer sets hold it already
    scat_store(D.id + dest, scat_load(S.id + i));
    if (D.lev) D.lev[dest] = A.lev[i];     // nullptr: every level is 0 and stays 0 (single-level runs)
    if (MOVE_ACC) {
      D.ax[dest] = S.ax[i]; D.ay[dest] = S.ay[i]; D.az[dest] = S.az[i]; D.pot[dest] = S.pot[i];
    }
  }
}

struct KaLaunch;
// host helper (defined in particles.hip): scan + scatter after a k_key_hist launch
int expamd_comp_finish_sort(exp_amd_comp *c, uint32_t nkeys, uint32_t ncell, bool move_acc,
                            const AdvSpec &adv, int level = -1, int level_hi = -1);
// level < 0: all slots; else the slots of levels [level, max(level, level_hi)]
SortRange expamd_sort_range(exp_amd_comp *c, int level, int level_hi = -1);
int expamd_comp_prepare_hist(exp_amd_comp *c, uint32_t nkeys);
int expamd_comp_propose_levels(exp_amd_comp *c, double dtime, const double dynfrac[5], int shiftlevl,
                               int multistep, int mfirst_mdrft, int first);
int expamd_comp_commit_levels(exp_amd_comp *c, size_t beg = 0);
int expamd_comp_kick_adjust(exp_amd_comp *c, double dtime, const double dynfrac[5], int shiftlevl,
                            int multistep, int mfirst_mdrft, int kick_lo, int first, double dt_min,
                            const unsigned long long **result, unsigned long long *host_out = nullptr,
                            unsigned long long seq = 0ull, bool *launched = nullptr, bool build_list = false,
                            void (*key_launch)(void *, const struct KaLaunch &) = nullptr, void *key_self = nullptr);
// (host_out: device address of 33 page-locked, host-coherent words -- the counters and, behind them, `seq` once they
// are all there: k_kick_adjust's last block writes them itself)
AdvanceArgs expamd_advance_args(exp_amd_comp *c, const AdvSpec &adv);
// the keys a closing sweep left for sub-step 0 (exp_amd_comp::mprekey_*) are this force's, for this smallest step, and nothing
// has touched the store since
bool expamd_comp_mprekey_ok(const exp_amd_comp *c, const void *owner, double dt_min);
