// Particle store: upload/download, leapfrog kick/drift, counting sort by (level, cell).
//
// Replaces (does not translate) src/cudaComponent.cu:621-792 (AoS mirror, zeroPotAccKernel,
// thrust level sort), src/cudaIncpos.cu:9-29 (coordDrift) and src/cudaIncvel.cu:9-41
// (velocityKick).  Layout is fp64 SoA so that every pass streams HBM with coalesced
// 8-byte-per-lane loads; a level is a contiguous slot range [lev_off[M], lev_off[M+1]).
#include "sort_kernels.h"
#include "kick_adjust.h"

#define TPB 256

// ---- leapfrog -------------------------------------------------------------------------------

// src/incpos.cc:15-69 : pos[k] += vel[k]*dt
__global__ void __launch_bounds__(TPB)
k_drift(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
        const double *__restrict__ vx, const double *__restrict__ vy,
        const double *__restrict__ vz, const uint32_t *__restrict__ lev_off, int lo, int hi,
        double dt)
{
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x; i < end;
       i += (size_t)gridDim.x * TPB) {
    // separate multiply and add: bit-identical to the reference's scalar `pos += vel*dt`
    x[i] = mul_then_add(x[i], vx[i], dt);
    y[i] = mul_then_add(y[i], vy[i], dt);
    z[i] = mul_then_add(z[i], vz[i], dt);
  }
}

// src/incvel.cc:15-88 : vel[k] += acc[k]*dt
__global__ void __launch_bounds__(TPB)
k_kick(double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
       const double *__restrict__ ax, const double *__restrict__ ay,
       const double *__restrict__ az, const uint32_t *__restrict__ lev_off, int lo, int hi,
       double dt)
{
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x; i < end;
       i += (size_t)gridDim.x * TPB) {
    vx[i] = mul_then_add(vx[i], ax[i], dt);
    vy[i] = mul_then_add(vy[i], ay[i], dt);
    vz[i] = mul_then_add(vz[i], az[i], dt);
  }
}

// first half of a block-multistep sub-step, in place: incr_velocity(DT(M)/2, M); incr_position(DT(M), M)
// for the slots of levels [lo, hi] (src/step.cc:126-148), DT(M) = dt_min 2^(multistep - M)
__global__ void __launch_bounds__(TPB)
k_advance_levels(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                 double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                 const double *__restrict__ ax, const double *__restrict__ ay,
                 const double *__restrict__ az, const uint8_t *__restrict__ lev,
                 const uint32_t *__restrict__ lev_off, int lo, int hi, double dt_min, int multistep,
                 double dt_kick0 /* a closing half-kick still owed by the levels >= kick0_lo: applied first */, int kick0_lo)
{
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x; i < end;
       i += (size_t)gridDim.x * TPB) {
    const int L = lev[i];
    const double dtd = level_dt(dt_min, multistep, L), dtk = 0.5 * dtd;
    double u = vx[i], v = vy[i], w = vz[i];
    const double a0 = ax[i], a1 = ay[i], a2 = az[i];
    if (dt_kick0 != 0.0 && L >= kick0_lo) {
      u = mul_then_add(u, a0, dt_kick0);
      v = mul_then_add(v, a1, dt_kick0);
      w = mul_then_add(w, a2, dt_kick0);
    }
    u = mul_then_add(u, a0, dtk);
    v = mul_then_add(v, a1, dtk);
    w = mul_then_add(w, a2, dtk);
    vx[i] = u; vy[i] = v; vz[i] = w;
    x[i] = mul_then_add(x[i], u, dtd);
    y[i] = mul_then_add(y[i], v, dtd);
    z[i] = mul_then_add(z[i], w, dtd);
  }
}

// src/ComponentContainer.cc:641-665
__global__ void __launch_bounds__(TPB)
k_zero_acc(double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
           double *__restrict__ pot, const uint32_t *__restrict__ lev_off, int lo, int hi)
{
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x; i < end;
       i += (size_t)gridDim.x * TPB) {
    ax[i] = 0.0;
    ay[i] = 0.0;
    az[i] = 0.0;
    pot[i] = 0.0;
  }
}

// ---- counting sort (pass kernels live in sort_kernels.h) -----------------------------------

// exclusive scan of hist[0..nkeys) in place; hist[nkeys] = total; lev_off[L] = start of key L*ncell,
// lev_off[nlev] = total.  range mode (range_lo >= 0): only the bins of the levels range_lo..range_hi
// are populated; positions start at lev_off[range_lo] and the level offsets outside are left alone.
// One block per chunk of SCAN_TPB x SCAN_SI bins; with several chunks k_scan_sums first leaves every
// chunk's total in `sums` and block b starts from the sum of the totals before it (the single-block
// loop this replaces took 18 us per chunk in sequence: 90 us for the cylinder's 5 x 32769 bins).
// SCAN_TPB = 256 (round 4; was 1024 threads with 135 KB of LDS a block): in the first sub-step of a two-component master
// step these small launches sit between the other stream's full-size passes, and a workgroup that needs sixteen free wave
// slots and most of a CU's LDS at once waited 100 us (sphere, one block) and 260 us (cylinder, five) for a CU to drain;
// four waves and 34 KB find room at once.
#define SCAN_TPB 256u
#define SCAN_NW (SCAN_TPB / 64u)
#define SCAN_SI 33u     // bins per thread (odd: a thread's run of words in the LDS stage starts on its own bank)

__device__ __forceinline__ uint32_t block_scan_tpb(uint32_t x, uint32_t *wsum /* [SCAN_NW] */, uint32_t &total)
{
  // inclusive scan of one value per thread over a SCAN_TPB-thread block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = (lane < (int)SCAN_NW) ? wsum[lane] : 0u;
#pragma unroll
    for (int off = 1; off < (int)SCAN_NW; off <<= 1) {
      const uint32_t y = __shfl_up(w, off);
      if (lane >= off) w += y;
    }
    if (lane < (int)SCAN_NW) wsum[lane] = w;           // inclusive prefix of the wave totals
  }
  __syncthreads();
  total = wsum[SCAN_NW - 1];
  return x + (wave ? wsum[wave - 1] : 0u);
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan_sums(const uint32_t *__restrict__ hist, uint32_t k0, uint32_t k1, uint32_t *__restrict__ sums)
{
  __shared__ uint32_t wsum[SCAN_NW];
  const uint32_t base = k0 + blockIdx.x * SCAN_TPB * SCAN_SI, top = min(k1, base + SCAN_TPB * SCAN_SI);
  uint32_t s = 0;
  for (uint32_t k = base + threadIdx.x; k < top; k += SCAN_TPB) s += hist[k];
  uint32_t total;
  (void)block_scan_tpb(s, wsum, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_TPB)
k_scan(uint32_t *__restrict__ hist, uint32_t nkeys, uint32_t *__restrict__ lev_off,
       uint32_t ncell, int nlev, int range_lo, int range_hi, const uint32_t *__restrict__ sums)
{
  __shared__ uint32_t wsum[SCAN_NW];
  const int t = threadIdx.x;
  const uint32_t k0 = (range_lo >= 0) ? (uint32_t)range_lo * ncell : 0u;
  const uint32_t k1 = (range_lo >= 0) ? (uint32_t)(range_hi + 1) * ncell : nkeys;
  uint32_t carry = (range_lo >= 0) ? lev_off[range_lo] : 0u;
  if (blockIdx.x) {                          // the chunks before this one
    uint32_t p = 0, tot;
    for (uint32_t j = t; j < blockIdx.x; j += SCAN_TPB) p += sums[j];
    (void)block_scan_tpb(p, wsum, tot);
    carry += tot;
    __syncthreads();
  }
  // SCAN_SI consecutive bins per thread (the pass is barrier-latency bound: the cylinder's bins took 200 us at one
  // bin per thread), staged through LDS so that the global loads and stores are coalesced: element j * SCAN_TPB + t in
  // pass j; a thread's own run of SCAN_SI words starts at word t * SCAN_SI -- an odd stride, no bank conflicts.
  __shared__ uint32_t stage[SCAN_TPB * SCAN_SI];
  const uint32_t cb = k0 + blockIdx.x * SCAN_TPB * SCAN_SI;           // first bin of this chunk
#pragma unroll
  for (uint32_t j = 0; j < SCAN_SI; j++) {
    const uint32_t e = j * SCAN_TPB + (uint32_t)t;
    stage[e] = (cb + e < k1) ? hist[cb + e] : 0u;
  }
  __syncthreads();
  uint32_t v[SCAN_SI], s = 0;
#pragma unroll
  for (uint32_t j = 0; j < SCAN_SI; j++) { v[j] = stage[(uint32_t)t * SCAN_SI + j]; s += v[j]; }
  uint32_t total;
  const uint32_t incl = block_scan_tpb(s, wsum, total);
  uint32_t excl = carry + (incl - s);
  // (a range of several levels re-partitions its slots: the inner level starts move)  next level boundary at or
  // after this thread's first bin -- one division here instead of a modulo per bin
  const uint32_t kb = cb + (uint32_t)t * SCAN_SI;
  const bool offs = range_lo < 0 || range_hi > range_lo;
  uint32_t lvl = (kb + ncell - 1u) / ncell, kbnd = lvl * ncell;
#pragma unroll
  for (uint32_t j = 0; j < SCAN_SI; j++) {
    const uint32_t k = kb + j;
    stage[(uint32_t)t * SCAN_SI + j] = excl;
    if (offs && k == kbnd && k < k1) { lev_off[lvl] = excl; lvl++; kbnd += ncell; }
    excl += v[j];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < SCAN_SI; j++) {
    const uint32_t e = j * SCAN_TPB + (uint32_t)t;
    if (cb + e < k1) hist[cb + e] = stage[e];
  }
  if (t == 0 && range_lo < 0 && blockIdx.x + 1 == gridDim.x) {
    hist[nkeys] = carry + total;
    lev_off[nlev] = carry + total;
  }
}

// chunk totals of multi-chunk scans: one small buffer per context (grown on demand)
static int scan_launch(exp_amd_ctx *ctx, hipStream_t st, uint32_t *hist, uint32_t nkeys, uint32_t *lev_off,
                       uint32_t ncell, int nlev, int range_lo, int range_hi)
{
  const uint32_t k0 = (range_lo >= 0) ? (uint32_t)range_lo * ncell : 0u;
  const uint32_t k1 = (range_lo >= 0) ? (uint32_t)(range_hi + 1) * ncell : nkeys;
  const uint32_t P = k1 > k0 ? (k1 - k0 + SCAN_TPB * SCAN_SI - 1u) / (SCAN_TPB * SCAN_SI) : 1u;
  auto &SS = ctx->scan_sums[(ctx->aux && st == ctx->aux) ? 1 : 0];
  if (P > 1u) {
    if (SS.n < P) {
      HIP_TRY(ctx, hipStreamSynchronize(st));
      if (SS.alloc((size_t)P + 64) != hipSuccess)
        return expamd_fail(ctx, EXP_AMD_ERR_HIP, "scan: hipMalloc failed");
    }
    k_scan_sums<<<P, SCAN_TPB, 0, st>>>(hist, k0, k1, SS.p);
  }
  k_scan<<<P, SCAN_TPB, 0, st>>>(hist, nkeys, lev_off, ncell, nlev, range_lo, range_hi, P > 1u ? SS.p : nullptr);
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

__global__ void __launch_bounds__(TPB)
k_iota(uint32_t *__restrict__ id, size_t n)
{
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) id[i] = (uint32_t)i;
}

// out[id[i]] = in[i]  (return to the caller's order)
__global__ void __launch_bounds__(TPB)
k_unpermute_f64(const double *__restrict__ in, const uint32_t *__restrict__ id, size_t n,
                double *__restrict__ out)
{
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[id[i]] = in[i];
}

__global__ void __launch_bounds__(TPB)
k_unpermute_lev(const uint8_t *__restrict__ in, const uint32_t *__restrict__ id, size_t n,
                int32_t *__restrict__ out)
{
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[id[i]] = (int32_t)in[i];
}

// out[id[i]] = v[i] + a[i] * dt (two roundings, as k_kick)
__global__ void __launch_bounds__(TPB)
k_unpermute_kicked(const double *__restrict__ v, const double *__restrict__ a, double dt,
                   const uint32_t *__restrict__ id, size_t n, double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[id[i]] = mul_then_add(v[i], a[i], dt);
}

// in[j] is in caller order; slot i holds caller index id[i]
__global__ void __launch_bounds__(TPB)
k_permute_lev(const int32_t *__restrict__ in, const uint32_t *__restrict__ id, size_t n,
              uint8_t *__restrict__ out)
{
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[i] = (uint8_t)in[id[i]];
}

__global__ void __launch_bounds__(TPB)
k_permute_f64(const double *__restrict__ in, const uint32_t *__restrict__ id, size_t n,
              double *__restrict__ out)
{
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[i] = in[id[i]];
}

static unsigned stream_grid(exp_amd_ctx *ctx, size_t n)
{
  size_t want = (n + TPB - 1) / TPB;
  size_t cap = (size_t)ctx->num_cu * 8;
  return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

int expamd_launch_scan(exp_amd_ctx *ctx, hipStream_t st, uint32_t *hist, uint32_t nkeys, uint32_t *lev_off,
                       uint32_t ncell, int nlev, int range_lo)
{
  return scan_launch(ctx, st, hist, nkeys, lev_off, ncell, nlev, range_lo, range_lo);
}

int expamd_launch_scan_full(exp_amd_ctx *ctx, hipStream_t st, uint32_t *hist, uint32_t nkeys, uint32_t *lev_off,
                            uint32_t ncell, int nlev)
{
  return scan_launch(ctx, st, hist, nkeys, lev_off, ncell, nlev, -1, -1);
}

int expamd_comp_prepare_hist(exp_amd_comp *c, uint32_t nkeys)
{
  exp_amd_ctx *ctx = c->ctx;
  c->mprekey_valid = false;            // (every sort uses the key array and moves slots: a caller that has such keys asked first)
  c->sort_win = 0;                     // (the force method's sort function sets it after this call, for its own sort)
  if (c->hist_cap < (size_t)nkeys + 1) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, c->hist.alloc((size_t)nkeys + 1));
    c->hist_cap = (size_t)nkeys + 1;
    c->hist_clean = 0;
  }
  if (c->hist_clean < (size_t)nkeys + 1)
    HIP_TRY(ctx, hipMemsetAsync(c->hist.p, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), ctx->stream));
  c->hist_clean = 0;           // (about to be counted into)
  return EXP_AMD_OK;
}

int expamd_comp_level_count(exp_amd_comp *c, int lo, int hi, size_t *count)
{
  exp_amd_ctx *ctx = c->ctx;
  if (c->nlevels <= 1) { *count = c->n; return EXP_AMD_OK; }
  if (!c->lev_host_valid) {
    HIP_TRY(ctx, hipMemcpyAsync(c->lev_host, c->lev_off.p, (size_t)(c->nlevels + 1) * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    c->lev_host_valid = true;
  }
  *count = (size_t)c->lev_host[hi + 1] - (size_t)c->lev_host[lo];
  return EXP_AMD_OK;
}

void expamd_comp_update_sparse(exp_amd_comp *c, int first, long long thresh)
{
  if (!c->lev_host_valid || c->nlevels <= 1) { c->sparse_mask = 0; return; }
  for (int L = first; L < c->nlevels; L++) {
    const uint32_t cnt = c->lev_host[L + 1] - c->lev_host[L];
    if ((long long)cnt < thresh) c->sparse_mask |= (1u << L);
    else c->sparse_mask &= ~(1u << L);
  }
}

// before a pass that advances the levels [lo, hi] with the half-kick owed folded in: it must hold all of the owing
// levels [pending_lo, top] or none of them -- otherwise the kick is applied now, by a launch of its own
int expamd_comp_settle_pending(exp_amd_comp *c, int lo, int hi, bool advancing)
{
  if (c->pending_kick == 0.0) return EXP_AMD_OK;
  const int top = c->nlevels - 1;
  const bool all = lo <= c->pending_lo && hi >= top, none = hi < c->pending_lo;
  if (advancing && (all || none)) return EXP_AMD_OK;
  if (!advancing && none) return EXP_AMD_OK;
  return expamd_comp_apply_pending(c);
}

// a pass that advances the levels [lo, hi] in place: a closing half-kick still owed (pending_kick, levels >= pending_lo)
// is taken along -- *k0 / *k0lo, settled here -- when the range holds all of the owing levels, left alone when it holds
// none of them, applied by a launch of its own otherwise
int expamd_comp_take_pending(exp_amd_comp *c, int lo, int hi, double *k0, int *k0lo)
{
  *k0 = 0.0;
  *k0lo = 0;
  if (c->pending_kick == 0.0) return EXP_AMD_OK;
  if (lo <= c->pending_lo && hi >= c->nlevels - 1) {
    *k0 = c->pending_kick;
    *k0lo = c->pending_lo;
    c->pending_kick = 0.0;
    c->pending_lo = 0;
    return EXP_AMD_OK;
  }
  if (hi >= c->pending_lo) return expamd_comp_apply_pending(c);
  return EXP_AMD_OK;
}

int expamd_comp_advance_levels(exp_amd_comp *c, int lo, int hi, double dt_min, int multistep)
{
  size_t nr = 0;
  int rc = expamd_comp_level_count(c, lo, hi, &nr);
  if (rc) return rc;
  double k0 = 0.0;
  int k0lo = 0;
  if ((rc = expamd_comp_take_pending(c, lo, hi, &k0, &k0lo))) return rc;
  if (nr == 0) return EXP_AMD_OK;
  ProfScope ps(c->ctx, "k_advance_levels");
  k_advance_levels<<<stream_grid(c->ctx, nr), TPB, 0, c->ctx->stream>>>(
      c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY),
      c->a(A_AZ), c->level[c->cur].p, c->lev_off.p, lo, hi, dt_min, multistep, k0, k0lo);
  HIP_TRY(c->ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// sort key of a force-independent re-partition: the level alone
struct LevelKeyFn {
  __device__ __forceinline__ uint32_t operator()(double, double, double, uint8_t lev) const { return lev; }
};

static int partition_by_level(exp_amd_comp *c)
{
  exp_amd_ctx *ctx = c->ctx;
  c->partition_stale = false;
  if (c->n == 0) return EXP_AMD_OK;
  const uint32_t nkeys = (uint32_t)c->nlevels;
  int rc = expamd_comp_prepare_hist(c, nkeys);
  if (rc) return rc;
  uint32_t keep[66];
  const bool had = c->lev_host_valid;
  for (int k = 0; k <= c->nlevels; k++) keep[k] = c->lev_host[k];
  {
    ProfScope ps(ctx, "k_key_hist");
    AdvanceArgs A = expamd_advance_args(c, AdvSpec());
    k_key_hist<LevelKeyFn><<<cdiv(c->n, HIST_TILE), SORT_TPB, 0, ctx->stream>>>(
        LevelKeyFn{}, A, expamd_sort_range(c, -1, -1), c->key.p, c->hist.p);
  }
  rc = expamd_comp_finish_sort(c, nkeys, 1u, true, AdvSpec(), -1, -1);
  if (rc) return rc;
  if (had) { for (int k = 0; k <= c->nlevels; k++) c->lev_host[k] = keep[k]; c->lev_host_valid = true; }
  c->sorted_for = nullptr;
  c->sparse_mask = 0;
  return EXP_AMD_OK;
}

int expamd_comp_flush_commit(exp_amd_comp *c)
{
  if (!c->commit_pending) return EXP_AMD_OK;
  c->commit_pending = false;
  const void *sf = c->sorted_for;
  int rc = expamd_comp_commit_levels(c, c->commit_beg);
  c->sorted_for = sf;                    // (it was cleared when the commit was put off)
  return rc;
}

bool expamd_comp_mprekey_ok(const exp_amd_comp *c, const void *owner, double dt_min)
{
  return c->mprekey_valid && c->mprekey_owner == owner && c->mprekey_epoch == c->ctx->force_epoch && c->mprekey_dt_min == dt_min &&
         c->mprekey_n == c->n && c->mprekey_center[0] == c->center[0] && c->mprekey_center[1] == c->center[1] &&
         c->mprekey_center[2] == c->center[2] && c->pending_kick == 0.0;
}

// ---- the appended store (particles.h; sph_kernels.h: AppDev) ---------------------------------------------------------------
// layout of a buffer set from the cells' populations: region c = [base[c], base[c + 1]), population + 1/64 of it + 192 slots,
// whole waves (a wave of the passes then never straddles two regions); the tail behind the last region; cursors and flag cleared
__global__ void __launch_bounds__(1024)
k_app_layout(const uint32_t *__restrict__ counts, uint32_t ncell, uint32_t tail, uint32_t *__restrict__ base,
             uint32_t *__restrict__ range, uint32_t *__restrict__ cursor, uint32_t *__restrict__ also_into, int tight)
{
  __shared__ uint32_t part[1024];
  const uint32_t per = (ncell + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per < ncell ? threadIdx.x * per : ncell, hi = lo + per < ncell ? lo + per : ncell;
  // (tight: no slack at all -- the test mode in which a pass runs out of room as soon as a population grows)
  const auto room = [tight](uint32_t n) { return tight ? (n + 63u) & ~63u : (n + n / 64u + 192u + 63u) & ~63u; };
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; k++) sum += room(counts[k]);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t n = counts[k];
    base[k] = run;
    if (also_into) also_into[k] = run;
    run += room(n);
  }
  __syncthreads();            // (counts may alias cursor: every count has been read)
  for (uint32_t k = lo; k < hi; k++) cursor[k] = 0u;
  if (threadIdx.x == 1023) {
    base[ncell] = part[1023];
    base[ncell + 1] = part[1023] + tail;
    range[0] = 0u;
    range[1] = part[1023] + tail;
    cursor[ncell] = 0u;          // tail
    cursor[ncell + 1] = 0u;      // flag
  }
}

// after the passes have placed their particles in a buffer set: x = +inf behind the filled part of every region and of the tail
#define APP_MARK_TAIL 256u
__global__ void __launch_bounds__(256)
k_app_mark(const uint32_t *__restrict__ base, const uint32_t *__restrict__ fill /* per cell: cursor, or the scatter's running
           offsets (absolute != 0) */, int absolute, uint32_t ncell, double *__restrict__ X)
{
  // ncell blocks for the regions, APP_MARK_TAIL more for the tail (one block alone would write its n / 128 slots in 0.1 ms)
  const uint32_t c = blockIdx.x < ncell ? blockIdx.x : ncell;
  const uint32_t b = base[c], e = base[c + 1];
  uint32_t f = absolute ? (c < ncell ? fill[c] - b : 0u) : fill[c];
  if (f > e - b) f = e - b;
  const uint32_t part = c < ncell ? 0u : blockIdx.x - ncell, nparts = c < ncell ? 1u : APP_MARK_TAIL;
  for (uint32_t s = b + f + part * 256u + threadIdx.x; s < e; s += 256u * nparts) X[s] = APP_EMPTY;
}

__global__ void __launch_bounds__(256)
k_app_fill(double *__restrict__ v, size_t n, double x)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = x;
}

// an appended set -> a dense one: the slots that hold a particle, block by block to a place reserved with one atomic
__global__ void __launch_bounds__(256)
k_app_densify(const double *__restrict__ X, const uint32_t *__restrict__ range, uint32_t *__restrict__ counter,
              const double *const *__restrict__ src, double *const *__restrict__ dst, int narr,
              const uint32_t *__restrict__ id, uint32_t *__restrict__ id_out,
              const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
              const uint32_t *__restrict__ pslot /* where the positions are: px[pslot[i]] (nullptr: px[i]) */,
              double *__restrict__ ox, double *__restrict__ oy, double *__restrict__ oz)
{
  __shared__ uint32_t s_cnt, s_base;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) s_cnt = 0u;
  __syncthreads();
  const bool have = i < range[1] && X[i] != APP_EMPTY;
  uint32_t my = 0;
  if (have) my = atomicAdd(&s_cnt, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(counter, s_cnt);
  __syncthreads();
  if (!have) return;
  const size_t o = (size_t)s_base + my;
  for (int a = 0; a < narr; a++) dst[a][o] = src[a][i];
  id_out[o] = id[i];
  const size_t q = pslot ? (size_t)pslot[i] : i;
  ox[o] = px[q]; oy[o] = py[q]; oz[o] = pz[q];
}

int expamd_comp_app_reserve(exp_amd_comp *c, size_t cap)
{
  exp_amd_ctx *ctx = c->ctx;
  if (!c->app_hflag) {
    HIP_TRY(ctx, hipHostMalloc((void **)&c->app_hflag, 4 * sizeof(uint32_t), hipHostMallocDefault));
    c->app_hflag[0] = c->app_hflag[1] = 0u;
  }
  if (!c->app_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&c->app_ev, hipEventDisableTiming));
  if (c->app_cap >= cap) return EXP_AMD_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  {
    // the larger arrays are made one at a time next to the ones they replace: is there room for the growth (and the new
    // index / scratch arrays) plus one array in flight?  If not the component keeps the ordinary step (app_refused)
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    const size_t grow = (cap - c->n) * (2 * (A_NARR * sizeof(double) + sizeof(uint32_t))) +
                        cap * (2 * (3 * sizeof(double) + sizeof(uint32_t))) + cap * sizeof(double);
    if (free_b < grow + (size_t)(1u << 30)) { c->app_refused = true; return EXP_AMD_OK; }
  }
  for (int w = 0; w < 2; w++) {
    for (int a = 0; a < A_NARR; a++) {
      DevBuf<double> nb;
      HIP_TRY(ctx, nb.alloc(cap));
      if (w == c->cur && c->n) HIP_TRY(ctx, hipMemcpy(nb.p, c->arr[w][a].p, c->n * sizeof(double), hipMemcpyDeviceToDevice));
      c->arr[w][a].release();
      c->arr[w][a] = nb;
    }
    DevBuf<uint32_t> ni;
    HIP_TRY(ctx, ni.alloc(cap));
    if (w == c->cur && c->n) HIP_TRY(ctx, hipMemcpy(ni.p, c->id[w].p, c->n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    c->id[w].release();
    c->id[w] = ni;
    for (int k = 0; k < 3; k++) HIP_TRY(ctx, c->xo[w][k].alloc(cap));
    HIP_TRY(ctx, c->app_src[w].alloc(cap));
  }
  // a uniform mass lives as the constant in the mass arrays of both sets: in all of their slots now
  if (c->uniform_mass)
    for (int w = 0; w < 2; w++) k_app_fill<<<4096, 256, 0, ctx->stream>>>(c->arr[w][A_M].p, cap, c->mass_value);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  c->app_cap = cap;
  return EXP_AMD_OK;
}

int expamd_comp_app_layout(exp_amd_comp *c, const uint32_t *counts, uint32_t ncell, int set, uint32_t *also_into)
{
  exp_amd_ctx *ctx = c->ctx;
  if (c->app_base[0].n < (size_t)ncell + 2) {
    for (int w = 0; w < 2; w++) { HIP_TRY(ctx, c->app_base[w].alloc((size_t)ncell + 2)); HIP_TRY(ctx, c->app_range[w].alloc(2)); }
    HIP_TRY(ctx, c->app_cursor.alloc((size_t)ncell + 2));
  }
  const bool tight = ctx->append_min < 0;
  c->app_ncell = ncell;
  c->app_tail = tight ? 64u : expamd_app_tail(c->n);
  k_app_layout<<<1, 1024, 0, ctx->stream>>>(counts, ncell, c->app_tail, c->app_base[set].p, c->app_range[set].p,
                                             c->app_cursor.p, also_into, tight ? 1 : 0);
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

int expamd_comp_app_finish(exp_amd_comp *c, int set, uint32_t *lost)
{
  exp_amd_ctx *ctx = c->ctx;
  k_app_mark<<<c->app_ncell + APP_MARK_TAIL, 256, 0, ctx->stream>>>(c->app_base[set].p, c->app_cursor.p, 0, c->app_ncell, c->arr[set][A_X].p);
  HIP_TRY(ctx, hipGetLastError());
  // {arrivals in the tail, particles that found no room at all}: to the page-locked words, behind an event (particles.h)
  HIP_TRY(ctx, hipMemcpyAsync(c->app_hflag, c->app_cursor.p + c->app_ncell, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(c->app_ev, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(c->app_ev));
  c->app_tail_used = c->app_hflag[0];
  *lost = c->app_hflag[1];
  static const bool app_debug = getenv("EXP_AMD_APP_DEBUG") != nullptr;
  if (app_debug) fprintf(stderr, "append step: %u particles in the tail, %u without room\n", c->app_tail_used, *lost);
  return EXP_AMD_OK;
}

void k_app_mark_launch(exp_amd_comp *c, int set, const uint32_t *offs)
{
  k_app_mark<<<c->app_ncell + APP_MARK_TAIL, 256, 0, c->ctx->stream>>>(c->app_base[set].p, offs, 1, c->app_ncell, c->arr[set][A_X].p);
}

int expamd_comp_densify(exp_amd_comp *c, bool state_positions)
{
  if (!c->appended) return EXP_AMD_OK;
  exp_amd_ctx *ctx = c->ctx;
  expamd_mutated();
  const int s = c->cur, d = 1 - c->cur;
  // positions: those of the completed step (xo), or -- a step that is being redone from its source -- the advanced ones
  const double *srcs[A_NARR];
  double *dsts[A_NARR];
  int na = 0;
  for (int a = A_VX; a < A_NARR; a++) {
    if (a == A_M && c->uniform_mass) continue;                   // (both sets hold the constant)
    srcs[na] = c->arr[s][a].p;
    dsts[na] = c->arr[d][a].p;
    na++;
  }
  // positions: state_positions -- those of the completed step, which are still in the OTHER set (the set this call fills:
  // they are gathered into scratch arrays that then take the other set's place); else the live set's own (a step being redone)
  const double *px = state_positions ? c->arr[d][A_X].p : c->arr[s][A_X].p;
  const double *py = state_positions ? c->arr[d][A_Y].p : c->arr[s][A_Y].p;
  const double *pz = state_positions ? c->arr[d][A_Z].p : c->arr[s][A_Z].p;
  // the pointer tables and the counter: scratch words behind the level sweeps' counters
  DevBuf<unsigned char> tab;
  HIP_TRY(ctx, tab.alloc(2 * A_NARR * sizeof(void *) + 16));
  HIP_TRY(ctx, hipMemcpyAsync(tab.p, srcs, na * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(tab.p + A_NARR * sizeof(void *), dsts, na * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
  uint32_t *counter = (uint32_t *)(tab.p + 2 * A_NARR * sizeof(void *));
  HIP_TRY(ctx, hipMemsetAsync(counter, 0, sizeof(uint32_t), ctx->stream));
  k_app_densify<<<cdiv(c->app_ns, 256), 256, 0, ctx->stream>>>(c->arr[s][A_X].p, c->app_range[s].p, counter,
                                                               (const double *const *)tab.p,
                                                               (double *const *)(tab.p + A_NARR * sizeof(void *)), na,
                                                               c->id[s].p, c->id[d].p, px, py, pz,
                                                               state_positions ? c->app_src[s].p : nullptr,
                                                               c->xo[d][0].p, c->xo[d][1].p, c->xo[d][2].p);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t got = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&got, counter, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  tab.release();
  if ((size_t)got != c->n)
    return expamd_fail(ctx, EXP_AMD_ERR_STATE, "appended store: %u particles found, %zu expected", got, c->n);
  for (int k = 0; k < 3; k++) std::swap(c->arr[d][A_X + k], c->xo[d][k]);
  c->app_wait = c->app_backoff;
  c->app_backoff = c->app_run >= 64 ? 8 : (c->app_backoff < 1024 ? 2 * c->app_backoff : 1024);
  c->app_run = 0;
  c->cur = d;
  c->appended = false;
  void *const owner = const_cast<void *>(c->app_owner);
  const bool reeval = state_positions && c->app_acc_stale && c->app_reeval && owner;
  c->app_acc_stale = false;
  c->app_owner = nullptr;
  expamd_app_unlist(c);
  c->sorted_for = nullptr;
  c->prekey_valid = false;
  c->split = false;
  c->hist_clean = 0;
  c->lev_host_valid = false;
  const uint32_t lo1[2] = {0u, (uint32_t)c->n};
  HIP_TRY(ctx, hipMemcpyAsync(c->lev_off.p, lo1, sizeof(lo1), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the state's acceleration and potential, which the placing passes did not carry along (particles.h: app_acc_stale)
  if (reeval) return c->app_reeval(owner, c);
  return EXP_AMD_OK;
}

void expamd_app_unlist(exp_amd_comp *c)
{
  auto &v = c->ctx->appended;
  for (size_t k = 0; k < v.size(); k++)
    if (v[k] == c) { v.erase(v.begin() + k); break; }
}

int expamd_comp_touch(exp_amd_comp *c)
{
  expamd_mutated();
  c->prekey_valid = false;
  c->mprekey_valid = false;
  if (c->commit_pending) { int rc = expamd_comp_flush_commit(c); if (rc) return rc; }
  if (c->partition_stale) { int rc = partition_by_level(c); if (rc) return rc; }
  return expamd_comp_apply_pending(c);
}

int expamd_comp_touch_keep_partition(exp_amd_comp *c)
{
  expamd_mutated();
  c->prekey_valid = false;
  return expamd_comp_apply_pending(c);
}

// A read-only consumer of the step-boundary velocities (fix_positions, Orient::accumulate): a closing half-kick
// still owed (pending_kick > 0) is applied -- exactly the operation the next step would do first --, velocities
// that are AHEAD by the next opening half-kick (pending_kick < 0, prekick) stay as they are and the caller forms
// v + a * (*back) on the fly, as exp_amd_comp_download does: un-kicking and re-kicking a rounded operation would
// move the trajectory by an ulp just because a diagnostic looked.  The keys recorded for the next step stay valid.
int expamd_comp_velocity_view(exp_amd_comp *c, double *back)
{
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }
  { int rc_ = expamd_comp_flush_commit(c); if (rc_) return rc_; }      // (these consumers read the level array too)
  *back = c->pending_kick < 0.0 ? c->pending_kick : 0.0;
  if (*back != 0.0) return EXP_AMD_OK;
  return expamd_comp_apply_pending(c);
}

int expamd_comp_apply_pending(exp_amd_comp *c)
{
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }     // (an appended store: every outside call sees an ordinary one)
  if (c->pending_kick != 0.0 && c->n) {
    const double dt = c->pending_kick;
    const int lo = c->pending_lo < c->nlevels ? c->pending_lo : 0;
    c->pending_kick = 0.0;
    c->pending_lo = 0;
    ProfScope ps(c->ctx, "k_kick");
    k_kick<<<stream_grid(c->ctx, c->n), TPB, 0, c->ctx->stream>>>(
        c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->lev_off.p, lo,
        c->nlevels - 1, dt);
    HIP_TRY(c->ctx, hipGetLastError());
  }
  c->pending_kick = 0.0;
  c->pending_lo = 0;
  return EXP_AMD_OK;
}

AdvanceArgs expamd_advance_args(exp_amd_comp *c, const AdvSpec &adv)
{
  AdvanceArgs A;
  A.x = c->a(A_X); A.y = c->a(A_Y); A.z = c->a(A_Z);
  A.vx = c->a(A_VX); A.vy = c->a(A_VY); A.vz = c->a(A_VZ);
  A.ax = c->a(A_AX); A.ay = c->a(A_AY); A.az = c->a(A_AZ);
  A.lev = c->commit_pending ? c->newlev.p : c->level[c->cur].p;     // (the caller has checked that the pass stays within
                                                                      // the slots the last sweep examined)
  A.dt_kick = adv.dt_kick; A.dt_drift = adv.dt_drift;
  A.dt_kick0 = adv.mode ? c->pending_kick : 0.0;     // deferred half-kick of the last fused step
  A.kick0_lo = c->pending_lo;
  // ... or the opposite: that force pass stored the velocities with THIS opening half-kick applied already
  // (pending_kick == -dt_kick, set together with the step's sort keys): nothing left to kick
  A.nokick = (adv.mode == 1 && c->pending_kick != 0.0 && c->pending_kick == -adv.dt_kick) ? 1 : 0;
  if (A.nokick) A.dt_kick0 = 0.0;
  A.advance = adv.mode;
  A.multistep = adv.multistep;
  A.lev_lo = adv.lev_lo;
  A.dt_min = adv.dt_min;
  return A;
}

// after k_key_hist: scan the histogram, scatter (with the same advance) into the other buffer set
SortRange expamd_sort_range(exp_amd_comp *c, int level, int level_hi)
{
  SortRange R;
  R.lev_off = (level >= 0) ? c->lev_off.p : nullptr;
  R.lo = R.hi = level < 0 ? 0 : level;
  if (level >= 0 && level_hi > level) R.hi = level_hi;
  R.n = c->n;
  return R;
}

// copy the slot range of one level back from the scatter target into the live buffer set
struct CopySet {
  double *dst[A_NARR];
  const double *src[A_NARR];
  uint32_t *did;
  const uint32_t *sid;
  int narr;
};

// hist[hz0, hz1): the histogram bins this range sort used (offsets by now), zeroed on the way: the next sort of this
// component then starts from a clean histogram without a memset (exp_amd_comp::hist_clean)
__global__ void __launch_bounds__(TPB)
k_copy_range(CopySet C, const uint32_t *__restrict__ lev_off, int level, int level_hi,
             uint8_t *__restrict__ dlev, const uint8_t *__restrict__ slev, uint32_t *__restrict__ hist, uint32_t hz0,
             uint32_t hz1)
{
  for (size_t k = hz0 + (size_t)blockIdx.x * TPB + threadIdx.x; k < hz1; k += (size_t)gridDim.x * TPB) hist[k] = 0u;
  const size_t beg = lev_off[level], end = lev_off[level_hi + 1];
  const size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= end) return;
  for (int a = 0; a < C.narr; a++) C.dst[a][i] = C.src[a][i];
  C.did[i] = C.sid[i];
  if (dlev) dlev[i] = slev[i];          // several levels: the level of a slot may have changed
}

int expamd_comp_finish_sort(exp_amd_comp *c, uint32_t nkeys, uint32_t ncell, bool move_acc,
                            const AdvSpec &adv, int level, int level_hi)
{
  const bool advance = adv.mode != 0;
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  if (level < 0 || level_hi < level) level_hi = level;
  // launches over a level range are sized for its population (host mirror of lev_off), not for n:
  // the upper levels are small and are sorted every sub-step
  size_t nr = c->n;
  if (level >= 0) { int rc_ = expamd_comp_level_count(c, level, level_hi, &nr); if (rc_) return rc_; }
  if (nr == 0) return EXP_AMD_OK;
  {
    ProfScope ps(ctx, "k_scan");
    { int rc_ = scan_launch(ctx, ctx->stream, c->hist.p, nkeys, c->lev_off.p, ncell, c->nlevels, level, level_hi); if (rc_) return rc_; }
    if (level < 0 || level_hi > level) c->lev_host_valid = false;
  }
  {
    ProfScope ps(ctx, "k_scatter_adv");
    AdvanceArgs A = expamd_advance_args(c, adv);
    ScatterSrc S{c->a(A_M), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), c->id[c->cur].p};
    ScatterDst D{c->b(A_X), c->b(A_Y), c->b(A_Z), c->b(A_VX), c->b(A_VY), c->b(A_VZ),
                 c->uniform_mass ? nullptr : c->b(A_M),
                 c->b(A_AX), c->b(A_AY), c->b(A_AZ), c->b(A_POT), c->id[1 - c->cur].p,
                 c->levels_zero ? nullptr : c->level[1 - c->cur].p};
    const SortRange R = expamd_sort_range(c, level, level_hi);
    if (nr <= SCAT_SHORT_MAX && level >= 0) {
      // (a level range of a sub-step: one slot a thread -- four times the blocks for a pass that is all latency there)
      const unsigned g = cdiv(nr, (size_t)SORT_TPB * SCAT_ITEMS_SHORT);
      if (move_acc)
        k_scatter_adv<true, SCAT_ITEMS_SHORT><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p);
      else
        k_scatter_adv<false, SCAT_ITEMS_SHORT><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p);
    } else {
    const uint32_t win = (level < 0 && c->sort_win) ? c->sort_win : (uint32_t)SORT_WIN;
    if (win < (uint32_t)SORT_WIN) {
      // (the dense one-level sort of the sphere's fused step: short window, short tiles -- sort_kernels.h)
      const unsigned g = cdiv(nr, (size_t)SORT_TPB * SCAT_ITEMS_DENSE);
      if (move_acc)
        k_scatter_adv<true, SCAT_ITEMS_DENSE><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p, win);
      else
        k_scatter_adv<false, SCAT_ITEMS_DENSE><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p, win);
    } else {
    const unsigned g = cdiv(nr, SCAT_TILE);
    if (move_acc)
      k_scatter_adv<true><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p, win);
    else
      k_scatter_adv<false><<<g, SORT_TPB, 0, ctx->stream>>>(A, S, D, R, c->key.p, c->hist.p, win);
    }
    }
    c->sort_win = 0;
  }
  HIP_TRY(ctx, hipGetLastError());
  // the scatter applied the half-kick still owed ahead of its own kick (a range sort: to the levels it holds -- callers
  // see to it that those are all of the owing levels or none, expamd_comp_settle_pending)
  if (advance && (level < 0 || c->pending_lo == 0 ||
                  (level <= c->pending_lo && (level_hi > level ? level_hi : level) >= c->nlevels - 1))) {
    c->pending_kick = 0.0;
    c->pending_lo = 0;
  }
  if (level < 0) {
    c->split = false;                      // one global order again
    c->cur = 1 - c->cur;
    return EXP_AMD_OK;
  }
  // one level only: bring its range back (levels do not change inside a level sort, so the u8
  // level array is already right)
  {
    ProfScope ps(ctx, "k_copy_range");
    CopySet C;
    C.narr = move_acc ? A_NARR : A_AX;
    for (int a = 0; a < A_NARR; a++) { C.dst[a] = c->a(a); C.src[a] = c->b(a); }
    C.did = c->id[c->cur].p;
    C.sid = c->id[1 - c->cur].p;
    const bool many = level_hi > level || c->commit_pending;
    // (range mode of the scan: only the bins of levels level..level_hi were populated, the rest is still zero)
    const uint32_t hz0 = (uint32_t)level * ncell, hz1 = (uint32_t)(level_hi + 1) * ncell;
    k_copy_range<<<cdiv(nr, TPB), TPB, 0, ctx->stream>>>(
        C, c->lev_off.p, level, level_hi, many ? c->level[c->cur].p : nullptr,
        many ? c->level[1 - c->cur].p : nullptr, c->hist.p, hz0, hz1);
    c->hist_clean = (size_t)nkeys + 1;
  }
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// ---- multistep level selection --------------------------------------------------------------------
// adjust_multistep_level_thread (src/multistep.cc:52-236) for the particles of levels
// [first, multistep]: five time-step criteria -> dtreq (rounded to float like Particle::dtreq,
// include/Particle.H:60) -> target level.  Levels are only PROPOSED here (newlev); the force
// method applies its coefficient differencing and the store commits + re-sorts afterwards.
__global__ void __launch_bounds__(TPB)
k_adjust_levels(AdjustArgs A, const double *__restrict__ vx, const double *__restrict__ vy,
                const double *__restrict__ vz, const double *__restrict__ ax,
                const double *__restrict__ ay, const double *__restrict__ az,
                const double *__restrict__ pot, const uint8_t *__restrict__ lev,
                uint8_t *__restrict__ newlev, const uint32_t *__restrict__ lev_off, int first,
                int last, size_t n, unsigned long long *__restrict__ nswitch, NsArgs N)
{
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const unsigned plev = lev[i];
  unsigned nlev = plev;
  if (i >= lev_off[first] && i < lev_off[last + 1]) {
    const double eps = 1.0e-10;
    const double v0 = vx[i], v1 = vy[i], v2 = vz[i], a0 = ax[i], a1 = ay[i], a2 = az[i];
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
    const float dtreq = ns_dtreq(N, i, dt, apply);       // ("noswitch": kick_adjust.h)
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
    if (nlev != plev) atomicAdd(nswitch, 1ull);
  }
  newlev[i] = (uint8_t)nlev;
}

__global__ void __launch_bounds__(TPB)
k_commit_levels(uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev, size_t beg, size_t n)
{
  const size_t i = beg + (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) lev[i] = newlev[i];
}

// the "noswitch" arguments of the sweep about to be launched (exp_amd_comp::ns_reset / ns_apply: set by its caller)
static NsArgs comp_ns_args(exp_amd_comp *c)
{
  NsArgs N;
  if (c->noswitch && c->d_dtreq.p) {
    N.dtreq = c->d_dtreq.p;
    N.id = c->id[c->cur].p;
    N.reset = c->ns_reset;
    N.apply = c->ns_apply;
  }
  return N;
}

int expamd_comp_propose_levels(exp_amd_comp *c, double dtime, const double dynfrac[5], int shiftlevl,
                               int multistep, int mfirst_mdrft, int first)
{
  exp_amd_ctx *ctx = c->ctx;
  // (the counter of this entry point is nswitch[64]; [0, 64) are the step driver's two counter sets)
  HIP_TRY(ctx, hipMemsetAsync(c->nswitch.p + 64, 0, sizeof(unsigned long long), ctx->stream));
  if (c->n == 0) return EXP_AMD_OK;
  AdjustArgs A{dtime, dynfrac[0], dynfrac[1], dynfrac[2], dynfrac[3], dynfrac[4], multistep,
               shiftlevl, mfirst_mdrft};
  ProfScope ps(ctx, "k_adjust_levels");
  k_adjust_levels<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(
      A, c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT),
      c->level[c->cur].p, c->newlev.p, c->lev_off.p, first, multistep, c->n, c->nswitch.p + 64, comp_ns_args(c));
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

int expamd_comp_commit_levels(exp_amd_comp *c, size_t beg)
{
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0 || beg >= c->n) return EXP_AMD_OK;
  c->levels_zero = false;
  k_commit_levels<<<cdiv(c->n - beg, TPB), TPB, 0, ctx->stream>>>(c->level[c->cur].p, c->newlev.p, beg, c->n);
  HIP_TRY(ctx, hipGetLastError());
  c->sorted_for = nullptr;
  return EXP_AMD_OK;
}

// Compaction of the movers (level != proposed level) of the slot range of levels [first, last].  Order-preserving
// inside a block of 256 x ML_ITEMS slots (per-thread bit masks + a block-level prefix); the blocks claim their stretch
// of the list with one atomic, in any order -- long stretches, so that where movers are dense (the case this list is
// for) a 64-entry group of the consumer still sees the (level, cell) runs of the store.  cnt = {0, count}.
#define ML_ITEMS 16
__global__ void __launch_bounds__(256)
k_mover_list(const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev,
             const uint32_t *__restrict__ lev_off, int first, int last, uint32_t *__restrict__ list,
             uint32_t cap, uint32_t *__restrict__ cnt, uint32_t *__restrict__ cnt_next /* the NEXT call's pair: zeroed here */)
{
  __shared__ uint32_t wsum[4], base;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cnt_next[0] = 0u; cnt_next[1] = 0u; }
  const size_t beg = lev_off[first], end = lev_off[last + 1];
  const size_t abeg = beg & ~(size_t)(ML_ITEMS - 1);          // 16-byte aligned reads
  const size_t s0 = abeg + ((size_t)blockIdx.x * 256 + threadIdx.x) * ML_ITEMS;
  uint32_t mask = 0;                                           // bit k: slot s0 + k is a mover
  if (s0 >= beg && s0 + ML_ITEMS <= end) {
    const uint4 a = *reinterpret_cast<const uint4 *>(lev + s0), b = *reinterpret_cast<const uint4 *>(newlev + s0);
    const uint32_t d[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
#pragma unroll
    for (int k = 0; k < ML_ITEMS; k++) if ((d[k >> 2] >> (8 * (k & 3))) & 0xffu) mask |= 1u << k;
  } else if (s0 < end) {
    for (int k = 0; k < ML_ITEMS; k++) {
      const size_t i = s0 + k;
      if (i >= beg && i < end && lev[i] != newlev[i]) mask |= 1u << k;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t mine = (uint32_t)__popc(mask);
  uint32_t incl = mine;                                        // inclusive prefix over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    base = tot ? atomicAdd(cnt + 1, tot) : 0u;
  }
  __syncthreads();
  if (mask) {
    uint32_t o = base + incl - mine;
    for (int w = 0; w < wave; w++) o += wsum[w];
    for (uint32_t m = mask; m; m &= m - 1u) {
      if (o < cap) list[o] = (uint32_t)(s0 + (size_t)(__ffs((int)m) - 1));
      o++;
    }
  }
}

int expamd_comp_mover_list(exp_amd_comp *c, int first, int last, size_t expected)
{
  exp_amd_ctx *ctx = c->ctx;
  if (c->mover_list_built) return EXP_AMD_OK;      // (k_kick_adjust did it: c->mover_cnt is its pair)
  // two {0, count} pairs used alternately: a call clears the pair the next one will count into (no memset in between)
  if (c->mover_cnt_buf.n == 0) {
    HIP_TRY(ctx, c->mover_cnt_buf.alloc(4));
    HIP_TRY(ctx, hipMemsetAsync(c->mover_cnt_buf.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    c->mover_flip = 0;
  }
  if (c->mover_list.n < expected) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    c->mover_list.release();
    HIP_TRY(ctx, c->mover_list.alloc(expected + expected / 4 + 1024));
  }
  uint32_t *cnt = c->mover_cnt_buf.p + 2 * c->mover_flip, *nxt = c->mover_cnt_buf.p + 2 * (1 - c->mover_flip);
  c->mover_cnt = cnt;
  size_t nr = 0;
  if (c->n) { int rc = expamd_comp_level_count(c, first, last, &nr); if (rc) return rc; }
  if (nr == 0) return EXP_AMD_OK;      // (the pair stays clean and is used again)
  k_mover_list<<<cdiv(nr + ML_ITEMS, (size_t)256 * ML_ITEMS), 256, 0, ctx->stream>>>(c->level[c->cur].p, c->newlev.p, c->lev_off.p, first, last,
                                                               c->mover_list.p, (uint32_t)c->mover_list.n, cnt, nxt);
  c->mover_flip ^= 1;
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// launches k_kick_adjust; *result receives the device address of this launch's counter set
// (u64[32]: level changes, then the proposals per level)
int expamd_comp_kick_adjust(exp_amd_comp *c, double dtime, const double dynfrac[5], int shiftlevl,
                            int multistep, int mfirst_mdrft, int kick_lo, int first, double dt_min,
                            const unsigned long long **result, unsigned long long *host_out, unsigned long long seq,
                            bool *launched, bool build_list, void (*key_launch)(void *, const KaLaunch &), void *key_self)
{
  exp_amd_ctx *ctx = c->ctx;
  unsigned long long *out = c->nswitch.p + 32 * c->nsw_flip, *nxt = c->nswitch.p + 32 * (1 - c->nsw_flip);
  *result = out;
  if (launched) *launched = false;
  const int lo = kick_lo < first ? kick_lo : first;
  size_t nr = 0;
  if (c->n) { int rc = expamd_comp_level_count(c, lo, multistep, &nr); if (rc) return rc; }
  if (nr == 0) return EXP_AMD_OK;        // (the set stays clean and is used again)
  AdjustArgs A{dtime, dynfrac[0], dynfrac[1], dynfrac[2], dynfrac[3], dynfrac[4], multistep,
               shiftlevl, mfirst_mdrft};
  // build_list: the slots of the movers are compacted by the sweep itself (room for every examined slot)
  uint32_t *lcnt = nullptr, *lnxt = nullptr;
  c->mover_list_built = false;
  if (build_list && first <= multistep) {
    size_t ne = 0;
    { int rc = expamd_comp_level_count(c, first, multistep, &ne); if (rc) return rc; }
    if (c->mover_cnt_buf.n == 0) {
      HIP_TRY(ctx, c->mover_cnt_buf.alloc(4));
      HIP_TRY(ctx, hipMemsetAsync(c->mover_cnt_buf.p, 0, 4 * sizeof(uint32_t), ctx->stream));
      c->mover_flip = 0;
    }
    if (c->mover_list.n < ne) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      c->mover_list.release();
      HIP_TRY(ctx, c->mover_list.alloc(c->n > ne ? c->n : ne));
    }
    lcnt = c->mover_cnt_buf.p + 2 * c->mover_flip;
    lnxt = c->mover_cnt_buf.p + 2 * (1 - c->mover_flip);
    c->mover_cnt = lcnt;
    c->mover_flip ^= 1;
    c->mover_list_built = true;
  }
  ProfScope ps(ctx, "k_kick_adjust");
  const size_t tiles = cdiv(nr, (size_t)KA_TPB);
  const int items = tiles >= 4096 * KA_ITEMS ? KA_ITEMS : (int)(tiles / 4096 > 1 ? tiles / 4096 : 1);
  KaLaunch L{cdiv(tiles, (size_t)items), ctx->stream, A, c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ),
             c->a(A_POT), c->level[c->cur].p, c->newlev.p, c->lev_off.p, kick_lo, first, multistep, dt_min, out, nxt, items,
             (unsigned int *)(c->nswitch.p + 70), host_out, seq, lcnt ? c->mover_list.p : nullptr, lcnt, lnxt,
             KaKeyArgs{c->a(A_X), c->a(A_Y), c->a(A_Z), c->key.p}, comp_ns_args(c)};
  c->mprekey_valid = false;
  // (key_launch: the sweep that closes a master step, every slot examined: the force method's instantiation writes the
  // next sub-step's sort keys on the way -- host.hip, kick_adjust.h)
  if (key_launch && lo == 0 && first == 0 && nr == c->n) key_launch(key_self, L);
  else { key_launch = nullptr; ka_launch_with(L, KaNoKey()); }
  HIP_TRY(ctx, hipGetLastError());
  c->mprekey_n = key_launch ? c->n : 0;        // (the caller completes the record: owner, step, centre)
  if (launched) *launched = true;
  c->nsw_flip ^= 1;
  return EXP_AMD_OK;
}

// ---- C ABI -----------------------------------------------------------------------------------

extern "C" int exp_amd_comp_create(exp_amd_ctx *ctx, size_t n, exp_amd_comp **out)
{
  if (!ctx || !out) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comp_create: NULL argument");
  if (n >= 0xffffffffull) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comp_create: n too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  exp_amd_comp *c = new exp_amd_comp;
  c->ctx = ctx;
  c->n = n;
  size_t na = n ? n : 1;
  for (int w = 0; w < 2; w++) {
    for (int a = 0; a < A_NARR; a++) {
      hipError_t e = c->arr[w][a].alloc(na);
      if (e != hipSuccess) {
        exp_amd_comp_destroy(c);
        return expamd_fail(ctx, EXP_AMD_ERR_HIP, "comp_create: hipMalloc failed: %s",
                           hipGetErrorString(e));
      }
    }
    if (c->id[w].alloc(na) != hipSuccess || c->level[w].alloc(na) != hipSuccess) {
      exp_amd_comp_destroy(c);
      return expamd_fail(ctx, EXP_AMD_ERR_HIP, "comp_create: hipMalloc failed");
    }
  }
  if (c->key.alloc(na) != hipSuccess || c->lev_off.alloc(64) != hipSuccess ||
      c->newlev.alloc(na) != hipSuccess || c->nswitch.alloc(72) != hipSuccess) {
    exp_amd_comp_destroy(c);
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "comp_create: hipMalloc failed");
  }
  for (int a = 0; a < A_NARR; a++)
    HIP_TRY(ctx, hipMemsetAsync(c->arr[0][a].p, 0, na * sizeof(double), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c->nswitch.p, 0, c->nswitch.bytes(), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c->level[0].p, 0, na, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(c->level[1].p, 0, na, ctx->stream));   // (levels_zero: never written then)
  k_iota<<<cdiv(na, TPB), TPB, 0, ctx->stream>>>(c->id[0].p, n);
  uint32_t lo[64];
  for (int i = 0; i < 64; i++) lo[i] = (uint32_t)n;
  lo[0] = 0;
  HIP_TRY(ctx, hipMemcpyAsync(c->lev_off.p, lo, sizeof(lo), hipMemcpyHostToDevice, ctx->stream));
  c->lev_host_valid = false;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *out = c;
  return EXP_AMD_OK;
}

extern "C" void exp_amd_comp_destroy(exp_amd_comp *c)
{
  if (!c) return;
  if (c->ctx->aux) (void)hipStreamSynchronize(c->ctx->aux);
  (void)hipStreamSynchronize(c->ctx->stream);
  expamd_forget_component(c->ctx, c);
  expamd_app_unlist(c);
  for (int w = 0; w < 2; w++) {
    for (int a = 0; a < A_NARR; a++) c->arr[w][a].release();
    c->id[w].release();
    c->level[w].release();
  }
  for (int w = 0; w < 2; w++) {
    for (int k = 0; k < 3; k++) c->xo[w][k].release();
    c->app_src[w].release();
    c->app_base[w].release();
    c->app_range[w].release();
  }
  c->app_cursor.release();
  if (c->app_ev) { (void)hipEventDestroy(c->app_ev); c->app_ev = nullptr; }
  if (c->app_hflag) { (void)hipHostFree(c->app_hflag); c->app_hflag = nullptr; }
  c->d_frz.release();
  c->d_escaped.release();
  c->d_dtreq.release();
  c->key.release();
  c->newlev.release();
  c->nswitch.release();
  c->hist.release();
  c->lev_off.release();
  c->half_off.release();
  c->com_lev.release();
  c->com_red.release();
  delete c;
}

extern "C" size_t exp_amd_comp_size(const exp_amd_comp *c) { return c ? c->n : 0; }

// min / max of a device array (two atomics per block on the order-preserving bit patterns)
__global__ void __launch_bounds__(TPB)
k_minmax(const double *__restrict__ v, size_t n, double *__restrict__ out /* {min, max}, preset */)
{
  double lo = 1.0e300, hi = -1.0e300;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) {
    const double x = v[i];
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    // masses are >= 0: their IEEE bit patterns order like unsigned integers
    atomicMin((unsigned long long *)out, (unsigned long long)__double_as_longlong(lo < 0.0 ? 0.0 : lo));
    atomicMax((unsigned long long *)(out + 1), (unsigned long long)__double_as_longlong(hi < 0.0 ? 0.0 : hi));
  }
}

// sum |v| (an upper bound is all that is needed: plain fp atomics)
__global__ void __launch_bounds__(TPB)
k_abs_sum(const double *__restrict__ v, size_t n, double *__restrict__ out)
{
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) s += fabs(v[i]);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0 && s != 0.0) unsafeAtomicAdd(out, s);
}

__global__ void __launch_bounds__(TPB)
k_fill_f64(double *__restrict__ v, size_t n, double x)
{
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) v[i] = x;
}

// after the mass array was (re)written: is it one value?  Then the other buffer set gets it too.
static int detect_uniform_mass(exp_amd_comp *c)
{
  exp_amd_ctx *ctx = c->ctx;
  c->uniform_mass = false;
  c->mass_abs_sum = 0.0;
  if (c->n == 0) return EXP_AMD_OK;
  double init[3] = {1.0e300, 0.0, 0.0}, got[3];
  double *scr = (double *)(c->nswitch.p + 66);   // three spare words of the counter block
  HIP_TRY(ctx, hipMemcpyAsync(scr, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  k_minmax<<<stream_grid(ctx, c->n), TPB, 0, ctx->stream>>>(c->a(A_M), c->n, scr);
  k_abs_sum<<<stream_grid(ctx, c->n), TPB, 0, ctx->stream>>>(c->a(A_M), c->n, scr + 2);
  HIP_TRY(ctx, hipMemcpyAsync(got, scr, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  c->mass_abs_sum = got[2] * (1.0 + 1e-9);         // (the fp sum itself is only known to rounding)
  if (got[0] == got[1] && got[0] >= 0.0) {
    k_fill_f64<<<stream_grid(ctx, c->n), TPB, 0, ctx->stream>>>(c->b(A_M), c->n, got[0]);
    HIP_TRY(ctx, hipGetLastError());
    c->uniform_mass = got[0] > 0.0;      // (a component of massless tracers keeps its stream)
    c->mass_value = got[0];
  }
  return EXP_AMD_OK;
}

static int upload_one(exp_amd_comp *c, int a, const double *h)
{
  exp_amd_ctx *ctx = c->ctx;
  if (!h) {
    HIP_TRY(ctx, hipMemsetAsync(c->a(a), 0, c->n * sizeof(double), ctx->stream));
    return EXP_AMD_OK;
  }
  // caller order -> current slot order through the scratch set
  HIP_TRY(ctx, hipMemcpyAsync(c->b(a), h, c->n * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  k_permute_f64<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(c->b(a), c->id[c->cur].p, c->n, c->a(a));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_upload(exp_amd_comp *c, const double *mass, const double *x,
                                   const double *y, const double *z, const double *vx,
                                   const double *vy, const double *vz)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (c) { c->app_wait = 0; c->app_backoff = 8; }      // (new particle data: the append step's hysteresis starts over)
  if (!c) return EXP_AMD_ERR_ARG;
  if (c->n == 0) return EXP_AMD_OK;
  const double *h[7] = {x, y, z, vx, vy, vz, mass};
  for (int a = 0; a < 7; a++) {
    int rc = upload_one(c, a, h[a]);
    if (rc) return rc;
  }
  HIP_TRY(c->ctx, hipStreamSynchronize(c->ctx->stream));
  c->sorted_for = nullptr;
  return detect_uniform_mass(c);
}

// ---- upload of positions as the caller holds them, with the expansion frame applied on the device ----------
// Basis::addFromArray / createFromReader (expui/BiorthBasis.cc:4616-4738, :4555-4562): every position becomes
// rot (x - ctr) before it is accumulated.  The caller's array is [n][3] (stride 3: x, y, z interleaved) or three
// columns (stride 1); it is copied as it lies into the scratch set and taken apart, shifted and rotated by one pass
// (numpy needs 0.17 s for the same on 1e7 particles: strided column copies and a [n,3] x [3,3] product).
struct Frame { double c[3], r[9]; int shift, rotate; };

__global__ void __launch_bounds__(TPB)
k_upload_frame(const double *__restrict__ b0, const double *__restrict__ b1, const double *__restrict__ b2, int stride,
               Frame F, const uint32_t *__restrict__ id, size_t n, double *__restrict__ X, double *__restrict__ Y,
               double *__restrict__ Z)
{
  const size_t s = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (s >= n) return;
  const size_t i = id[s];
  double p[3];
  if (stride == 1) { p[0] = b0[i]; p[1] = b1[i]; p[2] = b2[i]; }
  else {
#pragma unroll
    for (int k = 0; k < 3; k++) {                     // element 3 i + k of the array that lies across the three buffers
      const size_t f = 3 * i + k;
      p[k] = f < n ? b0[f] : (f < 2 * n ? b1[f - n] : b2[f - 2 * n]);
    }
  }
  if (F.shift) { p[0] -= F.c[0]; p[1] -= F.c[1]; p[2] -= F.c[2]; }
  if (F.rotate) {
    const double a = p[0], b = p[1], c = p[2];
    p[0] = F.r[0] * a + F.r[1] * b + F.r[2] * c;
    p[1] = F.r[3] * a + F.r[4] * b + F.r[5] * c;
    p[2] = F.r[6] * a + F.r[7] * b + F.r[8] * c;
  }
  X[s] = p[0]; Y[s] = p[1]; Z[s] = p[2];
}

static int upload_frame3(exp_amd_comp *c, int a0, const double *x, const double *y, const double *z, int stride,
                         const Frame &F)
{
  exp_amd_ctx *ctx = c->ctx;
  const size_t n = c->n, bytes = n * sizeof(double);
  const double *src[3] = {x, stride == 1 ? y : x + n, stride == 1 ? z : x + 2 * n};
  for (int k = 0; k < 3; k++)
    HIP_TRY(ctx, hipMemcpyAsync(c->b(a0 + k), src[k], bytes, hipMemcpyHostToDevice, ctx->stream));
  k_upload_frame<<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(c->b(a0), c->b(a0 + 1), c->b(a0 + 2), stride, F, c->id[c->cur].p, n,
                                                       c->a(a0), c->a(a0 + 1), c->a(a0 + 2));
  HIP_TRY(ctx, hipGetLastError());
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_upload_frame(exp_amd_comp *c, const double *mass, const double *x, const double *y,
                                         const double *z, const double *vx, const double *vy, const double *vz,
                                         int stride, const double center[3], const double rot[9])
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (c) { c->app_wait = 0; c->app_backoff = 8; }      // (new particle data: the append step's hysteresis starts over)
  if (!c || !x || (stride != 1 && stride != 3) || (stride == 1 && (!y || !z)) || (stride == 1 && vx && (!vy || !vz)))
    return EXP_AMD_ERR_ARG;
  if (c->n == 0) return EXP_AMD_OK;
  exp_amd_ctx *ctx = c->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Frame F{};
  for (int k = 0; k < 3; k++) { F.c[k] = center ? center[k] : 0.0; if (F.c[k] != 0.0) F.shift = 1; }
  for (int k = 0; k < 9; k++) { F.r[k] = rot ? rot[k] : (k % 4 == 0 ? 1.0 : 0.0); if (F.r[k] != (k % 4 == 0 ? 1.0 : 0.0)) F.rotate = 1; }
  int rc = upload_frame3(c, A_X, x, y, z, stride, F);
  if (rc) return rc;
  if (vx) {                                            // velocities are rotated, not shifted (:4566-4568)
    Frame V = F;
    V.shift = 0;
    if ((rc = upload_frame3(c, A_VX, vx, vy, vz, stride, V))) return rc;
  } else
    for (int a = A_VX; a <= A_VZ; a++) HIP_TRY(ctx, hipMemsetAsync(c->a(a), 0, c->n * sizeof(double), ctx->stream));
  if ((rc = upload_one(c, A_M, mass))) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  c->sorted_for = nullptr;
  return detect_uniform_mass(c);
}

extern "C" int exp_amd_comp_upload_acc(exp_amd_comp *c, const double *ax, const double *ay,
                                       const double *az, const double *pot)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  if (c->n == 0) return EXP_AMD_OK;
  const double *h[4] = {ax, ay, az, pot};
  for (int a = 0; a < 4; a++) {
    int rc = upload_one(c, A_AX + a, h[a]);
    if (rc) return rc;
  }
  HIP_TRY(c->ctx, hipStreamSynchronize(c->ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_upload_device(exp_amd_comp *c, const double *mass, const double *x,
                                          const double *y, const double *z, const double *vx,
                                          const double *vy, const double *vz)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (c) { c->app_wait = 0; c->app_backoff = 8; }      // (new particle data: the append step's hysteresis starts over)
  if (!c) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  const double *d[7] = {x, y, z, vx, vy, vz, mass};
  for (int a = 0; a < 7; a++) {
    if (!d[a]) {
      HIP_TRY(ctx, hipMemsetAsync(c->a(a), 0, c->n * sizeof(double), ctx->stream));
    } else {
      k_permute_f64<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(d[a], c->id[c->cur].p, c->n,
                                                              c->a(a));
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  c->sorted_for = nullptr;
  return detect_uniform_mass(c);
}

extern "C" int exp_amd_comp_upload_levels(exp_amd_comp *c, const int32_t *level)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c || !level) return expamd_fail(c ? c->ctx : nullptr, EXP_AMD_ERR_ARG, "comp_upload_levels: NULL argument");
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  c->levels_zero = false;
  int32_t *tmp = (int32_t *)c->b(A_X);   // scratch (8 B per slot >= 4 B needed)
  HIP_TRY(ctx, hipMemcpyAsync(tmp, level, c->n * sizeof(int32_t), hipMemcpyHostToDevice,
                              ctx->stream));
  k_permute_lev<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(tmp, c->id[c->cur].p, c->n,
                                                          c->level[c->cur].p);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  c->sorted_for = nullptr;   // level ranges are stale until the next sort
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_download(exp_amd_comp *c, double *mass, double *x, double *y,
                                     double *z, double *vx, double *vy, double *vz, double *ax,
                                     double *ay, double *az, double *pot)
{
  if (!c) return EXP_AMD_ERR_ARG;
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }
  // A fused step leaves the velocities either short of their closing half-kick (pending_kick > 0: applied
  // now, exactly the operation the next step would have done first) or AHEAD by the next step's opening
  // half-kick (pending_kick < 0).  The second state is left as it is -- undoing and redoing a rounded
  // operation would move the trajectory by an ulp just because someone looked -- and the velocities go
  // out through v + a * pending_kick.
  const double back = c->pending_kick < 0.0 ? c->pending_kick : 0.0;
  if (back == 0.0) { int rc_ = expamd_comp_apply_pending(c); if (rc_) return rc_; }
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  double *h[A_NARR] = {x, y, z, vx, vy, vz, mass, ax, ay, az, pot};
  for (int a = 0; a < A_NARR; a++) {
    if (!h[a]) continue;
    if (back != 0.0 && a >= A_VX && a <= A_VZ)
      k_unpermute_kicked<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(c->a(a), c->a(A_AX + (a - A_VX)), back,
                                                                   c->id[c->cur].p, c->n, c->b(a));
    else
    k_unpermute_f64<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(c->a(a), c->id[c->cur].p, c->n,
                                                              c->b(a));
    HIP_TRY(ctx, hipMemcpyAsync(h[a], c->b(a), c->n * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_download_levels(exp_amd_comp *c, int32_t *level)
{
  if (!c || !level) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (c->n == 0) return EXP_AMD_OK;
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }
  { int rc_ = expamd_comp_flush_commit(c); if (rc_) return rc_; }
  int32_t *tmp = (int32_t *)c->b(A_X);
  k_unpermute_lev<<<cdiv(c->n, TPB), TPB, 0, ctx->stream>>>(c->level[c->cur].p, c->id[c->cur].p,
                                                            c->n, tmp);
  HIP_TRY(ctx, hipMemcpyAsync(level, tmp, c->n * sizeof(int32_t), hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

static int comp_frz_refresh(exp_amd_comp *c);
extern "C" int exp_amd_comp_set_center(exp_amd_comp *c, const double center[3])
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c || !center) return EXP_AMD_ERR_ARG;
  for (int k = 0; k < 3; k++) c->center[k] = center[k];
  c->sorted_for = nullptr;
  return comp_frz_refresh(c);
}

const double *expamd_comp_frz(const exp_amd_comp *c) { return c->freeze_on ? c->d_frz.p : nullptr; }

// {com0, center, rtrunc^2} of Component::freeze to the device (called by the two setters that change them: rare -- the key is
// set once, the centre moves with an orientation estimator; ordered with every stream by being synchronous)
static int comp_frz_refresh(exp_amd_comp *c)
{
  if (!c->freeze_on) return EXP_AMD_OK;
  exp_amd_ctx *ctx = c->ctx;
  const double want[7] = {c->com0[0], c->com0[1], c->com0[2], c->center[0], c->center[1], c->center[2], c->rtrunc * c->rtrunc};
  if (!c->d_frz.p) HIP_TRY(ctx, c->d_frz.alloc(8));
  if (c->frz_valid && memcmp(want, c->frz_host, sizeof(want)) == 0) return EXP_AMD_OK;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(c->d_frz.p, want, sizeof(want), hipMemcpyHostToDevice));
  memcpy(c->frz_host, want, sizeof(want));
  c->frz_valid = true;
  return EXP_AMD_OK;
}

// Component::rtrunc and com0 (the "rtrunc" key of a component, src/Component.cc:69, :1023)
extern "C" int exp_amd_comp_set_rtrunc(exp_amd_comp *c, double rtrunc, const double com0[3])
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c || !(rtrunc > 0.0)) return expamd_fail(c ? c->ctx : nullptr, EXP_AMD_ERR_ARG, "comp_set_rtrunc: rtrunc must be positive");
  c->rtrunc = rtrunc;
  for (int k = 0; k < 3; k++) c->com0[k] = com0 ? com0[k] : 0.0;
  c->freeze_on = rtrunc < 1.0e20;
  return comp_frz_refresh(c);
}

extern "C" int exp_amd_comp_get_center(const exp_amd_comp *c, double center[3])
{
  if (!c || !center) return EXP_AMD_ERR_ARG;
  for (int k = 0; k < 3; k++) center[k] = c->center[k];
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_set_orientation(exp_amd_comp *c, const double body[9])
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  c->use_rot = body != nullptr;
  for (int k = 0; k < 9; k++) c->rot[k] = body ? body[k] : (k % 4 == 0 ? 1.0 : 0.0);
  c->sorted_for = nullptr;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_set_pseudo_accel(exp_amd_comp *c, const double accel[3], const double omega[3],
                                             const double domdt[3])
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  c->pseudo.center = accel != nullptr;
  c->pseudo.axis = omega != nullptr && domdt != nullptr;
  for (int k = 0; k < 3; k++) {
    c->pseudo.a[k] = accel ? accel[k] : 0.0;
    c->pseudo.om[k] = c->pseudo.axis ? omega[k] : 0.0;
    c->pseudo.dom[k] = c->pseudo.axis ? domdt[k] : 0.0;
  }
  return EXP_AMD_OK;
}

static void level_range(const exp_amd_comp *c, int mlevel, bool upward, int *lo, int *hi)
{
  if (mlevel < 0) { *lo = 0; *hi = c->nlevels - 1; }
  else if (upward) { *lo = mlevel; *hi = c->nlevels - 1; }
  else { *lo = mlevel; *hi = mlevel; }
  if (*lo > c->nlevels - 1) *lo = c->nlevels - 1;
}

extern "C" int exp_amd_comp_drift(exp_amd_comp *c, double dt, int mlevel)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  if (mlevel >= c->nlevels)
    return expamd_fail(c->ctx, EXP_AMD_ERR_ARG, "comp_drift: level %d beyond the component's %d level(s)", mlevel, c->nlevels);
  if (c->n == 0) return EXP_AMD_OK;
  int lo, hi;
  level_range(c, mlevel, false, &lo, &hi);
  ProfScope ps(c->ctx, "k_drift");
  k_drift<<<stream_grid(c->ctx, c->n), TPB, 0, c->ctx->stream>>>(
      c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->lev_off.p, lo, hi,
      dt);
  HIP_TRY(c->ctx, hipGetLastError());
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_kick(exp_amd_comp *c, double dt, int mlevel)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  if (mlevel >= c->nlevels)
    return expamd_fail(c->ctx, EXP_AMD_ERR_ARG, "comp_kick: level %d beyond the component's %d level(s)", mlevel, c->nlevels);
  if (c->n == 0) return EXP_AMD_OK;
  int lo, hi;
  level_range(c, mlevel, false, &lo, &hi);
  ProfScope ps(c->ctx, "k_kick");
  k_kick<<<stream_grid(c->ctx, c->n), TPB, 0, c->ctx->stream>>>(
      c->a(A_VX), c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->lev_off.p, lo,
      hi, dt);
  HIP_TRY(c->ctx, hipGetLastError());
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_zero_acc(exp_amd_comp *c, int mlevel)
{
  if (c) { int rc_ = expamd_comp_touch(c); if (rc_) return rc_; }
  if (!c) return EXP_AMD_ERR_ARG;
  // (as kick and drift: a level the store has not been partitioned into is refused, not clamped -- clamping zeroed
  // EVERY particle of a store that only had its levels uploaded; found by tests/fuzz/fuzz_store.py)
  if (mlevel > 0 && mlevel >= c->nlevels)
    return expamd_fail(c->ctx, EXP_AMD_ERR_ARG, "comp_zero_acc: level %d beyond the component's %d level(s)", mlevel, c->nlevels);
  if (c->n == 0) return EXP_AMD_OK;
  int lo, hi;
  level_range(c, mlevel, true, &lo, &hi);
  ProfScope ps(c->ctx, "k_zero_acc");
  k_zero_acc<<<stream_grid(c->ctx, c->n), TPB, 0, c->ctx->stream>>>(
      c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), c->lev_off.p, lo, hi);
  HIP_TRY(c->ctx, hipGetLastError());
  return EXP_AMD_OK;
}

// Component::freeze (src/Component.cc:4194-4202) for the sums below: frz = {com0[3], center[3], rtrunc^2} on the device
// (exp_amd_comp::d_frz), nullptr when rtrunc is not set; the reference's operation order
__device__ __forceinline__ bool comp_frozen(const double *__restrict__ F, double px, double py, double pz)
{
  if (!F) return false;
  const double dx = (px - F[0]) - F[3], dy = (py - F[1]) - F[4], dz = (pz - F[2]) - F[5];
  double r2 = dx * dx;
  r2 = mul_then_add(r2, dy, dy);
  r2 = mul_then_add(r2, dz, dz);
  return r2 > F[6];
}

// ---- centre of mass / velocity / acceleration (Component::fix_positions) -----------------------------
// src/Component.cc:3280-3351 (thread body: per-level sums of m, m x, m v, m a over the levels
// >= mlevel), :3354-3554 (levels below mlevel keep their previous sums, all-reduce over ranks,
// division by the total mass).  A frozen particle (beyond rtrunc, :3336) is skipped; with consp on (exp_amd_comp_set_consp)
// so is an escaped one, flagged here the first time it is found beyond rcom (:3317-3334).  The EJ orientation centre is
// orient.hip's.
#define COM_MAXLEV 16
// Component::escape_com (src/Component.cc:4204-4212): {com0[3], center[3], rcom^2}, the layout comp_frozen reads
struct EscapeArgs { double v[7]; };
__global__ void __launch_bounds__(256)
k_com_levels(const double *__restrict__ M, const double *__restrict__ X, const double *__restrict__ Y,
             const double *__restrict__ Z, const double *__restrict__ VX, const double *__restrict__ VY,
             const double *__restrict__ VZ, const double *__restrict__ AX, const double *__restrict__ AY,
             const double *__restrict__ AZ, double back, const uint8_t *__restrict__ lev, size_t n, int mlevel,
             int nlev, double *__restrict__ out /* [nlev][10] */, const double *__restrict__ frz,
             const uint32_t *__restrict__ id, uint8_t *__restrict__ escaped /* by id; nullptr: consp off */, EscapeArgs E)
{
  __shared__ double acc[COM_MAXLEV][10];
  for (int k = threadIdx.x; k < COM_MAXLEV * 10; k += 256) (&acc[0][0])[k] = 0.0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * 256;
  // per-lane partial sums of one level at a time, a wave reduction, then ONE LDS add per wave and
  // value (per-particle LDS atomics on ten words serialise: 5 ms at 1e8 particles)
  for (int L = mlevel; L < nlev; L++) {
    double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      if (nlev > 1 && lev[i] != L) continue;
      const double m = M[i], px = X[i], py = Y[i], pz = Z[i];
      if (escaped) {            // src/Component.cc:3317-3334 (every slot is examined by exactly one thread, once)
        const uint32_t pid = id[i];
        const uint8_t fl = escaped[pid];
        if (!fl && comp_frozen(E.v, px, py, pz)) { escaped[pid] = 1; continue; }
        if (fl == 1) continue;
      }
      if (comp_frozen(frz, px, py, pz)) continue;
      v[0] += m;
      v[1] = fma(m, px, v[1]);  v[2] = fma(m, py, v[2]);  v[3] = fma(m, pz, v[3]);
      const double ax = AX[i], ay = AY[i], az = AZ[i];
      double vx = VX[i], vy = VY[i], vz = VZ[i];
      if (back != 0.0) {        // prekicked store: the step-boundary velocity, formed on the fly (expamd_comp_velocity_view)
        vx = mul_then_add(vx, ax, back); vy = mul_then_add(vy, ay, back); vz = mul_then_add(vz, az, back);
      }
      v[4] = fma(m, vx, v[4]); v[5] = fma(m, vy, v[5]); v[6] = fma(m, vz, v[6]);
      v[7] = fma(m, ax, v[7]); v[8] = fma(m, ay, v[8]); v[9] = fma(m, az, v[9]);
    }
#pragma unroll
    for (int k = 0; k < 10; k++) {
      double t = v[k];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
      if ((threadIdx.x & 63) == 0 && t != 0.0) unsafeAtomicAdd(&acc[L][k], t);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nlev * 10; k += 256) {
    const double s = (&acc[0][0])[k];
    if (s != 0.0) unsafeAtomicAdd(out + k, s);
  }
}

// ---- the run log's sums (OutLog::Run, src/OutLog.cc:392-446) ------------------------------------------
// Per component: mass, m x, m v, angular momentum, kinetic energy, 0.5 m pot, the Clausius virial m x.a -- over every
// particle that is not frozen (beyond rtrunc, src/OutLog.cc:460), positions and velocities as stored (com_system off: Local = Inertial).
// Velocities are the step-boundary ones (expamd_comp_velocity_view), as at the reference's call after the second kick.
__global__ void __launch_bounds__(256)
k_log_sums(const double *__restrict__ M, const double *__restrict__ X, const double *__restrict__ Y,
           const double *__restrict__ Z, const double *__restrict__ VX, const double *__restrict__ VY,
           const double *__restrict__ VZ, const double *__restrict__ AX, const double *__restrict__ AY,
           const double *__restrict__ AZ, const double *__restrict__ P, double back, size_t n,
           double *__restrict__ out /* [13] */, const double *__restrict__ frz)
{
  __shared__ double acc[13];
  if (threadIdx.x < 13) acc[threadIdx.x] = 0.0;
  __syncthreads();
  double v[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const double m = M[i], x = X[i], y = Y[i], z = Z[i];
    if (comp_frozen(frz, x, y, z)) continue;
    const double ax = AX[i], ay = AY[i], az = AZ[i];
    double vx = VX[i], vy = VY[i], vz = VZ[i];
    if (back != 0.0) { vx = mul_then_add(vx, ax, back); vy = mul_then_add(vy, ay, back); vz = mul_then_add(vz, az, back); }
    v[0] += m;
    v[1] += m * x;  v[2] += m * y;  v[3] += m * z;
    v[4] += m * vx; v[5] += m * vy; v[6] += m * vz;
    v[7] += m * (y * vz - z * vy);
    v[8] += m * (z * vx - x * vz);
    v[9] += m * (x * vy - y * vx);
    v[10] += 0.5 * m * (vx * vx + vy * vy + vz * vz);
    v[11] += 0.5 * m * P[i];
    v[12] += m * (x * ax + y * ay + z * az);
  }
#pragma unroll
  for (int k = 0; k < 13; k++) {
    double t = v[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if ((threadIdx.x & 63) == 0 && t != 0.0) unsafeAtomicAdd(&acc[k], t);
  }
  __syncthreads();
  if (threadIdx.x < 13 && acc[threadIdx.x] != 0.0) unsafeAtomicAdd(out + threadIdx.x, acc[threadIdx.x]);
}

extern "C" int exp_amd_comp_log_sums(exp_amd_comp *c, double out[14])
{
  if (!c || !out) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double back = 0.0;
  { int rc_ = expamd_comp_velocity_view(c, &back); if (rc_) return rc_; }
  if (!c->com_red.p && c->com_red.alloc(COM_MAXLEV * 10) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "log_sums: hipMalloc failed");
  HIP_TRY(ctx, hipMemsetAsync(c->com_red.p, 0, 14 * sizeof(double), ctx->stream));
  if (c->n) {
    ProfScope ps(ctx, "k_log_sums");
    unsigned grid = cdiv(c->n, 256 * 16);
    if (grid > 2048) grid = 2048;
    k_log_sums<<<grid, 256, 0, ctx->stream>>>(c->a(A_M), c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX), c->a(A_VY),
                                             c->a(A_VZ), c->a(A_AX), c->a(A_AY), c->a(A_AZ), c->a(A_POT), back, c->n,
                                             c->com_red.p, expamd_comp_frz(c));
    HIP_TRY(ctx, hipGetLastError());
  }
  // the reference's MPI_Reduce of each sum (:448-478); the body count travels as the fourteenth double
  double nb = (double)c->n;
  HIP_TRY(ctx, hipMemcpyAsync(c->com_red.p + 13, &nb, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->nranks > 1 || ctx->ar_fn) {
    int rc = expamd_allreduce(ctx, c->com_red.p, 14);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(out, c->com_red.p, 14 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

// The component keys "noswitch", "freezeL", "dtreset" (src/Component.cc:253-255, :1036-1038), read by adjust_multistep_level's
// thread body (src/multistep.cc:136-158) and its device twin's caller (:528-534)
extern "C" int exp_amd_comp_set_level_policy(exp_amd_comp *c, int noswitch, int freeze_levels, int dtreset)
{
  if (!c) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (noswitch && !c->d_dtreq.p && c->n) {       // Particle::dtreq, one float per particle id (the first call resets it: firstCall)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (c->d_dtreq.alloc(c->n) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "comp_set_level_policy: hipMalloc failed");
    // (+inf, what a reset leaves: a run that never makes the first call -- the per-call entry point in the middle of a run --
    // then starts from "no step asked for yet" instead of a step of zero)
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)c->d_dtreq.p, 0x7f800000, c->n, ctx->stream));
  }
  c->noswitch = noswitch != 0;
  c->dtreset = dtreset != 0;
  c->freeze_levels = freeze_levels != 0;
  return EXP_AMD_OK;
}

// Component::consp / tidal / rcom (src/Component.cc:998-1000, :1024): the flags start at zero (iattrib as the body file gave
// them: exp_amd_comp_set_escaped), one byte per particle id
extern "C" int exp_amd_comp_set_consp(exp_amd_comp *c, int on, double rcom)
{
  if (!c) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (on && !(rcom > 0.0)) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comp_set_consp: rcom must be positive");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (on && !c->d_escaped.p && c->n) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (c->d_escaped.alloc(c->n) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "comp_set_consp: hipMalloc failed");
    HIP_TRY(ctx, hipMemsetAsync(c->d_escaped.p, 0, c->n, ctx->stream));
  }
  c->consp_on = on != 0;
  if (on) c->rcom = rcom;
  // (the cached per-level sums were formed under the other rule)
  if (c->com_lev.p) HIP_TRY(ctx, hipMemsetAsync(c->com_lev.p, 0, c->com_lev.bytes(), ctx->stream));
  return EXP_AMD_OK;
}

// iattrib[tidal] of every particle, in the caller's order (0 / 1): read back, or set (a restart: the body file's column)
extern "C" int exp_amd_comp_get_escaped(exp_amd_comp *c, unsigned char *flags)
{
  if (!c || !flags) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (!c->n && c->consp_on) return EXP_AMD_OK;
  if (!c->d_escaped.p) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "comp_get_escaped: consp was never switched on (exp_amd_comp_set_consp)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(flags, c->d_escaped.p, c->n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_set_escaped(exp_amd_comp *c, const unsigned char *flags)
{
  if (!c || !flags) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  if (!c->n && c->consp_on) return EXP_AMD_OK;
  if (!c->d_escaped.p) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "comp_set_escaped: consp was never switched on (exp_amd_comp_set_consp)");
  for (size_t i = 0; i < c->n; i++)
    if (flags[i] > 1) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comp_set_escaped: flag %zu is %d (0 or 1)", i, (int)flags[i]);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(c->d_escaped.p, flags, c->n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (c->com_lev.p) HIP_TRY(ctx, hipMemsetAsync(c->com_lev.p, 0, c->com_lev.bytes(), ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comp_fix_positions(exp_amd_comp *c, int mlevel, double out[10])
{
  if (!c || !out) return EXP_AMD_ERR_ARG;
  exp_amd_ctx *ctx = c->ctx;
  const int nlev = c->nlevels;
  if (nlev > COM_MAXLEV) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "fix_positions: more than %d levels", COM_MAXLEV);
  if (mlevel < 0) mlevel = 0;
  if (mlevel >= nlev) mlevel = nlev - 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double back = 0.0;
  { int rc_ = expamd_comp_velocity_view(c, &back); if (rc_) return rc_; }
  if (!c->com_lev.p) {
    if (c->com_lev.alloc(COM_MAXLEV * 10) != hipSuccess)
      return expamd_fail(ctx, EXP_AMD_ERR_HIP, "fix_positions: hipMalloc failed");
    HIP_TRY(ctx, hipMemsetAsync(c->com_lev.p, 0, c->com_lev.bytes(), ctx->stream));
    mlevel = 0;                           // nothing cached yet
  }
  // zero the level sums at and above mlevel (:3363-3369), then re-accumulate them
  HIP_TRY(ctx, hipMemsetAsync(c->com_lev.p + (size_t)mlevel * 10, 0,
                              (size_t)(nlev - mlevel) * 10 * sizeof(double), ctx->stream));
  EscapeArgs E;
  for (int k = 0; k < 3; k++) { E.v[k] = c->com0[k]; E.v[3 + k] = c->center[k]; }
  E.v[6] = c->rcom * c->rcom;
  if (c->n) {
    ProfScope ps(ctx, "k_com_levels");
    unsigned grid = cdiv(c->n, 256 * 16);
    if (grid > 2048) grid = 2048;
    k_com_levels<<<grid, 256, 0, ctx->stream>>>(c->a(A_M), c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_VX),
                                               c->a(A_VY), c->a(A_VZ), c->a(A_AX), c->a(A_AY),
                                               c->a(A_AZ), back, c->level[c->cur].p, c->n, mlevel, nlev,
                                               c->com_lev.p, expamd_comp_frz(c), c->id[c->cur].p,
                                               c->consp_on ? c->d_escaped.p : nullptr, E);
    HIP_TRY(ctx, hipGetLastError());
  }
  // sum the levels on the host side of one small read-back; ranks are combined first (:3500-3503)
  double host[COM_MAXLEV * 10];
  if (ctx->nranks > 1 || ctx->ar_fn) {
    // reduce a scratch copy so that the cached per-level sums stay rank-local
    if (!c->com_red.p && c->com_red.alloc(COM_MAXLEV * 10) != hipSuccess)
      return expamd_fail(ctx, EXP_AMD_ERR_HIP, "fix_positions: hipMalloc failed");
    HIP_TRY(ctx, hipMemcpyAsync(c->com_red.p, c->com_lev.p, (size_t)nlev * 10 * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    int rc = expamd_allreduce(ctx, c->com_red.p, (size_t)nlev * 10);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host, c->com_red.p, (size_t)nlev * 10 * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(host, c->com_lev.p, (size_t)nlev * 10 * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 10; k++) out[k] = 0.0;
  for (int L = 0; L < nlev; L++)
    for (int k = 0; k < 10; k++) out[k] += host[L * 10 + k];
  if (out[0] > 0.0)                       // :3541-3545
    for (int k = 1; k < 10; k++) out[k] /= out[0];
  return EXP_AMD_OK;
}
