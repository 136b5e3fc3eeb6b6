// Could the force pass write each particle straight to its place in the NEXT step's cell order and make the scatter pass
// (2.1 ms of the 10.1 ms step, 112 B per particle) unnecessary?  A particle's next cell is known inside the force pass (it
// computes the next step's sort key already), but its RANK inside that cell needs a cursor: one atomic with return per
// (block, destination cell) after an aggregation over the block, on ~2000 cells of which the ~6 the resident waves work on
// are hot.  An earlier experiment counted keys with one atomic per (wave, cell) and doubled the force pass' time
// (sph_kernels.h: sph_force_finish).  This microbenchmark has the traffic pattern of the aggregated scheme and nothing
// else: N particles in cell order (N / NCELL per cell), 256-thread blocks; a thread's destination cell is its own cell
// - 1, + 0 or + 1 (30 / 40 / 30 %); the block counts its (<= 3) destination cells in LDS, reserves with ONE atomicAdd per
// cell on one of S sub-cursors of the cell (S = 1, 4, 16: sub-cursor = blockIdx % S), and every thread writes 56 bytes
// (x, y, z, vx, vy, vz, id) at its reserved slot; beside it the same writes at the thread's OWN slot (no cursor).
//   hipcc --offload-arch=gfx950 -O2 -o tools/dbg/append_cursor tools/dbg/append_cursor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

template <int MODE, int WORK = 0>     // 0: own slot; 1: block-aggregated cursors; 2: wave-aggregated cursors; WORK: dependent fp64 FMAs per particle in front
// of the stores, in 8 independent chains (the force pass: ~1400 VALU instructions per wave), and the wider payload of the real scheme
// (reads 52 B: x, v, id; writes 108 B: state position, next position, v, a, pot, id)
__global__ void __launch_bounds__(256)
k_append(const double *__restrict__ src, double *__restrict__ dst, uint32_t *__restrict__ dstid, uint32_t *__restrict__ cursor,
         uint32_t n, uint32_t per_cell, uint32_t ncell, uint32_t S, uint32_t cap_sub /* capacity of a sub-region */)
{
  __shared__ uint32_t s_cnt[3], s_base[3];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const bool valid = i < n;
  const uint32_t cell = valid ? i / per_cell : 0;
  const uint32_t c0 = (blockIdx.x * 256) / per_cell;           // the block's first cell
  const uint32_t h = hash(i) % 10;
  int d = h < 3 ? -1 : h < 7 ? 0 : 1;
  uint32_t dc = (uint32_t)((int)cell + d);
  if (dc >= ncell) dc = cell;
  double v[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = valid ? src[(size_t)k * n + i] : 0.0;
  double w[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[0] + 1.0, v[1] + 1.0};
  if (WORK) {
#pragma unroll 1
    for (int it = 0; it < WORK / 8; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) w[q] = fma(w[q], 1.0000001, 1e-9);
    }
  }
  size_t slot = i;
  if (MODE == 1) {
    // rank within the block per destination cell (relative index dc - c0 + 1 in 0..2; a block that straddles a cell
    // boundary has up to 4: the last takes slot 2 -- only the traffic pattern matters here)
    uint32_t r = dc + 1 - c0; if (r > 2) r = 2;
    const uint32_t my = valid ? atomicAdd(&s_cnt[r], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < 3 && s_cnt[threadIdx.x]) {
      const uint32_t cc = c0 + threadIdx.x - 1 < ncell ? c0 + threadIdx.x - 1 : c0;
      s_base[threadIdx.x] = atomicAdd(&cursor[(size_t)cc * S + (blockIdx.x % S)], s_cnt[threadIdx.x]);
    }
    __syncthreads();
    const uint32_t cc = c0 + r - 1 < ncell ? c0 + r - 1 : c0;
    slot = ((size_t)cc * S + (blockIdx.x % S)) * cap_sub + (s_base[r] + my) % cap_sub;
  } else if (MODE == 2) {
    // one atomic per (wave, destination cell): the scheme of the earlier experiment
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const uint32_t c = __shfl(dc, lead);
      const unsigned long long m = __ballot(valid && dc == c);
      uint32_t base = 0;
      if ((threadIdx.x & 63) == lead) base = atomicAdd(&cursor[(size_t)c * S + (blockIdx.x % S)], (uint32_t)__popcll(m));
      base = __shfl(base, lead);
      if (valid && dc == c) slot = ((size_t)c * S + (blockIdx.x % S)) * cap_sub + (base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1))) % cap_sub;
      todo &= ~m;
    }
  }
  if (valid) {
    const size_t cap = (size_t)ncell * S * cap_sub;
#pragma unroll
    for (int k = 0; k < 6; k++) dst[(size_t)k * cap + slot] = v[k];
    if (WORK) {
#pragma unroll
      for (int k = 0; k < 7; k++) dst[(size_t)(6 + k) * cap + slot] = w[k];      // 13 doubles + id = 108 B
    }
    dstid[slot] = i;
  }
}

int main()
{
  const uint32_t N = 100000000u, NCELL = 1999u, PER = N / NCELL + 1;
  double *src, *dst; uint32_t *id, *cur;
  CK(hipMalloc(&src, (size_t)6 * N * 8)); CK(hipMemset(src, 0, (size_t)6 * N * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t Ss[] = {1, 4, 16};
  for (int mode = 0; mode < 3; mode++) {
    for (uint32_t S : Ss) {
      if (mode == 0 && S > 1) continue;
      const uint32_t cap_sub = (uint32_t)((double)PER / S * 1.15) + 512;
      const size_t cap = (size_t)NCELL * S * cap_sub;
      CK(hipMalloc(&dst, cap * 13 * 8)); CK(hipMalloc(&id, cap * 4)); CK(hipMalloc(&cur, (size_t)NCELL * S * 4));
      float best = 1e9;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipMemsetAsync(cur, 0, (size_t)NCELL * S * 4, s));
        CK(hipEventRecord(e0, s));
        const unsigned grid = (N + 255) / 256;
        if (mode == 0) k_append<0><<<grid, 256, 0, s>>>(src, dst, id, cur, N, PER, NCELL, S, cap_sub);
        else if (mode == 1) k_append<1><<<grid, 256, 0, s>>>(src, dst, id, cur, N, PER, NCELL, S, cap_sub);
        else k_append<2><<<grid, 256, 0, s>>>(src, dst, id, cur, N, PER, NCELL, S, cap_sub);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-46s S = %2u: %.3f ms  (%.2f TB/s of the 100 B per particle it moves)\n",
             mode == 0 ? "own slot (no cursor)" : mode == 1 ? "one atomic per (block, destination cell)" : "one atomic per (wave, destination cell)",
             S, best, 100.0 * N / (best * 1e-3) / 1e12);
      CK(hipFree(dst)); CK(hipFree(id)); CK(hipFree(cur));
    }
  }
  // the same with ~1400 dependent-chain FMAs per particle in front of the stores (the force pass is VALU-bound: do the wider
  // appended stores hide under it?) -- own slot against the block-aggregated append
  for (int mode = 0; mode < 2; mode++) {
    const uint32_t S = 1, cap_sub = (uint32_t)((double)PER * 1.15) + 512;
    const size_t cap = (size_t)NCELL * cap_sub;
    CK(hipMalloc(&dst, cap * 13 * 8)); CK(hipMalloc(&id, cap * 4)); CK(hipMalloc(&cur, (size_t)NCELL * 4));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemsetAsync(cur, 0, (size_t)NCELL * 4, s));
      CK(hipEventRecord(e0, s));
      const unsigned grid = (N + 255) / 256;
      if (mode == 0) k_append<0, 1400><<<grid, 256, 0, s>>>(src, dst, id, cur, N, PER, NCELL, S, cap_sub);
      else k_append<1, 1400><<<grid, 256, 0, s>>>(src, dst, id, cur, N, PER, NCELL, S, cap_sub);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("1400 FMAs per particle + 108 B stores, %-44s: %.3f ms\n", mode == 0 ? "own slot" : "one atomic per (block, destination cell)", best);
    CK(hipFree(dst)); CK(hipFree(id)); CK(hipFree(cur));
  }
  return 0;
}
