// Latency of the accumulate kernels' particle loads: every wave walks its own contiguous chunk of `chunk`
// particles in four 80 MB arrays, 64 x 8 B per array and iteration, with `spacing` dependent FMAs between
// issuing the loads of group k+1 and waiting for them (the one-deep software prefetch of the kernels).
//   hipcc -O2 --offload-arch=gfx950 tools/dbg/load_latency.hip -o build/load_latency
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 2) k(const double *X, const double *Y, const double *Z, const double *M, size_t n,
                                           int chunk, int spacing, int flush_every, double *sink, unsigned long long *out, double *W)
{
  __shared__ double pad[4400];        // 35 KB of LDS, as in k_cyl_accumulate
  const int lane = threadIdx.x & 63;
  const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t cbeg = wid * (size_t)chunk, cend = cbeg + chunk < n ? cbeg + chunk : n;
  if (cbeg >= n) return;
  pad[threadIdx.x] = 0;
  unsigned long long tot = 0;
  double nx = X[cbeg + lane], ny = Y[cbeg + lane], nz = Z[cbeg + lane], nm = M[cbeg + lane], acc = 0;
  int ng = 0;
  for (size_t i = cbeg + lane; i < cend; i += 64) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tot += __builtin_readcyclecounter() - t0;
    double a = nx + ny * nz + nm;
    if (i + 64 < cend) { nx = X[i + 64]; ny = Y[i + 64]; nz = Z[i + 64]; nm = M[i + 64]; }
#pragma unroll 8
    for (int s = 0; s < spacing; s++) a = a * 1.0000001 + 0.5;
    acc += a; ng++;
    if (flush_every && ng % flush_every == 0) {         // the flush of k_cyl_accumulate: 4 x 16 no-return fp64 atomics on a nearby node row
      const size_t base = ((i / 305) % 33000) * 13;
      for (int g = 0; g < 4; g++)
        if ((lane & 3) == 0) unsafeAtomicAdd(W + base + g * 16 + (lane >> 2), a);
    }
  }
  if (acc == 1.2345) sink[0] = acc + pad[5];
  if (lane == 0) { atomicAdd(&out[0], tot); atomicAdd(&out[1], (unsigned long long)ng); }
}
int main()
{
  const size_t n = 10000000;
  double *a[4], *sink; unsigned long long *o;
  for (auto &p : a) { hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8); }
  hipMalloc(&sink, 8); hipMalloc(&o, 16);
  double *W; hipMalloc(&W, 33200 * 13 * 8 + 4096); hipMemset(W, 0, 33200 * 13 * 8 + 4096);
  for (int chunk : {1024, 4096}) for (int spacing : {320, 1280}) for (int fe : {0, 11, 2}) {
    hipMemset(o, 0, 16);
    const int nw = (int)((n + chunk - 1) / chunk), nb = (nw + 3) / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 200; r++) k<<<nb, 256>>>(a[0], a[1], a[2], a[3], n, chunk, spacing, fe, sink, o, W);   // clocks up
    hipDeviceSynchronize();
    hipMemset(o, 0, 16);
    hipEventRecord(e0);
    for (int r = 0; r < 50; r++) k<<<nb, 256>>>(a[0], a[1], a[2], a[3], n, chunk, spacing, fe, sink, o, W);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("flush every %2d groups; chunk %5d particles per wave, %4d FMAs per group: mean wait for the prefetched loads %6.0f ticks per group; kernel %.3f ms = %.2f TB/s\n",
           fe, chunk, spacing, (double)h[0] / (double)h[1], ms, n * 32.0 / (ms * 1e9));
  }
  return 0;
}
