// How long until a wave's fp64 global atomics (no return) are all back?  Every wave issues 4 atomic
// instructions with 16 active lanes each (the flush pattern of k_cyl_accumulate) into a window of `span`
// doubles shared by all waves, then waits vmcnt(0); mean and max wait over the repetitions.
//   hipcc -O2 --offload-arch=gfx950 -munsafe-fp-atomics tools/dbg/atomic_latency.hip -o build/atomic_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(double *w, size_t span, int reps, int spacing, unsigned long long *out)
{
  const int lane = threadIdx.x & 63;
  const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned long long tot = 0, mx = 0;
  unsigned s = (unsigned)wid * 2654435761u + 12345u;
  for (int r = 0; r < reps; r++) {
    s = s * 1664525u + 1013904223u;
    const size_t base = ((size_t)s % (span / 64)) * 64;
    for (int g = 0; g < 4; g++)
      if ((lane & 3) == 0) unsafeAtomicAdd(w + base + g * 16 + (lane >> 2), 1.0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long d = __builtin_readcyclecounter() - t0;
    tot += d; mx = d > mx ? d : mx;
    double a = lane;
    for (int i = 0; i < spacing; i++) a = a * 1.0000001 + 0.5;       // work between flushes
    if (a == 1.5) w[0] = a;
  }
  if (lane == 0) { atomicAdd(&out[0], tot); atomicMax(&out[1], mx); }
}
int main()
{
  const int nblk = 512, reps = 200;
  for (size_t span : {(size_t)1 << 10, (size_t)1 << 14, (size_t)1 << 18, (size_t)1 << 22}) {
    for (int spacing : {0, 2000}) {
      double *w; unsigned long long *o;
      hipMalloc(&w, span * 8); hipMemset(w, 0, span * 8);
      hipMalloc(&o, 16); hipMemset(o, 0, 16);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      k<<<nblk, 256>>>(w, span, reps, spacing, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
      printf("window %8zu doubles, %4d FMAs between flushes: mean wait %7.0f ticks, max %8llu; kernel %.3f ms = %.2f G atomics/s\n", span, spacing,
             (double)h[0] / ((double)nblk * 4 * reps), h[1], ms, (double)nblk * 4 * reps * 64 / (ms * 1e6));
      hipFree(w); hipFree(o);
    }
  }
  return 0;
}
