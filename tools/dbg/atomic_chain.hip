// How long does ONE wave take to retire a chain of fp64 atomics without return?  (k_sph_mstep_update with a
// handful of movers: 196 atomic instructions per wave, ~55 us.)
//   hipcc -O2 --offload-arch=gfx950 -munsafe-fp-atomics tools/dbg/atomic_chain.hip -o build/atomic_chain && build/atomic_chain
#include <hip/hip_runtime.h>
#include <cstdio>
// mode 0: lane 0 issues n atomics to consecutive doubles; 1: the same n values spread over the lanes (n/64 instructions);
// 2: lane 0, stride 16 doubles (one cache line each); 3: lanes 0..3 each issue n atomics to the SAME n doubles;
// 4: lanes 0..3, each to its own n doubles
__global__ void __launch_bounds__(64) k(double *w, int n, int mode, long long *t)
{
  const int lane = threadIdx.x;
  const long long t0 = wall_clock64();
  if (mode == 0) { if (lane == 0) for (int j = 0; j < n; j++) unsafeAtomicAdd(w + j, 1.0); }
  else if (mode == 1) { for (int j = lane; j < n; j += 64) unsafeAtomicAdd(w + j, 1.0); }
  else if (mode == 2) { if (lane == 0) for (int j = 0; j < n; j++) unsafeAtomicAdd(w + 16 * j, 1.0); }
  else if (mode == 3) { if (lane < 4) for (int j = 0; j < n; j++) unsafeAtomicAdd(w + j, 1.0); }
  else if (mode == 4) { if (lane < 4) for (int j = 0; j < n; j++) unsafeAtomicAdd(w + 4096 * lane + j, 1.0); }
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) t[0] = wall_clock64() - t0;
}
int main()
{
  double *w; long long *t;
  hipMalloc(&w, 8 * 65536); hipMalloc(&t, 8);
  hipMemset(w, 0, 8 * 65536);
  const char *name[] = {"1 lane, consecutive", "64 lanes x n/64 instr", "1 lane, a line each", "4 lanes, same addresses", "4 lanes, own addresses"};
  for (int n : {64, 196, 484})
    for (int mode = 0; mode < 5; mode++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      k<<<1, 64>>>(w, n, mode, t); hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<1, 64>>>(w, n, mode, t);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long ht; hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
      printf("n %3d  %-26s kernel %7.1f us   in-kernel until drained %7.1f us (100 MHz clock)\n", n, name[mode], ms * 1e3, ht / 100.0);
    }
  return 0;
}
