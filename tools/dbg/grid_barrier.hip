// What does a phase boundary cost INSIDE one persistent launch?  G resident workgroups of 256 threads run N phases; a phase
// is a dependent global read-modify-write of a small array (what a tiny kernel of the thin sub-steps does) followed by a grid
// barrier on a device counter (one atomic per workgroup, spin on its value).  Beside it: the same N phases as N dependent
// launches of G workgroups (tools/dbg/launch_gap.hip measured 1.55 us per link for 1-block kernels).
//   hipcc --offload-arch=gfx950 -O2 -o tools/dbg/grid_barrier tools/dbg/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned goal)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                    // this workgroup's writes before its arrival
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < goal) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_persist(double *p, int n, int phases, unsigned *counter, int work)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  for (int ph = 0; ph < phases; ph++) {
    if (work && i < n) {
      // read what ANOTHER workgroup wrote in the last phase, write own slot: a dependent round trip through L2
      const double v = __hip_atomic_load(p + ((i + 256) % n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + i, v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(counter, (unsigned)(ph + 1) * gridDim.x);
  }
}

__global__ void __launch_bounds__(256) k_phase(double *p, int n)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[(i + 256) % n] + 1.0;
}

__global__ void k_block(double *p, long long cycles) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) ; if (threadIdx.x == 0) p[0] += 1.0; }

int main()
{
  double *d; unsigned *cnt; CK(hipMalloc(&d, 1 << 24)); CK(hipMemset(d, 0, 1 << 24)); CK(hipMalloc(&cnt, 256));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200;
  const int Gs[] = {1, 16, 64, 128, 256, 512, 1024};
  for (int G : Gs) {
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 3; rep++) {
      for (int variant = 0; variant < 3; variant++) {
        CK(hipMemsetAsync(cnt, 0, 256, s));
        CK(hipDeviceSynchronize());
        k_block<<<1, 64, 0, s>>>(d + (1 << 20), 100000);            // 1 ms: the chain is enqueued behind it
        CK(hipEventRecord(e0, s));
        if (variant == 0) k_persist<<<G, 256, 0, s>>>(d, G * 256, N, cnt, 0);        // barriers alone
        else if (variant == 1) k_persist<<<G, 256, 0, s>>>(d, G * 256, N, cnt, 1);   // phase work + barrier
        else for (int i = 0; i < N; i++) k_phase<<<G, 256, 0, s>>>(d, G * 256);      // the same phases as launches
        CK(hipEventRecord(e1, s));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best[variant]) best[variant] = ms;
      }
    }
    printf("%5d workgroups: barrier alone %.2f us, phase + barrier %.2f us, phase as a dependent launch %.2f us (per phase, %d phases)\n",
           G, 1e3 * best[0] / N, 1e3 * best[1] / N, 1e3 * best[2] / N, N);
  }
  return 0;
}
