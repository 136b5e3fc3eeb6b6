// Device-side gap between dependent launches, host out of the picture: a 3 ms blocker kernel goes first, the chain is
// enqueued behind it while it runs, events around the chain give its length on the GPU.  Variants: what the links are.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dbg/launch_gap tools/dbg/launch_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { double v[240]; };
__global__ void k_block(double *p, long long cycles) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) ; if (threadIdx.x == 0) p[0] += 1.0; }
__global__ void k_tiny(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
__global__ void k_bigarg(double *p, int n, Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += b.v[i & 127]; }
template <int K> __global__ void k_var(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += (double)K; }
__global__ void k_write(double *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (double)i; }
__global__ void k_read(const double *p, size_t n, double *o) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] < -1.0) o[0] = 1.0; }
__global__ void k_lds(double *p, int n) { __shared__ double s[8000]; int i = blockIdx.x * blockDim.x + threadIdx.x; s[threadIdx.x] = p[i & 255]; __syncthreads(); if (i < n) p[i] += s[(threadIdx.x + 1) & 255]; }
__global__ void k_atomic(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) unsafeAtomicAdd(p + (i & 1023), 1.0); }
typedef void (*vk)(double *, int);
int main()
{
  double *d, *big; CK(hipMalloc(&d, 1 << 22)); CK(hipMemset(d, 0, 1 << 22));
  const size_t NB = 16u << 20; CK(hipMalloc(&big, NB * 8));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  const int N = 200; Big b; for (int i = 0; i < 240; i++) b.v[i] = i;
  vk vars[8] = {k_var<0>, k_var<1>, k_var<2>, k_var<3>, k_var<4>, k_var<5>, k_var<6>, k_var<7>};
  auto run = [&](const char *name, int variant) -> int {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      k_block<<<1, 64, 0, s1>>>(d, 300000);           // 100 MHz wall clock: 3 ms
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < N; i++) switch (variant) {
        case 0: k_tiny<<<1, 256, 0, s1>>>(d, 256); break;
        case 1: k_tiny<<<2048, 256, 0, s1>>>(d, 2048 * 256); break;
        case 2: k_bigarg<<<1, 256, 0, s1>>>(d, 256, b); break;
        case 3: vars[i & 7]<<<1, 256, 0, s1>>>(d, 256); break;
        case 4: if (i & 1) k_read<<<64, 256, 0, s1>>>(big, 64 * 256, d); else k_write<<<(unsigned)(NB / 16 / 256), 256, 0, s1>>>(big, NB / 16); break;   // 8 MB written, then a dependent small read
        case 5: k_lds<<<1, 256, 0, s1>>>(d, 256); break;
        case 6: k_atomic<<<64, 256, 0, s1>>>(d, 64 * 256); break;
        case 7: if (i & 1) k_tiny<<<1, 256, 0, s1>>>(d, 256); else CK(hipMemsetAsync(d + 4096, 0, 4096, s1)); break;
        case 8: k_tiny<<<1, 256, 0, s1>>>(d, 256); CK(hipEventRecord(e2, s1)); break;      // an event record behind every launch
      }
      CK(hipEventRecord(e1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-64s %.2f us per link\n", name, 1e3 * best / N);
    return 0;
  };
  run("1 block", 0); run("2048 blocks", 1); run("1 block, 1.9 KB of arguments", 2); run("8 different kernels in turn", 3);
  run("8 MB write, then dependent 64-block read (per PAIR/2)", 4); run("1 block with 64 KB LDS", 5); run("64 blocks of atomics", 6);
  run("kernel, 4 KB memset in turn", 7); run("kernel + event record", 8);
  // two chains at once on two streams, each behind its own blocker
  {
    float best = 1e9, bestb = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      k_block<<<1, 64, 0, s1>>>(d, 300000); k_block<<<1, 64, 0, s2>>>(d + 8192, 300000);
      CK(hipEventRecord(e0, s1)); CK(hipEventRecord(e2, s2));
      for (int i = 0; i < N; i++) { k_tiny<<<1, 256, 0, s1>>>(d, 256); k_tiny<<<1, 256, 0, s2>>>(d + 16384, 256); }
      CK(hipEventRecord(e1, s1)); CK(hipEventRecord(e3, s2));
      CK(hipDeviceSynchronize());
      float ms, msb; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msb, e2, e3));
      if (ms < best) best = ms; if (msb < bestb) bestb = msb;
    }
    printf("%-64s %.2f / %.2f us per link\n", "two chains of 1-block kernels on two streams at once", 1e3 * best / N, 1e3 * bestb / N);
  }
  // cross-stream hops: s1 kernel -> event -> s2 kernel -> event -> s1 ...
  {
    float best = 1e9;
    std::vector<hipEvent_t> ev(2 * N); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      k_block<<<1, 64, 0, s1>>>(d, 600000);
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < N; i++) {
        k_tiny<<<1, 256, 0, s1>>>(d, 256); CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
        k_tiny<<<1, 256, 0, s2>>>(d, 256); CK(hipEventRecord(ev[2 * i + 1], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));
      }
      CK(hipEventRecord(e1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-64s %.2f us per hop\n", "hops between two streams through events", 1e3 * best / (2 * N));
  }
  // what a small device-to-host copy costs the stream: kernel, 256-byte copy into page-locked memory, next kernel
  {
    void *h; CK(hipHostMalloc(&h, 4096));
    auto chain = [&](const char *name, int variant) -> int {
      double *hd = nullptr;
      CK(hipHostGetDevicePointer((void **)&hd, h, 0));
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        k_block<<<1, 64, 0, s1>>>(d, 300000);
        CK(hipEventRecord(e0, s1));
        for (int i = 0; i < N; i++) {
          k_tiny<<<1, 256, 0, s1>>>(d, 256);
          if (variant == 0) CK(hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s1));
          else if (variant == 1) CK(hipMemcpyAsync(d + 65536, d, 256, hipMemcpyDeviceToDevice, s1));
          else k_tiny<<<1, 32, 0, s1>>>(hd, 32);
        }
        CK(hipEventRecord(e1, s1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-64s %.2f us per pair\n", name, 1e3 * best / N);
      return 0;
    };
    chain("kernel + 256-byte copy to page-locked host memory", 0);
    chain("kernel + 256-byte device-to-device copy", 1);
    chain("kernel + a kernel writing 256 bytes of page-locked host memory", 2);
  }
  return 0;
}
