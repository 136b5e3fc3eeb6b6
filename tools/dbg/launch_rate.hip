// Launch-rate microbenchmark: what does a chain of tiny dependent kernels cost per link on this box --
// enqueued one by one (host rate), on two streams alternately, and replayed from a HIP graph?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_rate tools/dbg/launch_rate.hip && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  double *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  for (int blocks : {1, 64, 512}) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipStreamSynchronize(s1));
      double t0 = now();
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < N; i++) k_tiny<<<blocks, 256, 0, s1>>>(d, blocks * 256);
      CK(hipEventRecord(e1, s1));
      double t1 = now();
      CK(hipStreamSynchronize(s1));
      double t2 = now(); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("one stream, %4d blocks: host enqueue %.2f us/launch, host total %.2f us/launch, GPU span %.2f us/launch\n", blocks,
                      1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N, 1e3 * ms / N);
    }
  }
  // two streams alternately
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < N; i++) { k_tiny<<<1, 256, 0, s1>>>(d, 256); k_tiny<<<1, 256, 0, s2>>>(d + 4096, 256); }
    double t1 = now();
    CK(hipDeviceSynchronize());
    double t2 = now();
    if (rep) printf("two streams alternately: host enqueue %.2f us/launch, total %.2f us per launch (%.2f per pair)\n", 1e6 * (t1 - t0) / (2 * N),
                    1e6 * (t2 - t0) / (2 * N), 1e6 * (t2 - t0) / N);
  }
  // chains of 20 with a stream sync after each (the sub-step pattern: launch chain, read back, decide)
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int c = 0; c < 100; c++) { for (int i = 0; i < 20; i++) k_tiny<<<1, 256, 0, s1>>>(d, 256); CK(hipStreamSynchronize(s1)); }
    double t2 = now();
    if (rep) printf("chains of 20 + sync: %.2f us per chain, %.2f per launch\n", 1e6 * (t2 - t0) / 100, 1e6 * (t2 - t0) / 2000);
  }
  // graph of 20 dependent launches
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; i++) k_tiny<<<1, 256, 0, s1>>>(d, 256);
  CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int c = 0; c < 100; c++) { CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1)); }
    double t2 = now();
    if (rep) printf("graph of 20 + sync: %.2f us per replay, %.2f per node\n", 1e6 * (t2 - t0) / 100, 1e6 * (t2 - t0) / 2000);
  }
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int c = 0; c < 100; c++) CK(hipGraphLaunch(ge, s1));
    CK(hipStreamSynchronize(s1));
    double t2 = now();
    if (rep) printf("graph of 20 back to back: %.2f us per replay, %.2f per node\n", 1e6 * (t2 - t0) / 100, 1e6 * (t2 - t0) / 2000);
  }
  // event record + wait across streams (the cross-force dependency)
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < 1000; i++) {
      k_tiny<<<1, 256, 0, s1>>>(d, 256); CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
      k_tiny<<<1, 256, 0, s2>>>(d, 256); CK(hipEventRecord(e1, s2)); CK(hipStreamWaitEvent(s1, e1, 0));
    }
    double t1 = now();
    CK(hipDeviceSynchronize());
    double t2 = now();
    if (rep) printf("ping-pong over two streams with events: host %.2f us, total %.2f us per hop\n", 1e6 * (t1 - t0) / 2000, 1e6 * (t2 - t0) / 2000);
  }
  // a 64-byte device-to-pinned-host copy + sync (the read-back)
  void *h; CK(hipHostMalloc(&h, 4096));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < 500; i++) { k_tiny<<<1, 256, 0, s1>>>(d, 256); CK(hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); }
    double t2 = now();
    if (rep) printf("kernel + 256-byte read-back + sync: %.2f us\n", 1e6 * (t2 - t0) / 500);
  }
  return 0;
}
