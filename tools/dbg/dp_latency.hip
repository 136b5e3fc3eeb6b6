// fp64 FMA: dependent-issue latency vs throughput on one SIMD.  ILP independent chains per wave, WPS waves per SIMD.
//   hipcc -O2 --offload-arch=gfx950 tools/dbg/dp_latency.hip -o build/dp_latency && build/dp_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void __launch_bounds__(64) k(double *out, int iters, double a, double b)
{
  double v[ILP];
  for (int j = 0; j < ILP; j++) v[j] = threadIdx.x + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int j = 0; j < ILP; j++) v[j] = fma(v[j], a, b);
  }
  double s = 0;
  for (int j = 0; j < ILP; j++) s += v[j];
  if (s == 1.2345) out[0] = s;
}
template <int ILP> void run(int wps)
{
  double *d;
  hipMalloc(&d, 8);
  const int iters = 20000, nwaves = 256 * 4 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<ILP><<<nwaves, 64>>>(d, 100, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<ILP><<<nwaves, 64>>>(d, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * 16 * ILP * wps);     // ns per FMA instruction per SIMD
  printf("ILP %d, %d wave(s)/SIMD: %.3f ns per wave64 FMA per SIMD (%.2f cycles at 2.1 GHz)\n", ILP, wps, per, per * 2.1);
  hipFree(d);
}
int main()
{
  for (int wps : {1, 2, 4}) { run<1>(wps); run<2>(wps); run<4>(wps); }
  return 0;
}
