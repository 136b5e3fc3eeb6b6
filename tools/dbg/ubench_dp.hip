// Micro-benchmark: fp64 VALU issue rate on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters)
{
  double x[12];
  for (int i = 0; i < 12; i++) x[i] = threadIdx.x * 1e-3 + i;
  const double sa = a, sb = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int i = 0; i < 12; i++) {
        if (MODE == 0) x[i] = fma(x[i], sa, sb);                 // 2 SGPR operands -> needs a mov? (a,b uniform)
        if (MODE == 1) x[i] = fma(x[(i + 1) % 12], sa, x[i]);    // 1 SGPR operand (fmac form)
        if (MODE == 2) x[i] = x[i] * sa;                         // mul
        if (MODE == 3) x[i] = x[i] + x[(i + 1) % 12];            // add
        if (MODE == 4) { float f = (float)x[i]; f = fmaf(f, 1.0001f, 0.5f); x[i] = f; } // cvt+f32
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 12; i++) s += x[i];
  if (s == 12345.678) out[threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wgs_per_cu)
{
  double *out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * wgs_per_cu;
  k<MODE><<<grid, 256>>>(out, 1.0000001, 1e-9, 10);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * 96 * wgs_per_cu;      // each WG = 4 waves, one per SIMD
  printf("%-28s wg/cu=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)\n", name, wgs_per_cu, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
  for (int w : {1, 2, 5, 8}) {
    if (w == 1) { run<0>("fma(v,s,s)", 1); run<1>("fmac(v,s,v)", 1); run<2>("mul(v,s)", 1); run<3>("add(v,v)", 1); }
    if (w == 2) { run<0>("fma(v,s,s)", 2); run<1>("fmac(v,s,v)", 2); run<2>("mul(v,s)", 2); run<3>("add(v,v)", 2); }
    if (w == 5) { run<0>("fma(v,s,s)", 5); run<1>("fmac(v,s,v)", 5); run<2>("mul(v,s)", 5); run<3>("add(v,v)", 5); }
    if (w == 8) { run<0>("fma(v,s,s)", 8); run<1>("fmac(v,s,v)", 8); run<2>("mul(v,s)", 8); run<3>("add(v,v)", 8); }
  }
  return 0;
}
