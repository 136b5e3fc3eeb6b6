// Issue cost of the fp64 transcendental / special instructions on one SIMD (4 waves per SIMD, 4 independent
// chains per wave): ns per wave64 instruction per SIMD, next to a plain FMA.
//   hipcc -O2 --offload-arch=gfx950 tools/dbg/dp_trans.hip -o build/dp_trans && build/dp_trans
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(64) k(double *out, int iters, double a, double b)
{
  double v[4];
  for (int j = 0; j < 4; j++) v[j] = 1.5 + 0.001 * (threadIdx.x + j);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (OP == 0) v[j] = fma(v[j], a, b);
        if (OP == 1) v[j] = __builtin_amdgcn_rcp(v[j]);
        if (OP == 2) v[j] = __builtin_amdgcn_rsq(v[j]);
        if (OP == 3) v[j] = __builtin_amdgcn_ldexp(v[j], 1) * 0.5;      // ldexp + mul
        if (OP == 4) v[j] = __builtin_amdgcn_frexp_mant(v[j]) + 1.0;    // frexp_mant + add
        if (OP == 5) v[j] = (double)(int)v[j] + 1.5;                    // cvt_i32_f64 + cvt_f64_i32 + add
        if (OP == 6) v[j] = fmax(v[j], a) * b;                          // max + mul
        if (OP == 7) v[j] = __builtin_amdgcn_trig_preop(v[j], 1) + 1.5; // (another quarter-rate candidate)
      }
  }
  double s = 0;
  for (int j = 0; j < 4; j++) s += v[j];
  if (s == 1.2345) out[0] = s;
}
template <int OP> void run(const char *name, int per)
{
  double *d;
  hipMalloc(&d, 8);
  const int iters = 20000, wps = 4, nwaves = 256 * 4 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<nwaves, 64>>>(d, 100, 1.0000001, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<nwaves, 64>>>(d, iters, 1.0000001, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double t = ms * 1e6 / ((double)iters * 8 * 4 * wps);     // ns per loop body statement per SIMD
  printf("%-28s %.2f ns per statement (%d instruction(s)) = %.1f cycles at 2.1 GHz\n", name, t, per, t * 2.1);
  hipFree(d);
}
int main()
{
  run<0>("v_fma_f64", 1);
  run<1>("v_rcp_f64", 1);
  run<2>("v_rsq_f64", 1);
  run<3>("v_ldexp_f64 + v_mul_f64", 2);
  run<4>("v_frexp_mant_f64 + v_add_f64", 2);
  run<5>("cvt_i32_f64 + cvt_f64_i32 + add", 3);
  run<6>("v_max_f64 + v_mul_f64", 2);
  run<7>("v_trig_preop_f64 + v_add_f64", 2);
  return 0;
}
