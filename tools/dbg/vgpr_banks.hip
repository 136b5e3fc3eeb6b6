// Does the register file charge for where the three 64-bit sources of an fp64 FMA live?  Four independent chains
// v_fma_f64 d, a, b, d per wave with hand-picked registers, 4 waves per SIMD.
//   hipcc -O2 --offload-arch=gfx950 tools/dbg/vgpr_banks.hip -o build/vgpr_banks && build/vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(D0, D1, D2, D3, A, B) \
  "v_fma_f64 " D0 ", " A ", " B ", " D0 "\n v_fma_f64 " D1 ", " A ", " B ", " D1 "\n v_fma_f64 " D2 ", " A ", " B ", " D2 "\n v_fma_f64 " D3 ", " A ", " B ", " D3 "\n"
template <int V>
__global__ void __launch_bounds__(64) k(double *out, int iters)
{
  asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n"
               "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
  for (int i = 0; i < iters; i++) {
    if (V == 0)      // a = v[40:41], b = v[44:45] (same bank pair as each other and as every destination)
      asm volatile(BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "v[40:41]", "v[44:45]") BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "v[40:41]", "v[44:45]")
                   ::: "v48", "v49", "v52", "v53", "v56", "v57", "v60", "v61");
    if (V == 1)      // a = v[40:41], b = v[42:43] (different pairs), destinations in v[48..]: pair of a
      asm volatile(BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "v[40:41]", "v[42:43]") BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "v[40:41]", "v[42:43]")
                   ::: "v48", "v49", "v52", "v53", "v56", "v57", "v60", "v61");
    if (V == 2)      // destinations alternate between the two pairs
      asm volatile(BODY("v[48:49]", "v[50:51]", "v[56:57]", "v[58:59]", "v[40:41]", "v[42:43]") BODY("v[48:49]", "v[50:51]", "v[56:57]", "v[58:59]", "v[40:41]", "v[42:43]")
                   ::: "v48", "v49", "v50", "v51", "v56", "v57", "v58", "v59");
    if (V == 3)      // one source from a scalar register pair
      asm volatile(BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "s[20:21]", "v[42:43]") BODY("v[48:49]", "v[52:53]", "v[56:57]", "v[60:61]", "s[20:21]", "v[42:43]")
                   ::: "v48", "v49", "v52", "v53", "v56", "v57", "v60", "v61", "s20", "s21");
  }
  if (iters < 0) out[0] = 1;
}
template <int V> void run(const char *name)
{
  double *d; hipMalloc(&d, 8);
  const int iters = 100000, nw = 256 * 4 * 4;
  for (int r = 0; r < 3; r++) k<V><<<nw, 64>>>(d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<V><<<nw, 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-70s %.3f ns per FMA per SIMD\n", name, ms * 1e6 / ((double)iters * 8 * 4));
}
int main()
{
  run<0>("sources a, b and the destinations all in bank pair {0,1}");
  run<1>("a in {0,1}, b in {2,3}, destinations in {0,1}");
  run<2>("a in {0,1}, b in {2,3}, destinations alternating");
  run<3>("a scalar, b in {2,3}, destinations in {0,1}");
  return 0;
}
