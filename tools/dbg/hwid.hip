// Where do the waves of a 256-thread block land?  Prints (XCC, SE, CU, SIMD, wave slot) of the four waves of a
// few blocks of a launch that, like k_sph_accumulate, fits two blocks per CU (LDS-limited).
//   hipcc -O2 --offload-arch=gfx950 tools/dbg/hwid.hip -o build/hwid && build/hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned *out, int spin)
{
  __shared__ double big[9000];      // 72 KB: two blocks per CU
  const unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20);
  big[threadIdx.x] = hw;
  double a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0000001 + 0.5;      // stay resident for a while
  if (a == 12345.0) big[1] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
}
int main()
{
  const int nb = 2048;
  unsigned *d;
  hipMalloc(&d, nb * 8 * sizeof(unsigned));
  k<<<nb, 256>>>(d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  int distinct = 0, sameslot = 0, slotpar[2] = {0, 0};
  for (int b = 0; b < nb; b++) {
    int simds = 0, slot0 = h[b * 8] & 15, same = 1;
    for (int w = 0; w < 4; w++) {
      const unsigned hw = h[(b * 4 + w) * 2];
      simds |= 1 << ((hw >> 4) & 3);
      if ((int)(hw & 15) != slot0) same = 0;
      if ((hw >> 4 & 3) != (unsigned)w) {}
    }
    distinct += simds == 0xF;
    sameslot += same;
    slotpar[slot0 & 1]++;
    if (b < 6 || (b > 600 && b < 606))
      for (int w = 0; w < 4; w++) {
        const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
        printf("block %4d wave %d: hw %08x  slot %2u simd %u pipe %u cu %2u sh %u se %u | xcc reg %08x\n", b, w, hw, hw & 15,
               (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x);
      }
  }
  printf("blocks with 4 distinct SIMDs: %d / %d; all four waves in the same slot: %d; first-wave slot parity even/odd: %d / %d\n",
         distinct, nb, sameslot, slotpar[0], slotpar[1]);
  return 0;
}
