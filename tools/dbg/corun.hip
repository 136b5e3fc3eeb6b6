// Experiment: a persistent, small-footprint streaming copy that runs beside the VALU-bound force pass.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/dbg/corun.hip -o build/libcorun.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));

// each block strides over the buffer; UNROLL x 16 B per lane in flight
template <int UNROLL>
__global__ void __launch_bounds__(256) k_copy_persistent(const d2 *__restrict__ src, d2 *__restrict__ dst, size_t n2)
{
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n2; base += stride) {
    d2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      if (base + (size_t)u * 256 < n2) v[u] = __builtin_nontemporal_load(src + base + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      if (base + (size_t)u * 256 < n2) __builtin_nontemporal_store(v[u], dst + base + (size_t)u * 256);
  }
}

extern "C" int corun_copy(const void *src, void *dst, size_t bytes, int blocks, int unroll, void *stream)
{
  const size_t n2 = bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  if (unroll == 4) k_copy_persistent<4><<<blocks, 256, 0, st>>>((const d2 *)src, (d2 *)dst, n2);
  else if (unroll == 8) k_copy_persistent<8><<<blocks, 256, 0, st>>>((const d2 *)src, (d2 *)dst, n2);
  else k_copy_persistent<16><<<blocks, 256, 0, st>>>((const d2 *)src, (d2 *)dst, n2);
  return (int)hipGetLastError();
}
