// One translation unit per LMAX (compiled with -DSPH_L=<k>): instantiates the accumulate and
// force kernels for that harmonic order so that the orders build in parallel.
#include "sph_kernels.h"

#ifndef SPH_L
#error "compile with -DSPH_L=<lmax>"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

template <int LMAX, int MLO, int MHI>
static void launch_acc_split(const SphAccArgs &a)
{
  const unsigned grid = cdiv(a.n, (size_t)ACC_WAVES * ACC_CHUNK);
  k_sph_accumulate<LMAX, MLO, MHI><<<grid, ACC_WAVES * 64, 0, a.stream>>>(
      a.S, a.X, a.Y, a.Z, a.M, a.lev_off, a.lo, a.hi, a.W, a.used);
}

// m-range splits keep the per-lane moment accumulators (2 per real row) in registers
void CAT(expamd_sph_acc_L, SPH_L)(const SphAccArgs &a)
{
  constexpr int LMAX = SPH_L;
  if constexpr (LMAX <= 4) {
    launch_acc_split<LMAX, 0, LMAX>(a);
  } else if constexpr (LMAX <= 7) {
    launch_acc_split<LMAX, 0, 1>(a);
    launch_acc_split<LMAX, 2, LMAX>(a);
  } else if constexpr (LMAX <= 10) {
    launch_acc_split<LMAX, 0, 1>(a);
    launch_acc_split<LMAX, 2, 3>(a);
    launch_acc_split<LMAX, 4, 6>(a);
    launch_acc_split<LMAX, 7, LMAX>(a);
  } else {
    launch_acc_split<LMAX, 0, 0>(a);
    launch_acc_split<LMAX, 1, 1>(a);
    launch_acc_split<LMAX, 2, 3>(a);
    launch_acc_split<LMAX, 4, 5>(a);
    launch_acc_split<LMAX, 6, 8>(a);
    launch_acc_split<LMAX, 9, LMAX>(a);
  }
}

void CAT(expamd_sph_force_L, SPH_L)(const SphForceArgs &a)
{
  constexpr int LMAX = SPH_L;
  const SphDev &S = a.S;
  // FLAGS=true always: besides honouring NO_L0/.../M0_only its wave-uniform row branches keep
  // the unrolled (l,m) nest in small basic blocks (the branch-free variant made hipcc spill).
  k_sph_force<LMAX, true><<<a.grid, 256, 0, a.stream>>>(S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi,
                                                        a.G, a.H, a.AX, a.AY, a.AZ, a.POT, a.VX,
                                                        a.VY, a.VZ, a.dt_kick, a.assign);
}
