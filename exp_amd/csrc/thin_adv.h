// The first half of a block-multistep sub-step folded into the thin accumulation kernels (k_sph_acc_thin,
// k_cyl_acc_thin): the tile's lane that loads a particle advances it first -- the operations of k_advance_levels
// (src/step.cc:126-148: incr_velocity(DT(M)/2, M); incr_position(DT(M), M)), a closing half-kick still owed ahead of them --
// and writes the new state back.  One launch less in every sub-step whose whole active range is thin.
#pragma once
#include <cstdint>
#include "common.h"

struct ThinAdv {
  double *x, *y, *z, *vx, *vy, *vz;
  const double *ax, *ay, *az;
  const uint8_t *lev;
  double dt_min;
  int multistep;
  double dt_kick0;             // a closing half-kick still owed by the levels >= kick0_lo (0: none)
  int kick0_lo;
  int on;                      // 0: the positions are used as they are
};

__device__ __forceinline__ void thin_advance(const ThinAdv &A, size_t i, double &x, double &y, double &z)
{
  const int L = A.lev[i];
  const double dtd = A.dt_min * (double)(1u << (A.multistep - L)), dtk = 0.5 * dtd;
  double u = A.vx[i], v = A.vy[i], w = A.vz[i];
  const double a0 = A.ax[i], a1 = A.ay[i], a2 = A.az[i];
  if (A.dt_kick0 != 0.0 && L >= A.kick0_lo) {
    u = mul_then_add(u, a0, A.dt_kick0);
    v = mul_then_add(v, a1, A.dt_kick0);
    w = mul_then_add(w, a2, A.dt_kick0);
  }
  u = mul_then_add(u, a0, dtk);
  v = mul_then_add(v, a1, dtk);
  w = mul_then_add(w, a2, dtk);
  A.vx[i] = u; A.vy[i] = v; A.vz[i] = w;
  x = mul_then_add(A.x[i], u, dtd);
  y = mul_then_add(A.y[i], v, dtd);
  z = mul_then_add(A.z[i], w, dtd);
  A.x[i] = x; A.y[i] = y; A.z[i] = z;
}
