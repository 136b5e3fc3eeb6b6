// Cylindrical BFE force method (cylinder / EmpCylSL): device-side description of the basis and the per-particle
// coordinate helpers every cylinder kernel shares (cyl_kernels.h, cyl_fields.hip).
#pragma once
#include <cstring>
#include "sort_kernels.h"
#include "thin_adv.h"
#include "force.h"

#include <type_traits>

#define DSMALL 1.0e-16        // src/expand.H:130
#define CYL_MAX_M 12            // azimuthal orders with unrolled kernels; above: k_cyl_moments_gen / k_cyl_force_gen
#define CYL_GEN_MAX_M 64        // ... up to this order

typedef const __attribute__((address_space(4))) double *cdp;

struct CylDev {
  int mmax, nmax, numx, numy, cmapr, cmapz, EVEN_M, ntrig;
  double ascale, hscale, rtable, xmin, dx, ymin, dy, rmax2;
  // reciprocals and products of the above, so that the per-particle code multiplies where the
  // reference divides by a constant (last-ulp differences only; the bilinear blend is continuous
  // across cell edges): 1/ascale, 1/hscale, 1/dx, 1/dy, rtable*ascale
  double inv_ascale, inv_hscale, inv_dx, inv_dy, rtab_abs, inv_rtab_abs;
  double cx, cy, cz;
  // Orient::transformBody of the component the basis belongs to (src/Cylinder.cc:799, :1352);
  // forces go back through its transpose, transformOrig (:1418)
  int use_rot;
  double rot[9];
  PseudoDev ps;     // frame acceleration of the TARGET component (force pass only)
  // deterministic mode: rounding-grid constants of the moment terms / of the in-cut mass (0: off)
  double detC, detCm;
  double umass;     // != 0: every particle of the component has this mass (accumulate does not read the stream)
  double mscale;    // Component::Adiabatic() of the basis' component at the time of the call: multiplies every mass the
                    // accumulation and the differencing read (src/Cylinder.cc:834, :1758; exp_amd_force_set_mass_scale)
  // Component::freeze (src/Component.cc:4194-4202) of the component whose particles the launch walks (the source of an
  // accumulation, the TARGET of a force pass: src/Cylinder.cc:788, :842, :1329, :1756): {com0[3], center[3], rtrunc^2} in
  // device memory (exp_amd_comp::d_frz; behind a pointer: see SphDev), nullptr: rtrunc not set
  const double *frz;
};

// Component::freeze, in the reference's operation order: r2 = sum_k (pos[k] - com0[k] - center[k])^2 > rtrunc^2
__device__ __forceinline__ bool cyl_frozen(const CylDev &C, double px, double py, double pz)
{
  if (!C.frz) return false;
  const double *F = C.frz;
  const double dx = (px - F[0]) - F[3], dy = (py - F[1]) - F[4], dz = (pz - F[2]) - F[5];
  double r2 = dx * dx;
  r2 = mul_then_add(r2, dy, dy);
  r2 = mul_then_add(r2, dz, dz);
  return r2 > F[6];
}

// centred, then rotated into the body frame
__device__ __forceinline__ void cyl_local(const CylDev &C, double x, double y, double z, double &xx,
                                          double &yy, double &zz)
{
  xx = x - C.cx; yy = y - C.cy; zz = z - C.cz;
  if (C.use_rot) {
    const double a = xx, b = yy, c = zz;
    xx = C.rot[0] * a + C.rot[1] * b + C.rot[2] * c;
    yy = C.rot[3] * a + C.rot[4] * b + C.rot[5] * c;
    zz = C.rot[6] * a + C.rot[7] * b + C.rot[8] * c;
  }
}

// ... for the passes that ADD particle contributions: a frozen particle (Component::freeze) is sent far off the grid and
// outside the rcylmax cut, where the pass neither counts nor adds it -- the `continue` of src/Cylinder.cc:842, the
// `return` of :1756
__device__ __forceinline__ void cyl_local_acc(const CylDev &C, double x, double y, double z, double &xx,
                                              double &yy, double &zz)
{
  cyl_local(C, x, y, z, xx, yy, zz);
  if (C.frz && cyl_frozen(C, x, y, z)) { xx = 1.0e150; yy = 0.0; zz = 0.0; }
}

template <int I, int N, class F>
__device__ __forceinline__ void cstatic_for(F &&f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cstatic_for<I + 1, N>(f);
  }
}

// 1 / r for r = sqrt(x^2 + y^2) + DSMALL (src/Cylinder.cc:1359) from irp ~ 1 / sqrt(x^2 + y^2): one Newton step squares
// the relative difference DSMALL / R between the two, which is below an ulp only from R ~ 1e-8 outwards; within 1e-9 of the
// axis the division itself (before: 1e-4 of the radial force's projection at R = 1e-14)
__device__ __forceinline__ double cyl_inv_r(double r, double irp, double r2)
{
  return r2 > 1.0e-18 ? rcp_refine(r, irp) : 1.0 / r;
}

// exputil/EmpCylSL.cc:6446-6463
__device__ __forceinline__ double cyl_r_to_xi(const CylDev &C, double r)
{
  if (C.cmapr > 0) { const double u = r * C.inv_ascale; return div_fast(u - 1.0, u + 1.0); }
  return r;
}

// exputil/EmpCylSL.cc:7109-7117
__device__ __forceinline__ double cyl_z_to_y(const CylDev &C, double z)
{
  if (C.cmapz == 1) return copysign(asinh_pos(fabs(z) * C.inv_hscale), z);     // sign(z) asinh|z/h|
  if (C.cmapz == 2) {
    double g, y;
    sqrt_rsqrt(z * z + C.hscale * C.hscale, g, y);
    return z * y;
  }
  return z;
}

// cell and bilinear weights (exputil/EmpCylSL.cc:5567-5597 == :5280-5314), enforce_limits false
__device__ __forceinline__ void cyl_weights(const CylDev &C, double r, double z, int &ix, int &iy,
                                            double &c00, double &c10, double &c01, double &c11)
{
  const double X = (cyl_r_to_xi(C, r) - C.xmin) * C.inv_dx;
  const double Y = (cyl_z_to_y(C, z) - C.ymin) * C.inv_dy;
  ix = (int)X;
  iy = (int)Y;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix >= C.numx) ix = C.numx - 1;
  if (iy >= C.numy) iy = C.numy - 1;
  const double delx0 = (double)ix + 1.0 - X, dely0 = (double)iy + 1.0 - Y;
  const double delx1 = X - (double)ix, dely1 = Y - (double)iy;
  c00 = delx0 * dely0;
  c10 = delx1 * dely0;
  c01 = delx0 * dely1;
  c11 = delx1 * dely1;
}

// sort key: level * (ncell+1) + cell, cell = ix*numy + iy; off-grid particles share bin ncell
struct CylKeyFn {
  static constexpr bool on = true;     // (k_kick_adjust: writes keys, kick_adjust.h)
  CylDev C;
  uint32_t sparse_mask;      // levels that are not cell-sorted: all their particles share bin 0
  __device__ __forceinline__ uint32_t operator()(double x, double y, double z, uint8_t lev) const
  {
    if ((sparse_mask >> lev) & 1u) return (uint32_t)lev * ((uint32_t)(C.numx * C.numy) + 1u);
    double xx, yy, zz;
    cyl_local(C, x, y, z, xx, yy, zz);
    const double r2 = xx * xx + yy * yy;
    double r, ir_, r3, ir3_;
    sqrt_rsqrt(r2, r, ir_);
    sqrt_rsqrt(r2 + zz * zz, r3, ir3_);
    const uint32_t ncell = (uint32_t)(C.numx * C.numy);
    uint32_t cell = ncell;
    if (!(r3 > C.rtab_abs)) {
      int ix, iy;
      double a, b, c, d;
      cyl_weights(C, r, zz, ix, iy, a, b, c, d);
      cell = (uint32_t)(ix * C.numy + iy);
    }
    return (uint32_t)lev * (ncell + 1u) + cell;
  }
};

