/* cyl_oracle.h -- CPU restatement of EXP's EmpCylSL/Cylinder hot path.  TEST INFRASTRUCTURE ONLY
 * (same scope statement as bfe_oracle.h: parity unpinned; never used by exp_amd/). */
#ifndef CYL_ORACLE_H
#define CYL_ORACLE_H
#include "bfe_oracle.h"      /* orc_slgrid: the helper spherical basis of the conditioning */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int mmax, norder, numx, numy, cmapr, cmapz, EVEN_M;
  double ascale, hscale;       /* ASCALE, HSCALE (acyl, hcyl)                               */
  double rtable;               /* Rtable = RMAX/sqrt(2)   (exputil/EmpCylSL.cc:2130)        */
  double xmin, dx, ymin, dy;   /* grid of setup_table (:2131-2137)                          */
  double rcylmax, acyl;        /* Cylinder's Rmax2 = (rcylmax*acyl)^2 cut (src/Cylinder.cc:752) */
  const double *tab;           /* [6][mmax+1][norder][numx+1][numy+1]                       */
} orc_cylgrid;

double orc_cyl_r_to_xi(const orc_cylgrid *g, double r);
double orc_cyl_z_to_y(const orc_cylgrid *g, double z);
void   orc_cyl_get_pot(const orc_cylgrid *g, double r, double z, double *Vc, double *Vs);
long   orc_cyl_accumulate(const orc_cylgrid *g, long n, const double *x, const double *y,
                          const double *z, const double *mass, const double *center,
                          double *cosN, double *sinN, double *cylmass);
void   orc_cyl_accumulated_eval(const orc_cylgrid *g, const double *accum_cos,
                                const double *accum_sin, double r, double z, double phi,
                                double *p0, double *p, double *fr, double *fz, double *fp);
void   orc_cyl_accel(const orc_cylgrid *g, long n, const double *x, const double *y,
                     const double *z, const double *center, const double *accum_cos,
                     const double *accum_sin, double cylmass, double *ax, double *ay, double *az,
                     double *pot);
/* field evaluation: EmpCylSL::accumulated_dens_eval (exputil/EmpCylSL.cc:5413-5502) and the pyEXP
 * wrappers Cylindrical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:1749-1849);
 * dens[2][mmax+1][norder][numx+1][numy+1] = densC, densS; out[n][9]                          */
double orc_cyl_accumulated_dens_eval(const orc_cylgrid *g, const double *dens,
                                     const double *accum_cos, const double *accum_sin, double r,
                                     double z, double phi, double *d0);
void   orc_pyexp_cyl_fields(const orc_cylgrid *g, const double *dens, const double *accum_cos,
                            const double *accum_sin, long n, const double *c1, const double *c2,
                            const double *c3, int coord, double *out);
/* pyEXP-literal twins.  Cylindrical::accumulate (expui/BiorthBasis.cc:1851-1857) = EmpCylSL::accumulate
 * on (R, z, phi): no rcylmax cut, only the table window; cosN/sinN +=, returns the number on the grid.
 * Cylindrical::computeAccel (:1804-1821) = accumulated_eval projected with the unguarded x/R, y/R:
 * no taper, no monopole continuation; acc[n][3], G = 1.                                           */
long   orc_pyexp_cyl_accumulate(const orc_cylgrid *g, long n, const double *x, const double *y,
                                const double *z, const double *mass, double *cosN, double *sinN);
void   orc_pyexp_cyl_accel(const orc_cylgrid *g, const double *accum_cos, const double *accum_sin,
                           long n, const double *x, const double *y, const double *z, double *acc);
/* Sub-sample covariance of the cylindrical coefficients: the `covar` branch of EmpCylSL::accumulate
 * (exputil/EmpCylSL.cc:4049-4146) as pyEXP drives it (Cylindrical::accumulate, expui/BiorthBasis.cc:
 * 1851-1857): particles on the grid only, whch = seq % sampT (seq NULL: the particle's index),
 * numbT/massT[sampT], VC[sampT][mmax+1][norder][2], MV[sampT][mmax+1][norder][norder][2] (re, im).
 * Accumulates into the outputs; returns the number of particles on the grid.                    */
long   orc_cyl_covariance(const orc_cylgrid *g, long n, const double *x, const double *y,
                          const double *z, const double *mass, const long *seq, int sampT,
                          long *numbT, double *massT, double *VC, double *MV);
/* conditioning the basis on the particles (precond: false): EmpCylSL::legendre_R (exputil/EmpCylSL.cc:6493-6569) and the
 * covariance sums of EmpCylSL::accumulate_eof (:2686-2862) under Cylinder's cut (src/Cylinder.cc:806-820) */
void   orc_emp_legendre_R(int lmax, double x, double *p);
long   orc_cyl_accumulate_eof(const orc_slgrid *sl, int M, double ascale, double rtable, double rmax2, long n,
                              const double *X, const double *Y, const double *Z, const double *mass,
                              double *SC, double *SS, double *cylmass);
#ifdef __cplusplus
}
#endif
#endif
