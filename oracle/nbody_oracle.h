/* nbody_oracle.h -- CPU restatement of EXP's block-multistep step loop over SEVERAL components with
 * mutual interactions (BASELINE config 4: disk + halo, multistep > 0).  TEST INFRASTRUCTURE ONLY
 * (same scope statement as bfe_oracle.h: parity unpinned; never used by exp_amd/).
 *
 * Follows: do_step (src/step.cc:67-325), begin_run (src/begin.cc:80-129),
 * ComponentContainer::compute_expansion / compute_potential / multistep_reset
 * (src/ComponentContainer.cc:1173-1241, :580-917), adjust_multistep_level (src/multistep.cc:344-627)
 * with the thread body (:52-236), SphericalBasis's N/L swap, multistep_update / _finish and
 * compute_multistep_coefficients (src/SphericalBasis.cc:785-792, :1033-1079, :1156-1333), the
 * cylinder's twins (src/Cylinder.cc:946-1199, :1752-1795; src/CylEXP.cc:45-282;
 * exputil/EmpCylSL.cc:1867-2030 setup_accumulation).
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H

#include "bfe_oracle.h"
#include "cyl_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int kind;                   /* 0 = sphereSL (sg, sp), 1 = cylinder (cg)                        */
  const orc_slgrid *sg;
  const orc_sph_params *sp;
  const orc_cylgrid *cg;
  long n;
  double *x, *y, *z, *vx, *vy, *vz, *ax, *ay, *az, *pot;
  const double *mass;
  int *level;                 /* Particle::level                                                 */
  double center[3];           /* Component::center (Local | Centered)                            */
  long ncoef;                 /* sph: (L+1)^2 nmax real rows; cyl: cos[m][n] then sin[m][n]      */
  double *coefN, *coefL;      /* [(multistep+1)][ncoef]  expcoefN/L, cosN/sinN, cosL/sinL        */
  double *coef;               /* [ncoef] the combined set of the last force evaluation           */
  double cylmass;             /* Cylinder::cylmass                                               */
  long used;                  /* PotAccel::used                                                  */
  double resetT;              /* Cylinder::resetT                                                */
  /* ---- keys that default to "off" (all zero = the defaults) ------------------------------------ */
  int has_rtrunc;             /* Component::rtrunc set (src/Component.cc:1023); freeze(): :4194-4202 */
  double rtrunc, com0[3];
  int adiabatic;              /* ton / toff / twid given (src/Component.cc:1040-1055): Adiabatic() :4214-4220 */
  double ton, toff, twid;
  int not_self_consistent;    /* "self_consistent: false" (src/SphericalBasis.cc:114-117, src/Cylinder.cc:557) */
  int coef_calls;             /* !firstime_coef once > 0 (src/SphericalBasis.cc:1001, src/Cylinder.cc:1198) */
  int fix_l0, have_c0;        /* FIX_L0 (src/SphericalBasis.cc:1689-1694); C0 = the saved l = 0 row [nmax]    */
  double *C0;
  int mlim;                   /* the cylinder's "mlim" key when has_mlim (src/Cylinder.cc:225)                */
  int has_mlim;
  int freeze_lev;             /* "freezeL" (Component::FreezeLev, src/Component.cc:255, :1037): levels are assigned on the first
                                 call of adjust_multistep_level only (src/multistep.cc:158, :534)               */
  int noswitch;               /* "noswitch" (Component::NoSwitch, src/Component.cc:253, :1036): Particle::dtreq keeps the smallest
                                 time step asked for since its last reset and levels are only assigned at the end of a master
                                 step (mdrft == Mstep) or on the first call (src/multistep.cc:136-147)            */
  int no_dtreset;             /* "dtreset: false" (:254, :1038): dtreq is not reset at mstep == 0 (src/multistep.cc:137)      */
  float *dtreq;               /* [n] Particle::dtreq when noswitch (caller's storage)                                         */
  void *noise;                /* the sphere's NOISE mode: orc_noise_create's handle (NULL: off) and [ncoef] of scratch; every force  */
  double *noise_buf;          /* evaluation, self or external, draws a set first (src/SphericalBasis.cc:395, :2150-2210)              */
  double ssfrac;              /* the sphere's "ssfrac" key (src/SphericalBasis.cc:149-152; subset when 0 < ssfrac < 1) ...      */
  int ss_nthrds;              /* ... and the thread count its partition of the level list depends on (0 is read as 1)         */
} orc_nbody_comp;

typedef struct {
  int ncomp;
  orc_nbody_comp *comp;
  int ninter;
  const int *inter;           /* ninter pairs (source, target): source's force acts on target    */
  int multistep;
  double dtime;
  double dynfrac[5];          /* D, V, S, A, P (src/global.cc:76-80)                             */
  int shiftlevl;
  long this_step;
  double tnow;
  int initializing;           /* the global of src/begin.cc:80, :129 (set by orc_nbody_init itself)  */
  int no_eqmotion;            /* the global "eqmotion: false" (src/global.cc:54): incr_position / incr_velocity return at once
                                 (src/incpos.cc:75, src/incvel.cc:93) -- fields, levels and time go on, nothing moves     */
} orc_nbody;

/* begin_run's initial expansion, potential and level assignment (src/begin.cc:80-129) */
void orc_nbody_init(orc_nbody *S);
void orc_nbody_init_pass0(orc_nbody *S);   /* its first half (test hook) */
/* one do_step (src/step.cc:67-325); nswitch[ncomp] (may be NULL) receives the level changes of
 * each component summed over the sub-steps                                                     */
void orc_nbody_step(orc_nbody *S, long *nswitch);

/* CylEXP::multistep_update (src/CylEXP.cc:159-188) for one particle given in the cylinder's centred
 * frame: val[2*(mmax+1)*norder] (cos block, sin block) receives `hold`; returns 0 when the particle
 * is off the grid (nothing to add).  Called by Cylinder::multistep_update (src/Cylinder.cc:1752-1773),
 * which applies neither the rcylmax cut nor the body rotation.                                  */
int orc_cyl_multistep_update(const orc_cylgrid *g, double xx, double yy, double zz, double mass,
                             double *val, double *vc, double *vs);

#ifdef __cplusplus
}
#endif
#endif
