/*
 * bfe_oracle.h -- CPU restatement of EXP's BFE hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for exp_amd.  It restates, in plain scalar fp64 C, the
 * arithmetic of the reference's CPU thread bodies (file:line citations are relative
 * to the EXP source tree and are given on every function in bfe_oracle.c).
 *
 * PARITY UNPINNED for coefficients and accelerations: the reference ships no golden vectors /
 * known-answer values for them and its C++ cannot be compiled in this
 * image (needs Eigen, yaml-cpp, HighFive, FFTW).  The oracle is therefore pinned by (i) line-by-
 * line correspondence with the cited code, (ii) the analytic known-answer tests under tests/
 * (orthogonality, shell theorem, rotation/reflection symmetry, multipole continuity, leapfrog
 * reversibility, Newton's theorem at 1e7 particles), (iii) the reference's own N-body acceptance
 * criterion (tests/Halo: mean 2T/VC) run on the device path, and (iv) for the TABLES only, the
 * reference's own Fortran SL solver exputil/sledge.f, which does build here (flang) into
 * oracle/_ref/ and agrees with exp_amd/slgrid.py to its tolerance (tests/test_ref_sledge.py).
 *
 * Nothing under exp_amd/ may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 */
#ifndef BFE_ORACLE_H
#define BFE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- spherical SL grid (SLGridSph tables; include/SLGridMP2.H:28, sltableMP2.H:16-24) ---- */
typedef struct {
  int lmax, nmax, numr, cmap;
  double rmin, rmax, rmap;   /* grid limits and mapping scale                 */
  double xmin, xmax, dxi;    /* SLGridMP2.cc:1355-1382                        */
  const double *xi;          /* [numr]                                        */
  const double *p0;          /* [numr]  background potential                  */
  const double *d0;          /* [numr]  4*pi*background density               */
  const double *ev;          /* [(lmax+1)][nmax]                              */
  const double *ef;          /* [(lmax+1)][nmax][numr]  (ef(n,i) of table l)  */
} orc_slgrid;

/* flags of SphericalBasis (src/SphericalBasis.cc:28-52) */
typedef struct {
  double scale;              /* "scale" key                                   */
  double rmin, rmax;         /* expansion window (unscaled r)                 */
  int NO_L0, NO_L1, EVEN_L, EVEN_M, M0_only;
  int N1, N2;                /* pyEXP only: radial window of the l >= 1 sums in sph_eval / computeAccel
                                (expui/BiorthBasis.cc:761, :876); N2 < 0 means no upper limit    */
} orc_sph_params;

/* ---- options of the n-body thread bodies that default to "off" (the reference's Component / Basis keys rtrunc, ton /
 * toff / twid, mlim).  They are per-CALL state of the calling thread: set before a call of an accumulate / accel /
 * multistep_update function of this library, cleared (NULL) after it.
 *   adb      Component::Adiabatic() of the basis' component (src/Component.cc:4214-4220): `mass = Mass(indx) * adb`
 *            (src/SphericalBasis.cc:441, :471, :1161; src/Cylinder.cc:834, :1758)
 *   frz      Component::freeze(indx) (src/Component.cc:4194-4202) of the component whose particles the call walks:
 *            sum_k (pos[k] - com0[k] - center[k])^2 > rtrunc^2  ->  `continue` (src/SphericalBasis.cc:468, :1521;
 *            src/Cylinder.cc:842, :1329) / `return` (src/SphericalBasis.cc:1159, src/Cylinder.cc:1756)
 *   mlim     EmpCylSL::MLIM (set_mlim, include/EmpCylSL.H:567): get_pot, accumulated_eval and accumulated_dens_eval
 *            loop to min(MLIM, MMAX) (exputil/EmpCylSL.cc:5602, :5317, :5465); < 0: no limit                      */
typedef struct {
  double adb;
  int    frz;
  double rtrunc, com0[3], fcenter[3];
  int    mlim;
  /* "ssfrac" of SphericalBasis (src/SphericalBasis.cc:149-152): subset when 0 < ssfrac < 1 -- thread id of nthrds walks
   * [n id / nthrds, floor(ssfrac * (n (id + 1) / nthrds))) of the level list (:438-439, :460) and every mass is divided by
   * ssfrac (:473).  The level list is the order of the arrays handed in.  nthrds < 1 is read as 1.                     */
  double ssfrac;
  int    nthrds;
} orc_call_opts;
void   orc_set_call_opts(const orc_call_opts *o);      /* NULL: the defaults (adb 1, no freeze, no mlim, no subset) */
int    orc_opt_subset(double *ssfrac, int *nthrds);    /* 1 when the subset is on */

/* The NOISE mode (noise_oracle.cc, C++: the reference's std::mt19937 / std::normal_distribution): compute_rms_coefs
 * (src/SphericalBasis.cc:2108-2147) and update_noise (:2150-2210)                                                    */
void   orc_sph_compute_rms_coefs(const orc_slgrid *g, double scale, int num, const double *rtab, const double *dtab,
                                 int numg, const double *knot, const double *weight, double *meanC, double *rmsC);
void  *orc_noise_create(int lmax, int nmax, const double *meanC, const double *rmsC, double noiseN, unsigned seedN);
void   orc_noise_update(void *h, double *expcoef);     /* [(lmax+1)^2][nmax] */
void   orc_noise_destroy(void *h);
double orc_opt_adb(void);
int    orc_opt_mlim(int mmax);                         /* min(MLIM, MMAX) */
int    orc_opt_frozen(double x, double y, double z);   /* Component::freeze of the position (component coordinates) */
/* Component::Adiabatic() (src/Component.cc:4214-4220) */
double orc_adiabatic(double tnow, double ton, double toff, double twid);

void   orc_legendre_R (int lmax, double x, double *p);               /* p[(lmax+1)*(lmax+1)], p[l*(lmax+1)+m] */
void   orc_dlegendre_R(int lmax, double x, double *p, double *dp);
void   orc_sinecosine_R(int mmax, double phi, double *c, double *s);
void   orc_factorial_table(int lmax, double *f);                     /* f[l*(lmax+1)+m] */

double orc_sl_r_to_xi  (const orc_slgrid *g, double r);
double orc_sl_xi_to_r  (const orc_slgrid *g, double xi);
double orc_sl_d_xi_to_r(const orc_slgrid *g, double xi);
void   orc_sl_get_pot  (const orc_slgrid *g, double r, double *mat); /* mat[(lmax+1)*nmax] */
void   orc_sl_get_force(const orc_slgrid *g, double r, double *mat);
void   orc_sl_get_dens (const orc_slgrid *g, double r, double *mat);
/* orthoCheck: ret[(lmax+1)*nmax*nmax]; knots/weights = Gauss-Legendre on [0,1] */
void   orc_sl_orthocheck(const orc_slgrid *g, int num, const double *knots,
                         const double *weights, double *ret);

/* coefficient accumulation: coef[(lmax+1)^2 * nmax] in the reference's real-row order.
 * returns number of particles used.  kahan!=0 -> compensated summation (arbiter mode). */
long   orc_sph_accumulate(const orc_slgrid *g, const orc_sph_params *P, long n,
                          const double *x, const double *y, const double *z,
                          const double *mass, const double *center,
                          double *coef, int kahan);

/* acceleration + potential: acc{x,y,z}[i] += ..., pot[i] += ...  */
void   orc_sph_accel(const orc_slgrid *g, const orc_sph_params *P, long n,
                     const double *x, const double *y, const double *z,
                     const double *center, const double *coef,
                     double *ax, double *ay, double *az, double *pot);

/* ... with the target component's frame pseudo-acceleration pseudo[n][3] (Component::getPseudoAccel)
 * subtracted by every Component::AddAcc call the thread body makes (src/Component.H:914-921,
 * src/SphericalBasis.cc:1645-1651): once for z, TWICE for x and y when x^2 + y^2 > DSMALL. */
void   orc_sph_accel_pseudo(const orc_slgrid *g, const orc_sph_params *P, long n,
                            const double *x, const double *y, const double *z,
                            const double *center, const double *coef, const double *pseudo,
                            double *ax, double *ay, double *az, double *pot);

/* leapfrog pieces (src/incpos.cc:15-69, src/incvel.cc:15-88) */
void   orc_drift(long n, double dt, double *x, double *y, double *z,
                 const double *vx, const double *vy, const double *vz);
void   orc_kick (long n, double dt, double *vx, double *vy, double *vz,
                 const double *ax, const double *ay, const double *az);

/* one multistep=0 KDK step of a single self-gravitating spherical component
 * (src/step.cc:271-323): kick/2, drift, coefficients, zero+force, kick/2.   */
void   orc_sph_step(const orc_slgrid *g, const orc_sph_params *P, long n, double dt,
                    double *x, double *y, double *z, double *vx, double *vy, double *vz,
                    double *ax, double *ay, double *az, double *pot,
                    const double *mass, const double *center, double *coef);

/* ---- multistep bookkeeping (src/multistep.cc:630-680) ---- */
typedef struct {
  int multistep, Mstep;
  int *mintvl;    /* [multistep+1]            */
  int *mfirst;    /* [Mstep+1]                */
  int *mactive;   /* [Mstep][multistep+1]     */
  int *dstepL;    /* [multistep+1][Mstep]     */
  int *dstepN;    /* [multistep+1][Mstep]     */
} orc_mstep_tables;

orc_mstep_tables *orc_mstep_create(int multistep);
void   orc_mstep_free(orc_mstep_tables *t);
/* flat copies for ctypes: arrays sized as documented above */
void   orc_mstep_export(const orc_mstep_tables *t, int *mintvl, int *mfirst,
                        int *mactive, int *dstepL, int *dstepN);

/* expcoef = sum_{M<mfirst[mdrft]} (a L + b N) + sum_{M>=mfirst} N
 * (src/SphericalBasis.cc:1231-1333; src/CylEXP.cc:192-282).  coefL/coefN are
 * [(multistep+1)][ncoef].                                                    */
void   orc_mstep_combine(const orc_mstep_tables *t, int mdrft, long ncoef,
                         const double *coefL, const double *coefN, double *coef);

/* time-step criterion + level choice (src/multistep.cc:52-236), one particle.
 * dynfrac = {D,V,S,A,P}; returns the new level; *dtreq receives dt.           */
/* its two halves: the smallest criterion (src/multistep.cc:94-130) and the level rule on the float dtreq (:160-196) */
double orc_level_dt(const double *dynfrac, double scale, const double *v, const double *a, double pot);
int    orc_level_rule(double dtime, int multistep, int mfirst_mdrft, int cur_level, int shiftlevl, float dtreq_f);
int    orc_level_select(double dtime, int multistep, int mfirst_mdrft, int cur_level,
                        int shiftlevl, const double *dynfrac, double scale,
                        const double *v, const double *a, double pot, double *dtreq);

/* One block-multistep master step (src/step.cc:98-269) of a single self-gravitating spherical
 * component.  level[n] in/out; coefN/coefL [(multistep+1)][(lmax+1)^2*nmax] in/out; coef_out
 * receives the last combined coefficient set; *nswitch the number of level changes.          */
void   orc_sph_multistep_step(const orc_slgrid *g, const orc_sph_params *P, int multistep,
                              double dtime, const double *dynfrac, int shiftlevl, long n,
                              double *x, double *y, double *z, double *vx, double *vy, double *vz,
                              double *ax, double *ay, double *az, double *pot, const double *mass,
                              int *level, const double *center, double *coefN, double *coefL,
                              int this_step, double *coef_out, long *nswitch);

/* begin_run's multistep initialisation (src/begin.cc:80-129) for the same single component. */
void   orc_sph_multistep_init(const orc_slgrid *g, const orc_sph_params *P, int multistep,
                              double dtime, const double *dynfrac, int shiftlevl, long n,
                              const double *x, const double *y, const double *z, const double *vx,
                              const double *vy, const double *vz, double *ax, double *ay, double *az,
                              double *pot, const double *mass, int *level, const double *center,
                              double *coefN, double *coefL, double *coef_out);

/* SphericalBasis::multistep_update (src/SphericalBasis.cc:1156-1228) for one particle given in the
 * centred frame: val[(lmax+1)^2*nmax]; returns 1 when the particle is inside the window r < rmax. */
int    orc_sph_multistep_update(const orc_slgrid *g, const orc_sph_params *P, double xx, double yy,
                                double zz, double mass, double *val);

/* Component::fix_positions (src/Component.cc:3280-3554): out = {mtot, com, cov, coa};
 * lev_sums[(multistep+1)][10] persists between calls (levels < mlevel are not re-summed).     */
void   orc_fix_positions(long n, const double *mass, const double *x, const double *y,
                         const double *z, const double *vx, const double *vy, const double *vz,
                         const double *ax, const double *ay, const double *az, const int *level,
                         int multistep, int mlevel, double *lev_sums, double *out);

/* ... with the escape bookkeeping (consp / tidal / rcom, src/Component.cc:3317-3334, :4204-4212; iattr [n] in/out, NULL: off)
 * and the freeze test (:3336, :4194-4202; rtrunc >= 1e20: off) of the thread body.                                      */
void   orc_fix_positions_opts(long n, const double *mass, const double *x, const double *y,
                              const double *z, const double *vx, const double *vy, const double *vz,
                              const double *ax, const double *ay, const double *az, const int *level,
                              int multistep, int mlevel, const double *com0, const double *center,
                              double rcom, int *iattr, double rtrunc, double *lev_sums, double *out);

/* ---- Orient (src/Orient.H, src/Orient.cc) -------------------------------------------------- */
#define ORC_ORIENT_HIST 64
typedef struct {
  int keep, many;
  unsigned oflags, cflags;        /* AXIS=1, CENTER=2;  DIAG=1, KE=2, EXTERNAL=4            */
  double deltaT, damp;
  int linear;
  double center[3], center0[3], cenvel0[3], axis[3], axis1[3], center1[3];
  double body[9], orig[9];        /* row-major                                              */
  double lasttime, Ecurr, sigA, sigC, sigCz, mtot;
  long used;
  int nA, nC;                     /* history lengths (sumsA, sumsC)                         */
  double tA[ORC_ORIENT_HIST], vA[ORC_ORIENT_HIST][3], tC[ORC_ORIENT_HIST], vC[ORC_ORIENT_HIST][3];
} orc_orient;
/* Orient::Orient (src/Orient.cc:38-80); the log-file restart is orc_orient_restart */
void   orc_orient_init(orc_orient *o, int keep, int many, unsigned oflags, unsigned cflags,
                       double deltaT, double damp);
/* Orient::accumulate + accumulate_cpu (src/Orient.cc:325-747), one process */
void   orc_orient_accumulate(orc_orient *o, double time, double dtime, long n, const double *mass,
                             const double *x, const double *y, const double *z, const double *vx,
                             const double *vy, const double *vz, const double *pot);
/* the log file: the restart block of the constructor (src/Orient.cc:84-335) and logEntry (:742-785) */
long   orc_orient_restart(orc_orient *o, const char *logfile, int restart, double tnow, double dtime,
                          int Mstep, int naccel, double *queue7, int *nq);
int    orc_orient_log_entry(const orc_orient *o, const char *logfile, double time, const double *com,
                            const double *com0, const double *accel, const double *omega, const double *domdt);
void   orc_euler_slater(double phi, double theta, double psi, int body, double *out9);
/* QuadLS (include/QuadLS.H:17-53): y = a x^2 + b x + c; out = {a, b, c} */
/* pyEXP coefficient covariance by sub-sampling: Spherical::accumulate with pcavar
 * (expui/BiorthBasis.cc:583-665).  used0 = accepted particles before this call; per sub-sample
 * T = used % sampT: counts[T], masses[T], mean[T][lm][n] (re, im) and covr[T][lm][n][n2] (the real
 * part of g g^dagger mass; its imaginary part is identically zero).  Accumulates; returns the
 * number of accepted particles. */
long   orc_pyexp_sph_covariance(const orc_slgrid *g, const orc_sph_params *P, long n,
                                const double *x, const double *y, const double *z, const double *mass,
                                int sampT, long used0, long *counts, double *masses, double *mean,
                                double *covr);
/* ---- tuned CPU baseline (tuned_cpu.c): the same arithmetic with the n-dependence hoisted out of
 * the particle loops (moments + contraction, projected table), an upper bound for what a CPU can
 * do with this algorithm; baseline infrastructure only ------------------------------------------ */
typedef struct orc_tuned orc_tuned;
orc_tuned *orc_tuned_create(const orc_slgrid *g);
void   orc_tuned_free(orc_tuned *t);
long   orc_tuned_moments(const orc_slgrid *g, const orc_tuned *t, const orc_sph_params *P, long n,
                         const double *x, const double *y, const double *z, const double *mass,
                         const double *center, double *W /* [(numr-1)][(lmax+1)^2][2], += */);
void   orc_tuned_contract(const orc_slgrid *g, const orc_tuned *t, const double *W, double *coef);
void   orc_tuned_project(const orc_slgrid *g, const orc_tuned *t, const double *coef, double *G);
void   orc_tuned_accel(const orc_slgrid *g, const orc_tuned *t, const orc_sph_params *P, long n,
                       const double *x, const double *y, const double *z, const double *center,
                       const double *G, double *ax, double *ay, double *az, double *pot);
void   orc_quadls(int n, const double *x, const double *y, double *out3);
/* PseudoAccel::operator() (include/PseudoAccel.H:45-91) on a full queue of n rows {t, c[3], a[3]}:
 * accel = 2a of the centre fits, omega = n x dn/dt, domdt = n x d2n/dt2 at the last time */
void   orc_pseudo_accel_fit(int n, const double *rows7, double *accel, double *omega, double *domdt);
/* Component::getPseudoAccel (src/Component.cc:4407-4427) */
void   orc_get_pseudo_accel(int center, int axis, const double *accel, const double *omega,
                            const double *domdt, const double *pos, const double *vel, double *out3);

/* pyEXP field evaluation (expui/BiorthBasis.cc:711-816, :930-958): out[n][9] =
 * {dens m=0, dens m>0, dens, potl m=0, potl m>0, potl, force x3 in the input coordinates};
 * coord 0: (r, cos theta, phi), 1: (R, z, phi), 2: (x, y, z).                                  */
/* pyEXP-literal twins of the n-body thread bodies.  Spherical::accumulate (expui/BiorthBasis.cc:583-665:
 * dsmall 1e-20, particles with r < rmin or r > rmax skipped, cos(m phi) / sin(m phi) evaluated
 * directly): coef[(lmax+1)^2][nmax] +=, returns the number used.  Spherical::computeAccel (:818-926:
 * 1e-18 added to R and r, tables evaluated at r/scale whatever r is, the l = 0 term summed over ALL n
 * and the l >= 1 terms over N1..N2, the azimuthal term divided by the unguarded x^2 + y^2): acc[n][3]. */
long   orc_pyexp_sph_accumulate(const orc_slgrid *g, const orc_sph_params *P, long n, const double *x,
                                const double *y, const double *z, const double *mass, double *coef);
void   orc_pyexp_sph_accel(const orc_slgrid *g, const orc_sph_params *P, const double *coef, long n,
                           const double *x, const double *y, const double *z, double *acc);

void   orc_pyexp_sph_fields(const orc_slgrid *g, const orc_sph_params *P, const double *coef,
                            long n, const double *c1, const double *c2, const double *c3,
                            int coord, double *out);

#ifdef __cplusplus
}
#endif
#endif
