/*
 * exp_amd.h -- C ABI of the MI355X-native BFE hot path (libexp_amd.so).
 *
 * This is the drop-in boundary: plain C, opaque handles, caller-owned host buffers,
 * library-owned device buffers, every call returns an int status (0 = ok) and never
 * throws.  One context per GPU; calls on a context are stream-ordered on its HIP
 * stream and must come from one host thread at a time (the same contract EXP's
 * force methods have: one MPI rank <-> one GPU <-> one stream,
 * src/Component.H:1054-1079).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * EXP source tree).  INTEGRATION.md shows the reference-side binding.
 *
 * Real rows of spherical coefficients use the reference's order
 * (src/SphericalBasis.cc:513-590): l=0; l=1: m0, m1 cos, m1 sin; l=2: ... ;
 * row(l,m,cs) = l*l + (m ? 2*m-1+cs : 0), each row holding nmax doubles.
 */
#ifndef EXP_AMD_H
#define EXP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXP_AMD_OK            0
#define EXP_AMD_ERR_ARG       1   /* bad argument                                  */
#define EXP_AMD_ERR_HIP       2   /* a HIP runtime call failed (see last_error)    */
#define EXP_AMD_ERR_STATE     3   /* call made in the wrong state (e.g. no basis)  */
#define EXP_AMD_ERR_NODEVICE  4   /* no HIP device / code object not loadable      */
#define EXP_AMD_ERR_COMM      5   /* RCCL failure                                  */

typedef struct exp_amd_ctx   exp_amd_ctx;    /* one per GPU                          */
typedef struct exp_amd_comp  exp_amd_comp;   /* particle store of one Component      */
typedef struct exp_amd_force exp_amd_force;  /* one force method (sphereSL/cylinder) */

/* ---- library / context ------------------------------------------------------------ */

/* Environment.  The library reads EIGHT environment variables, each once, where the object it concerns is created; none
 * is needed for correct results, everything a host would want to set at run time has a setter below.
 *   EXP_AMD_SPH_GENERIC=1 / EXP_AMD_CYL_GENERIC=1   (exp_amd_sph_create / exp_amd_cyl_create) every per-particle pass of
 *       that force through the any-order (run-time loop) kernels, which orders above 12 always take
 *       (tests/test_generic_order_gpu.py);
 *   EXP_AMD_NO_LITERAL=1    (exp_amd_sph_create, logarithmic radial map only) particles far outside the table keep the
 *       factored radial derivative instead of the reference's literal term-by-term form (tests/test_sph_gpu.py);
 *   EXP_AMD_CYL_TWIN=0      (exp_amd_cyl_create) the sine tables are fetched even where they equal the cosine tables bit
 *       for bit (tests/test_cyl_gpu.py);
 *   EXP_AMD_SIM_OVERLAP=0   (exp_amd_sim_init / exp_amd_sim_step) the two-component block-multistep driver keeps both
 *       components on the context's one stream (tests/test_config4_gpu.py);
 *   EXP_AMD_STEP_GRAPH=0    (exp_amd_step_kdk_n) never capture: every step eager (tests/test_rccl_gpu.py);
 *   EXP_AMD_POISON=1        (every device allocation) fresh device memory reads as NaN patterns (tests/test_poison_gpu.py);
 *   EXP_AMD_APPEND_MIN=n    (exp_amd_ctx_create) the default of exp_amd_ctx_set_append_min (tests/test_sph_gpu.py).
 *   EXP_AMD_APPEND_LEAN=1   (exp_amd_ctx_create) the default of exp_amd_ctx_set_append_lean.
 *   EXP_AMD_APP_DEBUG=1     (every append step) one line on stderr: particles the placing pass sent to the tail region, particles
 *                           that found no room (the step is then redone from its source).
 * The tuning and A/B switches of the development rounds (tile sizes, launch reductions that can be undone, ...) are
 * compile-time constants of the default build; `make EXPERIMENTAL=1` (-DEXP_AMD_EXPERIMENTAL) turns each
 * EXPAMD_EXPT("NAME", default) of exp_amd/csrc/ back into an environment variable for A/B runs.                       */

/* ABI version of this header (bumped on any signature change). */
int         exp_amd_abi_version(void);
/* Static description of the last error on this context (or of the failed create). */
const char *exp_amd_last_error(const exp_amd_ctx *ctx);
const char *exp_amd_last_global_error(void);

/* Replaces the per-rank device set-up of src/begin.cc:146-210 (one GPU per rank,
 * one stream per component).  `stream` may be NULL (the context then creates its own
 * non-blocking stream) or an existing hipStream_t the caller wants work ordered on. */
int  exp_amd_ctx_create(int device, void *stream, exp_amd_ctx **out);
void exp_amd_ctx_destroy(exp_amd_ctx *ctx);
int  exp_amd_ctx_synchronize(exp_amd_ctx *ctx);
/* The APPEND form of the fused step (exp_amd_step_kdk, spherical force, single level, steady state: same dt, same centre):
 * single-level components of at least `nmin` particles are stepped without sort passes -- the force pass, which knows where
 * every particle will be at the next step, places it in that step's cell order itself (regions per cell in the other buffer
 * set, one reservation per block and destination cell) -- and every other call first turns the store back into an ordinary
 * one.  Same trajectories up to the order of the coefficient sums.  Default: 2^20; 0 turns it off; nmin < 0: as -nmin with
 * regions that have no slack at all, so that a pass runs out of room and the step is redone from its source (the tests' way
 * into that path).  The layout needs about a quarter more device memory than the ordinary store (the slack, the source-slot
 * index, the scratch of the way back); a component for which that is not free keeps the ordinary step.  A caller that looks
 * at the particles every few steps is recognised: each exit keeps the mode off for 8, 16, ... 1024 further steps.          */
int  exp_amd_ctx_set_append_min(exp_amd_ctx *ctx, long long nmin);
/* ... with the LEAN payload (on != 0): the placing pass stores neither the acceleration nor the potential -- 32 of the 88
 * bytes a particle, which no pass of the next step reads; that pass is sensitive to what it stores: 5.7 -> 5.2 ms at 1e8 --
 * and the first call that looks at the component (which turns the store back into an ordinary one) has them evaluated at the
 * positions of the completed step from the coefficient set KEPT at that step, with the centre of that step, whatever has
 * been done to the force since; a force that is destroyed does so for its components first.  Values: those of an
 * evaluation of that set at those positions -- the ones the in-step pass kicked with up to the last place or two for the
 * few particles its general pass took.  Off by default (the state a step leaves in the store is then complete without a
 * further pass): for runs that step many times between two looks at the particles.  Reference: the acceleration and the
 * potential of a step are Particle::acc / ::pot after SphericalBasis::determine_acceleration_and_potential,
 * src/SphericalBasis.cc:1476-1660.                                                                                        */
int  exp_amd_ctx_set_append_lean(exp_amd_ctx *ctx, int on);
/* Tuning knob of exp_amd_step_kdk: single-level components of at least `nmin` particles are stepped
 * as two independently cell-sorted halves so that the HBM-bound sort passes of one half overlap the
 * VALU-bound accumulate / force passes of the other on a second HIP stream (same results up to the
 * order of the coefficient sums).  nmin <= 0 (the default) turns it off.
 * Measured on MI355X at 1e8 particles: 12.33 -> 12.12 ms per step only, because the
 * co-running kernels slow each other down (force 5.6 -> 7.1 ms, accumulate 2.9 -> 4.5 ms).        */
int  exp_amd_ctx_set_split_min(exp_amd_ctx *ctx, long long nmin);
/* The fused KDK step (exp_amd_step_kdk) ends with a force pass that knows the next step: it writes that
 * step's sort keys.  With prekick on (the default) the same pass stores the
 * velocities with BOTH half-kicks around the step boundary applied -- v + a dt/2 (closing), then + a dt/2
 * (opening), two rounding steps as src/incvel.cc:15-88 would take them -- so that the next step's
 * reordering pass only drifts and never reads the accelerations (24 B per particle-step less).  The
 * trajectory is bit-identical to the sequence kick, kick, drift.  Anything that reads velocities at the
 * step boundary sees v - a dt/2 formed on the way out (exp_amd_comp_download: non-destructively, within
 * one ulp of the closing-kick value; exp_amd_comp_fix_positions and exp_amd_orient_accumulate read the
 * same way: a diagnostic between two steps does not move the trajectory by a bit) or after the opening
 * half-kick has been taken back (every call that CHANGES the component; a following step_kdk with
 * another dt included).  Off: the closing half-kick is left to the
 * next reordering pass, which then reads the accelerations.  Measured at 1e8 / S10: the reordering pass
 * 2.94 -> 2.44 ms, the force pass 4.82 -> 5.01 ms for its three extra stores, the step 2 % faster.       */
int  exp_amd_ctx_set_prekick(exp_amd_ctx *ctx, int on);
/* Deterministic mode (off by default).  The coefficient sums
 * are reductions over millions of particles by fp64 atomics in whatever order the hardware serves
 * them, so two runs agree to rounding (~1e-15), not bit for bit -- like the reference's thread and MPI
 * reduction order.  With `on`, every particle's contribution is first rounded to a fixed absolute
 * grid 2^e chosen from a bound of the sum (4 pi x sum|m| x max|table|), fine enough to stay far below
 * the 1e-10 coefficient tolerance; all additions are then exact, hence independent of their order,
 * and a run is bit-reproducible on a given number of ranks.  Costs two extra fp64 operations per
 * accumulated term (~1.5x the accumulation kernels, ~10 % of a step).                             */
int  exp_amd_ctx_set_deterministic(exp_amd_ctx *ctx, int on);
/* Tuning knob of the block-multistep step loop (exp_amd_sim_step): time-step levels holding fewer
 * than `nmin` particles are kept level-contiguous but not cell-sorted -- advanced in place,
 * accumulated with per-particle atomics, forces by the gather path -- because a sparse level has
 * about one particle per basis cell and the cell order buys nothing.  Same results up to the order
 * of the sums.  nmin < 0 (the default): each force method's own
 * break-even (about 3e6-5e6 / moments per particle: ~51000 for lmax 6, ~20000 for lmax 10, ~58000 for mmax 6);
 * 0: every level is cell-sorted.                                                                  */
int  exp_amd_ctx_set_dense_min(exp_amd_ctx *ctx, long long nmin);
/* ... and what happens to a sub-step whose ACTIVE particles are few (the upper levels of a settled run: hundreds to
 * thousands of particles, stepped 2^level times per master step): when every active level of a component is sparse
 * and together they hold at most `nmax` particles, their coefficients are accumulated and the forces on them are
 * evaluated straight from the basis tables, particle by particle -- the reference's own formulation
 * (src/SphericalBasis.cc:429-599, :1476-1660; exputil/EmpCylSL.cc:4049-4146, :5256-5410) -- instead of through cell
 * moments and a projected table whose fixed costs (contraction over every cell, projection of every table row) dwarf
 * the particles' own work.  Any force evaluation on at most `nmax` target particles (a cross force on another
 * component's thin active set included) takes the same route.  Same results up to the order of the sums.
 * Default 8192 (the direct kernels cost ~5 ns per particle and kernel against the table
 * path's ~100 us of fixed costs per sub-step and component); 0: never.  The ACCUMULATION side alone takes the direct
 * route up to 4 x nmax particles (one launch there against three), and the level-change
 * differencing of at most nmax movers does too.                                                                   */
int  exp_amd_ctx_set_thin_max(exp_amd_ctx *ctx, long long nmax);
/* Second knob of the same loop: how the coefficient sets are differenced when particles change level
 * (multistep_update, src/SphericalBasis.cc:1156-1228, src/CylEXP.cc:159-188).  The slots of the movers of a
 * sweep are compacted into a list; below `nmin` movers each adds and subtracts its own contribution (straight
 * from the basis tables up to thin_max of them, through staged fp64 atomics above), from `nmin` on the list goes
 * through the accumulation kernels, which sum runs of equal
 * (level, cell) in registers (the first sweeps of a run move several per cent of a component at once).
 * Same results up to rounding (each way is order-independent in deterministic mode).  Default 8192;
 * 0: always the accumulation kernels; < 0: never.                */
int  exp_amd_ctx_set_mover_list_min(exp_amd_ctx *ctx, long long nmin);
void *exp_amd_ctx_stream(exp_amd_ctx *ctx);

/* Coefficient all-reduce across ranks.  Replaces the MPI_Allreduce calls of
 * src/SphericalBasis.cc:864-903 and exputil/EmpCylSL.cc:4188-4222 by ONE on-stream
 * reduction of the contiguous device coefficient buffer.
 *   (a) native RCCL: exchange the 128-byte id out of band (MPI_Bcast / torch), then
 *       every rank calls exp_amd_comm_init_rank;
 *   (b) host-provided: register a callback that reduces `count` doubles in place at
 *       device pointer `buf` on `stream` (e.g. torch.distributed.all_reduce or a
 *       GPU-aware MPI_Allreduce).  With neither, the context is single-rank.       */
int  exp_amd_comm_get_unique_id(void *id128);
int  exp_amd_comm_init_rank(exp_amd_ctx *ctx, const void *id128, int nranks, int rank);
typedef int (*exp_amd_allreduce_fn)(void *buf, size_t count, void *stream, void *user);
int  exp_amd_comm_set_callback(exp_amd_ctx *ctx, exp_amd_allreduce_fn fn, void *user);
/* The world the callback of (b) reduces over (MPI_Comm_size / MPI_Comm_rank of the host's communicator): what
 * exp_amd_comm_info then reports, and what rank-dependent host logic above the ABI may read back.  (a) sets both
 * itself; contradicting it is EXP_AMD_ERR_ARG.  A context told nranks > 1 that has neither a communicator nor a
 * callback fails its first reduction with EXP_AMD_ERR_COMM instead of silently keeping rank-local sums.          */
int  exp_amd_comm_set_world(exp_amd_ctx *ctx, int nranks, int rank);
/* Which reduction the context uses -- kind 0: none (single rank), 1: the library's RCCL communicator,
 * 2: the host's callback -- with the rank count / rank it was given and the number of all-reduces
 * issued so far; any output pointer may be NULL.                                                */
int  exp_amd_comm_info(exp_amd_ctx *ctx, int *kind, int *nranks, int *rank, long long *calls);
/* 2 when the context can reduce on two streams at once -- a single rank, a host callback (it is handed the stream), or an
 * RCCL communicator that could be split into a second one for the auxiliary stream (ncclCommSplit) -- and the
 * two-component step driver therefore keeps its two-stream schedule with several ranks; 1 otherwise.  A QUERY: no
 * collective is issued (one rank may ask alone); the split itself happens where the step driver decides its schedule
 * (exp_amd_sim_begin_run / exp_amd_sim_step), which every rank reaches alike.                                    */
int  exp_amd_comm_streams(exp_amd_ctx *ctx);
/* MAX over the ranks of one HOST number through the same transport (a sum of one-hot slots): for host logic that must
 * agree on a count before it issues collectives (e.g. how many batches a reader is cut into, each ending in one
 * all-reduce).  Single-rank contexts return at once; a callback context that was never told its world is
 * EXP_AMD_ERR_COMM.                                                                             */
int  exp_amd_comm_allreduce_max(exp_amd_ctx *ctx, double *value);
/* The collective itself: in-place sum over the ranks of `count` doubles at DEVICE pointer `buf`, on
 * the context's stream (what replaces MPI_Allreduce, src/SphericalBasis.cc:864-903), so that a host
 * can verify the communicator it has just set up.                                               */
int  exp_amd_comm_allreduce(exp_amd_ctx *ctx, void *buf, size_t count);

/* ---- particle store ------------------------------------------------------------------
 * Replaces the CUDA particle mirror of src/cudaComponent.cu:621-727
 * (ParticlesToCuda / CudaToParticles) with an SoA store in HBM.  Host arrays are
 * length n, fp64; any pointer may be NULL on upload (=> zeros) or download (=> skip).
 * The store may physically reorder particles (cell sort); downloads always return the
 * caller's original order.                                                           */
int  exp_amd_comp_create(exp_amd_ctx *ctx, size_t n, exp_amd_comp **out);
void exp_amd_comp_destroy(exp_amd_comp *c);
size_t exp_amd_comp_size(const exp_amd_comp *c);
int  exp_amd_comp_upload(exp_amd_comp *c, const double *mass,
                         const double *x, const double *y, const double *z,
                         const double *vx, const double *vy, const double *vz);
/* Positions (and velocities) as the caller holds them -- three columns (stride 1) or one [n][3] array (stride 3, passed as
 * x; y and z ignored) -- with the expansion frame of Basis::addFromArray / createFromReader applied on the device:
 * x' = rot (x - center), v' = rot v (expui/BiorthBasis.cc:4555-4568, :4616-4738); center / rot may be NULL (no shift / no
 * rotation; rot is row-major).  Saves the caller the strided column copies and the [n,3] x [3,3] product.           */
int  exp_amd_comp_upload_frame(exp_amd_comp *c, const double *mass, const double *x, const double *y, const double *z,
                               const double *vx, const double *vy, const double *vz, int stride,
                               const double center[3], const double rot[9]);
int  exp_amd_comp_upload_acc(exp_amd_comp *c, const double *ax, const double *ay,
                             const double *az, const double *pot);
/* Per-particle multistep levels (Particle::level).  The store keeps its particles PARTITIONED by level -- the
 * counterpart of Component::levlist -- and level-specific calls (kick, drift, zero_acc with mlevel >= 0) work on those
 * ranges: uploaded levels take effect for them at the next accumulation of a multistep force on this component, which
 * re-partitions the store (every driver accumulates before it advances); until a multistep force has done so the store
 * has one level, and a call for a level beyond it is refused with EXP_AMD_ERR_ARG.                                 */
int  exp_amd_comp_upload_levels(exp_amd_comp *c, const int32_t *level);
int  exp_amd_comp_download(exp_amd_comp *c, double *mass, double *x, double *y, double *z,
                           double *vx, double *vy, double *vz,
                           double *ax, double *ay, double *az, double *pot);
int  exp_amd_comp_download_levels(exp_amd_comp *c, int32_t *level);
/* Adopt device-resident SoA arrays (device pointers, length n) without a host trip:
 * contents are copied device-to-device on the context stream.                      */
int  exp_amd_comp_upload_device(exp_amd_comp *c, const double *mass,
                                const double *x, const double *y, const double *z,
                                const double *vx, const double *vy, const double *vz);
/* Expansion centre subtracted from positions (Component::Centered, src/Component.H:748-757). */
int  exp_amd_comp_set_center(exp_amd_comp *c, const double center[3]);
/* Component::center as it stands (set by the caller or by the orientation estimator of a sim; the C(x), C(y), C(z)
 * columns of the run log, src/OutLog.cc:444) */
int  exp_amd_comp_get_center(const exp_amd_comp *c, double center[3]);
/* Body-frame rotation applied after centring by the cylindrical force method (row-major 3x3 =
 * Orient::transformBody; positions go in as body * (x - centre), forces come back through the
 * transpose, transformOrig: src/Cylinder.cc:799-800, :1352-1353, :1417-1418).  NULL = none.
 * The spherical method ignores it, as Sphere does.                                            */
int  exp_amd_comp_set_orientation(exp_amd_comp *c, const double body[9]);
/* Component::rtrunc with the com0 it is measured from (the "rtrunc" key of a component, src/Component.cc:69, :213,
 * :1023): Component::freeze(i) (:4194-4202) is |pos - com0 - center|^2 > rtrunc^2 with `center` the one of
 * exp_amd_comp_set_center.  A frozen particle takes no part in ANY force method's accumulation, level-change differencing
 * or force pass (src/SphericalBasis.cc:468, :1159, :1521; src/Cylinder.cc:788, :842, :1329, :1756): nothing is added to its
 * acc / pot -- not even the frame's pseudo-acceleration -- while kicks and drifts still move it.  The test is made
 * with the positions of the moment of each call, in the reference's operation order.  com0 NULL = zeros; rtrunc >= 1e20
 * (the reference's default) switches the test off.                                                                  */
int  exp_amd_comp_set_rtrunc(exp_amd_comp *c, double rtrunc, const double com0[3]);
/* Acceleration of the component's non-inertial frame, subtracted by every force applied to its
 * particles (Component::AddAcc -> getPseudoAccel, src/Component.H:914-921, src/Component.cc:4407-4427):
 * accel (the EJ centre's acceleration; NULL: none) plus, when omega and domdt are both given, the
 * Coriolis, Euler and centrifugal terms 2 omega x v + domdt x x + omega x (omega x x) of the
 * rotating axis frame, from the stored position and velocity.                                  */
int  exp_amd_comp_set_pseudo_accel(exp_amd_comp *c, const double accel[3], const double omega[3],
                                   const double domdt[3]);

/* Leapfrog pieces.  Replace incr_position(dt, mlevel) (src/incpos.cc:72) and
 * incr_velocity(dt, mlevel) (src/incvel.cc:90); mlevel < 0 means all levels.       */
int  exp_amd_comp_drift(exp_amd_comp *c, double dt, int mlevel);
int  exp_amd_comp_kick (exp_amd_comp *c, double dt, int mlevel);
/* Replaces the zeroing loop of ComponentContainer::compute_potential
 * (src/ComponentContainer.cc:641-665): acc = pot = 0 for levels >= mlevel.         */
int  exp_amd_comp_zero_acc(exp_amd_comp *c, int mlevel);

/* Centre of mass / velocity / acceleration of the component: replaces Component::fix_positions
 * (src/Component.cc:3280-3554; CUDA twin src/cudaComponent.cu:800-933) without the orientation centre (that is
 * exp_amd_orient_*); frozen particles (exp_amd_comp_set_rtrunc) and, with exp_amd_comp_set_consp, escaped ones are left
 * out as in the thread body (:3317-3336).  com_system is off in this scope (no comE / covE sums).  Only the levels
 * >= mlevel are re-summed (the others keep their previous per-level sums, as the reference
 * does); ranks are combined with the context's all-reduce.  out = {mtot, com[3], cov[3], coa[3]}
 * (the three vectors divided by mtot when mtot > 0).                                        */
int  exp_amd_comp_fix_positions(exp_amd_comp *c, int mlevel, double out[10]);
/* The component keys "noswitch", "freezeL", "dtreset" (src/Component.cc:253-255, :1036-1038), read by
 * adjust_multistep_level (src/multistep.cc:136-158, :528-534).  freeze_levels: levels are assigned on the first call only
 * (`if (not firstCall and c->FreezeLev()) apply = false;`) -- exp_amd_force_adjust_multistep_level and the step driver's
 * sweeps then move nothing of this component.  noswitch: Particle::dtreq (a float per particle, kept on the device by particle
 * id) holds the smallest time step asked for since its last reset -- at mstep == 0 when dtreset is set, and on the first call
 * (:136-141) -- and levels are only assigned at the end of a master step (mdrft == Mstep) or on the first call (:147); the
 * sweeps in between examine their levels for dtreq's sake and move nothing.  dtreset is only read with noswitch.          */
int  exp_amd_comp_set_level_policy(exp_amd_comp *c, int noswitch, int freeze_levels, int dtreset);
/* The escape bookkeeping of Component::fix_positions: the component keys "tidal" (which switches `consp` on and names the
 * integer attribute that holds the flag, src/Component.cc:998-1000) and "rcom" (:1024).  With it on, fix_positions flags a
 * particle of the examined levels that is beyond rcom of com0 + center (Component::escape_com, :4204-4212; com0 is the one of
 * exp_amd_comp_set_rtrunc, zeros by default) -- iattrib[tidal] = 1 -- and leaves it out of the sums from then on (:3317-3334).
 * The flags live on the device, one byte per particle in the caller's order; they start at zero.  get / set: the attribute
 * column as a body file holds it (a restart sets it before the first step).  on = 0 switches the test off and keeps the flags. */
int  exp_amd_comp_set_consp(exp_amd_comp *c, int on, double rcom);
int  exp_amd_comp_get_escaped(exp_amd_comp *c, unsigned char *flags /* [n] */);
int  exp_amd_comp_set_escaped(exp_amd_comp *c, const unsigned char *flags /* [n], 0 or 1 */);
/* The per-component sums of the run log (OutLog::Run, src/OutLog.cc:392-478): out = {mass, m x [3], m v [3], angular
 * momentum [3], kinetic energy, 0.5 m pot, Clausius virial m x.a, number of bodies}, reduced over the ranks; a frozen
 * particle (beyond rtrunc, exp_amd_comp_set_rtrunc) is left out of the sums as in src/OutLog.cc:460 and still counted in the
 * number of bodies; positions and velocities as stored (com_system off), velocities at the step boundary.              */
int  exp_amd_comp_log_sums(exp_amd_comp *c, double out[14]);

/* ---- orientation / expansion-centre estimator ("EJ") ----------------------------------------
 * Replaces class Orient (src/Orient.H:31-204, src/Orient.cc:38-790; CUDA twin
 * src/cudaOrient.cu:109-199), created by Component::initialize with (nEJkeep, nEJwant, EJ flags,
 * EJkinE, EJdT, EJdamp) (src/Component.cc:1323-1370) and consulted by Component::fix_positions
 * (:3569-3582) and the cylinder force (src/Cylinder.cc:799, :1352).
 *   oflags: 1 = AXIS, 2 = CENTER (Orient::OrientFlags);  cflags: 2 = KE (Orient::ControlFlags; DIAG
 *   is ignored; 4 = EXTERNAL is accepted and adds nothing: Particle::potext only holds the potential of the External
 *   force plug-ins, which are outside this build -- cross forces between components add to pot, src/SphericalBasis.cc:
 *   1652, src/Cylinder.cc:1416).  keep >= 1.
 * accumulate(time, dtime, c) is Orient::accumulate(time, c) with the global time step passed in:
 * the `want` most bound particles (E = pot [+ v^2/2]) are selected on the device -- exactly, over
 * all ranks of the context -- and their mass-weighted position and angular momentum about the
 * current centre enter the damped least-squares histories; calls closer than deltaT are skipped.
 * get: centre, axis, the body/original rotations (row-major 3x3, return_euler_slater
 * exputil/euler_slater.cc:46) and stats = {Ecurr, used, sigA, sigC, sigCz, mtot, axis1[3],
 * center1[3], center0[3]}; any output pointer may be NULL.                                    */
typedef struct exp_amd_orient exp_amd_orient;
int  exp_amd_orient_create(exp_amd_ctx *ctx, int keep, int want, unsigned oflags, unsigned cflags,
                           double deltaT, double damp, exp_amd_orient **out);
void exp_amd_orient_destroy(exp_amd_orient *o);
int  exp_amd_orient_set_center(exp_amd_orient *o, const double center[3]);   /* Orient::set_center */
int  exp_amd_orient_set_cenvel(exp_amd_orient *o, const double vel[3]);      /* Orient::set_cenvel */
int  exp_amd_orient_set_linear(exp_amd_orient *o);                           /* Orient::set_linear */
int  exp_amd_orient_accumulate(exp_amd_orient *o, double time, double dtime, exp_amd_comp *c);
unsigned exp_amd_orient_flags(const exp_amd_orient *o);                     /* the orient flags */
/* The pseudo-acceleration helper of Orient (Naccel constructor argument, include/PseudoAccel.H):
 * quadratic least squares over the last `naccel` (time, centre, axis) estimates; accel = centre
 * acceleration (CENTER), omega / domdt = angular velocity of the axis and its rate (AXIS).  The step
 * loop hands them to the component (exp_amd_comp_set_pseudo_accel) together with the centre.    */
int  exp_amd_orient_set_naccel(exp_amd_orient *o, int naccel);
int  exp_amd_orient_accel(exp_amd_orient *o, double accel[3], double omega[3], double domdt[3]);
int  exp_amd_orient_get(const exp_amd_orient *o, double center[3], double axis[3], double body[9],
                        double orig[9], double stats[15]);
/* Orient's Logfile constructor argument with the restart block of the constructor
 * (src/Orient.cc:84-335).  Call after create / set_naccel.  Rank 0 does the file work:
 *   - no file: the two header rows are written (:236-284);
 *   - a file: it is moved to <logfile>.bak and, with flags & 1 (the global `restart`), its data rows
 *     up to tnow + 0.1*dtime/Mstep are copied into a fresh <logfile> and rebuild Ecurr, axis, centre,
 *     centre0, the last `keep` (time, axis1) / (time, centre1) pairs of the two histories, the
 *     pseudo-acceleration queue and the body/orig rotations; the state then goes to every rank.
 * The reference queues the row's logged pseudo-ACCELERATION where accumulate() queues centre1
 * (:185 vs :711); that is reproduced unless flags & 2 asks for centre1.  *rows (may be NULL) = data
 * rows taken.  Values come back with the 6 significant digits the log holds.
 * log_entry is Orient::logEntry(time, c) (:742-785): one 33-column row appended by rank 0 -- time,
 * Ecurr, used, axis, axis1, centre, centre0, centre1, com, com0 (NULL: zeros), pseudo-acceleration,
 * omega, domega/dt.  A sim whose estimator has a log open writes the row itself after each
 * accumulate (src/ComponentContainer.cc:1386-1389).                                          */
int  exp_amd_orient_open_log(exp_amd_orient *o, const char *logfile, unsigned flags, double tnow,
                             double dtime, int Mstep, long long *rows);
int  exp_amd_orient_log_entry(exp_amd_orient *o, double time, const double com[3], const double com0[3]);

/* ---- spherical force method (sphereSL) -----------------------------------------------
 * Replaces class Sphere : SphericalBasis (src/Sphere.cc:28-96, src/SphericalBasis.cc)
 * given the SLGridSph tables (exputil/SLGridMP2.cc: ev, ef, p0 on the xi grid).     */
typedef struct {
  int    lmax, nmax, numr, cmap;       /* Lmax, nmax, numr, cmap keys                  */
  double rmap;                         /* rmapping                                     */
  double scale;                        /* scale                                        */
  double rmin, rmax;                   /* expansion window in unscaled r               */
  double xmin, dxi;                    /* xi grid origin / spacing (SLGridMP2.cc:1355) */
  int    NO_L0, NO_L1, EVEN_L, EVEN_M, M0_only;   /* src/SphericalBasis.cc:28-52       */
  int    multistep;                    /* number of extra time-step levels (0 = none)  */
} exp_amd_sph_config;

/* xi[numr], p0[numr], ev[(lmax+1)*nmax], ef[(lmax+1)*nmax*numr] (ef(n,i) of table l).
 * lmax: any order the reference's YAML may ask for (src/Sphere.cc:28-96), 0 <= lmax <= 64; up to 12 the per-particle
 * kernels are unrolled at compile time, above that run-time-loop kernels take over (slower, same results;
 * EXP_AMD_SPH_GENERIC=1 selects them at any order).  EXP_AMD_ERR_ARG outside that range.           */
int  exp_amd_sph_create(exp_amd_ctx *ctx, const exp_amd_sph_config *cfg,
                        const double *xi, const double *p0,
                        const double *ev, const double *ef, exp_amd_force **out);
void exp_amd_force_destroy(exp_amd_force *f);
/* continuation != 0 (default): r > rmax uses the exterior multipole continuation of the n-body
 * force (src/SphericalBasis.cc:1555-1560, :1605-1628); 0: tables evaluated at r/scale as in
 * pyEXP's Spherical::computeAccel (expui/BiorthBasis.cc:818-926).                          */
int  exp_amd_sph_set_exterior(exp_amd_force *f, int continuation);
/* FIX_L0 (src/SphericalBasis.cc:34, :119, :1689-1694): with on != 0 the next force evaluation saves the l = 0 row of the
 * coefficient set (nmax values) and every later one -- self or external -- copies it back into the active set first.  */
int  exp_amd_sph_set_fix_l0(exp_amd_force *f, int on);
/* The "ssfrac" key (src/SphericalBasis.cc:149-152, :437-440, :459-460, :472-473): with 0 < ssfrac < 1 the coefficients are
 * accumulated from a sub-sample -- thread id of `nthrds` takes the entries [n id / nthrds, floor(ssfrac n (id + 1) / nthrds))
 * of the level list (the END index is scaled, as in the reference: with several threads the later slices are short or empty)
 * and every mass is divided by ssfrac.  The level list is the CALLER's particle order (the reference's is the iteration
 * order of its particle map).  Any other ssfrac switches it off, as the reference's sanity check does.  Single-level forces
 * only: with block multistep the level lists' order is the history of the level changes (EXP_AMD_ERR_STATE).  The fused
 * step (exp_amd_step_kdk) then advances the particles in passes of their own.                                         */
int  exp_amd_sph_set_subset(exp_amd_force *f, double ssfrac, int nthrds);
/* The NOISE keys of SphericalBasis (`NOISE`, `noiseN`, `noise_model_file`, `seedN`: src/SphericalBasis.cc:79-81, :135-147): every
 * force evaluation -- self or external, `if (NOISE) update_noise();` opens get_acceleration_and_potential, :395 -- replaces the
 * coefficient set by draws from the noise model (update_noise, :2150-2210): sqrt(|rmsC(l,n) - meanC[n]^2| factorial(l,m) / noiseN)
 * times a standard normal deviate, plus meanC[n] on the l = 0 row; one std::mt19937 + std::normal_distribution per force,
 * seeded with seedN at the first evaluation after this call.  A self call of a multistep force consumes its draws and leaves
 * the set alone (compute_multistep_coefficients rebuilds it right after, :1680-1685).  meanC[nmax] and rmsC[lmax+1][nmax] are
 * SphericalBasis::compute_rms_coefs (:2108-2147) of the noise model file, computed on the host (exp_amd/slgrid.py:
 * compute_rms_coefs).  meanC == NULL switches the mode off.                                                             */
int  exp_amd_sph_set_noise(exp_amd_force *f, const double *meanC, const double *rmsC, double noiseN, unsigned seedN);
/* M0_only in the accumulation: the n-body code skips the m > 0 sums altogether (src/SphericalBasis.cc:550), pyEXP's
 * Spherical::accumulate applies no flag at all (expui/BiorthBasis.cc:583-665: the coefficients it returns hold every m;
 * only the evaluation drops them, :851).  all_m = 1 selects the latter; the default is the former.                */
int  exp_amd_sph_set_accumulate_all_m(exp_amd_force *f, int all_m);
/* The small number added to r before any division: 1e-16 (DSMALL, src/expand.H:130) by default, as in
 * the n-body code; pyEXP's Spherical::accumulate adds 1e-20 and computeAccel 1e-18
 * (expui/BiorthBasis.cc:588, :824-825).  Only the origin and the polar axis can tell them apart.  */
int  exp_amd_sph_set_dsmall(exp_amd_force *f, double dsmall);

/* PotAccel::set_multistep_level (src/PotAccel.H:285) */
int  exp_amd_force_set_level(exp_amd_force *f, int mlevel);

/* PotAccel::determine_coefficients(Component*) (src/PotAccel.H:178-180 ->
 * SphericalBasis::determine_coefficients_particles, src/SphericalBasis.cc:682-1002):
 * accumulate the particles of level `mlevel` of `c` (all particles if the force has
 * multistep == 0), reduce across ranks, leave the result on the device.             */
int  exp_amd_force_determine_coefficients(exp_amd_force *f, exp_amd_comp *c);

/* Host copies of the current coefficient set, reference real-row order
 * ((lmax+1)^2 rows x nmax).  set_coefs replaces HtoD_coefs
 * (src/cudaSphericalBasis.cu:1437 ff.); get_coefs replaces DtoH_coefs.             */
int  exp_amd_force_get_coefs(exp_amd_force *f, double *coef, size_t count);
int  exp_amd_force_set_coefs(exp_amd_force *f, const double *coef, size_t count);
size_t exp_amd_force_ncoef(const exp_amd_force *f);
/* Per-level coefficient sets of a multistep force: which = 0 -> expcoefN[level] (new),
 * which = 1 -> expcoefL[level] (last)  (src/SphericalBasis.cc:785-792).              */
int  exp_amd_force_get_level_coefs(exp_amd_force *f, int level, int which, double *coef,
                                   size_t count);
/* PotAccel::Used() (src/PotAccel.H:207): particles inside the window at the last
 * accumulation, summed over ranks.                                                  */
int  exp_amd_force_used(exp_amd_force *f, long long *used);

/* PotAccel::get_acceleration_and_potential(Component*) (src/PotAccel.H:173 ->
 * SphericalBasis::determine_acceleration_and_potential, src/SphericalBasis.cc:1663):
 * acc += force, pot += potential for the particles of levels >= mlevel of `target`.
 * `external` != 0 is SetExternal() (src/PotAccel.H:215): `target` is another
 * component evaluated in this force's centred frame.                                */
int  exp_amd_force_get_acceleration(exp_amd_force *f, exp_amd_comp *target, int external);

/* Component::Adiabatic() (src/Component.cc:4214-4220: 0.25 (1 + erf((tnow - ton)/twid)) (1 + erf((toff - tnow)/twid)), the
 * keys ton / toff / twid of the component the basis belongs to) as the host has evaluated it for the current time: every
 * mass read by the accumulation and by the level-change differencing is multiplied by it (src/SphericalBasis.cc:441,
 * :471, :1161; src/Cylinder.cc:834, :1758 -- the cylinder's in-cut mass tally included, :864).  1 by default.  The step
 * driver does this itself for components given to exp_amd_sim_set_adiabatic.                                        */
int  exp_amd_force_set_mass_scale(exp_amd_force *f, double adiabatic);
/* The "self_consistent" key (src/SphericalBasis.cc:33, :114-117; src/Cylinder.cc:76, :557-558).  With on = 0 the
 * coefficients are held fixed once the FIRST determine_coefficients call has completed and the host is not
 * `initializing` (the global of src/begin.cc:80-129, mirrored by exp_amd_force_set_initializing; exp_amd_sim_init sets
 * it around its own work): exp_amd_force_determine_coefficients returns at once (src/SphericalBasis.cc:694,
 * src/Cylinder.cc:959), exp_amd_force_compute_multistep_coefficients leaves the combined set alone (:1682, :1469), the
 * level changes are not differenced (src/Cylinder.cc:1755), exp_amd_step_kdk and the step driver only advance the
 * particles and evaluate the force of the set as it is.  exp_amd_force_coefs_frozen tells whether that point is reached. */
int  exp_amd_force_set_self_consistent(exp_amd_force *f, int on);
int  exp_amd_force_set_initializing(exp_amd_force *f, int on);
int  exp_amd_force_coefs_frozen(const exp_amd_force *f);

/* Multistep coefficient bookkeeping (src/SphericalBasis.cc:1231-1333, :1013-1079).  */
int  exp_amd_force_multistep_reset(exp_amd_force *f);
int  exp_amd_force_compute_multistep_coefficients(exp_amd_force *f, int mdrft);

/* adjust_multistep_level() (src/multistep.cc:344-627) for one component and its force:
 * time-step criteria (dynfrac = {dynfracD, dynfracV, dynfracS, dynfracA, dynfracP},
 * src/global.cc:76-80) -> new levels (shiftlevl-limited, clamped to [mfirst[mdrft], multistep]);
 * PotAccel::multistep_update_begin / multistep_update / multistep_update_finish
 * (src/PotAccel.H:271-281) for the particles that change level; Component::reset_level_lists.
 * first_step != 0 examines every level (src/multistep.cc:451-453).  *nswitch (may be NULL)
 * receives the number of level changes on this rank.                                   */
int  exp_amd_force_adjust_multistep_level(exp_amd_force *f, exp_amd_comp *c, double dtime,
                                          const double dynfrac[5], int shiftlevl, int mdrft,
                                          int first_step, long long *nswitch);

/* ---- cylindrical force method (cylinder) -------------------------------------------------
 * Replaces class Cylinder (src/Cylinder.cc) + EmpCylSL's accumulate / accumulated_eval
 * (exputil/EmpCylSL.cc:4049-4146, :5256-5410) given the EOF tables.  Coefficients are
 * ncoef = 2*(mmax+1)*nmax doubles: accum_cos[m][n] followed by accum_sin[m][n] (row m = 0 of the
 * sine block is zero).                                                                       */
typedef struct {
  int    mmax, nmax;                   /* mmax, nmax (NORDER)                                  */
  int    numx, numy;                   /* ncylnx, ncylny                                       */
  int    cmapr, cmapz;                 /* cmapr, cmapz                                         */
  double ascale, hscale;               /* acyl, hcyl                                           */
  double rtable;                       /* Rtable = RMAX/sqrt(2)  (exputil/EmpCylSL.cc:2130)    */
  double xmin, dx, ymin, dy;           /* grid of EmpCylSL::setup_table (:2131-2137)           */
  double rcylmax;                      /* accumulation cut r^2+z^2 < (rcylmax*acyl)^2          */
  int    EVEN_M;
  int    multistep;
} exp_amd_cyl_config;

/* tab[6][mmax+1][nmax][numx+1][numy+1]: potC, rforceC, zforceC, potS, rforceS, zforceS.
 * mmax: 0 <= mmax <= 64 (src/Cylinder.cc:473); unrolled kernels up to 12, run-time-loop kernels above
 * (EXP_AMD_CYL_GENERIC=1: at any order).                                                            */
int  exp_amd_cyl_create(exp_amd_ctx *ctx, const exp_amd_cyl_config *cfg, const double *tab,
                        exp_amd_force **out);
/* The "mlim" key (src/Cylinder.cc:40, :225 -> EmpCylSL::set_mlim; pyEXP: expui/BiorthBasis.cc:1466, :1620): harmonics
 * m > mlim take no part -- EmpCylSL::get_pot fills Vc / Vs up to min(MLIM, MMAX) (exputil/EmpCylSL.cc:5602), so nothing
 * is accumulated there, and accumulated_eval / accumulated_dens_eval sum up to it (:5317, :5465).  The coefficients of
 * m > mlim read back as zero (the reference leaves them unspecified: rows of Vc it never fills, :4078).  mlim >= mmax
 * changes nothing; mlim < 0 is EXP_AMD_ERR_ARG (the key's default -1 means "do not call"); once set it can only be
 * lowered (EXP_AMD_ERR_STATE otherwise: the tables above it are gone from the device).                              */
int  exp_amd_cyl_set_mlim(exp_amd_force *f, int mlim);
/* Mass of the particles inside the accumulation cut at the last accumulation (Cylinder's
 * cylmass, src/Cylinder.cc:1081-1098), used for the off-grid monopole blend (:1364-1414);
 * set_cylmass overrides it (playback / external coefficient sets).                       */
int  exp_amd_cyl_get_cylmass(exp_amd_force *f, double *mass);
int  exp_amd_cyl_set_cylmass(exp_amd_force *f, double mass);

/* ---- field evaluation at points (pyEXP getFields) ------------------------------------------
 * Spherical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:711-816, :930-958) for the
 * force's current coefficient set.  coord: 0 = (r, cos theta, phi), 1 = (R, z, phi),
 * 2 = (x, y, z); host arrays of n points; out[n][9] = {dens m=0, dens m>0, dens, potl m=0,
 * potl m>0, potl, force x 3 in the input coordinates} (labels: BiorthBasis.cc:71-97).  The
 * density needs SLGridSph's d0 table (4 pi rho0 on the xi grid, exputil/SLGridMP2.cc:913-950),
 * supplied once with exp_amd_sph_set_density.                                               */
int  exp_amd_sph_set_density(exp_amd_force *f, const double *d0 /* [numr] */);
int  exp_amd_sph_fields(exp_amd_force *f, size_t n, const double *c1, const double *c2,
                        const double *c3, int coord, double *out /* [n][9] */);

/* Coefficient covariance by sub-sampling (pyEXP: the `pcavar` / `subsamp` keys, enableCoefCovariance,
 * getCoefCovariance, getCovarSamples; Spherical::accumulate expui/BiorthBasis.cc:583-665, :342-378,
 * expui/BiorthBasis.H:425-470).  enable(sampT) allocates and zeroes (sampT <= 0 frees); accumulate
 * files every particle of `c` inside the expansion window, in the CALLER's order, under sub-sample
 * (used_before + its running count) % sampT and adds its g and g g^dagger; get returns
 * counts[sampT], masses[sampT], mean[sampT][(L+1)(L+2)/2][nmax][2] (re, im) and
 * covr[sampT][(L+1)(L+2)/2][nmax][nmax] (real: the phase cancels).                              */
int  exp_amd_sph_cov_enable(exp_amd_force *f, int sampT);
int  exp_amd_sph_cov_reset(exp_amd_force *f);
int  exp_amd_sph_cov_accumulate(exp_amd_force *f, exp_amd_comp *c, long long used_before,
                                long long *accepted);
int  exp_amd_sph_cov_get(exp_amd_force *f, long long *counts, double *masses, double *mean, double *covr);

/* Cylindrical twin: Cylindrical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:1749-1849)
 * = EmpCylSL::accumulated_eval (exputil/EmpCylSL.cc:5256-5410) + accumulated_dens_eval
 * (:5413-5502) for the current coefficient set; dens = densC, densS tables
 * [2][mmax+1][norder][numx+1][numy+1] of compute_eof_grid (:1507-1534).  Same coord / out as
 * exp_amd_sph_fields.                                                                       */
int  exp_amd_cyl_set_density(exp_amd_force *f, const double *dens);
int  exp_amd_cyl_fields(exp_amd_force *f, size_t n, const double *c1, const double *c2,
                        const double *c3, int coord, double *out /* [n][9] */);

/* ---- the basis functions themselves (pyEXP getBasis / orthoCheck / getMass) ------------------------
 * SphericalSL::getBasis (expui/BiorthBasis.cc:960-993; pyEXP/BasisWrappers.cc:2142): SLGridSph::get_pot /
 * get_dens / get_force (exputil/SLGridMP2.cc:872-989) of every (l, n) at n host radii r[] -- taken as they
 * are, not divided by `scale` --; out[3][lmax+1][nmax][n] = potential, density, radial force (= -get_force).
 * Needs exp_amd_sph_set_density.                                                                       */
int  exp_amd_sph_basis(exp_amd_force *f, size_t n, const double *r, double *out);
/* Mass of the particles of `c` inside the expansion window rmin <= r <= rmax, r = sqrt(r^2) + dsmall: what
 * Spherical::accumulate adds to totalMass (expui/BiorthBasis.cc:596-607) and BiorthBasis::getMass returns
 * (expui/BiorthBasis.H:189; pyEXP/BasisWrappers.cc:1729).                                              */
int  exp_amd_sph_window_mass(exp_amd_force *f, exp_amd_comp *c, double *mass);
/* Cylindrical::getBasis (expui/BiorthBasis.cc:1930-1974; pyEXP/BasisWrappers.cc:1811): EmpCylSL::get_all(m, n,
 * R, z, phi = 0) (exputil/EmpCylSL.cc:5635-5800) of every (m, n) at n host points -- the cosine tables'
 * bilinear blend on the grid, the monopole of cylmass beyond the table radius;
 * out[4][mmax+1][nmax][n] = potential, density, radial force, vertical force.  Needs exp_amd_cyl_set_density. */
int  exp_amd_cyl_basis(exp_amd_force *f, size_t n, const double *R, const double *z, double *out);
/* EmpCylSL::orthoCheck (exputil/EmpCylSL.cc:7199-7260) behind Cylindrical.orthoCheck (pyEXP/BasisWrappers.cc:
 * 1854): the pot x dens overlap integrals on the table grid; out[mmax+1][nmax][nmax].                    */
int  exp_amd_cyl_orthocheck(exp_amd_force *f, double *out);

/* Sub-sample covariance of the cylindrical coefficients (pyEXP: Cylindrical::enableCoefCovariance,
 * getCoefCovariance, getCovarSamples; the `covar` branch of EmpCylSL::accumulate, exputil/EmpCylSL.cc:
 * 4049-4146, :4554-4575, :4974-5015).  accumulate files every particle of `c` that lies on the grid
 * under sub-sample seq % sampT, seq[] (caller order; NULL: the caller index) being the `seq` argument
 * of EmpCylSL::accumulate; get returns counts[sampT], masses[sampT], VC[sampT][mmax+1][nmax][2] and
 * MV[sampT][mmax+1][nmax][nmax][2] (re, im).                                                      */
int  exp_amd_cyl_cov_enable(exp_amd_force *f, int sampT);
int  exp_amd_cyl_cov_reset(exp_amd_force *f);
int  exp_amd_cyl_cov_accumulate(exp_amd_force *f, exp_amd_comp *c, const uint32_t *seq, long long *on_grid);
int  exp_amd_cyl_cov_get(exp_amd_force *f, long long *counts, double *masses, double *vc, double *mv);

/* ---- fused step ------------------------------------------------------------------------
 * One multistep=0 KDK step of a single self-gravitating component
 * (src/step.cc:271-323): kick dt/2, drift dt, coefficients, zero + force, kick dt/2.
 * Same results as the unfused sequence of calls above; fewer passes over HBM: kick and
 * drift are applied inside the cell-sort passes; the closing half-kick is deferred (applied,
 * as its own rounding step, by the next fused step's scatter pass, or before any other call
 * reads or changes the component); the force pass also records where each particle will be after the NEXT
 * call's kick+drift, so that consecutive calls with the same dt skip the key pass.  Any
 * other call on the component in between (upload, kick, drift, set_center, zero_acc,
 * another force, a different dt) discards that record.                                */
int  exp_amd_step_kdk(exp_amd_force *f, exp_amd_comp *c, double dt);
/* `nsteps` such steps (the do_step loop of src/expand.cc:423-470 at multistep 0 for one component).  Once
 * the steps are in their steady state -- same dt, same centre, the force pass of each step has written the
 * sort keys of the next -- PAIRS of steps are captured once in a HIP graph on the context's stream (the
 * host-side state of a step alternates with period two) and replayed, the RCCL all-reduce of
 * exp_amd_comm_init_rank included: bit-identical to nsteps calls of exp_amd_step_kdk, without the ~12
 * launch gaps per step (3-4 % of a step at 1.25e7 particles per GPU).  A host all-reduce callback, the
 * profiler (exp_amd_profile_enable), the split step or EXP_AMD_STEP_GRAPH=0 keep every step eager.      */
int  exp_amd_step_kdk_n(exp_amd_force *f, exp_amd_comp *c, double dt, int nsteps);

/* ---- step loop ---------------------------------------------------------------------------
 * do_step (src/step.cc:67-325) and begin_run's initial expansion (src/begin.cc:80-129) over a
 * set of components, their self-gravity force methods and pairwise interactions
 * (ComponentContainer, src/ComponentContainer.cc:580-917, :1173-1226): the C++ host
 * orchestration of the path.  dynfrac = {dynfracD, dynfracV, dynfracS, dynfracA, dynfracP}
 * (NULL -> src/global.cc:76-80 defaults).                                               */
typedef struct exp_amd_sim exp_amd_sim;
int  exp_amd_sim_create(exp_amd_ctx *ctx, int multistep, double dtime, const double dynfrac[5],
                        int shiftlevl, exp_amd_sim **out);
void exp_amd_sim_destroy(exp_amd_sim *s);
int  exp_amd_sim_add_component(exp_amd_sim *s, exp_amd_comp *c, exp_amd_force *f, int *index);
int  exp_amd_sim_add_interaction(exp_amd_sim *s, int source, int target);
/* Give component `index` an orientation estimator (the EJ keys of Component, src/Component.cc:1323-
 * 1370): whenever level `centerlevl` (< 0: multistep/2, src/ComponentContainer.cc:42-45) is active,
 * the force evaluation first sets the component's expansion centre to the estimator's current
 * centre (not with dryrun != 0; Component::fix_positions :3357, :3569-3582) and then lets the
 * estimator take in the present state (ComponentContainer::fix_positions :1386-1389).  With the AXIS
 * flag the body rotation is handed to the component too (used by the cylindrical method).  The
 * sim does not own the estimator.                               */
int  exp_amd_sim_set_orient(exp_amd_sim *s, int index, exp_amd_orient *o, int dryrun, int centerlevl);
/* The global `restart` (src/global.cc): the estimators take in the state of the first force
 * evaluation too, where a fresh run waits for potentials (src/ComponentContainer.cc:1386).   */
int  exp_amd_sim_set_restart(exp_amd_sim *s, int on);
/* The global "eqmotion" (src/global.cc:54): with 0, incr_position and incr_velocity return at once (src/incpos.cc:75,
 * src/incvel.cc:93) -- the driver's steps then evaluate expansions, forces and level proposals as the time goes on, and move
 * nothing (a fixed-particle run).  Default 1.                                                                          */
int  exp_amd_sim_set_eqmotion(exp_amd_sim *s, int on);
/* The component key "ctr_name" (Component::c0: find_ctr_component src/Component.cc:284-310, fix_positions :3584-3587): component
 * `index` takes the expansion centre of component `source` at every centre update of the driver (after its own estimator's,
 * which it overrides); components are visited in the order they were added, as the reference visits its list.  source < 0
 * switches it off.                                                                                                       */
int  exp_amd_sim_set_center_from(exp_amd_sim *s, int index, int source);
/* The adiabatic turn-on / turn-off of component `index` (its keys ton, toff, twid; src/Component.cc:1040-1055): the driver
 * evaluates Component::Adiabatic() at its tnow before every accumulation (the time at the START of the sub-step, as
 * do_step has it, src/step.cc:126-160) and before every level-change differencing (the time at its end) and hands it to
 * the component's force method (exp_amd_force_set_mass_scale).  exp_amd_sim_set_time sets tnow (a restart).          */
int  exp_amd_sim_set_adiabatic(exp_amd_sim *s, int index, double ton, double toff, double twid);
int  exp_amd_sim_set_time(exp_amd_sim *s, double tnow);
int  exp_amd_sim_init(exp_amd_sim *s);
int  exp_amd_sim_step(exp_amd_sim *s, int nsteps);
double exp_amd_sim_time(const exp_amd_sim *s);
long long exp_amd_sim_last_switches(const exp_amd_sim *s);   /* level changes of the last adjustment */
long long exp_amd_sim_step_switches(const exp_amd_sim *s);   /* ... summed over the last exp_amd_sim_step call */

/* Host-only: out[bin[i]] += val[i] (float accumulator, double addend, the particles in the order given; bins outside
 * [0, nbins) are skipped) -- the arithmetic of FieldGenerator::histogram2d / histogram1d / histo1dlog
 * (expui/FieldGenerator.cc:776-1009), whose float sums depend on the order of the particles.              */
int  exp_amd_host_binsum_f32(long long n, const int *bin, const double *val, int nbins, float *out);

/* Host-only: n packed particle records of a PSP file (exputil/Particle.cc:333-388; PParticle::read,
 * include/ParticleReader.H:276-315), `rec_size` bytes apart -- [unsigned long indx]? real mass, pos[3], vel[3], pot;
 * int iattrib[niatr]; real dattrib[ndatr], real = float (r_size 4) or double (8) -- into separate arrays, reals
 * widened to double (what a reader hands to exp_amd_comp_upload).                                              */
int  exp_amd_host_psp_unpack(long long n, const void *rec, long long rec_size, int r_size, int indexed, int niatr,
                             int ndatr, unsigned long long *indx, double *mass, double *pos, double *vel, double *pot,
                             int *iattrib, double *dattrib);

/* Timing of the last fused step's dominant kernels (ms, HIP events on the context
 * stream); names are static strings.  Used by bench.py for the roofline figure.    */
int  exp_amd_profile_enable(exp_amd_ctx *ctx, int on);
int  exp_amd_profile_get(exp_amd_ctx *ctx, int idx, const char **name, double *ms_total,
                         long long *launches);
int  exp_amd_profile_reset(exp_amd_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
