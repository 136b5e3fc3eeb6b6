// CylForce: the cylindrical force object behind exp_amd_force (shared by cyl.hip and cyl_fields.hip).
#pragma once
#include "cyl_dev.h"
#include "force.h"

struct CylForce : exp_amd_force {
  exp_amd_cyl_config cfg{};
  CylDev dev{};
  DevBuf<double> d_tab, d_Wn, d_TF;
  bool tab_twin = false;            // the three sine tables equal the three cosine tables bit for bit (m >= 1)
  DevBuf<double> d_cpart;           // stage-1 sums of the contraction: [level][CYL_CSEG][ncoef]
  DevBuf<uint32_t> d_work;          // work list of the force pass' tail launch (lanes beyond 0.75 of the table radius)
  size_t work_cap = 0;              // ... in waves; the two counters behind it are used alternately
  int work_flip = 0;
  int step_parity() const override { return work_flip; }
  int mlim = -1;                    // the "mlim" key (exp_amd_cyl_set_mlim): harmonics above it are dropped; < 0: none
  bool generic = false;             // mmax > CYL_MAX_M (or EXP_AMD_CYL_GENERIC=1): the run-time-order kernels throughout
  // k_cyl_acc_thin keeps the table blends of at least four particles in LDS (one or two sets of (mmax+1) nmax values
  // each): beyond 120 KB the moment path takes the work
  bool thin_lds_ok() const { return (size_t)4 * 2 * (cfg.mmax + 1) * cfg.nmax * sizeof(double) <= 120 * 1024; }
  bool adv_owed = false;            // substep_expansion: the advance of the active range is left to k_cyl_acc_thin
  double adv_dt_min = 0.0;
  bool cpart_clean = false;         // ... all zero (what k_cyl_acc_thin adds to; its summing kernels keep them so)
  DevBuf<double> d_tabT;            // node-major copy tabT[node][kind][m][n] for the thin path (made on first use)
  int tabT_nk = 0;                  // kinds it holds: 3 (sine tables == cosine tables) or 6
  int ensure_tabT();
  DevBuf<double> d_Wnd, d_differ;   // multistep differencing
  DevBuf<double> d_dens;            // densC / densS tables (field evaluation only)
  // sub-sample covariance (pyEXP pcavar, analysis only): node moments U[T][node][ntrig], cell
  // moments Q[T][cell][10], counts / masses [T], results
  int cov_T = 0;
  DevBuf<double> cov_U, cov_Q, cov_mass, cov_vc, cov_mv;
  DevBuf<unsigned long long> cov_cnt, cov_used;
  DevBuf<uint32_t> cov_seq;
  size_t cov_seq_cap = 0;
  DevBuf<double> d_mass;            // {cylmass, used}: in-cut mass / count of the current master step
  DevBuf<double> d_tailpart;        // [CYL_TAILS][2]: the same tallies of ONE accumulation launch, spread over slots
  bool mass_open = true;            // still within the first sub-step (tnow == resetT)
  bool wn_clean = false;            // every per-level moment buffer of d_Wn is zero (substep_expansion's contraction keeps it so)
  bool wnd_clean = false;           // ... and d_Wnd, d_differ's tails (multistep_update)
  bool tails_clean = false;         // the {mass, count} tails of all expcoefN sets are zero (substep_expansion keeps them so)
  size_t nnode = 0;

  int determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift,
                             bool have_keys = false) override;
  int accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick, double nk_dtk = 0.0,
                 double nk_dtd = 0.0, bool *prekey_done = nullptr, bool defer_kick = false) override;
  int multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft) override;
  bool prekey_launcher(exp_amd_comp *c, ka_launch_fn *fn, void **self) override;
  CylDev ka_C;
  int substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine = -1, int phase = 0) override;
  long long sparse_threshold() const override { return 3000000LL / (4 * dev.ntrig); }
  int resort(exp_amd_comp *c, int first = 0) override;
  int multistep_reset() override
  {
    // Cylinder::multistep_reset: used = 0, cylmass = 0, resetT = tnow (src/Cylinder.cc:1209-1216)
    HIP_TRY(ctx, hipMemsetAsync(d_mass.p, 0, 2 * sizeof(double), ctx->stream));
    mass_open = true;
    return EXP_AMD_OK;
  }
  int sort(exp_amd_comp *c, bool move_acc, const AdvSpec &adv, int level = -1, bool have_keys = false,
           int level_hi = -1);
  void release() override
  {
    cov_U.release(); cov_Q.release(); cov_mass.release(); cov_vc.release(); cov_mv.release();
    cov_cnt.release(); cov_used.release(); cov_seq.release();
    d_tab.release(); d_Wn.release(); d_TF.release(); d_Wnd.release(); d_differ.release(); d_cpart.release();
    d_tabT.release(); d_work.release();
    d_mass.release();
    d_tailpart.release();
    d_dens.release();
  }
  int get_used(long long *used) override
  {
    double u = 0.0;
    HIP_TRY(ctx, hipMemcpyAsync(&u, d_mass.p + 1, sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *used = (long long)(u + 0.5);
    return EXP_AMD_OK;
  }
};

static inline CylDev cdev_frame(const CylForce *f, const double *center, bool use_rot, const double *rot)
{
  CylDev C = f->dev;
  C.cx = center[0]; C.cy = center[1]; C.cz = center[2];
  C.use_rot = use_rot ? 1 : 0;
  for (int k = 0; k < 9; k++) C.rot[k] = rot[k];
  return C;
}

static inline CylDev cdev_for(const CylForce *f, const exp_amd_comp *c)
{
  return cdev_frame(f, c->center, c->use_rot, c->rot);
}

// Component::freeze of the component whose particles a launch walks (the source of an accumulation, the target of a force)
static inline void cdev_freeze(CylDev &C, const exp_amd_comp *c)
{
  C.frz = expamd_comp_frz(c);
}

// ... for the passes that ADD particle contributions: with the deterministic mode on, the rounding
// grids that keep every partial sum of this component exact (|-4 pi m c_k trig| <= 4 pi |m| x 2 for the
// bilinear weights; the in-cut mass itself)
static inline CylDev cdev_acc(const CylForce *f, const exp_amd_comp *c)
{
  CylDev C = cdev_for(f, c);
  C.detC = expamd_det_constant(f->ctx->deterministic, c->mass_abs_sum * 4.0 * M_PI * 2.0);
  C.detCm = expamd_det_constant(f->ctx->deterministic, c->mass_abs_sum);
  C.umass = c->uniform_mass ? c->mass_value : 0.0;
  C.mscale = f->mass_scale;
  cdev_freeze(C, c);
  return C;
}


#define MMAX_DISPATCH(M, CALL)                                                        \
  switch (M) {                                                                        \
    case 0: CALL(0); break;  case 1: CALL(1); break;  case 2: CALL(2); break;        \
    case 3: CALL(3); break;  case 4: CALL(4); break;  case 5: CALL(5); break;        \
    case 6: CALL(6); break;  case 7: CALL(7); break;  case 8: CALL(8); break;        \
    case 9: CALL(9); break;  case 10: CALL(10); break; case 11: CALL(11); break;     \
    case 12: CALL(12); break;                                                        \
  }
