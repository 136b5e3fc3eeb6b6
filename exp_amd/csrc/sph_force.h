// SphForce: the spherical force object behind exp_amd_force (shared by sph.hip and sph_fields.hip).
#pragma once
#include "sph_kernels.h"
#include "force.h"
#include <random>

struct SphForce : exp_amd_force {
  exp_amd_sph_config cfg{};
  SphDev dev{};
  DevBuf<double> d_xi, d_p0, d_E, d_lc;
  DevBuf<double> d_litef;           // raw eigenfunctions at the first and last force stencil (SphDev::lit_ef)
  DevBuf<uint32_t> d_litlist;       // [1 + capacity]: slots left to the literal pass (SphDev::lit_list), cmap 2 only
  bool lit_on = false;
  // FIX_L0 (src/SphericalBasis.cc:34, :1689-1694): the monopole row of the coefficient set is saved at the first force
  // evaluation and copied back into the active set at every later one (exp_amd_sph_set_fix_l0)
  bool fix_l0 = false, have_c0 = false;
  DevBuf<double> d_c0;              // [nmax]
  DevBuf<unsigned char> d_gen_slot; // (l, m | sine flag) of the projected table's slots (SphDev::gen_slot)
  DevBuf<double> d_gen_ac, d_gen_e; // run-time recurrence constants of the any-order kernels (SphDev::gen_ac, gen_e)
  bool generic = false;             // lmax > SPH_MAX_L (or EXP_AMD_SPH_GENERIC=1): every per-particle pass through sph_gen.hip
  DevBuf<double> d_W, d_part, d_G, d_T4;
  DevBuf<double> d_coef_app;        // [2][ncoef]: the coefficient set of the last append step (whose accelerations the placing pass
                                    // does not store: sph.hip, sph_app_reeval) and scratch for the set in force meanwhile
  DevBuf<int> d_rowmap;
  DevBuf<double> d_tscale, d_wscale;   // 1/s(l,m) per table slot / per coefficient row
  DevBuf<double> d_Wd, d_differ;    // multistep differencing: moments / coefficients per level
  DevBuf<double> d_stage;            // staged differencing of few movers: values [mover][nrows][2] ...
  DevBuf<int> d_stage_keys;          // ... and their two W offsets (k_sph_mstep_update<L, true>, k_mstep_apply)
  DevBuf<double> d_ev, d_d0, d_Gd;  // field evaluation (pyEXP getFields): ev[l][n], d0[numr], Gd[numr][rows]
  void *cov = nullptr;              // sub-sample covariance state (sph_cov.hip), analysis only
  DevBuf<uint32_t> d_work;          // slow-path work list of the force pass + count (last slot)
  size_t work_cap = 0;
  double term_max = 0.0;            // max |P0| x max |Ph(l,m)| x 4 pi x 2: bound of one unit-mass contribution
  int work_flip = 0;                // which of the two work-list counters the next fast pass counts into
  // ... the same for the staged evaluation of ANOTHER component's particles (its special lanes): a list of its own, because
  // that launch may run on the target's stream beside this force's self evaluation
  DevBuf<uint32_t> d_xwork;
  size_t xwork_cap = 0;
  int xwork_flip = 0;
  // PotAccel::used of a multistep run: the counts of every level accumulated while tnow == resetT,
  // i.e. during the first sub-step of a master step (src/SphericalBasis.cc:796, :860-862, :1004-1010);
  // d_used[0] is what Used() reports, d_used[1] takes the counts of the later accumulations
  bool used_open = true;
  bool wd_clean = false;            // ... and of d_Wd (multistep_update)
  // the tiled direct kernels (k_sph_acc_thin / k_sph_diff_thin) keep the harmonics and table blends of at least four
  // particles in LDS: where that does not fit (a small lmax with a very large nmax) the moment path takes the work
  bool thin_lds_ok() const
  {
    const size_t nrows = (size_t)(cfg.lmax + 1) * (cfg.lmax + 1), lsn = (size_t)(cfg.lmax + 1) * cfg.nmax;
    return ((((size_t)4 * nrows + 1) & ~(size_t)1) + (size_t)4 * lsn) * sizeof(double) <= 120 * 1024;
  }
  bool adv_owed = false;            // substep_expansion: the advance of the active range is left to k_sph_acc_thin
  double adv_dt_min = 0.0;
  bool part_clean = false;          // d_part is all zero (what the thin accumulation adds to; its summing kernels keep it so)
  bool w_clean = false;             // every per-level moment buffer of d_W is zero (substep_expansion keeps it so)
  // "ssfrac" (src/SphericalBasis.cc:149-152, :437-440, :459-460, :472-473): the coefficients from a sub-sample of the level
  // list.  Thread id of `nthrds` takes the entries [n id / nthrds, floor(ssfrac n (id + 1) / nthrds)) of the list -- the
  // END INDEX is scaled, not the slice length, so with several threads the later slices come out short or empty, as in the
  // reference -- and a particle's mass is divided by ssfrac.  The level list is the caller's particle order here (the
  // reference's is the iteration order of its particle map).  exp_amd_sph_set_subset; single-level forces only.
  bool subset_on = false;
  double ssfrac = 1.0;
  int ss_nthrds = 1;
  exp_amd_comp *ss_comp = nullptr;  // the sub-sample as a component of its own (rebuilt at every accumulation)
  DevBuf<double> d_ss[4];           // gathered m, x, y, z
  DevBuf<uint32_t> d_ss_prefix;     // entries of the compacted set in front of each thread's slice
  size_t ss_cap = 0;
  int determine_coefficients_subset(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift);
  // The NOISE keys (src/SphericalBasis.cc:79-81, :135-147, :355, :395, :2108-2210): every get_acceleration_and_potential call
  // REPLACES the coefficient set by draws from a noise model -- sqrt(|rmsC(l,n) - meanC[n]^2| factorial(l,m) / noiseN) times a
  // standard normal deviate, plus meanC[n] on the l = 0 row -- from one std::mt19937 / std::normal_distribution pair seeded
  // once with seedN (the same standard-library objects the reference holds, :340-341 of its header).  A self call of a
  // multistep force draws too and is then overwritten by compute_multistep_coefficients (:1680-1685), so nothing is uploaded
  // for it.  exp_amd_sph_set_noise; meanC / rmsC come from the host (SphericalBasis::compute_rms_coefs, exp_amd/slgrid.py).
  bool noise_on = false, noise_setup = true;
  std::vector<double> n_mean, n_rms, n_host;
  double noiseN = 1.0e-6;
  unsigned seedN = 0;
  std::mt19937 rgen;
  std::normal_distribution<> nrand;
  long long noise_calls = 0;
  int update_noise(bool self_call);
  int step_parity() const override { return work_flip; }
  bool step_graph_ok() const override { return !noise_on && !subset_on; }
  int multistep_reset() override
  {
    HIP_TRY(ctx, hipMemsetAsync(d_used.p, 0, sizeof(unsigned long long), ctx->stream));
    used_open = true;
    return EXP_AMD_OK;
  }

  int determine_coefficients(exp_amd_comp *c, bool advance, double dt_kick, double dt_drift,
                             bool have_keys = false) override;
  int accelerate(exp_amd_comp *t, int external, bool assign, double dt_kick, double nk_dtk = 0.0,
                 double nk_dtd = 0.0, bool *prekey_done = nullptr, bool defer_kick = false) override;
  int multistep_update(exp_amd_comp *c, int first, int mfirst_mdrft) override;
  bool prekey_launcher(exp_amd_comp *c, ka_launch_fn *fn, void **self) override;
  SphDev ka_S;                      // ... the frame its key function was given (the component's centre at the sweep)
  int substep_expansion(exp_amd_comp *c, int lo, double dt_min, int mdrft_combine = -1, int phase = 0) override;
  // (the spherical basis has few cells, numr - 1: a level stays worth sorting down to a few particles
  // per cell, and per-particle atomics on so few addresses contend)
  // (break-even of the un-sorted treatment -- staged per-particle accumulation + gather forces, no sort -- against the
  // cell-sorted one, measured at S6 and S10 with tools/dbg/acc_staged.py: about 5e6 / (values per particle))
  long long sparse_threshold() const override
  {
    const long long a = 4LL * (cfg.numr - 1), b = 5000000LL / (2LL * (cfg.lmax + 1) * (cfg.lmax + 1));
    return a > b ? a : b;
  }
  int resort(exp_amd_comp *c, int first = 0) override;
  int fused_step_split(exp_amd_comp *c, double dt, bool have_keys, bool *handled) override;
  int fused_step_append(exp_amd_comp *c, double dt, bool have_keys, bool *handled) override;
  void release() override;
};


void expamd_sph_cov_release(SphForce *f);
int sph_project(SphForce *f);     // coefficients -> G / T4 tables (no-op when they are current)
