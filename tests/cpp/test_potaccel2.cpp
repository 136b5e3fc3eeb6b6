// Two components through the C++ adaptor (include/exp_amd_potaccel.hpp), no Python in the process: a sphereSL halo and a
// cylinder disk, each with its self force and each acting on the other -- begin_run (src/begin.cc:80-129) and one
// block-multistep master step (do_step, src/step.cc:98-269) written call for call as the reference's loops make them:
// ComponentContainer::compute_expansion (src/ComponentContainer.cc:1173-1226), ::compute_potential with its interaction
// list -- SetExternal(); set_multistep_level(mlevel); get_acceleration_and_potential(other); ClearExternal()
// (:698-716, :785-822) --, adjust_multistep_level (src/multistep.cc:344-627), multistep_reset (:1241).
//
// The fixture tests/golden/adaptor_case2.bin (tests/golden/make_adaptor_case.py) holds the inputs and, per scenario, the
// results of oracle/nbody_oracle.c.  Scenario 0 is the plain run; the others switch on the keys that default to off --
// rtrunc / com0 (Component::freeze), ton / toff / twid (Component::Adiabatic), FIX_L0, mlim, self_consistent: false.
//
//   test_potaccel2 <path to adaptor_case2.bin>          exit code 0 = all checks passed
#include "potaccel_test_util.hpp"

#include <memory>

struct Opts {                          // one record of make_adaptor_case.py::_opt_record
  bool has_rtrunc; double rtrunc, com0[3];
  bool adiabatic; double ton, toff, twid;
  bool self_consistent, fix_l0; int mlim;
};
static Opts rd_opts(std::ifstream &f)
{
  double v[12];
  f.read(reinterpret_cast<char *>(v), sizeof(v));
  return Opts{v[0] != 0.0, v[1], {v[2], v[3], v[4]}, v[5] != 0.0, v[6], v[7], v[8], v[9] != 0.0, v[10] != 0.0, (int)v[11]};
}
struct Expected {
  std::vector<std::int32_t> lev0, lev1;
  std::vector<double> acc0, pot0, coef0, pos, vel, acc, pot, coef;
};
static Expected rd_expected(std::ifstream &f, std::size_t n, std::size_t ncoef)
{
  Expected e;
  e.lev0.resize(n); e.lev1.resize(n);
  f.read(reinterpret_cast<char *>(e.lev0.data()), (std::streamsize)(n * 4));
  f.read(reinterpret_cast<char *>(e.lev1.data()), (std::streamsize)(n * 4));
  e.acc0 = rd(f, 3 * n); e.pot0 = rd(f, n); e.coef0 = rd(f, ncoef);
  e.pos = rd(f, 3 * n); e.vel = rd(f, 3 * n); e.acc = rd(f, 3 * n); e.pot = rd(f, n); e.coef = rd(f, ncoef);
  return e;
}

int main(int argc, char **argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: %s adaptor_case2.bin\n", argv[0]); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
  char magic[8];
  f.read(magic, 8);
  if (!f || std::memcmp(magic, "EXPAMD02", 8) != 0) { std::fprintf(stderr, "bad fixture\n"); return 2; }
  std::int32_t hd[12], nn[2];
  f.read(reinterpret_cast<char *>(hd), sizeof(hd));
  f.read(reinterpret_cast<char *>(nn), sizeof(nn));
  const int lmax = hd[0], nmax = hd[1], numr = hd[2], cmap = hd[3], mmax = hd[4], norder = hd[5], numx = hd[6], numy = hd[7],
            cmapr = hd[8], cmapz = hd[9], ms = hd[10], nscen = hd[11];
  const std::size_t nh = (std::size_t)nn[0], nd = (std::size_t)nn[1];
  double sp[6], cp[8], td[6];
  f.read(reinterpret_cast<char *>(sp), sizeof(sp));
  f.read(reinterpret_cast<char *>(cp), sizeof(cp));
  f.read(reinterpret_cast<char *>(td), sizeof(td));
  const double dtime = td[0];
  const double *dyn = td + 1;
  const std::size_t ncs = (std::size_t)(lmax + 1) * (lmax + 1) * nmax, ncc = (std::size_t)2 * (mmax + 1) * norder;
  auto xi = rd(f, numr), p0 = rd(f, numr), ev = rd(f, (std::size_t)(lmax + 1) * nmax), ef = rd(f, (std::size_t)(lmax + 1) * nmax * numr),
       tab = rd(f, (std::size_t)6 * (mmax + 1) * norder * (numx + 1) * (numy + 1)), hm = rd(f, nh), hp = rd(f, 3 * nh),
       hv = rd(f, 3 * nh), dm = rd(f, nd), dp = rd(f, 3 * nd), dv = rd(f, 3 * nd);
  if (!f) { std::fprintf(stderr, "short fixture\n"); return 2; }

  auto fresh = [](const std::vector<double> &m, const std::vector<double> &p, const std::vector<double> &v) {
    auto c = std::make_unique<VecComponent>(m.size());
    for (std::size_t i = 0; i < m.size(); i++) {
      c->m[i] = m[i];
      c->x[i] = p[3 * i]; c->y[i] = p[3 * i + 1]; c->z[i] = p[3 * i + 2];
      c->vx[i] = v[3 * i]; c->vy[i] = v[3 * i + 1]; c->vz[i] = v[3 * i + 2];
    }
    return c;
  };
  const int Mstep = 1 << ms;
  std::vector<int> mintvl(ms + 1), mfirst(Mstep + 1, 0);              // initialize_multistep, src/multistep.cc:630-680
  mintvl[0] = Mstep;
  for (int k = 1; k <= ms; k++) mintvl[k] = mintvl[k - 1] / 2;
  for (int s = 0; s <= Mstep; s++)
    for (int M = 0; M <= ms; M++)
      if (s == 0 || s % (1 << (ms - M)) == 0) { mfirst[s] = M; break; }

  try {
    exp_amd::Context ctx(0);
    for (int sc = 0; sc < nscen; sc++) {
      const Opts oh = rd_opts(f), od = rd_opts(f);
      long long nsw_ref[2];
      double cylmass_ref;
      f.read(reinterpret_cast<char *>(nsw_ref), sizeof(nsw_ref));
      f.read(reinterpret_cast<char *>(&cylmass_ref), sizeof(double));
      const Expected eh = rd_expected(f, nh, ncs), ed = rd_expected(f, nd, ncc);
      if (!f) { std::fprintf(stderr, "short fixture (scenario %d)\n", sc); return 2; }
      std::printf("---- scenario %d: halo rtrunc %s, FIX_L0 %d, self_consistent %d | disk rtrunc %s, adiabatic %d, mlim %d, self_consistent %d\n",
                  sc, oh.has_rtrunc ? "set" : "-", (int)oh.fix_l0, (int)oh.self_consistent, od.has_rtrunc ? "set" : "-",
                  (int)od.adiabatic, od.mlim, (int)od.self_consistent);

      double tnow = 0.0;                                              // EXP's global
      auto halo = fresh(hm, hp, hv), disk = fresh(dm, dp, dv);
      VecComponent *comp[2] = {halo.get(), disk.get()};
      const Opts *opt[2] = {&oh, &od};
      for (int k = 0; k < 2; k++) {
        comp[k]->tnow_ = &tnow;
        if (opt[k]->has_rtrunc) { comp[k]->rtrunc_ = opt[k]->rtrunc; for (int j = 0; j < 3; j++) comp[k]->com0_[j] = opt[k]->com0[j]; }
        if (opt[k]->adiabatic) { comp[k]->adiabatic_ = true; comp[k]->ton_ = opt[k]->ton; comp[k]->toff_ = opt[k]->toff; comp[k]->twid_ = opt[k]->twid; }
      }
      exp_amd::Mirror mirror(ctx);
      exp_amd_sph_config scfg{lmax, nmax, numr, cmap, sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], 0, 0, 0, 0, 0, ms};
      exp_amd_cyl_config ccfg{mmax, norder, numx, numy, cmapr, cmapz, cp[0], cp[1], cp[2], cp[3], cp[4], cp[5], cp[6], cp[7], 0, ms};
      exp_amd::SphereAMD fh(ctx, mirror, halo.get(), scfg, xi.data(), p0.data(), ev.data(), ef.data());
      exp_amd::CylinderAMD fd(ctx, mirror, disk.get(), ccfg, tab.data());
      exp_amd::PotAccelAMD *force[2] = {&fh, &fd};
      // the keys of the two force methods
      if (oh.fix_l0) fh.set_fix_l0(true);
      if (!oh.self_consistent) fh.set_self_consistent(false);
      if (od.mlim >= 0) fd.set_mlim(od.mlim);
      if (!od.self_consistent) fd.set_self_consistent(false);
      // the interaction list: each component's force acts on the other (src/ComponentContainer.cc:785-853)
      const int inter[2][2] = {{0, 1}, {1, 0}};

      // ComponentContainer::compute_expansion(M) (:1173-1226)
      auto compute_expansion = [&](int M) {
        for (int k = 0; k < 2; k++) { force[k]->set_multistep_level((unsigned)M); force[k]->determine_coefficients(comp[k]); }
      };
      // ComponentContainer::compute_potential(mlevel) (:580-917): zero, self forces, interactions
      auto compute_potential = [&](int mlevel, int mdrft) {
        for (int k = 0; k < 2; k++) exp_amd::zero_acceleration(ctx, mirror, comp[k], mlevel);
        for (int k = 0; k < 2; k++) {
          force[k]->set_multistep_level((unsigned)mlevel);                       // :698
          force[k]->set_mdrft(mdrft);
          force[k]->get_acceleration_and_potential(comp[k]);                    // :714
        }
        for (auto &pr : inter) {
          exp_amd::PotAccelAMD *src = force[pr[0]];
          src->SetExternal();                                                   // :817
          src->set_multistep_level((unsigned)mlevel);                           // :819
          src->get_acceleration_and_potential(comp[pr[1]]);                     // :820
          src->ClearExternal();                                                 // :822
        }
      };
      long long nswitch[2] = {0, 0};
      // adjust_multistep_level (src/multistep.cc:344-627): _begin for every component, the sweep, _finish
      auto adjust = [&](int mdrft, bool first_step, bool count) {
        for (int k = 0; k < 2; k++) force[k]->multistep_update_begin();
        for (int k = 0; k < 2; k++) {
          const long long u = force[k]->multistep_update_device(comp[k], dtime, dyn, 0, mdrft, first_step);
          if (count) nswitch[k] += u;
        }
        for (int k = 0; k < 2; k++) force[k]->multistep_update_finish();
      };
      // ---- begin_run (src/begin.cc:80-129) ------------------------------------------------------------------------
      for (auto fp : force) fp->set_initializing(true);                          // `initializing = true` (:80)
      for (auto fp : force) fp->multistep_reset();
      for (int M = 0; M <= ms; M++) compute_expansion(M);
      compute_potential(0, 0);
      adjust(0, true, false);
      for (auto fp : force) fp->multistep_reset();
      for (int M = 0; M <= ms; M++) compute_expansion(M);
      compute_potential(0, 0);
      for (auto fp : force) fp->set_initializing(false);                         // (:129)
      for (int k = 0; k < 2; k++) mirror.download(comp[k]);
      {
        const Expected *e[2] = {&eh, &ed};
        const char *nm[2] = {"halo", "disk"};
        for (int k = 0; k < 2; k++) {
          char what[96];
          int bad = 0;
          for (std::size_t i = 0; i < comp[k]->level.size(); i++) bad += comp[k]->level[i] != e[k]->lev0[i];
          std::snprintf(what, sizeof what, "begin_run %s: levels (mismatches)", nm[k]);
          expect(what, (double)bad, 0.0);
          std::snprintf(what, sizeof what, "begin_run %s: accelerations", nm[k]);
          expect(what, maxdiff3(comp[k]->ax, comp[k]->ay, comp[k]->az, e[k]->acc0), 1e-9 * maxabs(e[k]->acc0));
          std::snprintf(what, sizeof what, "begin_run %s: potential", nm[k]);
          expect(what, maxdiff(comp[k]->pot, e[k]->pot0), 1e-9 * maxabs(e[k]->pot0));
          std::snprintf(what, sizeof what, "begin_run %s: combined coefficients", nm[k]);
          expect(what, maxdiff(force[k]->get_coefs(), e[k]->coef0), 1e-10 * maxabs(e[k]->coef0));
        }
      }
      // ---- do_step (src/step.cc:98-269) -------------------------------------------------------------------------------
      for (auto fp : force) fp->multistep_reset();                               // :84
      const double dts = dtime / Mstep;
      for (int mstep = 0; mstep < Mstep; mstep++) {
        for (int M = mfirst[mstep]; M <= ms; M++) {
          const double DT = dts * mintvl[M];
          for (int k = 0; k < 2; k++) exp_amd::incr_velocity(ctx, mirror, comp[k], 0.5 * DT, M);
          for (int k = 0; k < 2; k++) exp_amd::incr_position(ctx, mirror, comp[k], DT, M);
          compute_expansion(M);
        }
        tnow += dts;                                                             // :163
        const int mdrft = mstep + 1;
        compute_potential(mfirst[mstep], mdrft);
        for (int M = mfirst[mdrft]; M <= ms; M++)
          for (int k = 0; k < 2; k++) exp_amd::incr_velocity(ctx, mirror, comp[k], 0.5 * dts * mintvl[M], M);
        adjust(mdrft, mstep == 0, true);
      }
      for (int k = 0; k < 2; k++) mirror.download(comp[k]);
      {
        const Expected *e[2] = {&eh, &ed};
        const char *nm[2] = {"halo", "disk"};
        for (int k = 0; k < 2; k++) {
          char what[96];
          int bad = 0;
          for (std::size_t i = 0; i < comp[k]->level.size(); i++) bad += comp[k]->level[i] != e[k]->lev1[i];
          std::snprintf(what, sizeof what, "master step %s: levels (mismatches)", nm[k]);
          expect(what, (double)bad, 0.0);
          std::snprintf(what, sizeof what, "master step %s: level changes", nm[k]);
          expect(what, std::fabs((double)(nswitch[k] - nsw_ref[k])), 0.0);
          std::snprintf(what, sizeof what, "master step %s: positions", nm[k]);
          expect(what, maxdiff3(comp[k]->x, comp[k]->y, comp[k]->z, e[k]->pos), 1e-11);
          std::snprintf(what, sizeof what, "master step %s: velocities", nm[k]);
          expect(what, maxdiff3(comp[k]->vx, comp[k]->vy, comp[k]->vz, e[k]->vel), 1e-9 * maxabs(e[k]->vel));
          std::snprintf(what, sizeof what, "master step %s: accelerations", nm[k]);
          expect(what, maxdiff3(comp[k]->ax, comp[k]->ay, comp[k]->az, e[k]->acc), 1e-8 * maxabs(e[k]->acc));
          std::snprintf(what, sizeof what, "master step %s: potential", nm[k]);
          expect(what, maxdiff(comp[k]->pot, e[k]->pot), 1e-8 * maxabs(e[k]->pot));
          std::snprintf(what, sizeof what, "master step %s: combined coefficients", nm[k]);
          expect(what, maxdiff(force[k]->get_coefs(), e[k]->coef), 1e-10 * maxabs(e[k]->coef));
        }
        expect("master step disk: cylmass", std::fabs(fd.cylmass() - cylmass_ref), 1e-12 * std::fmax(std::fabs(cylmass_ref), 1e-300) + 1e-300);
        if (!oh.self_consistent) expect("halo: coefficients held fixed (coefs_frozen)", fh.coefs_frozen() ? 0.0 : 1.0, 0.0);
        if (!od.self_consistent) expect("disk: coefficients held fixed (coefs_frozen)", fd.coefs_frozen() ? 0.0 : 1.0, 0.0);
        if (od.mlim >= 0) {
          // harmonics above mlim read back as zero (exputil/EmpCylSL.cc:5602: get_pot never fills them)
          const std::vector<double> c = fd.get_coefs();
          double worst = 0.0;
          const std::size_t half = (std::size_t)(mmax + 1) * norder;
          for (int m = od.mlim + 1; m <= mmax; m++)
            for (int n = 0; n < norder; n++)
              worst = std::fmax(worst, std::fmax(std::fabs(c[(std::size_t)m * norder + n]), std::fabs(c[half + (std::size_t)m * norder + n])));
          expect("disk: coefficients above mlim are zero", worst, 0.0);
        }
      }
    }
  } catch (const exp_amd::Error &e) {
    std::fprintf(stderr, "exp_amd::Error: %s\n", e.what());
    return 3;
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
