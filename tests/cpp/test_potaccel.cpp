// Drives the C++ adaptor (include/exp_amd_potaccel.hpp) the way EXP's step loop drives a PotAccel --
// no Python in the process: a multistep=0 KDK step (do_step, src/step.cc:271-323) and begin_run + one
// block-multistep master step (src/begin.cc:80-129, src/step.cc:98-269) of a sphereSL component,
// checked against the oracle's results stored in tests/golden/adaptor_case.bin
// (tests/golden/make_adaptor_case.py).  Build and run: see tests/test_adaptor_gpu.py / Makefile.
//
//   test_potaccel <path to adaptor_case.bin> [file for dump_coefs]     exit code 0 = all checks passed
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "exp_amd_potaccel.hpp"

#include "potaccel_test_util.hpp"

int main(int argc, char **argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: %s adaptor_case.bin\n", argv[0]); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
  char magic[8];
  f.read(magic, 8);
  if (!f || std::memcmp(magic, "EXPAMD01", 8) != 0) { std::fprintf(stderr, "bad fixture\n"); return 2; }
  std::int32_t hd[6];
  f.read(reinterpret_cast<char *>(hd), sizeof(hd));
  const int lmax = hd[0], nmax = hd[1], numr = hd[2], cmap = hd[3], n = hd[4], ms = hd[5];
  double sc[8], dyn[5];
  f.read(reinterpret_cast<char *>(sc), sizeof(sc));
  f.read(reinterpret_cast<char *>(dyn), sizeof(dyn));
  long long nsw_ref = 0, used_ref = 0;
  f.read(reinterpret_cast<char *>(&nsw_ref), sizeof(nsw_ref));
  f.read(reinterpret_cast<char *>(&used_ref), sizeof(used_ref));
  const double rmap = sc[0], scale = sc[1], rmin = sc[2], rmax = sc[3], xmin = sc[4], dxi = sc[5], dt = sc[6], dtime = sc[7];
  const std::size_t ncoef = (std::size_t)(lmax + 1) * (lmax + 1) * nmax;
  auto xi = rd(f, numr), p0 = rd(f, numr), ev = rd(f, (std::size_t)(lmax + 1) * nmax),
       ef = rd(f, (std::size_t)(lmax + 1) * nmax * numr), mass = rd(f, n), pos = rd(f, 3 * (std::size_t)n),
       vel = rd(f, 3 * (std::size_t)n), coef0 = rd(f, ncoef), acc0 = rd(f, 3 * (std::size_t)n), pot0 = rd(f, n),
       spos = rd(f, 3 * (std::size_t)n), svel = rd(f, 3 * (std::size_t)n), sacc = rd(f, 3 * (std::size_t)n),
       spot = rd(f, n), scoef = rd(f, ncoef);
  std::vector<std::int32_t> mlev(n);
  f.read(reinterpret_cast<char *>(mlev.data()), (std::streamsize)(n * sizeof(std::int32_t)));
  auto mpos = rd(f, 3 * (std::size_t)n), mvel = rd(f, 3 * (std::size_t)n), macc = rd(f, 3 * (std::size_t)n), mcoef = rd(f, ncoef);
  if (!f) { std::fprintf(stderr, "short fixture\n"); return 2; }

  auto fresh = [&]() {
    VecComponent c((std::size_t)n);
    for (int i = 0; i < n; i++) {
      c.m[i] = mass[i];
      c.x[i] = pos[3 * i]; c.y[i] = pos[3 * i + 1]; c.z[i] = pos[3 * i + 2];
      c.vx[i] = vel[3 * i]; c.vy[i] = vel[3 * i + 1]; c.vz[i] = vel[3 * i + 2];
    }
    return c;
  };
  try {
    exp_amd::Context ctx(0);
    // ---- multistep = 0: initial field, then do_step's KDK block (src/step.cc:271-323) ---------------
    {
      VecComponent comp = fresh();
      exp_amd::Mirror mirror(ctx);
      exp_amd_sph_config cfg{lmax, nmax, numr, cmap, rmap, scale, rmin, rmax, xmin, dxi, 0, 0, 0, 0, 0, 0};
      exp_amd::SphereAMD force(ctx, mirror, &comp, cfg, xi.data(), p0.data(), ev.data(), ef.data());
      // begin_run: compute_expansion(0); compute_potential(0)
      force.set_multistep_level(0);
      force.determine_coefficients(&comp);
      exp_amd::zero_acceleration(ctx, mirror, &comp, 0);
      force.get_acceleration_and_potential(&comp);
      mirror.download(&comp);
      const double cmax = maxabs(coef0), amax = maxabs(acc0);
      expect("coefficients (begin_run)", maxdiff(force.get_coefs(), coef0), 1e-10 * cmax);
      expect("accelerations (begin_run)", maxdiff3(comp.ax, comp.ay, comp.az, acc0), 1e-9 * amax);
      expect("potential (begin_run)", maxdiff(comp.pot, pot0), 1e-9 * maxabs(pot0));
      // do_step
      exp_amd::incr_velocity(ctx, mirror, &comp, 0.5 * dt);
      exp_amd::incr_position(ctx, mirror, &comp, dt);
      force.set_multistep_level(0);
      force.determine_coefficients(&comp);
      exp_amd::zero_acceleration(ctx, mirror, &comp, 0);
      force.get_acceleration_and_potential(&comp);
      exp_amd::incr_velocity(ctx, mirror, &comp, 0.5 * dt);
      mirror.download(&comp);
      expect("positions after one KDK step", maxdiff3(comp.x, comp.y, comp.z, spos), 1e-12);
      expect("velocities after one KDK step", maxdiff3(comp.vx, comp.vy, comp.vz, svel), 1e-9 * maxabs(svel));
      expect("accelerations after one KDK step", maxdiff3(comp.ax, comp.ay, comp.az, sacc), 1e-9 * maxabs(sacc));
      expect("potential after one KDK step", maxdiff(comp.pot, spot), 1e-9 * maxabs(spot));
      expect("coefficients after one KDK step", maxdiff(force.get_coefs(), scoef), 1e-10 * maxabs(scoef));
      {
        // the run log's sums on the device (OutLog::Run's particle loop) against the same loop over the downloaded state
        const exp_amd::LogSums ls = exp_amd::log_sums(ctx, mirror, &comp);
        double mt = 0.0, ek = 0.0, vc = 0.0;
        for (size_t i = 0; i < comp.m.size(); i++) {
          mt += comp.m[i];
          ek += 0.5 * comp.m[i] * (comp.vx[i] * comp.vx[i] + comp.vy[i] * comp.vy[i] + comp.vz[i] * comp.vz[i]);
          vc += comp.m[i] * (comp.x[i] * comp.ax[i] + comp.y[i] * comp.ay[i] + comp.z[i] * comp.az[i]);
        }
        expect("log sums: mass", std::fabs(ls.v[0] - mt), 1e-12 * mt);
        expect("log sums: kinetic energy", std::fabs(ls.v[10] - ek), 1e-11 * ek);
        expect("log sums: Clausius virial", std::fabs(ls.v[12] - vc), 1e-11 * std::fabs(vc));
        expect("log sums: bodies", std::fabs(ls.v[13] - (double)comp.m.size()), 0.0);
      }
      (void)used_ref;      // (the stored count is that of the INITIAL accumulation; one particle may cross rmax in the step)
      expect("PotAccel::Used()", std::fabs((double)(force.Used() - used_ref)), 1.0);
      // PotAccel::dump_coefs(ostream&) (src/PotAccel.H:224, src/SphericalBasis.cc:1829-1879): one native record;
      // read back here field by field, and by exp_amd.coefs.SphCoefs.readNativeCoefs in tests/test_adaptor_gpu.py
      if (argc > 2) {
        std::ofstream out(argv[2], std::ios::binary);
        force.dump_coefs(out, 0.125);
        force.dump_coefs(out, 0.25);
        out.close();
        std::ifstream in(argv[2], std::ios::binary);
        unsigned int mg = 0, hs = 0;
        in.read(reinterpret_cast<char *>(&mg), 4);
        in.read(reinterpret_cast<char *>(&hs), 4);
        std::string hdr(hs, ' ');
        in.read(&hdr[0], hs);
        std::vector<double> rec(ncoef);
        in.read(reinterpret_cast<char *>(rec.data()), (std::streamsize)(ncoef * sizeof(double)));
        const std::vector<double> cf = force.get_coefs();
        double worst = 0;
        const int nrows = (lmax + 1) * (lmax + 1);
        for (int ir = 0; ir < nmax; ir++)
          for (int row = 0; row < nrows; row++)
            worst = std::fmax(worst, std::fabs(rec[(std::size_t)ir * nrows + row] - cf[(std::size_t)row * nmax + ir]));
        expect("dump_coefs: magic", mg == 0xc0a57a2u ? 0.0 : 1.0, 0.0);
        expect("dump_coefs: header names the force id", hdr.find("id: sphereSL") == 0 && hdr.find("normed: true") != std::string::npos ? 0.0 : 1.0, 0.0);
        expect("dump_coefs: record = coefficients, n-major", worst, 0.0);
      }
    }
    // ---- error paths of the C ABI: documented status codes and exp_amd_last_error, nothing thrown, nothing lost ----
    {
      exp_amd_ctx *c = ctx.get();
      auto code = [&](const char *what, int rc, int want) {
        const char *msg = exp_amd_last_error(c);
        const bool ok = rc == want && (want == EXP_AMD_OK || (msg && msg[0]));
        std::printf("%-44s rc %d (want %d) %s  %s\n", what, rc, want, ok ? "ok" : "FAIL", want ? (msg ? msg : "(null)") : "");
        if (!ok) failures++;
      };
      exp_amd_comp *comp = nullptr;
      exp_amd_force *force = nullptr;
      exp_amd_sph_config cfg{lmax, nmax, numr, cmap, rmap, scale, rmin, rmax, xmin, dxi, 0, 0, 0, 0, 0, 0};
      code("comp_create(NULL out)", exp_amd_comp_create(c, 10, nullptr), EXP_AMD_ERR_ARG);
      code("sph_create(NULL tables)", exp_amd_sph_create(c, &cfg, nullptr, p0.data(), ev.data(), ef.data(), &force), EXP_AMD_ERR_ARG);
      exp_amd_sph_config bad = cfg;
      bad.lmax = 99;
      code("sph_create(lmax out of range)", exp_amd_sph_create(c, &bad, xi.data(), p0.data(), ev.data(), ef.data(), &force), EXP_AMD_ERR_ARG);
      bad = cfg;
      bad.numr = 1;
      code("sph_create(numr < 3)", exp_amd_sph_create(c, &bad, xi.data(), p0.data(), ev.data(), ef.data(), &force), EXP_AMD_ERR_ARG);
      code("sph_create (good)", exp_amd_sph_create(c, &cfg, xi.data(), p0.data(), ev.data(), ef.data(), &force), EXP_AMD_OK);
      code("comp_create (good)", exp_amd_comp_create(c, (std::size_t)n, &comp), EXP_AMD_OK);
      code("determine_coefficients(NULL component)", exp_amd_force_determine_coefficients(force, nullptr), EXP_AMD_ERR_ARG);
      code("determine_coefficients(NULL force)", exp_amd_force_determine_coefficients(nullptr, comp), EXP_AMD_ERR_ARG);
      std::vector<double> buf(ncoef + 7);
      code("get_coefs(ncoef mismatch)", exp_amd_force_get_coefs(force, buf.data(), ncoef + 7), EXP_AMD_ERR_ARG);
      code("set_coefs(ncoef mismatch)", exp_amd_force_set_coefs(force, buf.data(), ncoef - 1), EXP_AMD_ERR_ARG);
      code("set_level(beyond multistep)", exp_amd_force_set_level(force, 3), EXP_AMD_ERR_ARG);
      code("get_level_coefs(level out of range)", exp_amd_force_get_level_coefs(force, 5, 0, buf.data(), ncoef), EXP_AMD_ERR_ARG);
      code("comp_kick(level out of range)", exp_amd_comp_kick(comp, 0.1, 40), EXP_AMD_ERR_ARG);
      code("sph_fields before set_density", exp_amd_sph_fields(force, 1, buf.data(), buf.data(), buf.data(), 2, buf.data()), EXP_AMD_ERR_STATE);
      code("comm_init_rank(rank >= nranks)", exp_amd_comm_init_rank(c, buf.data(), 2, 5), EXP_AMD_ERR_ARG);
      code("comm_init_rank(nranks < 1)", exp_amd_comm_init_rank(c, buf.data(), 0, 0), EXP_AMD_ERR_ARG);
      // a failed call leaves the objects usable
      code("determine_coefficients after the failures", exp_amd_force_determine_coefficients(force, comp), EXP_AMD_OK);
      exp_amd_force_destroy(force);
      exp_amd_comp_destroy(comp);
      exp_amd_force_destroy(nullptr);       // destroying NULL is a no-op
      exp_amd_comp_destroy(nullptr);
    }
    // ---- block multistep: begin_run + one master step, call for call as the reference's loop ---------
    {
      VecComponent comp = fresh();
      exp_amd::Mirror mirror(ctx);
      exp_amd_sph_config cfg{lmax, nmax, numr, cmap, rmap, scale, rmin, rmax, xmin, dxi, 0, 0, 0, 0, 0, ms};
      exp_amd::SphereAMD force(ctx, mirror, &comp, cfg, xi.data(), p0.data(), ev.data(), ef.data());
      const int Mstep = 1 << ms;
      std::vector<int> mintvl(ms + 1), mfirst(Mstep + 1, 0);            // src/multistep.cc:630-680
      mintvl[0] = Mstep;
      for (int k = 1; k <= ms; k++) mintvl[k] = mintvl[k - 1] / 2;
      for (int s = 0; s <= Mstep; s++)
        for (int M = 0; M <= ms; M++)
          if (s == 0 || s % (1 << (ms - M)) == 0) { mfirst[s] = M; break; }
      auto compute_expansion = [&](int M) { force.set_multistep_level((unsigned)M); force.determine_coefficients(&comp); };
      auto compute_potential = [&](int mlevel, int mdrft) {
        exp_amd::zero_acceleration(ctx, mirror, &comp, mlevel);
        force.set_multistep_level((unsigned)mlevel);
        force.set_mdrft(mdrft);
        force.get_acceleration_and_potential(&comp);
      };
      long long nswitch = 0;
      // begin_run (src/begin.cc:80-129)
      force.multistep_reset();
      for (int M = 0; M <= ms; M++) compute_expansion(M);
      compute_potential(0, 0);
      force.multistep_update_begin();
      force.multistep_update_device(&comp, dtime, dyn, 0, 0, true);
      force.multistep_update_finish();
      force.multistep_reset();
      for (int M = 0; M <= ms; M++) compute_expansion(M);
      compute_potential(0, 0);
      // do_step (src/step.cc:98-269)
      force.multistep_reset();
      const double dts = dtime / Mstep;
      for (int mstep = 0; mstep < Mstep; mstep++) {
        for (int M = mfirst[mstep]; M <= ms; M++) {
          const double DT = dts * mintvl[M];
          exp_amd::incr_velocity(ctx, mirror, &comp, 0.5 * DT, M);
          exp_amd::incr_position(ctx, mirror, &comp, DT, M);
          compute_expansion(M);
        }
        const int mdrft = mstep + 1;
        compute_potential(mfirst[mstep], mdrft);
        for (int M = mfirst[mdrft]; M <= ms; M++) exp_amd::incr_velocity(ctx, mirror, &comp, 0.5 * dts * mintvl[M], M);
        force.multistep_update_begin();
        nswitch += force.multistep_update_device(&comp, dtime, dyn, 0, mdrft, mstep == 0);
        force.multistep_update_finish();
      }
      mirror.download(&comp);
      int nlev = 0;
      for (int i = 0; i < n; i++) nlev += comp.level[i] != mlev[i];
      expect("multistep: levels (mismatches)", (double)nlev, 0.0);
      expect("multistep: level changes", std::fabs((double)(nswitch - nsw_ref)), 0.0);
      expect("multistep: positions", maxdiff3(comp.x, comp.y, comp.z, mpos), 1e-11);
      expect("multistep: velocities", maxdiff3(comp.vx, comp.vy, comp.vz, mvel), 1e-9 * maxabs(mvel));
      expect("multistep: accelerations", maxdiff3(comp.ax, comp.ay, comp.az, macc), 1e-8 * maxabs(macc));
      expect("multistep: combined coefficients", maxdiff(force.get_coefs(), mcoef), 1e-10 * maxabs(mcoef));
    }
  } catch (const exp_amd::Error &e) {
    std::fprintf(stderr, "exp_amd::Error: %s\n", e.what());
    return 3;
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
