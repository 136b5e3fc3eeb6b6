// Shared pieces of the g++-built adaptor tests (tests/cpp/test_potaccel.cpp, test_potaccel2.cpp): a Component reduced to
// what the adaptor asks of it, fixture reading, comparisons.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "exp_amd_potaccel.hpp"

using exp_amd::ComponentView;

// a Component reduced to what the adaptor asks of it
struct VecComponent : ComponentView {
  std::vector<double> m, x, y, z, vx, vy, vz, ax, ay, az, pot;
  std::vector<std::int32_t> level;
  // the component keys that default to off: rtrunc / com0 (src/Component.cc:213, :1023), ton / toff / twid (:1040-1055)
  double ctr_[3] = {0, 0, 0}, rtrunc_ = 1.0e20, com0_[3] = {0, 0, 0};
  bool adiabatic_ = false;
  double ton_ = -1.0e20, toff_ = 1.0e20, twid_ = 0.1;
  const double *tnow_ = nullptr;             // EXP's global tnow
  void center(double c[3]) const override { for (int k = 0; k < 3; k++) c[k] = ctr_[k]; }
  double rtrunc() const override { return rtrunc_; }
  void com0(double c[3]) const override { for (int k = 0; k < 3; k++) c[k] = com0_[k]; }
  double Adiabatic() const override                                    // src/Component.cc:4214-4220
  {
    if (!adiabatic_) return 1.0;
    return 0.25 * (1.0 + std::erf((*tnow_ - ton_) / twid_)) * (1.0 + std::erf((toff_ - *tnow_) / twid_));
  }
  explicit VecComponent(std::size_t n)
      : m(n), x(n), y(n), z(n), vx(n), vy(n), vz(n), ax(n, 0.0), ay(n, 0.0), az(n, 0.0), pot(n, 0.0), level(n, 0) {}
  std::size_t Number() const override { return m.size(); }
  void gather(double *M, double *X, double *Y, double *Z, double *VX, double *VY, double *VZ, double *AX,
              double *AY, double *AZ, double *POT, std::int32_t *LEV) const override
  {
    auto cp = [&](double *dst, const std::vector<double> &src) { if (dst) std::memcpy(dst, src.data(), src.size() * sizeof(double)); };
    cp(M, m); cp(X, x); cp(Y, y); cp(Z, z); cp(VX, vx); cp(VY, vy); cp(VZ, vz); cp(AX, ax); cp(AY, ay); cp(AZ, az); cp(POT, pot);
    if (LEV) std::memcpy(LEV, level.data(), level.size() * sizeof(std::int32_t));
  }
  void scatter(const double *X, const double *Y, const double *Z, const double *VX, const double *VY,
               const double *VZ, const double *AX, const double *AY, const double *AZ, const double *POT,
               const std::int32_t *LEV) override
  {
    auto cp = [&](std::vector<double> &dst, const double *src) { if (src) std::memcpy(dst.data(), src, dst.size() * sizeof(double)); };
    cp(x, X); cp(y, Y); cp(z, Z); cp(vx, VX); cp(vy, VY); cp(vz, VZ); cp(ax, AX); cp(ay, AY); cp(az, AZ); cp(pot, POT);
    if (LEV) std::memcpy(level.data(), LEV, level.size() * sizeof(std::int32_t));
  }
};

static std::vector<double> rd(std::ifstream &f, std::size_t n)
{
  std::vector<double> v(n);
  f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * sizeof(double)));
  return v;
}

static int failures = 0;
static void expect(const char *what, double err, double tol)
{
  const bool ok = err <= tol;
  std::printf("%-44s err %.3e  tol %.1e  %s\n", what, err, tol, ok ? "ok" : "FAIL");
  if (!ok) failures++;
}
static double maxabs(const std::vector<double> &a) { double s = 0; for (double v : a) s = std::fmax(s, std::fabs(v)); return s; }
static double maxdiff3(const std::vector<double> &a, const std::vector<double> &b, const std::vector<double> &c,
                       const std::vector<double> &ref /* [n][3] */)
{
  double s = 0;
  for (std::size_t i = 0; i < a.size(); i++) {
    s = std::fmax(s, std::fabs(a[i] - ref[3 * i]));
    s = std::fmax(s, std::fabs(b[i] - ref[3 * i + 1]));
    s = std::fmax(s, std::fabs(c[i] - ref[3 * i + 2]));
  }
  return s;
}
static double maxdiff(const std::vector<double> &a, const std::vector<double> &ref)
{
  double s = 0;
  for (std::size_t i = 0; i < a.size(); i++) s = std::fmax(s, std::fabs(a[i] - ref[i]));
  return s;
}

