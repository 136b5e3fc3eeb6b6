// exp_amd_potaccel.hpp -- C++ adaptor over the C ABI (exp_amd.h) with the method names of EXP's
// force-method interface, so that ComponentContainer-style callers read unchanged.
//
// What it mirrors (paths relative to the EXP source tree):
//   class PotAccel                          src/PotAccel.H:39-324
//     determine_coefficients(Component*)      :178-180        get_acceleration_and_potential(Component*) :173
//     set_multistep_level(unsigned)           :285            SetExternal() / ClearExternal()            :215-218
//     multistep_reset()                       :288            multistep_update_begin / _finish           :271-281
//     multistep_update_cuda()                 :283  (the device path replaces the per-particle
//                                                    multistep_update(cur, next, c, i, id) by one sweep)
//     Used()                                  :207            setScale / getScale                        :294-297
//   free functions the step loop calls        incr_position(dt, mlevel) src/incpos.cc:72, incr_velocity
//                                             src/incvel.cc:90, adjust_multistep_level src/multistep.cc:344
//   the particle mirror                       Component::ParticlesToCuda / CudaToParticles
//                                             src/cudaComponent.cu:621-727
//
// Header-only, C++17, no dependency beyond the C ABI.  On the EXP side `ComponentView` is implemented
// by a ten-line wrapper over `Component` (Number(), Pos/Vel/Mass accessors, level, centre); the rest
// of this file is what `class SphereAMD : public PotAccel` forwards to.  tests/cpp/test_potaccel.cpp
// drives a KDK step and a block-multistep master step through it with no Python in the process.
#ifndef EXP_AMD_POTACCEL_HPP
#define EXP_AMD_POTACCEL_HPP

#include <cstdint>
#include <cstdio>
#include <map>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "exp_amd.h"

namespace exp_amd {

// EXP reports errors by exceptions that unwind to main (GenericError, src/expand.cc catch blocks);
// the ABI never throws, the adaptor turns a non-zero status into one.
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc, const exp_amd_ctx *ctx)
{
  if (rc == EXP_AMD_OK) return;
  const char *m = ctx ? exp_amd_last_error(ctx) : exp_amd_last_global_error();
  throw Error(rc, std::string("exp_amd error ") + std::to_string(rc) + ": " + (m ? m : "?"));
}

// What the adaptor needs to see of a Component (src/Component.H): the particle count, the phase
// space in the component's own particle order, the multistep levels and the expansion centre.
struct ComponentView {
  virtual ~ComponentView() = default;
  virtual std::size_t Number() const = 0;
  // fill host arrays of length Number(); any pointer may be null (= not wanted)
  virtual void gather(double *mass, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                      double *ax, double *ay, double *az, double *pot, std::int32_t *level) const = 0;
  // take the device state back (CudaToParticles); any pointer may be null (= not provided)
  virtual void scatter(const double *x, const double *y, const double *z, const double *vx, const double *vy,
                       const double *vz, const double *ax, const double *ay, const double *az,
                       const double *pot, const std::int32_t *level) = 0;
  // Component::center (Local | Centered positions subtract it, src/Component.H:748-757)
  virtual void center(double c[3]) const { c[0] = c[1] = c[2] = 0.0; }
  // Component::rtrunc and com0 (src/Component.cc:213, :4194-4202): beyond rtrunc of com0 + center a particle is frozen
  // -- no force method accumulates it, differences it or accelerates it.  1e20 (the reference's default): never
  virtual double rtrunc() const { return 1.0e20; }
  virtual void com0(double c[3]) const { c[0] = c[1] = c[2] = 0.0; }
  // Component::tidal / rcom (src/Component.cc:998-1000, :1024): tidal >= 0 switches `consp` on -- Component::fix_positions
  // (exp_amd_comp_fix_positions) flags a particle beyond rcom of com0 + center in iattrib[tidal] and leaves it out of the
  // centre-of-mass sums from then on (:3317-3334); exp_amd_comp_get_escaped / _set_escaped are that attribute column
  virtual int tidal() const { return -1; }
  virtual double rcom() const { return 1.0e20; }
  // Component::NoSwitch / FreezeLev / DTreset (keys noswitch, freezeL, dtreset; src/Component.cc:253-255): read by
  // adjust_multistep_level (src/multistep.cc:136-158); handed to the store when the component is uploaded
  // (exp_amd_comp_set_level_policy)
  virtual bool NoSwitch() const { return false; }
  virtual bool FreezeLev() const { return false; }
  virtual bool DTreset() const { return true; }
  // double Component::Adiabatic() (src/Component.cc:4214-4220) at the caller's current tnow; 1 without ton / toff
  virtual double Adiabatic() const { return 1.0; }
};

// One GPU / one stream / one rank, as in EXP (src/Component.H:1054-1079).
class Context {
  exp_amd_ctx *h_ = nullptr;

public:
  explicit Context(int device = 0, void *stream = nullptr) { check(exp_amd_ctx_create(device, stream, &h_), nullptr); }
  ~Context() { exp_amd_ctx_destroy(h_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  exp_amd_ctx *get() const { return h_; }
  void synchronize() { check(exp_amd_ctx_synchronize(h_), h_); }
  // the coefficient all-reduce over the ranks (replaces src/SphericalBasis.cc:864-903):
  //   char id[128]; if (myid == 0) Context::unique_id(id); MPI_Bcast(id, 128, MPI_BYTE, 0, comm);
  //   ctx.init_comm(id, numprocs, myid);
  static void unique_id(void *id128) { check(exp_amd_comm_get_unique_id(id128), nullptr); }
  void init_comm(const void *id128, int nranks, int rank) { check(exp_amd_comm_init_rank(h_, id128, nranks, rank), h_); }
  void set_allreduce(exp_amd_allreduce_fn fn, void *user) { check(exp_amd_comm_set_callback(h_, fn, user), h_); }
  // ... with numprocs / myid of the communicator the callback reduces over
  void set_allreduce(exp_amd_allreduce_fn fn, void *user, int nranks, int rank)
  {
    check(exp_amd_comm_set_world(h_, nranks, rank), h_);
    check(exp_amd_comm_set_callback(h_, fn, user), h_);
  }
};

// Device stores of the Components a force method has met: ParticlesToCuda on first use,
// CudaToParticles on request.
class Mirror {
  Context &ctx_;
  std::map<ComponentView *, exp_amd_comp *> dev_;

public:
  explicit Mirror(Context &ctx) : ctx_(ctx) {}
  ~Mirror() { for (auto &kv : dev_) exp_amd_comp_destroy(kv.second); }
  Mirror(const Mirror &) = delete;
  Mirror &operator=(const Mirror &) = delete;

  bool has(ComponentView *c) const { return dev_.count(c) != 0; }

  // Component::ParticlesToCuda (src/cudaComponent.cu:621-686)
  exp_amd_comp *upload(ComponentView *c)
  {
    const std::size_t n = c->Number();
    auto it = dev_.find(c);
    bool fresh = false;
    if (it != dev_.end() && exp_amd_comp_size(it->second) != n) {
      exp_amd_comp_destroy(it->second);
      dev_.erase(it);
      it = dev_.end();
    }
    if (it == dev_.end()) {
      // (no map entry is left behind when the creation throws: dev() must never hand out a null store)
      exp_amd_comp *nd = nullptr;
      check(exp_amd_comp_create(ctx_.get(), n, &nd), ctx_.get());
      it = dev_.emplace(c, nd).first;
      fresh = true;
    }
    exp_amd_comp *d = it->second;
    std::vector<double> v[11];
    for (auto &a : v) a.resize(n ? n : 1);
    std::vector<std::int32_t> lev(n ? n : 1);
    c->gather(v[0].data(), v[1].data(), v[2].data(), v[3].data(), v[4].data(), v[5].data(), v[6].data(),
              v[7].data(), v[8].data(), v[9].data(), v[10].data(), lev.data());
    check(exp_amd_comp_upload(d, v[0].data(), v[1].data(), v[2].data(), v[3].data(), v[4].data(), v[5].data(),
                              v[6].data()), ctx_.get());
    check(exp_amd_comp_upload_acc(d, v[7].data(), v[8].data(), v[9].data(), v[10].data()), ctx_.get());
    // a store that already exists may hold levels from an earlier upload: they are replaced even when every host
    // level has gone back to zero; only a fresh store (all zero by construction) can skip an all-zero upload
    bool any = !fresh;
    for (std::size_t i = 0; i < n && !any; i++) any = lev[i] != 0;
    if (any) check(exp_amd_comp_upload_levels(d, lev.data()), ctx_.get());
    double ctr[3];
    c->center(ctr);
    check(exp_amd_comp_set_center(d, ctr), ctx_.get());
    if (c->rtrunc() < 1.0e20 || c->tidal() >= 0) {    // Component::freeze / escape_com (src/Component.cc:4194-4212): com0
      double c0[3];
      c->com0(c0);
      check(exp_amd_comp_set_rtrunc(d, c->rtrunc(), c0), ctx_.get());
    }
    if (c->tidal() >= 0) check(exp_amd_comp_set_consp(d, 1, c->rcom()), ctx_.get());
    if (c->NoSwitch() || c->FreezeLev())
      check(exp_amd_comp_set_level_policy(d, c->NoSwitch() ? 1 : 0, c->FreezeLev() ? 1 : 0, c->DTreset() ? 1 : 0), ctx_.get());
    return d;
  }

  // the device store of `c` (uploaded on first use)
  exp_amd_comp *dev(ComponentView *c)
  {
    auto it = dev_.find(c);
    return it != dev_.end() ? it->second : upload(c);
  }

  // Component::CudaToParticles (src/cudaComponent.cu:688-727)
  void download(ComponentView *c)
  {
    exp_amd_comp *d = dev(c);
    const std::size_t n = exp_amd_comp_size(d);
    std::vector<double> v[10];
    for (auto &a : v) a.resize(n ? n : 1);
    std::vector<std::int32_t> lev(n ? n : 1);
    check(exp_amd_comp_download(d, nullptr, v[0].data(), v[1].data(), v[2].data(), v[3].data(), v[4].data(),
                                v[5].data(), v[6].data(), v[7].data(), v[8].data(), v[9].data()), ctx_.get());
    check(exp_amd_comp_download_levels(d, lev.data()), ctx_.get());
    c->scatter(v[0].data(), v[1].data(), v[2].data(), v[3].data(), v[4].data(), v[5].data(), v[6].data(),
               v[7].data(), v[8].data(), v[9].data(), lev.data());
  }
};

// The PotAccel-shaped front of one force method.  Concrete methods (SphereAMD, CylinderAMD below)
// only differ in how the ABI object is created.
class PotAccelAMD {
protected:
  Context &ctx_;
  Mirror &mirror_;
  exp_amd_force *force_ = nullptr;
  ComponentView *component_ = nullptr;     // PotAccel::component (the owner)
  ComponentView *cC_ = nullptr;            // PotAccel::cC (the current target)
  unsigned mlevel_ = 0;
  bool use_external_ = false;
  int multistep_ = 0;
  int mdrft_ = 0;                          // EXP's global `mdrft` (src/global.cc), set by the step loop

  PotAccelAMD(Context &ctx, Mirror &mirror, ComponentView *c0, int multistep)
      : ctx_(ctx), mirror_(mirror), component_(c0), cC_(c0), multistep_(multistep) {}

public:
  virtual ~PotAccelAMD() { exp_amd_force_destroy(force_); }
  PotAccelAMD(const PotAccelAMD &) = delete;
  PotAccelAMD &operator=(const PotAccelAMD &) = delete;
  exp_amd_force *get() const { return force_; }

  // ---- PotAccel's interface -------------------------------------------------------------------
  void RegisterComponent(ComponentView *c) { component_ = c; }                     // :195
  void set_multistep_level(unsigned n)                                              // :285
  {
    mlevel_ = n;
    check(exp_amd_force_set_level(force_, (int)n), ctx_.get());
  }
  void SetExternal() { use_external_ = true; }                                      // :215
  void ClearExternal() { use_external_ = false; }                                   // :218
  void determine_coefficients(ComponentView *c) { cC_ = c; determine_coefficients(); }   // :179-180
  void determine_coefficients()                                                     // :178
  {
    // `double adb = component->Adiabatic();` (src/SphericalBasis.cc:441, src/Cylinder.cc:834): the OWNER's, at this tnow
    check(exp_amd_force_set_mass_scale(force_, component_->Adiabatic()), ctx_.get());
    // (with "self_consistent: false" the library returns at once after the first completed call unless `initializing`,
    // src/SphericalBasis.cc:694, src/Cylinder.cc:959)
    check(exp_amd_force_determine_coefficients(force_, mirror_.dev(cC_)), ctx_.get());
  }
  // the "self_consistent" key (src/SphericalBasis.cc:114-117, src/Cylinder.cc:557-558) and EXP's global `initializing`
  // (src/begin.cc:80, :129), which begin_run sets around its expansions
  void set_self_consistent(bool on) { check(exp_amd_force_set_self_consistent(force_, on ? 1 : 0), ctx_.get()); }
  void set_initializing(bool on) { check(exp_amd_force_set_initializing(force_, on ? 1 : 0), ctx_.get()); }
  bool coefs_frozen() const { return exp_amd_force_coefs_frozen(force_) != 0; }
  // src/SphericalBasis.cc:381, :1663-1777 / src/Cylinder.cc:1448-1500: the self call recombines the
  // per-level coefficient sets first (use_external == false branch)
  void get_acceleration_and_potential(ComponentView *c)                             // :173
  {
    cC_ = c;
    if (multistep_ && !use_external_)
      check(exp_amd_force_compute_multistep_coefficients(force_, mdrft_), ctx_.get());
    check(exp_amd_force_get_acceleration(force_, mirror_.dev(c), use_external_ ? 1 : 0), ctx_.get());
  }
  void multistep_reset() { check(exp_amd_force_multistep_reset(force_), ctx_.get()); }   // :288
  // The CPU path calls multistep_update(cur, next, c, i, id) once per particle that changes level
  // between _begin and _finish (src/multistep.cc:202, :539).  On the device the whole of
  // adjust_multistep_level for this method's component -- criteria, differencing, all-reduce, level
  // lists -- is one call, exactly where EXP's CUDA path has multistep_update_cuda (:283,
  // src/multistep.cc:361-366); _begin / _finish keep their places in the caller and do nothing.
  void multistep_update_begin() {}                                                  // :274
  void multistep_update_finish() {}                                                 // :280
  long long multistep_update_device(ComponentView *c, double dtime, const double dynfrac[5], int shiftlevl,
                                    int mdrft, bool first_step)
  {
    long long nswitch = 0;
    // `double mass = c->Mass(i) * component->Adiabatic();` (src/SphericalBasis.cc:1161, src/Cylinder.cc:1758)
    check(exp_amd_force_set_mass_scale(force_, component_->Adiabatic()), ctx_.get());
    check(exp_amd_force_adjust_multistep_level(force_, mirror_.dev(c), dtime, dynfrac, shiftlevl, mdrft,
                                               first_step ? 1 : 0, &nswitch), ctx_.get());
    return nswitch;
  }
  long long Used()                                                                  // :207
  {
    long long u = 0;
    check(exp_amd_force_used(force_, &u), ctx_.get());
    return u;
  }
  // EXP's global drifted sub-step index (src/step.cc:113, :176), needed by the coefficient
  // interpolation of get_acceleration_and_potential
  void set_mdrft(int mdrft) { mdrft_ = mdrft; }

  // ---- coefficient access (HtoD_coefs / DtoH_coefs, dump_coefs) --------------------------------
  std::vector<double> get_coefs()
  {
    std::vector<double> c(exp_amd_force_ncoef(force_));
    check(exp_amd_force_get_coefs(force_, c.data(), c.size()), ctx_.get());
    return c;
  }
  void set_coefs(const std::vector<double> &c) { check(exp_amd_force_set_coefs(force_, c.data(), c.size()), ctx_.get()); }

  // PotAccel::dump_coefs(ostream&) (src/PotAccel.H:224): one record of the native coefficient stream at time tnow
  virtual void dump_coefs(std::ostream &out, double tnow) = 0;

protected:
  // magic number + YAML header of a new-style record (NewCoefs; src/SphericalBasis.cc:1831-1861,
  // exputil/EmpCylSL.cc:5870-5898): `cmagic` (0xc0a57a2 for the sphere, src/SphericalBasis.H:368; 0xc0a57a3 for the cylinder, include/EmpCylSL.H:222), the header's byte count, the header
  static void write_header(std::ostream &out, unsigned int cmagic, const std::string &yaml)
  {
    const unsigned int hsize = (unsigned int)yaml.size();
    out.write(reinterpret_cast<const char *>(&cmagic), sizeof(unsigned int));
    out.write(reinterpret_cast<const char *>(&hsize), sizeof(unsigned int));
    out.write(yaml.data(), hsize);
  }
  static std::string num(double v)
  {
    char b[40];
    std::snprintf(b, sizeof b, "%.17g", v);
    return b;
  }
};

// sphereSL: class Sphere : SphericalBasis (src/Sphere.cc:28-96) given SLGridSph's tables
class SphereAMD : public PotAccelAMD {
public:
  // xi[numr], p0[numr], ev[(lmax+1)*nmax], ef[(lmax+1)*nmax*numr] as SLGridSph holds them
  // (exputil/SLGridMP2.cc:1321-1382); cfg carries Lmax, nmax, numr, cmap, rmapping, scale, the
  // expansion window, the xi grid and the SphericalBasis flags
  SphereAMD(Context &ctx, Mirror &mirror, ComponentView *c0, const exp_amd_sph_config &cfg, const double *xi,
            const double *p0, const double *ev, const double *ef)
      : PotAccelAMD(ctx, mirror, c0, cfg.multistep)
  {
    check(exp_amd_sph_create(ctx.get(), &cfg, xi, p0, ev, ef, &force_), ctx.get());
    lmax_ = cfg.lmax; nmax_ = cfg.nmax; scale_ = cfg.scale;
  }
  // SphericalBasis::dump_coefs (src/SphericalBasis.cc:1829-1879): header {id, time, scale, nmax, lmax, normed},
  // then for every radial order the real rows in (l, m; cos, sin) order -- the row order of the device buffer
  void dump_coefs(std::ostream &out, double tnow) override
  {
    write_header(out, 0xc0a57a2u, "id: sphereSL\ntime: " + num(tnow) + "\nscale: " + num(scale_) + "\nnmax: " +
                          std::to_string(nmax_) + "\nlmax: " + std::to_string(lmax_) + "\nnormed: true");
    const std::vector<double> c = get_coefs();               // [(lmax+1)^2][nmax]
    const int nrows = (lmax_ + 1) * (lmax_ + 1);
    for (int ir = 0; ir < nmax_; ir++)
      for (int row = 0; row < nrows; row++)
        out.write(reinterpret_cast<const char *>(&c[(std::size_t)row * nmax_ + ir]), sizeof(double));
  }

  // FIX_L0 (src/SphericalBasis.cc:119, :1689-1694)
  void set_fix_l0(bool on) { check(exp_amd_sph_set_fix_l0(force_, on ? 1 : 0), ctx_.get()); }

private:
  int lmax_ = 0, nmax_ = 0;
  double scale_ = 1.0;
};

// cylinder: class Cylinder (src/Cylinder.cc) given EmpCylSL's tables
class CylinderAMD : public PotAccelAMD {
public:
  // tab[6][mmax+1][nmax][numx+1][numy+1]: potC, rforceC, zforceC, potS, rforceS, zforceS
  CylinderAMD(Context &ctx, Mirror &mirror, ComponentView *c0, const exp_amd_cyl_config &cfg, const double *tab)
      : PotAccelAMD(ctx, mirror, c0, cfg.multistep)
  {
    check(exp_amd_cyl_create(ctx.get(), &cfg, tab, &force_), ctx.get());
    mmax_ = cfg.mmax; nmax_ = cfg.nmax;
  }
  // Cylinder::dump_coefs -> EmpCylSL::dump_coefs_binary (src/Cylinder.cc:1618-1621, exputil/EmpCylSL.cc:5868-5920):
  // header {time, mmax, nmax}, then per m the cosine row and, for m > 0, the sine row
  void dump_coefs(std::ostream &out, double tnow) override
  {
    write_header(out, 0xc0a57a3u, "time: " + num(tnow) + "\nmmax: " + std::to_string(mmax_) + "\nnmax: " + std::to_string(nmax_));
    const std::vector<double> c = get_coefs();               // [2][mmax+1][nmax]: cos block, sin block
    const std::size_t half = (std::size_t)(mmax_ + 1) * nmax_;
    for (int mm = 0; mm <= mmax_; mm++) {
      out.write(reinterpret_cast<const char *>(&c[(std::size_t)mm * nmax_]), sizeof(double) * nmax_);
      if (mm) out.write(reinterpret_cast<const char *>(&c[half + (std::size_t)mm * nmax_]), sizeof(double) * nmax_);
    }
  }

private:
  int mmax_ = 0, nmax_ = 0;

public:
  // the "mlim" key: `if (mlim>=0) ortho->set_mlim(mlim);` (src/Cylinder.cc:225)
  void set_mlim(int mlim) { if (mlim >= 0) check(exp_amd_cyl_set_mlim(force_, mlim), ctx_.get()); }
  double cylmass()
  {
    double m = 0.0;
    check(exp_amd_cyl_get_cylmass(force_, &m), ctx_.get());
    return m;
  }
};

// ---- the free functions of the step loop ---------------------------------------------------------
// incr_position(dt, mlevel) (src/incpos.cc:72) / incr_velocity(dt, mlevel) (src/incvel.cc:90) for one
// component; mlevel < 0: all levels.  ComponentContainer's zeroing loop (src/ComponentContainer.cc:
// 641-665) likewise.
inline void incr_position(Context &ctx, Mirror &m, ComponentView *c, double dt, int mlevel = -1)
{
  check(exp_amd_comp_drift(m.dev(c), dt, mlevel), ctx.get());
}
inline void incr_velocity(Context &ctx, Mirror &m, ComponentView *c, double dt, int mlevel = -1)
{
  check(exp_amd_comp_kick(m.dev(c), dt, mlevel), ctx.get());
}
// OutLog::Run's particle loop (src/OutLog.cc:392-478; with CUDA the reference copies the particles back first, :377-385):
// {mass, m x [3], m v [3], angular momentum [3], kinetic energy, 0.5 m pot, Clausius virial, number of bodies}, already
// reduced over the ranks -- what a device-resident OutLog divides and prints
struct LogSums { double v[14]; };
inline LogSums log_sums(Context &ctx, Mirror &m, ComponentView *c)
{
  LogSums s{};
  check(exp_amd_comp_log_sums(m.dev(c), s.v), ctx.get());
  return s;
}

inline void zero_acceleration(Context &ctx, Mirror &m, ComponentView *c, int mlevel = 0)
{
  check(exp_amd_comp_zero_acc(m.dev(c), mlevel), ctx.get());
}

}  // namespace exp_amd
#endif
