// Driver for two more files of the reference that compile on their own (std only), where they lie:
//   exputil/gaussQ.cc   LegeQuad: the Gauss-Legendre knots and weights on [0, 1] of makeFromFunction / computeQuadrature
//                       (expui/BiorthBasis.cc:5230-5458)
//   expui/UnitValidator.cc  the unit type / name alias tables of Coefs::setUnits (expui/Coefficients.cc:61)
//   include/KDtree.H    the k-nearest-neighbour search of Utility::getDensityCenter (expui/Centering.cc)
//   exputil/VtkGrid.cc  the rectilinear-grid writer of a build without the VTK library: FieldGenerator::file_slices /
//                       file_volumes (expui/FieldGenerator.cc:512-564, 725-774)
// Test infrastructure only (tests/test_ref_util.py).
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "gaussQ.H"
#include "VtkGrid.H"

extern "C" int ref_legequad(int n, double *knots, double *weights)
{
  try {
    LegeQuad lq(n);
    for (int i = 0; i < n; i++) { knots[i] = lq.knot(i); weights[i] = lq.weight(i); }
    return 0;
  } catch (std::exception &) { return -1; }
}

// VtkGrid(nx, ny, nz, bounds); Add(data_k, name_k) for every field; Write(name) -> <name>.vtr
extern "C" int ref_vtk_write(const char *name, int nx, int ny, int nz, const double *bounds, int nfields,
                             const char *const *names, const double *data)
{
  try {
    VtkGrid g(nx, ny, nz, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5]);
    const size_t n = (size_t)nx * ny * nz;
    for (int k = 0; k < nfields; k++) {
      std::vector<double> v(data + k * n, data + (k + 1) * n);
      g.Add(v, names[k]);
    }
    g.Write(name);
    return 0;
  } catch (std::exception &) { return -1; }
}

// expui/UnitValidator.cc: the type / unit alias tables behind Coefs::setUnits
#include "UnitValidator.H"

static void join(const std::vector<std::string> &v, char *out, int cap)
{
  std::string s;
  for (auto &x : v) { s += x; s += '\n'; }
  strncpy(out, s.c_str(), cap - 1); out[cap - 1] = 0;
}

// ok, canonical type, canonical unit for (type, unit)
extern "C" int ref_unit_check(const char *type, const char *unit, char *ctype, char *cunit, int cap)
{
  UnitValidator check;
  auto [ok, t, u] = check(type, unit);
  strncpy(ctype, t.c_str(), cap - 1); ctype[cap - 1] = 0;
  strncpy(cunit, u.c_str(), cap - 1); cunit[cap - 1] = 0;
  return ok ? 1 : 0;
}

// which: 0 getAllowedTypes(), 1 getAllowedTypeAliases(type), 2 getAllowedUnits(type) -> newline-separated
extern "C" int ref_unit_list(int which, const char *type, char *out, int cap)
{
  UnitValidator check;
  try {
    if (which == 0) join(check.getAllowedTypes(), out, cap);
    else if (which == 1) join(check.getAllowedTypeAliases(type), out, cap);
    else join(check.getAllowedUnits(type), out, cap);
    return 0;
  } catch (std::exception &) { out[0] = 0; return -1; }
}

// The old-style PSP info string as PSPout::PSPout takes it apart (exputil/ParticleReader.cc:1405-1412):
// StringTok<string> tokens(info); name = trim_copy(tokens(":")); id, cparam, fparam likewise (include/StringTok.H,
// exputil/Sutils.cc)
#include "StringTok.H"
#include "Sutils.H"

extern "C" void ref_old_info(const char *info, char *out4, int cap)
{
  StringTok<std::string> tokens(info);
  std::string s;
  for (int k = 0; k < 4; k++) { s += trim_copy(tokens(":")); s += '\n'; }
  strncpy(out4, s.c_str(), cap - 1); out4[cap - 1] = 0;
}

// include/KDtree.H: the k-nearest-neighbour density of getDensityCenter (expui/Centering.cc:66-96), one value per point:
// nearestN(points[i], Ndens) -> (summed mass of the neighbours, the point itself among them) / (4 pi/3 r_N^3) / total mass
#include <iostream>
#include <cmath>
#include "KDtree.H"

extern "C" int ref_kd_density(int n, const double *pos, const double *mass, int Ndens, double *density)
{
  using point3 = KDtree::point<double, 3>;
  using tree3 = KDtree::kdtree<double, 3>;
  try {
    std::vector<point3> points;
    double KDmass = 0.0;
    for (int i = 0; i < n; i++) {
      KDmass += mass[i];
      points.push_back(point3({pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, mass[i]));
    }
    tree3 tree(points.begin(), points.end());
    for (int i = 0; i < n; i++) {
      auto ret = tree.nearestN(points[i], Ndens);
      double volume = 4.0 * M_PI / 3.0 * std::pow(std::get<2>(ret), 3.0);
      density[i] = (volume > 0.0 && KDmass > 0.0) ? std::get<1>(ret) / volume / KDmass : 0.0;
    }
    return 0;
  } catch (std::exception &) { return -1; }
}
