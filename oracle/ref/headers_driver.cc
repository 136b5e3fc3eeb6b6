// Driver for the two self-contained headers of the reference that sit on the path (compiled against
// them WHERE THEY LIE, -I$(REF)/include; nothing is copied):
//   include/QuadLS.H -- the quadratic least squares behind Orient's PseudoAccel (include/PseudoAccel.H:45-91)
//   include/coef.H   -- SphCoefHeader / CylCoefHeader, the headers of the legacy native coefficient streams
// Test infrastructure only (tests/test_ref_headers.py pins oracle/bfe_oracle.c:orc_quadls and the struct
// formats of exp_amd/coefs.py against them).
#include <cmath>
#include <cstddef>
#include <vector>

#include "QuadLS.H"
#include "coef.H"

extern "C" void ref_quadls(int n, const double *x, const double *y, double *out3)
{
  std::vector<double> X(x, x + n), Y(y, y + n);
  QuadLS<std::vector<double>> q(X, Y);
  auto c = q.coefs();
  out3[0] = std::get<0>(c);
  out3[1] = std::get<1>(c);
  out3[2] = std::get<2>(c);
}

// {sizeof(SphCoefHeader), offsets of id, tnow, scale, nmax, Lmax, sizeof(CylCoefHeader), offsets of time, mmax, nmax}
extern "C" void ref_coef_layout(long *out)
{
  out[0] = (long)sizeof(SphCoefHeader);
  out[1] = (long)offsetof(SphCoefHeader, id);
  out[2] = (long)offsetof(SphCoefHeader, tnow);
  out[3] = (long)offsetof(SphCoefHeader, scale);
  out[4] = (long)offsetof(SphCoefHeader, nmax);
  out[5] = (long)offsetof(SphCoefHeader, Lmax);
  out[6] = (long)sizeof(CylCoefHeader);
  out[7] = (long)offsetof(CylCoefHeader, time);
  out[8] = (long)offsetof(CylCoefHeader, mmax);
  out[9] = (long)offsetof(CylCoefHeader, nmax);
}
