// Driver for the two self-contained headers of the reference that sit on the path (compiled against
// them WHERE THEY LIE, -I$(REF)/include; nothing is copied):
//   include/QuadLS.H -- the quadratic least squares behind Orient's PseudoAccel (include/PseudoAccel.H:45-91)
//   include/coef.H   -- SphCoefHeader / CylCoefHeader, the headers of the legacy native coefficient streams
//   include/P2Quantile.H -- the streaming median of ParticleReader::PrintSummary (exputil/ParticleReader.cc:2298-2407)
//   include/gadget.H -- the Gadget-2 snapshot header GadgetNative reads (exputil/ParticleReader.cc:77-148)
// Test infrastructure only (tests/test_ref_headers.py pins oracle/bfe_oracle.c:orc_quadls and the struct
// formats of exp_amd/coefs.py against them).
#include <cmath>
#include <cstddef>
#include <vector>

#include <algorithm>
#include <stdexcept>

#include "QuadLS.H"
#include "coef.H"
#include "P2Quantile.H"
#include "gadget.H"

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

extern "C" double ref_p2quantile(long n, const double *x, double prob)
{
  P2Quantile q(prob);
  for (long i = 0; i < n; i++) q.addValue(x[i]);
  return q.getQuantile();
}

// {sizeof(gadget_header), offsets of npart, mass, time, redshift, flag_sfr, npartTotal, num_files, BoxSize, flag_metals,
//  npartTotalHighWord, fill}
extern "C" void ref_gadget_layout(long *out)
{
  out[0] = (long)sizeof(gadget_header);
  out[1] = (long)offsetof(gadget_header, npart);
  out[2] = (long)offsetof(gadget_header, mass);
  out[3] = (long)offsetof(gadget_header, time);
  out[4] = (long)offsetof(gadget_header, redshift);
  out[5] = (long)offsetof(gadget_header, flag_sfr);
  out[6] = (long)offsetof(gadget_header, npartTotal);
  out[7] = (long)offsetof(gadget_header, num_files);
  out[8] = (long)offsetof(gadget_header, BoxSize);
  out[9] = (long)offsetof(gadget_header, flag_metals);
  out[10] = (long)offsetof(gadget_header, npartTotalHighWord);
  out[11] = (long)offsetof(gadget_header, fill);
}
