// Host-side helpers of the C-ABI that have no device part.
#include "../../include/exp_amd.h"

// out[bin[i]] += val[i] for every i with 0 <= bin[i] < nbins, one particle at a time in the order given: the float
// accumulators of FieldGenerator::histogram2d / histogram1d / histo1dlog (expui/FieldGenerator.cc:776-1009) take a
// double addend each (`ret(i, j) += p->mass * fac`: promoted, added in double, rounded to float), so the result depends
// on the order and a parallel reduction would not reproduce it.
extern "C" int exp_amd_host_binsum_f32(long long n, const int *bin, const double *val, int nbins, float *out)
{
  if (n < 0 || nbins < 0 || (n > 0 && (!bin || !val)) || (nbins > 0 && !out)) return EXP_AMD_ERR_ARG;
  for (long long i = 0; i < n; i++) {
    const int b = bin[i];
    if (b >= 0 && b < nbins) out[b] = (float)((double)out[b] + val[i]);
  }
  return EXP_AMD_OK;
}
