// Host-side helpers of the C-ABI that have no device part.
#include "../../include/exp_amd.h"

#include <cstring>

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

// The packed particle records of a PSP file (exputil/Particle.cc:333-388; PParticle::read, include/ParticleReader.H:
// 276-315) -> separate arrays, reals widened to double: [unsigned long indx]? real mass, pos[3], vel[3], pot;
// int iattrib[niatr]; real dattrib[ndatr].  `stride` records apart (a rank's share of a file: first, first + stride, ...).
// numpy takes the fields out of such unaligned records at ~0.15 GB/s; this loop runs at memory speed.
template <typename real>
static void psp_unpack(long long n, const unsigned char *rec, size_t rec_size, int indexed, int niatr, int ndatr,
                       unsigned long long *indx, double *mass, double *pos, double *vel, double *pot, int *iattrib,
                       double *dattrib)
{
  for (long long i = 0; i < n; i++) {
    const unsigned char *p = rec + (size_t)i * rec_size;
    if (indexed) { memcpy(&indx[i], p, 8); p += 8; }
    real v[8];
    memcpy(v, p, 8 * sizeof(real)); p += 8 * sizeof(real);
    mass[i] = v[0];
    for (int k = 0; k < 3; k++) { pos[3 * i + k] = v[1 + k]; vel[3 * i + k] = v[4 + k]; }
    pot[i] = v[7];
    if (niatr) { memcpy(iattrib + (size_t)i * niatr, p, 4 * (size_t)niatr); p += 4 * (size_t)niatr; }
    for (int k = 0; k < ndatr; k++) { real d; memcpy(&d, p, sizeof(real)); p += sizeof(real); dattrib[(size_t)i * ndatr + k] = d; }
  }
}

extern "C" int exp_amd_host_psp_unpack(long long n, const void *rec, long long rec_size, int r_size, int indexed, int niatr,
                                       int ndatr, unsigned long long *indx, double *mass, double *pos, double *vel,
                                       double *pot, int *iattrib, double *dattrib)
{
  if (n < 0 || (r_size != 4 && r_size != 8) || niatr < 0 || ndatr < 0) return EXP_AMD_ERR_ARG;
  const long long need = (indexed ? 8 : 0) + 8LL * r_size + 4LL * niatr + (long long)r_size * ndatr;
  if (rec_size < need) return EXP_AMD_ERR_ARG;
  if (n > 0 && (!rec || !mass || !pos || !vel || !pot || (indexed && !indx) || (niatr && !iattrib) || (ndatr && !dattrib)))
    return EXP_AMD_ERR_ARG;
  if (r_size == 4) psp_unpack<float>(n, (const unsigned char *)rec, (size_t)rec_size, indexed, niatr, ndatr, indx, mass, pos, vel, pot, iattrib, dattrib);
  else psp_unpack<double>(n, (const unsigned char *)rec, (size_t)rec_size, indexed, niatr, ndatr, indx, mass, pos, vel, pot, iattrib, dattrib);
  return EXP_AMD_OK;
}
