/*
 * psp_oracle.c -- TEST INFRASTRUCTURE ONLY (see bfe_oracle.h): CPU restatement of the reference's phase-space file
 * format and of the particle histograms that read it.  Only tests/ may call it.
 *
 * The reference's READERS (exputil/ParticleReader.cc) need yaml-cpp and HighFive, which this image lacks, so their logic
 * is restated here statement by statement.  The RECORD code (exputil/Particle.cc, exputil/header.cc, include/tipsy.H) does
 * compile -- it needs only <mpi.h>, present under /opt/conda -- into oracle/_ref/libref_particle.so, and
 * tests/test_ref_particle.py checks this restatement's writer byte for byte against it; tests/test_ref_pspformat.py
 * checks order, widths and constants against the reference's SOURCE TEXT as well.
 *
 *   orc_psp_write            OutPSN::Run (src/OutPSN.cc:141-169) -> Component::write_binary (src/Component.cc:2385-2454)
 *                            -> ComponentHeader::write (exputil/header.cc:7-19), Particle::writeBinary
 *                            (exputil/Particle.cc:333-388): one out->write per field, in that order
 *   orc_psp_scan/_read       PSPout::PSPout (exputil/ParticleReader.cc:1298-1469), PParticle::read
 *                            (include/ParticleReader.H:276-315), PSPout::nextParticle (:1689-1735: stagger by myid,
 *                            stride numprocs)
 *   orc_histo2d/1d/1dlog     FieldGenerator::histogram2d / histogram1d / histo1dlog (expui/FieldGenerator.cc:776-1009):
 *                            float accumulators, one particle at a time, in reader order
 *   orc_outlog_sums/_row     the particle loop and the data row of OutLog::Run (src/OutLog.cc:392-446, :480-590)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const unsigned long psp_magic = 0xadbfabc0;     /* include/ParticleReader.H:340 */
static const unsigned long psp_mmask = 0xf;

typedef struct {
  int nbod, niatr, ndatr, ninfochar;
  unsigned long r_size, index_size;
  long pspos;                     /* offset of the first particle */
  char info[8192];
} orc_psp_stanza;

/* Particle::writeBinary (exputil/Particle.cc:333-388) */
static void write_particle(FILE *out, unsigned rsize, int indexing, unsigned long indx, double mass, const double *pos,
                           const double *vel, double pot, double potext, int niatr, const int *iattrib, int ndatr,
                           const double *dattrib)
{
  float tf;
  if (indexing) fwrite(&indx, sizeof(unsigned long), 1, out);
  if (rsize == sizeof(float)) { tf = (float)mass; fwrite(&tf, sizeof(float), 1, out); }
  else fwrite(&mass, sizeof(double), 1, out);
  for (int i = 0; i < 3; i++) {
    double pv = pos[i];
    if (rsize == sizeof(float)) { tf = (float)pv; fwrite(&tf, sizeof(float), 1, out); }
    else fwrite(&pv, sizeof(double), 1, out);
  }
  for (int i = 0; i < 3; i++) {
    double pv = vel[i];
    if (rsize == sizeof(float)) { tf = (float)pv; fwrite(&tf, sizeof(float), 1, out); }
    else fwrite(&pv, sizeof(double), 1, out);
  }
  double pot0 = pot + potext;
  if (rsize == sizeof(float)) { tf = (float)pot0; fwrite(&tf, sizeof(float), 1, out); }
  else fwrite(&pot0, sizeof(double), 1, out);
  for (int k = 0; k < niatr; k++) fwrite(&iattrib[k], sizeof(int), 1, out);
  for (int k = 0; k < ndatr; k++) {
    if (rsize == sizeof(float)) { tf = (float)dattrib[k]; fwrite(&tf, sizeof(float), 1, out); }
    else fwrite(&dattrib[k], sizeof(double), 1, out);
  }
}

/* One file with `ncomp` components.  Per component c: nbod[c] particles, info[c] (the YAML stanza, NUL-terminated),
 * indexing[c]; the particle arrays of all components are concatenated (pos / vel [n][3], iattrib [n][niatr[c]], ...). */
int orc_psp_write(const char *path, double time, int ncomp, const int *nbod, const int *niatr, const int *ndatr,
                  const char *const *info, const int *indexing, int real4, const unsigned long *indx,
                  const double *mass, const double *pos, const double *vel, const double *pot, const double *potext,
                  const int *iattrib, const double *dattrib)
{
  FILE *out = fopen(path, "wb");
  if (!out) return -1;
  /* struct MasterHeader {double time; int ntot; int ncomp;} written whole (src/OutPSN.cc:143-148) */
  struct { double time; int ntot; int ncomp; } master;
  master.time = time; master.ntot = 0; master.ncomp = ncomp;
  for (int c = 0; c < ncomp; c++) master.ntot += nbod[c];
  fwrite(&master, sizeof(master), 1, out);
  long p0 = 0, ia0 = 0, da0 = 0;
  for (int c = 0; c < ncomp; c++) {
    /* ComponentHeader(): ninfochar = defaultInfoSize = 1024, info zero-filled; grown to the stanza's size when that is
     * longer (src/Component.cc:2399-2411) */
    int ninfochar = 1024;
    size_t len = strlen(info[c]);
    if ((size_t)ninfochar < len) ninfochar = (int)len;
    char *buf = (char *)calloc((size_t)ninfochar + 1, 1);
    strncpy(buf, info[c], (size_t)ninfochar);
    unsigned rsize = real4 ? sizeof(float) : sizeof(double);
    unsigned long cmagic = psp_magic + rsize;
    fwrite(&cmagic, sizeof(unsigned long), 1, out);
    fwrite(&nbod[c], sizeof(int), 1, out);               /* ComponentHeader::write (exputil/header.cc:9-13) */
    fwrite(&niatr[c], sizeof(int), 1, out);
    fwrite(&ndatr[c], sizeof(int), 1, out);
    fwrite(&ninfochar, sizeof(int), 1, out);
    fwrite(buf, 1, (size_t)ninfochar, out);
    free(buf);
    for (long i = 0; i < nbod[c]; i++) {
      const long p = p0 + i;
      write_particle(out, rsize, indexing[c], indx ? indx[p] : (unsigned long)(i + 1), mass[p], pos + 3 * p, vel + 3 * p,
                     pot[p], potext ? potext[p] : 0.0, niatr[c], iattrib ? iattrib + ia0 + i * niatr[c] : NULL,
                     ndatr[c], dattrib ? dattrib + da0 + i * ndatr[c] : NULL);
    }
    p0 += nbod[c];
    ia0 += (long)nbod[c] * niatr[c];
    da0 += (long)nbod[c] * ndatr[c];
  }
  fclose(out);
  return 0;
}

/* PSPout::PSPout without the YAML parse: the caller says which stanzas are indexed (the reader takes that from the info
 * string) -> number of stanzas found */
int orc_psp_scan(const char *path, double *time, int *ntot, int cap, orc_psp_stanza *st, const int *indexing)
{
  FILE *in = fopen(path, "rb");
  if (!in) return -1;
  struct { double time; int ntot; int ncomp; } master;
  if (fread(&master, sizeof(master), 1, in) != 1) { fclose(in); return -2; }
  *time = master.time; *ntot = master.ntot;
  int found = 0;
  for (int i = 0; i < master.ncomp && i < cap; i++) {
    unsigned long ret;
    if (fread(&ret, sizeof(unsigned long), 1, in) != 1) break;
    unsigned long rsize = sizeof(double);
    if ((ret & ~psp_mmask) == psp_magic) rsize = ret & psp_mmask;
    orc_psp_stanza *s = st + i;
    if (fread(&s->nbod, sizeof(int), 1, in) != 1) break;
    if (fread(&s->niatr, sizeof(int), 1, in) != 1) break;
    if (fread(&s->ndatr, sizeof(int), 1, in) != 1) break;
    if (fread(&s->ninfochar, sizeof(int), 1, in) != 1) break;
    memset(s->info, 0, sizeof(s->info));
    if (s->ninfochar >= (int)sizeof(s->info)) { fclose(in); return -3; }
    if (fread(s->info, 1, (size_t)s->ninfochar, in) != (size_t)s->ninfochar) break;
    s->pspos = ftell(in);
    s->r_size = rsize;
    s->index_size = indexing[i] ? sizeof(unsigned long) : 0;
    fseek(in, (long)s->nbod * (long)(s->index_size + 8 * s->r_size + s->niatr * sizeof(int) + s->ndatr * s->r_size), SEEK_CUR);
    found++;
  }
  fclose(in);
  return found;
}

/* firstParticle / nextParticle of one stanza for rank `myid` of `numprocs` -> number of particles returned */
long orc_psp_read(const char *path, const orc_psp_stanza *s, int numprocs, int myid, unsigned long *indx, double *mass,
                  double *pos, double *vel, double *pot, int *iattrib, double *dattrib)
{
  FILE *in = fopen(path, "rb");
  if (!in) return -1;
  fseek(in, s->pspos, SEEK_SET);
  const long skipsize = (long)(8 * s->r_size + s->niatr * sizeof(int) + s->ndatr * s->r_size + s->index_size);
  long pcount = 0, got = 0;
  for (int n = 0; n < myid; n++)
    if (pcount < s->nbod) { fseek(in, skipsize, SEEK_CUR); pcount++; }
  while (pcount < s->nbod) {
    unsigned long id = (unsigned long)pcount;          /* indx = pcount when the file holds none (:283) */
    if (s->index_size) { if (fread(&id, sizeof(unsigned long), 1, in) != 1) break; }
    double v[8];
    if (s->r_size == 4) {
      float f[8];
      if (fread(f, sizeof(float), 8, in) != 8) break;
      for (int k = 0; k < 8; k++) v[k] = f[k];
    } else if (fread(v, sizeof(double), 8, in) != 8) break;
    indx[got] = id;
    mass[got] = v[0];
    for (int k = 0; k < 3; k++) { pos[3 * got + k] = v[1 + k]; vel[3 * got + k] = v[4 + k]; }
    pot[got] = v[7];
    for (int k = 0; k < s->niatr; k++)
      if (fread(&iattrib[got * s->niatr + k], sizeof(int), 1, in) != 1) break;
    for (int k = 0; k < s->ndatr; k++) {
      if (s->r_size == 4) { float f; if (fread(&f, sizeof(float), 1, in) != 1) break; dattrib[got * s->ndatr + k] = f; }
      else if (fread(&dattrib[got * s->ndatr + k], sizeof(double), 1, in) != 1) break;
    }
    pcount++; got++;
    for (int n = 0; n < numprocs - 1; n++)
      if (pcount < s->nbod) { fseek(in, skipsize, SEEK_CUR); pcount++; }
  }
  fclose(in);
  return got;
}

/* FieldGenerator::histogram2d (expui/FieldGenerator.cc:776-855): out = xy [g0][g1], xz [g0][g2], yz [g1][g2] (each only
 * when both grid sizes are positive), float, row-major */
void orc_histo2d(long n, const double *mass, const double *pos, const double *ctr, const double *pmin, const double *pmax,
                 const int *grid, float *xy, float *xz, float *yz)
{
  double del[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < 3; k++) if (grid[k] > 0) del[k] = (pmax[k] - pmin[k]) / grid[k];
  const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  float *out[3] = {xy, xz, yz};
  double fac[3] = {0.0, 0.0, 0.0};
  for (int q = 0; q < 3; q++)
    if (grid[pairs[q][0]] > 0 && grid[pairs[q][1]] > 0) fac[q] = 1.0 / (del[pairs[q][0]] * del[pairs[q][1]]);
  for (long i = 0; i < n; i++) {
    double pp[3]; int bb[3];
    for (int k = 0; k < 3; k++) {
      pp[k] = pos[3 * i + k] - ctr[k];
      bb[k] = pp[k] >= pmin[k] && pp[k] < pmax[k] && del[k] > 0.0;
    }
    for (int q = 0; q < 3; q++) {
      const int a = pairs[q][0], b = pairs[q][1];
      if (!(grid[a] > 0 && grid[b] > 0)) continue;
      if (bb[a] && bb[b]) {
        int i1 = (int)floor((pp[a] - pmin[a]) / del[a]);
        int i2 = (int)floor((pp[b] - pmin[b]) / del[b]);
        if (i1 >= 0 && i1 < grid[a] && i2 >= 0 && i2 < grid[b])
          out[q][(long)i1 * grid[b] + i2] += mass[i] * fac[q];       /* float += double */
      }
    }
  }
}

/* FieldGenerator::histogram1d (:857-920); proj: 0 xy, 1 xz, 2 yz, 3 r */
void orc_histo1d(long n, const double *mass, const double *pos, const double *ctr, double rmax, int nbins, int proj,
                 float *ret)
{
  const double pi = 3.14159265358979323846;
  double del = rmax / nbins;
  for (int i = 0; i < nbins; i++) ret[i] = 0.0f;
  for (long i = 0; i < n; i++) {
    double rad = 0.0;
    for (int k = 0; k < 3; k++) {
      double pp = pos[3 * i + k] - ctr[k];
      if (proj == 0 && (k == 0 || k == 1)) rad += pp * pp;
      else if (proj == 1 && (k == 0 || k == 2)) rad += pp * pp;
      else if (proj == 2 && (k == 1 || k == 2)) rad += pp * pp;
      else if (proj == 3) rad += pp * pp;
    }
    int indx = (int)floor(sqrt(rad) / del);
    if (indx >= 0 && indx < nbins) ret[indx] += mass[i];
  }
  for (int i = 0; i < nbins; i++) {
    if (proj == 3) ret[i] /= 4.0 * pi / 3.0 * del * del * del * (3 * i * (i + 1) + 1);
    else ret[i] /= pi * del * del * (2 * i + 1);
  }
}

/* FieldGenerator::histo1dlog (:922-1009) -> rad, ret (density), vel (dispersion), float [nbins] each */
void orc_histo1dlog(long n, const double *mass, const double *pos, const double *velo, const double *ctr, double rmin,
                    double rmax, int nbins, float *rad, float *ret, float *vel)
{
  const double pi = 3.14159265358979323846;
  float *vc1 = (float *)calloc((size_t)nbins * 3, sizeof(float));
  float *vc2 = (float *)calloc((size_t)nbins * 3, sizeof(float));
  for (int i = 0; i < nbins; i++) rad[i] = ret[i] = vel[i] = 0.0f;
  double lrmin = log(rmin), lrmax = log(rmax);
  double del = (lrmax - lrmin) / nbins;
  for (long i = 0; i < n; i++) {
    double r2 = 0.0;
    for (int k = 0; k < 3; k++) { double pp = pos[3 * i + k] - ctr[k]; r2 += pp * pp; }
    int indx = (int)floor((log(sqrt(r2)) - lrmin) / del);
    if (indx >= 0 && indx < nbins) {
      ret[indx] += mass[i];
      for (int k = 0; k < 3; k++) {
        double v = velo[3 * i + k];
        vc1[k * nbins + indx] += mass[i] * v;           /* Eigen::MatrixXf(nbins, 3): column-major, element (indx, k) */
        vc2[k * nbins + indx] += mass[i] * v * v;
      }
    }
  }
  double d3 = exp(3.0 * del);
  double rf = 4.0 * pi / 3.0 * (d3 - 1.0);
  for (int i = 0; i < nbins; i++) {
    double sig = 0.0;
    if (ret[i] > 0) {
      for (int k = 0; k < 3; k++) {
        vc1[k * nbins + i] /= ret[i];
        vc2[k * nbins + i] /= ret[i];
        sig += vc2[k * nbins + i] - vc1[k * nbins + i] * vc1[k * nbins + i];   /* double += float - float*float */
      }
      rad[i] = exp(lrmin + del * (0.5 + i));
      ret[i] /= exp(3.0 * (lrmin + del * i)) * rf;
      vel[i] = sqrt(fabs(sig));
    } else {
      rad[i] = exp(lrmin + del * (0.5 + i));
      ret[i] = 0.0f;
      vel[i] = 0.0f;
    }
  }
  free(vc1); free(vc2);
}

/* The particle loop of OutLog::Run (src/OutLog.cc:392-446) for one component with com_system off and no frozen
 * particle: out = {mtot, com[3], cov[3], angm[3], ektot, eptot, clausius} (sums, not yet divided) */
void orc_outlog_sums(long n, const double *mass, const double *pos, const double *vel, const double *acc,
                     const double *pot, double *out)
{
  double mtot1 = 0.0, com1[3] = {0, 0, 0}, cov1[3] = {0, 0, 0}, angm1[3] = {0, 0, 0};
  double ektot1 = 0.0, eptot1 = 0.0, clausius1 = 0.0;
  for (long i = 0; i < n; i++) {
    const double m = mass[i];
    const double *posL = pos + 3 * i, *velL = vel + 3 * i;
    mtot1 += m;
    for (int k = 0; k < 3; k++) {
      com1[k] += m * posL[k];
      cov1[k] += m * velL[k];
    }
    angm1[0] += m * (posL[1] * velL[2] - posL[2] * velL[1]);
    angm1[1] += m * (posL[2] * velL[0] - posL[0] * velL[2]);
    angm1[2] += m * (posL[0] * velL[1] - posL[1] * velL[0]);
    eptot1 += 0.5 * m * pot[i];
    for (int k = 0; k < 3; k++) {
      ektot1 += 0.5 * m * velL[k] * velL[k];
      clausius1 += m * posL[k] * acc[3 * i + k];
    }
  }
  out[0] = mtot1;
  for (int k = 0; k < 3; k++) { out[1 + k] = com1[k]; out[4 + k] = cov1[k]; out[7 + k] = angm1[k]; }
  out[10] = ektot1; out[11] = eptot1; out[12] = clausius1;
}

/* The data row of OutLog::Run (src/OutLog.cc:480-590) written with the C library's %*.*e -- the same characters as
 * `out << std::scientific << setprecision(p) << setw(w)`.  sums: [ncomp][13] as above, nbodies, used, ctr [ncomp][3]. */
int orc_outlog_row(char *buf, int cap, double tnow, int ncomp, const double *sums, const int *nbodies, const int *used,
                   const double *ctr, double wtime, int precision)
{
  const int cwid = 10 + precision;
  int o = 0;
#define PUT_D(v) o += snprintf(buf + o, (size_t)(cap - o), "%s%*.*e", o ? "|" : "", cwid, precision, (double)(v))
#define PUT_I(v) o += snprintf(buf + o, (size_t)(cap - o), "|%*d", cwid, (int)(v))
  PUT_D(tnow);
  double mtot0 = 0.0;
  for (int i = 0; i < ncomp; i++) mtot0 += sums[13 * i];
  PUT_D(mtot0);
  int nbodies0 = 0;
  for (int i = 0; i < ncomp; i++) nbodies0 += nbodies[i];
  PUT_I(nbodies0);
  double com0[3] = {0, 0, 0}, cov0[3] = {0, 0, 0}, angm0[3] = {0, 0, 0};
  for (int i = 0; i < ncomp; i++)
    for (int j = 0; j < 3; j++) { com0[j] += sums[13 * i + 1 + j]; cov0[j] += sums[13 * i + 4 + j]; angm0[j] += sums[13 * i + 7 + j]; }
  for (int j = 0; j < 3; j++) PUT_D(mtot0 > 0.0 ? com0[j] / mtot0 : 0.0);
  for (int j = 0; j < 3; j++) PUT_D(mtot0 > 0.0 ? cov0[j] / mtot0 : 0.0);
  for (int j = 0; j < 3; j++) PUT_D(angm0[j]);
  double ektot0 = 0.0, eptot0 = 0.0, clausius0 = 0.0;
  for (int i = 0; i < ncomp; i++) ektot0 += sums[13 * i + 10];
  PUT_D(ektot0);
  for (int i = 0; i < ncomp; i++) eptot0 += sums[13 * i + 11] + 0.0;
  PUT_D(eptot0);
  for (int i = 0; i < ncomp; i++) clausius0 += sums[13 * i + 12];
  PUT_D(clausius0);
  PUT_D(ektot0 + clausius0);
  PUT_D(clausius0 != 0.0 ? -2.0 * ektot0 / clausius0 : 0.0);
  PUT_D(wtime);
  int usedT = 0;
  for (int i = 0; i < ncomp; i++) usedT += used[i];
  PUT_I(usedT);
  for (int i = 0; i < ncomp; i++) {
    const double *s = sums + 13 * i;
    PUT_D(s[0]);
    PUT_I(nbodies[i]);
    for (int j = 0; j < 3; j++) PUT_D(s[0] > 0.0 ? s[1 + j] / s[0] : 0.0);
    for (int j = 0; j < 3; j++) PUT_D(s[0] > 0.0 ? s[4 + j] / s[0] : 0.0);
    for (int j = 0; j < 3; j++) PUT_D(s[7 + j]);
    for (int j = 0; j < 3; j++) PUT_D(ctr[3 * i + j]);
    double vbar2 = 0.0;
    if (s[0] > 0.0) {
      for (int j = 0; j < 3; j++) vbar2 += s[4 + j] * s[4 + j];
      vbar2 /= s[0] * s[0];
    }
    double ek = s[10];
    if (nbodies[i] > 1) ek -= 0.5 * s[0] * vbar2;
    PUT_D(ek);
    PUT_D(s[11] + 0.0);
    PUT_D(s[12]);
    PUT_D(ek + s[12]);
    PUT_D(s[12] != 0.0 ? -2.0 * ek / s[12] : 0.0);
    PUT_I(used[i]);
  }
  o += snprintf(buf + o, (size_t)(cap - o), "\n");
#undef PUT_D
#undef PUT_I
  return o;
}
