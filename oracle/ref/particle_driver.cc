// Driver for the REFERENCE's own particle-record and header code, compiled where it lies (nothing is copied):
//   exputil/Particle.cc   Particle::writeBinary / readBinary / readAscii / writeAscii  (the PSP record, the body file)
//   exputil/header.cc     ComponentHeader::write / read                               (the PSP component header)
//   exputil/libvars.cc, exputil/localmpi.cc   the globals those use (__EXP__::multistep; numprocs, myid)
//   include/tipsy.H       TipsyReader::TipsyNative                                    (Tipsy native files, rank blocks)
// They need <mpi.h> and libmpi, which this image has under /opt/conda (the HDF5 of the image is built on it), and
// nothing else.  Test infrastructure only: tests/test_ref_particle.py pins exp_amd/reader.py and oracle/psp_oracle.c
// against them byte for byte.
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Particle.H"
#include "header.H"
#include "tipsy.H"

// MasterHeader + per component: magic, ComponentHeader::write, Particle::writeBinary -- the calls of OutPSN::Run /
// Component::write_binary (src/OutPSN.cc:143-169, src/Component.cc:2385-2454) with the reference's own functions
extern "C" int ref_psp_write(const char *path, double time, int ncomp, const int *nbod, const int *niatr, const int *ndatr,
                             const char *const *info, const int *indexing, int real4, const unsigned long *indx,
                             const double *mass, const double *pos, const double *vel, const double *pot,
                             const double *potext, const int *iattrib, const double *dattrib)
{
  std::ofstream out(path, std::ios::binary);
  if (!out) return -1;
  MasterHeader master;
  master.time = time; master.ntot = 0; master.ncomp = ncomp;
  for (int c = 0; c < ncomp; c++) master.ntot += nbod[c];
  out.write((char *)&master, sizeof(MasterHeader));
  const unsigned long magic = 0xadbfabc0;                       // src/Component.H
  long p0 = 0, ia0 = 0, da0 = 0;
  for (int c = 0; c < ncomp; c++) {
    ComponentHeader header;                                     // ninfochar = defaultInfoSize
    header.nbod = nbod[c]; header.niatr = niatr[c]; header.ndatr = ndatr[c];
    std::string s(info[c]);
    if ((size_t)header.ninfochar < s.size()) {                  // src/Component.cc:2399-2408
      header.ninfochar = s.size();
      header.info = std::shared_ptr<char>(new char[header.ninfochar + 1], std::default_delete<char[]>());
      std::fill(header.info.get(), header.info.get() + header.ninfochar + 1, 0);
    }
    strncpy(header.info.get(), s.c_str(), header.ninfochar);
    unsigned rsize = real4 ? sizeof(float) : sizeof(double);
    unsigned long cmagic = magic + rsize;
    out.write((const char *)&cmagic, sizeof(unsigned long));
    if (!header.write(&out)) return -2;
    for (long i = 0; i < nbod[c]; i++) {
      Particle p(niatr[c], ndatr[c]);
      const long q = p0 + i;
      p.indx = indx[q]; p.mass = mass[q]; p.pot = pot[q]; p.potext = potext[q];
      for (int k = 0; k < 3; k++) { p.pos[k] = pos[3 * q + k]; p.vel[k] = vel[3 * q + k]; }
      for (int k = 0; k < niatr[c]; k++) p.iattrib[k] = iattrib[ia0 + i * niatr[c] + k];
      for (int k = 0; k < ndatr[c]; k++) p.dattrib[k] = dattrib[da0 + i * ndatr[c] + k];
      p.writeBinary(rsize, indexing[c], &out);
    }
    p0 += nbod[c]; ia0 += (long)nbod[c] * niatr[c]; da0 += (long)nbod[c] * ndatr[c];
  }
  return out ? 0 : -3;
}

// One component of a PSP file read back with ComponentHeader::read and Particle::readBinary (the restart path,
// src/Component.cc read_bodies_and_distribute_binary_out): component `which`, its header fields and particles
extern "C" long ref_psp_read(const char *path, int which, int indexing_of_each[], int *niatr, int *ndatr, int *ninfochar,
                             char *info, int info_cap, unsigned long *rsize_out, unsigned long *indx, double *mass,
                             double *pos, double *vel, double *pot, int *iattrib, double *dattrib, long cap)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) return -1;
  MasterHeader master;
  in.read((char *)&master, sizeof(MasterHeader));
  if (!in) return -2;
  const unsigned long magic = 0xadbfabc0, mmask = 0xf, nmask = ~mmask;
  for (int c = 0; c < master.ncomp; c++) {
    unsigned long cmagic;
    in.read((char *)&cmagic, sizeof(unsigned long));
    unsigned long rsize = sizeof(double);
    if ((cmagic & nmask) == magic) rsize = cmagic & mmask;
    ComponentHeader header;
    if (!header.read(&in)) return -3;
    const bool idx = indexing_of_each[c] != 0;
    if (c != which) {
      in.seekg((std::streamoff)header.nbod * ((idx ? 8 : 0) + 8 * rsize + header.niatr * sizeof(int) + header.ndatr * rsize), std::ios::cur);
      continue;
    }
    *niatr = header.niatr; *ndatr = header.ndatr; *ninfochar = header.ninfochar; *rsize_out = rsize;
    strncpy(info, header.info.get(), info_cap - 1); info[info_cap - 1] = 0;
    if (header.nbod > cap) return -4;
    for (long i = 0; i < header.nbod; i++) {
      Particle p(header.niatr, header.ndatr);
      p.readBinary(rsize, idx, i + 1, &in);                      // seq = i (1-based) without an index in the file
      indx[i] = p.indx; mass[i] = p.mass; pot[i] = p.pot;
      for (int k = 0; k < 3; k++) { pos[3 * i + k] = p.pos[k]; vel[3 * i + k] = p.vel[k]; }
      for (int k = 0; k < header.niatr; k++) iattrib[i * header.niatr + k] = p.iattrib[k];
      for (int k = 0; k < header.ndatr; k++) dattrib[i * header.ndatr + k] = p.dattrib[k];
    }
    return in ? header.nbod : -5;
  }
  return -6;
}

// A body file read line by line with Particle::readAscii (src/Component.cc:1479-1526)
extern "C" long ref_bodies_read(const char *path, int aindex, long cap, int *niatr, int *ndatr, unsigned long *indx,
                                double *mass, double *pos, double *vel, int *iattrib, double *dattrib, int attr_cap)
{
  std::ifstream fin(path);
  if (!fin) return -1;
  const int nline = 2048;
  char line[nline];
  fin.getline(line, nline);
  std::istringstream ins(line);
  long nbodies_tot = 0; int ni = 0, nd = 0;
  ins >> nbodies_tot;
  if (!ins) return -2;
  ins >> ni;
  if (!ins) ni = 0;
  ins >> nd;
  if (!ins) nd = 0;
  if (nbodies_tot > cap || ni > attr_cap || nd > attr_cap) return -3;
  *niatr = ni; *ndatr = nd;
  for (long i = 1; i <= nbodies_tot; i++) {
    Particle p(ni, nd);
    p.readAscii(aindex != 0, i, &fin);
    const long q = i - 1;
    indx[q] = p.indx; mass[q] = p.mass;
    for (int k = 0; k < 3; k++) { pos[3 * q + k] = p.pos[k]; vel[3 * q + k] = p.vel[k]; }
    for (int k = 0; k < ni; k++) iattrib[q * ni + k] = p.iattrib[k];
    for (int k = 0; k < nd; k++) dattrib[q * nd + k] = p.dattrib[k];
  }
  return nbodies_tot;
}

// TipsyReader::TipsyNative for rank `rank` of `nranks` -> what it holds of the group `ptype` (0 gas, 1 dark, 2 star):
// count, index offset, masses, positions, velocities, and the two trailing floats (eps / metals.., phi) of each record
extern "C" long ref_tipsy_read(const char *path, int nranks, int rank, int ptype, double *time, unsigned long *offset,
                               float *mass, float *pos, float *vel, float *phi, long cap)
{
  numprocs = nranks; myid = rank;
  try {
    TipsyReader::TipsyNative ps(path);
    ps.readParticles();
    *time = ps.header.time;
    long n = 0;
    if (ptype == 0) {
      n = ps.gas_particles.size(); *offset = ps.getIndexOffset(TipsyReader::Ptype::gas);
      if (n > cap) return -2;
      for (long i = 0; i < n; i++) { auto &p = ps.gas_particles[i]; mass[i] = p.mass; phi[i] = p.phi; for (int k = 0; k < 3; k++) { pos[3 * i + k] = p.pos[k]; vel[3 * i + k] = p.vel[k]; } }
    } else if (ptype == 1) {
      n = ps.dark_particles.size(); *offset = ps.getIndexOffset(TipsyReader::Ptype::dark);
      if (n > cap) return -2;
      for (long i = 0; i < n; i++) { auto &p = ps.dark_particles[i]; mass[i] = p.mass; phi[i] = p.phi; for (int k = 0; k < 3; k++) { pos[3 * i + k] = p.pos[k]; vel[3 * i + k] = p.vel[k]; } }
    } else {
      n = ps.star_particles.size(); *offset = ps.getIndexOffset(TipsyReader::Ptype::star);
      if (n > cap) return -2;
      for (long i = 0; i < n; i++) { auto &p = ps.star_particles[i]; mass[i] = p.mass; phi[i] = p.phi; for (int k = 0; k < 3; k++) { pos[3 * i + k] = p.pos[k]; vel[3 * i + k] = p.vel[k]; } }
    }
    numprocs = 1; myid = 0;
    return n;
  } catch (std::exception &e) {
    numprocs = 1; myid = 0;
    return -1;
  }
}

// Bonsai ids as the reference forms them (include/tipsy.H: dark_particle::ID, ID2)
extern "C" void ref_tipsy_ids(float eps, float phi, int *id, unsigned long *id2)
{
  TipsyReader::dark_particle p;
  p.eps = eps; p.phi = phi;
  *id = p.ID(); *id2 = p.ID2();
}
