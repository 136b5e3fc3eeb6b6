/*
 * h5part.c -- the HDF5 phase-space files of EXP and Gadget through the HDF5 C library (host side, no GPU code): what
 * exp_amd/reader_h5.py reads and writes.  Part of exp_amd/libexp_amd_h5.so.
 *
 * Files:
 *   OutHDF5 (src/OutHDF5.cc:400-560 "gadget4" style, :645-780 PSP style; Component::write_HDF5 / write_H5,
 *   src/Component.cc:2456-2690) -- groups Header {MassTable, NumPart_ThisFile, Time, Flag_DoublePrecision, HubbleParam,
 *   Omega0, OmegaBaryon, OmegaLambda, Redshift, NumFilesPerSnapshot, NumPart_Total}, Config {PSPstyle, NTYPES,
 *   DOUBLEPRECISION, Niattrib, Ndattrib}, Parameters {Git_commit, Git_branch, Compile_date, ComponentNames, ForceMethods,
 *   ForceConfigurations[, EXPConfiguration]} and one PartType<k> per component holding either the Gadget-4 datasets
 *   (Masses?, ParticleIDs?, Coordinates, Velocities, Potential, PotentialExt, IntAttributes?, RealAttributes?) or ONE
 *   compound dataset "particles" {id: native int at the offset of an unsigned long, mass, pos[3], vel[3], pot, potext,
 *   iattrib: vlen int, dattrib: vlen real} (exputil/ParticleReader.cc:981-993, :1079-1104);
 *   Gadget HDF5 snapshots (exputil/ParticleReader.cc:361-660): Header {Time, MassTable, NumPart_ThisFile}, PartType<k>
 *   {Coordinates, Velocities, Masses?, ParticleIDs}.
 *
 * Every entry point takes the file's path and opens / closes it: these are bulk calls, a handful per file.
 */
#include <hdf5.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static hid_t open_obj(hid_t file, const char *obj)
{
  if (!obj || !obj[0] || (obj[0] == '/' && !obj[1])) return H5Gopen2(file, "/", H5P_DEFAULT);
  return H5Oopen(file, obj, H5P_DEFAULT);
}

static void quiet(void) { H5Eset_auto2(H5E_DEFAULT, NULL, NULL); }

/* 1: the link exists, 0: it does not, <0: the file cannot be opened */
int exp_h5p_exists(const char *path, const char *obj)
{
  if (!path || !obj || !obj[0]) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  /* every component of the path must exist for H5Lexists */
  char buf[1024];
  strncpy(buf, obj, sizeof(buf) - 1); buf[sizeof(buf) - 1] = '\0';
  int ok = 1;
  for (char *p = buf[0] ? buf + 1 : buf; ok && (p = strchr(p, '/')); p++) {
    *p = '\0';
    if (H5Lexists(f, buf, H5P_DEFAULT) <= 0) ok = 0;
    *p = '/';
  }
  if (ok && H5Lexists(f, buf, H5P_DEFAULT) <= 0) ok = 0;
  H5Fclose(f);
  return ok;
}

/* numeric attribute (scalar or 1-D, any integer or float type) -> doubles */
int exp_h5p_attr_f64(const char *path, const char *obj, const char *name, double *out, int cap, int *n)
{
  if (!path || !obj || !name || !out || !n || cap < 0) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t o = open_obj(f, obj);
  if (o >= 0) {
    if (H5Aexists(o, name) > 0) {
      hid_t a = H5Aopen(o, name, H5P_DEFAULT);
      hid_t s = H5Aget_space(a);
      hssize_t np = H5Sget_simple_extent_npoints(s);
      *n = (int)np;
      if (np <= cap) rc = H5Aread(a, H5T_NATIVE_DOUBLE, out) < 0 ? -3 : 0;
      else rc = -4;
      H5Sclose(s); H5Aclose(a);
    }
    H5Oclose(o);
  }
  H5Fclose(f);
  return rc;
}

/* string attribute, scalar or 1-D, variable- or fixed-length: element `index` -> out; *count = number of elements */
int exp_h5p_attr_str(const char *path, const char *obj, const char *name, int index, char *out, int cap, int *count)
{
  if (!path || !obj || !name || !count || !out || cap < 1) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  out[0] = '\0';
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t o = open_obj(f, obj);
  if (o >= 0) {
    if (H5Aexists(o, name) > 0) {
      hid_t a = H5Aopen(o, name, H5P_DEFAULT);
      hid_t s = H5Aget_space(a);
      hid_t ft = H5Aget_type(a);
      hssize_t np = H5Sget_simple_extent_npoints(s);
      *count = (int)np;
      if (index >= 0 && index < np) {
        if (H5Tis_variable_str(ft) > 0) {
          char **p = (char **)calloc((size_t)np, sizeof(char *));
          hid_t mt = H5Tcopy(H5T_C_S1);
          H5Tset_size(mt, H5T_VARIABLE);
          H5Tset_cset(mt, H5Tget_cset(ft));
          if (H5Aread(a, mt, p) >= 0) {
            if (p[index]) { strncpy(out, p[index], (size_t)cap - 1); out[cap - 1] = '\0'; }
            rc = 0;
            H5Dvlen_reclaim(mt, s, H5P_DEFAULT, p);
          } else rc = -3;
          H5Tclose(mt);
          free(p);
        } else {
          size_t w = H5Tget_size(ft);
          char *buf = (char *)calloc((size_t)np * w + 1, 1);
          hid_t mt = H5Tcopy(H5T_C_S1);
          H5Tset_size(mt, w);
          if (H5Aread(a, mt, buf) >= 0) {
            size_t len = w < (size_t)cap - 1 ? w : (size_t)cap - 1;
            memcpy(out, buf + (size_t)index * w, len); out[len] = '\0';
            rc = 0;
          } else rc = -3;
          H5Tclose(mt);
          free(buf);
        }
      } else rc = np == 0 ? 0 : -4;
      H5Tclose(ft); H5Sclose(s); H5Aclose(a);
    }
    H5Oclose(o);
  }
  H5Fclose(f);
  return rc;
}

int exp_h5p_dset_shape(const char *path, const char *dset, int *rank, long long *dims, long long *storage)
{
  if (!path || !dset || !rank || !dims) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t d = H5Dopen2(f, dset, H5P_DEFAULT);
  if (d >= 0) {
    hid_t s = H5Dget_space(d);
    hsize_t dm[8] = {0};
    int r = H5Sget_simple_extent_ndims(s);
    if (r >= 0 && r <= 4) {
      H5Sget_simple_extent_dims(s, dm, NULL);
      *rank = r;
      for (int k = 0; k < 4; k++) dims[k] = k < r ? (long long)dm[k] : 1;
      *storage = (long long)H5Dget_storage_size(d);
      rc = 0;
    }
    H5Sclose(s); H5Dclose(d);
  }
  H5Fclose(f);
  return rc;
}

static hid_t mem_type(char kind)
{
  switch (kind) {
  case 'd': return H5T_NATIVE_DOUBLE;
  case 'f': return H5T_NATIVE_FLOAT;
  case 'i': return H5T_NATIVE_INT;
  case 'u': return H5T_NATIVE_UINT;
  case 'l': return H5T_NATIVE_LONG;
  case 'L': return H5T_NATIVE_ULONG;
  }
  return -1;
}

/* the whole dataset converted to `kind` */
int exp_h5p_dset_read(const char *path, const char *dset, char kind, void *out)
{
  if (!path || !dset || !out) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t mt = mem_type(kind);
  if (mt < 0) return -5;
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t d = H5Dopen2(f, dset, H5P_DEFAULT);
  if (d >= 0) {
    rc = H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0 ? -3 : 0;
    H5Dclose(d);
  }
  H5Fclose(f);
  return rc;
}

/* template <typename T> struct H5Particle (exputil/ParticleReader.cc:981-993, src/Component.cc:2563-2575) */
typedef struct { unsigned long id; double mass, pos[3], vel[3], pot, potext; hvl_t iattrib, dattrib; } part_d;
typedef struct { unsigned long id; float  mass, pos[3], vel[3], pot, potext; hvl_t iattrib, dattrib; } part_f;

static hid_t particle_type(int real4)
{
  hid_t real = real4 ? H5T_NATIVE_FLOAT : H5T_NATIVE_DOUBLE;
  hsize_t three[1] = {3};
  hid_t a3 = H5Tarray_create2(real, 1, three);
  hid_t vi = H5Tvlen_create(H5T_NATIVE_INT);
  hid_t vr = H5Tvlen_create(real);
  hid_t t;
#define MEMBERS(S)                                                                          \
  t = H5Tcreate(H5T_COMPOUND, sizeof(S));                                                   \
  H5Tinsert(t, "id", HOFFSET(S, id), H5T_NATIVE_INT);   /* as the reference declares it */  \
  H5Tinsert(t, "mass", HOFFSET(S, mass), real);                                             \
  H5Tinsert(t, "pos", HOFFSET(S, pos), a3);                                                 \
  H5Tinsert(t, "vel", HOFFSET(S, vel), a3);                                                 \
  H5Tinsert(t, "pot", HOFFSET(S, pot), real);                                               \
  H5Tinsert(t, "potext", HOFFSET(S, potext), real);                                         \
  H5Tinsert(t, "iattrib", HOFFSET(S, iattrib), vi);                                         \
  H5Tinsert(t, "dattrib", HOFFSET(S, dattrib), vr);
  if (real4) { MEMBERS(part_f) } else { MEMBERS(part_d) }
#undef MEMBERS
  H5Tclose(a3); H5Tclose(vi); H5Tclose(vr);
  return t;
}

/* the compound dataset "particles" of a PSP-style file -> arrays (reals widened to double) */
int exp_h5p_particles_read(const char *path, const char *dset, int real4, long long n, int niatr, int ndatr,
                           long long *id, double *mass, double *pos, double *vel, double *pot, double *potext,
                           int *iattrib, double *dattrib)
{
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t d = H5Dopen2(f, dset, H5P_DEFAULT);
  if (d >= 0) {
    hid_t s = H5Dget_space(d);
    hsize_t dm[1] = {0};
    H5Sget_simple_extent_dims(s, dm, NULL);
    if ((long long)dm[0] != n) rc = -6;
    else {
      hid_t t = particle_type(real4);
      const size_t sz = real4 ? sizeof(part_f) : sizeof(part_d);
      char *buf = (char *)calloc((size_t)n + 1, sz);
      if (H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) >= 0) {
        for (long long i = 0; i < n; i++) {
          hvl_t *vi, *vd;
          if (real4) {
            part_f *p = (part_f *)(buf + (size_t)i * sz);
            id[i] = (long long)p->id; mass[i] = p->mass; pot[i] = p->pot; potext[i] = p->potext;
            for (int k = 0; k < 3; k++) { pos[3 * i + k] = p->pos[k]; vel[3 * i + k] = p->vel[k]; }
            vi = &p->iattrib; vd = &p->dattrib;
            for (int j = 0; j < ndatr; j++) dattrib[i * ndatr + j] = (size_t)j < vd->len ? ((float *)vd->p)[j] : 0.0;
          } else {
            part_d *p = (part_d *)(buf + (size_t)i * sz);
            id[i] = (long long)p->id; mass[i] = p->mass; pot[i] = p->pot; potext[i] = p->potext;
            for (int k = 0; k < 3; k++) { pos[3 * i + k] = p->pos[k]; vel[3 * i + k] = p->vel[k]; }
            vi = &p->iattrib; vd = &p->dattrib;
            for (int j = 0; j < ndatr; j++) dattrib[i * ndatr + j] = (size_t)j < vd->len ? ((double *)vd->p)[j] : 0.0;
          }
          for (int j = 0; j < niatr; j++) iattrib[i * niatr + j] = (size_t)j < vi->len ? ((int *)vi->p)[j] : 0;
        }
        H5Dvlen_reclaim(t, s, H5P_DEFAULT, buf);
        rc = 0;
      } else rc = -3;
      free(buf);
      H5Tclose(t);
    }
    H5Sclose(s); H5Dclose(d);
  }
  H5Fclose(f);
  return rc;
}

/* ---- writing ---- */
int exp_h5p_create(const char *path)
{
  if (!path) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (f < 0) return -1;
  H5Fclose(f);
  return 0;
}

int exp_h5p_group(const char *path, const char *group)
{
  if (!path || !group || !group[0]) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return -1;
  hid_t g = H5Gcreate2(f, group, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int rc = g < 0 ? -2 : 0;
  if (g >= 0) H5Gclose(g);
  H5Fclose(f);
  return rc;
}

/* n < 0: scalar; otherwise a 1-D attribute of n elements.  The file type is the native type, as HighFive / H5:: write it. */
int exp_h5p_attr_write(const char *path, const char *obj, const char *name, char kind, int n, const void *data)
{
  if (!path || !obj || !name || n < -1 || (n != 0 && !data)) return -9;   /* n = -1: a scalar */     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t mt = mem_type(kind);
  if (mt < 0) return -5;
  hid_t f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t o = open_obj(f, obj);
  if (o >= 0) {
    hsize_t dm[1] = {(hsize_t)(n < 0 ? 1 : n)};
    hid_t s = n < 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(1, dm, NULL);
    hid_t a = H5Acreate2(o, name, mt, s, H5P_DEFAULT, H5P_DEFAULT);
    if (a >= 0) { rc = (n == 0 || H5Awrite(a, mt, data) >= 0) ? 0 : -3; H5Aclose(a); }
    H5Sclose(s); H5Oclose(o);
  }
  H5Fclose(f);
  return rc;
}

int exp_h5p_attr_write_str(const char *path, const char *obj, const char *name, int n, const char *const *vals)
{
  if (!path || !obj || !name || n < -1 || (n != 0 && !vals)) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = -2;
  hid_t o = open_obj(f, obj);
  if (o >= 0) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);
    hsize_t dm[1] = {(hsize_t)(n < 0 ? 1 : n)};
    hid_t s = n < 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(1, dm, NULL);
    hid_t a = H5Acreate2(o, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    if (a >= 0) { rc = (n == 0 || H5Awrite(a, t, vals) >= 0) ? 0 : -3; H5Aclose(a); }
    H5Sclose(s); H5Tclose(t); H5Oclose(o);
  }
  H5Fclose(f);
  return rc;
}

static hid_t creation_props(int rank, const hsize_t *dims, int chunk, int deflate, int shuffle)
{
  /* src/Component.cc:2504-2540: chunk clamped to [1, nbodies] (nbodies / 8 when the request is not below nbodies); no
   * filter at all for an empty dataset */
  hid_t p = H5Pcreate(H5P_DATASET_CREATE);
  if ((chunk > 0 || deflate > 0) && dims[0] > 0) {
    long long c = chunk;
    if (c >= (long long)dims[0]) c = (long long)dims[0] / 8;
    if (c < 1) c = 1;
    if (c > (long long)dims[0]) c = (long long)dims[0];
    hsize_t cd[4];
    for (int k = 0; k < rank; k++) cd[k] = k == 0 ? (hsize_t)c : dims[k];
    H5Pset_chunk(p, rank, cd);
    if (shuffle) H5Pset_shuffle(p);
    if (deflate > 0) H5Pset_deflate(p, (unsigned)deflate);
  }
  return p;
}

int exp_h5p_dset_write(const char *path, const char *dset, char kind, int rank, const long long *dims, const void *data,
                       int chunk, int deflate, int shuffle)
{
  if (!path || !dset || rank < 0 || (rank > 0 && !dims) || !data) return -9;     /* (C-ABI entry point: arguments are checked before anything is touched) */
  quiet();
  hid_t mt = mem_type(kind);
  if (mt < 0 || rank < 1 || rank > 4) return -5;
  hid_t f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return -1;
  hsize_t dm[4];
  for (int k = 0; k < rank; k++) dm[k] = (hsize_t)dims[k];
  hid_t s = H5Screate_simple(rank, dm, NULL);
  hid_t p = creation_props(rank, dm, chunk, deflate, shuffle);
  hid_t d = H5Dcreate2(f, dset, mt, s, H5P_DEFAULT, p, H5P_DEFAULT);
  int rc = -2;
  if (d >= 0) {
    int empty = 0;
    for (int k = 0; k < rank; k++) if (dm[k] == 0) empty = 1;
    rc = (empty || H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) >= 0) ? 0 : -3;
    H5Dclose(d);
  }
  H5Pclose(p); H5Sclose(s); H5Fclose(f);
  return rc;
}

/* Component::write_H5<T> (src/Component.cc:2590-2690): one compound dataset of n particles */
int exp_h5p_particles_write(const char *path, const char *dset, int real4, long long n, int niatr, int ndatr,
                            const long long *id, const double *mass, const double *pos, const double *vel,
                            const double *pot, const double *potext, const int *iattrib, const double *dattrib,
                            int chunk, int deflate, int shuffle)
{
  quiet();
  hid_t f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return -1;
  const size_t sz = real4 ? sizeof(part_f) : sizeof(part_d);
  char *buf = (char *)calloc((size_t)n + 1, sz);
  float *fattr = real4 && ndatr ? (float *)malloc(sizeof(float) * (size_t)n * ndatr + 1) : NULL;
  for (long long i = 0; i < n; i++) {
    hvl_t *vi, *vd;
    if (real4) {
      part_f *p = (part_f *)(buf + (size_t)i * sz);
      p->id = (unsigned long)id[i]; p->mass = (float)mass[i]; p->pot = (float)pot[i]; p->potext = (float)potext[i];
      for (int k = 0; k < 3; k++) { p->pos[k] = (float)pos[3 * i + k]; p->vel[k] = (float)vel[3 * i + k]; }
      vi = &p->iattrib; vd = &p->dattrib;
      for (int j = 0; j < ndatr; j++) fattr[i * ndatr + j] = (float)dattrib[i * ndatr + j];
      vd->p = ndatr ? (void *)(fattr + i * ndatr) : NULL;
    } else {
      part_d *p = (part_d *)(buf + (size_t)i * sz);
      p->id = (unsigned long)id[i]; p->mass = mass[i]; p->pot = pot[i]; p->potext = potext[i];
      for (int k = 0; k < 3; k++) { p->pos[k] = pos[3 * i + k]; p->vel[k] = vel[3 * i + k]; }
      vi = &p->iattrib; vd = &p->dattrib;
      vd->p = ndatr ? (void *)(dattrib + i * ndatr) : NULL;
    }
    vi->len = (size_t)niatr; vi->p = niatr ? (void *)(iattrib + i * niatr) : NULL;
    vd->len = (size_t)ndatr;
  }
  hid_t t = particle_type(real4);
  hsize_t dm[1] = {(hsize_t)n};
  hid_t s = H5Screate_simple(1, dm, NULL);
  hid_t p = creation_props(1, dm, chunk, deflate, shuffle);
  hid_t d = H5Dcreate2(f, dset, t, s, H5P_DEFAULT, p, H5P_DEFAULT);
  int rc = -2;
  if (d >= 0) {
    rc = (n == 0 || H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) >= 0) ? 0 : -3;
    H5Dclose(d);
  }
  H5Pclose(p); H5Sclose(s); H5Tclose(t); H5Fclose(f);
  free(buf); free(fattr);
  return rc;
}
