/*
 * h5cache.c -- EXP's HDF5 basis-cache files through the HDF5 C library (host side, no GPU code).
 *
 * SLGridSph cache, as written / read by SLGridSph::WriteH5Cache / ReadH5Cache
 * (exputil/SLGridMP2.cc:490-696): root attributes
 *     geometry = "sphere", forceID = "SLGridSph", Version = "1.0" (include/SLGridMP2.H:95), model
 *     (strings: HighFive std::string -> variable-length UTF-8), lmax, nmax, numr, cmap, diverge
 *     (native int), rmin, rmax, rmapping, dfac (native double; "scale" is accepted for "rmapping"
 *     on reading, :577-581)
 * and, per harmonic order l, the group Harmonic/<l> with the datasets
 *     ev  [nmax]          eigenvalues
 *     ef  [nmax][numr]    eigenfunctions (Eigen::MatrixXd(nmax, numr) serialised row-major by
 *                         HighFive's Eigen inspector -- the layout the "Version" attribute pins;
 *                         a [numr][nmax] dataset from the older API is transposed on reading).
 * The reference compares the header with its own parameters and silently rebuilds on mismatch;
 * here the header is returned to the caller.  Built only where hdf5.h is available
 * (exp_amd/libexp_amd_h5.so); no HighFive, no C++.
 */
#include <stddef.h>
#include <hdf5.h>
#include <unistd.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  char geometry[64], forceID[64], version[32], model[512];
  int lmax, nmax, numr, cmap, diverge;
  double rmin, rmax, rmapping, dfac;
} exp_h5_slgrid_hdr;

static int put_str(hid_t loc, const char *name, const char *val)
{
  hid_t t = H5Tcopy(H5T_C_S1);
  H5Tset_size(t, H5T_VARIABLE);
  H5Tset_cset(t, H5T_CSET_UTF8);
  hid_t s = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (a < 0) ? -1 : (H5Awrite(a, t, &val) < 0 ? -1 : 0);
  if (a >= 0) H5Aclose(a);
  H5Sclose(s);
  H5Tclose(t);
  return rc;
}

static int put_int(hid_t loc, const char *name, int v)
{
  hid_t s = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(loc, name, H5T_NATIVE_INT, s, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (a < 0) ? -1 : (H5Awrite(a, H5T_NATIVE_INT, &v) < 0 ? -1 : 0);
  if (a >= 0) H5Aclose(a);
  H5Sclose(s);
  return rc;
}

static int put_dbl(hid_t loc, const char *name, double v)
{
  hid_t s = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(loc, name, H5T_NATIVE_DOUBLE, s, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (a < 0) ? -1 : (H5Awrite(a, H5T_NATIVE_DOUBLE, &v) < 0 ? -1 : 0);
  if (a >= 0) H5Aclose(a);
  H5Sclose(s);
  return rc;
}

static int get_str(hid_t loc, const char *name, char *out, size_t cap)
{
  out[0] = '\0';
  if (H5Aexists(loc, name) <= 0) return -1;
  hid_t a = H5Aopen(loc, name, H5P_DEFAULT);
  if (a < 0) return -1;
  hid_t ft = H5Aget_type(a);
  int rc = 0;
  if (H5Tis_variable_str(ft) > 0) {
    char *p = NULL;
    hid_t mt = H5Tcopy(H5T_C_S1);
    H5Tset_size(mt, H5T_VARIABLE);
    H5Tset_cset(mt, H5Tget_cset(ft));
    if (H5Aread(a, mt, &p) < 0 || !p) rc = -1;
    else { strncpy(out, p, cap - 1); out[cap - 1] = '\0'; H5free_memory(p); }
    H5Tclose(mt);
  } else {
    size_t n = H5Tget_size(ft);
    char *buf = (char *)calloc(n + 1, 1);
    hid_t mt = H5Tcopy(H5T_C_S1);
    H5Tset_size(mt, n);
    if (H5Aread(a, mt, buf) < 0) rc = -1;
    else { strncpy(out, buf, cap - 1); out[cap - 1] = '\0'; }
    H5Tclose(mt);
    free(buf);
  }
  H5Tclose(ft);
  H5Aclose(a);
  return rc;
}

static int get_int(hid_t loc, const char *name, int *v)
{
  if (H5Aexists(loc, name) <= 0) return -1;
  hid_t a = H5Aopen(loc, name, H5P_DEFAULT);
  int rc = (a < 0 || H5Aread(a, H5T_NATIVE_INT, v) < 0) ? -1 : 0;
  if (a >= 0) H5Aclose(a);
  return rc;
}

static int get_dbl(hid_t loc, const char *name, double *v)
{
  if (H5Aexists(loc, name) <= 0) return -1;
  hid_t a = H5Aopen(loc, name, H5P_DEFAULT);
  int rc = (a < 0 || H5Aread(a, H5T_NATIVE_DOUBLE, v) < 0) ? -1 : 0;
  if (a >= 0) H5Aclose(a);
  return rc;
}

static int put_array(hid_t loc, const char *name, int rank, const hsize_t *dims, const double *data)
{
  hid_t s = H5Screate_simple(rank, dims, NULL);
  hid_t d = H5Dcreate2(loc, name, H5T_NATIVE_DOUBLE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (d < 0) ? -1 : (H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? -1 : 0);
  if (d >= 0) H5Dclose(d);
  H5Sclose(s);
  return rc;
}

/* exputil/SLGridMP2.cc:622-696 */
int exp_h5_slgrid_write_ex(const char *path, const exp_h5_slgrid_hdr *h, const double *ev,
                           const double *ef, int old_layout);

int exp_h5_slgrid_write(const char *path, const exp_h5_slgrid_hdr *h, const double *ev,
                        const double *ef)
{
  return exp_h5_slgrid_write_ex(path, h, ev, ef, 0);
}

/* old_layout != 0: ef as [numr][nmax] (what pre-"Version" caches hold), for reader tests */
int exp_h5_slgrid_write_ex(const char *path, const exp_h5_slgrid_hdr *h, const double *ev,
                           const double *ef, int old_layout)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= put_str(f, "geometry", "sphere");
  rc |= put_str(f, "forceID", "SLGridSph");
  rc |= put_str(f, "Version", h->version[0] ? h->version : "1.0");
  rc |= put_str(f, "model", h->model);
  rc |= put_int(f, "lmax", h->lmax);
  rc |= put_int(f, "nmax", h->nmax);
  rc |= put_int(f, "numr", h->numr);
  rc |= put_int(f, "cmap", h->cmap);
  rc |= put_dbl(f, "rmin", h->rmin);
  rc |= put_dbl(f, "rmax", h->rmax);
  rc |= put_dbl(f, "rmapping", h->rmapping);
  rc |= put_int(f, "diverge", h->diverge);
  rc |= put_dbl(f, "dfac", h->dfac);
  hid_t harm = H5Gcreate2(f, "Harmonic", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  if (harm < 0) rc = -1;
  for (int l = 0; l <= h->lmax && rc == 0; l++) {
    char nm[32];
    snprintf(nm, sizeof nm, "%d", l);
    hid_t g = H5Gcreate2(harm, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) { rc = -1; break; }
    hsize_t d1[1] = {(hsize_t)h->nmax};
    hsize_t d2[2] = {(hsize_t)h->nmax, (hsize_t)h->numr};
    rc |= put_array(g, "ev", 1, d1, ev + (size_t)l * h->nmax);
    const double *src = ef + (size_t)l * h->nmax * h->numr;
    if (!old_layout) rc |= put_array(g, "ef", 2, d2, src);
    else {
      double *t = (double *)malloc(sizeof(double) * (size_t)h->nmax * h->numr);
      for (int n = 0; n < h->nmax; n++)
        for (int i = 0; i < h->numr; i++) t[(size_t)i * h->nmax + n] = src[(size_t)n * h->numr + i];
      hsize_t d3[2] = {(hsize_t)h->numr, (hsize_t)h->nmax};
      rc |= put_array(g, "ef", 2, d3, t);
      free(t);
    }
    H5Gclose(g);
  }
  if (harm >= 0) H5Gclose(harm);
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* exputil/SLGridMP2.cc:490-620: header only (0 = ok, -1 = not an SLGridSph cache / unreadable) */
int exp_h5_slgrid_read_header(const char *path, exp_h5_slgrid_hdr *h)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  memset(h, 0, sizeof *h);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= get_str(f, "geometry", h->geometry, sizeof h->geometry);
  rc |= get_str(f, "forceID", h->forceID, sizeof h->forceID);
  get_str(f, "Version", h->version, sizeof h->version);          /* absent in pre-1.0 caches */
  rc |= get_str(f, "model", h->model, sizeof h->model);
  rc |= get_int(f, "lmax", &h->lmax);
  rc |= get_int(f, "nmax", &h->nmax);
  rc |= get_int(f, "numr", &h->numr);
  rc |= get_int(f, "cmap", &h->cmap);
  rc |= get_dbl(f, "rmin", &h->rmin);
  rc |= get_dbl(f, "rmax", &h->rmax);
  if (get_dbl(f, "rmapping", &h->rmapping) != 0) rc |= get_dbl(f, "scale", &h->rmapping);
  rc |= get_int(f, "diverge", &h->diverge);
  rc |= get_dbl(f, "dfac", &h->dfac);
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* tables: ev[(lmax+1)][nmax], ef[(lmax+1)][nmax][numr] */
int exp_h5_slgrid_read_tables(const char *path, int lmax, int nmax, int numr, double *ev, double *ef)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  double *tmp = (double *)malloc(sizeof(double) * (size_t)nmax * numr);
  for (int l = 0; l <= lmax && rc == 0; l++) {
    char nm[64];
    snprintf(nm, sizeof nm, "Harmonic/%d/ev", l);
    hid_t d = H5Dopen2(f, nm, H5P_DEFAULT);
    if (d < 0) { rc = -1; break; }
    {
      hid_t s1 = H5Dget_space(d);
      if (H5Sget_simple_extent_npoints(s1) != (hssize_t)nmax) rc = -1;   /* never over-read */
      H5Sclose(s1);
    }
    if (rc == 0 && H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, ev + (size_t)l * nmax) < 0) rc = -1;
    H5Dclose(d);
    if (rc) break;
    snprintf(nm, sizeof nm, "Harmonic/%d/ef", l);
    d = H5Dopen2(f, nm, H5P_DEFAULT);
    if (d < 0) { rc = -1; break; }
    hid_t s = H5Dget_space(d);
    hsize_t dims[2] = {0, 0};
    if (H5Sget_simple_extent_ndims(s) != 2) rc = -1;
    else H5Sget_simple_extent_dims(s, dims, NULL);
    double *dst = ef + (size_t)l * nmax * numr;
    if (rc == 0 && dims[0] == (hsize_t)nmax && dims[1] == (hsize_t)numr) {
      if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) < 0) rc = -1;
    } else if (rc == 0 && dims[0] == (hsize_t)numr && dims[1] == (hsize_t)nmax) {
      if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, tmp) < 0) rc = -1;
      else
        for (int n = 0; n < nmax; n++)
          for (int i = 0; i < numr; i++) dst[(size_t)n * numr + i] = tmp[(size_t)i * nmax + n];
    } else
      rc = -1;
    H5Sclose(s);
    H5Dclose(d);
  }
  free(tmp);
  H5Fclose(f);
  return rc;
}

/* ---- EmpCylSL cache ------------------------------------------------------------------------------
 * EmpCylSL::WriteH5Cache / ReadH5Cache (exputil/EmpCylSL.cc:7378-7460, :7486-7640): root attributes
 * geometry = "cylinder", forceID = "Cylinder", Version = "1.0", model (e.g. "Exponential"), mmax,
 * numx, numy, nmax (= NORDER), lmaxfid, nmaxfid, neven, nodd, cmapr, cmapz (int), rmin, rmax, ascl,
 * hscl, cmass (double); groups Cosine/<m>/<n> with potC, rforceC, zforceC, densC and (m >= 1)
 * Sine/<m>/<n> with potS, rforceS, zforceS, densS, each a [numx+1][numy+1] matrix.
 * tab[6][mmax+1][norder][numx+1][numy+1] = potC, rforceC, zforceC, potS, rforceS, zforceS;
 * dens[2][mmax+1][norder][numx+1][numy+1] = densC, densS.                                       */
typedef struct {
  char geometry[64], forceID[64], version[32], model[128];
  int mmax, numx, numy, nmax, lmaxfid, nmaxfid, neven, nodd, cmapr, cmapz;
  double rmin, rmax, ascl, hscl, cmass;
} exp_h5_cyl_hdr;

int exp_h5_cyl_write(const char *path, const exp_h5_cyl_hdr *h, const double *tab, const double *dens)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= put_str(f, "geometry", "cylinder");
  rc |= put_str(f, "forceID", "Cylinder");
  rc |= put_str(f, "Version", h->version[0] ? h->version : "1.0");
  rc |= put_str(f, "model", h->model);
  rc |= put_int(f, "mmax", h->mmax);
  rc |= put_int(f, "numx", h->numx);
  rc |= put_int(f, "numy", h->numy);
  rc |= put_int(f, "nmax", h->nmax);
  rc |= put_int(f, "lmaxfid", h->lmaxfid);
  rc |= put_int(f, "nmaxfid", h->nmaxfid);
  rc |= put_int(f, "neven", h->neven);
  rc |= put_int(f, "nodd", h->nodd);
  rc |= put_int(f, "cmapr", h->cmapr);
  rc |= put_int(f, "cmapz", h->cmapz);
  rc |= put_dbl(f, "rmin", h->rmin);
  rc |= put_dbl(f, "rmax", h->rmax);
  rc |= put_dbl(f, "ascl", h->ascl);
  rc |= put_dbl(f, "hscl", h->hscl);
  rc |= put_dbl(f, "cmass", h->cmass);
  const size_t nnode = (size_t)(h->numx + 1) * (h->numy + 1);
  const size_t kind = (size_t)(h->mmax + 1) * h->nmax * nnode;
  hsize_t d2[2] = {(hsize_t)h->numx + 1, (hsize_t)h->numy + 1};
  static const char *cn[4] = {"potC", "rforceC", "zforceC", "densC"};
  static const char *sn[4] = {"potS", "rforceS", "zforceS", "densS"};
  for (int cs = 0; cs < 2 && rc == 0; cs++) {
    hid_t top = H5Gcreate2(f, cs ? "Sine" : "Cosine", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (top < 0) { rc = -1; break; }
    for (int m = cs; m <= h->mmax && rc == 0; m++) {
      char nm[32];
      snprintf(nm, sizeof nm, "%d", m);
      hid_t gm = H5Gcreate2(top, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      for (int n = 0; n < h->nmax && rc == 0; n++) {
        snprintf(nm, sizeof nm, "%d", n);
        hid_t gn = H5Gcreate2(gm, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        const size_t off = ((size_t)m * h->nmax + n) * nnode;
        for (int k = 0; k < 3; k++)
          rc |= put_array(gn, cs ? sn[k] : cn[k], 2, d2, tab + (size_t)(3 * cs + k) * kind + off);
        rc |= put_array(gn, cs ? sn[3] : cn[3], 2, d2, dens + (size_t)cs * kind + off);
        H5Gclose(gn);
      }
      H5Gclose(gm);
    }
    H5Gclose(top);
  }
  H5Fclose(f);
  return rc ? -1 : 0;
}

int exp_h5_cyl_read_header(const char *path, exp_h5_cyl_hdr *h)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  memset(h, 0, sizeof *h);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= get_str(f, "geometry", h->geometry, sizeof h->geometry);
  rc |= get_str(f, "forceID", h->forceID, sizeof h->forceID);
  get_str(f, "Version", h->version, sizeof h->version);
  rc |= get_str(f, "model", h->model, sizeof h->model);
  rc |= get_int(f, "mmax", &h->mmax);
  rc |= get_int(f, "numx", &h->numx);
  rc |= get_int(f, "numy", &h->numy);
  rc |= get_int(f, "nmax", &h->nmax);
  rc |= get_int(f, "lmaxfid", &h->lmaxfid);
  rc |= get_int(f, "nmaxfid", &h->nmaxfid);
  rc |= get_int(f, "neven", &h->neven);
  rc |= get_int(f, "nodd", &h->nodd);
  rc |= get_int(f, "cmapr", &h->cmapr);
  rc |= get_int(f, "cmapz", &h->cmapz);
  rc |= get_dbl(f, "rmin", &h->rmin);
  rc |= get_dbl(f, "rmax", &h->rmax);
  rc |= get_dbl(f, "ascl", &h->ascl);
  rc |= get_dbl(f, "hscl", &h->hscl);
  rc |= get_dbl(f, "cmass", &h->cmass);
  H5Fclose(f);
  return rc ? -1 : 0;
}

static int get_matrix(hid_t f, const char *name, int nx1, int ny1, double *dst)
{
  hid_t d = H5Dopen2(f, name, H5P_DEFAULT);
  if (d < 0) return -1;
  hid_t s = H5Dget_space(d);
  hsize_t dims[2] = {0, 0};
  int rc = 0;
  if (H5Sget_simple_extent_ndims(s) != 2) rc = -1;
  else H5Sget_simple_extent_dims(s, dims, NULL);
  if (rc == 0 && (dims[0] != (hsize_t)nx1 || dims[1] != (hsize_t)ny1)) rc = -1;
  if (rc == 0 && H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) < 0) rc = -1;
  H5Sclose(s);
  H5Dclose(d);
  return rc;
}

int exp_h5_cyl_read_tables(const char *path, int mmax, int nmax, int numx, int numy, double *tab,
                           double *dens)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  const size_t nnode = (size_t)(numx + 1) * (numy + 1);
  const size_t kind = (size_t)(mmax + 1) * nmax * nnode;
  static const char *cn[4] = {"potC", "rforceC", "zforceC", "densC"};
  static const char *sn[4] = {"potS", "rforceS", "zforceS", "densS"};
  int rc = 0;
  for (int cs = 0; cs < 2 && rc == 0; cs++)
    for (int m = cs; m <= mmax && rc == 0; m++)
      for (int n = 0; n < nmax && rc == 0; n++) {
        const size_t off = ((size_t)m * nmax + n) * nnode;
        char nm[96];
        for (int k = 0; k < 4 && rc == 0; k++) {
          snprintf(nm, sizeof nm, "%s/%d/%d/%s", cs ? "Sine" : "Cosine", m, n, cs ? sn[k] : cn[k]);
          double *dst = (k < 3) ? tab + (size_t)(3 * cs + k) * kind + off : dens + (size_t)cs * kind + off;
          rc |= get_matrix(f, nm, numx + 1, numy + 1, dst);
        }
      }
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* ---- spherical coefficient files -------------------------------------------------------------------
 * Coefs::WriteH5Coefs (expui/Coefficients.cc:3100-3163), SphCoefs::WriteH5Params (:841-853),
 * SphCoefs::WriteH5Times (:907-944) and the reading constructor (:228-330): root attributes
 * CoefficientOutputVersion = "1.0" (Coefficients.H:95), geometry = "sphere", name, config, lmax,
 * nmax (int), scale (double), forceID; dataset "count" (unsigned); group snapshots/%08d with the
 * attributes Time (double), Center (double[3]), Rotation (double[3][3]) and the dataset
 * "coefficients" [(L+1)(L+2)/2][nmax] of std::complex<double> (HighFive's compound {r, i}).
 * The optional "Units" dataset is not written (the reader takes it only if present, :171-181).
 * coefs[ntimes][ldim][nmax][2] = (re, im).                                                      */
static hid_t complex_type(void)
{
  hid_t t = H5Tcreate(H5T_COMPOUND, 16);
  H5Tinsert(t, "r", 0, H5T_NATIVE_DOUBLE);
  H5Tinsert(t, "i", 8, H5T_NATIVE_DOUBLE);
  return t;
}

static int put_dbl_array_attr(hid_t loc, const char *name, int rank, const hsize_t *dims, const double *v)
{
  hid_t s = H5Screate_simple(rank, dims, NULL);
  hid_t a = H5Acreate2(loc, name, H5T_NATIVE_DOUBLE, s, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (a < 0) ? -1 : (H5Awrite(a, H5T_NATIVE_DOUBLE, v) < 0 ? -1 : 0);
  if (a >= 0) H5Aclose(a);
  H5Sclose(s);
  return rc;
}

/* Coefs::WriteH5Coefs (expui/Coefficients.cc:3100-3163) with the per-geometry WriteH5Params /
 * WriteH5Times: sphere (:841-853, :907-944: lmax, nmax, scale; (l, m>=0) rows) and cylinder
 * (:1323-1332, :1375-1405: mmax, nmax; m rows).  `ldim` is the number of complex rows.            */
/* Coefs::WriteH5Units / ReadH5Units (expui/Coefficients.cc:20-30, :152-182): dataset "Units", one compound record
 * {char name[16]; char unit[16]; float value;} per unit (HighFive: AtomicType<char[16]> = a 16-byte UTF-8 string).
 * The records for the next coefficient file are handed over with exp_h5_coef_set_units. */
typedef struct { char name[16]; char unit[16]; float value; } exp_h5_unit;
static exp_h5_unit g_units[16];
static int g_nunits = 0;

int exp_h5_coef_set_units(int n, const void *records /* n x 36 bytes */)
{
  if (n < 0 || n > 16) return -1;
  memcpy(g_units, records, (size_t)n * sizeof(exp_h5_unit));
  g_nunits = n;
  return 0;
}

static hid_t unit_type(void)
{
  hid_t st = H5Tcopy(H5T_C_S1);
  H5Tset_size(st, 16);
  H5Tset_cset(st, H5T_CSET_UTF8);
  hid_t ct = H5Tcreate(H5T_COMPOUND, sizeof(exp_h5_unit));
  H5Tinsert(ct, "name", offsetof(exp_h5_unit, name), st);
  H5Tinsert(ct, "unit", offsetof(exp_h5_unit, unit), st);
  H5Tinsert(ct, "value", offsetof(exp_h5_unit, value), H5T_NATIVE_FLOAT);
  H5Tclose(st);
  return ct;
}

static int put_units(hid_t f)
{
  hsize_t dims[1] = {(hsize_t)g_nunits};
  hid_t ct = unit_type(), sp = H5Screate_simple(1, dims, NULL);
  hid_t d = H5Dcreate2(f, "Units", ct, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int rc = 0;
  if (d < 0 || (g_nunits && H5Dwrite(d, ct, H5S_ALL, H5S_ALL, H5P_DEFAULT, g_units) < 0)) rc = -1;
  if (d >= 0) H5Dclose(d);
  H5Sclose(sp);
  H5Tclose(ct);
  return rc;
}

/* *n = -1 when the file has no "Units" dataset (older files: the reader keeps its default) */
int exp_h5_coef_read_units(const char *path, int cap, void *records, int *n)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  *n = -1;
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  if (H5Lexists(f, "Units", H5P_DEFAULT) > 0) {
    hid_t d = H5Dopen2(f, "Units", H5P_DEFAULT);
    hid_t sp = d >= 0 ? H5Dget_space(d) : -1;
    hsize_t dims[1] = {0};
    if (sp >= 0 && H5Sget_simple_extent_ndims(sp) == 1) H5Sget_simple_extent_dims(sp, dims, NULL);
    if (d < 0 || (int)dims[0] > cap) rc = -1;
    else {
      hid_t ct = unit_type();
      if (dims[0] && H5Dread(d, ct, H5S_ALL, H5S_ALL, H5P_DEFAULT, records) < 0) rc = -1;
      else *n = (int)dims[0];
      H5Tclose(ct);
    }
    if (sp >= 0) H5Sclose(sp);
    if (d >= 0) H5Dclose(d);
  }
  H5Fclose(f);
  return rc;
}

static int coef_write(const char *path, const char *geometry, const char *name, const char *config,
                      const char *forceID, const char *key1, int val1, int nmax, int has_scale,
                      double scale, int ldim, int ntimes, const double *times, const double *centers,
                      const double *rots, const double *coefs, int extend)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  /* extend != 0: Coefs::ExtendH5Coefs (expui/Coefficients.cc:3165-3204) -- open read-write, continue
   * the snapshot numbering at the stored count, update the count (the parameter check,
   * CheckH5Params, is the caller's) */
  hid_t f = extend ? H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT)
                   : H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  unsigned first = 0;
  hid_t cnt = -1;
  if (!extend) {
    rc |= put_str(f, "CoefficientOutputVersion", "1.0");
    rc |= put_str(f, "geometry", geometry);
    rc |= put_str(f, "name", name);
    rc |= put_str(f, "config", config);
    rc |= put_units(f);
    rc |= put_int(f, key1, val1);
    rc |= put_int(f, "nmax", nmax);
    if (has_scale) rc |= put_dbl(f, "scale", scale);
    rc |= put_str(f, "forceID", forceID);
    hid_t s = H5Screate(H5S_SCALAR);
    cnt = H5Dcreate2(f, "count", H5T_NATIVE_UINT, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Sclose(s);
  } else {
    cnt = H5Dopen2(f, "count", H5P_DEFAULT);
    if (cnt < 0 || H5Dread(cnt, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &first) < 0) rc = -1;
  }
  hid_t snaps = extend ? H5Gopen2(f, "snapshots", H5P_DEFAULT)
                       : H5Gcreate2(f, "snapshots", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  hid_t ct = complex_type();
  for (int k = 0; k < ntimes && rc == 0 && snaps >= 0; k++) {
    char nm[16];
    snprintf(nm, sizeof nm, "%08u", first + (unsigned)k);
    hid_t g = H5Gcreate2(snaps, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) { rc = -1; break; }
    rc |= put_dbl(g, "Time", times[k]);
    hsize_t d3[1] = {3}, d33[2] = {3, 3};
    rc |= put_dbl_array_attr(g, "Center", 1, d3, centers + 3 * k);
    rc |= put_dbl_array_attr(g, "Rotation", 2, d33, rots + 9 * k);
    hsize_t dc[2] = {(hsize_t)ldim, (hsize_t)nmax};
    hid_t s = H5Screate_simple(2, dc, NULL);
    hid_t d = H5Dcreate2(g, "coefficients", ct, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (d < 0 || H5Dwrite(d, ct, H5S_ALL, H5S_ALL, H5P_DEFAULT, coefs + (size_t)k * ldim * nmax * 2) < 0) rc = -1;
    if (d >= 0) H5Dclose(d);
    H5Sclose(s);
    H5Gclose(g);
  }
  H5Tclose(ct);
  if (snaps >= 0) H5Gclose(snaps); else rc = -1;
  if (cnt >= 0) {
    const unsigned count = first + (unsigned)ntimes;
    if (rc == 0 && H5Dwrite(cnt, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &count) < 0) rc = -1;
    H5Dclose(cnt);
  } else
    rc = -1;
  H5Fclose(f);
  return rc ? -1 : 0;
}

int exp_h5_sphcoef_write(const char *path, const char *name, const char *config, const char *forceID,
                         int lmax, int nmax, double scale, int ntimes, const double *times,
                         const double *centers, const double *rots, const double *coefs)
{
  return coef_write(path, "sphere", name, config, forceID, "lmax", lmax, nmax, 1, scale,
                    (lmax + 1) * (lmax + 2) / 2, ntimes, times, centers, rots, coefs, 0);
}

/* Coefs::ExtendH5Coefs for either geometry: ldim complex rows per snapshot */
int exp_h5_coef_extend(const char *path, int ldim, int nmax, int ntimes, const double *times,
                       const double *centers, const double *rots, const double *coefs)
{
  return coef_write(path, "", "", "", "", "", 0, nmax, 0, 0.0, ldim, ntimes, times, centers, rots, coefs, 1);
}

int exp_h5_cylcoef_write(const char *path, const char *name, const char *config, const char *forceID,
                         int mmax, int nmax, int ntimes, const double *times, const double *centers,
                         const double *rots, const double *coefs)
{
  return coef_write(path, "cylinder", name, config, forceID, "mmax", mmax, nmax, 0, 0.0, mmax + 1,
                    ntimes, times, centers, rots, coefs, 0);
}

/* Coefs::factory's first question (expui/Coefficients.cc:2917-2931): is this an HDF5 file with a
 * `geometry` attribute?  0 and the string, or -1.                                                */
int exp_h5_coef_geometry(const char *path, char *geometry, int cap)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = get_str(f, "geometry", geometry, (size_t)cap);
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* header of a cylindrical coefficient file: CylCoefs(HighFive::File&, ...) (expui/Coefficients.cc:1075-1095) */
int exp_h5_cylcoef_info(const char *path, int *mmax, int *nmax, int *count, char *name, int name_cap,
                        char *forceID, int id_cap, char *geometry, int geo_cap, int *has_version)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= get_int(f, "mmax", mmax);
  rc |= get_int(f, "nmax", nmax);
  get_str(f, "name", name, (size_t)name_cap);
  get_str(f, "forceID", forceID, (size_t)id_cap);
  rc |= get_str(f, "geometry", geometry, (size_t)geo_cap);
  *has_version = H5Aexists(f, "CoefficientOutputVersion") > 0;
  unsigned c = 0;
  hid_t d = H5Dopen2(f, "count", H5P_DEFAULT);
  if (d < 0 || H5Dread(d, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &c) < 0) rc = -1;
  if (d >= 0) H5Dclose(d);
  *count = (int)c;
  H5Fclose(f);
  return rc ? -1 : 0;
}

int exp_h5_sphcoef_info(const char *path, int *lmax, int *nmax, double *scale, int *count,
                        char *name, int name_cap, char *forceID, int id_cap, char *geometry,
                        int geo_cap, int *has_version)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= get_int(f, "lmax", lmax);
  rc |= get_int(f, "nmax", nmax);
  rc |= get_dbl(f, "scale", scale);
  get_str(f, "name", name, (size_t)name_cap);
  get_str(f, "forceID", forceID, (size_t)id_cap);
  rc |= get_str(f, "geometry", geometry, (size_t)geo_cap);
  *has_version = H5Aexists(f, "CoefficientOutputVersion") > 0;
  unsigned c = 0;
  hid_t d = H5Dopen2(f, "count", H5P_DEFAULT);
  if (d < 0 || H5Dread(d, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &c) < 0) rc = -1;
  if (d >= 0) H5Dclose(d);
  *count = (int)c;
  H5Fclose(f);
  return rc ? -1 : 0;
}

static int coef_read(const char *path, int count, int ldim, int nmax, double *times, double *centers,
                     double *rots, double *coefs);

int exp_h5_sphcoef_read(const char *path, int count, int lmax, int nmax, double *times,
                        double *centers, double *rots, double *coefs)
{
  return coef_read(path, count, (lmax + 1) * (lmax + 2) / 2, nmax, times, centers, rots, coefs);
}

/* the snapshots of a cylindrical file, (mmax+1) x nmax complex each (expui/Coefficients.cc:1100-1170) */
int exp_h5_cylcoef_read(const char *path, int count, int mmax, int nmax, double *times,
                        double *centers, double *rots, double *coefs)
{
  return coef_read(path, count, mmax + 1, nmax, times, centers, rots, coefs);
}

static int coef_read(const char *path, int count, int ldim, int nmax, double *times, double *centers,
                     double *rots, double *coefs)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  hid_t ct = complex_type();
  int rc = 0;
  for (int k = 0; k < count && rc == 0; k++) {
    char nm[64];
    snprintf(nm, sizeof nm, "snapshots/%08d", k);
    hid_t g = H5Gopen2(f, nm, H5P_DEFAULT);
    if (g < 0) { rc = -1; break; }
    rc |= get_dbl(g, "Time", times + k);
    double *c3 = centers + 3 * k, *r9 = rots + 9 * k;
    c3[0] = c3[1] = c3[2] = 0.0;
    for (int q = 0; q < 9; q++) r9[q] = (q % 4 == 0) ? 1.0 : 0.0;
    if (H5Aexists(g, "Center") > 0) {
      hid_t a = H5Aopen(g, "Center", H5P_DEFAULT);
      hid_t s = H5Aget_space(a);
      if (H5Sget_simple_extent_npoints(s) == 3) H5Aread(a, H5T_NATIVE_DOUBLE, c3);
      H5Sclose(s); H5Aclose(a);
    }
    if (H5Aexists(g, "Rotation") > 0) {
      hid_t a = H5Aopen(g, "Rotation", H5P_DEFAULT);
      hid_t s = H5Aget_space(a);
      if (H5Sget_simple_extent_npoints(s) == 9) H5Aread(a, H5T_NATIVE_DOUBLE, r9);
      H5Sclose(s); H5Aclose(a);
    }
    hid_t d = H5Dopen2(g, "coefficients", H5P_DEFAULT);
    if (d < 0) rc = -1;
    else {
      hid_t s = H5Dget_space(d);
      hsize_t dims[2] = {0, 0};
      if (H5Sget_simple_extent_ndims(s) != 2) rc = -1;
      else H5Sget_simple_extent_dims(s, dims, NULL);
      if (rc == 0 && (dims[0] != (hsize_t)ldim || dims[1] != (hsize_t)nmax)) rc = -1;
      if (rc == 0 && H5Dread(d, ct, H5S_ALL, H5S_ALL, H5P_DEFAULT, coefs + (size_t)k * ldim * nmax * 2) < 0) rc = -1;
      H5Sclose(s);
      H5Dclose(d);
    }
    H5Gclose(g);
  }
  H5Tclose(ct);
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* ---- coefficient covariance store ----------------------------------------------------------------
 * SubsampleCovariance::writeCoefCovariance / extendCoefCovariance / writeCovarH5
 * (expui/Covariance.cc:17-417; include/Covariance.H) with the per-basis parameters of
 * Spherical::writeCovarH5Params / Cylindrical::writeCovarH5Params (expui/BiorthBasis.cc:5192-5210),
 * file version "1.1", FloatSize 8.  Root: attributes CovarianceFileVersion, BasisID, FloatSize and
 * the basis parameters; dataset `count` (unsigned); group snapshots/%08d with attributes Time
 * (rounded to 1e-8), sampleSize, angularSize, rankSize and the datasets sampleCounts (int),
 * sampleMasses, coefficients_real / _imag [T*ltot*nmax] and either covariance_real_total / _imag_total
 * [ltot*nmax(nmax+1)/2] (summed over the sub-samples, upper triangles, the default) or covariance_real /
 * _imag per sub-sample (upper triangles if `covar`, else diagonals).  Eigen vectors are stored by
 * HighFive as [n][1] datasets, chunked, shuffled and deflated (level 5) as the reference does.
 *   kind 0: sphere    ipar = {lmax, nmax},  dpar = {scale, rmin, rmax}
 *   kind 1: cylinder  ipar = {mmax, nmax},  dpar = {rcylmin, rcylmax, acyl, bias, hcyl}
 * mean[T][ltot][nmax][2], covr[T][ltot][nmax][nmax][2] (re, im); covr may be NULL (no covariance). */
static int put_uint(hid_t loc, const char *name, unsigned v)
{
  hid_t s = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(loc, name, H5T_NATIVE_UINT, s, H5P_DEFAULT, H5P_DEFAULT);
  int rc = (a < 0) ? -1 : (H5Awrite(a, H5T_NATIVE_UINT, &v) < 0 ? -1 : 0);
  if (a >= 0) H5Aclose(a);
  H5Sclose(s);
  return rc;
}

/* SubsampleCovariance::setCovarH5Compress(level, chunksize, shuffle, szip) (include/Covariance.H:147-153;
 * defaults :60-63 of the class: level 5, chunk 1024*1024, shuffle on): level 0 = no filters, contiguous.
 * szip is refused (the HDF5 library of this image has no szip encoder).                                  */
static unsigned g_covar_level = 5, g_covar_chunk = 1048576;
static int g_covar_shuffle = 1;

int exp_h5_covar_set_compress(unsigned level, unsigned chunksize, int shuffle, int szip)
{
  if (szip || level > 9 || chunksize == 0) return -1;
  g_covar_level = level;
  g_covar_chunk = chunksize;
  g_covar_shuffle = shuffle ? 1 : 0;
  return 0;
}

static int put_vec(hid_t loc, const char *name, hid_t type, size_t n, const void *data)
{
  hsize_t dims[2] = {(hsize_t)n, 1}, chunk[2] = {(hsize_t)(n < g_covar_chunk ? (n ? n : 1) : g_covar_chunk), 1};
  hid_t s = H5Screate_simple(2, dims, NULL);
  hid_t p = H5Pcreate(H5P_DATASET_CREATE);
  if (n && g_covar_level) {
    H5Pset_chunk(p, 2, chunk);
    if (g_covar_shuffle) H5Pset_shuffle(p);
    H5Pset_deflate(p, g_covar_level);
  }
  hid_t d = H5Dcreate2(loc, name, type, s, H5P_DEFAULT, p, H5P_DEFAULT);
  int rc = (d < 0) ? -1 : (H5Dwrite(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? -1 : 0);
  if (d >= 0) H5Dclose(d);
  H5Pclose(p);
  H5Sclose(s);
  return rc;
}

static double covar_round_time(double t) { return floor(t * 1.0e8 + 0.5) / 1.0e8; }

int exp_h5_covar_append(const char *path, const char *basisID, int kind, const int *ipar, const double *dpar,
                        int summed, int covar, double time, int sampT, int ltot, int nmax, const int *counts,
                        const double *masses, const double *mean, const double *covr)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  if (summed) covar = 1;                                   /* expui/Covariance.cc:14 */
  long total = 0;
  for (int t = 0; t < sampT; t++) total += counts[t];
  if (total == 0) return 1;                                /* "no data": nothing is written (:309-312) */
  int rc = 0;
  unsigned count = 0;
  hid_t f = -1, snaps = -1, cd = -1;
  /* expui/Covariance.cc:283-417: the file is opened ReadWrite | Create.  An existing HDF5 file WITHOUT the
   * version attribute gets the attributes and groups added to it (nothing of it is dropped); a path that
   * exists and is not HDF5 is an error, never truncated.                                               */
  htri_t isf = H5Fis_hdf5(path);
  if (isf > 0) {
    f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
    if (f < 0) return -1;
  } else if (access(path, F_OK) == 0) {
    return -2;                                             /* exists, not HDF5 */
  }
  if (f >= 0 && H5Aexists(f, "CovarianceFileVersion") > 0) {          /* extendCoefCovariance */
    cd = H5Dopen2(f, "count", H5P_DEFAULT);
    if (cd < 0 || H5Dread(cd, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &count) < 0) rc = -1;
    snaps = H5Gopen2(f, "snapshots", H5P_DEFAULT);
  } else {
    if (f < 0) f = H5Fcreate(path, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return -1;
    if (H5Lexists(f, "count", H5P_DEFAULT) > 0 || H5Lexists(f, "snapshots", H5P_DEFAULT) > 0) {
      H5Fclose(f);
      return -2;                                           /* a foreign file that already uses these names */
    }
    rc |= put_str(f, "CovarianceFileVersion", "1.1");
    rc |= put_str(f, "BasisID", basisID);
    rc |= put_int(f, "FloatSize", 8);
    if (kind == 0) {
      rc |= put_int(f, "lmax", ipar[0]); rc |= put_int(f, "nmax", ipar[1]);
      rc |= put_dbl(f, "scale", dpar[0]); rc |= put_dbl(f, "rmin", dpar[1]); rc |= put_dbl(f, "rmax", dpar[2]);
    } else {
      rc |= put_int(f, "mmax", ipar[0]); rc |= put_int(f, "nmax", ipar[1]);
      rc |= put_dbl(f, "rcylmin", dpar[0]); rc |= put_dbl(f, "rcylmax", dpar[1]); rc |= put_dbl(f, "acyl", dpar[2]);
      rc |= put_dbl(f, "bias", dpar[3]); rc |= put_dbl(f, "hcyl", dpar[4]);
    }
    hid_t s = H5Screate(H5S_SCALAR);
    cd = H5Dcreate2(f, "count", H5T_NATIVE_UINT, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Sclose(s);
    snaps = H5Gcreate2(f, "snapshots", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  }
  if (snaps < 0 || cd < 0) rc = -1;
  if (rc == 0) {
    char nm[16];
    snprintf(nm, sizeof nm, "%08u", count);
    hid_t g = H5Gcreate2(snaps, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) rc = -1;
    else {
      rc |= put_dbl(g, "Time", covar_round_time(time));
      rc |= put_vec(g, "sampleCounts", H5T_NATIVE_INT, (size_t)sampT, counts);
      rc |= put_vec(g, "sampleMasses", H5T_NATIVE_DOUBLE, (size_t)sampT, masses);
      rc |= put_uint(g, "sampleSize", (unsigned)sampT);
      rc |= put_uint(g, "angularSize", (unsigned)ltot);
      rc |= put_uint(g, "rankSize", (unsigned)nmax);
      const size_t nc = (size_t)sampT * ltot * nmax;
      double *re = (double *)malloc(sizeof(double) * (nc ? nc : 1)), *im = (double *)malloc(sizeof(double) * (nc ? nc : 1));
      for (size_t c = 0; c < nc; c++) { re[c] = mean[2 * c]; im[c] = mean[2 * c + 1]; }
      rc |= put_vec(g, "coefficients_real", H5T_NATIVE_DOUBLE, nc, re);
      rc |= put_vec(g, "coefficients_imag", H5T_NATIVE_DOUBLE, nc, im);
      free(re); free(im);
      if (covr) {
        const size_t diag = covar ? (size_t)nmax * (nmax + 1) / 2 : (size_t)nmax;
        const size_t nv = (size_t)ltot * diag * (summed ? 1 : (size_t)sampT);
        re = (double *)calloc(nv ? nv : 1, sizeof(double));
        im = (double *)calloc(nv ? nv : 1, sizeof(double));
        for (int T = 0; T < sampT; T++) {
          size_t c = summed ? 0 : (size_t)T * ltot * diag;
          for (int l = 0; l < ltot; l++)
            for (int n1 = 0; n1 < nmax; n1++) {
              const double *row = covr + ((((size_t)T * ltot + l) * nmax + n1) * nmax) * 2;
              if (covar) {
                for (int n2 = n1; n2 < nmax; n2++, c++) { re[c] += row[2 * n2]; im[c] += row[2 * n2 + 1]; }
              } else {
                re[c] = row[2 * n1]; im[c] = row[2 * n1 + 1]; c++;
              }
            }
        }
        rc |= put_vec(g, summed ? "covariance_real_total" : "covariance_real", H5T_NATIVE_DOUBLE, nv, re);
        rc |= put_vec(g, summed ? "covariance_imag_total" : "covariance_imag", H5T_NATIVE_DOUBLE, nv, im);
        free(re); free(im);
      }
      H5Gclose(g);
      count++;
      if (H5Dwrite(cd, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &count) < 0) rc = -1;
    }
  }
  if (cd >= 0) H5Dclose(cd);
  if (snaps >= 0) H5Gclose(snaps);
  H5Fclose(f);
  return rc ? -1 : 0;
}

/* header of a covariance file: BasisID, version, the number of snapshots and the sizes of snapshot 0 */
int exp_h5_covar_info(const char *path, char *basisID, int cap, char *version, int vcap, int *floatsize,
                      int *count, int *sampT, int *ltot, int *nmax, int *summed, int *has_covar, int *full)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  int rc = 0;
  rc |= get_str(f, "CovarianceFileVersion", version, (size_t)vcap);
  rc |= get_str(f, "BasisID", basisID, (size_t)cap);
  rc |= get_int(f, "FloatSize", floatsize);
  unsigned cnt = 0;
  hid_t cd = H5Dopen2(f, "count", H5P_DEFAULT);
  if (cd < 0 || H5Dread(cd, H5T_NATIVE_UINT, H5S_ALL, H5S_ALL, H5P_DEFAULT, &cnt) < 0) rc = -1;
  if (cd >= 0) H5Dclose(cd);
  *count = (int)cnt;
  *sampT = *ltot = *nmax = *summed = *has_covar = *full = 0;
  if (rc == 0 && cnt > 0) {
    hid_t g = H5Gopen2(f, "snapshots/00000000", H5P_DEFAULT);
    if (g < 0) rc = -1;
    else {
      rc |= get_int(g, "sampleSize", sampT); rc |= get_int(g, "angularSize", ltot); rc |= get_int(g, "rankSize", nmax);
      *summed = H5Lexists(g, "covariance_real_total", H5P_DEFAULT) > 0;
      *has_covar = *summed || H5Lexists(g, "covariance_real", H5P_DEFAULT) > 0;
      if (*has_covar) {
        hid_t d = H5Dopen2(g, *summed ? "covariance_real_total" : "covariance_real", H5P_DEFAULT);
        hid_t s = H5Dget_space(d);
        const hssize_t npts = H5Sget_simple_extent_npoints(s);
        const hssize_t tri = (hssize_t)(*ltot) * (*nmax) * (*nmax + 1) / 2 * (*summed ? 1 : *sampT);
        *full = npts == tri;
        H5Sclose(s); H5Dclose(d);
      }
      H5Gclose(g);
    }
  }
  H5Fclose(f);
  return rc ? -1 : 0;
}

static int get_vec(hid_t g, const char *name, hid_t type, void *out)
{
  hid_t d = H5Dopen2(g, name, H5P_DEFAULT);
  if (d < 0) return -1;
  int rc = H5Dread(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0 ? -1 : 0;
  H5Dclose(d);
  return rc;
}

/* snapshot `index`: time, counts[T], masses[T], mean[T][ltot][nmax][2] and the covariance vectors as stored
 * (cre / cim: ltot * diag * (summed ? 1 : T) values; may be NULL) */
int exp_h5_covar_read(const char *path, int index, int sampT, int ltot, int nmax, int summed, double *time,
                      int *counts, double *masses, double *mean, double *cre, double *cim)
{
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return -1;
  char nm[32];
  snprintf(nm, sizeof nm, "snapshots/%08d", index);
  hid_t g = H5Gopen2(f, nm, H5P_DEFAULT);
  int rc = g < 0 ? -1 : 0;
  if (rc == 0) {
    rc |= get_dbl(g, "Time", time);
    rc |= get_vec(g, "sampleCounts", H5T_NATIVE_INT, counts);
    rc |= get_vec(g, "sampleMasses", H5T_NATIVE_DOUBLE, masses);
    const size_t nc = (size_t)sampT * ltot * nmax;
    double *re = (double *)malloc(sizeof(double) * (nc ? nc : 1)), *im = (double *)malloc(sizeof(double) * (nc ? nc : 1));
    rc |= get_vec(g, "coefficients_real", H5T_NATIVE_DOUBLE, re);
    rc |= get_vec(g, "coefficients_imag", H5T_NATIVE_DOUBLE, im);
    for (size_t c = 0; c < nc && rc == 0; c++) { mean[2 * c] = re[c]; mean[2 * c + 1] = im[c]; }
    free(re); free(im);
    if (cre && cim) {
      rc |= get_vec(g, summed ? "covariance_real_total" : "covariance_real", H5T_NATIVE_DOUBLE, cre);
      rc |= get_vec(g, summed ? "covariance_imag_total" : "covariance_imag", H5T_NATIVE_DOUBLE, cim);
    }
    H5Gclose(g);
  }
  H5Fclose(f);
  return rc ? -1 : 0;
}
