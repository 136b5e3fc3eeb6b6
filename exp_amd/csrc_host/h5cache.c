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
#include <hdf5.h>
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
