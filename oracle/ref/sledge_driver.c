/*
 * sledge_driver.c -- calls the REFERENCE's own Sturm-Liouville solver (exputil/sledge.f, SLEDGE
 * 2.2, compiled from /root/reference where it lies, see oracle/ref/Makefile) exactly the way
 * SLGridSph::compute_table does (exputil/SLGridMP2.cc:1103-1200: cons, tol, type, endfin, job,
 * invec; coefficient callback coeff_ :3632-3664; sign convention :1318-1330).
 *
 * TEST INFRASTRUCTURE ONLY.  It exists to check exp_amd/slgrid.py (our own solver of the same
 * problem, needed because the product cannot ship Fortran) against the reference's table
 * construction.  The output goes to oracle/_ref/ (git-ignored); nothing under exp_amd/ uses it.
 *
 * The background model (Phi0, 4 pi rho0, dPhi0/dr) is supplied by the caller as C callbacks so
 * that the very same model objects feed both solvers.
 */
#include <stdlib.h>
#include <string.h>

typedef int    logical;      /* exputil/SLGridMP2.cc:39-41 */
typedef double doublereal;
typedef int    integer;

extern void sledge_(logical *job, doublereal *cons, logical *endfin, integer *invec,
                    doublereal *tol, logical *type, doublereal *ev, integer *numx,
                    doublereal *xef, doublereal *ef, doublereal *pdef, doublereal *t,
                    doublereal *rho, integer *iflag, doublereal *store);

typedef double (*model_fn)(double);
static model_fn g_pot, g_dens, g_dpot;
static double g_L2;

/* exputil/SLGridMP2.cc:3632-3664 (spherical branch) */
int coeff_(doublereal *x, doublereal *px, doublereal *qx, doublereal *rx)
{
  double f = g_pot(*x);
  double rho = g_dens(*x);
  *px = (*x) * (*x) * f * f;
  *qx = (g_L2 * f - rho * (*x) * (*x)) * f;
  *rx = -rho * (*x) * (*x) * f;
  return 0;
}

/* One harmonic order: r[num] is the output mesh (the table's radial grid between the inner and
 * outer boundary), ev[nmax], ef[nmax][num] (sign convention applied), iflag[nmax].            */
int ref_sledge_order(int l, int nmax, int num, const double *r, double rmap, int nevsign,
                     model_fn pot, model_fn dens, model_fn dpot, double *ev_out, double *ef_out,
                     int *iflag_out)
{
  doublereal cons[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  doublereal tol[6] = {1.0e-4 * rmap, 1.0e-6, 1.0e-4 * rmap, 1.0e-6, 1.0e-4 * rmap, 1.0e-6};
  logical type[8] = {0, 0, 1, 0, 0, 0, 1, 0};
  logical endfin[2] = {1, 1};
  integer NUM = num, N = nmax;

  g_pot = pot; g_dens = dens; g_dpot = dpot;
  g_L2 = (double)l * (l + 1);

  cons[6] = r[0];
  cons[7] = r[num - 1];

  integer *iflag = (integer *)calloc(nmax, sizeof(integer));
  integer *invec = (integer *)calloc(nmax + 3, sizeof(integer));
  doublereal *ev = (doublereal *)calloc(N, sizeof(doublereal));
  doublereal *store = (doublereal *)calloc(26 * (NUM + 16), sizeof(doublereal));
  doublereal *xef = (doublereal *)calloc(NUM + 16, sizeof(doublereal));
  doublereal *ef = (doublereal *)calloc((size_t)NUM * N, sizeof(doublereal));
  doublereal *pdef = (doublereal *)calloc((size_t)NUM * N, sizeof(doublereal));
  doublereal tdum = 0.0, rhodum = 0.0;
  double f;

  f = pot(cons[6]);                         /* inner BC */
  if (l == 0) {
    cons[0] = dpot(cons[6]) / f;
    cons[2] = 1.0 / (cons[6] * cons[6] * f * f);
  } else
    cons[0] = 1.0;

  f = pot(cons[7]);                         /* outer BC */
  cons[4] = (1.0 + l) / cons[7] + dpot(cons[7]) / f;
  cons[5] = 1.0 / (cons[7] * cons[7] * f * f);

  invec[0] = 0;
  invec[1] = 3;
  invec[2] = N;
  for (int i = 0; i < N; i++) invec[3 + i] = i;

  logical job[5] = {0, 1, 0, 1, 0};

  for (int i = 0; i < NUM; i++) xef[i] = r[i];

  sledge_(job, cons, endfin, invec, tol, type, ev, &NUM, xef, ef, pdef, &tdum, &rhodum, iflag,
          store);

  int nfid = (nevsign < NUM ? nevsign : NUM) - 1;
  for (int j = 0; j < N; j++) {
    double sgn = (ef[(size_t)j * NUM + nfid] < 0.0) ? -1.0 : 1.0;
    ev_out[j] = ev[j];
    iflag_out[j] = iflag[j];
    for (int i = 0; i < NUM; i++) ef_out[(size_t)j * num + i] = ef[(size_t)j * NUM + i] * sgn;
  }
  free(iflag); free(invec); free(ev); free(store); free(xef); free(ef); free(pdef);
  return 0;
}
