/*
 * nbody_oracle.c -- CPU restatement of EXP's block-multistep step loop over several components
 * with mutual interactions.  TEST INFRASTRUCTURE ONLY; see nbody_oracle.h / bfe_oracle.h for the
 * scope statement (parity unpinned).  One process, one thread: the reference's thread sums and
 * MPI_Allreduce calls reduce to plain assignments.
 */
#include "nbody_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- CylEXP::multistep_update (src/CylEXP.cc:159-188) ------------------------------------------- */
int orc_cyl_multistep_update(const orc_cylgrid *g, double xx, double yy, double zz, double mass,
                             double *val, double *vc, double *vs)
{
  /* Cylinder::multistep_update (src/Cylinder.cc:1752-1773): r, phi from the centred position */
  const int N = g->norder, half = (g->mmax + 1) * N;
  double r2 = (xx * xx + yy * yy);
  double r = sqrt(r2);
  double phi = atan2(yy, xx);
  double z = zz;

  double rr = sqrt(r * r + z * z);
  if (rr / g->ascale > g->rtable) return 0;

  double norm = -4.0 * M_PI;
  memset(vc, 0, sizeof(double) * half);
  memset(vs, 0, sizeof(double) * half);
  orc_cyl_get_pot(g, r, z, vc, vs);
  memset(val, 0, sizeof(double) * 2 * half);
  for (int mm = 0; mm <= g->mmax; mm++) {
    double mcos = cos(phi * mm);
    double msin = sin(phi * mm);
    for (int nn = 0; nn < N; nn++) {
      double hold = norm * mass * mcos * vc[mm * N + nn];
      val[mm * N + nn] = hold;
      if (mm > 0) {
        hold = norm * mass * msin * vs[mm * N + nn];
        val[half + mm * N + nn] = hold;
      }
    }
  }
  return 1;
}

/* ---- helpers ------------------------------------------------------------------------------------- */

/* the per-call options (bfe_oracle.h) of a call that walks the particles of `walked` with the force method of `basis`:
 * adb = basis' component->Adiabatic() at tnow, freeze = walked->freeze(), mlim = the basis' */
static void call_opts(const orc_nbody *S, const orc_nbody_comp *basis, const orc_nbody_comp *walked)
{
  orc_call_opts o = {1.0, 0, 1.0e20, {0, 0, 0}, {0, 0, 0}, -1, 1.0, 1};
  if (basis == walked && basis->ssfrac > 0.0 && basis->ssfrac < 1.0) {      /* (only the accumulation reads it) */
    o.ssfrac = basis->ssfrac;
    o.nthrds = basis->ss_nthrds < 1 ? 1 : basis->ss_nthrds;
  }
  if (basis->adiabatic) o.adb = orc_adiabatic(S->tnow, basis->ton, basis->toff, basis->twid);
  if (walked->has_rtrunc) {
    o.frz = 1;
    o.rtrunc = walked->rtrunc;
    for (int k = 0; k < 3; k++) { o.com0[k] = walked->com0[k]; o.fcenter[k] = walked->center[k]; }
  }
  if (basis->has_mlim) o.mlim = basis->mlim;
  orc_set_call_opts(&o);
}

/* `if (!self_consistent && !firstime_coef && !initializing) return;` (src/SphericalBasis.cc:694, src/Cylinder.cc:959) */
static int coefs_fixed(const orc_nbody *S, const orc_nbody_comp *c)
{
  return c->not_self_consistent && c->coef_calls > 0 && !S->initializing;
}

typedef struct {
  double *tx, *ty, *tz, *tm;     /* gathered level lists */
  double *val, *differ, *vc, *vs;
  long cap, ncoef_max;
} nb_work;

static void work_alloc(nb_work *w, const orc_nbody *S)
{
  long nmax = 0, cmax = 0, hmax = 1;
  for (int k = 0; k < S->ncomp; k++) {
    if (S->comp[k].n > nmax) nmax = S->comp[k].n;
    if (S->comp[k].ncoef > cmax) cmax = S->comp[k].ncoef;
    if (S->comp[k].kind == 1) {
      long h = (long)(S->comp[k].cg->mmax + 1) * S->comp[k].cg->norder;
      if (h > hmax) hmax = h;
    }
  }
  if (nmax < 1) nmax = 1;
  w->cap = nmax;
  w->ncoef_max = cmax;
  w->tx = (double *)malloc(sizeof(double) * nmax);
  w->ty = (double *)malloc(sizeof(double) * nmax);
  w->tz = (double *)malloc(sizeof(double) * nmax);
  w->tm = (double *)malloc(sizeof(double) * nmax);
  w->val = (double *)malloc(sizeof(double) * (cmax + 1));
  w->differ = (double *)malloc(sizeof(double) * (cmax + 1) * (S->multistep + 1));
  w->vc = (double *)malloc(sizeof(double) * hmax);
  w->vs = (double *)malloc(sizeof(double) * hmax);
}

static void work_free(nb_work *w)
{
  free(w->tx); free(w->ty); free(w->tz); free(w->tm); free(w->val); free(w->differ);
  free(w->vc); free(w->vs);
}

/* PotAccel::multistep_reset: SphericalBasis (src/SphericalBasis.cc:1004-1010: used = 0, resetT),
 * Cylinder (src/Cylinder.cc:1786-1795: used = 0, cylmass = 0, resetT = tnow) */
static void multistep_reset(orc_nbody *S)
{
  for (int k = 0; k < S->ncomp; k++) {
    orc_nbody_comp *c = &S->comp[k];
    c->used = 0;
    if (c->kind == 1) c->cylmass = 0.0;
    c->resetT = S->tnow;
  }
}

/* ComponentContainer::compute_expansion(M) (src/ComponentContainer.cc:1173-1226): every component's
 * determine_coefficients for level M.  Spherical: expcoefL[M] <-> expcoefN[M], N zeroed and
 * accumulated from levlist[M] (src/SphericalBasis.cc:785-792, :429-599).  Cylinder:
 * setup_accumulation(M) swaps cosL/cosN, sinL/sinN and zeroes N (exputil/EmpCylSL.cc:2010-2030), the
 * thread body accumulates levlist[M] (src/Cylinder.cc:823-890), used / cylmass are added only while
 * tnow == resetT (:1091-1099); the compute_multistep_coefficients() call at :1112 only fills
 * accum_cos/sin, which the next force evaluation overwrites (:1469-1471).                        */
static void compute_expansion(orc_nbody *S, int M, nb_work *w)
{
  for (int k = 0; k < S->ncomp; k++) {
    orc_nbody_comp *c = &S->comp[k];
    if (coefs_fixed(S, c)) continue;              /* (the return precedes the N/L swap) */
    double *N = c->coefN + (size_t)M * c->ncoef, *L = c->coefL + (size_t)M * c->ncoef;
    memcpy(L, N, sizeof(double) * c->ncoef);      /* after the swap, L holds the old N */
    call_opts(S, c, c);
    long cnt = 0;
    for (long i = 0; i < c->n; i++)
      if (c->level[i] == M) {
        w->tx[cnt] = c->x[i]; w->ty[cnt] = c->y[i]; w->tz[cnt] = c->z[i]; w->tm[cnt] = c->mass[i];
        cnt++;
      }
    if (c->kind == 0) {
      long use = orc_sph_accumulate(c->sg, c->sp, cnt, w->tx, w->ty, w->tz, w->tm, c->center, N, 0);
      if (S->multistep == 0) c->used = 0;                               /* :796 */
      if (S->multistep == 0 || S->tnow == c->resetT) c->used += use;   /* :860-862 */
    } else {
      const long half = (long)(c->cg->mmax + 1) * c->cg->norder;
      double cm = 0.0;
      long use = orc_cyl_accumulate(c->cg, cnt, w->tx, w->ty, w->tz, w->tm, c->center, N, N + half, &cm);
      /* src/Cylinder.cc:1091-1099 has no `multistep==0 or` (SphericalBasis.cc:860 has): with
       * multistep = 0, do_step's multistep_reset zeroes cylmass and moves tnow past resetT before
       * the expansion, so the literal reference blends the off-grid monopole with mass 0 (and reads
       * an uninitialised resetT in begin_run).  Deliberate deviation, stated in DESIGN.md: with
       * multistep = 0 cylmass / used are those of the last accumulation.                       */
      if (S->multistep == 0) { c->used = use; c->cylmass = cm; }
      else if (S->tnow == c->resetT) { c->used += use; c->cylmass += cm; }
    }
    orc_set_call_opts(NULL);
    c->coef_calls++;                              /* firstime_coef = false (:1001, :1198) */
  }
}

/* compute_multistep_coefficients (src/SphericalBasis.cc:1231-1333; src/CylEXP.cc:192-282) */
static void combine(const orc_nbody *S, const orc_mstep_tables *T, orc_nbody_comp *c, int mdrft)
{
  /* `if (multistep && (self_consistent || initializing)) compute_multistep_coefficients();`
   * (src/SphericalBasis.cc:1682, src/Cylinder.cc:1469); with multistep = 0 the set that is not re-made is the one copied */
  if (S->multistep && c->not_self_consistent && !S->initializing) return;
  if (S->multistep == 0) {
    memcpy(c->coef, c->coefN, sizeof(double) * c->ncoef);
    return;
  }
  orc_mstep_combine(T, mdrft, c->ncoef, c->coefL, c->coefN, c->coef);
}

/* one force method applied to the particles of `t` with level >= mlevel */
static void apply_force(const orc_nbody *S, orc_nbody_comp *src, orc_nbody_comp *t, int mlevel)
{
  if (src->kind == 0 && src->noise) {
    /* `if (NOISE) update_noise();` opens get_acceleration_and_potential (src/SphericalBasis.cc:395); the self call of a
     * multistep force then rebuilds the set from the per-level ones (:1680-1685: combine() has run above), an external call
     * and a single-level self call evaluate the draws */
    orc_noise_update(src->noise, src->noise_buf);
    const int rebuilt = src == t && S->multistep && (!src->not_self_consistent || S->initializing);
    if (!rebuilt) memcpy(src->coef, src->noise_buf, sizeof(double) * src->ncoef);
  }
  if (src->kind == 0 && src->fix_l0) {            /* src/SphericalBasis.cc:1689-1694 (self and external calls alike) */
    const int nmax = src->sg->nmax;
    if (!src->have_c0) { memcpy(src->C0, src->coef, sizeof(double) * nmax); src->have_c0 = 1; }
    else memcpy(src->coef, src->C0, sizeof(double) * nmax);
  }
  call_opts(S, src, t);
  for (long i = 0; i < t->n; i++) {
    if (t->level[i] < mlevel) continue;
    if (src->kind == 0)
      orc_sph_accel(src->sg, src->sp, 1, t->x + i, t->y + i, t->z + i, src->center, src->coef,
                    t->ax + i, t->ay + i, t->az + i, t->pot + i);
    else {
      const long half = (long)(src->cg->mmax + 1) * src->cg->norder;
      orc_cyl_accel(src->cg, 1, t->x + i, t->y + i, t->z + i, src->center, src->coef,
                    src->coef + half, src->cylmass, t->ax + i, t->ay + i, t->az + i, t->pot + i);
    }
  }
  orc_set_call_opts(NULL);
}

/* ComponentContainer::compute_potential(mlevel) (src/ComponentContainer.cc:580-917): per component
 * zero pot/potext/acc of levels >= mlevel (:641-665) and apply its own force (:698-716; the self
 * call recombines the level sets, src/SphericalBasis.cc:1680-1694, src/Cylinder.cc:1466-1472);
 * then the interaction list, SetExternal (:785-853): the source's coefficient set as the self call
 * left it, the target's positions in the source's centred frame.  pot and potext are kept in one
 * array (adjust_multistep_level only uses their sum, src/multistep.cc:112).                       */
static void compute_potential(orc_nbody *S, const orc_mstep_tables *T, int mlevel, int mdrft)
{
  for (int k = 0; k < S->ncomp; k++) {
    orc_nbody_comp *c = &S->comp[k];
    for (long i = 0; i < c->n; i++)
      if (c->level[i] >= mlevel) c->ax[i] = c->ay[i] = c->az[i] = c->pot[i] = 0.0;
    combine(S, T, c, mdrft);
    apply_force(S, c, c, mlevel);
  }
  for (int q = 0; q < S->ninter; q++)
    apply_force(S, &S->comp[S->inter[2 * q]], &S->comp[S->inter[2 * q + 1]], mlevel);
}

/* adjust_multistep_level (src/multistep.cc:344-627): multistep_update_begin for every component,
 * the level sweep of every component (levels first..multistep of the OLD level lists; a particle is
 * visited once because the lists are rebuilt only by reset_level_lists), multistep_update_finish. */
static void adjust_levels(orc_nbody *S, const orc_mstep_tables *T, int mdrft, int all_levels,
                          nb_work *w, long *nswitch)
{
  const int ms = S->multistep;
  if (!ms) return;
  const int mf = T->mfirst[mdrft];
  int first = mf;
  if (all_levels) first = 0;                      /* this_step == 0 and mstep == 0 (:451-453) */
  for (int k = 0; k < S->ncomp; k++) {
    orc_nbody_comp *c = &S->comp[k];
    const long nc = c->ncoef;
    for (int M = mf; M <= ms; M++) memset(w->differ + (size_t)M * nc, 0, sizeof(double) * nc);
    long switched = 0;
    /* `if (not firstCall and c->FreezeLev()) apply = false;` (src/multistep.cc:158): no level is proposed, nothing moves */
    /* (firstCall = this_step == 0 and mdrft == 0: begin_run's call; the first sub-step of the run also does all levels, :453) */
    const int first_call = all_levels && mdrft == 0;
    const int mstep = mdrft - 1;                  /* do_step: mdrft = mstep + 1 at its call (src/step.cc:188, :221) */
    const int frozen_levels = c->freeze_lev && !first_call;
    for (int lev = first; lev <= ms && !frozen_levels; lev++) {
      for (long i = 0; i < c->n; i++) {
        if (c->level[i] != lev) continue;
        double v[3] = {c->vx[i], c->vy[i], c->vz[i]}, a[3] = {c->ax[i], c->ay[i], c->az[i]};
        /* src/multistep.cc:132-158: Particle::dtreq (a float) and whether this sweep assigns levels */
        const double dt = orc_level_dt(S->dynfrac, 0.0, v, a, c->pot[i]);
        float dtreq;
        if (c->noswitch) {
          if ((!c->no_dtreset && mstep == 0) || first_call) c->dtreq[i] = HUGE_VALF;   /* (float)DBL_MAX */
          if (dt < c->dtreq[i]) c->dtreq[i] = (float)dt;
          dtreq = c->dtreq[i];
          if (!(mdrft == (1 << ms) || first_call)) continue;                           /* apply (:147) */
        } else dtreq = (float)dt;
        int nlev = orc_level_rule(S->dtime, ms, mf, lev, S->shiftlevl, dtreq);
        if (nlev == lev) continue;
        double xx = c->x[i] - c->center[0], yy = c->y[i] - c->center[1], zz = c->z[i] - c->center[2];
        int inside;
        /* `if (c->freeze(i)) return; double mass = c->Mass(i) * component->Adiabatic();` (src/SphericalBasis.cc:
         * 1159-1161, src/Cylinder.cc:1756-1758); the cylinder alone tests self_consistent first (:1755) */
        call_opts(S, c, c);
        const double mas = c->mass[i] * orc_opt_adb();
        if (orc_opt_frozen(c->x[i], c->y[i], c->z[i])) inside = 0;
        else if (c->kind == 0) inside = orc_sph_multistep_update(c->sg, c->sp, xx, yy, zz, mas, w->val);
        else if (c->not_self_consistent) inside = 0;
        else              inside = orc_cyl_multistep_update(c->cg, xx, yy, zz, mas, w->val, w->vc, w->vs);
        orc_set_call_opts(NULL);
        if (inside)
          for (long q = 0; q < nc; q++) {
            /* differ[from] -= val; differ[to] += val; only M >= mfirst[mdrft] is cleared by
             * _begin and added by _finish */
            if (lev >= mf) w->differ[(size_t)lev * nc + q] -= w->val[q];
            if (nlev >= mf) w->differ[(size_t)nlev * nc + q] += w->val[q];
          }
        c->level[i] = -(nlev + 1);                /* p->level = nlev; seen once (old lists) */
        switched++;
      }
    }
    for (long i = 0; i < c->n; i++)
      if (c->level[i] < 0) c->level[i] = -c->level[i] - 1;
    for (int M = mf; M <= ms; M++)
      for (long q = 0; q < nc; q++) c->coefN[(size_t)M * nc + q] += w->differ[(size_t)M * nc + q];
    if (nswitch) nswitch[k] += switched;
  }
}

/* first half of begin_run only (expansion at every level, potential, first level assignment with
 * its differencing): lets a test compare the differenced expcoefN / cosN with a fresh accumulation
 * of the new level lists at the same positions                                                   */
void orc_nbody_init_pass0(orc_nbody *S)
{
  const int ms = S->multistep;
  orc_mstep_tables *T = orc_mstep_create(ms);
  nb_work w;
  work_alloc(&w, S);
  S->this_step = 0;
  S->initializing = 1;
  multistep_reset(S);
  for (int M = 0; M <= ms; M++) compute_expansion(S, M, &w);
  compute_potential(S, T, 0, 0);
  adjust_levels(S, T, 0, 1, &w, NULL);
  S->initializing = 0;
  work_free(&w);
  orc_mstep_free(T);
}

void orc_nbody_init(orc_nbody *S)
{
  const int ms = S->multistep;
  orc_mstep_tables *T = orc_mstep_create(ms);
  nb_work w;
  work_alloc(&w, S);
  S->this_step = 0;
  S->initializing = 1;                                        /* src/begin.cc:80 */
  if (ms) {
    multistep_reset(S);
    for (int M = 0; M <= ms; M++) compute_expansion(S, M, &w);
    compute_potential(S, T, 0, 0);
    adjust_levels(S, T, 0, 1, &w, NULL);
  }
  if (ms) multistep_reset(S);
  for (int M = 0; M <= ms; M++) compute_expansion(S, M, &w);
  compute_potential(S, T, 0, 0);
  S->initializing = 0;                                        /* :129 */
  work_free(&w);
  orc_mstep_free(T);
}

void orc_nbody_step(orc_nbody *S, long *nswitch)
{
  const int ms = S->multistep;
  orc_mstep_tables *T = orc_mstep_create(ms);
  nb_work w;
  work_alloc(&w, S);
  if (nswitch) for (int k = 0; k < S->ncomp; k++) nswitch[k] = 0;

  multistep_reset(S);                                        /* src/step.cc:84 */

  if (ms) {
    const int Mstep = T->Mstep;
    const double dt = S->dtime / Mstep;
    for (int mstep = 0; mstep < Mstep; mstep++) {
      int mdrft = mstep;
      for (int M = T->mfirst[mstep]; M <= ms; M++) {
        double DT = dt * T->mintvl[M];
        /* incr_velocity(0.5*DT, M); incr_position(DT, M) over all components */
        for (int k = 0; k < S->ncomp; k++) {
          orc_nbody_comp *c = &S->comp[k];
          for (long i = 0; i < c->n && !S->no_eqmotion; i++)
            if (c->level[i] == M) {
              c->vx[i] += c->ax[i] * (0.5 * DT); c->vy[i] += c->ay[i] * (0.5 * DT); c->vz[i] += c->az[i] * (0.5 * DT);
            }
        }
        for (int k = 0; k < S->ncomp; k++) {
          orc_nbody_comp *c = &S->comp[k];
          for (long i = 0; i < c->n && !S->no_eqmotion; i++)
            if (c->level[i] == M) {
              c->x[i] += c->vx[i] * DT; c->y[i] += c->vy[i] * DT; c->z[i] += c->vz[i] * DT;
            }
        }
        compute_expansion(S, M, &w);
      }
      S->tnow += dt;
      mdrft = mstep + 1;
      compute_potential(S, T, T->mfirst[mstep], mdrft);
      for (int M = T->mfirst[mdrft]; M <= ms; M++) {
        double DT = dt * T->mintvl[M];
        for (int k = 0; k < S->ncomp; k++) {
          orc_nbody_comp *c = &S->comp[k];
          for (long i = 0; i < c->n && !S->no_eqmotion; i++)
            if (c->level[i] == M) {
              c->vx[i] += c->ax[i] * (0.5 * DT); c->vy[i] += c->ay[i] * (0.5 * DT); c->vz[i] += c->az[i] * (0.5 * DT);
            }
        }
      }
      adjust_levels(S, T, mdrft, (S->this_step == 0 && mstep == 0), &w, nswitch);
    }
  } else {
    /* src/step.cc:271-323 */
    S->tnow += S->dtime;
    for (int k = 0; k < S->ncomp; k++) {
      orc_nbody_comp *c = &S->comp[k];
      if (!S->no_eqmotion) orc_kick(c->n, 0.5 * S->dtime, c->vx, c->vy, c->vz, c->ax, c->ay, c->az);
    }
    for (int k = 0; k < S->ncomp; k++) {
      orc_nbody_comp *c = &S->comp[k];
      if (!S->no_eqmotion) orc_drift(c->n, S->dtime, c->x, c->y, c->z, c->vx, c->vy, c->vz);
    }
    compute_expansion(S, 0, &w);
    compute_potential(S, T, 0, 1);
    for (int k = 0; k < S->ncomp; k++) {
      orc_nbody_comp *c = &S->comp[k];
      if (!S->no_eqmotion) orc_kick(c->n, 0.5 * S->dtime, c->vx, c->vy, c->vz, c->ax, c->ay, c->az);
    }
  }
  S->this_step++;
  work_free(&w);
  orc_mstep_free(T);
}
