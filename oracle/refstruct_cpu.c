/*
 * refstruct_cpu.c -- the CPU baseline in the reference's DATA STRUCTURE: the same
 * arithmetic as bfe_oracle.c (it calls it), reached the way EXP's CPU path reaches its particles:
 *
 *   * particles are individually allocated objects (~200 B: mass, pos, vel, acc, pot, potext, level, index,
 *     attribute space -- include/Particle.H:17-60) held in a hash map keyed by the particle index
 *     (PartMap = unordered_map<unsigned long, shared_ptr<Particle>>, include/Particle.H:168-170);
 *   * a level list of keys per multistep level (Component::levlist, src/Component.H:460);
 *   * every access goes through Component::Mass(i) / Pos(i, k) / AddAcc(i, k, v) / AddPot(i, v), each a map
 *     lookup (src/Component.H:738-757, :914-921);
 *   * one step is FIVE separate passes over the level list -- incr_velocity, incr_position,
 *     determine_coefficients, get_acceleration_and_potential, incr_velocity (src/step.cc:271-323) -- each
 *     forked over nthrds pthreads on contiguous slices of the list (exp_thread_fork, src/PotAccel.cc:97-130),
 *     per-thread coefficient arrays summed afterwards (src/SphericalBasis.cc:855-858).
 *
 * TEST INFRASTRUCTURE / BASELINE ONLY (bfe_oracle.h).  A restatement of the structure, not a build of EXP.
 */
#include "bfe_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double mass, pos[3], vel[3], acc[3], pot, potext;
  float dtreq, scale;
  unsigned level;
  unsigned long indx;
  double dattrib[4];         /* attribute space a halo particle typically carries */
  int iattrib[2];
} rs_particle;

typedef struct rs_node { unsigned long key; rs_particle *p; struct rs_node *next; } rs_node;

typedef struct {
  rs_node **bucket;
  size_t nbucket;
  long n;
  unsigned long *levlist;    /* keys of level 0 (the single level of a multistep = 0 run) */
  double *coef;              /* [(L+1)^2][nmax], the reduced set */
  int ncoef;
} orc_refstruct;

static inline rs_particle *rs_find(const orc_refstruct *S, unsigned long key)
{
  for (rs_node *q = S->bucket[key % S->nbucket]; q; q = q->next)
    if (q->key == key) return q->p;
  return NULL;
}

orc_refstruct *orc_refstruct_create(long n, const double *mass, const double *pos, const double *vel)
{
  orc_refstruct *S = (orc_refstruct *)calloc(1, sizeof(*S));
  S->n = n;
  S->nbucket = (size_t)(n * 1.3) + 17;          /* load factor ~0.77 (libstdc++ keeps it <= 1) */
  S->bucket = (rs_node **)calloc(S->nbucket, sizeof(rs_node *));
  S->levlist = (unsigned long *)malloc(sizeof(unsigned long) * (size_t)(n ? n : 1));
  for (long i = 0; i < n; i++) {
    rs_particle *p = (rs_particle *)calloc(1, sizeof(rs_particle));
    rs_node *q = (rs_node *)malloc(sizeof(rs_node));
    const unsigned long key = (unsigned long)i + 1;           /* EXP's indices start at 1 */
    p->mass = mass[i];
    for (int k = 0; k < 3; k++) { p->pos[k] = pos[3 * i + k]; p->vel[k] = vel[3 * i + k]; }
    p->indx = key;
    q->key = key; q->p = p;
    q->next = S->bucket[key % S->nbucket];
    S->bucket[key % S->nbucket] = q;
    S->levlist[i] = key;
  }
  return S;
}

void orc_refstruct_free(orc_refstruct *S)
{
  if (!S) return;
  for (size_t b = 0; b < S->nbucket; b++)
    for (rs_node *q = S->bucket[b]; q;) { rs_node *nx = q->next; free(q->p); free(q); q = nx; }
  free(S->bucket); free(S->levlist); free(S->coef); free(S);
}

/* state back out in index order: pos, vel, acc [n][3], pot [n] */
void orc_refstruct_get(const orc_refstruct *S, double *pos, double *vel, double *acc, double *pot)
{
  for (long i = 0; i < S->n; i++) {
    const rs_particle *p = rs_find(S, (unsigned long)i + 1);
    for (int k = 0; k < 3; k++) { pos[3 * i + k] = p->pos[k]; vel[3 * i + k] = p->vel[k]; acc[3 * i + k] = p->acc[k]; }
    pot[i] = p->pot;
  }
}

const double *orc_refstruct_coef(const orc_refstruct *S) { return S->coef; }

typedef struct {
  orc_refstruct *S;
  const orc_slgrid *g;
  const orc_sph_params *P;
  int pass, id, nthrds;
  double dt;
  double *coef;      /* this thread's coefficient array (pass 2) */
  long used;
} rs_job;

#define RS_CHUNK 256

static void *rs_thread(void *arg)
{
  rs_job *J = (rs_job *)arg;
  orc_refstruct *S = J->S;
  const long n = S->n;
  const long nbeg = n * J->id / J->nthrds, nend = n * (J->id + 1) / J->nthrds;       /* src/SphericalBasis.cc:437-440 */
  const double center[3] = {0.0, 0.0, 0.0};
  if (J->pass == 0) {                                  /* incr_velocity_thread (src/incvel.cc:15-88) */
    for (long q = nbeg; q < nend; q++) {
      const unsigned long key = S->levlist[q];
      for (int k = 0; k < 3; k++) {
        rs_particle *p = rs_find(S, key);              /* c->Part(i): a lookup per access */
        p->vel[k] += p->acc[k] * J->dt;
      }
    }
  } else if (J->pass == 1) {                           /* incr_position_thread (src/incpos.cc:15-69) */
    for (long q = nbeg; q < nend; q++) {
      const unsigned long key = S->levlist[q];
      for (int k = 0; k < 3; k++) {
        rs_particle *p = rs_find(S, key);
        p->pos[k] += p->vel[k] * J->dt;
      }
    }
  } else if (J->pass == 2) {                           /* determine_coefficients_thread (src/SphericalBasis.cc:429-599) */
    double x[RS_CHUNK], y[RS_CHUNK], z[RS_CHUNK], m[RS_CHUNK];
    double *part = (double *)malloc(sizeof(double) * S->ncoef);
    memset(J->coef, 0, sizeof(double) * S->ncoef);
    J->used = 0;
    for (long q0 = nbeg; q0 < nend; q0 += RS_CHUNK) {
      const long cnt = (nend - q0 < RS_CHUNK) ? nend - q0 : RS_CHUNK;
      for (long j = 0; j < cnt; j++) {
        const unsigned long key = S->levlist[q0 + j];
        m[j] = rs_find(S, key)->mass;                  /* cC->Mass(indx) */
        x[j] = rs_find(S, key)->pos[0];                /* cC->Pos(indx, 0, Component::Local | Centered) */
        y[j] = rs_find(S, key)->pos[1];
        z[j] = rs_find(S, key)->pos[2];
      }
      J->used += orc_sph_accumulate(J->g, J->P, cnt, x, y, z, m, center, part, 0);
      for (int k = 0; k < S->ncoef; k++) J->coef[k] += part[k];
    }
    free(part);
  } else {                                             /* determine_acceleration_and_potential_thread (:1476-1660) */
    double x[RS_CHUNK], y[RS_CHUNK], z[RS_CHUNK], ax[RS_CHUNK], ay[RS_CHUNK], az[RS_CHUNK], pot[RS_CHUNK];
    for (long q0 = nbeg; q0 < nend; q0 += RS_CHUNK) {
      const long cnt = (nend - q0 < RS_CHUNK) ? nend - q0 : RS_CHUNK;
      for (long j = 0; j < cnt; j++) {
        const unsigned long key = S->levlist[q0 + j];
        x[j] = rs_find(S, key)->pos[0];
        y[j] = rs_find(S, key)->pos[1];
        z[j] = rs_find(S, key)->pos[2];
        ax[j] = ay[j] = az[j] = pot[j] = 0.0;
      }
      orc_sph_accel(J->g, J->P, cnt, x, y, z, center, S->coef, ax, ay, az, pot);
      for (long j = 0; j < cnt; j++) {
        const unsigned long key = S->levlist[q0 + j];
        rs_find(S, key)->acc[0] += ax[j];              /* cC->AddAcc(indx, k, v) x3 (+2 for the azimuthal term) */
        rs_find(S, key)->acc[1] += ay[j];
        rs_find(S, key)->acc[2] += az[j];
        rs_find(S, key)->acc[0] += 0.0;
        rs_find(S, key)->acc[1] += 0.0;
        rs_find(S, key)->pot += pot[j];                /* cC->AddPot(indx, potl) */
      }
    }
  }
  return NULL;
}

static void rs_fork(orc_refstruct *S, const orc_slgrid *g, const orc_sph_params *P, int pass, double dt,
                    int nthrds, double **tcoef, long *used)
{
  pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * nthrds);
  rs_job *J = (rs_job *)calloc(nthrds, sizeof(rs_job));
  for (int i = 0; i < nthrds; i++) {
    J[i].S = S; J[i].g = g; J[i].P = P; J[i].pass = pass; J[i].id = i; J[i].nthrds = nthrds; J[i].dt = dt;
    J[i].coef = tcoef ? tcoef[i] : NULL;
    pthread_create(&t[i], NULL, rs_thread, &J[i]);
  }
  long u = 0;
  for (int i = 0; i < nthrds; i++) { pthread_join(t[i], NULL); u += J[i].used; }
  if (used) *used = u;
  free(t); free(J);
}

/* the coefficient and force passes alone (begin_run's compute_expansion + compute_potential) */
long orc_refstruct_field(orc_refstruct *S, const orc_slgrid *g, const orc_sph_params *P, int nthrds)
{
  S->ncoef = (g->lmax + 1) * (g->lmax + 1) * g->nmax;
  if (!S->coef) S->coef = (double *)calloc(S->ncoef, sizeof(double));
  double **tcoef = (double **)malloc(sizeof(double *) * nthrds);
  for (int i = 0; i < nthrds; i++) tcoef[i] = (double *)malloc(sizeof(double) * S->ncoef);
  long used = 0;
  rs_fork(S, g, P, 2, 0.0, nthrds, tcoef, &used);
  memset(S->coef, 0, sizeof(double) * S->ncoef);
  for (int i = 0; i < nthrds; i++) {                                  /* thread sum (:855-858) */
    for (int k = 0; k < S->ncoef; k++) S->coef[k] += tcoef[i][k];
    free(tcoef[i]);
  }
  free(tcoef);
  for (long i = 0; i < S->n; i++) {                                   /* ComponentContainer zeroes acc / pot (:641-665) */
    rs_particle *p = rs_find(S, S->levlist[i]);
    p->acc[0] = p->acc[1] = p->acc[2] = 0.0;
    p->pot = 0.0;
  }
  rs_fork(S, g, P, 3, 0.0, nthrds, NULL, NULL);
  return used;
}

/* do_step at multistep = 0 (src/step.cc:271-323): five passes */
long orc_refstruct_step(orc_refstruct *S, const orc_slgrid *g, const orc_sph_params *P, double dt, int nthrds)
{
  rs_fork(S, g, P, 0, 0.5 * dt, nthrds, NULL, NULL);
  rs_fork(S, g, P, 1, dt, nthrds, NULL, NULL);
  long used = orc_refstruct_field(S, g, P, nthrds);
  rs_fork(S, g, P, 0, 0.5 * dt, nthrds, NULL, NULL);
  return used;
}
