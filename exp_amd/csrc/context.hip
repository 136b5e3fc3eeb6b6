// Context, error reporting, profiling and the coefficient all-reduce of libexp_amd.
#include "common.h"
#include <chrono>

#include <dlfcn.h>

thread_local std::string g_exp_amd_global_err;

int expamd_fail(exp_amd_ctx *ctx, int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  g_exp_amd_global_err = buf;
  return code;
}

extern "C" int exp_amd_abi_version(void) { return 1; }

extern "C" const char *exp_amd_last_error(const exp_amd_ctx *ctx)
{
  return ctx ? ctx->err.c_str() : g_exp_amd_global_err.c_str();
}

extern "C" const char *exp_amd_last_global_error(void) { return g_exp_amd_global_err.c_str(); }

extern "C" int exp_amd_ctx_create(int device, void *stream, exp_amd_ctx **out)
{
  if (!out) return expamd_fail(nullptr, EXP_AMD_ERR_ARG, "ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return expamd_fail(nullptr, EXP_AMD_ERR_NODEVICE,
                       "ctx_create: no HIP device available (%s); the BFE hot path has no "
                       "CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return expamd_fail(nullptr, EXP_AMD_ERR_ARG, "ctx_create: device %d out of range [0,%d)",
                       device, ndev);
  exp_amd_ctx *ctx = new exp_amd_ctx;
  ctx->device = device;
  // (tuning constants of the block-multistep engine without a setter of their own)
  ctx->thin_acc_scale = EXPAMD_EXPT("EXP_AMD_THIN_ACC_SCALE", ctx->thin_acc_scale) > 0 ? EXPAMD_EXPT("EXP_AMD_THIN_ACC_SCALE", ctx->thin_acc_scale) : 1;
  ctx->mover_slices_min = EXPAMD_EXPT("EXP_AMD_MOVER_SLICES_MIN", ctx->mover_slices_min);
  ctx->stage_max = EXPAMD_EXPT("EXP_AMD_STAGE_MAX", ctx->stage_max);
  // EXP_AMD_APPEND_MIN (include/exp_amd.h, "Environment"): the default of exp_amd_ctx_set_append_min
  if (const char *e = getenv("EXP_AMD_APPEND_MIN")) ctx->append_min = atoll(e);
  if (const char *e = getenv("EXP_AMD_APPEND_LEAN")) ctx->append_lean = atoll(e) != 0;
  HIP_TRY(ctx, hipSetDevice(device));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      int rc = expamd_fail(nullptr, EXP_AMD_ERR_HIP, "hipStreamCreate failed: %s",
                           hipGetErrorString(e));
      delete ctx;
      return rc;
    }
    ctx->own_stream = true;
  }
  *out = ctx;
  return EXP_AMD_OK;
}

extern "C" void exp_amd_ctx_destroy(exp_amd_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &s : ctx->slots)
    for (auto &p : s.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
  if (ctx->rccl_comm && ctx->rccl_lib) {
    typedef int (*destroy_fn)(void *);
    destroy_fn d = (destroy_fn)dlsym(ctx->rccl_lib, "ncclCommDestroy");
    if (d && ctx->rccl_comm2) d(ctx->rccl_comm2);
    if (d) d(ctx->rccl_comm);
  }
  if (ctx->aux) {
    (void)hipStreamSynchronize(ctx->aux);
    (void)hipStreamDestroy(ctx->aux);
    for (int k = 0; k < 2; k++) { (void)hipEventDestroy(ctx->ev_sorted[k]); (void)hipEventDestroy(ctx->ev_forced[k]); }
  }
  for (auto &ss : ctx->scan_sums) if (ss.p) (void)hipFree(ss.p);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int exp_amd_ctx_set_prekick(exp_amd_ctx *ctx, int on)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->prekick = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_deterministic(exp_amd_ctx *ctx, int on)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->deterministic = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_dense_min(exp_amd_ctx *ctx, long long nmin)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->dense_min = nmin;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_thin_max(exp_amd_ctx *ctx, long long nmax)
{
  expamd_mutated();
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->thin_max = nmax < 0 ? 0 : nmax;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_mover_list_min(exp_amd_ctx *ctx, long long nmin)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->mover_list_min = nmin;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_append_min(exp_amd_ctx *ctx, long long nmin)
{
  expamd_mutated();
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->append_min = nmin;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_append_lean(exp_amd_ctx *ctx, int on)
{
  expamd_mutated();
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->append_lean = on != 0;          // (read by every append step: a store placed the other way stays valid either way)
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_set_split_min(exp_amd_ctx *ctx, long long nmin)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->split_min = nmin;
  return EXP_AMD_OK;
}

int expamd_ctx_aux(exp_amd_ctx *ctx)
{
  if (ctx->aux) return EXP_AMD_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // (stream priorities were tried for the aux stream: no effect on the overlap, 12.12-12.33 ms)
  // (experimental builds: EXP_AMD_AUX_CUS = n > 0 confines the auxiliary stream to n compute units spread over the XCDs --
  // does a bounded footprint of the HBM-bound passes leave the fp64-bound ones more of the part?  profiles/r06_overlap_ab.txt)
  const long long aux_cus = EXPAMD_EXPT("EXP_AMD_AUX_CUS", 0);
  if (aux_cus > 0 && aux_cus < 256) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // CU index = xcd + 8 * k in the mask's order on this part (round-robin over the XCDs): the first n bits spread evenly
    for (long long k = 0; k < aux_cus; k++) mask[k / 32] |= 1u << (k % 32);
    HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&ctx->aux, 8, mask));
  } else
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_sorted[k], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_forced[k], hipEventDisableTiming));
  }
  return EXP_AMD_OK;
}

extern "C" int exp_amd_ctx_synchronize(exp_amd_ctx *ctx)
{
  if (!ctx) return EXP_AMD_ERR_ARG;
  if (ctx->aux) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" void *exp_amd_ctx_stream(exp_amd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---- profiling ---------------------------------------------------------------------------

static hipEvent_t get_event(exp_amd_ctx *ctx)
{
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(exp_amd_ctx *c, const char *name, hipStream_t on) : ctx(c)
{
  if (!ctx || !ctx->profile) return;
  st = on ? on : ctx->stream;
  for (size_t i = 0; i < ctx->slots.size(); i++)
    if (ctx->slots[i].name == name || !strcmp(ctx->slots[i].name, name)) { slot = (int)i; break; }
  if (slot < 0) {
    ProfileSlot s;
    s.name = name;
    ctx->slots.push_back(s);
    slot = (int)ctx->slots.size() - 1;
  }
  e0 = get_event(ctx);
  e1 = get_event(ctx);
  (void)hipEventRecord(e0, st);
}

ProfScope::~ProfScope()
{
  if (slot < 0) return;
  (void)hipEventRecord(e1, st);
  ctx->slots[slot].pending.emplace_back(e0, e1);
}

static void drain_profile(exp_amd_ctx *ctx)
{
  for (auto &s : ctx->slots) {
    for (auto &p : s.pending) {
      (void)hipEventSynchronize(p.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
        s.ms_total += ms;
        s.launches += 1;
      }
      ctx->event_pool.push_back(p.first);
      ctx->event_pool.push_back(p.second);
    }
    s.pending.clear();
  }
}

extern "C" int exp_amd_profile_enable(exp_amd_ctx *ctx, int on)
{
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->profile = on != 0;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_profile_get(exp_amd_ctx *ctx, int idx, const char **name, double *ms_total,
                                   long long *launches)
{
  if (!ctx) return EXP_AMD_ERR_ARG;
  drain_profile(ctx);
  if (idx < 0 || idx >= (int)ctx->slots.size()) return EXP_AMD_ERR_ARG;
  if (name) *name = ctx->slots[idx].name;
  if (ms_total) *ms_total = ctx->slots[idx].ms_total;
  if (launches) *launches = ctx->slots[idx].launches;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_profile_reset(exp_amd_ctx *ctx)
{
  if (!ctx) return EXP_AMD_ERR_ARG;
  drain_profile(ctx);
  for (auto &s : ctx->slots) { s.ms_total = 0.0; s.launches = 0; }
  return EXP_AMD_OK;
}

// ---- collectives ---------------------------------------------------------------------------
// RCCL is bound lazily with dlopen so that the library loads (and single-rank runs work)
// on hosts without it.  Only ncclAllReduce(sum, double) on the coefficient buffer is used.

typedef struct { char internal[128]; } rccl_unique_id;
typedef int (*fn_get_uid)(rccl_unique_id *);
typedef int (*fn_init_rank)(void **, int, rccl_unique_id, int);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_errstr)(int);

static void *open_rccl()
{
  static void *lib = nullptr;
  if (lib) return lib;
  const char *names[] = {"librccl.so.1", "librccl.so", nullptr};
  for (int i = 0; names[i] && !lib; i++) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  return lib;
}

extern "C" int exp_amd_comm_get_unique_id(void *id128)
{
  void *lib = open_rccl();
  if (!lib) return expamd_fail(nullptr, EXP_AMD_ERR_COMM, "librccl not found: %s", dlerror());
  fn_get_uid f = (fn_get_uid)dlsym(lib, "ncclGetUniqueId");
  if (!f) return expamd_fail(nullptr, EXP_AMD_ERR_COMM, "ncclGetUniqueId missing");
  rccl_unique_id id;
  int rc = f(&id);
  if (rc) return expamd_fail(nullptr, EXP_AMD_ERR_COMM, "ncclGetUniqueId -> %d", rc);
  memcpy(id128, &id, sizeof(id));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comm_init_rank(exp_amd_ctx *ctx, const void *id128, int nranks, int rank)
{
  expamd_mutated();
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comm_init_rank: bad arguments");
  void *lib = open_rccl();
  if (!lib) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "librccl not found: %s", dlerror());
  fn_init_rank f = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
  if (!f) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "ncclCommInitRank missing");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  void *comm = nullptr;
  int rc = f(&comm, nranks, id, rank);
  if (rc) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "ncclCommInitRank -> %d", rc);
  void *ar = dlsym(lib, "ncclAllReduce");
  if (!ar) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "ncclAllReduce missing");
  ctx->rccl_allreduce = ar;
  ctx->rccl_lib = lib;
  ctx->rccl_comm = comm;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return EXP_AMD_OK;
}

extern "C" int exp_amd_comm_set_callback(exp_amd_ctx *ctx, exp_amd_allreduce_fn fn, void *user)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  if (!ctx) return EXP_AMD_ERR_ARG;
  ctx->ar_fn = fn;
  ctx->ar_user = user;
  return EXP_AMD_OK;
}

// the world a host-provided callback reduces over (exp_amd_comm_init_rank states its own)
extern "C" int exp_amd_comm_set_world(exp_amd_ctx *ctx, int nranks, int rank)
{
  expamd_mutated();
  if (!ctx || nranks < 1 || rank < 0 || rank >= nranks)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comm_set_world: bad arguments");
  if (ctx->rccl_comm && (nranks != ctx->nranks || rank != ctx->rank))
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comm_set_world: (%d, %d) contradicts the RCCL communicator's (%d, %d)",
                       nranks, rank, ctx->nranks, ctx->rank);
  ctx->nranks = nranks;
  ctx->rank = rank;
  return EXP_AMD_OK;
}

// The step driver runs two components' chains on two streams (host.hip).  Collectives of one RCCL communicator are
// ordered among themselves whatever stream they are given, so the second stream gets a communicator of its own: the first
// one split with every rank in the same colour (ncclCommSplit, RCCL >= 2.18).  Each rank issues the collectives of each
// communicator in the same order -- the driver's launch order does not depend on the data -- which is what NCCL asks of
// concurrently used communicators.  No split available: one stream, as before.
typedef int (*fn_comm_split)(void *, int, int, void **, void *);
static fn_comm_split comm_split_fn(exp_amd_ctx *ctx)
{
  return ctx->rccl_lib ? (fn_comm_split)dlsym(ctx->rccl_lib, "ncclCommSplit") : nullptr;
}

// A QUERY, free of side effects (exp_amd_comm_streams / comm_info may be called on one rank alone): the answer the
// collective below would give -- the twin exists, or it has not been asked for yet and the library can split.
bool expamd_comm_two_streams(exp_amd_ctx *ctx)
{
  if (ctx->ar_fn) return true;                       // (the callback is told the stream)
  if (!ctx->rccl_comm) return ctx->nranks <= 1;
  if (ctx->rccl_comm2) return true;
  return !ctx->rccl_comm2_tried && comm_split_fn(ctx) != nullptr;
}

// COLLECTIVE over the ranks of the RCCL communicator (ncclCommSplit is): makes the twin on first use.  Called by the
// step driver where it decides its schedule (host.hip: overlap_begin), which every rank reaches with the same
// components and settings.
bool expamd_comm_prepare_two_streams(exp_amd_ctx *ctx)
{
  if (ctx->ar_fn) return true;
  if (!ctx->rccl_comm) return ctx->nranks <= 1;
  if (!ctx->rccl_comm2 && !ctx->rccl_comm2_tried) {
    ctx->rccl_comm2_tried = true;
    fn_comm_split f = comm_split_fn(ctx);
    void *c2 = nullptr;
    if (f && f(ctx->rccl_comm, 0, ctx->rank, &c2, nullptr) == 0) ctx->rccl_comm2 = c2;
  }
  return ctx->rccl_comm2 != nullptr;
}

int expamd_allreduce(exp_amd_ctx *ctx, double *dev, size_t count)
{
  if (ctx->nranks > 1 && !ctx->ar_fn && !ctx->rccl_comm)
    return expamd_fail(ctx, EXP_AMD_ERR_COMM, "all-reduce over %d ranks asked for, but the context has neither an RCCL "
                       "communicator nor a callback", ctx->nranks);
  if (ctx->ar_fn) {
    ProfScope ps(ctx, "allreduce(callback)");
    int rc = ctx->ar_fn((void *)dev, count, (void *)ctx->stream, ctx->ar_user);
    if (rc) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "all-reduce callback returned %d", rc);
    ctx->ar_calls++;
    return EXP_AMD_OK;
  }
  if (ctx->rccl_comm) {
    ProfScope ps(ctx, "ncclAllReduce(coef)");
    fn_allreduce f = (fn_allreduce)ctx->rccl_allreduce;
    // ncclDouble = 8 (ncclFloat64), ncclSum = 0  (rccl.h:448, :467)
    void *comm = (ctx->aux && ctx->stream == ctx->aux && ctx->rccl_comm2) ? ctx->rccl_comm2 : ctx->rccl_comm;
    int rc = f(dev, dev, count, 8, 0, comm, ctx->stream);
    if (rc) return expamd_fail(ctx, EXP_AMD_ERR_COMM, "ncclAllReduce -> %d", rc);
    ctx->ar_calls++;
  }
  return EXP_AMD_OK;
}

// MAX over the ranks of one host number, on the transport the coefficient all-reduce uses (a SUM of one-hot slots): what
// rank-dependent host logic needs to agree on a count -- e.g. the number of batches, each ending in one all-reduce, that
// every rank must issue alike
extern "C" int exp_amd_comm_allreduce_max(exp_amd_ctx *ctx, double *value)
{
  if (!ctx || !value) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comm_allreduce_max: NULL argument");
  if (!ctx->ar_fn && !ctx->rccl_comm) return EXP_AMD_OK;                  // single rank
  if (ctx->nranks <= 1 && ctx->ar_fn)
    return expamd_fail(ctx, EXP_AMD_ERR_COMM, "comm_allreduce_max: the context has an all-reduce callback but was not told "
                       "its world (exp_amd_comm_set_world)");
  const int nr = ctx->nranks;
  std::vector<double> h((size_t)nr, 0.0);
  h[(size_t)ctx->rank] = *value;
  DevBuf<double> d;
  HIP_TRY(ctx, d.alloc((size_t)nr));
  HIP_TRY(ctx, hipMemcpyAsync(d.p, h.data(), nr * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = expamd_allreduce(ctx, d.p, (size_t)nr);
  if (rc == EXP_AMD_OK) {
    hipError_t e = hipMemcpyAsync(h.data(), d.p, nr * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = expamd_fail(ctx, EXP_AMD_ERR_HIP, "comm_allreduce_max: %s", hipGetErrorString(e));
  }
  d.release();
  if (rc) return rc;
  double m = h[0];
  for (int k = 1; k < nr; k++) m = h[k] > m ? h[k] : m;
  *value = m;
  return EXP_AMD_OK;
}

// on how many streams the context can reduce at once: 2 = the step driver keeps its two-stream schedule with this transport
// (a single rank, a host callback, or an RCCL communicator with a split twin), 1 = one stream
extern "C" int exp_amd_comm_streams(exp_amd_ctx *ctx) { return ctx && expamd_comm_two_streams(ctx) ? 2 : 1; }

// which all-reduce the context uses: kind 0 = none (single rank), 1 = the library's RCCL
// communicator, 2 = host-provided callback; calls = all-reduces issued so far
extern "C" int exp_amd_comm_info(exp_amd_ctx *ctx, int *kind, int *nranks, int *rank, long long *calls)
{
  if (!ctx) return EXP_AMD_ERR_ARG;
  if (kind) *kind = ctx->ar_fn ? 2 : ctx->rccl_comm ? 1 : 0;
  if (nranks) *nranks = ctx->nranks;
  if (rank) *rank = ctx->rank;
  if (calls) *calls = (long long)ctx->ar_calls;
  return EXP_AMD_OK;
}

// in-place sum of `count` doubles at device pointer `dev` over the ranks, on the context's stream:
// the collective the force methods use, exposed so that a host can check the communicator it set up
extern "C" int exp_amd_comm_allreduce(exp_amd_ctx *ctx, void *dev, size_t count)
{
  if (!ctx || !dev) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "comm_allreduce: NULL argument");
  return expamd_allreduce(ctx, (double *)dev, count);
}
