// RCCL glue: one process per GPU, communicator attached to the context.
//
// librccl.so is opened lazily with dlopen so that single-GPU use neither loads nor needs
// it.  Only three collectives exist on this path: an all-gather of
// the operand slice before every operator application, a SUM all-reduce after every
// reduction, and a tiny all-gather of the row counts when an operator is created.
//
// A second, in-process backend ("loopback") runs the same collectives between several contexts of
// ONE process, one host thread per rank, through a host barrier and device-to-device copies.  It
// exists so that the multi-rank code (row partition, column remap, split sweeps, partial sums from
// several ranks) can be rehearsed on a box with a single GPU, where RCCL refuses two ranks on one
// device; it is host-synchronous and not meant to be fast.
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <time.h>
#include "common.h"

typedef void* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId_t;
enum { NCCL_SUM = 0, NCCL_INT64 = 4, NCCL_FLOAT64 = 8 };

struct RcclApi {
  void* handle;
  int (*GetUniqueId)(ncclUniqueId_t*);
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int);
  int (*CommDestroy)(ncclComm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(int);
};
static RcclApi g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

static char g_rccl_path[512] = "";

// ROCm's own RCCL first: a bare "librccl.so" resolves to whatever copy is already mapped into the
// process (e.g. the one bundled with a PyTorch wheel, built for another ROCm), which made the library
// used depend on import order.  HIPEIG_RCCL_LIB overrides the search.
static int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {getenv("HIPEIG_RCCL_LIB"), "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1",
                         "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    hipeig_set_error("cannot dlopen librccl.so: %s", dlerror());
    return 3;
  }
#define LOAD(field, sym)                                              \
  *(void**)(&g_rccl.field) = dlsym(h, sym);                           \
  if (!g_rccl.field) {                                                \
    hipeig_set_error("librccl.so lacks symbol %s", sym);              \
    return 3;                                                         \
  }
  LOAD(GetUniqueId, "ncclGetUniqueId");
  LOAD(CommInitRank, "ncclCommInitRank");
  LOAD(CommDestroy, "ncclCommDestroy");
  LOAD(AllReduce, "ncclAllReduce");
  LOAD(AllGather, "ncclAllGather");
  LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
  g_rccl.handle = h;
  Dl_info di;
  if (dladdr((void*)g_rccl.AllReduce, &di) && di.dli_fname) snprintf(g_rccl_path, sizeof(g_rccl_path), "%s", di.dli_fname);
  return 0;
}

#define RCCL_CHECK(expr)                                                                  \
  do {                                                                                    \
    int _r = (expr);                                                                      \
    if (_r != 0) {                                                                        \
      hipeig_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
      return 4;                                                                           \
    }                                                                                     \
  } while (0)

// ---- loopback group ---------------------------------------------------------------------
#define LOOP_MAX_RANKS 64
struct LoopGroup {
  int n;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int arrived;
  unsigned generation;
  const void* ptr[LOOP_MAX_RANKS];
};

// Generation barrier; a rank that waits longer than 120 s reports an error instead of hanging.
static int loop_barrier(LoopGroup* g) {
  struct timespec dl;
  clock_gettime(CLOCK_REALTIME, &dl);
  dl.tv_sec += 120;
  pthread_mutex_lock(&g->mu);
  const unsigned gen = g->generation;
  if (++g->arrived == g->n) {
    g->arrived = 0;
    ++g->generation;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
    return 0;
  }
  int rc = 0;
  while (gen == g->generation && rc == 0) rc = pthread_cond_timedwait(&g->cv, &g->mu, &dl);
  const bool ok = gen != g->generation;
  if (!ok) --g->arrived;
  pthread_mutex_unlock(&g->mu);
  if (!ok) {
    hipeig_set_error("loopback collective: a peer rank did not arrive within 120 s");
    return 4;
  }
  return 0;
}

extern "C" int hipeig_loopback_group_create(int nranks, void** out) {
  HIPEIG_REQUIRE(out && nranks >= 1 && nranks <= LOOP_MAX_RANKS, "bad loopback group size");
  LoopGroup* g = (LoopGroup*)calloc(1, sizeof(LoopGroup));
  HIPEIG_REQUIRE(g != nullptr, "out of host memory");
  g->n = nranks;
  pthread_mutex_init(&g->mu, nullptr);
  pthread_cond_init(&g->cv, nullptr);
  *out = g;
  return 0;
}

extern "C" int hipeig_loopback_group_destroy(void* group) {
  LoopGroup* g = (LoopGroup*)group;
  if (!g) return 0;
  pthread_cond_destroy(&g->cv);
  pthread_mutex_destroy(&g->mu);
  free(g);
  return 0;
}

static void comm_flags_from_env(hipeig_ctx* c, int nranks) {
  // HIPEIG_FORCE_COLLECTIVES=1 keeps the all-gather / all-reduce path active on a one-rank
  // communicator, so the RCCL plumbing can be exercised on a single-GPU box.
  const char* force = getenv("HIPEIG_FORCE_COLLECTIVES");
  c->collectives = (nranks > 1) || (force && atoi(force) != 0);
  // HIPEIG_OVERLAP=0 turns the all-gather / local-window overlap off (default on)
  const char* ov = getenv("HIPEIG_OVERLAP");
  c->overlap = c->collectives && !(ov && atoi(ov) == 0);
}

extern "C" int hipeig_comm_init_loopback(hipeig_ctx* c, void* group, int rank) {
  LoopGroup* g = (LoopGroup*)group;
  HIPEIG_REQUIRE(g != nullptr && rank >= 0 && rank < g->n, "bad loopback group / rank");
  HIPEIG_REQUIRE(c->comm == nullptr && c->loop == nullptr, "communicator already attached");
  c->loop = g;
  c->nranks = g->n;
  c->rank = rank;
  comm_flags_from_env(c, g->n);
  c->row_counts = (int64_t*)calloc((size_t)g->n, sizeof(int64_t));
  return 0;
}

// All ranks contribute `bytes` from `send`; rank r's block lands at recv + r*bytes (in place when
// send == recv + rank*bytes).  Host-synchronous.
static int loop_allgather(hipeig_ctx* c, const void* send, void* recv, size_t bytes, hipStream_t s) {
  LoopGroup* g = (LoopGroup*)c->loop;
  HIPEIG_CHECK(hipStreamSynchronize(s));                 // my block is final
  g->ptr[c->rank] = send;
  if (loop_barrier(g)) return 4;
  for (int r = 0; r < g->n; ++r) {
    char* dst = (char*)recv + (size_t)r * bytes;
    if ((const void*)dst != g->ptr[r])
      HIPEIG_CHECK(hipMemcpyAsync(dst, g->ptr[r], bytes, hipMemcpyDeviceToDevice, s));
  }
  HIPEIG_CHECK(hipStreamSynchronize(s));
  return loop_barrier(g);                                // nobody rewrites its block before all have read it
}

// SUM in rank order, so every rank obtains the identical value.
static int loop_allreduce_f64(hipeig_ctx* c, double* buf, int count, hipStream_t s) {
  LoopGroup* g = (LoopGroup*)c->loop;
  HIPEIG_CHECK(hipStreamSynchronize(s));
  g->ptr[c->rank] = buf;
  if (loop_barrier(g)) return 4;
  double* acc = (double*)calloc((size_t)count * 2, sizeof(double));
  HIPEIG_REQUIRE(acc != nullptr, "out of host memory");
  double* tmp = acc + count;
  for (int r = 0; r < g->n; ++r) {
    const hipError_t e = hipMemcpy(tmp, g->ptr[r], (size_t)count * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(acc); HIPEIG_CHECK(e); }
    for (int i = 0; i < count; ++i) acc[i] += tmp[i];
  }
  int rc = loop_barrier(g);                              // all have read before anyone overwrites
  if (rc == 0) {
    const hipError_t e = hipMemcpy(buf, acc, (size_t)count * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { free(acc); HIPEIG_CHECK(e); }
  }
  free(acc);
  return rc;
}

static int coll_allgather(hipeig_ctx* c, const void* send, void* recv, size_t count, int type, hipStream_t s) {
  if (c->loop) return loop_allgather(c, send, recv, count * 8, s);      // both types used here are 8 bytes wide
  RCCL_CHECK(g_rccl.AllGather(send, recv, count, type, (ncclComm_t)c->comm, s));
  return 0;
}

// Path of the RCCL shared object the collectives run on ("" before the first communicator call).
extern "C" int hipeig_comm_library(char* path, int path_len) {
  HIPEIG_REQUIRE(path && path_len > 0, "bad buffer");
  snprintf(path, (size_t)path_len, "%s", g_rccl_path);
  return 0;
}

extern "C" int hipeig_comm_unique_id(void* id128) {
  if (load_rccl()) return 3;
  ncclUniqueId_t id;
  RCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int hipeig_comm_init(hipeig_ctx* c, int nranks, int rank, const void* id128) {
  HIPEIG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank/nranks");
  HIPEIG_REQUIRE(c->comm == nullptr && c->loop == nullptr, "communicator already attached");
  if (load_rccl()) return 3;
  HIPEIG_CHECK(hipSetDevice(c->device));
  ncclUniqueId_t id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  RCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->comm = comm;
  c->nranks = nranks;
  c->rank = rank;
  comm_flags_from_env(c, nranks);
  c->row_counts = (int64_t*)calloc((size_t)nranks, sizeof(int64_t));
  return 0;
}

extern "C" int hipeig_comm_destroy(hipeig_ctx* c) {
  if (c->comm) {
    hipStreamSynchronize(c->stream);
    g_rccl.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  c->loop = nullptr;                 // the group belongs to whoever created it
  c->collectives = 0;
  c->nranks = 1;
  c->rank = 0;
  return 0;
}

extern "C" int hipeig_comm_info(hipeig_ctx* c, int* nranks, int* rank) {
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return 0;
}

// stats[0] = collectives (operand all-gathers + all-reduces) the most recent hipeig_minres call issued on
// this rank; stats[1..3] reserved.
extern "C" int hipeig_comm_stats(hipeig_ctx* c, int64_t stats[4]) {
  stats[0] = c->mr_collectives; stats[1] = stats[2] = stats[3] = 0;
  return 0;
}

// SUM all-reduce of `count` doubles in place on the compute stream.
int hipeig_allreduce_sum(hipeig_ctx* c, double* d_buf, int count) {
  if (!c->collectives) return 0;
  if (c->loop) return loop_allreduce_f64(c, d_buf, count, c->stream);
  RCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM,
                              (ncclComm_t)c->comm, c->stream));
  return 0;
}

// Replica mode (FEAST contour points spread over GPUs): the communicator stays
// attached but operators and vectors are whole on every rank, so reductions and products must NOT
// go through the collectives; only hipeig_vec_allreduce does.  partitioned = 1 restores the default.
extern "C" int hipeig_comm_set_partitioned(hipeig_ctx* c, int partitioned) {
  HIPEIG_REQUIRE(c->comm != nullptr || c->loop != nullptr, "no communicator attached");
  if (partitioned) comm_flags_from_env(c, c->nranks);
  else { c->collectives = 0; c->overlap = 0; }
  return 0;
}

// SUM over all ranks of a whole vector, in place (compute stream).
extern "C" int hipeig_vec_allreduce(hipeig_ctx* c, double* v, int64_t n) {
  HIPEIG_REQUIRE(c->comm != nullptr || c->loop != nullptr, "no communicator attached");
  HIPEIG_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "bad length");
  if (n == 0) return 0;
  if (c->loop) return loop_allreduce_f64(c, v, (int)n, c->stream);
  RCCL_CHECK(g_rccl.AllReduce(v, v, (size_t)n, NCCL_FLOAT64, NCCL_SUM, (ncclComm_t)c->comm, c->stream));
  return 0;
}

// Gather the row counts of every rank (host result in ctx->row_counts) and size the
// gathered-operand buffer: rank r's slice lives at x_full + r*stride, stride = max count.
int hipeig_comm_setup_rows(hipeig_ctx* c, int64_t nrows_local, int64_t* stride_out) {
  if (!c->collectives) {
    *stride_out = nrows_local;
    return 0;
  }
  int64_t* d = (int64_t*)c->d_scalars;
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));            // the pinned staging word below is free
  int64_t* staged = (int64_t*)c->h_scalars;
  staged[0] = nrows_local;
  HIPEIG_CHECK(hipMemcpyAsync(d + c->rank, staged, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  if (coll_allgather(c, d + c->rank, d, 1, NCCL_INT64, c->stream)) return 4;
  HIPEIG_CHECK(hipMemcpyAsync(c->row_counts, d, sizeof(int64_t) * c->nranks, hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  int64_t stride = 0;
  for (int r = 0; r < c->nranks; ++r) stride = c->row_counts[r] > stride ? c->row_counts[r] : stride;
  const int64_t need = stride * c->nranks;
  if (need > c->x_full_n) {
    if (c->x_full) HIPEIG_CHECK(hipFree(c->x_full));
    HIPEIG_CHECK(hipMalloc((void**)&c->x_full, (size_t)need * sizeof(double)));
    HIPEIG_CHECK(hipMemsetAsync(c->x_full, 0, (size_t)need * sizeof(double), c->stream));
    c->x_full_n = need;
  }
  *stride_out = stride;
  return 0;
}

// All-gather of the operand: every rank contributes its slice, in place inside x_full.
int hipeig_allgather_x(hipeig_ctx* c, const double* x_local, int64_t n_local, int64_t stride,
                       const double** x_full_out) {
  if (!c->collectives) {
    *x_full_out = x_local;
    return 0;
  }
  HIPEIG_REQUIRE(stride >= n_local && stride * c->nranks <= c->x_full_n, "operand buffer smaller than the partition");
  double* mine = c->x_full + (int64_t)c->rank * stride;
  HIPEIG_CHECK(hipMemcpyAsync(mine, x_local, (size_t)n_local * sizeof(double),
                              hipMemcpyDeviceToDevice, c->stream));
  if (coll_allgather(c, mine, c->x_full, (size_t)stride, NCCL_FLOAT64, c->stream)) return 4;
  *x_full_out = c->x_full;
  return 0;
}

// All-gather of `count` doubles per rank into recv (rank r's block at recv + r*count; `send` may be the
// caller's own block inside recv), on the compute stream.
int hipeig_allgather_f64(hipeig_ctx* c, const double* send, double* recv, size_t count) {
  HIPEIG_REQUIRE(c->collectives, "no communicator");
  return coll_allgather(c, send, recv, count, NCCL_FLOAT64, c->stream);
}

// Split form of the operand all-gather: `begin` issues the copy + ncclAllGather on the
// communication stream (ordered after everything already queued on the compute stream),
// `end` makes the compute stream wait for it.  Between the two the caller may launch work that
// reads only x_local.
int hipeig_allgather_x_begin(hipeig_ctx* c, const double* x_local, int64_t n_local, int64_t stride) {
  HIPEIG_REQUIRE(c->collectives, "no communicator");
  HIPEIG_REQUIRE(stride >= n_local && stride * c->nranks <= c->x_full_n, "operand buffer smaller than the partition");
  double* mine = c->x_full + (int64_t)c->rank * stride;
  HIPEIG_CHECK(hipEventRecord(c->ev_x, c->stream));
  HIPEIG_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_x, 0));
  HIPEIG_CHECK(hipMemcpyAsync(mine, x_local, (size_t)n_local * sizeof(double), hipMemcpyDeviceToDevice, c->comm_stream));
  if (coll_allgather(c, mine, c->x_full, (size_t)stride, NCCL_FLOAT64, c->comm_stream)) return 4;
  HIPEIG_CHECK(hipEventRecord(c->ev_comm, c->comm_stream));
  return 0;
}

int hipeig_allgather_x_end(hipeig_ctx* c, const double** x_full_out) {
  HIPEIG_CHECK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
  *x_full_out = c->x_full;
  return 0;
}
