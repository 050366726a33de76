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
  const char* gc = getenv("HIPEIG_GATHER_CHUNKS");          // 0 / unset: automatic (pick_gather_chunks)
  c->gather_chunks = gc ? atoi(gc) : 0;
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

int hipeig_direct_allreduce(hipeig_ctx* c, double* buf, int64_t count, hipStream_t s);      // comm_direct.hip
bool hipeig_direct_ready(const hipeig_ctx* c);

static int coll_allgather(hipeig_ctx* c, const void* send, void* recv, size_t count, int type, hipStream_t s) {
  if (c->loop) return loop_allgather(c, send, recv, count * 8, s);      // both types used here are 8 bytes wide
  if (!c->comm) {
    hipeig_set_error("this exchange (a block operand, or an operator created before the direct exchange was attached) "
                     "needs the RCCL communicator; the context has only the direct peer-write backend");
    return 4;
  }
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

int hipeig_direct_destroy(hipeig_ctx* c);

// A communicator WITHOUT RCCL: rank / size only; every exchange then goes through the direct peer-write backend, which
// must be attached (hipeig_direct_alloc / _attach) before the first operator is created.  What it cannot do is the
// exchange of block operands (block products and lock-step block solves on a partitioned operator need RCCL).
extern "C" int hipeig_comm_init_direct(hipeig_ctx* c, int nranks, int rank) {
  HIPEIG_REQUIRE(nranks >= 1 && nranks <= HIPEIG_MAX_RANKS && rank >= 0 && rank < nranks, "bad rank/nranks");
  HIPEIG_REQUIRE(c->comm == nullptr && c->loop == nullptr, "communicator already attached");
  c->nranks = nranks;
  c->rank = rank;
  comm_flags_from_env(c, nranks);
  c->gather_backend = 1;
  c->allreduce_backend = 1;
  free(c->row_counts);
  c->row_counts = (int64_t*)calloc((size_t)nranks, sizeof(int64_t));
  return 0;
}

extern "C" int hipeig_comm_destroy(hipeig_ctx* c) {
  if (c->direct) hipeig_direct_destroy(c);
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
  if (!c->comm || (c->allreduce_backend == 1 && count <= 1024 && hipeig_direct_ready(c)))
    return hipeig_direct_allreduce(c, d_buf, count, c->stream);
  RCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM,
                              (ncclComm_t)c->comm, c->stream));
  return 0;
}

// Replica mode (FEAST contour points spread over GPUs): the communicator stays
// attached but operators and vectors are whole on every rank, so reductions and products must NOT
// go through the collectives; only hipeig_vec_allreduce does.  partitioned = 1 restores the default.
extern "C" int hipeig_comm_set_partitioned(hipeig_ctx* c, int partitioned) {
  HIPEIG_REQUIRE(c->comm != nullptr || c->loop != nullptr || hipeig_direct_ready(c), "no communicator attached");
  if (partitioned) comm_flags_from_env(c, c->nranks);
  else { c->collectives = 0; c->overlap = 0; }
  return 0;
}

// SUM over all ranks of a whole vector, in place (compute stream).
extern "C" int hipeig_vec_allreduce(hipeig_ctx* c, double* v, int64_t n) {
  HIPEIG_REQUIRE(c->comm != nullptr || c->loop != nullptr || hipeig_direct_ready(c), "no communicator attached");
  HIPEIG_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "bad length");
  if (n == 0) return 0;
  if (c->loop) return loop_allreduce_f64(c, v, (int)n, c->stream);
  if (!c->comm) return hipeig_direct_allreduce(c, v, n, c->stream);
  RCCL_CHECK(g_rccl.AllReduce(v, v, (size_t)n, NCCL_FLOAT64, NCCL_SUM, (ncclComm_t)c->comm, c->stream));
  return 0;
}

// Chunks the operand exchange is cut into: HIPEIG_GATHER_CHUNKS, else 2 when the slices are long enough for the sweep
// of one chunk's column windows to hide the transfer of the next (>= 4 windows of 2^17 columns per chunk) and the
// overlap is on, else 1.
static int pick_gather_chunks(const hipeig_ctx* c, int64_t max_rows) {
  int n = c->gather_chunks;
  if (n <= 0) n = (c->overlap && c->nranks > 1 && max_rows >= ((int64_t)8 << 17)) ? 2 : 1;
  if (n > HIPEIG_GATHER_MAX_CHUNKS) n = HIPEIG_GATHER_MAX_CHUNKS;
  while (n > 1 && max_rows / n < 16) --n;
  return n;
}

// Layout for slices of at most max_rows rows (the same on every rank).
void hipeig_gather_layout(const hipeig_ctx* c, int64_t max_rows, GatherLayout* gl) {
  const int nch = pick_gather_chunks(c, max_rows);
  int64_t h = (max_rows + nch - 1) / nch;
  const int64_t W = (int64_t)1 << 17;                       // the widest column window of the blocked sweeps
  if (h >= 4 * W) h = (h + W - 1) / W * W;                   // slices start on window boundaries: every window has one owner
  else h = (h + 15) / 16 * 16;                               // whole 128-byte lines
  if (h < 16) h = 16;
  gl->nranks = c->nranks; gl->nchunks = nch; gl->h = h;
  int64_t pos = 0;
  for (int k = 0; k < nch; ++k) { gl->cbase[k] = pos; pos += (int64_t)c->nranks * gl->cstride(k); }
  for (int k = nch; k <= HIPEIG_GATHER_MAX_CHUNKS; ++k) gl->cbase[k] = pos;
}

int hipeig_direct_reserve(hipeig_ctx* c, int64_t doubles);     // comm_direct.hip
int hipeig_direct_begin(hipeig_ctx* c, const GatherLayout& gl, int64_t n_local);
int hipeig_direct_wait_chunk(hipeig_ctx* c, const GatherLayout& gl, int chunk);
double* hipeig_direct_next_buffer(hipeig_ctx* c);
double* hipeig_direct_current_buffer(hipeig_ctx* c);

// Gather the row counts of every rank (host result in ctx->row_counts), fix the layout of the gathered operand for
// slices of that size and make sure the buffer holds it.
int hipeig_comm_setup_rows(hipeig_ctx* c, int64_t nrows_local, GatherLayout* gl) {
  if (!c->collectives) {
    gl->nranks = 1; gl->nchunks = 1; gl->h = nrows_local > 16 ? nrows_local : 16;
    gl->cbase[0] = 0;
    for (int k = 1; k <= HIPEIG_GATHER_MAX_CHUNKS; ++k) gl->cbase[k] = gl->h + HIPEIG_SLOT_DOUBLES;
    return 0;
  }
  int64_t* d = (int64_t*)c->d_scalars;
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));            // the pinned staging word below is free
  int64_t* staged = (int64_t*)c->h_scalars;
  if (!c->comm && !c->loop) {
    // direct-only communicator: the counts as a SUM of one-hot records of doubles (exact below 2^53)
    HIPEIG_REQUIRE(hipeig_direct_ready(c), "attach the direct exchange before creating an operator");
    double* hd = c->h_scalars;
    for (int r = 0; r < c->nranks; ++r) hd[r] = (r == c->rank) ? (double)nrows_local : 0.0;
    HIPEIG_CHECK(hipMemcpyAsync(c->d_scalars, hd, sizeof(double) * c->nranks, hipMemcpyHostToDevice, c->stream));
    if (hipeig_direct_allreduce(c, c->d_scalars, c->nranks, c->stream)) return 4;
    HIPEIG_CHECK(hipMemcpyAsync(hd, c->d_scalars, sizeof(double) * c->nranks, hipMemcpyDeviceToHost, c->stream));
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->nranks; ++r) c->row_counts[r] = (int64_t)hd[r];
  } else {
    staged[0] = nrows_local;
    HIPEIG_CHECK(hipMemcpyAsync(d + c->rank, staged, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (coll_allgather(c, d + c->rank, d, 1, NCCL_INT64, c->stream)) return 4;
    HIPEIG_CHECK(hipMemcpyAsync(c->row_counts, d, sizeof(int64_t) * c->nranks, hipMemcpyDeviceToHost, c->stream));
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  }
  int64_t max_rows = 0;
  for (int r = 0; r < c->nranks; ++r) max_rows = c->row_counts[r] > max_rows ? c->row_counts[r] : max_rows;
  hipeig_gather_layout(c, max_rows, gl);
  const int64_t need = gl->total();
  if (need > c->x_full_n) {
    HIPEIG_CHECK(hipStreamSynchronize(c->comm_stream));
    if (c->x_full) HIPEIG_CHECK(hipFree(c->x_full));
    c->x_full = nullptr; c->x_full_n = 0;
    HIPEIG_CHECK(hipMalloc((void**)&c->x_full, (size_t)need * sizeof(double)));
    HIPEIG_CHECK(hipMemsetAsync(c->x_full, 0, (size_t)need * sizeof(double), c->stream));
    c->x_full_n = need;
  }
  // the direct buffers must hold the operand only where they are (or will be, checked at the switch) the ones in use
  if (hipeig_direct_ready(c) && c->gather_backend == 1 && hipeig_direct_reserve(c, need)) return 4;
  return 0;
}

// The buffer the NEXT operand exchange of this context fills (the direct backend alternates between two).
static double* gather_buffer_next(hipeig_ctx* c) { return (c->direct && c->gather_backend == 1) ? hipeig_direct_next_buffer(c) : c->x_full; }
static double* gather_buffer_current(hipeig_ctx* c) { return (c->direct && c->gather_backend == 1) ? hipeig_direct_current_buffer(c) : c->x_full; }

// This rank's scalar slot inside the buffer of the NEXT exchange: what a kernel stores there before
// hipeig_allgather_x[_begin] travels with the operand and is found at gathered + gl.slot(r) on every rank.
double* hipeig_gather_slot(hipeig_ctx* c, const GatherLayout& gl) {
  if (!c->collectives) return nullptr;
  return gather_buffer_next(c) + gl.slot(c->rank);
}

// Phase events of a partitioned product (hipeig_phase_timing): 0 begin, 1 local sweep done, 2 first remote launch may
// start, 3 product done (compute stream); 4 exchange starts, 5 exchange done (communication stream).
static inline void phase_mark(hipeig_ctx* c, int k, hipStream_t s) {
  if (c->phase_timing) hipEventRecord(c->ev_ph[k], s);
}
void hipeig_phase_mark(hipeig_ctx* c, int k) { phase_mark(c, k, c->stream); }

// Start the operand exchange: the slice goes into its places in the gathered buffer on the compute stream, then the
// communication stream moves chunk after chunk (one collective each, in place) and records an event per chunk.
int hipeig_allgather_x_begin(hipeig_ctx* c, const GatherLayout& gl, const double* x_local, int64_t n_local) {
  HIPEIG_REQUIRE(c->collectives, "no communicator");
  HIPEIG_REQUIRE(gl.nranks == c->nranks && n_local <= gl.h * gl.nchunks && gl.total() <= c->x_full_n,
                 "operand buffer smaller than the partition");
  phase_mark(c, 0, c->stream);
  double* buf = gather_buffer_next(c);
  for (int k = 0; k < gl.nchunks; ++k) {
    const int64_t lo = (int64_t)k * gl.h, hi = (lo + gl.h < n_local) ? lo + gl.h : n_local;
    if (hi > lo)
      HIPEIG_CHECK(hipMemcpyAsync(buf + gl.pos(c->rank, lo), x_local + lo, (size_t)(hi - lo) * sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
  }
  if (c->exchange_off) return 0;                             // measurement aid: the peers' parts keep their previous values
  HIPEIG_CHECK(hipEventRecord(c->ev_x, c->stream));
  HIPEIG_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_x, 0));
  phase_mark(c, 4, c->comm_stream);
  if (c->direct && c->gather_backend == 1) {
    if (hipeig_direct_begin(c, gl, n_local)) return 4;
  } else {
    for (int k = 0; k < gl.nchunks; ++k) {
      double* region = buf + gl.cbase[k];
      if (coll_allgather(c, region + (int64_t)c->rank * gl.cstride(k), region, (size_t)gl.cstride(k), NCCL_FLOAT64, c->comm_stream)) return 4;
      HIPEIG_CHECK(hipEventRecord(c->ev_chunk[k], c->comm_stream));
    }
  }
  phase_mark(c, 5, c->comm_stream);
  return 0;
}

// The compute stream waits until chunk `chunk` (and every earlier one) of the exchange begun last has arrived.
int hipeig_allgather_x_wait_chunk(hipeig_ctx* c, const GatherLayout& gl, int chunk) {
  if (c->exchange_off) return 0;
  if (c->direct && c->gather_backend == 1) return hipeig_direct_wait_chunk(c, gl, chunk);
  HIPEIG_CHECK(hipStreamWaitEvent(c->stream, c->ev_chunk[chunk], 0));
  return 0;
}

const double* hipeig_gathered(hipeig_ctx* c) { return gather_buffer_current(c); }

// Whole exchange, compute stream ordered after it.
int hipeig_allgather_x(hipeig_ctx* c, const GatherLayout& gl, const double* x_local, int64_t n_local, const double** x_full_out) {
  if (!c->collectives) {
    *x_full_out = x_local;
    return 0;
  }
  if (hipeig_allgather_x_begin(c, gl, x_local, n_local)) return 4;
  if (hipeig_allgather_x_wait_chunk(c, gl, gl.nchunks - 1)) return 4;
  *x_full_out = gather_buffer_current(c);
  return 0;
}

// All-gather of `count` doubles per rank into recv (rank r's block at recv + r*count; `send` may be the
// caller's own block inside recv), on the compute stream.
int hipeig_allgather_f64(hipeig_ctx* c, const double* send, double* recv, size_t count) {
  HIPEIG_REQUIRE(c->collectives, "no communicator");
  return coll_allgather(c, send, recv, count, NCCL_FLOAT64, c->stream);
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
block_rows_copy_kernel(int64_t n16, const double2* __restrict__ src, double2* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// Interleaved block operand of width K: the layout of the single-vector exchange with every position K doubles wide
// (chunk c at cbase[c]*K, per-rank stride cstride(c)*K; the slot of a rank is the K*HIPEIG_SLOT_DOUBLES doubles behind
// its rows of the last chunk: K scalars per rank ride along).  On the compute stream, one collective per chunk.
int hipeig_block_reserve(hipeig_ctx* c, const GatherLayout& gl, int K) {
  if (!c->collectives) return 0;
  const int64_t need = gl.total() * K;
  if (c->xb_full_n < need) {
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    if (c->xb_full) HIPEIG_CHECK(hipFree(c->xb_full));
    c->xb_full = nullptr; c->xb_full_n = 0;
    HIPEIG_CHECK(hipMalloc((void**)&c->xb_full, (size_t)need * sizeof(double)));
    HIPEIG_CHECK(hipMemsetAsync(c->xb_full, 0, (size_t)need * sizeof(double), c->stream));
    c->xb_full_n = need;
  }
  return 0;
}

int hipeig_allgather_block(hipeig_ctx* c, const GatherLayout& gl, int K, const double* xb_local, int64_t n_local,
                           const double** xb_full_out) {
  if (!c->collectives) { *xb_full_out = xb_local; return 0; }
  if (hipeig_block_reserve(c, gl, K)) return 1;
  for (int k = 0; k < gl.nchunks; ++k) {
    const int64_t lo = (int64_t)k * gl.h, hi = (lo + gl.h < n_local) ? lo + gl.h : n_local;
    if (hi > lo) {
      const int64_t n16 = (hi - lo) * K / 2;
      hipLaunchKernelGGL(block_rows_copy_kernel, dim3(grid_stream(n16 * 2)), dim3(HIPEIG_BLOCK), 0, c->stream, n16,
                         reinterpret_cast<const double2*>(xb_local + lo * K),
                         reinterpret_cast<double2*>(c->xb_full + gl.pos(c->rank, lo) * K));
    }
  }
  HIPEIG_CHECK(hipGetLastError());
  for (int k = 0; k < gl.nchunks; ++k) {
    double* region = c->xb_full + gl.cbase[k] * K;
    if (coll_allgather(c, region + (int64_t)c->rank * gl.cstride(k) * K, region, (size_t)(gl.cstride(k) * K), NCCL_FLOAT64, c->stream)) return 4;
  }
  *xb_full_out = c->xb_full;
  return 0;
}

// this rank's K-scalar slot inside the block buffer (see hipeig_allgather_block)
double* hipeig_gather_block_slot(hipeig_ctx* c, const GatherLayout& gl, int K) {
  if (!c->collectives || !c->xb_full) return nullptr;
  return c->xb_full + gl.slot(c->rank) * K;
}

// Chunks of the operand exchange for operators created FROM NOW ON (0 = automatic, pick_gather_chunks); existing operators
// keep the layout their columns were remapped to.  Every rank must set the same value.
// Measurement aid: on = 0 makes the products of this context place their own slice in the gathered buffer and skip the
// exchange with the peers (whose parts keep the values of the last real exchange), so that ONE rank's sweeps - own
// windows + remaining windows, the compute part of a partitioned product - can be timed alone on a shared GPU.
// Results are meaningless while it is off.  Not collective; switch it back on before the next collective product.
extern "C" int hipeig_comm_set_exchange(hipeig_ctx* c, int on) {
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->comm_stream));
  c->exchange_off = on ? 0 : 1;
  return 0;
}

extern "C" int hipeig_comm_set_gather_chunks(hipeig_ctx* c, int nchunks) {
  HIPEIG_REQUIRE(nchunks >= 0 && nchunks <= HIPEIG_GATHER_MAX_CHUNKS, "chunks must be 0 (automatic) .. 4");
  c->gather_chunks = nchunks;
  return 0;
}

// ---- measurement hooks (bench.py) -------------------------------------------------------------------------------
// Phase timing of row-partitioned products: on != 0 makes every product record events on both streams.
extern "C" int hipeig_phase_timing(hipeig_ctx* c, int on) {
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->comm_stream));
  c->phase_timing = on ? 1 : 0;
  return 0;
}

// Times of the most recent product, in ms (a negative value: that phase did not occur): out[0] the operand exchange
// (communication stream), [1] sweep of this rank's own windows (under the exchange), [2] sweep of the remaining windows
// incl. the waits for later chunks, [3] whole product on the compute stream, [4] compute stream idle between the own
// windows and the first chunk's arrival.
extern "C" int hipeig_phase_get(hipeig_ctx* c, double out[8]) {
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->comm_stream));
  const int pairs[5][2] = {{4, 5}, {0, 1}, {2, 3}, {0, 3}, {1, 2}};
  for (int k = 0; k < 8; ++k) out[k] = -1.0;
  for (int k = 0; k < 5; ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_ph[pairs[k][0]], c->ev_ph[pairs[k][1]]) == hipSuccess) out[k] = ms;
    else (void)hipGetLastError();
  }
  return 0;
}

// reps SUM all-reduces of `count` doubles back to back on the compute stream; *ms_each = average time of one.
extern "C" int hipeig_comm_bench_allreduce(hipeig_ctx* c, int count, int reps, double* ms_each) {
  HIPEIG_REQUIRE(count >= 1 && count <= 1024 && reps >= 1 && ms_each, "bad arguments");
  double* buf = c->d_scalars + 2048;
  HIPEIG_CHECK(hipMemsetAsync(buf, 0, (size_t)count * sizeof(double), c->stream));
  if (hipeig_allreduce_sum(c, buf, count)) return 4;                   // warm-up
  HIPEIG_CHECK(hipEventRecord(c->ev0, c->stream));
  for (int k = 0; k < reps; ++k)
    if (hipeig_allreduce_sum(c, buf, count)) return 4;
  HIPEIG_CHECK(hipEventRecord(c->ev1, c->stream));
  HIPEIG_CHECK(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  HIPEIG_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_each = ms / reps;
  return 0;
}
