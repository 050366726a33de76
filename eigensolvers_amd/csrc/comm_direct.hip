// Direct all-gather of the operand: every rank WRITES its slice into the gathered buffers of its peers.
//
// xGMI on an MI355X node is point to point - 7 links per GPU, one per peer - so the exchange of a
// row-partitioned product is at its best when a rank sends its slice over all 7 links at once; a ring is bound by
// one link.  This backend does exactly that with plain stores from one kernel:
//
//   * every rank owns TWO gathered buffers (exchange e uses buffer e & 1) in one allocation and an array of arrival
//     flags in fine-grained memory; both are exported with hipIpcGetMemHandle and mapped by every peer
//     (hipeig_direct_alloc -> the host side all-gathers the handles over its TCP group -> hipeig_direct_attach);
//   * `push` (communication stream): one kernel per chunk of the layout reads the rank's piece once and stores it
//     into the same place of every peer's buffer; its last workgroup, after a system-scope release, stores the exchange
//     number into flag[buffer][chunk][this rank] of every peer;
//   * `wait` (compute stream): a one-wavefront kernel spins (system-scope acquire loads) until the flags of all peers for
//     that chunk have reached the exchange number, with a wall-clock limit after which it raises an error word in
//     mapped host memory and returns - no wave ever waits forever.  The sweep that needs the chunk is the NEXT kernel,
//     so remote data is only read behind a kernel boundary.
//
// Why two buffers suffice without any "done reading" handshake: a peer starts pushing exchange e + 2 only after its
// own sweep of exchange e + 1, which waited for MY push of e + 1, which my compute stream issued after my sweep of e.
// So nobody overwrites a buffer that is still being read, provided every product waits for all chunks (it does).
//
// The small all-reduces (MINRES scalars, dots) have a direct form too - mailboxes in the same fine-grained area, see the
// end of this file - so a communicator can run without RCCL altogether (hipeig_comm_init_direct); whole-vector sums and
// the exchange of block operands stay on RCCL.
#include "common.h"

#define DIRECT_MBOX_DOUBLES 1024                   // doubles one small all-reduce moves per rank
#define DIRECT_GATHER_FLAGS (2 * HIPEIG_GATHER_MAX_CHUNKS * HIPEIG_MAX_RANKS)
#define DIRECT_MBOX_FLAGS (2 * HIPEIG_MAX_RANKS)
// fine-grained area of a rank, in 8-byte words: arrival flags of the exchange, flags of the mailboxes, the mailboxes
#define DIRECT_MBOX_OFFSET (DIRECT_GATHER_FLAGS + DIRECT_MBOX_FLAGS)
#define DIRECT_FLAG_WORDS (DIRECT_MBOX_OFFSET + 2 * HIPEIG_MAX_RANKS * DIRECT_MBOX_DOUBLES)
#define DIRECT_WAIT_LIMIT_S 120                        // HIPEIG_DIRECT_WAIT_S: the skew between ranks a wait tolerates

struct DirectComm {
  int nranks, rank;
  int64_t capacity;                         // doubles per buffer
  double* base;                             // own allocation: 2 * capacity doubles
  uint64_t* flags;                          // own flags (fine-grained): [2][chunks][ranks]
  double* peer_base[HIPEIG_MAX_RANKS];      // base of every rank as mapped here (own: base)
  uint64_t* peer_flags[HIPEIG_MAX_RANKS];
  bool attached;
  uint64_t seq;                             // exchanges begun so far; the last one uses buffer seq & 1
  uint64_t rseq;                            // small all-reduces begun so far (mailbox rseq & 1)
  unsigned* d_ticket;
  int* h_err;                               // mapped host word: a wait kernel gave up
  int* d_err;
  int64_t wait_limit_ticks;                 // of the 100 MHz wall clock
};

struct PeerTable {
  double* dst[HIPEIG_MAX_RANKS];
  uint64_t* flag[HIPEIG_MAX_RANKS];
  int n;
};

// src -> the same offset of every peer buffer (16-byte stores), then flag them.
__global__ void __launch_bounds__(HIPEIG_BLOCK)
direct_push_kernel(const double2* __restrict__ src, int64_t n16, PeerTable peers, uint64_t seq, unsigned* ticket) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const double2 v = src[i];
    for (int p = 0; p < peers.n; ++p) reinterpret_cast<double2*>(peers.dst[p])[i] = v;
  }
  __threadfence_system();                              // this thread's stores have reached their destinations
  __shared__ int sh_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (sh_last && (int)threadIdx.x < peers.n) {
    __hip_atomic_store(peers.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Spin until flags[r] >= seq for every r < n with r != self (one lane per rank); gives up after `limit` clock ticks.
__global__ void __launch_bounds__(64)
direct_wait_kernel(const uint64_t* flags, int n, int self, uint64_t seq, int64_t limit, int* err) {
  const int r = threadIdx.x;
  if (r >= n || r == self) return;
  const int64_t t0 = wall_clock64();
  while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    if (wall_clock64() - t0 > limit) {
      __hip_atomic_store(err, 1 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}

static void direct_close_peers(DirectComm* d) {
  for (int r = 0; r < d->nranks; ++r) {
    if (r == d->rank) continue;
    if (d->peer_base[r]) hipIpcCloseMemHandle(d->peer_base[r]);
    if (d->peer_flags[r]) hipIpcCloseMemHandle(d->peer_flags[r]);
    d->peer_base[r] = nullptr; d->peer_flags[r] = nullptr;
  }
  d->attached = false;
}

int hipeig_direct_destroy(hipeig_ctx* c) {
  DirectComm* d = c->direct;
  if (!d) return 0;
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->comm_stream);
  direct_close_peers(d);
  if (d->base) hipFree(d->base);
  if (d->flags) hipFree(d->flags);
  if (d->d_ticket) hipFree(d->d_ticket);
  if (d->h_err) hipHostFree(d->h_err);
  free(d);
  c->direct = nullptr;
  c->gather_backend = 0;
  return 0;
}

// Allocate this rank's two gathered buffers (capacity doubles each) and its flags; handles_out receives a 192-byte record:
// the two 64-byte hipIpc handles (buffers, flags) and the PCI bus id of this rank's device (so that a peer can check that
// it may address this device before it ever stores to it).  A previous allocation is released first.
extern "C" int hipeig_direct_alloc(hipeig_ctx* c, int64_t capacity_doubles, void* handles_out /* 192 bytes */) {
  HIPEIG_REQUIRE(capacity_doubles > 0 && handles_out, "bad arguments");
  HIPEIG_REQUIRE(c->nranks <= HIPEIG_MAX_RANKS, "too many ranks for the direct exchange");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  HIPEIG_CHECK(hipSetDevice(c->device));
  if (c->direct) hipeig_direct_destroy(c);
  DirectComm* d = (DirectComm*)calloc(1, sizeof(DirectComm));
  HIPEIG_REQUIRE(d != nullptr, "out of host memory");
  d->nranks = c->nranks; d->rank = c->rank;
  d->capacity = (capacity_doubles + 31) & ~(int64_t)31;
  c->direct = d;                                       // owned by the context from here on (destroy releases what exists)
  HIPEIG_CHECK(hipMalloc((void**)&d->base, (size_t)2 * d->capacity * sizeof(double)));
  HIPEIG_CHECK(hipMemset(d->base, 0, (size_t)2 * d->capacity * sizeof(double)));
  HIPEIG_CHECK(hipExtMallocWithFlags((void**)&d->flags, DIRECT_FLAG_WORDS * sizeof(uint64_t), hipDeviceMallocFinegrained));
  HIPEIG_CHECK(hipMemset(d->flags, 0, DIRECT_FLAG_WORDS * sizeof(uint64_t)));
  HIPEIG_CHECK(hipMalloc((void**)&d->d_ticket, 64));
  HIPEIG_CHECK(hipMemset(d->d_ticket, 0, 64));
  HIPEIG_CHECK(hipHostMalloc((void**)&d->h_err, 64, hipHostMallocMapped));
  *d->h_err = 0;
  HIPEIG_CHECK(hipHostGetDevicePointer((void**)&d->d_err, d->h_err, 0));
  d->wait_limit_ticks = (int64_t)DIRECT_WAIT_LIMIT_S * 100000000LL;
  if (const char* e = getenv("HIPEIG_DIRECT_WAIT_S")) d->wait_limit_ticks = (int64_t)(atof(e) * 1e8);
  HIPEIG_CHECK(hipDeviceSynchronize());
  hipIpcMemHandle_t hb, hf;
  HIPEIG_CHECK(hipIpcGetMemHandle(&hb, d->base));
  HIPEIG_CHECK(hipIpcGetMemHandle(&hf, d->flags));
  memcpy(handles_out, &hb, 64);
  memcpy((char*)handles_out + 64, &hf, 64);
  char bus[64];
  memset(bus, 0, sizeof(bus));
  HIPEIG_CHECK(hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, c->device));
  memcpy((char*)handles_out + 128, bus, 64);
  return 0;
}

// Map every peer's buffers and flags: all_handles holds nranks records of 192 bytes in rank order (this rank's own
// record is skipped).  Collective in effect: every rank must have allocated before any rank attaches.  Refuses - before
// anything is mapped or written - when a peer's device is not visible here or cannot be addressed from this one: a store
// to an unreachable peer would be a GPU fault, not an error code.
#define DIRECT_RECORD 192
extern "C" int hipeig_direct_attach(hipeig_ctx* c, const void* all_handles) {
  DirectComm* d = c->direct;
  HIPEIG_REQUIRE(d != nullptr && all_handles, "hipeig_direct_alloc first");
  HIPEIG_CHECK(hipSetDevice(c->device));
  int ndev = 0;
  HIPEIG_CHECK(hipGetDeviceCount(&ndev));
  char mine[64];
  memset(mine, 0, sizeof(mine));
  HIPEIG_CHECK(hipDeviceGetPCIBusId(mine, (int)sizeof(mine) - 1, c->device));
  for (int r = 0; r < d->nranks; ++r) {
    if (r == d->rank) continue;
    char bus[64];
    memcpy(bus, (const char*)all_handles + (size_t)r * DIRECT_RECORD + 128, 64);
    bus[63] = 0;
    if (strcmp(bus, mine) == 0) continue;                // a rank on this very device (several processes on one GPU)
    int peer = -1;
    for (int p = 0; p < ndev && peer < 0; ++p) {
      char pb[64];
      memset(pb, 0, sizeof(pb));
      if (hipDeviceGetPCIBusId(pb, (int)sizeof(pb) - 1, p) == hipSuccess && strcmp(pb, bus) == 0) peer = p;
    }
    (void)hipGetLastError();
    if (peer < 0) {
      hipeig_set_error("direct exchange: the device of rank %d (%s) is not visible in this process", r, bus);
      return 4;
    }
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, c->device, peer) != hipSuccess || !can) {
      (void)hipGetLastError();
      hipeig_set_error("direct exchange: device %d cannot address the device of rank %d (%s)", c->device, r, bus);
      return 4;
    }
    (void)hipDeviceEnablePeerAccess(peer, 0);              // already enabled: fine
    (void)hipGetLastError();
  }
  direct_close_peers(d);
  for (int r = 0; r < d->nranks; ++r) {
    if (r == d->rank) { d->peer_base[r] = d->base; d->peer_flags[r] = d->flags; continue; }
    hipIpcMemHandle_t hb, hf;
    memcpy(&hb, (const char*)all_handles + (size_t)r * DIRECT_RECORD, 64);
    memcpy(&hf, (const char*)all_handles + (size_t)r * DIRECT_RECORD + 64, 64);
    HIPEIG_CHECK(hipIpcOpenMemHandle((void**)&d->peer_base[r], hb, hipIpcMemLazyEnablePeerAccess));
    HIPEIG_CHECK(hipIpcOpenMemHandle((void**)&d->peer_flags[r], hf, hipIpcMemLazyEnablePeerAccess));
  }
  d->attached = true;
  d->seq = 0;
  d->rseq = 0;
  if (!c->comm && !c->loop) { c->gather_backend = 1; c->allreduce_backend = 1; }     // a direct-only communicator has nothing else
  return 0;
}

// 0 = RCCL's all-gather, 1 = direct peer writes (needs hipeig_direct_attach).  Collective in effect: every rank must
// switch at the same point of its call sequence, with no exchange in flight.
extern "C" int hipeig_comm_set_gather_backend(hipeig_ctx* c, int backend) {
  HIPEIG_REQUIRE(backend == 0 || backend == 1, "backend must be 0 (RCCL) or 1 (direct)");
  HIPEIG_REQUIRE(backend == 0 || (c->direct && c->direct->attached), "the direct exchange is not attached");
  HIPEIG_REQUIRE(backend == 1 || c->comm || c->loop, "no RCCL communicator to switch to");
  HIPEIG_REQUIRE(backend == 0 || c->x_full_n <= c->direct->capacity,
                 "the direct exchange buffers are smaller than the gathered operand of an existing operator");
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->comm_stream));
  c->gather_backend = backend;
  return 0;
}

// The same choice for the small all-reduces (<= 1024 doubles: MINRES scalars, dots, Gram blocks).
extern "C" int hipeig_comm_set_allreduce_backend(hipeig_ctx* c, int backend) {
  HIPEIG_REQUIRE(backend == 0 || backend == 1, "backend must be 0 (RCCL) or 1 (direct)");
  HIPEIG_REQUIRE(backend == 0 || (c->direct && c->direct->attached), "the direct exchange is not attached");
  HIPEIG_REQUIRE(backend == 1 || c->comm || c->loop, "no RCCL communicator to switch to");
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  c->allreduce_backend = backend;
  return 0;
}

// info[0] = backend in use, [1] = direct exchange attached, [2] = capacity (doubles per buffer), [3] = exchanges begun,
// [4] = error word of the wait kernels (0 = none; 1 + r: rank r's data did not arrive within the limit)
extern "C" int hipeig_comm_gather_info(hipeig_ctx* c, int64_t info[8]) {
  memset(info, 0, 8 * sizeof(int64_t));
  info[0] = c->gather_backend;
  if (c->direct) {
    info[1] = c->direct->attached; info[2] = c->direct->capacity; info[3] = (int64_t)c->direct->seq;
    info[4] = *c->direct->h_err;
  }
  info[5] = c->gather_chunks;
  info[6] = c->allreduce_backend;
  return 0;
}

int hipeig_direct_reserve(hipeig_ctx* c, int64_t doubles) {
  DirectComm* d = c->direct;
  if (!d) return 0;
  if (doubles > d->capacity) {
    hipeig_set_error("the direct exchange buffers hold %lld doubles, this operator needs %lld: enlarge them "
                     "(HipContext.enable_direct_gather) before creating it", (long long)d->capacity, (long long)doubles);
    return 4;
  }
  return 0;
}

double* hipeig_direct_next_buffer(hipeig_ctx* c) {
  DirectComm* d = c->direct;
  return d->base + (int64_t)((d->seq + 1) & 1) * d->capacity;
}

double* hipeig_direct_current_buffer(hipeig_ctx* c) {
  DirectComm* d = c->direct;
  return d->base + (int64_t)(d->seq & 1) * d->capacity;
}

static inline uint64_t* flag_slot(uint64_t* flags, int buf, int chunk, int src) {
  return flags + ((size_t)buf * HIPEIG_GATHER_MAX_CHUNKS + chunk) * HIPEIG_MAX_RANKS + src;
}

// Called by hipeig_allgather_x_begin after this rank's pieces are in its NEXT buffer and the communication stream has
// been ordered behind that copy: push chunk after chunk to every peer.
int hipeig_direct_begin(hipeig_ctx* c, const GatherLayout& gl, int64_t n_local) {
  DirectComm* d = c->direct;
  HIPEIG_REQUIRE(d && d->attached, "the direct exchange is not attached");
  HIPEIG_REQUIRE(gl.total() <= d->capacity, "the direct exchange buffers are smaller than the gathered operand");
  if (*d->h_err) {
    hipeig_set_error("direct exchange: the slice of rank %d did not arrive within the wait limit", *d->h_err - 1);
    return 4;
  }
  const uint64_t e = ++d->seq;
  const int buf = (int)(e & 1);
  for (int k = 0; k < gl.nchunks; ++k) {
    // the piece of this chunk incl. (last chunk) the scalar slot behind it; whole 128-byte lines
    const int64_t lo = (int64_t)k * gl.h;
    int64_t rows = n_local - lo;
    if (rows > gl.h) rows = gl.h;
    if (rows < 0) rows = 0;
    int64_t len = (k == gl.nchunks - 1) ? gl.cstride(k) : ((rows + 15) & ~(int64_t)15);
    if (len > gl.cstride(k)) len = gl.cstride(k);
    const int64_t off = (int64_t)buf * d->capacity + gl.cbase[k] + (int64_t)c->rank * gl.cstride(k);
    PeerTable pt;
    pt.n = 0;
    for (int r = 0; r < d->nranks; ++r) {
      if (r == d->rank) continue;
      pt.dst[pt.n] = d->peer_base[r] + off;
      pt.flag[pt.n] = flag_slot(d->peer_flags[r], buf, k, d->rank);
      ++pt.n;
    }
    if (pt.n == 0) continue;
    const int64_t n16 = len / 2;
    int g = (int)((n16 + HIPEIG_BLOCK * 4 - 1) / (HIPEIG_BLOCK * 4));
    if (g < 1) g = 1;
    if (g > 2 * c->num_cu) g = 2 * c->num_cu;           // the links, not the CUs, bound this kernel; leave the sweep its CUs
    hipLaunchKernelGGL(direct_push_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->comm_stream,
                       reinterpret_cast<const double2*>(d->base + off), n16, pt, e, d->d_ticket);
  }
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

int hipeig_direct_wait_chunk(hipeig_ctx* c, const GatherLayout& gl, int chunk) {
  DirectComm* d = c->direct;
  if (d->nranks == 1) return 0;
  const int buf = (int)(d->seq & 1);
  hipLaunchKernelGGL(direct_wait_kernel, dim3(1), dim3(64), 0, c->stream, flag_slot(d->flags, buf, chunk, 0), d->nranks,
                     d->rank, d->seq, d->wait_limit_ticks, d->d_err);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// ---- small all-reduce through the peers' mailboxes -----------------------------------------------------------------
// A SUM over the ranks of <= DIRECT_MBOX_DOUBLES doubles (MINRES: two per iteration) without RCCL: every rank stores its
// values into mailbox [rseq & 1][this rank] of EVERY rank (its own included) and then flags them; a second one-workgroup
// kernel waits (bounded) for the flags of all ranks and adds the mailboxes in rank order, so every rank obtains the
// identical sum.  Two mailboxes suffice for the reason given at the top: a rank begins all-reduce s + 2 only after it
// has finished s + 1, for which it needed every peer's s + 1, which each peer sent after finishing s.
struct MboxTable {
  double* box[HIPEIG_MAX_RANKS];           // mailbox of THIS rank inside rank p's area
  uint64_t* flag[HIPEIG_MAX_RANKS];
  int n;
};

__global__ void __launch_bounds__(HIPEIG_BLOCK)
direct_mbox_push_kernel(const double* __restrict__ src, int count, MboxTable t, uint64_t seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const double v = src[i];
    for (int p = 0; p < t.n; ++p) __hip_atomic_store(t.box[p] + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < t.n) __hip_atomic_store(t.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
direct_mbox_sum_kernel(double* __restrict__ dst, int count, const double* boxes /* [ranks][DIRECT_MBOX_DOUBLES] */,
                       const uint64_t* flags, int n, uint64_t seq, int64_t limit, int* err) {
  __shared__ int sh_bad;
  if (threadIdx.x == 0) sh_bad = 0;
  __syncthreads();
  if ((int)threadIdx.x < n) {
    const int64_t t0 = wall_clock64();
    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > limit) {
        __hip_atomic_store(err, 1 + (int)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sh_bad = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (sh_bad) return;
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    double a = 0.0;
    for (int r = 0; r < n; ++r) a += __hip_atomic_load(boxes + (size_t)r * DIRECT_MBOX_DOUBLES + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    dst[i] = a;
  }
}

// In-place SUM all-reduce of `count` doubles on stream s (pieces of DIRECT_MBOX_DOUBLES).
int hipeig_direct_allreduce(hipeig_ctx* c, double* buf, int64_t count, hipStream_t s) {
  DirectComm* d = c->direct;
  HIPEIG_REQUIRE(d && d->attached, "the direct exchange is not attached");
  if (*d->h_err) {
    hipeig_set_error("direct all-reduce: the contribution of rank %d did not arrive within the wait limit", *d->h_err - 1);
    return 4;
  }
  for (int64_t o = 0; o < count; o += DIRECT_MBOX_DOUBLES) {
    const int n = (int)((count - o < DIRECT_MBOX_DOUBLES) ? count - o : DIRECT_MBOX_DOUBLES);
    const uint64_t e = ++d->rseq;
    const int par = (int)(e & 1);
    MboxTable t;
    t.n = d->nranks;
    for (int p = 0; p < d->nranks; ++p) {
      double* area = reinterpret_cast<double*>(d->peer_flags[p] + DIRECT_MBOX_OFFSET);
      t.box[p] = area + ((size_t)par * HIPEIG_MAX_RANKS + d->rank) * DIRECT_MBOX_DOUBLES;
      t.flag[p] = d->peer_flags[p] + DIRECT_GATHER_FLAGS + (size_t)par * HIPEIG_MAX_RANKS + d->rank;
    }
    hipLaunchKernelGGL(direct_mbox_push_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, s, buf + o, n, t, e);
    const double* mine = reinterpret_cast<const double*>(d->flags + DIRECT_MBOX_OFFSET) + (size_t)par * HIPEIG_MAX_RANKS * DIRECT_MBOX_DOUBLES;
    hipLaunchKernelGGL(direct_mbox_sum_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, s, buf + o, n, mine,
                       d->flags + DIRECT_GATHER_FLAGS + (size_t)par * HIPEIG_MAX_RANKS, d->nranks, e, d->wait_limit_ticks, d->d_err);
  }
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

bool hipeig_direct_ready(const hipeig_ctx* c) { return c->direct && c->direct->attached; }

int hipeig_sync_checked(hipeig_ctx* c) {
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  if (c->direct && c->direct->h_err && *c->direct->h_err) {
    hipeig_set_error("direct exchange: the data of rank %d did not arrive within the wait limit (HIPEIG_DIRECT_WAIT_S); "
                     "results computed since are invalid", *c->direct->h_err - 1);
    return 4;
  }
  return 0;
}

// Wall-clock limit of the wait kernels from now on (seconds; <= 0 restores the default / HIPEIG_DIRECT_WAIT_S).  A trial of
// the backend on hardware it has never run on uses a short one so that a dead link costs seconds, not minutes.
extern "C" int hipeig_comm_set_wait_limit(hipeig_ctx* c, double seconds) {
  DirectComm* d = c->direct;
  HIPEIG_REQUIRE(d != nullptr, "no direct exchange allocated");
  if (seconds <= 0.0) {
    seconds = DIRECT_WAIT_LIMIT_S;
    if (const char* e = getenv("HIPEIG_DIRECT_WAIT_S")) seconds = atof(e);
  }
  d->wait_limit_ticks = (int64_t)(seconds * 1e8);
  return 0;
}

// Release the buffers of the direct exchange (a collective fall-back to RCCL would otherwise keep two gathered operands
// per rank allocated for nothing).
extern "C" int hipeig_direct_release(hipeig_ctx* c) { return hipeig_direct_destroy(c); }
