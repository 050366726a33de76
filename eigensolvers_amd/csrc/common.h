// Internal definitions shared by the translation units of libhipeig.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/hipeig.h"

#define HIPEIG_WAVE 64
#define HIPEIG_BLOCK 256                 // 4 wavefronts per workgroup
#define HIPEIG_MAX_PARTIALS 2048         // reduction grid cap: 256 CUs x 8 workgroups
#define HIPEIG_MAX_COLS 16               // basis columns handled per tall-skinny launch
#define HIPEIG_WIDE_PARTIALS 32768       // grid cap of the reductions that finish in their LAST workgroup (no consumer sums partials)
#define HIPEIG_GATHER_MAX_CHUNKS 4
#define HIPEIG_SLOT_DOUBLES 16           // per-rank scalar slot riding on the operand all-gather (one 128-byte line)
#define HIPEIG_MAX_RANKS 64

// Layout of the all-gathered operand of a row-partitioned operator ("chunk-major"): every rank's row slice is cut
// into `nchunks` pieces of h rows; chunk c of ALL ranks is contiguous ([chunk][rank][h]), so that one collective per
// chunk fills one contiguous column range and the sweep of chunk c's column windows can run while chunk c+1 is still
// travelling.  The last chunk's per-rank stride is h + HIPEIG_SLOT_DOUBLES: the extra line carries a few scalars of
// the sending rank (MINRES: its share of <y,y>), which therefore reach every rank WITH the operand, no second collective.
struct GatherLayout {
  int32_t nranks, nchunks;
  int64_t h;                                       // rows per (rank, chunk)
  int64_t cbase[HIPEIG_GATHER_MAX_CHUNKS + 1];     // first position of chunk c; cbase[nchunks] = total length
  __host__ __device__ int64_t cstride(int c) const { return h + (c == nchunks - 1 ? HIPEIG_SLOT_DOUBLES : 0); }
  __host__ __device__ int64_t pos(int rank, int64_t i) const {      // local row i of `rank` -> position in the gathered operand
    int c = (int)(i / h);
    if (c >= nchunks) c = nchunks - 1;             // never taken for i < nchunks*h; keeps a bad index in range
    return cbase[c] + (int64_t)rank * cstride(c) + (i - (int64_t)c * h);
  }
  __host__ __device__ int64_t slot(int rank) const { return cbase[nchunks - 1] + (int64_t)rank * cstride(nchunks - 1) + h; }
  __host__ __device__ int64_t total() const { return cbase[nchunks]; }
};

void hipeig_set_error(const char* fmt, ...);

#define HIPEIG_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      hipeig_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

#define HIPEIG_REQUIRE(cond, msg)                                                       \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      hipeig_set_error("%s:%d: requirement failed: %s (%s)", __FILE__, __LINE__, #cond, msg); \
      return 2;                                                                         \
    }                                                                                   \
  } while (0)

// Device-resident recurrence state of the MINRES driver (see minres.hip).
struct MinresState {
  double beta1, beta, oldb, alfa;
  double dbar, epsln, oldeps, delta, gbar, gamma, cs, sn, phi, phibar;
  double tnorm2, gmax, gmin, root;
  double Anorm, ynorm, rnorm, test1, test2, Acond;
  double s;          // 1/beta of the Lanczos vector being built (v = s*r)
  double denom;      // 1/gamma
  int itn;           // completed iterations
  int istop;         // SciPy stop code, 0 = running
  int pending_m1;    // istop = -1 requested at itn == 1
  int done;          // iteration finished: later kernels of the chunk exit immediately
};

struct hipeig_ctx {
  int device;
  int num_cu;
  hipStream_t stream;        // compute
  hipStream_t comm_stream;   // collectives / copies overlapped with compute
  hipEvent_t ev0, ev1, ev_comm, ev_x;
  // reduction workspace
  double* d_partials;        // HIPEIG_MAX_PARTIALS * (HIPEIG_MAX_COLS*HIPEIG_MAX_COLS) doubles
  size_t partials_doubles;
  double* d_scalars;         // small device scalar area (results of reductions)
  double* h_scalars;         // pinned host mirror
  double* h_scalars_dev;     // the same buffer as the device sees it (null if it cannot be mapped)
  size_t scalars_doubles;
  const double** d_ptrs;     // device pointer tables for tall-skinny kernels
  const double** h_ptrs;     // pinned staging of the same
  size_t ptrs_count;
  // MINRES workspace (7 vectors: r1 r2 y, w1 w2 w, x) cached across solves
  double* mr_ws;
  int64_t mr_ws_n;
  // hipGraph of one 18-iteration MINRES chunk, replayed while its key (operator layout, shift,
  // tolerances, workspace) is unchanged
  hipGraphExec_t mr_graph;
  void* mr_graph_key;        // malloc'ed copy of the key the graph was captured for
  size_t mr_graph_key_bytes;
  int use_graph;             // HIPEIG_GRAPH (default 0)
  MinresState* d_mr_state;   // ring of 3
  MinresState* h_mr_state;   // pinned
  // distributed
  void* comm;                // ncclComm_t
  void* loop;                // in-process loopback group (rehearsal backend), or null
  int nranks, rank;
  int collectives;           // 1 when reductions / operator applications must go through RCCL
  int64_t mr_collectives;    // collectives issued by the most recent hipeig_minres call (all-gathers + all-reduces)
  double* x_full;            // all-gathered operand of the operator
  int64_t x_full_n;
  int64_t* row_counts;       // rows per rank (host), length nranks
  double* blk_ws;            // interleaved operand / result blocks of hipeig_spmm
  size_t blk_ws_doubles;
  double* xb_full;           // all-gathered interleaved operand block ([stride*nranks][8])
  int64_t xb_full_n;         // its capacity in doubles
  // block MINRES (minres_block.hip): 7 interleaved blocks and 3 x 8 + 8 recurrence records
  double* mrb_ws;
  int64_t mrb_ws_n;
  MinresState* d_mrb_state;  // ring of 3 x 8 records
  MinresState* h_mrb_state;  // pinned, 8 records
  int overlap;               // 1: all-gather on the comm stream while the local-column windows are swept
  double* ytmp;              // raw partial sums handed from the local-window launch to the remote one
  int64_t ytmp_n;
  int gather_chunks;         // chunks the operand all-gather is cut into (HIPEIG_GATHER_CHUNKS; 0 = automatic)
  hipEvent_t ev_chunk[HIPEIG_GATHER_MAX_CHUNKS];   // chunk c of the current all-gather has arrived (comm stream)
  unsigned* d_counters;      // arrival counters of the reductions that finish in their last workgroup (zero between kernels)
  double* d_group_partials;  // group sums of the record reductions (blas1.hip, finish_records): 32 groups x 1024 values
  // phase timing of a partitioned product (hipeig_phase_timing): events on both streams
  int phase_timing;
  hipEvent_t ev_ph[8];
  hipEvent_t ev_stage;       // behind the last asynchronous copy out of the pinned staging buffers (hipeig_lincomb_block)
  hipEvent_t ev_slot[16];    // one per pinned result slot of the split Arnoldi step (hipeig_pair_arnoldi_step_begin)
  // side streams of the split Arnoldi step (hipeig_pair_arnoldi_step_begin): the steps of the right-hand sides of a lock-step
  // block solve are independent, so slot s runs on side stream s % arn_nstreams with that stream's own workspace
  hipStream_t arn_stream[16];
  hipEvent_t ev_arn_in[16];  // "the compute stream has produced this slot's operands"
  int arn_nstreams;          // 0: not created yet; 1: everything on the compute stream
  double* d_arn_ws;          // per side stream: partial areas, two total records, the step's result record
  unsigned* d_arn_cnt;       // per side stream: ticket counters
  void* h_arn_items;         // pinned, mapped: the items of a batched Arnoldi step (hipeig_pair_arnoldi_step_batch_begin)
  void* d_arn_items;         // the same as the device sees it
  // direct all-gather backend (comm_direct.hip): peers' operand buffers and flags mapped through hipIpc
  struct DirectComm* direct;
  int gather_backend;        // operand exchange: 0 = RCCL (or loopback), 1 = direct peer writes
  int allreduce_backend;     // small (<= 1024 doubles) all-reduces: 0 = RCCL, 1 = the peers' mailboxes (comm_direct.hip)
  int exchange_off;          // measurement aid (hipeig_comm_set_exchange): products place the own slice and skip the exchange
};

struct hipeig_csr {
  int64_t nrows, ncols, nnz, row_offset;
  int32_t* d_rowptr;         // nrows+1
  int32_t* d_col;            // nnz
  double* d_val;             // nnz
  int32_t* d_row_blocks;     // n_row_blocks+1 row indices: block b owns rows [rb[b], rb[b+1])
  int32_t n_row_blocks;
  // column-window blocked copy (see spmv_device.h, "TCOO"); built on first use
  uint32_t* t_idx;
  double* t_val;
  uint32_t* t_off;
  int32_t t_nunits, t_nwin, t_wbits, t_rw, t_wgs_per_sweep, t_prefetch;
  // workgroup-wide column-bucketed copy ("TCOO-W", variant 4); built on first use
  uint32_t* w_idx;
  double* w_val;
  uint32_t* w_off;
  int32_t w_nunits, w_nwin, w_wbits, w_rw, w_wgs_per_sweep;
  int32_t w_csplit;          // workgroups sharing one row block (column splits), 1 = none
  int32_t w_binbits, w_align; // columns per bin (log2) and whether bins are aligned to 64-element instruction groups
  int64_t w_slots;           // stream length incl. padding slots (== nnz without alignment)
  // copy of the same layout for the pair sweep (two accumulators per row: units of half the rows), built
  // on the first hipeig_spmv_shift_pair of a large operator
  uint32_t* p_idx;
  double* p_val;
  uint32_t* p_off;
  int32_t p_nunits, p_nwin, p_wbits, p_rw, p_wgs_per_sweep;
  int32_t last_pair_fused;   // 1 when the most recent pair product ran as one sweep
  int reproducible;          // automatic choice restricted to bitwise reproducible kernels (hipeig_csr_set_reproducible)
  double absrow_max;         // max_i sum_j |a_ij| over the local rows: overflow bound of the fixed-point sweep (variant 5)
  // block-operand copies ("TCOO-B", spmm_device.h): one per interleave width, [0]: K = 4, [1]: K = 8, [2]: K = 16; built on first use
  struct BcooLayout {
    uint32_t* idx;
    double* val;
    uint32_t* off;
    int32_t nunits, nwin, wbits, rw, wgs_per_sweep;
    int32_t state;           // 0 = undecided, 1 = built, 2 = not suited (row-owner kernel is used)
  } bl[3];
  int32_t block_variant;     // 0 = automatic, 1 = row-owner CSR, 2 = TCOO-B
  int32_t last_block_variant, last_block_k;
  int64_t gather_len;        // length of the gathered operand (ncols, or gl.total())
  GatherLayout gl;           // layout of the gathered operand when the columns were remapped (col_stride > 0)
  int variant;               // 0 = auto, 1 = CSR-vector, 2 = CSR-stream, 3 = TCOO (wave units), 4 = TCOO-W, 5 = TCOO-W with fixed-point accumulators
  int last_variant;          // variant used by the most recent launch (0 = none yet)
  int last_launches;         // kernel launches (sweeps) one product with that variant takes
  int lanes_per_row;         // sub-wave width used to reduce one row
  int64_t col_stride;        // > 0 when the columns were remapped to the gathered layout `gl` (= gl.h); 0 = global columns
  int64_t bytes;
};

// Synchronise the compute stream and report a direct-exchange wait that gave up (comm_direct.hip): every path that hands
// a result to the host goes through this, so a timed-out wait is an error of THAT call, not of some later one.
int hipeig_sync_checked(hipeig_ctx* c);

// ---- collectives (comm.hip); no-ops without a communicator ---------------------------
int hipeig_comm_setup_rows(hipeig_ctx* ctx, int64_t nrows_local, GatherLayout* gl_out);
int hipeig_allreduce_sum(hipeig_ctx* ctx, double* d_buf, int count);
// whole exchange on the compute stream; *x_full_out = the gathered operand (x_local itself without a communicator)
int hipeig_allgather_x(hipeig_ctx* ctx, const GatherLayout& gl, const double* x_local, int64_t n_local,
                       const double** x_full_out);
int hipeig_allgather_f64(hipeig_ctx* ctx, const double* send, double* recv, size_t count);
// split form: `begin` places this rank's slice in the gathered buffer (compute stream) and starts the exchange chunk by
// chunk on the communication stream; `wait_chunk` makes the compute stream wait until chunk c of every rank is there.
int hipeig_allgather_x_begin(hipeig_ctx* ctx, const GatherLayout& gl, const double* x_local, int64_t n_local);
int hipeig_allgather_x_wait_chunk(hipeig_ctx* ctx, const GatherLayout& gl, int chunk);
// interleaved block operand of width K (doubles per position): same layout scaled by K, in ctx->xb_full
int hipeig_allgather_block(hipeig_ctx* ctx, const GatherLayout& gl, int K, const double* xb_local, int64_t n_local,
                           const double** xb_full_out);
double* hipeig_gather_slot(hipeig_ctx* ctx, const GatherLayout& gl);     // this rank's scalar slot inside x_full

// ---- device helpers ------------------------------------------------------------------
// Separately rounded multiply / add.  hipcc contracts a*b+c into an FMA by default and the
// __dmul_rn/__dadd_rn intrinsics are plain operators in ROCm, so an explicit pragma is the
// only way to keep two roundings where the reference's NumPy expression has two.
__host__ __device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__host__ __device__ __forceinline__ double add_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}

// Sum over the workgroup; result valid in thread 0.  lds must hold >= 4 doubles.
__device__ __forceinline__ double block_reduce_sum(double v, double* lds) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    r = lds[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w];
  }
  __syncthreads();
  return r;
}

// Every thread of the workgroup obtains the fixed-order sum of p[0..count) (count <=
// HIPEIG_MAX_PARTIALS).  Used in kernel prologues so that a reduction never needs its own
// launch or a host round trip; every workgroup computes the identical value.
// The summation tree is that of a 256-thread workgroup whatever the launch shape (the 1024-thread sweep kernels
// leave their upper waves out), so a scalar does not depend on WHICH kernel's prologue reduces it.
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ p, int count,
                                                     double* lds) {
  const int nt = blockDim.x < HIPEIG_BLOCK ? (int)blockDim.x : HIPEIG_BLOCK;
  double a = 0.0;
  if ((int)threadIdx.x < nt)
    for (int i = threadIdx.x; i < count; i += nt) a += p[i];
  a = wave_reduce_sum(a);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) lds[wid] = a;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < (nt >> 6); ++w) r += lds[w];
  __syncthreads();
  return r;
}

// ---- reductions that finish in their last workgroup ------------------------------------------------------------
// Every workgroup stores its partial sum, then takes a ticket; the workgroup holding the last ticket adds ALL partials
// in fixed order (the tree of block_sum_partials, whichever workgroup happens to be last) and stores the total.  The
// consumers read one double instead of summing <= 2048 partials in their prologues, so the grid no longer has to be
// capped at 2048 workgroups: the streaming reductions run on the same n/512-workgroup grids as the element-wise
// kernels.  `counter` must be zero when the kernel starts and is zero again when it ends.
// Usage inside a kernel (after every thread that holds a partial has stored it with store_partial):
//   if (last_block_ticket(counters, tickets, my_slot)) { t = sum_partials_agent(p, count, lds); if (threadIdx.x == 0) *total = t; ...;
//                                             release_ticket_counter(counter); }
__device__ __forceinline__ void wait_for_my_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The storing thread itself waits until the store has been performed (vmcnt counts stores on gfx9): a workgroup-scope
// release fence emits NO wait on gfx950 (checked in the ISA: store, s_waitcnt lgkmcnt(0), s_barrier, atomic), so without
// this nothing would order the partial before the ticket that announces it.
__device__ __forceinline__ void store_partial(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  wait_for_my_stores();
}

// True, uniformly over the workgroup, in the workgroup that takes the last of `tickets` tickets (my_slot: this workgroup's
// number among them).  The barrier in front makes the partial stores of ALL threads of this workgroup precede thread 0's.
// Tickets are counted on two levels - workgroup b takes a ticket of group b / 64 and the workgroup that completes a
// group takes one of the master counter - because ~10^4 atomic adds on ONE address serialise (measured: 7 ns each, 70 us
// for the 9766 workgroups of an element-wise kernel at N = 1e7).  `counters`: 1 + ceil(tickets / 64) words, all zero
// before the kernel and after it.
//
// The partials were stored with agent-scope atomic stores (written through to the point where all XCDs agree) and are
// read back with agent-scope atomic loads, so what has to be ordered is only "my store has been performed before my
// ticket is counted": an explicit `s_waitcnt vmcnt(0)` in every storing thread (store_partial) in front of the barrier
// below, and once more in thread 0 in front of the ticket - NOT an agent-scope release, which on this chip writes back
// and invalidates the XCD's whole L2 at the end of every workgroup (measured: +15 us per kernel at N = 1e6), and not a
// workgroup-scope release either, which compiles to no wait at all.
#define HIPEIG_TICKET_GROUP 64
#define HIPEIG_TICKET_WORDS (1 + HIPEIG_WIDE_PARTIALS / HIPEIG_TICKET_GROUP + 7)
__device__ __forceinline__ bool last_block_ticket(unsigned* counters, unsigned tickets, unsigned my_slot) {
  __shared__ int sh_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ng = (tickets + HIPEIG_TICKET_GROUP - 1) / HIPEIG_TICKET_GROUP;
    const unsigned g = my_slot / HIPEIG_TICKET_GROUP;
    const unsigned gsize = (g == ng - 1) ? tickets - g * HIPEIG_TICKET_GROUP : HIPEIG_TICKET_GROUP;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wait_for_my_stores();
    int last = 0;
    if (__hip_atomic_fetch_add(counters + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(counters + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    sh_last = last;
  }
  __syncthreads();
  const bool last = sh_last != 0;
  __syncthreads();                                   // sh_last may be rewritten by the next call
  return last;
}

// Fixed-order sum of p[0..count) read past the (per-XCD, mutually incoherent) caches; every thread obtains it.  Same tree
// as block_sum_partials.
__device__ __forceinline__ double sum_partials_agent(const double* p, int count, double* lds) {
  const int nt = blockDim.x < HIPEIG_BLOCK ? (int)blockDim.x : HIPEIG_BLOCK;
  double a = 0.0;
  if ((int)threadIdx.x < nt)
    for (int i = threadIdx.x; i < count; i += nt) a += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = wave_reduce_sum(a);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) lds[wid] = a;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < (nt >> 6); ++w) r += lds[w];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void release_ticket_counter(unsigned* counter) {
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid of a reduction that finishes in its last workgroup: n/(256*per_thread) workgroups, at most HIPEIG_WIDE_PARTIALS.
static inline int grid_wide(int64_t n, int per_thread) {
  int64_t g = (n + (int64_t)HIPEIG_BLOCK * per_thread - 1) / ((int64_t)HIPEIG_BLOCK * per_thread);
  if (g < 1) g = 1;
  if (g > HIPEIG_WIDE_PARTIALS) g = HIPEIG_WIDE_PARTIALS;
  return (int)g;
}

// Grid of a kernel WITHOUT a reduction (nothing is left per workgroup, so the cap above does not apply): one 16-byte
// access per thread and stream.  Measured on MI355X (tools/stream_forms_bench.hip, y = a*x on 1e8 doubles): the same
// grid-stride kernel runs at 4.9-5.2 TB/s on 2048 workgroups and at 6.0-6.6 TB/s on n/512 workgroups.
static inline int grid_stream(int64_t n) {
  int64_t g = (n + (int64_t)HIPEIG_BLOCK * 2 - 1) / ((int64_t)HIPEIG_BLOCK * 2);
  if (g < 1) g = 1;
  if (g > ((int64_t)1 << 22)) g = (int64_t)1 << 22;
  return (int)g;
}

static inline int grid_for(int64_t n, int per_thread) {
  int64_t g = (n + (int64_t)HIPEIG_BLOCK * per_thread - 1) / ((int64_t)HIPEIG_BLOCK * per_thread);
  if (g < 1) g = 1;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return (int)g;
}
