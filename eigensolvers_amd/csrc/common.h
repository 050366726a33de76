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
  // copy of the same layout for the pair sweep (two accumulators per row: units of half the rows), built
  // on the first hipeig_spmv_shift_pair of a large operator
  uint32_t* p_idx;
  double* p_val;
  uint32_t* p_off;
  int32_t p_nunits, p_nwin, p_wbits, p_rw, p_wgs_per_sweep;
  int32_t last_pair_fused;   // 1 when the most recent pair product ran as one sweep
  int reproducible;          // automatic choice restricted to bitwise reproducible kernels (hipeig_csr_set_reproducible)
  double absrow_max;         // max_i sum_j |a_ij| over the local rows: overflow bound of the fixed-point sweep (variant 5)
  // block-operand copies ("TCOO-B", spmm_device.h): one per interleave width, [0]: K = 4, [1]: K = 8; built on first use
  struct BcooLayout {
    uint32_t* idx;
    double* val;
    uint32_t* off;
    int32_t nunits, nwin, wbits, rw, wgs_per_sweep;
    int32_t state;           // 0 = undecided, 1 = built, 2 = not suited (row-owner kernel is used)
  } bl[2];
  int32_t block_variant;     // 0 = automatic, 1 = row-owner CSR, 2 = TCOO-B
  int32_t last_block_variant, last_block_k;
  int64_t gather_len;        // length of the gathered operand (ncols, or stride*nranks)
  int variant;               // 0 = auto, 1 = CSR-vector, 2 = CSR-stream, 3 = TCOO (wave units), 4 = TCOO-W, 5 = TCOO-W with fixed-point accumulators
  int last_variant;          // variant used by the most recent launch (0 = none yet)
  int last_launches;         // kernel launches (sweeps) one product with that variant takes
  int lanes_per_row;         // sub-wave width used to reduce one row
  int64_t col_stride;        // x_full stride per rank when columns were remapped (0 = global)
  int64_t bytes;
};

// ---- collectives (comm.hip); no-ops without a communicator ---------------------------
int hipeig_comm_setup_rows(hipeig_ctx* ctx, int64_t nrows_local, int64_t* stride_out);
int hipeig_allreduce_sum(hipeig_ctx* ctx, double* d_buf, int count);
int hipeig_allgather_x(hipeig_ctx* ctx, const double* x_local, int64_t n_local, int64_t stride,
                       const double** x_full_out);
int hipeig_allgather_f64(hipeig_ctx* ctx, const double* send, double* recv, size_t count);
int hipeig_allgather_x_begin(hipeig_ctx* ctx, const double* x_local, int64_t n_local, int64_t stride);
int hipeig_allgather_x_end(hipeig_ctx* ctx, const double** x_full_out);

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
