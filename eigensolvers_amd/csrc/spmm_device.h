// Block product Y = H X for a block of K = 4 or 8 operands held INTERLEAVED ([row][K], 8K bytes per row):
// the tall-skinny SpMM of block Lanczos (inexact_Lanczos.py:319-320 runs the nBlock solves of one
// iteration on the same operator and shift) and of matrixRepresentation (numpyVector.py:184-185).
// One index + one value fetched per non-zero serve K FMAs and the operand gather per non-zero is ONE
// contiguous 8K-byte read.  Blocks of <= 4 operands use K = 4 (twice the accumulator rows per workgroup,
// half the vector traffic), larger ones chunks of K = 8.  Sizes below are for K = 8 (K = 4 in brackets).
//
// "TCOO-B" layout = the column-window blocked storage of spmv_device.h sized for blocks:
//   * a row block (unit) of <= 2520 [5048] rows is owned by one 1024-thread workgroup whose LDS holds
//     the K accumulators of every row (64 [32] B per row);
//   * the columns are cut into windows of 2^wbits operand rows (2^11 [2^12] x 8K B = 128 KiB of X: 32
//     windows fit one XCD's L2, so workgroups that drift apart still hit) and a unit's non-zeros are stored
//     window after window as (row_local << wbits | col_local, value) - 12 bytes per non-zero like CSR;
//   * K/2 lanes serve one non-zero (16 B each of the operand row, two accumulators each), so one wave
//     instruction gathers 16 [32] non-zeros; the (idx, val) stream is loaded one element per lane and
//     handed to the lane groups by ds_bpermute; 8 gather instructions per wave are kept in flight.
// Accumulation is LDS fp64 atomics (measured, tools/l1_forms_bench.hip: the quad pattern costs +0.6 ms
// per 6.5e8 non-zeros over no accumulation at all; read-modify-write without atomics +0.4 ms but would
// need conflict-free instruction groups).
//
// When it pays: the lines of X that an XCD pulls into its L2 for one window are re-used by the 32
// workgroups of that XCD only if their tiles are dense enough:
//   touches per line = 32 * rows_per_unit * nnz_per_row * (rows of X per 128-byte line) / ncols.
// N = 1e6, 33 nnz/row, K = 8: 4.1 (worth it, X = 64 MB); N = 1e7, 65 nnz/row: 1.05 (every gather a miss:
// measured 8.7 ms for the gathers alone) - there hipeig_spmm keeps the row-owner kernel.  Either way every
// gather moves a whole 128-byte line from L2 to L1 for 64 [32] useful bytes: 4.2 GB per product at
// N = 1e6, ~0.17 ms of the 0.36 [0.24] ms a product takes.
#pragma once
#include "common.h"

#define BCOO_KMAX 8                       // widest interleave of the block SOLVES (operands per block, zero padded); K = 4 for blocks of <= 4
#define BCOO_KPACK 16                     // widest interleave of a block PRODUCT: 8 complex operands of a contour point (128 B = one line per operand row)
#define BCOO_THREADS 1024
#define BCOO_GATHERS 8                    // gather instructions a wave keeps in flight (measured: 4 / 8 / 16 -> 0.41 / 0.385 / 0.53 ms)
#define BCOO_MAX_WIN 4096                 // (nwin + 2) offsets share the dynamic LDS with the accumulators
#define HIPEIG_BCOO_LDS_MAX ((size_t)161792)     // dynamic LDS; 2 KiB of the CU's 160 KiB stay for the kernels' static arrays

template <int K> struct BcooShape {
  static_assert(K == 4 || K == 8 || K == 16, "interleave width is 4, 8 or 16");
  static constexpr int LPN = K / 2;               // lanes per non-zero: each lane owns 16 B (two operands) of the operand row
  static constexpr int NPI = 64 / LPN;            // non-zeros per gather instruction
  static constexpr int ROUNDS = 64 / NPI;         // gather instructions per 64 non-zeros
  static constexpr int NB = BCOO_GATHERS / ROUNDS > 0 ? BCOO_GATHERS / ROUNDS : 1; // sub-batches of 64 non-zeros per wave step
  static constexpr int MAX_RW = (int)(HIPEIG_BCOO_LDS_MAX / (K * 8)) - 8;     // accumulator rows per workgroup
};

struct BcooView {
  const uint32_t* __restrict__ idx;
  const double* __restrict__ val;
  const uint32_t* __restrict__ off;      // nunits*nwin + 1 offsets, unit-major
  int32_t nunits, nwin, wbits, rw;
  int32_t unit_begin;
  int64_t nrows;
};

__device__ __forceinline__ void lds_add_f64_blk(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// operand gather of the block sweep (16 B of an operand row).  BCOO_GATHER_NT (build-time experiment): non-temporal load.
#ifndef BCOO_GATHER_NT
#define BCOO_GATHER_NT 0
#endif
typedef double bcoo_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 bcoo_gather(const double2* p) {
#if BCOO_GATHER_NT
  const bcoo_d2 v = __builtin_nontemporal_load(reinterpret_cast<const bcoo_d2*>(p));
  return make_double2(v.x, v.y);
#else
  return *p;
#endif
}

__device__ __forceinline__ uint32_t bperm_u32(int src_lane, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ double bperm_f64(int src_lane, double v) {
  const uint64_t u = __double_as_longlong(v);
  const uint32_t lo = bperm_u32(src_lane, (uint32_t)u), hi = bperm_u32(src_lane, (uint32_t)(u >> 32));
  return __longlong_as_double(((uint64_t)hi << 32) | lo);
}

// Epi must provide: __device__ void elem(int64_t row, int j, double sum, double& acc) const;
// it is called once per (row, operand) with threadIdx.x % K == j, so `acc` is a per-operand partial.
template <int K, class Epi>
__device__ __forceinline__ void bcoo_wg_sweep(const BcooView& T, const double* __restrict__ X, const Epi& epi,
                                              double& acc, double* yacc /* rw*K doubles + (nwin+2) uint32 */) {
  using Sh = BcooShape<K>;
  constexpr int LPN = Sh::LPN, NPI = Sh::NPI, ROUNDS = Sh::ROUNDS, NB = Sh::NB;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int grp = lane / LPN, sub = lane % LPN;
  const uint32_t cmask = (1u << T.wbits) - 1u;
  const int u = T.unit_begin + blockIdx.x;
  if (u >= T.nunits) return;                         // uniform for the workgroup
  uint32_t* offL = reinterpret_cast<uint32_t*>(yacc + (size_t)T.rw * K);
  const int64_t r0 = (int64_t)u * T.rw;
  for (int k = threadIdx.x; k < T.rw * K; k += blockDim.x) yacc[k] = 0.0;
  for (int k = threadIdx.x; k <= T.nwin; k += blockDim.x) offL[k] = T.off[(size_t)u * T.nwin + k];
  __syncthreads();
  const uint32_t sbeg = offL[0], send = offL[T.nwin];
  // A wave takes NB * 64 consecutive non-zeros per step (one (idx, val) pair per lane and sub-batch);
  // the next step's stream loads are issued before this step's ROUNDS * NB gather instructions, which
  // are all issued before the first of them is waited for.
  const uint32_t step = (uint32_t)nw * 64 * NB;
  const double2* __restrict__ X2 = reinterpret_cast<const double2*>(X);
  int c = 0;
  uint32_t idA[NB], idB[NB];
  double vA[NB], vB[NB];
  // Every load is unconditional (clamped index / column 0 for padding lanes) so that the compiler can
  // keep the stream loads and all gathers of a step in flight behind counted s_waitcnt vmcnt(N); with
  // the loads inside exec-masked blocks it drained the queue (vmcnt(0)) after every single gather.
#define BCOO_LOAD(ID, V, BASE)                                             \
  _Pragma("unroll") for (int b = 0; b < NB; ++b) {                         \
    const uint32_t q = (BASE) + 64 * b + lane;                             \
    const uint32_t qc = q < send ? q : send - 1;                           \
    ID[b] = __builtin_nontemporal_load(T.idx + qc);                        \
    V[b] = __builtin_nontemporal_load(T.val + qc);                         \
  }
#define BCOO_CONSUME(ID, V, BASE)                                          \
  {                                                                        \
    uint32_t row_t[ROUNDS * NB];                                           \
    double v_t[ROUNDS * NB];                                               \
    double2 g[ROUNDS * NB];                                                \
    _Pragma("unroll") for (int b = 0; b < NB; ++b) {                       \
      const uint32_t q = (BASE) + 64 * b + lane;                           \
      while (c + 1 < T.nwin && q >= offL[c + 1]) ++c;                      \
      const bool live = q < send;            /* padding lanes re-read the last element: masked here */ \
      const uint32_t col = live ? ((uint32_t)c << T.wbits) + (ID[b] & cmask) : 0u; \
      const uint32_t row = live ? (ID[b] >> T.wbits) : 0xFFFFFFFFu;        \
      _Pragma("unroll") for (int t = 0; t < ROUNDS; ++t) {                 \
        const int src = t * NPI + grp;                                     \
        const uint32_t col_t = bperm_u32(src, col);                        \
        row_t[ROUNDS * b + t] = bperm_u32(src, row);                       \
        v_t[ROUNDS * b + t] = bperm_f64(src, V[b]);                        \
        g[ROUNDS * b + t] = bcoo_gather(X2 + (size_t)col_t * (K / 2) + sub); \
      }                                                                    \
    }                                                                      \
    _Pragma("unroll") for (int t = 0; t < ROUNDS * NB; ++t) {              \
      if (row_t[t] != 0xFFFFFFFFu) {                                       \
        double* a = yacc + (size_t)row_t[t] * K + sub * 2;                 \
        lds_add_f64_blk(a, v_t[t] * g[t].x);                               \
        lds_add_f64_blk(a + 1, v_t[t] * g[t].y);                           \
      }                                                                    \
    }                                                                      \
  }
  uint32_t base = sbeg + (uint32_t)wid * 64 * NB;
  if (base < send) {                                 // uniform per wave (send > sbeg: the clamp above is in range)
    BCOO_LOAD(idA, vA, base)
    while (true) {
      const uint32_t nb = base + step;
      BCOO_LOAD(idB, vB, nb)                         // unconditional: past the end it re-reads the last element
      BCOO_CONSUME(idA, vA, base)
      if (nb >= send) break;
      const uint32_t nb2 = nb + step;
      BCOO_LOAD(idA, vA, nb2)
      BCOO_CONSUME(idB, vB, nb)
      if (nb2 >= send) break;
      base = nb2;
    }
  }
#undef BCOO_LOAD
#undef BCOO_CONSUME
  __syncthreads();
  const int nk = T.rw * K;
  for (int k = threadIdx.x; k < nk; k += blockDim.x) {          // blockDim % K == 0: k % K is fixed per thread
    const int64_t r = r0 + k / K;
    if (r < T.nrows) epi.elem(r, k % K, yacc[k], acc);
  }
}

// Row-owner form (no windows): a wavefront owns one row at a time, lane = (slot, operand), 64 / K non-zeros
// in flight; partial sums over the slots are folded with shuffles.  Every gather is an 8K-byte read from
// wherever X lives (Infinity Cache / HBM) - what remains when the tiles are too sparse for L2 reuse.  The
// terms of a row are added in a fixed order: this kernel is bitwise reproducible.
template <int K, class Epi>
__device__ __forceinline__ void csr_rowowner_block_sweep(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const double* __restrict__ val, int64_t nrows,
                                                         const double* __restrict__ X, const Epi& epi, double& acc) {
  const int lane = threadIdx.x & 63;
  const int r = lane % K, sl = lane / K;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int NPI = 64 / K;                 // non-zeros per wave instruction
#ifndef ROWOWNER_UNROLL
#define ROWOWNER_UNROLL 2                     // gather instructions in flight per wave (a lane adds its terms in ascending order
#endif                                        // whatever the unroll: same bits)
  for (int64_t row = wave; row < nrows; row += nwaves) {
    const int s = rowptr[row], e = rowptr[row + 1];
    double a = 0.0;
    for (int p0 = s + sl; p0 - sl < e; p0 += ROWOWNER_UNROLL * NPI) {
      double v[ROWOWNER_UNROLL], xv[ROWOWNER_UNROLL];
      int cc[ROWOWNER_UNROLL];
      // unconditional loads from clamped positions (e - 1 is a term of this row), masked afterwards: the compiler
      // then keeps all the gathers of a batch in flight instead of draining them one by one (exec-masked loads)
#pragma unroll
      for (int u = 0; u < ROWOWNER_UNROLL; ++u) {
        const int p = p0 + u * NPI;
        const int pc = p < e ? p : e - 1;
        cc[u] = __builtin_nontemporal_load(col + pc);
        v[u] = __builtin_nontemporal_load(val + pc);
      }
#pragma unroll
      for (int u = 0; u < ROWOWNER_UNROLL; ++u) xv[u] = X[(int64_t)cc[u] * K + r];
#pragma unroll
      for (int u = 0; u < ROWOWNER_UNROLL; ++u)
        if (p0 + u * NPI < e) a = fma(v[u], xv[u], a);
    }
#pragma unroll
    for (int off = 32; off >= K; off >>= 1) a += __shfl_down(a, off, 64);
    if (sl == 0) epi.elem(row, r, a, acc);
  }
}

// Sum over the workgroup of a per-operand partial (threadIdx.x % K = operand): out[j], j < K, valid in
// threads 0..K-1 after the call.  lds must hold (blockDim/64)*K doubles.
template <int K>
__device__ __forceinline__ double block_reduce_cols(double v, double* lds) {
#pragma unroll
  for (int off = K; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane < K) lds[wid * K + lane] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < K) {
    r = lds[threadIdx.x];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w * K + threadIdx.x];
  }
  __syncthreads();
  return r;
}

// Every thread obtains the fixed-order sum of the partials of ITS operand (threadIdx.x % K):
// p holds count records of K doubles.  lds must hold (blockDim/64)*K doubles.
template <int K>
__device__ __forceinline__ double block_sum_partials_cols(const double* __restrict__ p, int count, double* lds) {
  double a = 0.0;
  for (int i = threadIdx.x; i < count * K; i += blockDim.x) a += p[i];      // blockDim % K == 0
#pragma unroll
  for (int off = K; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane < K) lds[wid * K + lane] = a;
  __syncthreads();
  const int j = threadIdx.x % K;
  double r = lds[j];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w * K + j];
  __syncthreads();
  return r;
}
