// Block product Y = H X for a block of <= 8 operands held INTERLEAVED ([row][8], 64 bytes per
// row): the tall-skinny SpMM of block Lanczos (inexact_Lanczos.py:319-320 runs the nBlock
// solves of one iteration on the same operator and shift) and of matrixRepresentation
// (numpyVector.py:184-185).  One index + one value fetched per non-zero serve 8 FMAs and
// the operand gather per non-zero is ONE contiguous 64-byte read.
//
// "TCOO-B" layout = the column-window blocked storage of spmv_device.h sized for blocks:
//   * a row block (unit) of <= BCOO_MAX_RW rows is owned by one 1024-thread workgroup whose
//     LDS holds the 8 accumulators of every row (64 B per row);
//   * the columns are cut into windows of 2^wbits columns (2^14 x 64 B = 1 MiB of X, L2
//     resident) and a unit's non-zeros are stored window after window as
//     (row_local << wbits | col_local, value) - 12 bytes per non-zero like CSR;
//   * 4 lanes serve one non-zero (16 B each of the 64-byte operand row, two accumulators
//     each), so one wave instruction gathers 16 non-zeros; the (idx, val) stream is loaded
//     one element per lane, 64 per wave, double-buffered, and handed to the quads by
//     ds_bpermute.
// Accumulation is LDS fp64 atomics (measured, tools/l1_forms_bench.hip: the quad pattern
// costs +0.6 ms per 6.5e8 non-zeros over no accumulation at all; read-modify-write without
// atomics +0.4 ms but would need conflict-free instruction groups).
//
// When it pays: the lines of X that an XCD pulls into its L2 for one window are re-used by
// the 32 workgroups of that XCD only if their tiles are dense enough:
//   touches per line = 32 * rows_per_unit * nnz_per_row * 2 / ncols.
// N = 1e6, 33 nnz/row: 4.1 (worth it, X = 64 MB); N = 1e7, 65 nnz/row: 1.05 (every gather a
// miss: measured 8.7 ms for the gathers alone) - there hipeig_spmm keeps the row-owner kernel.
#pragma once
#include "common.h"

#define BCOO_K 8                          // interleave width (operands per block, zero padded)
#define BCOO_THREADS 1024
#ifndef BCOO_NB
#define BCOO_NB 2                         // sub-batches of 64 non-zeros per wave step: 4 * BCOO_NB gathers in flight per wave
#endif
#define BCOO_MAX_RW 2520                  // 161,280 B of accumulators (+ the unit's window offsets)
#define BCOO_MAX_WIN 4096                 // (nwin + 2) offsets share the dynamic LDS with the accumulators
#define HIPEIG_BCOO_LDS_MAX ((size_t)161792)     // dynamic LDS; 2 KiB of the CU's 160 KiB stay for the kernels' static arrays

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

__device__ __forceinline__ uint32_t bperm_u32(int src_lane, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ double bperm_f64(int src_lane, double v) {
  const uint64_t u = __double_as_longlong(v);
  const uint32_t lo = bperm_u32(src_lane, (uint32_t)u), hi = bperm_u32(src_lane, (uint32_t)(u >> 32));
  return __longlong_as_double(((uint64_t)hi << 32) | lo);
}

// Epi must provide: __device__ void elem(int64_t row, int j, double sum, double& acc) const;
// it is called once per (row, operand) with threadIdx.x % 8 == j, so `acc` is a per-operand partial.
template <class Epi>
__device__ __forceinline__ void bcoo_wg_sweep(const BcooView& T, const double* __restrict__ X, const Epi& epi,
                                              double& acc, double* yacc /* rw*8 doubles + (nwin+2) uint32 */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int quad = lane >> 2, sub = lane & 3;
  const uint32_t cmask = (1u << T.wbits) - 1u;
  const int u = T.unit_begin + blockIdx.x;
  if (u >= T.nunits) return;                         // uniform for the workgroup
  uint32_t* offL = reinterpret_cast<uint32_t*>(yacc + (size_t)T.rw * BCOO_K);
  const int64_t r0 = (int64_t)u * T.rw;
  for (int k = threadIdx.x; k < T.rw * BCOO_K; k += blockDim.x) yacc[k] = 0.0;
  for (int k = threadIdx.x; k <= T.nwin; k += blockDim.x) offL[k] = T.off[(size_t)u * T.nwin + k];
  __syncthreads();
  const uint32_t sbeg = offL[0], send = offL[T.nwin];
  // A wave takes BCOO_NB * 64 consecutive non-zeros per step (one (idx, val) pair per lane and sub-batch);
  // the next step's stream loads are issued before this step's 4 * BCOO_NB gather instructions, which
  // are all issued before the first of them is waited for.
  const uint32_t step = (uint32_t)nw * 64 * BCOO_NB;
  const double2* __restrict__ X2 = reinterpret_cast<const double2*>(X);
  int c = 0;
  uint32_t idA[BCOO_NB], idB[BCOO_NB];
  double vA[BCOO_NB], vB[BCOO_NB];
  // Every load is unconditional (clamped index / column 0 for padding lanes) so that the compiler can
  // keep the stream loads and all gathers of a step in flight behind counted s_waitcnt vmcnt(N); with
  // the loads inside exec-masked blocks it drained the queue (vmcnt(0)) after every single gather.
#define BCOO_LOAD(ID, V, BASE)                                             \
  _Pragma("unroll") for (int b = 0; b < BCOO_NB; ++b) {                    \
    const uint32_t q = (BASE) + 64 * b + lane;                             \
    const uint32_t qc = q < send ? q : send - 1;                           \
    ID[b] = __builtin_nontemporal_load(T.idx + qc);                        \
    V[b] = __builtin_nontemporal_load(T.val + qc);                         \
  }
#define BCOO_CONSUME(ID, V, BASE)                                          \
  {                                                                        \
    uint32_t row_t[4 * BCOO_NB];                                           \
    double v_t[4 * BCOO_NB];                                               \
    double2 g[4 * BCOO_NB];                                                \
    _Pragma("unroll") for (int b = 0; b < BCOO_NB; ++b) {                  \
      const uint32_t q = (BASE) + 64 * b + lane;                           \
      while (c + 1 < T.nwin && q >= offL[c + 1]) ++c;                      \
      const bool live = q < send;            /* padding lanes re-read the last element: masked here */ \
      const uint32_t col = live ? ((uint32_t)c << T.wbits) + (ID[b] & cmask) : 0u; \
      const uint32_t row = live ? (ID[b] >> T.wbits) : 0xFFFFFFFFu;        \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                      \
        const int src = t * 16 + quad;                                     \
        const uint32_t col_t = bperm_u32(src, col);                        \
        row_t[4 * b + t] = bperm_u32(src, row);                            \
        v_t[4 * b + t] = bperm_f64(src, V[b]);                             \
        g[4 * b + t] = X2[(size_t)col_t * (BCOO_K / 2) + sub];             \
      }                                                                    \
    }                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4 * BCOO_NB; ++t) {              \
      if (row_t[t] != 0xFFFFFFFFu) {                                       \
        double* a = yacc + (size_t)row_t[t] * BCOO_K + sub * 2;            \
        lds_add_f64_blk(a, v_t[t] * g[t].x);                               \
        lds_add_f64_blk(a + 1, v_t[t] * g[t].y);                           \
      }                                                                    \
    }                                                                      \
  }
  uint32_t base = sbeg + (uint32_t)wid * 64 * BCOO_NB;
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
  const int nk = T.rw * BCOO_K;
  for (int k = threadIdx.x; k < nk; k += blockDim.x) {          // blockDim % 8 == 0: k % 8 is fixed per thread
    const int64_t r = r0 + (k >> 3);
    if (r < T.nrows) epi.elem(r, k & 7, yacc[k], acc);
  }
}

// Row-owner form (no windows): a wavefront owns one row at a time, lane = (slot, operand), 8 non-zeros
// in flight; partial sums over the slots are folded with shuffles.  Every gather is a 64-byte read
// from wherever X lives (Infinity Cache / HBM) - what remains when the tiles are too sparse for L2 reuse.
template <class Epi>
__device__ __forceinline__ void csr_rowowner_block_sweep(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const double* __restrict__ val, int64_t nrows,
                                                         const double* __restrict__ X, const Epi& epi, double& acc) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 7, sl = lane >> 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave; row < nrows; row += nwaves) {
    const int s = rowptr[row], e = rowptr[row + 1];
    double a = 0.0;
    for (int p = s + sl; p < e; p += 8)
      a = fma(__builtin_nontemporal_load(val + p), X[(int64_t)__builtin_nontemporal_load(col + p) * BCOO_K + r], a);
    a += __shfl_down(a, 32, 64);
    a += __shfl_down(a, 16, 64);
    a += __shfl_down(a, 8, 64);
    if (sl == 0) epi.elem(row, r, a, acc);
  }
}

// Sum over the workgroup of a per-operand partial (threadIdx.x % 8 = operand): out[j], j < 8, valid in
// threads 0..7 after the call.  lds must hold (blockDim/64)*8 doubles.
__device__ __forceinline__ double block_reduce_cols8(double v, double* lds) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane < 8) lds[wid * 8 + lane] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < 8) {
    r = lds[threadIdx.x];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w * 8 + threadIdx.x];
  }
  __syncthreads();
  return r;
}

// Every thread obtains the fixed-order sum of the partials of ITS operand (threadIdx.x % 8):
// p holds count records of 8 doubles.  lds must hold (blockDim/64)*8 doubles.
__device__ __forceinline__ double block_sum_partials_cols8(const double* __restrict__ p, int count, double* lds) {
  double a = 0.0;
  for (int i = threadIdx.x; i < count * 8; i += blockDim.x) a += p[i];      // blockDim % 8 == 0
  a += __shfl_xor(a, 8, 64);
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane < 8) lds[wid * 8 + lane] = a;
  __syncthreads();
  const int j = threadIdx.x & 7;
  double r = lds[j];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w * 8 + j];
  __syncthreads();
  return r;
}
