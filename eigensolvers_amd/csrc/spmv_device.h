// Device-side CSR row products for gfx950, shared by spmv.hip and minres.hip.
//
// Two kernel bodies, both parameterised by an epilogue functor so that the fused variants
// (shift term, MINRES vector update, dot partial) reuse the same sweep:
//
//  * csr_stream_sweep  - "LDS-staged": a workgroup owns a block of consecutive rows whose
//    non-zeros (<= SPMV_NNZ_PER_BLOCK) form ONE contiguous range of val/col.  All 256 lanes
//    stream that range with fully coalesced non-temporal loads, gather x, and park the
//    products in LDS; then sub-wave groups reduce each row's LDS segment and finish with a
//    wave shuffle.  Every lane is busy whatever the row lengths are.
//  * csr_vector_sweep  - one sub-wave group per row straight from global memory (kept as
//    the baseline / ablation variant).
//
// Workgroups are persistent (grid <= HIPEIG_MAX_PARTIALS) and walk row blocks
// b = blockIdx.x, blockIdx.x + gridDim.x, ... so that the resident workgroups always work
// on one contiguous band of the matrix and a fused reduction yields <= 2048 partials.
#pragma once
#include "common.h"

#define SPMV_NNZ_PER_BLOCK 2048          // 16 KiB of LDS products -> 8 workgroups per CU
#define SPMV_PER_THREAD (SPMV_NNZ_PER_BLOCK / HIPEIG_BLOCK)

struct CsrView {
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ col;
  const double* __restrict__ val;
  const int32_t* __restrict__ row_blocks;
  int32_t n_row_blocks;
  int64_t nrows;
  int group;                       // lanes that cooperate on one row (power of two, <= 64)
};

__device__ __forceinline__ double group_reduce_sum(double v, int group) {
  for (int off = group >> 1; off > 0; off >>= 1) v += __shfl_down(v, off, group);
  return v;   // valid in lane 0 of the group
}

// Epi must provide:  __device__ void row(int64_t r, double sum, double& acc) const;
template <class Epi>
__device__ __forceinline__ void csr_stream_sweep(const CsrView& A, const double* __restrict__ x,
                                                 const Epi& epi, double& acc, double* prod /*LDS*/) {
  const int tid = threadIdx.x;
  const int group = A.group;
  const int gid = tid / group, glane = tid % group;
  const int ngroups = HIPEIG_BLOCK / group;
  for (int b = blockIdx.x; b < A.n_row_blocks; b += gridDim.x) {
    const int r0 = A.row_blocks[b], r1 = A.row_blocks[b + 1];
    const int p0 = A.rowptr[r0], p1 = A.rowptr[r1];
    const int nn = p1 - p0;
    if (nn <= SPMV_NNZ_PER_BLOCK) {
      // phase 1: stream val/col (read once -> non-temporal), gather x, park products in LDS
      int cidx[SPMV_PER_THREAD];
      double v[SPMV_PER_THREAD];
#pragma unroll
      for (int u = 0; u < SPMV_PER_THREAD; ++u) {
        const int k = tid + u * HIPEIG_BLOCK;
        if (k < nn) {
          cidx[u] = __builtin_nontemporal_load(A.col + p0 + k);
          v[u] = __builtin_nontemporal_load(A.val + p0 + k);
        }
      }
#pragma unroll
      for (int u = 0; u < SPMV_PER_THREAD; ++u) {
        const int k = tid + u * HIPEIG_BLOCK;
        if (k < nn) v[u] *= x[cidx[u]];
      }
#pragma unroll
      for (int u = 0; u < SPMV_PER_THREAD; ++u) {
        const int k = tid + u * HIPEIG_BLOCK;
        if (k < nn) prod[k] = v[u];
      }
      __syncthreads();
      // phase 2: one sub-wave group per row reduces its LDS segment
      for (int r = r0 + gid; r < r1; r += ngroups) {
        const int s = A.rowptr[r] - p0, e = A.rowptr[r + 1] - p0;
        double sum = 0.0;
        for (int k = s + glane; k < e; k += group) sum += prod[k];
        sum = group_reduce_sum(sum, group);
        if (glane == 0) epi.row(r, sum, acc);
      }
      __syncthreads();
    } else {
      // a single row longer than the LDS tile: the whole workgroup strides over it
      double sum = 0.0;
      for (int p = p0 + tid; p < p1; p += HIPEIG_BLOCK)
        sum = fma(__builtin_nontemporal_load(A.val + p), x[__builtin_nontemporal_load(A.col + p)], sum);
      sum = block_reduce_sum(sum, prod);
      if (tid == 0) epi.row(r0, sum, acc);
    }
  }
}

template <class Epi>
__device__ __forceinline__ void csr_vector_sweep(const CsrView& A, const double* __restrict__ x,
                                                 const Epi& epi, double& acc) {
  const int group = A.group;
  const int64_t gid = ((int64_t)blockIdx.x * HIPEIG_BLOCK + threadIdx.x) / group;
  const int glane = threadIdx.x % group;
  const int64_t ngroups = (int64_t)gridDim.x * HIPEIG_BLOCK / group;
  for (int64_t r = gid; r < A.nrows; r += ngroups) {
    const int s = A.rowptr[r], e = A.rowptr[r + 1];
    double sum = 0.0;
    for (int p = s + glane; p < e; p += group)
      sum = fma(__builtin_nontemporal_load(A.val + p), x[__builtin_nontemporal_load(A.col + p)], sum);
    sum = group_reduce_sum(sum, group);
    if (glane == 0) epi.row(r, sum, acc);
  }
}
