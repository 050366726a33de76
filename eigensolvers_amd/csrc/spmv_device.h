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

// ---- column-window blocked layout ("TCOO") ----------------------------------------------
// Random columns make every x gather a separate L2 request, and beyond the 4 MiB L2 of an
// XCD each one costs a full 128-byte fabric fetch (measured: 57 Ggather/s from an 80 MB
// table against 215 Ggather/s from an L2-resident one, tools/gather_bench*.hip).  This
// layout makes the gathers L2-resident: the columns are cut into windows of W = 2^wbits
// (<= 256 Ki columns = 2 MiB of x), the rows into units of RW rows owned by ONE wavefront,
// and the non-zeros of a unit are stored window by window, each as a packed 32-bit
// (row_local << wbits | col_local) plus the fp64 value - 12 bytes per non-zero like CSR.
// A wave keeps its unit's RW partial sums in LDS, sweeps the windows in order (all waves do,
// so the chip works on one x window at a time) and scatter-adds v*x into LDS with ds_add_f64.
// No inter-wave communication at all; adds to one row come from one wave in stream order,
// so the summation order is fixed and results are reproducible.
#define TCOO_MAX_WBITS 18
#define TCOO_MAX_RW 2560                 // 20 KiB of LDS per wave, 8 waves per CU
#define TCOO_UNROLL 8

struct TcooView {
  const uint32_t* __restrict__ idx;
  const double* __restrict__ val;
  const uint32_t* __restrict__ off;      // nunits*nwin + 1 offsets, unit-major
  int32_t nunits, nwin, wbits, rw;
  int64_t nrows;
};

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <class Epi>
__device__ __forceinline__ void tcoo_sweep(const TcooView& T, const double* __restrict__ x, const Epi& epi,
                                           double& acc, double* lds /* blockDim/64 * rw doubles */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  double* yacc = lds + (size_t)wid * T.rw;
  const uint32_t cmask = (1u << T.wbits) - 1u;
  const int nwaves = gridDim.x * waves_per_block;
  for (int u = blockIdx.x * waves_per_block + wid; u < T.nunits; u += nwaves) {
    for (int k = lane; k < T.rw; k += 64) yacc[k] = 0.0;
    const uint32_t* offu = T.off + (size_t)u * T.nwin;
    for (int c = 0; c < T.nwin; ++c) {
      const uint32_t beg = offu[c], end = offu[c + 1];
      const double* __restrict__ xw = x + ((size_t)c << T.wbits);
      uint32_t p = beg + lane;
      for (; p + 64 * (TCOO_UNROLL - 1) < end; p += 64 * TCOO_UNROLL) {
        uint32_t id[TCOO_UNROLL];
        double v[TCOO_UNROLL];
#pragma unroll
        for (int j = 0; j < TCOO_UNROLL; ++j) {
          id[j] = __builtin_nontemporal_load(T.idx + p + 64 * j);
          v[j] = __builtin_nontemporal_load(T.val + p + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < TCOO_UNROLL; ++j) v[j] *= xw[id[j] & cmask];
#pragma unroll
        for (int j = 0; j < TCOO_UNROLL; ++j) lds_add_f64(yacc + (id[j] >> T.wbits), v[j]);
      }
      for (; p < end; p += 64) {
        const uint32_t id = __builtin_nontemporal_load(T.idx + p);
        const double v = __builtin_nontemporal_load(T.val + p);
        lds_add_f64(yacc + (id >> T.wbits), v * xw[id & cmask]);
      }
    }
    const int64_t r0 = (int64_t)u * T.rw;
    for (int k = lane; k < T.rw && r0 + k < T.nrows; k += 64) epi.row(r0 + k, yacc[k], acc);
  }
}
