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
#define TCOO_MAX_WIN 128
#define HIPEIG_TCOO_LDS_MAX ((size_t)4 * TCOO_MAX_RW * sizeof(double))
#ifndef TCOO_UNROLL
#define TCOO_UNROLL 4
#endif
#ifndef TCOOW_INTERLEAVE
#define TCOOW_INTERLEAVE 0   // 1: the waves of a workgroup take adjacent 64-element groups (instruction-level interleave)
#endif

// Timing experiments (skip gathers / LDS adds / the value stream / window switching) are compiled
// in only with -DHIPEIG_EXPERIMENTS (make EXPERIMENTS=1); the shipped kernels carry none of it.
#ifdef HIPEIG_EXPERIMENTS
#define TCOO_ABL(T, bit) ((T).ablate & (bit))
#else
#define TCOO_ABL(T, bit) 0
#endif

struct TcooView {
  const uint32_t* __restrict__ idx;
  const double* __restrict__ val;
  const uint32_t* __restrict__ off;      // nunits*nwin + 1 offsets, unit-major
  int32_t nunits, nwin, wbits, rw;
  int32_t unit_begin;                    // first unit of this launch (one unit per wave)
  int32_t prefetch;                      // dense L2 prefetch of the next x window (0/1)
  int32_t ablate;                        // timing experiments only: 1 = skip gathers, 2 = skip LDS adds,
                                         // 4 = skip the value stream, 8 = gather from window 0 only
  int64_t nrows, gather_len;
  // split sweeps (multi-GPU overlap): process only the windows of `nrun` ascending, disjoint runs
  // [run_lo[i], run_hi[i]) (nrun = 0: all windows); start the accumulators from yinit instead of zero;
  // store raw sums to raw_out instead of running the epilogue.  Defaults: all windows, 0, no.
  int32_t nrun;
  int32_t run_lo[4], run_hi[4];
  const double* yinit;
  double* raw_out;
  // column splits (TCOO-W only): when an operator (slab) has fewer row blocks than the GPU has CUs,
  // `csplit` workgroups share one row block, each taking an equal share of the block's non-zero
  // stream (a contiguous run of column windows), so that every CU works but x is streamed through
  // the L1s once per ROW BLOCK instead of once per CU.  Workgroup (block b, share s) stores raw sums
  // to raw_out + (part_base + s) * part_stride; a combine kernel adds the slabs and runs the epilogue.
  int32_t csplit, part_base;
  int64_t part_stride;
  // fixed-point accumulation (TCOO-W, public variant 5): the accumulators are int64 sums of
  // rint(v*x*2^e); e is chosen in the kernel prologue from fx_bound = max_i sum_j |a_ij| (set when the
  // layout is built) and max|x| (fx_count per-workgroup maxima left by absmax_kernel in fx_xmax), so that no
  // row can overflow 2^62.  Integer adds commute: the result does not depend on the order in which waves
  // reach a row nor on the slot order of the layout build - bitwise reproducible, at the speed of the
  // atomic-fp64 form.  Absolute error per row <= (nnz_row/2 + 1) * 2^-61 * fx_bound * max|x|.
  const double* fx_xmax;
  int32_t fx_count;
  double fx_bound;
};

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Dense read of this wave's share of an x window (16 B per lane, whole 128-byte lines) so
// that the window is already in the XCD's L2 when the gathers of the next tile arrive; without
// it every window switch starts with a storm of cold 8-byte misses, each a full line fetch.
// Workgroups b and b+8 are observed to share an XCD, so wave (b/8)*wpb + wid of every XCD
// takes slice number that; a different placement only costs speed, never correctness.
__device__ __forceinline__ double tcoo_prefetch_window(const double* __restrict__ x, int64_t gather_len,
                                                       int c, int wbits, int lane, int wid, int wpb) {
  const int64_t w0 = (int64_t)c << wbits;
  int64_t wlen = (int64_t)1 << wbits;
  if (w0 + wlen > gather_len) wlen = gather_len - w0;
  if (wlen <= 0) return 0.0;
  const int nslices = ((gridDim.x + 7) / 8) * wpb;
  const int rank = (blockIdx.x / 8) * wpb + wid;
  int64_t per = ((wlen + nslices - 1) / nslices + 15) & ~(int64_t)15;     // whole 128-byte lines
  const int64_t s0 = (int64_t)rank * per;
  int64_t s1 = s0 + per;
  if (s1 > wlen) s1 = wlen;
  double pf = 0.0;
  const double* base = x + w0;
  for (int64_t i = s0 + 2 * lane; i + 1 < s1; i += 128) {
    const double2 t = *reinterpret_cast<const double2*>(base + i);
    pf += t.x + t.y;
  }
  return pf;
}

template <class Epi>
__device__ __forceinline__ void tcoo_sweep(const TcooView& T, const double* __restrict__ x, const Epi& epi,
                                           double& acc, double* lds /* blockDim/64 * rw doubles */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  double* yacc = lds + (size_t)wid * T.rw;
  const uint32_t cmask = (1u << T.wbits) - 1u;
  // One launch = one sweep: every resident wave owns exactly one unit and all of them walk
  // the windows together; the kernel boundary keeps consecutive sweeps from overlapping
  // (two sweeps in flight would keep two x windows alive and thrash the L2).
  const int u = T.unit_begin + blockIdx.x * waves_per_block + wid;
  double pf = 0.0;
  if (T.prefetch) pf += tcoo_prefetch_window(x, T.gather_len, 0, T.wbits, lane, wid, waves_per_block);
  if (u < T.nunits) {
    for (int k = lane; k < T.rw; k += 64) yacc[k] = 0.0;
    const uint32_t* offu = T.off + (size_t)u * T.nwin;
    uint32_t end = offu[0];
    for (int c = 0; c < T.nwin; ++c) {
      const uint32_t beg = end;
      end = offu[c + 1];
      if (T.prefetch && c + 1 < T.nwin)
        pf += tcoo_prefetch_window(x, T.gather_len, c + 1, T.wbits, lane, wid, waves_per_block);
      const double* __restrict__ xw = x + ((size_t)c << T.wbits);
      // batches of 64*TCOO_UNROLL non-zeros, every slot predicated: no serial tail
      for (uint32_t base = beg; base < end; base += 64 * TCOO_UNROLL) {
        uint32_t id[TCOO_UNROLL];
        double v[TCOO_UNROLL];
#pragma unroll
        for (int j = 0; j < TCOO_UNROLL; ++j) {
          const uint32_t q = base + lane + 64 * j;
          const bool ok = q < end;
          id[j] = ok ? __builtin_nontemporal_load(T.idx + q) : 0xFFFFFFFFu;
          v[j] = ok ? __builtin_nontemporal_load(T.val + q) : 0.0;
        }
        if (!(TCOO_ABL(T, 1))) {
#pragma unroll
          for (int j = 0; j < TCOO_UNROLL; ++j)
            if (id[j] != 0xFFFFFFFFu) v[j] *= xw[id[j] & cmask];
        }
        if (!(TCOO_ABL(T, 2))) {
#pragma unroll
          for (int j = 0; j < TCOO_UNROLL; ++j)
            if (id[j] != 0xFFFFFFFFu) lds_add_f64(yacc + (id[j] >> T.wbits), v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < TCOO_UNROLL; ++j) pf += v[j] + (double)id[j];
        }
      }
    }
    const int64_t r0 = (int64_t)u * T.rw;
    for (int k = lane; k < T.rw && r0 + k < T.nrows; k += 64) epi.row(r0 + k, yacc[k], acc);
  }
  asm volatile("" ::"v"(pf));      // keep the prefetch loads alive
}

// ---- workgroup-wide units ("TCOO-W") ------------------------------------------------------
// Same storage, but a unit (row block) is owned by a whole 1024-thread workgroup (one per CU, all
// 160 KiB of LDS as the y accumulator: up to 20224 rows) and the non-zeros of a (row block, window)
// tile are bucketed by 32-column bins before being laid out, so lanes that need the same 128-byte
// line of x sit next to each other in ONE gather instruction - the only place where the 256-line
// L1 can serve the second access (measured: a line does not survive until the wave's next gather).
// Lines fetched per non-zero drop from 1 to about (1 - exp(-k))/k with k = (rows per block) *
// (nnz per row) * 16 / ncols (k ~ 2.1 at N = 1e7, 65 nnz/row).  What bounds the sweep is the L1
// pipeline's rate for this hit/miss mix, not L2 or HBM.  Adds to one row
// come from several waves, so the summation ORDER inside a row is not fixed run to run (values
// agree to rounding); the wave-owned TCOO variant above is the reproducible one.
#ifndef TCOOW_THREADS
#define TCOOW_THREADS 1024
#endif
#define TCOOW_MAX_RW 20224               // 161,792 B of LDS (+ the unit's window offsets)
#define TCOOW_MAX_WIN 256
#define HIPEIG_TCOOW_LDS_MAX ((size_t)TCOOW_MAX_RW * sizeof(double) + (TCOOW_MAX_WIN + 2) * sizeof(uint32_t))

__device__ __forceinline__ void lds_add_f64_wg(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// x gather of the TCOO-W sweep.  TCOOW_GATHER_MODE (build-time experiment): 0 plain load,
// 1 non-temporal, 2 sc1 (bypasses the CU's L1), 3 sc0 sc1.
#ifndef TCOOW_GATHER_MODE
#define TCOOW_GATHER_MODE 0
#endif
#ifndef TCOOW_GIF
#define TCOOW_GIF 1          // gather instructions of a batch in flight per wave (1: each waited for before the next)
#endif
__device__ __forceinline__ double tcoow_gather(const double* p) {
#if TCOOW_GATHER_MODE == 1
  return __builtin_nontemporal_load(p);
#elif TCOOW_GATHER_MODE == 2
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load ... sc1
#elif TCOOW_GATHER_MODE == 3
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // global_load ... sc0 sc1
#else
  return *p;
#endif
}

// Sum of the raw partial slabs of a split sweep (fixed order) followed by the row epilogue.
template <class Epi>
__device__ __forceinline__ void tcoow_combine_sweep(const double* __restrict__ parts, int nparts, int64_t stride,
                                                    int64_t nrows, const Epi& epi, double& acc) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += step) {
    double s = parts[r];
    for (int p = 1; p < nparts; ++p) s += parts[(int64_t)p * stride + r];
    epi.row(r, s, acc);
  }
}

// max over the workgroup of per-thread values (exact, order-free); every thread obtains it.  lds: 16 doubles.
__device__ __forceinline__ double block_max_all(double v, double* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, lds[w]);
  __syncthreads();
  return r;
}

// 2^e with bound * xmax * 2^e < 2^62 (one bit of headroom below int64 for the rounding of the terms).
// A non-finite operand gives NaN, which the epilogue's multiplication by 1/scale spreads over the result.
__device__ __forceinline__ double fixed_point_scale(double bound, double xmax) {
  const double v = bound * xmax;
  if (!(v > 0.0)) return (v == 0.0) ? 1.0 : __builtin_nan("");
  if (!(v < 1.7976931348623157e308)) return __builtin_nan("");
  int ex;
  frexp(v, &ex);                                   // v < 2^ex
  int e = 61 - ex;
  e = e > 1000 ? 1000 : (e < -1000 ? -1000 : e);
  return ldexp(1.0, e);
}

__device__ __forceinline__ void lds_add_i64_wg(double* slot, double scaled) {
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double2ll_rn(scaled),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// PAIR = 1: two operands at once (the real and imaginary halves of a complex vector, FEAST's contour solves):
// x is the interleaved operand [column][2], a gather is ONE 16-byte read, every row has two accumulators
// (yacc[2*row], yacc[2*row + 1]: T.rw <= TCOOW_MAX_RW / 2) and the epilogue is `row2(r, sum0, sum1, acc)`.
// The (index, value) stream is read once for both products.  Plain sweeps only (no split / overlap / fixed point).
template <class Epi, int FIXED = 0, int PAIR = 0>
__device__ __forceinline__ void tcoo_wg_sweep(const TcooView& T, const double* __restrict__ x, const Epi& epi,
                                              double& acc, double* yacc /* rw doubles + (nwin+1) uint32 of LDS */,
                                              double* red16 = nullptr /* FIXED: 16 doubles of LDS */) {
  static_assert(!(FIXED && PAIR), "the pair sweep has no fixed-point form");
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const uint32_t cmask = (1u << T.wbits) - 1u;
  const int vb = T.unit_begin + blockIdx.x;
  const int u = (T.csplit > 1) ? vb / T.csplit : vb;
  const int cs = (T.csplit > 1) ? vb - u * T.csplit : 0;
  if (u >= T.nunits) return;                         // uniform for the workgroup
  uint32_t* offL = reinterpret_cast<uint32_t*>(yacc + (PAIR ? 2 : 1) * T.rw);     // this unit's window offsets
  const int64_t r0 = (int64_t)u * T.rw;
  double fx_scale = 1.0, fx_inv = 1.0;
  if (FIXED) {
    double m = 0.0;
    for (int k = threadIdx.x; k < T.fx_count; k += blockDim.x) m = fmax(m, T.fx_xmax[k]);
    m = block_max_all(m, red16);
    fx_scale = fixed_point_scale(T.fx_bound, m);
    fx_inv = 1.0 / fx_scale;                         // exact: a power of two
  }
  if (PAIR) {
    for (int k = threadIdx.x; k < 2 * T.rw; k += blockDim.x) yacc[k] = 0.0;
  } else {
    for (int k = threadIdx.x; k < T.rw; k += blockDim.x) {
      const double y0 = (T.yinit && r0 + k < T.nrows) ? T.yinit[r0 + k] : 0.0;
      if (FIXED) reinterpret_cast<long long*>(yacc)[k] = __double2ll_rn(y0 * fx_scale);
      else yacc[k] = y0;
    }
  }
  for (int k = threadIdx.x; k <= T.nwin; k += blockDim.x) offL[k] = T.off[(size_t)u * T.nwin + k];
  __syncthreads();
  // The unit's non-zeros are ONE contiguous stream (window after window).  Waves take
  // batches of 64*TCOO_UNROLL round-robin; each lane tracks the window of its elements by
  // walking the offsets (they only move forward), so a batch may straddle windows and the
  // stream loads of batch k+1 are in flight while batch k gathers and scatters.
  const uint32_t step = (uint32_t)nw * 64 * TCOO_UNROLL;
#if TCOOW_INTERLEAVE
  // element j of a batch: the 16 waves cover adjacent 64-element groups at every j, so the
  // whole CU works on one narrow column range at a time (L1 reuse across waves)
  const uint32_t jstride = (uint32_t)nw * 64;
  const uint32_t wave_off = (uint32_t)wid * 64;
#else
  const uint32_t jstride = 64;
  const uint32_t wave_off = (uint32_t)wid * 64 * TCOO_UNROLL;
#endif
  int c = 0;
  double sink = 0.0;
  uint32_t idA[TCOO_UNROLL], idB[TCOO_UNROLL];
  double vA[TCOO_UNROLL], vB[TCOO_UNROLL];
#define TCOO_LOAD(ID, V, BASE)                                                         \
  _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) {                            \
    const uint32_t q = (BASE) + lane + jstride * j;                                         \
    const bool ok = q < send;                                                          \
    /* experiment bit 16: the stream re-reads the unit's first 4096 elements (L2-resident stream: what a perfect */ \
    /* prefetch of the stream into L2 could buy; wrong results)                                                  */ \
    const uint32_t ql = (TCOO_ABL(T, 16)) ? sbeg + ((q - sbeg) & 4095u) : q;           \
    ID[j] = ok ? __builtin_nontemporal_load(T.idx + ql) : 0xFFFFFFFFu;                 \
    V[j] = (ok && !(TCOO_ABL(T, 4))) ? __builtin_nontemporal_load(T.val + ql) : 1.0;    \
  }
#define TCOO_CONSUME(ID, V, BASE)                                                      \
  {                                                                                    \
    int cw[TCOO_UNROLL];                                                               \
    _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) {                          \
      const uint32_t q = (BASE) + lane + jstride * j;                                       \
      /* padding lanes (q >= send) must not move the cursor: it would run into the windows skipped  */ \
      /* between two runs and the next run's first elements would gather from there                */ \
      while (q < send && c + 1 < T.nwin && q >= offL[c + 1]) ++c;                      \
      cw[j] = c;                                                                       \
    }                                                                                  \
    double V2[TCOO_UNROLL];                                                            \
    if (TCOOW_GIF > 1 && !PAIR) {                                                      \
      /* build-time experiment: all gathers of the batch issued before the first is waited for (clamped index */ \
      /* for padding lanes, whose products are never added)                                                  */ \
      double xg[TCOO_UNROLL];                                                          \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) {                        \
        const size_t gi = (ID[j] != 0xFFFFFFFFu) ? ((size_t)cw[j] << T.wbits) + (ID[j] & cmask) : 0; \
        xg[j] = tcoow_gather(x + gi);                                                  \
      }                                                                                \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) V[j] *= xg[j];           \
    } else if (!(TCOO_ABL(T, 1))) {                                                      \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j)                          \
        if (ID[j] != 0xFFFFFFFFu) {                                                    \
          const size_t gi = ((size_t)((TCOO_ABL(T, 8)) ? 0 : cw[j]) << T.wbits) + (ID[j] & cmask); \
          if (PAIR) {                                                                  \
            const double2 xv = *reinterpret_cast<const double2*>(x + 2 * gi);          \
            V2[j] = V[j] * xv.y;                                                       \
            V[j] *= xv.x;                                                              \
          } else {                                                                     \
            V[j] *= tcoow_gather(x + gi);                                              \
          }                                                                            \
        }                                                                              \
    } else if (PAIR) {                                                                 \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) V2[j] = V[j];            \
    }                                                                                  \
    if (!(TCOO_ABL(T, 2))) {                                                             \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j)                          \
        if (ID[j] != 0xFFFFFFFFu) {                                                    \
          if (PAIR) {                                                                  \
            lds_add_f64_wg(yacc + 2 * (ID[j] >> T.wbits), V[j]);                       \
            lds_add_f64_wg(yacc + 2 * (ID[j] >> T.wbits) + 1, V2[j]);                  \
          } else if (FIXED) lds_add_i64_wg(yacc + (ID[j] >> T.wbits), V[j] * fx_scale); \
          else lds_add_f64_wg(yacc + (ID[j] >> T.wbits), V[j]);                        \
        }                                                                              \
    } else {                                                                           \
      _Pragma("unroll") for (int j = 0; j < TCOO_UNROLL; ++j) sink += V[j] + (double)ID[j]; \
    }                                                                                  \
  }
  // stream ranges of this launch: all windows, or the windows of the runs
  const int nparts = T.nrun ? T.nrun : 1;
  for (int part = 0; part < nparts; ++part) {
    uint32_t sbeg = T.nrun ? offL[T.run_lo[part]] : offL[0];
    uint32_t send = T.nrun ? offL[T.run_hi[part]] : offL[T.nwin];
    if (T.csplit > 1) {                              // this workgroup's share of the range (64-element granules)
      const uint64_t len = send - sbeg;
      const uint32_t b0 = sbeg + (uint32_t)((len * (uint64_t)cs / (uint64_t)T.csplit) & ~(uint64_t)63);
      const uint32_t b1 = (cs + 1 == T.csplit) ? send : sbeg + (uint32_t)((len * (uint64_t)(cs + 1) / (uint64_t)T.csplit) & ~(uint64_t)63);
      sbeg = b0; send = b1;
    }
    uint32_t base = sbeg + wave_off;
    if (T.nrun && c < T.run_lo[part]) c = T.run_lo[part];      // a run starts behind the windows skipped in front of it
    if (base < send) {
      TCOO_LOAD(idA, vA, base)
      while (true) {
        const uint32_t nb = base + step;
        const bool more = nb < send;                 // uniform per wave
        if (more) { TCOO_LOAD(idB, vB, nb) }
        TCOO_CONSUME(idA, vA, base)
        if (!more) break;
        const uint32_t nb2 = nb + step;
        const bool more2 = nb2 < send;
        if (more2) { TCOO_LOAD(idA, vA, nb2) }
        TCOO_CONSUME(idB, vB, nb)
        if (!more2) break;
        base = nb2;
      }
    }
  }
#undef TCOO_LOAD
#undef TCOO_CONSUME
  asm volatile("" ::"v"(sink));
  __syncthreads();
  if constexpr (PAIR) {
    for (int k = threadIdx.x; k < T.rw && r0 + k < T.nrows; k += blockDim.x)
      epi.row2(r0 + k, yacc[2 * k], yacc[2 * k + 1], acc);
  } else {
    if (T.raw_out) {
      double* dst = T.raw_out + (int64_t)(T.part_base + cs) * T.part_stride;
      for (int k = threadIdx.x; k < T.rw && r0 + k < T.nrows; k += blockDim.x)
        dst[r0 + k] = FIXED ? (double)reinterpret_cast<const long long*>(yacc)[k] * fx_inv : yacc[k];
    } else {
      for (int k = threadIdx.x; k < T.rw && r0 + k < T.nrows; k += blockDim.x)
        epi.row(r0 + k, FIXED ? (double)reinterpret_cast<const long long*>(yacc)[k] * fx_inv : yacc[k], acc);
    }
  }
}

// per-workgroup max |x_i| -> partials[blockIdx.x] (operand of the fixed-point sweep)
__global__ void __launch_bounds__(HIPEIG_BLOCK) absmax_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ partials);
