// Block operands: interleaved [row][K] storage (K = 4 or 8), the TCOO-B layout build and the block
// product Y = H X (hipeig_spmm).  Kernel bodies live in spmm_device.h; the lock-step block MINRES
// (minres_block.hip) fuses its vector updates into the same sweeps.
#include <vector>
#include "spmm_device.h"

struct PtrTable8 { const double* p[BCOO_KPACK]; };
struct OutTable8 { double* p[BCOO_KPACK]; };

static inline int layout_slot(int K) { return K == 4 ? 0 : K == 8 ? 1 : 2; }

// ---- interleave / de-interleave ---------------------------------------------------------
// Thread t owns the 16-byte pair (2t, 2t+1) of the block: row t / (K/2), operands 2(t % (K/2)) and the next,
// so the block side is one fully coalesced 16-byte access per lane and every column is touched in
// runs of 64 / (K/2) consecutive rows per wave instruction.  Missing operands (k < K) read as zero.
__device__ __forceinline__ const double* pick8(const PtrTable8& t, int j) {
  const double* p = t.p[0];
#pragma unroll
  for (int q = 1; q < BCOO_KPACK; ++q) p = (j == q) ? t.p[q] : p;
  return p;
}
__device__ __forceinline__ double* pick8(const OutTable8& t, int j) {
  double* p = t.p[0];
#pragma unroll
  for (int q = 1; q < BCOO_KPACK; ++q) p = (j == q) ? t.p[q] : p;
  return p;
}

template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
block_pack_kernel(int64_t n, int k, PtrTable8 cols, double* __restrict__ blk) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;           // multiple of K/2: the operand pair is fixed
  const int j0 = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % (K / 2)) * 2);
  const double* c0 = j0 < k ? pick8(cols, j0) : nullptr;
  const double* c1 = j0 + 1 < k ? pick8(cols, j0 + 1) : nullptr;
  double2* out = reinterpret_cast<double2*>(blk);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * (K / 2); t += stride) {
    const int64_t row = t / (K / 2);
    double2 v;
    v.x = c0 ? c0[row] : 0.0;
    v.y = c1 ? c1[row] : 0.0;
    out[t] = v;
  }
}

template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
block_unpack_kernel(int64_t n, int k, const double* __restrict__ blk, OutTable8 cols) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int j0 = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % (K / 2)) * 2);
  double* c0 = j0 < k ? pick8(cols, j0) : nullptr;
  double* c1 = j0 + 1 < k ? pick8(cols, j0 + 1) : nullptr;
  const double2* in = reinterpret_cast<const double2*>(blk);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * (K / 2); t += stride) {
    const int64_t row = t / (K / 2);
    const double2 v = in[t];
    if (c0) c0[row] = v.x;
    if (c1) c1[row] = v.y;
  }
}

int hipeig_block_pack(hipeig_ctx* c, int K, int64_t n, int k, const double* const* cols, double* blk) {
  HIPEIG_REQUIRE((K == 4 || K == 8 || K == 16) && k >= 1 && k <= K, "a block holds 1..K operands, K = 4, 8 or 16");
  if (n == 0) return 0;
  PtrTable8 t;
  for (int j = 0; j < BCOO_KPACK; ++j) t.p[j] = j < k ? cols[j] : nullptr;
  const int g = grid_stream(n * (K / 2) * 2);
  if (K == 4) hipLaunchKernelGGL(block_pack_kernel<4>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, t, blk);
  else if (K == 8) hipLaunchKernelGGL(block_pack_kernel<8>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, t, blk);
  else hipLaunchKernelGGL(block_pack_kernel<16>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, t, blk);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

int hipeig_block_unpack(hipeig_ctx* c, int K, int64_t n, int k, const double* blk, double* const* cols) {
  HIPEIG_REQUIRE((K == 4 || K == 8 || K == 16) && k >= 1 && k <= K, "a block holds 1..K operands, K = 4, 8 or 16");
  if (n == 0) return 0;
  OutTable8 t;
  for (int j = 0; j < BCOO_KPACK; ++j) t.p[j] = j < k ? cols[j] : nullptr;
  const int g = grid_stream(n * (K / 2) * 2);
  if (K == 4) hipLaunchKernelGGL(block_unpack_kernel<4>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, blk, t);
  else if (K == 8) hipLaunchKernelGGL(block_unpack_kernel<8>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, blk, t);
  else hipLaunchKernelGGL(block_unpack_kernel<16>, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, blk, t);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// ---- TCOO-B construction -------------------------------------------------------------------
// One cursor per (unit, window) tile: count, exclusive scan on the host, scatter.  The slot a
// non-zero takes inside its tile depends on scheduling; the set of non-zeros of a tile does not
// (the sweep adds with atomics, so no order inside a tile is promised anyway).
__global__ void __launch_bounds__(256)
bcoo_bin_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                int64_t nrows, int rw, int wbits, int nwin, uint32_t* __restrict__ cursor,
                uint32_t* __restrict__ b_idx, double* __restrict__ b_val, int fill) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t wmask = (1u << wbits) - 1u;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t unit = r / rw;
    const uint32_t rl = (uint32_t)(r - unit * rw);
    uint32_t* cur = cursor + unit * nwin;
    const int s = rowptr[r], e = rowptr[r + 1];
    for (int p = s + lane; p < e; p += 64) {
      const uint32_t cc = (uint32_t)col[p];
      const uint32_t slot = atomicAdd(cur + (cc >> wbits), 1u);
      if (fill) {
        b_idx[slot] = (rl << wbits) | (cc & wmask);
        b_val[slot] = val[p];
      }
    }
  }
}

size_t hipeig_bcoo_lds_bytes(const hipeig_csr* A, int K) {
  const auto& L = A->bl[layout_slot(K)];
  return (size_t)L.rw * K * sizeof(double) + ((size_t)L.nwin + 2) * sizeof(uint32_t);
}

BcooView hipeig_bcoo_view(const hipeig_csr* A, int K) {
  const auto& L = A->bl[layout_slot(K)];
  BcooView t;
  t.idx = L.idx; t.val = L.val; t.off = L.off;
  t.nunits = L.nunits; t.nwin = L.nwin; t.wbits = L.wbits; t.rw = L.rw;
  t.unit_begin = 0;
  t.nrows = A->nrows;
  return t;
}

int hipeig_bcoo_grid(const hipeig_csr* A, int K) {
  const auto& L = A->bl[layout_slot(K)];
  return L.wgs_per_sweep < L.nunits ? L.wgs_per_sweep : L.nunits;
}

// Decide between the window-blocked and the row-owner block kernel for interleave width K and build the
// former's layout (idempotent).  Returns 2 (TCOO-B) or 1 (row-owner), -1 on failure.  for_solve: the caller is the block
// MINRES, whose sweeps leave per-workgroup partials - it cannot take a layout that needs more than HIPEIG_MAX_PARTIALS
// workgroups per product and keeps the row-owner kernel there (N = 1e7: 16 sweeps of 256).
int hipeig_block_pick_variant(hipeig_ctx* c, hipeig_csr* A, int K, int for_solve) {
  auto& L = A->bl[layout_slot(K)];
  A->last_block_k = K;
  if (A->block_variant == 1 || A->nnz == 0 || A->nrows == 0) return A->last_block_variant = 1;
  // an operator pinned to a reproducible kernel (variants 1-3, 5) keeps that promise for block products and block
  // solves too: the row-owner kernel adds a row's terms in a fixed order, the window-blocked one uses fp64 atomics
  if (A->block_variant == 0 && (A->reproducible || (A->variant != 0 && A->variant != 4))) return A->last_block_variant = 1;
  auto fits_solve = [&](const hipeig_csr::BcooLayout& l) {
    const int64_t g = l.wgs_per_sweep < l.nunits ? l.wgs_per_sweep : l.nunits;
    return g > 0 && ((l.nunits + g - 1) / g) * g <= HIPEIG_MAX_PARTIALS;
  };
  if (L.state == 1) return A->last_block_variant = (for_solve && A->block_variant == 0 && !fits_solve(L)) ? 1 : 2;
  if (L.state == 2 && A->block_variant == 0) return A->last_block_variant = 1;
  int wbits = K == 16 ? 10 : K == 8 ? 11 : 12;           // 128 KiB of the operand block per window (measured best of 2^8..2^15 rows at N = 1e6,
                                                         // K = 8: 32 windows fit one L2, so workgroups that drift apart still hit)
  if (const char* e = getenv("HIPEIG_BCOO_WBITS")) wbits = atoi(e);          // tuning knob
  if (wbits < 8 || wbits > 20) { hipeig_set_error("HIPEIG_BCOO_WBITS out of range"); return -1; }
  while (wbits > 8 && ((int64_t)1 << (wbits - 1)) >= A->gather_len) --wbits;
  int64_t nwin = (A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits;
  while (nwin > BCOO_MAX_WIN && wbits < 20) { ++wbits; nwin = (A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits; }
  int64_t rw_max = ((int64_t)HIPEIG_BCOO_LDS_MAX - (nwin + 2) * 4) / (K * 8);
  if (rw_max > ((int64_t)1 << (32 - wbits)) - 1) rw_max = ((int64_t)1 << (32 - wbits)) - 1;    // 0xFFFFFFFF stays the padding mark
  int64_t sweeps = (A->nrows + (int64_t)c->num_cu * rw_max - 1) / ((int64_t)c->num_cu * rw_max);
  if (sweeps < 1) sweeps = 1;
  int64_t rw = (A->nrows + sweeps * c->num_cu - 1) / (sweeps * c->num_cu);
  if (rw < 8) rw = 8;
  if (rw > rw_max) rw = rw_max;
  if (const char* e = getenv("HIPEIG_BCOO_RW")) { rw = atoi(e); if (rw < 1 || rw > rw_max) { hipeig_set_error("HIPEIG_BCOO_RW out of range"); return -1; } }
  // L2 reuse of the operand lines inside one XCD (32 workgroups share a window): round 2 kept the windows only above ~2
  // touches per line (spmm_device.h).  Round 4 measured the other end (tools/experiments/pair_block_width.py, block_bench.py):
  // once the operand block has outgrown the 256 MiB Infinity Cache the row-owner kernel's gathers go to HBM one line each
  // and the windows win although hardly a line is touched twice - N = 1e7: K = 8 15.2 -> 11.8 ms per product, 8 complex
  // operands (K = 16) 2.90 -> 1.85 ms per operand - and the 16-wide block wins with them at N = 1e6 too (0.134 -> 0.109).
  // A block that fits one L2 needs no windows.
  const double rows_per_line = 128.0 / (K * 8);
  const double touches = 32.0 * (double)rw * ((double)A->nnz / (double)A->nrows) * rows_per_line / (double)A->gather_len;
  const bool fits_l2 = A->gather_len * (int64_t)(K * 8) <= ((int64_t)3 << 20);
  const bool beyond_mall = A->gather_len * (int64_t)(K * 8) > ((int64_t)256 << 20);
  if (A->block_variant == 0 && (fits_l2 || (touches < 2.0 && !beyond_mall && K != 16))) {
    L.state = 2;
    return A->last_block_variant = 1;
  }
  if (for_solve && A->block_variant == 0) {            // would the layout fit the solve?  (same arithmetic as below)
    const int64_t nu = (A->nrows + rw - 1) / rw, g = nu < c->num_cu ? nu : c->num_cu;
    if (((nu + g - 1) / g) * g > HIPEIG_MAX_PARTIALS) return A->last_block_variant = 1;      // state stays undecided: a product may still build it
  }
  const int64_t nunits = (A->nrows + rw - 1) / rw;
  const size_t ntile = (size_t)nunits * (size_t)nwin;
  if (ntile >= ((size_t)1 << 30)) { L.state = 2; return A->last_block_variant = 1; }
  uint32_t* d_cur = nullptr;
  if (hipMalloc((void**)&d_cur, ntile * sizeof(uint32_t)) != hipSuccess) { hipeig_set_error("out of device memory (TCOO-B cursors)"); return -1; }
  hipMemsetAsync(d_cur, 0, ntile * sizeof(uint32_t), c->stream);
  const int grid = 8 * c->num_cu;
  hipLaunchKernelGGL(bcoo_bin_kernel, dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->nrows,
                     (int)rw, wbits, (int)nwin, d_cur, (uint32_t*)nullptr, (double*)nullptr, 0);
  std::vector<uint32_t> off(ntile + 1);
  if (hipMemcpyAsync(off.data(), d_cur, ntile * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) { hipFree(d_cur); hipeig_set_error("TCOO-B count pass failed"); return -1; }
  uint64_t run = 0;
  for (size_t i = 0; i < ntile; ++i) { const uint32_t n = off[i]; off[i] = (uint32_t)run; run += n; }
  off[ntile] = (uint32_t)run;
  if (run != (uint64_t)A->nnz) { hipFree(d_cur); hipeig_set_error("TCOO-B count pass lost non-zeros"); return -1; }
  bool ok = hipMemcpyAsync(d_cur, off.data(), ntile * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
  ok = ok && hipMalloc((void**)&L.off, (ntile + 1) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc((void**)&L.idx, (size_t)A->nnz * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc((void**)&L.val, (size_t)A->nnz * sizeof(double)) == hipSuccess;
  ok = ok && hipMemcpyAsync(L.off, off.data(), (ntile + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(bcoo_bin_kernel, dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->nrows,
                       (int)rw, wbits, (int)nwin, d_cur, L.idx, L.val, 1);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;   // `off` must outlive the copies
  }
  hipFree(d_cur);
  if (!ok) {
    if (L.off) hipFree(L.off);
    if (L.idx) hipFree(L.idx);
    if (L.val) hipFree(L.val);
    L.off = nullptr; L.idx = nullptr; L.val = nullptr;
    hipeig_set_error("TCOO-B build failed (out of device memory?)");
    return -1;
  }
  L.nunits = (int)nunits; L.nwin = (int)nwin; L.wbits = wbits; L.rw = (int)rw;
  L.wgs_per_sweep = c->num_cu;                          // one 1024-thread workgroup (all of the LDS) per CU
  L.state = 1;
  A->bytes += (int64_t)A->nnz * 12 + (int64_t)(ntile + 1) * 4;
  return A->last_block_variant = 2;
}

// All-gather of an interleaved block of width K into the layout the remapped column indices address (the chunk-major
// layout of the single-vector exchange, K doubles per position; comm.hip).  Single rank: the local block is the operand.
int hipeig_block_allgather(hipeig_ctx* c, hipeig_csr* A, int K, const double* xb_local, const double** xb_full) {
  if (!c->collectives) { *xb_full = xb_local; return 0; }
  HIPEIG_REQUIRE(A->col_stride > 0 && A->gl.h * A->gl.nchunks >= A->nrows, "operator was not prepared for this communicator");
  return hipeig_allgather_block(c, A->gl, K, xb_local, A->nrows, xb_full);
}

// ---- plain block product -------------------------------------------------------------------
template <int K>
struct StoreBlockEpilogue {
  double* __restrict__ Y;
  __device__ __forceinline__ void elem(int64_t r, int j, double sum, double& acc) const { Y[r * K + j] = sum; }
};

// Columns (2p, 2p + 1) of the block are the real and imaginary halves of complex operand p; the epilogue applies the
// complex shift of a contour solve, y_p = sign*(z*x_p - H x_p) (feast.py:83-90 -> numpyVector.py:152-161), with the
// roundings of PairEpilogue (spmv.hip): the real part of the shift and the operator sum separately, then the zi term.
// xl: the packed operand block restricted to this operator's rows.
template <int K>
struct ShiftPairBlockEpilogue {
  double ar, ai, as;                                   // sign*zr, sign*zi, -sign
  const double* __restrict__ xl;
  double* __restrict__ Y;
  __device__ __forceinline__ void elem(int64_t r, int j, double sum, double& acc) const {
    const int64_t base = r * K + (j & ~1);
    const double vr = xl[base], vi = xl[base + 1];
    const double own = (j & 1) ? vi : vr;
    const double t = add_rn(mul_rn(ar, own), mul_rn(as, sum));
    Y[r * K + j] = (j & 1) ? fma(ai, vr, t) : fma(-ai, vi, t);
  }
};

template <int K, class Epi>
__global__ void __launch_bounds__(BCOO_THREADS)
spmm_bcoo_kernel(BcooView T, const double* __restrict__ X, Epi epi) {
  extern __shared__ double bcoo_lds[];
  double acc = 0.0;
  bcoo_wg_sweep<K>(T, X, epi, acc, bcoo_lds);
}

template <int K, class Epi>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmm_rowowner_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                     int64_t nrows, const double* __restrict__ X, Epi epi) {
  double acc = 0.0;
  csr_rowowner_block_sweep<K>(rowptr, col, val, nrows, X, epi, acc);
}

int hipeig_rowowner_grid(const hipeig_ctx* c, const hipeig_csr* A) {
  int64_t g = (A->nrows + 3) / 4;
  if (g > 8 * (int64_t)c->num_cu) g = 8 * (int64_t)c->num_cu;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return g < 1 ? 1 : (int)g;
}

// Block product on interleaved blocks of width K (local rows) with the epilogue `epi`; Xb is this rank's slice.
template <int K, class Epi>
static int spmm_block_run(hipeig_ctx* c, hipeig_csr* A, const double* Xb, const Epi& epi) {
  if (A->nrows == 0) return 0;
  const int bv = hipeig_block_pick_variant(c, A, K, 0);
  if (bv < 0) return 1;
  const double* xg = nullptr;
  if (hipeig_block_allgather(c, A, K, Xb, &xg)) return 4;
  if (bv == 2) {
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmm_bcoo_kernel<K, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIPEIG_BCOO_LDS_MAX));
    BcooView t = hipeig_bcoo_view(A, K);
    const int g = hipeig_bcoo_grid(A, K);
    for (int ub = 0; ub < t.nunits; ub += g) {                // one launch per sweep of the windows
      t.unit_begin = ub;
      hipLaunchKernelGGL((spmm_bcoo_kernel<K, Epi>), dim3(g), dim3(BCOO_THREADS), hipeig_bcoo_lds_bytes(A, K), c->stream, t, xg, epi);
    }
  } else {
    hipLaunchKernelGGL((spmm_rowowner_kernel<K, Epi>), dim3(hipeig_rowowner_grid(c, A)), dim3(HIPEIG_BLOCK), 0, c->stream,
                       A->d_rowptr, A->d_col, A->d_val, A->nrows, xg, epi);
  }
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// Yb = H Xb
template <int K>
static int spmm_block_impl(hipeig_ctx* c, hipeig_csr* A, const double* Xb, double* Yb) {
  return spmm_block_run<K>(c, A, Xb, StoreBlockEpilogue<K>{Yb});
}

static int ensure_blk_ws(hipeig_ctx* c, size_t doubles) {
  if (c->blk_ws_doubles >= doubles) return 0;
  if (c->blk_ws) HIPEIG_CHECK(hipFree(c->blk_ws));
  c->blk_ws = nullptr; c->blk_ws_doubles = 0;
  HIPEIG_CHECK(hipMalloc((void**)&c->blk_ws, doubles * sizeof(double)));
  c->blk_ws_doubles = doubles;
  return 0;
}

extern "C" int hipeig_spmm(hipeig_ctx* c, hipeig_csr* A, int k, const double* const* X, double* const* Y) {
  HIPEIG_REQUIRE(k >= 1 && X && Y, "bad arguments");
  if (A->nrows == 0) return 0;
  if (c->collectives && !c->comm && !c->loop) {
    // a communicator without RCCL has no exchange for block operands: k products through the direct exchange instead
    for (int j = 0; j < k; ++j)
      if (hipeig_spmv(c, A, X[j], Y[j])) return 1;
    return 0;
  }
  // operand slice: a partitioned run hands over local slices; a row slab applied to full-length
  // operands (single process) gathers from the whole operand, so the block has ncols rows
  const int64_t nx = c->collectives ? A->nrows : A->ncols;
  const int64_t ny = A->nrows;
  if (ensure_blk_ws(c, (size_t)(nx + ny) * BCOO_KMAX)) return 1;
  double* Xi = c->blk_ws;
  double* Yi = c->blk_ws + (size_t)nx * BCOO_KMAX;
  for (int j0 = 0; j0 < k;) {
    const int K = (k - j0 <= 4) ? 4 : 8;                     // blocks of <= 4 take the narrow interleave (twice the rows per workgroup)
    const int kk = (k - j0 < K) ? k - j0 : K;
    if (hipeig_block_pack(c, K, nx, kk, X + j0, Xi)) return 1;
    if (K == 4 ? spmm_block_impl<4>(c, A, Xi, Yi) : spmm_block_impl<8>(c, A, Xi, Yi)) return 1;
    if (hipeig_block_unpack(c, K, ny, kk, Yi, Y + j0)) return 1;
    j0 += kk;
  }
  return 0;
}

// The complex matvec of the contour solves for SEVERAL right-hand sides at once (feast.py:198-200: the m0 solves of one
// contour point share operator and shift): y_p = sign*(z*x_p - H x_p), p < npairs, complex operands as (re, im) buffers.
// Four complex operands fill an 8-wide block (two a 4-wide one), so the (index, value) stream of the operator is read
// once per four operands instead of once each; the shift is applied in the block product's epilogue.  A row-partitioned
// or direct-only context and npairs = 1 take hipeig_spmv_shift_pair per operand.
extern "C" int hipeig_spmm_shift_pairs(hipeig_ctx* c, hipeig_csr* A, int npairs, double zr, double zi, double sign,
                                       const double* const* Xre, const double* const* Xim, double* const* Yre, double* const* Yim) {
  HIPEIG_REQUIRE(npairs >= 1 && Xre && Xim && Yre && Yim, "bad arguments");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0, "sign must be +1 or -1");
  if (A->nrows == 0) return 0;
  if (npairs == 1 || c->collectives) {
    for (int p = 0; p < npairs; ++p)
      if (hipeig_spmv_shift_pair(c, A, zr, zi, sign, Xre[p], Xim[p], Yre[p], Yim[p])) return 1;
    return 0;
  }
  const int64_t nx = A->ncols, ny = A->nrows;
  // complex operands per pass over the operator: 4 (an 8-wide block, 64 B per operand row: a gather uses half a line) or 8
  // (16 wide, 128 B = one whole line per gather; HIPEIG_PAIR_BLOCK_WIDTH, see EXPERIMENTS.md R4-block16)
  // Measured per complex operand, 4 / 8 per pass (tools/experiments/pair_block_width.py): N = 1e6 0.107 / 0.109, 2e6 0.387 /
  // 0.365, 4e6 0.933 / 0.973, 1e7 2.98 / 1.85 ms (single pair products: 0.177 / 0.51 / 1.46 / 3.94).  Within +-6 % while the
  // 8-wide block (64 B per row) still sits in the 256 MiB Infinity Cache; once it does not, its gathers pay a whole line from
  // HBM for half a line of data and the 16-wide block wins by 38 %.
  int wide = (nx * (int64_t)64 > ((int64_t)288 << 20)) ? 8 : 4;
  if (const char* e = getenv("HIPEIG_PAIR_BLOCK_WIDTH")) wide = atoi(e) >= 8 ? 8 : 4;
  if (ensure_blk_ws(c, (size_t)(nx + ny) * BCOO_KPACK)) return 1;
  double* Xi = c->blk_ws;
  double* Yi = c->blk_ws + (size_t)nx * BCOO_KPACK;
  const double ar = sign * zr, ai = sign * zi, as = -sign;
  for (int p0 = 0; p0 < npairs;) {
    const int left = npairs - p0;
    const int np = (wide == 8 && left > 4) ? (left < 8 ? left : 8) : (left < 4 ? left : 4);
    const int K = (np <= 2) ? 4 : (np <= 4) ? 8 : 16;
    const double* cols[BCOO_KPACK];
    double* outs[BCOO_KPACK];
    for (int p = 0; p < np; ++p) {
      cols[2 * p] = Xre[p0 + p]; cols[2 * p + 1] = Xim[p0 + p];
      outs[2 * p] = Yre[p0 + p]; outs[2 * p + 1] = Yim[p0 + p];
    }
    if (hipeig_block_pack(c, K, nx, 2 * np, cols, Xi)) return 1;
    int rc;
    if (K == 4) rc = spmm_block_run<4>(c, A, Xi, ShiftPairBlockEpilogue<4>{ar, ai, as, Xi + A->row_offset * 4, Yi});
    else if (K == 8) rc = spmm_block_run<8>(c, A, Xi, ShiftPairBlockEpilogue<8>{ar, ai, as, Xi + A->row_offset * 8, Yi});
    else rc = spmm_block_run<16>(c, A, Xi, ShiftPairBlockEpilogue<16>{ar, ai, as, Xi + A->row_offset * 16, Yi});
    if (rc) return rc;
    if (hipeig_block_unpack(c, K, ny, 2 * np, Yi, outs)) return 1;
    p0 += np;
  }
  return 0;
}

extern "C" int hipeig_csr_block_info(hipeig_csr* A, int64_t info[4]) {
  const auto& L = A->bl[layout_slot(A->last_block_k == 4 ? 4 : A->last_block_k == 16 ? 16 : 8)];
  info[0] = A->last_block_variant; info[1] = L.nunits; info[2] = L.nwin; info[3] = L.rw;
  return 0;
}

extern "C" int hipeig_csr_set_block_variant(hipeig_csr* A, int variant) {
  HIPEIG_REQUIRE(variant >= 0 && variant <= 2, "unknown block variant (0 auto, 1 row-owner, 2 window-blocked)");
  A->block_variant = variant;
  return 0;
}
