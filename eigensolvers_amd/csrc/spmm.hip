// Block operands: interleaved [row][8] storage, the TCOO-B layout build and the block product
// Y = H X (hipeig_spmm).  Kernel bodies live in spmm_device.h; the lock-step block MINRES
// (minres_block.hip) fuses its vector updates into the same sweeps.
#include <vector>
#include "spmm_device.h"

struct PtrTable8 { const double* p[BCOO_K]; };
struct OutTable8 { double* p[BCOO_K]; };

// ---- interleave / de-interleave ---------------------------------------------------------
// Thread t owns the 16-byte pair (2t, 2t+1) of the block: row t/4, operands 2(t%4) and 2(t%4)+1, so the
// block side is one fully coalesced 16-byte access per lane and every column is touched in 128-byte
// runs (16 consecutive rows per wave instruction).  Missing operands (k < 8) read as zero.
__device__ __forceinline__ const double* pick8(const PtrTable8& t, int j) {
  const double* p = t.p[0];
#pragma unroll
  for (int q = 1; q < BCOO_K; ++q) p = (j == q) ? t.p[q] : p;
  return p;
}
__device__ __forceinline__ double* pick8(const OutTable8& t, int j) {
  double* p = t.p[0];
#pragma unroll
  for (int q = 1; q < BCOO_K; ++q) p = (j == q) ? t.p[q] : p;
  return p;
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
block_pack_kernel(int64_t n, int k, PtrTable8 cols, double* __restrict__ blk) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;           // multiple of 4: the operand pair is fixed
  const int j0 = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & 3) * 2);
  const double* c0 = j0 < k ? pick8(cols, j0) : nullptr;
  const double* c1 = j0 + 1 < k ? pick8(cols, j0 + 1) : nullptr;
  double2* out = reinterpret_cast<double2*>(blk);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 4; t += stride) {
    const int64_t row = t >> 2;
    double2 v;
    v.x = c0 ? c0[row] : 0.0;
    v.y = c1 ? c1[row] : 0.0;
    out[t] = v;
  }
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
block_unpack_kernel(int64_t n, int k, const double* __restrict__ blk, OutTable8 cols) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int j0 = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & 3) * 2);
  double* c0 = j0 < k ? pick8(cols, j0) : nullptr;
  double* c1 = j0 + 1 < k ? pick8(cols, j0 + 1) : nullptr;
  const double2* in = reinterpret_cast<const double2*>(blk);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 4; t += stride) {
    const int64_t row = t >> 2;
    const double2 v = in[t];
    if (c0) c0[row] = v.x;
    if (c1) c1[row] = v.y;
  }
}

int hipeig_block_pack(hipeig_ctx* c, int64_t n, int k, const double* const* cols, double* blk) {
  HIPEIG_REQUIRE(k >= 1 && k <= BCOO_K, "a block holds 1..8 operands");
  if (n == 0) return 0;
  PtrTable8 t;
  for (int j = 0; j < BCOO_K; ++j) t.p[j] = j < k ? cols[j] : nullptr;
  hipLaunchKernelGGL(block_pack_kernel, dim3(grid_for(n * 4, 2)), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, t, blk);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

int hipeig_block_unpack(hipeig_ctx* c, int64_t n, int k, const double* blk, double* const* cols) {
  HIPEIG_REQUIRE(k >= 1 && k <= BCOO_K, "a block holds 1..8 operands");
  if (n == 0) return 0;
  OutTable8 t;
  for (int j = 0; j < BCOO_K; ++j) t.p[j] = j < k ? cols[j] : nullptr;
  hipLaunchKernelGGL(block_unpack_kernel, dim3(grid_for(n * 4, 2)), dim3(HIPEIG_BLOCK), 0, c->stream, n, k, blk, t);
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

size_t hipeig_bcoo_lds_bytes(const hipeig_csr* A) {
  return (size_t)A->b_rw * BCOO_K * sizeof(double) + ((size_t)A->b_nwin + 2) * sizeof(uint32_t);
}

BcooView hipeig_bcoo_view(const hipeig_csr* A) {
  BcooView t;
  t.idx = A->b_idx; t.val = A->b_val; t.off = A->b_off;
  t.nunits = A->b_nunits; t.nwin = A->b_nwin; t.wbits = A->b_wbits; t.rw = A->b_rw;
  t.unit_begin = 0;
  t.nrows = A->nrows;
  return t;
}

// Decide between the window-blocked and the row-owner block kernel and build the former's layout
// (idempotent).  Returns 2 (TCOO-B) or 1 (row-owner), -1 on failure.
int hipeig_block_pick_variant(hipeig_ctx* c, hipeig_csr* A) {
  if (A->block_variant == 1 || A->nnz == 0 || A->nrows == 0) return A->last_block_variant = 1;
  // an operator pinned to a reproducible kernel (variants 1-3, 5) keeps that promise for block products and block
  // solves too: the row-owner kernel adds a row's terms in a fixed order, the window-blocked one uses fp64 atomics
  if (A->block_variant == 0 && A->variant != 0 && A->variant != 4) return A->last_block_variant = 1;
  if (A->b_state == 1) return A->last_block_variant = 2;
  if (A->b_state == 2 && A->block_variant == 0) return A->last_block_variant = 1;
  int wbits = 11;                                        // 2 Ki columns x 64 B = 128 KiB of X per window (measured best of 8..15 at N = 1e6: 32 windows fit one L2, so workgroups that drift apart still hit)
  if (const char* e = getenv("HIPEIG_BCOO_WBITS")) wbits = atoi(e);          // tuning knob
  if (wbits < 8 || wbits > 20) { hipeig_set_error("HIPEIG_BCOO_WBITS out of range"); return -1; }
  while (wbits > 8 && ((int64_t)1 << (wbits - 1)) >= A->gather_len) --wbits;
  int64_t nwin = (A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits;
  while (nwin > BCOO_MAX_WIN && wbits < 20) { ++wbits; nwin = (A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits; }
  int64_t rw_max = ((int64_t)HIPEIG_BCOO_LDS_MAX - (nwin + 2) * 4) / (BCOO_K * 8);
  if (rw_max > BCOO_MAX_RW) rw_max = BCOO_MAX_RW;
  if (rw_max > ((int64_t)1 << (32 - wbits)) - 1) rw_max = ((int64_t)1 << (32 - wbits)) - 1;    // 0xFFFFFFFF stays the padding mark
  int64_t sweeps = (A->nrows + (int64_t)c->num_cu * rw_max - 1) / ((int64_t)c->num_cu * rw_max);
  if (sweeps < 1) sweeps = 1;
  int64_t rw = (A->nrows + sweeps * c->num_cu - 1) / (sweeps * c->num_cu);
  if (rw < 8) rw = 8;
  if (rw > rw_max) rw = rw_max;
  if (const char* e = getenv("HIPEIG_BCOO_RW")) { rw = atoi(e); if (rw < 1 || rw > rw_max) { hipeig_set_error("HIPEIG_BCOO_RW out of range"); return -1; } }
  // L2 reuse of the operand lines inside one XCD (32 workgroups share a window): below ~2 touches per
  // line the windows buy nothing (spmm_device.h); a block that fits one L2 needs no windows either.
  const double touches = 32.0 * (double)rw * ((double)A->nnz / (double)A->nrows) * 2.0 / (double)A->gather_len;
  const bool fits_l2 = A->gather_len * (int64_t)(BCOO_K * 8) <= ((int64_t)3 << 20);
  if (A->block_variant == 0 && (touches < 2.0 || fits_l2)) {
    A->b_state = 2;
    return A->last_block_variant = 1;
  }
  const int64_t nunits = (A->nrows + rw - 1) / rw;
  const size_t ntile = (size_t)nunits * (size_t)nwin;
  if (ntile >= ((size_t)1 << 30)) { A->b_state = 2; return A->last_block_variant = 1; }
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
  ok = ok && hipMalloc((void**)&A->b_off, (ntile + 1) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc((void**)&A->b_idx, (size_t)A->nnz * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc((void**)&A->b_val, (size_t)A->nnz * sizeof(double)) == hipSuccess;
  ok = ok && hipMemcpyAsync(A->b_off, off.data(), (ntile + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(bcoo_bin_kernel, dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->nrows,
                       (int)rw, wbits, (int)nwin, d_cur, A->b_idx, A->b_val, 1);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;   // `off` must outlive the copies
  }
  hipFree(d_cur);
  if (!ok) {
    if (A->b_off) hipFree(A->b_off);
    if (A->b_idx) hipFree(A->b_idx);
    if (A->b_val) hipFree(A->b_val);
    A->b_off = nullptr; A->b_idx = nullptr; A->b_val = nullptr;
    hipeig_set_error("TCOO-B build failed (out of device memory?)");
    return -1;
  }
  A->b_nunits = (int)nunits; A->b_nwin = (int)nwin; A->b_wbits = wbits; A->b_rw = (int)rw;
  A->b_wgs_per_sweep = c->num_cu;                       // one 1024-thread workgroup (all of the LDS) per CU
  A->b_state = 1;
  A->bytes += (int64_t)A->nnz * 12 + (int64_t)(ntile + 1) * 4;
  return A->last_block_variant = 2;
}

// All-gather of an interleaved block: rank r's rows land at xb_full + r*stride*8 (the layout the
// remapped column indices address).  Single rank: the local block is the operand.
int hipeig_block_allgather(hipeig_ctx* c, hipeig_csr* A, const double* xb_local, const double** xb_full) {
  if (!c->collectives) { *xb_full = xb_local; return 0; }
  const int64_t stride = A->col_stride, need = stride * c->nranks * BCOO_K;
  HIPEIG_REQUIRE(stride >= A->nrows, "operator was not prepared for this communicator");
  if (c->xb_full_n < need) {
    if (c->xb_full) HIPEIG_CHECK(hipFree(c->xb_full));
    c->xb_full = nullptr; c->xb_full_n = 0;
    HIPEIG_CHECK(hipMalloc((void**)&c->xb_full, (size_t)need * sizeof(double)));
    HIPEIG_CHECK(hipMemsetAsync(c->xb_full, 0, (size_t)need * sizeof(double), c->stream));
    c->xb_full_n = need;
  }
  double* mine = c->xb_full + (int64_t)c->rank * stride * BCOO_K;
  HIPEIG_CHECK(hipMemcpyAsync(mine, xb_local, (size_t)A->nrows * BCOO_K * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (hipeig_allgather_f64(c, mine, c->xb_full, (size_t)stride * BCOO_K)) return 4;
  *xb_full = c->xb_full;
  return 0;
}

// ---- plain block product -------------------------------------------------------------------
struct StoreBlockEpilogue {
  double* __restrict__ Y;
  __device__ __forceinline__ void elem(int64_t r, int j, double sum, double& acc) const { Y[r * BCOO_K + j] = sum; }
};

__global__ void __launch_bounds__(BCOO_THREADS)
spmm_bcoo_kernel(BcooView T, const double* __restrict__ X, StoreBlockEpilogue epi) {
  extern __shared__ double bcoo_lds[];
  double acc = 0.0;
  bcoo_wg_sweep(T, X, epi, acc, bcoo_lds);
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmm_rowowner_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                     int64_t nrows, const double* __restrict__ X, StoreBlockEpilogue epi) {
  double acc = 0.0;
  csr_rowowner_block_sweep(rowptr, col, val, nrows, X, epi, acc);
}

int hipeig_rowowner_grid(const hipeig_ctx* c, const hipeig_csr* A) {
  int64_t g = (A->nrows + 3) / 4;
  if (g > 8 * (int64_t)c->num_cu) g = 8 * (int64_t)c->num_cu;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return g < 1 ? 1 : (int)g;
}

// Yb = H Xb on interleaved blocks (local rows); Xb is this rank's slice.
int hipeig_spmm_block(hipeig_ctx* c, hipeig_csr* A, const double* Xb, double* Yb) {
  if (A->nrows == 0) return 0;
  const int bv = hipeig_block_pick_variant(c, A);
  if (bv < 0) return 1;
  const double* xg = nullptr;
  if (hipeig_block_allgather(c, A, Xb, &xg)) return 4;
  StoreBlockEpilogue epi{Yb};
  if (bv == 2) {
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmm_bcoo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIPEIG_BCOO_LDS_MAX));
    BcooView t = hipeig_bcoo_view(A);
    const int g = A->b_wgs_per_sweep < A->b_nunits ? A->b_wgs_per_sweep : A->b_nunits;
    for (int ub = 0; ub < A->b_nunits; ub += g) {             // one launch per sweep of the windows
      t.unit_begin = ub;
      hipLaunchKernelGGL(spmm_bcoo_kernel, dim3(g), dim3(BCOO_THREADS), hipeig_bcoo_lds_bytes(A), c->stream, t, xg, epi);
    }
  } else {
    hipLaunchKernelGGL(spmm_rowowner_kernel, dim3(hipeig_rowowner_grid(c, A)), dim3(HIPEIG_BLOCK), 0, c->stream,
                       A->d_rowptr, A->d_col, A->d_val, A->nrows, xg, epi);
  }
  HIPEIG_CHECK(hipGetLastError());
  return 0;
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
  // operand slice: a partitioned run hands over local slices; a row slab applied to full-length
  // operands (single process) gathers from the whole operand, so the block has ncols rows
  const int64_t nx = c->collectives ? A->nrows : A->ncols;
  const int64_t ny = A->nrows;
  if (ensure_blk_ws(c, (size_t)(nx + ny) * BCOO_K)) return 1;
  double* Xi = c->blk_ws;
  double* Yi = c->blk_ws + (size_t)nx * BCOO_K;
  for (int j0 = 0; j0 < k; j0 += BCOO_K) {
    const int kk = (k - j0 < BCOO_K) ? k - j0 : BCOO_K;
    if (hipeig_block_pack(c, nx, kk, X + j0, Xi)) return 1;
    if (hipeig_spmm_block(c, A, Xi, Yi)) return 1;
    if (hipeig_block_unpack(c, ny, kk, Yi, Y + j0)) return 1;
  }
  return 0;
}

extern "C" int hipeig_csr_block_info(hipeig_csr* A, int64_t info[4]) {
  info[0] = A->last_block_variant; info[1] = A->b_nunits; info[2] = A->b_nwin; info[3] = A->b_rw;
  return 0;
}

extern "C" int hipeig_csr_set_block_variant(hipeig_csr* A, int variant) {
  HIPEIG_REQUIRE(variant >= 0 && variant <= 2, "unknown block variant (0 auto, 1 row-owner, 2 window-blocked)");
  A->block_variant = variant;
  return 0;
}
