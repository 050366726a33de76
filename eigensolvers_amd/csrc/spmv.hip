// Sparse operator: creation, layout preparation and the y = Hx / y = sign*(sigma*x - Hx)
// entry points.  Kernel bodies live in spmv_device.h.
#include <vector>
#include "spmv_device.h"

// y[r] = a_self*xl[r] + a_sum*sum  (a_self = 0, a_sum = 1: plain product;
// a_self = sign*sigma, a_sum = -sign: the shifted operator of numpyVector.py:152/154).
// Two roundings like the reference's sigma*x - H@x (no contraction into an FMA).
struct AxpyEpilogue {
  double a_self, a_sum;
  const double* __restrict__ xl;
  double* __restrict__ y;
  __device__ __forceinline__ void row(int64_t r, double sum, double& acc) const {
    const double t = (a_self == 0.0) ? 0.0 : __dmul_rn(a_self, xl[r]);
    y[r] = __dadd_rn(t, __dmul_rn(a_sum, sum));
  }
};

__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmv_stream_kernel(CsrView A, const double* __restrict__ x, AxpyEpilogue epi) {
  __shared__ double prod[SPMV_NNZ_PER_BLOCK];
  double acc = 0.0;
  csr_stream_sweep(A, x, epi, acc, prod);
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmv_vector_kernel(CsrView A, const double* __restrict__ x, AxpyEpilogue epi) {
  double acc = 0.0;
  csr_vector_sweep(A, x, epi, acc);
}

CsrView hipeig_csr_view(const hipeig_csr* A) {
  CsrView v;
  v.rowptr = A->d_rowptr;
  v.col = A->d_col;
  v.val = A->d_val;
  v.row_blocks = A->d_row_blocks;
  v.n_row_blocks = A->n_row_blocks;
  v.nrows = A->nrows;
  v.group = A->lanes_per_row;
  return v;
}

int hipeig_spmv_grid(const hipeig_csr* A, int variant) {
  int64_t g;
  if (variant == 1) {
    const int64_t groups_per_block = HIPEIG_BLOCK / A->lanes_per_row;
    g = (A->nrows + groups_per_block - 1) / groups_per_block;
  } else {
    g = A->n_row_blocks;
  }
  if (g < 1) g = 1;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return (int)g;
}

static int launch_spmv(hipeig_ctx* c, hipeig_csr* A, double a_self, double a_sum,
                       const double* x, double* y) {
  if (A->nrows == 0) return 0;
  const double* xg = nullptr;
  if (hipeig_allgather_x(c, x, A->nrows, &xg)) return 4;
  AxpyEpilogue epi{a_self, a_sum, x, y};
  const int variant = A->variant ? A->variant : 2;
  const CsrView v = hipeig_csr_view(A);
  const int g = hipeig_spmv_grid(A, variant);
  if (variant == 1)
    hipLaunchKernelGGL(spmv_vector_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, v, xg, epi);
  else
    hipLaunchKernelGGL(spmv_stream_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, v, xg, epi);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hipeig_spmv(hipeig_ctx* c, hipeig_csr* A, const double* x, double* y) {
  HIPEIG_REQUIRE(x != y, "in-place product is not supported");
  return launch_spmv(c, A, 0.0, 1.0, x, y);
}

extern "C" int hipeig_spmv_shift(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign,
                                 const double* x, double* y) {
  HIPEIG_REQUIRE(x != y, "in-place product is not supported");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0, "sign must be +1 or -1");
  return launch_spmv(c, A, sign * sigma, -sign, x, y);
}

// ---- layout preparation ------------------------------------------------------------------
__global__ void remap_cols_kernel(int32_t* __restrict__ col, int64_t nnz, const int64_t* __restrict__ offs,
                                  int nranks, int64_t stride) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += step) {
    const int64_t j = col[p];
    int r = 0;
    while (r + 1 < nranks && j >= offs[r + 1]) ++r;
    col[p] = (int32_t)(r * stride + (j - offs[r]));
  }
}

// Partition the rows into blocks of <= SPMV_NNZ_PER_BLOCK non-zeros (a longer single row
// gets a block of its own) and pick the sub-wave width from the mean row length.
// rowptr32: host copy of the (int32) row pointer.
int hipeig_csr_finalize(hipeig_ctx* c, hipeig_csr* A, const int32_t* rowptr32) {
  std::vector<int32_t> rb;
  rb.reserve((size_t)(A->nnz / (SPMV_NNZ_PER_BLOCK / 2) + 16));
  rb.push_back(0);
  int64_t r = 0;
  while (r < A->nrows) {
    const int32_t base = rowptr32[r];
    int64_t e = r + 1;
    while (e < A->nrows && rowptr32[e + 1] - base <= SPMV_NNZ_PER_BLOCK && e - r < 1024) ++e;
    rb.push_back((int32_t)e);
    r = e;
  }
  A->n_row_blocks = (int32_t)rb.size() - 1;
  HIPEIG_CHECK(hipMalloc((void**)&A->d_row_blocks, rb.size() * sizeof(int32_t)));
  HIPEIG_CHECK(hipMemcpyAsync(A->d_row_blocks, rb.data(), rb.size() * sizeof(int32_t),
                              hipMemcpyHostToDevice, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  const double mean = A->nrows ? (double)A->nnz / (double)A->nrows : 0.0;
  A->lanes_per_row = mean < 6 ? 4 : mean < 24 ? 8 : mean < 96 ? 16 : mean < 384 ? 32 : 64;
  A->bytes = (A->nrows + 1) * 4 + A->nnz * 12 + (int64_t)rb.size() * 4;

  // distributed: gather row counts, size x_full and remap global columns to its layout
  int64_t stride = 0;
  if (hipeig_comm_setup_rows(c, A->nrows, &stride)) return 4;
  if (c->comm && c->nranks > 1) {
    std::vector<int64_t> offs(c->nranks + 1, 0);
    for (int k = 0; k < c->nranks; ++k) offs[k + 1] = offs[k] + c->row_counts[k];
    HIPEIG_REQUIRE(offs[c->nranks] == A->ncols, "row counts over ranks must add up to ncols");
    HIPEIG_REQUIRE(offs[c->rank] == A->row_offset, "row_offset does not match the rank order");
    HIPEIG_REQUIRE(stride * c->nranks < (int64_t)1 << 31, "gathered operand too long for int32 columns");
    int64_t* d_offs = (int64_t*)(c->d_scalars + 1024);
    HIPEIG_CHECK(hipMemcpyAsync(d_offs, offs.data(), sizeof(int64_t) * (c->nranks + 1),
                                hipMemcpyHostToDevice, c->stream));
    if (A->nnz > 0) {
      hipLaunchKernelGGL(remap_cols_kernel, dim3(2048), dim3(HIPEIG_BLOCK), 0, c->stream,
                         A->d_col, A->nnz, d_offs, c->nranks, stride);
      HIPEIG_CHECK(hipGetLastError());
    }
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    A->col_stride = stride;
  }
  return 0;
}

extern "C" int hipeig_csr_create(hipeig_ctx* c, int64_t nrows, int64_t ncols, int64_t row_offset,
                                 const int64_t* rowptr, const int32_t* col, const double* val,
                                 hipeig_csr** out) {
  HIPEIG_REQUIRE(out && rowptr, "null argument");
  HIPEIG_REQUIRE(nrows >= 0 && ncols >= 0 && ncols < ((int64_t)1 << 31), "bad shape");
  HIPEIG_REQUIRE(rowptr[0] == 0, "rowptr[0] must be 0");
  const int64_t nnz = rowptr[nrows];
  HIPEIG_REQUIRE(nnz >= 0 && nnz < ((int64_t)1 << 31), "local nnz must fit int32");
  std::vector<int32_t> rp32((size_t)nrows + 1);
  for (int64_t i = 0; i <= nrows; ++i) {
    if (i > 0) HIPEIG_REQUIRE(rowptr[i] >= rowptr[i - 1], "rowptr must be non-decreasing");
    rp32[i] = (int32_t)rowptr[i];
  }
  for (int64_t p = 0; p < nnz; ++p)
    HIPEIG_REQUIRE(col[p] >= 0 && col[p] < ncols, "column index out of range");
  hipeig_csr* A = (hipeig_csr*)calloc(1, sizeof(hipeig_csr));
  HIPEIG_REQUIRE(A != nullptr, "out of host memory");
  A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->row_offset = row_offset;
  HIPEIG_CHECK(hipMalloc((void**)&A->d_rowptr, (size_t)(nrows + 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_col, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_val, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double)));
  HIPEIG_CHECK(hipMemcpyAsync(A->d_rowptr, rp32.data(), (size_t)(nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (nnz > 0) {
    HIPEIG_CHECK(hipMemcpyAsync(A->d_col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPEIG_CHECK(hipMemcpyAsync(A->d_val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  int rc = hipeig_csr_finalize(c, A, rp32.data());
  if (rc) { hipeig_csr_destroy(c, A); return rc; }
  *out = A;
  return 0;
}

extern "C" int hipeig_csr_destroy(hipeig_ctx* c, hipeig_csr* A) {
  if (!A) return 0;
  hipStreamSynchronize(c->stream);
  if (A->d_rowptr) hipFree(A->d_rowptr);
  if (A->d_col) hipFree(A->d_col);
  if (A->d_val) hipFree(A->d_val);
  if (A->d_row_blocks) hipFree(A->d_row_blocks);
  free(A);
  return 0;
}

extern "C" int hipeig_csr_info(hipeig_csr* A, int64_t info[8]) {
  info[0] = A->nrows; info[1] = A->ncols; info[2] = A->nnz; info[3] = A->row_offset;
  info[4] = A->variant; info[5] = A->bytes; info[6] = A->n_row_blocks; info[7] = A->lanes_per_row;
  return 0;
}

extern "C" int hipeig_csr_set_variant(hipeig_csr* A, int variant) {
  HIPEIG_REQUIRE(variant >= 0 && variant <= 2, "unknown variant");
  A->variant = variant;
  return 0;
}

extern "C" int hipeig_csr_download(hipeig_ctx* c, hipeig_csr* A, int64_t* rowptr, int32_t* col, double* val) {
  HIPEIG_REQUIRE(A->col_stride == 0 || col == nullptr, "columns of a partitioned operator are remapped");
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  if (rowptr) {
    std::vector<int32_t> rp((size_t)A->nrows + 1);
    HIPEIG_CHECK(hipMemcpy(rp.data(), A->d_rowptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < rp.size(); ++i) rowptr[i] = rp[i];
  }
  if (col && A->nnz) HIPEIG_CHECK(hipMemcpy(col, A->d_col, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (val && A->nnz) HIPEIG_CHECK(hipMemcpy(val, A->d_val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
