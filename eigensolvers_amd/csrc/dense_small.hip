// linearSolver = "pardiso" (numpyVector.py:166-170): the reference's exact branch - spsolve of sigma*I - H - which it
// keeps "only for comparing with fortran" (the 4 x 4 known-answer system of unittests/test_feast_fortran.py).  Here: the
// n x (n+1) complex system [sign*(z I - H) | b] is built in the LDS of ONE workgroup from the CSR operator and solved by
// Gaussian elimination with partial pivoting, n <= HIPEIG_DENSE_MAX.  Nothing of it is on the hot path.
#include "common.h"

#define HIPEIG_DENSE_MAX 96                // 96 x 97 complex = 149 KB of the CU's 160 KB LDS
#define DENSE_THREADS 256

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cdiv(double2 a, double2 b) {
  // Smith's algorithm: no overflow in |b|^2
  if (fabs(b.x) >= fabs(b.y)) {
    const double r = b.y / b.x, d = b.x + b.y * r;
    return make_double2((a.x + a.y * r) / d, (a.y - a.x * r) / d);
  }
  const double r = b.x / b.y, d = b.x * r + b.y;
  return make_double2((a.x * r + a.y) / d, (a.y * r - a.x) / d);
}

__global__ void __launch_bounds__(DENSE_THREADS)
dense_solve_small_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                         int n, double zr, double zi, double sign, const double* __restrict__ b_re,
                         const double* __restrict__ b_im, double* __restrict__ x_re, double* __restrict__ x_im,
                         int* __restrict__ singular) {
  extern __shared__ double2 M[];            // row-major, n rows of n + 1
  __shared__ int piv;
  __shared__ int bad;
  const int ld = n + 1, tid = threadIdx.x;
  for (int k = tid; k < n * ld; k += DENSE_THREADS) M[k] = make_double2(0.0, 0.0);
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int r = tid; r < n; r += DENSE_THREADS) {                 // one thread per row: duplicates of an entry add up
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) M[r * ld + col[p]].x -= val[p];
    M[r * ld + r].x += zr;
    M[r * ld + r].y += zi;
    M[r * ld + n] = make_double2(b_re[r], b_im ? b_im[r] : 0.0);
  }
  __syncthreads();
  if (sign < 0.0) {
    for (int k = tid; k < n * ld; k += DENSE_THREADS)
      if (k % ld != n) M[k] = make_double2(-M[k].x, -M[k].y);
    __syncthreads();
  }
  for (int k = 0; k < n; ++k) {
    if (tid == 0) {                                              // partial pivoting, first largest |re| + |im|
      int best = k;
      double bm = fabs(M[k * ld + k].x) + fabs(M[k * ld + k].y);
      for (int i = k + 1; i < n; ++i) {
        const double m = fabs(M[i * ld + k].x) + fabs(M[i * ld + k].y);
        if (m > bm) { bm = m; best = i; }
      }
      piv = best;
      if (!(bm > 0.0)) bad = 1;
    }
    __syncthreads();
    if (bad) break;                                              // uniform
    if (piv != k)
      for (int j = k + tid; j <= n; j += DENSE_THREADS) {
        const double2 t = M[k * ld + j];
        M[k * ld + j] = M[piv * ld + j];
        M[piv * ld + j] = t;
      }
    __syncthreads();
    const double2 pk = M[k * ld + k];
    const int rows = n - k - 1, cols = n - k;                    // columns k+1 .. n
    for (int e = tid; e < rows * cols; e += DENSE_THREADS) {
      const int i = k + 1 + e / cols, j = k + 1 + e % cols;
      const double2 f = cdiv(M[i * ld + k], pk);
      const double2 t = cmul(f, M[k * ld + j]);
      M[i * ld + j].x -= t.x;
      M[i * ld + j].y -= t.y;
    }
    __syncthreads();
  }
  if (tid == 0) {
    *singular = bad;
    if (!bad) {
      for (int i = n - 1; i >= 0; --i) {                           // back substitution
        double2 s = M[i * ld + n];
        for (int j = i + 1; j < n; ++j) {
          const double2 t = cmul(M[i * ld + j], M[j * ld + n]);
          s.x -= t.x; s.y -= t.y;
        }
        M[i * ld + n] = cdiv(s, M[i * ld + i]);
      }
    }
  }
  __syncthreads();
  if (!bad)
    for (int r = tid; r < n; r += DENSE_THREADS) {
      x_re[r] = M[r * ld + n].x;
      if (x_im) x_im[r] = M[r * ld + n].y;
    }
}

extern "C" int hipeig_dense_solve_small(hipeig_ctx* c, hipeig_csr* A, double zr, double zi, double sign, const double* b_re,
                                        const double* b_im, double* x_re, double* x_im, int* singular) {
  HIPEIG_REQUIRE(singular != nullptr && b_re != nullptr && x_re != nullptr, "null argument");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0, "sign must be +1 or -1");
  HIPEIG_REQUIRE(!c->collectives && A->nrows == A->ncols && A->col_stride == 0, "the exact small solve needs the whole square operator on one device");
  HIPEIG_REQUIRE(A->nrows >= 1 && A->nrows <= HIPEIG_DENSE_MAX, "the exact small solve is limited to n <= 96 (it exists for the Fortran known-answer comparison)");
  HIPEIG_REQUIRE((zi == 0.0 && b_im == nullptr) || x_im != nullptr, "a complex system needs x_im");
  const int n = (int)A->nrows;
  const size_t lds = (size_t)n * (n + 1) * sizeof(double2);
  HIPEIG_CHECK(hipFuncSetAttribute((const void*)dense_solve_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int* d_flag = reinterpret_cast<int*>(c->d_scalars);
  hipLaunchKernelGGL(dense_solve_small_kernel, dim3(1), dim3(DENSE_THREADS), lds, c->stream, A->d_rowptr, A->d_col, A->d_val, n,
                     zr, zi, sign, b_re, b_im, x_re, x_im, d_flag);
  HIPEIG_CHECK(hipGetLastError());
  int* h_flag = reinterpret_cast<int*>(c->h_scalars);
  HIPEIG_CHECK(hipMemcpyAsync(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  *singular = *h_flag;
  return 0;
}
