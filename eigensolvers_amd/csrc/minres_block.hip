// Lock-step block MINRES: the nBlock inner solves of one block-Lanczos iteration
// (inexact_Lanczos.py:319-320 calls typeClass.solve once per block vector, on the same operator
// and shift; feast.py:189-201 does the same for the m0 right-hand sides of a contour point)
// advanced TOGETHER, one tall-skinny block product per iteration instead of nBlock operator sweeps.
//
// Every column runs exactly the recurrences of the single-vector driver (minres.hip; the scalar
// code is shared through minres_device.h): its own MinresState record, its own stopping tests in
// SciPy's order, its own iteration count.  A column that has stopped is masked: its iterate is no
// longer touched, so what is returned for it is the iterate of the iteration it stopped at, as
// scipy.sparse.linalg.minres (numpyVector.py:163) would return it.  All vectors are interleaved
// blocks ([row][8]); one iteration = three kernels like the single-vector driver:
//   KA  Y = A V - (beta/oldb) R1, V = R2/beta    (block product, fused)  + partials <v_j, y_j>
//   KC  Y -= (alfa/beta) R2                                               + partials <y_j, y_j>
//   KD  W = (V - oldeps W1 - delta W2)/gamma ; X += phi W                  + partials <x_j, x_j>
#include "minres_device.h"
#include "spmm_device.h"

int hipeig_block_pick_variant(hipeig_ctx* c, hipeig_csr* A, int K, int for_solve);
BcooView hipeig_bcoo_view(const hipeig_csr* A, int K);
size_t hipeig_bcoo_lds_bytes(const hipeig_csr* A, int K);
int hipeig_bcoo_grid(const hipeig_csr* A, int K);
int hipeig_block_allgather(hipeig_ctx* c, hipeig_csr* A, int K, const double* xb_local, const double** xb_full);
int hipeig_block_pack(hipeig_ctx* c, int K, int64_t n, int k, const double* const* cols, double* blk);
int hipeig_block_unpack(hipeig_ctx* c, int K, int64_t n, int k, const double* blk, double* const* cols);
int hipeig_rowowner_grid(const hipeig_ctx* c, const hipeig_csr* A);
double* hipeig_gather_block_slot(hipeig_ctx* c, const GatherLayout& gl, int K);
int hipeig_block_reserve(hipeig_ctx* c, const GatherLayout& gl, int K);

#define MRB_PART_STRIDE (HIPEIG_MAX_PARTIALS * BCOO_KMAX)     // doubles between the three partial areas

// Per-operand sum of `count` partial records for the operand threadIdx.x % K (count == 1: the record
// has already been reduced, e.g. by an all-reduce).
template <int K>
__device__ __forceinline__ double sum_or_value_cols(const double* p, int count, double* lds) {
  if (count == 1) return p[threadIdx.x % K];
  return block_sum_partials_cols<K>(p, count, lds);
}

template <int K>
struct MinresBlockEpilogue {
  double sigma, sign, s, c1;         // s, c1: the scalars of THIS thread's operand (threadIdx.x % 8)
  int use_r1;
  const double* __restrict__ r2l;    // local rows of R2 (v = s*r2)
  const double* __restrict__ r1;
  double* __restrict__ y;
  __device__ __forceinline__ void elem(int64_t r, int j, double sum, double& acc) const {
    const int64_t i = r * K + j;
    const double v = s * r2l[i];
    double yv = sign * (mul_rn(sigma, v) - s * sum);
    if (use_r1) yv -= c1 * r1[i];
    y[i] = yv;
    acc = fma(v, yv, acc);
  }
};

// VARIANT 2: window-blocked (TCOO-B) sweep, 1024 threads; VARIANT 1: row-owner CSR sweep, 256 threads.
// LATE_TESTS = 1: the form of a row-partitioned run - the stopping tests of the previous iteration are not evaluated
// here but in KC's prologue, where its <x,x> arrives with the all-reduce that also carries this sweep's <v,y>.
template <int VARIANT, int K, int LATE_TESTS = 0>
__global__ void __launch_bounds__(VARIANT == 2 ? BCOO_THREADS : HIPEIG_BLOCK)
minres_block_ka_kernel(BcooView T, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                       const double* __restrict__ val, int64_t nrows, const double* __restrict__ xg, MinresArgs a,
                       const MinresState* __restrict__ Sin, MinresState* __restrict__ Sout,
                       const double* __restrict__ r2l, const double* __restrict__ r1, double* __restrict__ y,
                       double* __restrict__ partials) {
  __shared__ double red[(VARIANT == 2 ? BCOO_THREADS : HIPEIG_BLOCK) / 64 * K];
  __shared__ double sh_s[K], sh_c1[K];
  __shared__ int sh_use[K], sh_live;
  extern __shared__ double bcoo_lds[];
  const double xx = LATE_TESTS ? 0.0 : sum_or_value_cols<K>(a.pD, a.nD, red);
  if (threadIdx.x == 0) sh_live = 0;
  __syncthreads();
  if (threadIdx.x < K) {
    MinresState S = Sin[threadIdx.x];
    if (!LATE_TESTS) minres_tests(S, (S.itn > 0 && !S.done) ? xx : 0.0, a);
    if (blockIdx.x == 0) Sout[threadIdx.x] = S;
    const int use = (!S.done && S.itn >= 1);
    sh_s[threadIdx.x] = S.done ? 0.0 : S.s;
    sh_c1[threadIdx.x] = use ? S.beta / S.oldb : 0.0;
    sh_use[threadIdx.x] = use;
    if (!S.done) atomicOr(&sh_live, 1);
  }
  __syncthreads();
  if (!sh_live) return;                                   // every column has stopped
  MinresBlockEpilogue<K> epi;
  const int j = threadIdx.x % K;
  epi.sigma = a.sigma; epi.sign = a.sign; epi.s = sh_s[j]; epi.c1 = sh_c1[j]; epi.use_r1 = sh_use[j];
  epi.r2l = r2l; epi.r1 = r1; epi.y = y;
  double acc = 0.0;
  if (VARIANT == 2) bcoo_wg_sweep<K>(T, xg, epi, acc, bcoo_lds);
  else csr_rowowner_block_sweep<K>(rowptr, col, val, nrows, xg, epi, acc);
  const double tot = block_reduce_cols<K>(acc, red);
  if (threadIdx.x < K) partials[(size_t)blockIdx.x * K + threadIdx.x] = tot;
}

// Fold (a0, a1) - partials of operands 2(t % (K/2)) and the next - over the workgroup; record in threads 0..K-1.
template <int K>
__device__ __forceinline__ void block_reduce_pairs(double a0, double a1, double* lds, double* __restrict__ out_record) {
#pragma unroll
  for (int off = K / 2; off < 64; off <<= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane < K / 2) { lds[wid * K + lane * 2] = a0; lds[wid * K + lane * 2 + 1] = a1; }
  __syncthreads();
  if (threadIdx.x < K) {
    double r = lds[threadIdx.x];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w * K + threadIdx.x];
    out_record[threadIdx.x] = r;
  }
}

template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
minres_block_kc_kernel(int64_t n, MinresArgs a, const MinresState* __restrict__ Sin, MinresState* __restrict__ Sout,
                       const double* __restrict__ r2, double* __restrict__ y, double* __restrict__ partials, int test_prev) {
  __shared__ double red[HIPEIG_BLOCK / 64 * K];
  __shared__ double sh_c[K];
  const double alfa = sum_or_value_cols<K>(a.pA, a.nA, red);
  const double xx = test_prev ? sum_or_value_cols<K>(a.pD, a.nD, red) : 0.0;
  if (threadIdx.x < K) {
    MinresState S = Sin[threadIdx.x];
    if (test_prev) minres_tests(S, (S.itn > 0 && !S.done) ? xx : 0.0, a);      // row-partitioned run: the tests of the iteration before
    if (!S.done) S.alfa = alfa;
    if (blockIdx.x == 0) Sout[threadIdx.x] = S;
    sh_c[threadIdx.x] = S.done ? 0.0 : S.alfa / S.beta;
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;           // multiple of K/2: operand pair fixed per thread
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = (int)(t0 % (K / 2)) * 2;
  const double c0 = sh_c[j0], c1 = sh_c[j0 + 1];
  const double2* r22 = reinterpret_cast<const double2*>(r2);
  double2* y2 = reinterpret_cast<double2*>(y);
  double a0 = 0.0, a1 = 0.0;
  for (int64_t t = t0; t < n * (K / 2); t += stride) {
    const double2 rv = r22[t];
    double2 yv = y2[t];
    yv.x -= c0 * rv.x; yv.y -= c1 * rv.y;
    y2[t] = yv;
    a0 = fma(yv.x, yv.x, a0); a1 = fma(yv.y, yv.y, a1);
  }
  block_reduce_pairs<K>(a0, a1, red, partials + (size_t)blockIdx.x * K);
}

template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
minres_block_kd_kernel(int64_t n, MinresArgs a, const MinresState* __restrict__ Sin, MinresState* __restrict__ Sout,
                       const double* __restrict__ r2old, const double* __restrict__ w1, const double* __restrict__ w2,
                       double* __restrict__ w, double* __restrict__ x, double* __restrict__ partials) {
  __shared__ double red[HIPEIG_BLOCK / 64 * K];
  __shared__ double sh_sold[K], sh_oldeps[K], sh_delta[K], sh_denom[K], sh_phi[K];
  __shared__ int sh_done[K];
  double bb;
  if (a.sC > 0) {                                       // one share of <y_j,y_j> per rank, in the slots of the operand exchange
    bb = 0.0;
    if (threadIdx.x < K)
      for (int r = 0; r < a.nC; ++r) bb += a.pC[(int64_t)r * a.sC + threadIdx.x];
  } else {
    bb = sum_or_value_cols<K>(a.pC, a.nC, red);
  }
  if (threadIdx.x < K) {
    MinresState S = Sin[threadIdx.x];
    sh_sold[threadIdx.x] = S.s;
    if (!S.done) minres_advance(S, bb);
    if (blockIdx.x == 0) Sout[threadIdx.x] = S;
    sh_oldeps[threadIdx.x] = S.oldeps; sh_delta[threadIdx.x] = S.delta;
    sh_denom[threadIdx.x] = S.denom; sh_phi[threadIdx.x] = S.phi;
    sh_done[threadIdx.x] = S.done;
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = (int)(t0 % (K / 2)) * 2, j1 = j0 + 1;
  const double s0 = sh_sold[j0], e0 = sh_oldeps[j0], d0 = sh_delta[j0], q0 = sh_denom[j0], p0 = sh_phi[j0];
  const double s1 = sh_sold[j1], e1 = sh_oldeps[j1], d1 = sh_delta[j1], q1 = sh_denom[j1], p1 = sh_phi[j1];
  const bool live0 = !sh_done[j0], live1 = !sh_done[j1];
  const double2* r2 = reinterpret_cast<const double2*>(r2old);
  const double2* w12 = reinterpret_cast<const double2*>(w1);
  const double2* w22 = reinterpret_cast<const double2*>(w2);
  double2* wn2 = reinterpret_cast<double2*>(w);
  double2* x2 = reinterpret_cast<double2*>(x);
  double a0 = 0.0, a1 = 0.0;
  if (live0 || live1) {
    for (int64_t t = t0; t < n * (K / 2); t += stride) {
      const double2 rv = r2[t], b1 = w12[t], b2 = w22[t];
      double2 xv = x2[t], wn;
      // a stopped column keeps its iterate; its w is never read again, so what is stored there is irrelevant
      wn.x = (s0 * rv.x - e0 * b1.x - d0 * b2.x) * q0;
      wn.y = (s1 * rv.y - e1 * b1.y - d1 * b2.y) * q1;
      if (live0) xv.x += p0 * wn.x;
      if (live1) xv.y += p1 * wn.y;
      wn2[t] = wn;
      x2[t] = xv;
      a0 = fma(xv.x, xv.x, a0); a1 = fma(xv.y, xv.y, a1);
    }
  }
  block_reduce_pairs<K>(a0, a1, red, partials + (size_t)blockIdx.x * K);
}

// Row-partitioned run: this rank's two records (<v,y> of the sweep, <x,x> of the KD before it) of K sums each, at
// out + 0 / 8, from the partial areas - ONE all-reduce of 16 doubles follows.  One workgroup per record.
template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
sum_two_records_kernel(const double* __restrict__ pA, int nA, const double* __restrict__ pD, int nD, double* __restrict__ out) {
  __shared__ double red[HIPEIG_BLOCK / 64 * K];
  const double* p = blockIdx.x == 0 ? pA : pD;
  const int cnt = blockIdx.x == 0 ? nA : nD;
  const double v = block_sum_partials_cols<K>(p, cnt, red);
  if (threadIdx.x < K) out[blockIdx.x * 8 + threadIdx.x] = v;
}

// End-of-chunk evaluation of the stopping tests (what KA's prologue would do next).
template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
minres_block_check_kernel(MinresArgs a, MinresState* __restrict__ S0) {
  __shared__ double red[HIPEIG_BLOCK / 64 * K];
  const double xx = sum_or_value_cols<K>(a.pD, a.nD, red);
  if (threadIdx.x < K) {
    MinresState S = S0[threadIdx.x];
    minres_tests(S, (S.itn > 0 && !S.done) ? xx : 0.0, a);
    S0[threadIdx.x] = S;
  }
}

// one record of 8 per-operand sums from `count` partial records (distributed path, before the all-reduce)
template <int K>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
sum_partials_cols_kernel(const double* __restrict__ p, int count, double* __restrict__ out) {
  __shared__ double red[HIPEIG_BLOCK / 64 * K];
  const double v = block_sum_partials_cols<K>(p, count, red);
  if (threadIdx.x < K) out[threadIdx.x] = v;
}

template <int K>
static int minres_block_impl(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign, int k,
                             const double* const* b, double* const* x, double rtol, int maxiter,
                             int* info, double* out_stats) {
  HIPEIG_REQUIRE(info != nullptr && b != nullptr && x != nullptr, "null argument");
  HIPEIG_REQUIRE(k >= 1 && k <= K, "more right-hand sides than the interleave width");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0, "sign must be +1 or -1");
  HIPEIG_REQUIRE(maxiter >= 1, "maxiter must be positive");
  HIPEIG_REQUIRE(c->collectives || A->nrows == A->ncols, "the inner solve needs a square operator (or a row partition)");
  const int64_t n = A->nrows;
  for (int j = 0; j < k; ++j) {
    info[j] = 0;
    HIPEIG_REQUIRE(b[j] != x[j], "x must not alias b");
  }
  if (out_stats) memset(out_stats, 0, (size_t)k * 8 * sizeof(double));
  if (n == 0) return 0;

  // recurrence records: <b_j, b_j> with the same reduction as the single-vector driver
  if (!c->d_mrb_state) {
    HIPEIG_CHECK(hipMalloc((void**)&c->d_mrb_state, 3 * BCOO_KMAX * sizeof(MinresState)));
    HIPEIG_CHECK(hipHostMalloc((void**)&c->h_mrb_state, BCOO_KMAX * sizeof(MinresState), hipHostMallocDefault));
  }
  MinresState* h = c->h_mrb_state;
  int live = 0;
  for (int j = 0; j < K; ++j) {
    double bb = 0.0;
    if (j < k && hipeig_dot(c, n, b[j], b[j], &bb)) return 1;
    if (bb > 0.0) { minres_init_state(&h[j], bb); ++live; }
    else {                                   // padding column, or beta1 == 0: the exact solution is x0 = 0
      memset(&h[j], 0, sizeof(MinresState));
      h[j].done = 1;
    }
  }
  if (live == 0) {
    for (int j = 0; j < k; ++j) if (hipeig_vec_fill(c, x[j], n, 0.0)) return 1;
    return 0;
  }

  // workspace: R[3] (r1, r2, y rotate), W[3] (w1, w2, w rotate) and the iterate block
  const int64_t nb = n * K;
  if (c->mrb_ws_n < nb) {
    if (c->mrb_ws) HIPEIG_CHECK(hipFree(c->mrb_ws));
    c->mrb_ws = nullptr; c->mrb_ws_n = 0;
    HIPEIG_CHECK(hipMalloc((void**)&c->mrb_ws, (size_t)nb * 7 * sizeof(double)));
    c->mrb_ws_n = nb;
  }
  double* R[3] = {c->mrb_ws, c->mrb_ws + c->mrb_ws_n, c->mrb_ws + 2 * c->mrb_ws_n};
  double* W[3] = {c->mrb_ws + 3 * c->mrb_ws_n, c->mrb_ws + 4 * c->mrb_ws_n, c->mrb_ws + 5 * c->mrb_ws_n};
  double* xw = c->mrb_ws + 6 * c->mrb_ws_n;
  if (hipeig_block_pack(c, K, n, k, b, R[0])) return 1;
  HIPEIG_CHECK(hipMemsetAsync(R[1], 0, (size_t)c->mrb_ws_n * 6 * sizeof(double), c->stream));
  MinresState* V = c->d_mrb_state;
  HIPEIG_CHECK(hipMemcpyAsync(V, h, K * sizeof(MinresState), hipMemcpyHostToDevice, c->stream));
  if (hipeig_sync_checked(c)) return 4;       // the pinned records are rewritten by the first copy-back

  const int bv = hipeig_block_pick_variant(c, A, K, 1);
  if (bv < 0) return 1;
  const BcooView tview = hipeig_bcoo_view(A, K);
  if (bv == 2)
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)minres_block_ka_kernel<2, K>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)HIPEIG_BCOO_LDS_MAX));
  const int gA = (bv == 2) ? hipeig_bcoo_grid(A, K) : hipeig_rowowner_grid(c, A);
  const int nsweepA = (bv == 2) ? (tview.nunits + gA - 1) / gA : 1;
  HIPEIG_REQUIRE((int64_t)nsweepA * gA <= HIPEIG_MAX_PARTIALS, "too many sweeps for the partial-sum buffer");
  // element-wise kernels: every workgroup sums the previous kernel's gE x K partials in its prologue, so FEWER, fatter
  // workgroups pay twice (less prologue traffic, fewer partials).  Measured (tools/experiments/mrb_grid_sweep.sh, N = 1e6,
  // K = 8, ms per block iteration): 4 / 8 / 16 / 32 / 64 items per thread -> 0.483 / 0.483 / 0.469 / 0.463 / 0.496.
  int per_thread = 32;
  if (const char* e = getenv("HIPEIG_MRB_PER_THREAD")) per_thread = atoi(e) > 0 ? atoi(e) : 32;
  const int gE = grid_for(n * (K / 2), per_thread);
  double* pA = c->d_partials;
  double* pC = c->d_partials + MRB_PART_STRIDE;
  double* pD = c->d_partials + 2 * MRB_PART_STRIDE;
  HIPEIG_REQUIRE(c->partials_doubles >= (size_t)3 * MRB_PART_STRIDE, "partial-sum workspace too small");
  const bool dist = c->collectives != 0;
  // row-partitioned run: reduced records of 8 - [0..7] <v,y>, [8..15] <x,x> (ONE all-reduce), [16..23] / [24..31] the
  // end-of-solve flush
  double* red = c->d_scalars + 2048;
  MinresArgs a;
  a.sigma = sigma; a.sign = sign; a.rtol = rtol; a.maxiter = maxiter; a.sC = 0;
  const int nPA = gA * nsweepA;
  a.pA = dist ? red + 0 : pA; a.nA = dist ? 1 : nPA;
  a.pC = pC; a.nC = gE;
  a.pD = dist ? red + 8 : pD; a.nD = dist ? 1 : gE;
  double* slot = nullptr;
  MinresArgs a_kd = a;                                  // KD of a partitioned run: <y_j,y_j> as one share per rank from the exchange's slots
  if (dist) {
    HIPEIG_CHECK(hipMemsetAsync(red, 0, 32 * sizeof(double), c->stream));
    if (hipeig_block_reserve(c, A->gl, K)) return 1;
    slot = hipeig_gather_block_slot(c, A->gl, K);
    HIPEIG_REQUIRE(slot != nullptr, "no block exchange buffer");
    a_kd.pC = c->xb_full + A->gl.slot(0) * K; a_kd.nC = c->nranks; a_kd.sC = A->gl.cstride(A->gl.nchunks - 1) * K;
  }
  c->mr_collectives = 0;

  auto launch_ka = [&](const double* xg, double* r2, double* r1, double* yb, bool late) {
    if (bv == 2) {
      BcooView tv = tview;
      for (int sw = 0; sw < nsweepA; ++sw) {
        tv.unit_begin = sw * gA;
        if (late) hipLaunchKernelGGL((minres_block_ka_kernel<2, K, 1>), dim3(gA), dim3(BCOO_THREADS), hipeig_bcoo_lds_bytes(A, K), c->stream,
                                     tv, A->d_rowptr, A->d_col, A->d_val, n, xg, a, V + 0, V + 8, r2, r1, yb, pA + (size_t)sw * gA * K);
        else hipLaunchKernelGGL((minres_block_ka_kernel<2, K>), dim3(gA), dim3(BCOO_THREADS), hipeig_bcoo_lds_bytes(A, K), c->stream,
                                tv, A->d_rowptr, A->d_col, A->d_val, n, xg, a, V + 0, V + 8, r2, r1, yb, pA + (size_t)sw * gA * K);
      }
    } else if (late) {
      hipLaunchKernelGGL((minres_block_ka_kernel<1, K, 1>), dim3(gA), dim3(HIPEIG_BLOCK), 0, c->stream,
                         tview, A->d_rowptr, A->d_col, A->d_val, n, xg, a, V + 0, V + 8, r2, r1, yb, pA);
    } else {
      hipLaunchKernelGGL((minres_block_ka_kernel<1, K>), dim3(gA), dim3(HIPEIG_BLOCK), 0, c->stream,
                         tview, A->d_rowptr, A->d_col, A->d_val, n, xg, a, V + 0, V + 8, r2, r1, yb, pA);
    }
  };
  // KD of iteration `it` (buffers of that iteration)
  auto launch_kd = [&](int it, const MinresArgs& ak) {
    hipLaunchKernelGGL(minres_block_kd_kernel<K>, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, ak, V + 16, V + 0,
                       R[it % 3], W[(it + 1) % 3], W[(it + 2) % 3], W[it % 3], xw, pD);
  };

  if (bv == 2)
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)minres_block_ka_kernel<2, K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)HIPEIG_BCOO_LDS_MAX));

  auto enqueue_iteration = [&](int it) -> int {
    double* r2 = R[it % 3];
    double* yb = R[(it + 1) % 3];
    double* r1 = R[(it + 2) % 3];
    const double* xg = nullptr;
    if (hipeig_block_allgather(c, A, K, r2, &xg)) return 4;
    if (dist) {
      // Two collectives per iteration, as for one right-hand side (minres.hip): the exchange of the operand block carries
      // every rank's shares of <y_j,y_j> (so beta is the directly reduced norm of the updated y, not <y,y> - alfa^2), and
      // one all-reduce carries <v_j,y_j> together with the <x_j,x_j> of the KD that ran just before the sweep.
      ++c->mr_collectives;
      if (it > 0) launch_kd(it - 1, a_kd);
      else HIPEIG_CHECK(hipMemsetAsync(pD, 0, (size_t)gE * K * sizeof(double), c->stream));
      launch_ka(xg, r2, r1, yb, true);
      hipLaunchKernelGGL(sum_two_records_kernel<K>, dim3(2), dim3(HIPEIG_BLOCK), 0, c->stream, pA, nPA, pD, gE, red);
      if (hipeig_allreduce_sum(c, red, 16)) return 4;
      ++c->mr_collectives;
      hipLaunchKernelGGL(minres_block_kc_kernel<K>, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, V + 8, V + 16, r2, yb, pC, it > 0 ? 1 : 0);
      hipLaunchKernelGGL(sum_partials_cols_kernel<K>, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, pC, gE, slot);   // travels with the next exchange
      return 0;
    }
    launch_ka(xg, r2, r1, yb, false);
    hipLaunchKernelGGL(minres_block_kc_kernel<K>, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, V + 8, V + 16, r2, yb, pC, 0);
    launch_kd(it, a);
    return 0;
  };

  const int chunk = 16;
  int it = 0;
  bool all_done = false;
  while (it < maxiter && !all_done) {
    const int iend = (it + chunk < maxiter) ? it + chunk : maxiter;
    for (; it < iend; ++it) {
      const int rc = enqueue_iteration(it);
      if (rc) return rc;
    }
    HIPEIG_CHECK(hipGetLastError());
    if (dist && iend < maxiter) {
      // no flush at a chunk boundary: the records KC left hold the tests of the iteration before; the pending KD runs at
      // the start of the next chunk
      HIPEIG_CHECK(hipMemcpyAsync(h, V + 16, K * sizeof(MinresState), hipMemcpyDeviceToHost, c->stream));
    } else {
      MinresArgs ac = a;
      if (dist) {
        // end of the solve: the last KD and the last tests on their own (their sums have not travelled: two small all-reduces)
        HIPEIG_CHECK(hipMemcpyAsync(red + 16, slot, K * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (hipeig_allreduce_sum(c, red + 16, K)) return 4;
        ++c->mr_collectives;
        MinresArgs ak = a;
        ak.pC = red + 16; ak.nC = 1; ak.sC = 0;
        launch_kd(iend - 1, ak);
        hipLaunchKernelGGL(sum_partials_cols_kernel<K>, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, pD, gE, red + 24);
        if (hipeig_allreduce_sum(c, red + 24, K)) return 4;
        ++c->mr_collectives;
        ac.pD = red + 24;
      }
      hipLaunchKernelGGL(minres_block_check_kernel<K>, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, ac, V + 0);
      HIPEIG_CHECK(hipMemcpyAsync(h, V, K * sizeof(MinresState), hipMemcpyDeviceToHost, c->stream));
    }
    if (hipeig_sync_checked(c)) return 4;
    all_done = true;
    for (int j = 0; j < K; ++j) all_done = all_done && h[j].done;
  }
  HIPEIG_REQUIRE(all_done, "block MINRES left the iteration loop without a stop code in every column");
  if (hipeig_block_unpack(c, K, n, k, xw, x)) return 1;
  for (int j = 0; j < k; ++j) {
    info[j] = (h[j].istop == 6) ? maxiter : 0;
    if (out_stats) {
      double* st = out_stats + (size_t)j * 8;
      st[0] = h[j].itn; st[1] = h[j].istop; st[2] = h[j].rnorm; st[3] = h[j].Anorm;
      st[4] = h[j].ynorm; st[5] = h[j].test1; st[6] = h[j].test2; st[7] = h[j].Acond;
    }
  }
  return 0;
}

extern "C" int hipeig_minres_block(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign, int k,
                                   const double* const* b, double* const* x, double rtol, int maxiter,
                                   int* info, double* out_stats) {
  HIPEIG_REQUIRE(k >= 1 && k <= BCOO_KMAX, "a block solve takes 1..8 right-hand sides");
  // blocks of <= 4 use the narrow interleave: twice the accumulator rows per workgroup, half the vector traffic
  if (k <= 4) return minres_block_impl<4>(c, A, sigma, sign, k, b, x, rtol, maxiter, info, out_stats);
  return minres_block_impl<8>(c, A, sigma, sign, k, b, x, rtol, maxiter, info, out_stats);
}
