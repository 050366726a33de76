// Device-resident MINRES for sign*(sigma*I - H) x = b  (NumpyVector.solve with
// linearSolver="minres", numpyVector.py:147-178; the recurrences are SciPy 1.15.3's
// scipy.sparse.linalg.minres, the third-party routine the reference calls at :163).
//
// One iteration = three steps, no host round trip:
//   KA  y = A v - (beta/oldb) r1,  v = r2/beta        (CSR sweep, fused)   + <v,y>
//   KC  y -= (alfa/beta) r2                                              + <y,y>
//   KD  w = (v - oldeps*w1 - delta*w2)/gamma ; x += phi*w                 + <x,x>
// in two kernels: KD(k) is element-wise and needs nothing but beta_{k+1} = sqrt(<y,y>), which KA(k+1) needs as well, so it
// runs in the row epilogue of KA(k+1)'s sweep (r1 of that sweep IS the r2 that KD(k) reads); the stopping tests of
// iteration k then sit in KC(k+1)'s prologue, where <x,x> has arrived.  Two global reductions per iteration (alfa, beta)
// is the least SciPy's recurrences allow.  HIPEIG_MINRES_FUSE_KD=0 keeps the three-kernel form on one GPU.
//
// Reductions finish in the LAST workgroup of the kernel that produces them (common.h): consumers read one double, and the
// element-wise kernels run on uncapped grids.  The recurrence scalars live in a ring of three MinresState records in
// device memory: each kernel reads one record and workgroup 0 writes the next.  Once `done` is set the remaining kernels
// of a chunk return immediately and x is left untouched, so x is exactly SciPy's iterate.
//
// Row-partitioned run: the SAME two kernels and two collectives per iteration -
//   * every rank's share of <y,y> rides on the operand all-gather of the next sweep (a scalar slot behind its slice,
//     GatherLayout::slot), so beta_{k+1}^2 is the directly reduced <y,y> of the updated y, exactly as on one GPU
//     (round 2 took it as <y,y> - alfa^2 from the un-updated y, which cancels when the right-hand side is close to an
//     eigenvector - restart vectors, converged Ritz vectors);
//   * <v,y> and the <x,x> of the KD riding on the sweep share ONE all-reduce of two doubles.
#include <math.h>
#include "spmv_device.h"

CsrView hipeig_csr_view(const hipeig_csr* A);
TcooView hipeig_tcoo_view(const hipeig_csr* A);
TcooView hipeig_tcoow_view(const hipeig_csr* A);
size_t hipeig_tcoow_lds_bytes(const hipeig_csr* A);
int hipeig_tcoow_run_plan(hipeig_ctx* c, hipeig_csr* A, const double* x_local, int fixed, TcooView* last, bool* has_last,
                          const double** xg, int* ncombine);
int64_t hipeig_tcoow_part_stride(const hipeig_csr* A);
int hipeig_tcoow_reserve(hipeig_ctx* c, const hipeig_csr* A);
int hipeig_spmv_grid(const hipeig_csr* A, int variant);
int hipeig_csr_pick_variant(hipeig_ctx* c, hipeig_csr* A);
size_t hipeig_tcoo_lds_bytes(const hipeig_csr* A);
void hipeig_phase_mark(hipeig_ctx* c, int k);

#include "minres_device.h"

// KD's element: w = (v - oldeps*w1 - delta*w2)*denom ; x += phi*w with v = s_old*r2old.  One definition for the
// stand-alone kernel and for the epilogue form, so that both evaluate the same expression.
struct MinresKd {
  double s_old, oldeps, delta, denom, phi;
  __device__ __forceinline__ void apply(double r2old, double a1, double a2, double& wn, double& xv) const {
    // spelled out with explicit fused operations (and contraction off) so that every instantiation - the stand-alone
    // kernel and the epilogue of each sweep layout - rounds identically: the two forms then agree bit for bit
#pragma clang fp contract(off)
    wn = fma(-delta, a2, fma(-oldeps, a1, s_old * r2old)) * denom;
    xv = fma(phi, wn, xv);
  }
};

struct MinresRowEpilogue {
  double sigma, sign, s, c1;
  int use_r1;
  const double* __restrict__ r2l;   // local slice of r2 (v = s*r2)
  const double* __restrict__ r1;
  double* __restrict__ y;
  // KD of the previous iteration riding on this sweep: r1 is the r2 it reads
  int do_kd;
  MinresKd kd;
  const double* __restrict__ w1;
  const double* __restrict__ w2;
  double* __restrict__ w;
  double* __restrict__ x;
  double* xx;                       // running <x,x> of this thread
  __device__ __forceinline__ void row(int64_t r, double sum, double& acc) const {
    const double v = s * r2l[r];
    double yv = sign * (mul_rn(sigma, v) - s * sum);
    const double r1v = (use_r1 || do_kd) ? r1[r] : 0.0;
    if (use_r1) yv -= c1 * r1v;
    y[r] = yv;
    acc = fma(v, yv, acc);
    if (do_kd) {
      double wn, xv = x[r];
      kd.apply(r1v, w1[r], w2[r], wn, xv);
      w[r] = wn; x[r] = xv;
      *xx = fma(xv, xv, *xx);
    }
  }
};

// What the KD riding on a KA launch needs besides the state record.
struct MinresKdArgs {
  int do_kd;
  const double* w1;
  const double* w2;
  double* w;
  double* x;
  MinresRed red;                    // <x,x>: partials laid out like the launch's <v,y> partials
};

// VARIANT 1-4: the operator sweep of that layout.  VARIANT 5: the combine step of a split TCOO-W sweep
// (T.raw_out holds T.part_base slabs of raw sums, see spmv_device.h) - same prologue, same epilogue.
// FIXED = 1 (VARIANT 4 only): fixed-point accumulators, public kernel variant 5 (spmv_device.h).
// `zero_xx`: a row-partitioned run all-reduces (<v,y>, <x,x>) as one record, so a sweep without a riding KD stores 0 there.
template <int VARIANT, int FIXED = 0>
__global__ void __launch_bounds__(VARIANT == 4 ? TCOOW_THREADS : HIPEIG_BLOCK)
minres_ka_kernel(CsrView A, TcooView T, const double* __restrict__ xg, MinresArgs a, const MinresState* __restrict__ Sin,
                 MinresState* __restrict__ Sout, const double* __restrict__ r2l,
                 const double* __restrict__ r1, double* __restrict__ y, MinresRed ra, MinresKdArgs kda, int zero_xx) {
  __shared__ double prod[VARIANT == 2 ? SPMV_NNZ_PER_BLOCK : 8];
  __shared__ double red[16];
  extern __shared__ double tcoo_lds[];
  MinresState S = *Sin;
  MinresRowEpilogue epi;
  epi.do_kd = 0;
  if (kda.do_kd) {
    // Sin is the record KC left: the scalar half of KD (minres_kd_kernel) happens here, its vector half in the
    // epilogue; the tests of that iteration wait for its <x,x> (KC's prologue)
    if (!S.done) {
      const double bb = minres_yy(a);
      epi.kd.s_old = S.s;
      minres_advance(S, bb);
      epi.kd.oldeps = S.oldeps; epi.kd.delta = S.delta; epi.kd.denom = S.denom; epi.kd.phi = S.phi;
      epi.do_kd = 1;
    }
  } else {
    const double xx = (S.itn > 0 && !S.done) ? a.pD[0] : 0.0;
    minres_tests(S, xx, a);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *Sout = S;
  if (S.done) return;
  epi.sigma = a.sigma; epi.sign = a.sign; epi.s = S.s;
  epi.use_r1 = S.itn >= 1;
  epi.c1 = epi.use_r1 ? S.beta / S.oldb : 0.0;
  epi.r2l = r2l; epi.r1 = r1; epi.y = y;
  double acc = 0.0, acc_xx = 0.0;
  epi.w1 = kda.w1; epi.w2 = kda.w2; epi.w = kda.w; epi.x = kda.x; epi.xx = &acc_xx;
  if (VARIANT == 5) tcoow_combine_sweep(T.raw_out, T.part_base, T.part_stride, T.nrows, epi, acc);
  else if (VARIANT == 4) tcoo_wg_sweep<MinresRowEpilogue, FIXED>(T, xg, epi, acc, tcoo_lds, red);
  else if (VARIANT == 3) tcoo_sweep(T, xg, epi, acc, tcoo_lds);
  else if (VARIANT == 2) csr_stream_sweep(A, xg, epi, acc, prod);
  else csr_vector_sweep(A, xg, epi, acc);
  acc = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) store_partial(ra.part + blockIdx.x, acc);
  if (kda.do_kd) {                                           // uniform: every workgroup saw the same record
    acc_xx = block_reduce_sum(acc_xx, red);
    if (threadIdx.x == 0) store_partial(kda.red.part + blockIdx.x, acc_xx);
  }
  if (last_block_ticket(ra.counter, ra.tickets, (unsigned)(ra.part - ra.base) + blockIdx.x)) {
    const double vy = sum_partials_agent(ra.base, ra.count, red);
    const double xx = kda.do_kd ? sum_partials_agent(kda.red.base, kda.red.count, red) : 0.0;
    if (threadIdx.x == 0) {
      *ra.tot = vy;
      if (kda.do_kd || zero_xx) *kda.red.tot = xx;
    }
    release_ticket_counter(ra.counter);
  }
}

// y -= (alfa/beta) r2 and <y,y>; with test_prev the stopping tests of the previous iteration first (its KD ran in the
// epilogue of the sweep before this kernel and has left <x,x>).
__global__ void __launch_bounds__(HIPEIG_BLOCK)
minres_kc_kernel(int64_t n, MinresArgs a, const MinresState* __restrict__ Sin, MinresState* __restrict__ Sout,
                 const double* __restrict__ r2, double* __restrict__ y, MinresRed rc, int test_prev) {
  __shared__ double red[4];
  MinresState S = *Sin;
  if (test_prev && !S.done) {
    const double xx = S.itn > 0 ? a.pD[0] : 0.0;
    minres_tests(S, xx, a);
  }
  if (S.done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *Sout = S;
    return;
  }
  S.alfa = a.pA[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) *Sout = S;
  const double c = S.alfa / S.beta;
  const int64_t n2 = n >> 1;
  const double2* r22 = reinterpret_cast<const double2*>(r2);
  double2* y2 = reinterpret_cast<double2*>(y);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 rv = r22[i];
    double2 yv = y2[i];
    yv.x -= c * rv.x; yv.y -= c * rv.y;
    y2[i] = yv;
    acc = fma(yv.x, yv.x, acc); acc = fma(yv.y, yv.y, acc);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double yv = y[n - 1] - c * r2[n - 1];
    y[n - 1] = yv;
    acc = fma(yv, yv, acc);
  }
  acc = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) store_partial(rc.part + blockIdx.x, acc);
  if (last_block_ticket(rc.counter, rc.tickets, blockIdx.x)) {
    const double yy = sum_partials_agent(rc.base, rc.count, red);
    if (threadIdx.x == 0) {
      *rc.tot = yy;
      if (rc.tot2) *rc.tot2 = yy;                            // this rank's share, in the slot that rides on the next operand exchange
    }
    release_ticket_counter(rc.counter);
  }
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
minres_kd_kernel(int64_t n, MinresArgs a, const MinresState* __restrict__ Sin, MinresState* __restrict__ Sout,
                 const double* __restrict__ r2old, const double* __restrict__ w1, const double* __restrict__ w2,
                 double* __restrict__ w, double* __restrict__ x, MinresRed rd) {
  __shared__ double red[4];
  MinresState S = *Sin;
  if (S.done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *Sout = S;
    return;
  }
  const double bb = minres_yy(a);
  const double s_old = S.s;
  minres_advance(S, bb);      // scalar recurrences (every thread, identical)
  if (blockIdx.x == 0 && threadIdx.x == 0) *Sout = S;
  // ---- w = (v - oldeps*w1 - delta*w2)*denom ; x += phi*w ----
  const MinresKd kd{s_old, S.oldeps, S.delta, S.denom, S.phi};
  const int64_t n2 = n >> 1;
  const double2* r2 = reinterpret_cast<const double2*>(r2old);
  const double2* w12 = reinterpret_cast<const double2*>(w1);
  const double2* w22 = reinterpret_cast<const double2*>(w2);
  double2* wn2 = reinterpret_cast<double2*>(w);
  double2* x2 = reinterpret_cast<double2*>(x);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 rv = r2[i], a1 = w12[i], a2 = w22[i];
    double2 xv = x2[i], wn;
    kd.apply(rv.x, a1.x, a2.x, wn.x, xv.x);
    kd.apply(rv.y, a1.y, a2.y, wn.y, xv.y);
    wn2[i] = wn;
    x2[i] = xv;
    acc = fma(xv.x, xv.x, acc); acc = fma(xv.y, xv.y, acc);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double wn, xv = x[i];
    kd.apply(r2old[i], w1[i], w2[i], wn, xv);
    w[i] = wn; x[i] = xv;
    acc = fma(xv, xv, acc);
  }
  acc = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) store_partial(rd.part + blockIdx.x, acc);
  if (last_block_ticket(rd.counter, rd.tickets, blockIdx.x)) {
    const double xx = sum_partials_agent(rd.base, rd.count, red);
    if (threadIdx.x == 0) *rd.tot = xx;
    release_ticket_counter(rd.counter);
  }
}

// End-of-chunk evaluation of the stopping tests (what KA's prologue would do next).
__global__ void minres_check_kernel(MinresArgs a, MinresState* __restrict__ S0) {
  MinresState S = *S0;
  const double xx = (S.itn > 0 && !S.done) ? a.pD[0] : 0.0;
  minres_tests(S, xx, a);
  if (threadIdx.x == 0) *S0 = S;
}

extern "C" int hipeig_minres_x0(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign, const double* b, const double* x0,
                                double* x, double rtol, int maxiter, int* info, double out_stats[8]);

extern "C" int hipeig_minres(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign, const double* b,
                             double* x, double rtol, int maxiter, int* info, double out_stats[8]) {
  return hipeig_minres_x0(c, A, sigma, sign, b, nullptr, x, rtol, maxiter, info, out_stats);
}

// x0 != NULL: SciPy's minres(A, b, x0) - r1 = b - A x0, the iterate starts at x0 (so ||x|| in the stopping tests
// includes it), beta1 = ||r1||; beta1 == 0 returns x0, b == 0 returns b (scipy.sparse.linalg.minres, the x0 branch).
extern "C" int hipeig_minres_x0(hipeig_ctx* c, hipeig_csr* A, double sigma, double sign, const double* b, const double* x0,
                                double* x, double rtol, int maxiter, int* info, double out_stats[8]) {
  HIPEIG_REQUIRE(info != nullptr, "null info");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0, "sign must be +1 or -1");
  HIPEIG_REQUIRE(maxiter >= 1, "maxiter must be positive");
  HIPEIG_REQUIRE(b != x, "x must not alias b");
  const int64_t n = A->nrows;
  HIPEIG_REQUIRE(c->collectives || A->nrows == A->ncols, "the inner solve needs a square operator (or a row partition)");
  *info = 0;
  if (out_stats) memset(out_stats, 0, 8 * sizeof(double));
  double bb = 0.0;
  if (hipeig_dot(c, n, b, b, &bb)) return 1;
  if (bb == 0.0) return hipeig_vec_fill(c, x, n, 0.0);            // beta1 == 0 (or b == 0): the exact solution is x = 0 = b

  // workspace: R[3] (r1, r2, y rotate), W[3] (w1, w2, w rotate) and the iterate xw.  The iterate
  // lives in the workspace so that every kernel argument of a chunk is the same from solve to
  // solve, which is what lets the captured chunk be replayed; it is copied to x at the end.
  if (c->mr_ws_n < n) {
    if (c->mr_ws) HIPEIG_CHECK(hipFree(c->mr_ws));
    c->mr_ws = nullptr; c->mr_ws_n = 0;
    const int64_t npad = (n + 31) & ~(int64_t)31;
    HIPEIG_CHECK(hipMalloc((void**)&c->mr_ws, (size_t)npad * 7 * sizeof(double)));
    c->mr_ws_n = n;
  }
  const int64_t npad = (c->mr_ws_n + 31) & ~(int64_t)31;
  double* R[3] = {c->mr_ws, c->mr_ws + npad, c->mr_ws + 2 * npad};
  double* W[3] = {c->mr_ws + 3 * npad, c->mr_ws + 4 * npad, c->mr_ws + 5 * npad};
  double* xw = c->mr_ws + 6 * npad;
  HIPEIG_CHECK(hipMemsetAsync(W[0], 0, (size_t)npad * 4 * sizeof(double), c->stream));   // W[0..2] and xw
  if (x0) {
    if (hipeig_spmv_shift(c, A, sigma, sign, x0, R[0])) return 1;          // A x0 with the two roundings of the lambda
    if (hipeig_axpby(c, n, 1.0, b, -1.0, R[0])) return 1;                  // r1 = b - A x0
    if (hipeig_dot(c, n, R[0], R[0], &bb)) return 1;
    if (bb == 0.0) {                                                        // beta1 == 0: x0 is the exact solution
      if (x != x0) HIPEIG_CHECK(hipMemcpyAsync(x, x0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      return 0;
    }
    HIPEIG_CHECK(hipMemcpyAsync(xw, x0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  } else {
    HIPEIG_CHECK(hipMemcpyAsync(R[0], b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }

  MinresState* h = c->h_mr_state;
  minres_init_state(h, bb);
  MinresState* V = c->d_mr_state;
  HIPEIG_CHECK(hipMemcpyAsync(V, h, sizeof(MinresState), hipMemcpyHostToDevice, c->stream));
  // the pinned record is rewritten by the first chunk's copy-back; the upload above must have read it
  if (hipeig_sync_checked(c)) return 4;

  int variant = hipeig_csr_pick_variant(c, A);
  if (variant < 0) return 1;
  const bool fixed = (variant == 5);                    // TCOO-W with fixed-point accumulators: same structure as 4
  if (fixed) variant = 4;
  if (variant == 4 && hipeig_tcoow_reserve(c, A)) return 1;
  const CsrView view = hipeig_csr_view(A);
  const TcooView tview = (variant == 4) ? hipeig_tcoow_view(A) : hipeig_tcoo_view(A);
  if (variant == 4) {
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)minres_ka_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIPEIG_TCOOW_LDS_MAX));
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)minres_ka_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIPEIG_TCOOW_LDS_MAX));
  }
  if (variant == 3)
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)minres_ka_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)HIPEIG_TCOO_LDS_MAX));
  const int gA = hipeig_spmv_grid(A, variant);
  int per_thread = 24;       // measured (tools/experiments/mr_grid_sweep.sh): 4 / 8 / 16 / 24 / 32 / 64 elements per thread -> 0.138 / 0.136 / 0.134 / 0.1335 / 0.135 / 0.140 ms per iteration at N = 1e6, 2.32 / 2.24 / 2.23 / 2.21 / 2.22 / 2.20 at N = 1e7: the fixed cost of a workgroup (state record, ticket), not the grid cap, is what these kernels feel
  if (const char* e = getenv("HIPEIG_MR_PER_THREAD")) per_thread = atoi(e) > 0 ? atoi(e) : 24;     // tuning knob
  const int gE = grid_wide(n, per_thread);                   // element-wise kernels: their reductions finish in the last workgroup
  // partial-sum areas (HIPEIG_WIDE_PARTIALS apart) and the totals their last workgroups leave
  double* pA = c->d_partials;
  double* pC = c->d_partials + HIPEIG_WIDE_PARTIALS;
  double* pD = c->d_partials + 2 * HIPEIG_WIDE_PARTIALS;
  double* tot = c->d_scalars + 3072;                         // [0] <v,y>, [1] <x,x> (one all-reduce record), [2] <y,y>, [4..5] end-of-solve flush
  unsigned* cntA = c->d_counters + 0;
  unsigned* cntC = c->d_counters + HIPEIG_TICKET_WORDS;
  unsigned* cntD = c->d_counters + 2 * HIPEIG_TICKET_WORDS;
  const bool dist = c->collectives != 0;
  MinresArgs a;
  a.sigma = sigma; a.sign = sign; a.rtol = rtol; a.maxiter = maxiter;
  const bool split = (variant == 4) && A->w_csplit > 1;      // raw slabs + combine launch
  const int nsweepA = split ? 1 : (variant == 4) ? (A->w_nunits + gA - 1) / gA
                    : (variant == 3) ? (A->t_nunits + gA * 4 - 1) / (gA * 4) : 1;
  HIPEIG_REQUIRE(nsweepA * gA <= HIPEIG_WIDE_PARTIALS, "too many sweeps for the partial-sum buffer");
  // combine launch of a column-split sweep (slab of a many-GPU run): rows per thread.  It carries the whole KA epilogue with
  // the riding KD (10 streams + the slabs), not a 2-stream update: 24 / 12 / 8 / 4 / 2 rows per thread -> 0.177 / 0.160 /
  // 0.156 / 0.160 / 0.166 ms per iteration (N = 1e6, 5 splits forced; profiles/r04_minres_combine_grid.txt)
  int ct = 8;
  if (const char* e = getenv("HIPEIG_MR_COMBINE_PER_THREAD")) ct = atoi(e) > 0 ? atoi(e) : 8;   // tuning knob
  const int gC = grid_wide(n, ct);
  const int nPA = split ? gC : gA * nsweepA;                 // partial <v,y> sums one iteration leaves in pA
  a.pA = tot + 0; a.nA = 1;
  a.pD = tot + 1; a.nD = 1;
  a.pC = tot + 2; a.nC = 1; a.sC = 0;
  HIPEIG_CHECK(hipMemsetAsync(tot, 0, 8 * sizeof(double), c->stream));
  c->mr_collectives = 0;
  // Row-partitioned run: <y,y> arrives as one share per rank in the scalar slots of the operand exchange (the pointer
  // is set per sweep: the direct backend alternates between two buffers)
  const int64_t slot0 = dist ? A->gl.slot(0) : 0;
  const int64_t slot_stride = dist ? A->gl.cstride(A->gl.nchunks - 1) : 0;

  // KD(k) rides on the sweep of KA(k+1) (two kernels per iteration, see the header); a partitioned run always does.
  const char* fk_env = getenv("HIPEIG_MINRES_FUSE_KD");
  const bool fuse_kd = dist || !(fk_env && atoi(fk_env) == 0);

  const MinresRed redC{pC, pC, gE, (unsigned)gE, cntC, tot + 2, nullptr};
  const MinresRed redD{pD, pD, gE, (unsigned)gE, cntD, tot + 1, nullptr};

  // The operator sweep of iteration k (all variants).  `kd`: the pending KD of the previous iteration (do_kd = 0: none),
  // `Sin`: the record the sweep starts from.
  auto enqueue_ka = [&](double* r2, double* r1, double* yb, const MinresState* Sin, MinresKdArgs kd) -> int {
    const double* xg = nullptr;
    TcooView tv = tview;
    MinresArgs al = a;
    const int zero_xx = dist ? 1 : 0;
#define KA_LAUNCH(VAR, FIX, GRID, THREADS, LDS, TV, OFF)                                                            \
    do {                                                                                                            \
      MinresKdArgs kl = kd;                                                                                         \
      kl.red = MinresRed{pD + (OFF), pD, nPA, (unsigned)nPA, cntD, tot + 1, nullptr};              \
      const MinresRed ra{pA + (OFF), pA, nPA, (unsigned)nPA, cntA, tot + 0, nullptr};              \
      hipLaunchKernelGGL((minres_ka_kernel<VAR, FIX>), dim3(GRID), dim3(THREADS), LDS, c->stream, view, TV, xg, al, Sin, V + 1, r2, r1, yb, ra, kl, zero_xx); \
    } while (0)
    if (variant == 4) {
      bool has_last = true;
      int ncombine = 0;                                           // own windows under the exchange, chunk by chunk behind it
      if (hipeig_tcoow_run_plan(c, A, r2, fixed ? 1 : 0, &tv, &has_last, &xg, &ncombine)) return 4;
      if (dist) { ++c->mr_collectives; al.pC = xg + slot0; al.nC = c->nranks; al.sC = slot_stride; }
      if (has_last) {
        for (int sw = 0; sw < nsweepA; ++sw) {
          tv.unit_begin = sw * gA;
          if (fixed) KA_LAUNCH(4, 1, gA, TCOOW_THREADS, hipeig_tcoow_lds_bytes(A), tv, sw * gA);
          else KA_LAUNCH(4, 0, gA, TCOOW_THREADS, hipeig_tcoow_lds_bytes(A), tv, sw * gA);
        }
      }
      if (ncombine) {
        TcooView tc = tv;
        tc.raw_out = c->ytmp;
        tc.part_base = ncombine;                                  // number of slabs to add
        KA_LAUNCH(5, 0, gC, HIPEIG_BLOCK, 0, tc, 0);
      }
      hipeig_phase_mark(c, 3);
      return 0;
    }
    if (hipeig_allgather_x(c, A->gl, r2, n, &xg)) return 4;
    if (dist) { ++c->mr_collectives; al.pC = xg + slot0; al.nC = c->nranks; al.sC = slot_stride; }
    if (variant == 3) {
      for (int sw = 0; sw < nsweepA; ++sw) {       // one launch per sweep; partials side by side
        tv.unit_begin = sw * gA * 4;
        KA_LAUNCH(3, 0, gA, HIPEIG_BLOCK, hipeig_tcoo_lds_bytes(A), tv, sw * gA);
      }
    } else if (variant == 1) {
      KA_LAUNCH(1, 0, gA, HIPEIG_BLOCK, 0, tview, 0);
    } else {
      KA_LAUNCH(2, 0, gA, HIPEIG_BLOCK, 0, tview, 0);
    }
#undef KA_LAUNCH
    return 0;
  };

  const MinresKdArgs no_kd{0, nullptr, nullptr, nullptr, nullptr, MinresRed{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr}};
  // One iteration's launches on the compute stream (buffer roles rotate with period 3).  `first`: nothing is pending
  // from the iteration before (one GPU: the first iteration of a chunk - the chunk before ended with a stand-alone KD;
  // partitioned run: only the very first iteration, the pending KD is carried over the chunk boundary).
  auto enqueue_iteration = [&](int k, bool first) -> int {
    double* r2 = R[k % 3];
    double* yb = R[(k + 1) % 3];
    double* r1 = R[(k + 2) % 3];
    double* wn = W[k % 3];
    double* w1 = W[(k + 1) % 3];
    double* w2 = W[(k + 2) % 3];
    if (fuse_kd) {
      // KD(k-1): w = W[(k-1)%3] from w1 = W[k%3], w2 = W[(k+1)%3] and r2 of that iteration = this one's r1
      const bool pending = !first;
      MinresKdArgs kd = no_kd;
      kd.do_kd = pending ? 1 : 0; kd.w1 = wn; kd.w2 = w1; kd.w = w2; kd.x = xw;
      if (enqueue_ka(r2, r1, yb, pending ? V + 2 : V + 0, kd)) return 4;
      if (dist) {                                                // (<v,y>, <x,x>) in one all-reduce
        if (hipeig_allreduce_sum(c, tot, 2)) return 4;
        ++c->mr_collectives;
      }
      MinresRed rc = redC;
      if (dist) rc.tot2 = hipeig_gather_slot(c, A->gl);          // travels with the NEXT exchange
      hipLaunchKernelGGL(minres_kc_kernel, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, V + 1, V + 2, r2, yb, rc, pending ? 1 : 0);
      return 0;
    }
    if (enqueue_ka(r2, r1, yb, V + 0, no_kd)) return 4;
    hipLaunchKernelGGL(minres_kc_kernel, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, V + 1, V + 2, r2, yb, redC, 0);
    hipLaunchKernelGGL(minres_kd_kernel, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, V + 2, V + 0, r2, w1, w2, wn, xw, redD);
    return 0;
  };
  // The KD of iteration k as its own kernel (nothing follows to ride on): the end of a chunk on one GPU, the end of the
  // solve on a partitioned run (whose <y,y> shares have not travelled: one extra all-reduce).
  auto enqueue_last_kd = [&](int k) -> int {
    if (!fuse_kd) return 0;
    MinresArgs al = a;
    if (dist) {
      if (hipeig_allreduce_sum(c, tot + 2, 1)) return 4;
      ++c->mr_collectives;
    }
    hipLaunchKernelGGL(minres_kd_kernel, dim3(gE), dim3(HIPEIG_BLOCK), 0, c->stream, n, al, V + 2, V + 0, R[k % 3], W[(k + 1) % 3], W[(k + 2) % 3], W[k % 3], xw, redD);
    return 0;
  };
  // diagnostic knobs for the rocprofv3-inside-capture question (profiles/r02_hipgraph_under_rocprofv3.txt):
  // HIPEIG_GRAPH_COPY=0 keeps the device-to-host copy of the state record OUT of the captured graph,
  // HIPEIG_GRAPH_MODE=0/2 captures in global / relaxed instead of thread-local mode
  const char* gc_env = getenv("HIPEIG_GRAPH_COPY");
  const bool graph_copy_node = !(gc_env && atoi(gc_env) == 0);
  const char* gm_env = getenv("HIPEIG_GRAPH_MODE");
  const int graph_mode = gm_env ? atoi(gm_env) : 1;
  bool capturing = false;
  const bool gtrace = getenv("HIPEIG_GRAPH_TRACE") != nullptr;       // stderr markers around the graph API calls
#define GTRACE(msg) do { if (gtrace) { fprintf(stderr, "[hipeig graph] %s\n", msg); fflush(stderr); } } while (0)
  auto enqueue_check = [&]() -> int {
    if (dist) {
      if (hipeig_allreduce_sum(c, tot + 1, 1)) return 4;      // the stand-alone KD's <x,x>
      ++c->mr_collectives;
    }
    hipLaunchKernelGGL(minres_check_kernel, dim3(1), dim3(64), 0, c->stream, a, V + 0);
    if (!capturing || graph_copy_node)
      HIPEIG_CHECK(hipMemcpyAsync(h, V, sizeof(MinresState), hipMemcpyDeviceToHost, c->stream));
    return 0;
  };

  if (c->use_graph && !dist) {
    // Single-GPU: a chunk of 18 iterations (a multiple of the rotation period, so every chunk has
    // the same arguments) + the stop check + the copy-back is captured once as a hipGraph and
    // replayed; iterations past maxiter are no-ops (KA's prologue raises istop = 6).  The graph is
    // kept across solves while nothing it has baked in changes.
    struct GraphKey {
      const void *A, *rowptr, *col, *val, *tidx, *widx, *ws, *state, *parts;
      int64_t n, nnz, variant, gA, nsweep, units, lds, maxiter, csplit;
      double sigma, sign, rtol;
    } key;
    memset(&key, 0, sizeof(key));
    key.A = A; key.rowptr = A->d_rowptr; key.col = A->d_col; key.val = A->d_val;
    key.tidx = A->t_idx; key.widx = A->w_idx; key.ws = c->mr_ws; key.state = V; key.parts = c->ytmp;
    key.n = n; key.nnz = A->nnz; key.variant = variant; key.gA = gA; key.nsweep = nsweepA;
    key.units = (variant == 4) ? A->w_nunits : (variant == 3) ? A->t_nunits : A->n_row_blocks;
    key.lds = (variant == 4) ? (int64_t)hipeig_tcoow_lds_bytes(A) : (variant == 3) ? (int64_t)hipeig_tcoo_lds_bytes(A) : 0;
    key.csplit = (variant == 4) ? A->w_csplit + (fixed ? 1000 : 0) : 0;
    key.maxiter = maxiter; key.sigma = sigma; key.sign = sign; key.rtol = rtol;
    key.csplit += fuse_kd ? 100000 : 0;
    const int gchunk = 18;
    if (!c->mr_graph || c->mr_graph_key_bytes != sizeof(key) || memcmp(c->mr_graph_key, &key, sizeof(key)) != 0) {
      if (c->mr_graph) { GTRACE("hipGraphExecDestroy"); hipGraphExecDestroy(c->mr_graph); c->mr_graph = nullptr; }
      hipGraph_t g = nullptr;
      GTRACE("hipStreamBeginCapture");
      HIPEIG_CHECK(hipStreamBeginCapture(c->stream, graph_mode == 0 ? hipStreamCaptureModeGlobal
                                                    : graph_mode == 2 ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
      capturing = true;
      int rc = 0;
      for (int k = 0; k < gchunk && rc == 0; ++k) rc = enqueue_iteration(k, k == 0);
      if (rc == 0) rc = enqueue_last_kd(gchunk - 1);
      if (rc == 0) rc = enqueue_check();
      GTRACE("hipStreamEndCapture");
      const hipError_t ce = hipStreamEndCapture(c->stream, &g);
      capturing = false;
      GTRACE("hipGraphInstantiate");
      if (rc) { if (g) hipGraphDestroy(g); return rc; }
      HIPEIG_CHECK(ce);
      const hipError_t ie = hipGraphInstantiate(&c->mr_graph, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      HIPEIG_CHECK(ie);
      if (!c->mr_graph_key) c->mr_graph_key = malloc(sizeof(key));
      HIPEIG_REQUIRE(c->mr_graph_key != nullptr, "out of host memory");
      memcpy(c->mr_graph_key, &key, sizeof(key));
      c->mr_graph_key_bytes = sizeof(key);
    }
    for (int k = 0; k < maxiter; k += gchunk) {
      GTRACE("hipGraphLaunch");
      HIPEIG_CHECK(hipGraphLaunch(c->mr_graph, c->stream));
      GTRACE("hipGraphLaunch returned");
      if (!graph_copy_node) HIPEIG_CHECK(hipMemcpyAsync(h, V, sizeof(MinresState), hipMemcpyDeviceToHost, c->stream));
      if (hipeig_sync_checked(c)) return 4;
      if (h->done) break;
    }
  } else {
    // iterations between two looks at the state record (check kernel + copy-back + host sync: ~70 us at N = 1e6, where
    // an iteration takes 140 us).  Kernels launched past the stopping iteration return at once (~5 us each), so a longer
    // chunk wastes at most that.  One GPU: a chunk ends with the stand-alone KD and the check kernel.  Partitioned run:
    // no flush at a chunk boundary (it would cost two extra collectives) - the host looks at the record KC left, which
    // holds the tests of the iteration before, and the pending KD is carried into the next chunk; only when the
    // iteration limit is reached are the last KD and the last tests run on their own.
    const int chunk = dist ? 16 : 32;
    int k = 0;
    while (k < maxiter) {
      const int kend = (k + chunk < maxiter) ? k + chunk : maxiter;
      const int kfirst = k;
      for (; k < kend; ++k) {
        const int rc = enqueue_iteration(k, dist ? (k == 0) : (k == kfirst));
        if (rc) return rc;
      }
      HIPEIG_CHECK(hipGetLastError());
      if (dist && kend < maxiter) {
        HIPEIG_CHECK(hipMemcpyAsync(h, V + 2, sizeof(MinresState), hipMemcpyDeviceToHost, c->stream));
      } else {
        if (enqueue_last_kd(kend - 1)) return 1;
        HIPEIG_CHECK(hipGetLastError());
        if (enqueue_check()) return 1;
      }
      if (hipeig_sync_checked(c)) return 4;
      if (h->done) break;
    }
  }
  HIPEIG_REQUIRE(h->done, "MINRES left the iteration loop without a stop code");
  HIPEIG_CHECK(hipMemcpyAsync(x, xw, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  *info = (h->istop == 6) ? maxiter : 0;
  if (out_stats) {
    out_stats[0] = h->itn; out_stats[1] = h->istop; out_stats[2] = h->rnorm; out_stats[3] = h->Anorm;
    out_stats[4] = h->ynorm; out_stats[5] = h->test1; out_stats[6] = h->test2; out_stats[7] = h->Acond;
  }
  return 0;
}
