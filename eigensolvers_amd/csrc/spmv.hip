// Sparse operator: creation, layout preparation and the y = Hx / y = sign*(sigma*x - Hx)
// entry points.  Kernel bodies live in spmv_device.h.
#include <vector>
#include "spmv_device.h"

int hipeig_csr_pick_variant(hipeig_ctx* c, hipeig_csr* A);
size_t hipeig_tcoo_lds_bytes(const hipeig_csr* A);
size_t hipeig_tcoow_lds_bytes(const hipeig_csr* A);
static int build_tcoow_layout(hipeig_ctx* c, hipeig_csr* A, int pair);

// y[r] = a_self*xl[r] + a_sum*sum  (a_self = 0, a_sum = 1: plain product;
// a_self = sign*sigma, a_sum = -sign: the shifted operator of numpyVector.py:152/154).
// Two roundings like the reference's sigma*x - H@x (no contraction into an FMA).
struct AxpyEpilogue {
  double a_self, a_sum;
  const double* __restrict__ xl;
  double* __restrict__ y;
  __device__ __forceinline__ void row(int64_t r, double sum, double& acc) const {
    const double t = (a_self == 0.0) ? 0.0 : mul_rn(a_self, xl[r]);
    y[r] = add_rn(t, mul_rn(a_sum, sum));
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

__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmv_tcoo_kernel(TcooView T, const double* __restrict__ x, AxpyEpilogue epi) {
  extern __shared__ double tcoo_lds[];
  double acc = 0.0;
  tcoo_sweep(T, x, epi, acc, tcoo_lds);
}

template <int FIXED>
__global__ void __launch_bounds__(TCOOW_THREADS)
spmv_tcoow_kernel(TcooView T, const double* __restrict__ x, AxpyEpilogue epi) {
  extern __shared__ double tcoo_lds[];
  __shared__ double red16[16];
  double acc = 0.0;
  tcoo_wg_sweep<AxpyEpilogue, FIXED>(T, x, epi, acc, tcoo_lds, red16);
}

// Complex operand, real operator, complex shift (the contour solves of feast.py:83-90 through GCROT): with
// x = xr + i xi and z = zr + i zi,  y = sign*(z*x - H x):
//   yr = sign*(zr*xr - H xr) - sign*zi*xi,   yi = sign*(zr*xi - H xi) + sign*zi*xr.
// ar = sign*zr, ai = sign*zi, as = -sign (ar = ai = 0, as = 1: the plain product).  The real part of the
// shift and the operator sum are rounded separately, like the reference's sigma*x - H@x.
struct PairEpilogue {
  double ar, ai, as;
  const double* __restrict__ xr;
  const double* __restrict__ xi;
  double* __restrict__ yr;
  double* __restrict__ yi;
  __device__ __forceinline__ void row2(int64_t r, double sr, double si, double& acc) const {
    const double vr = xr[r], vi = xi[r];
    const double tr = add_rn(mul_rn(ar, vr), mul_rn(as, sr));
    const double ti = add_rn(mul_rn(ar, vi), mul_rn(as, si));
    yr[r] = fma(-ai, vi, tr);
    yi[r] = fma(ai, vr, ti);
  }
};

__global__ void __launch_bounds__(TCOOW_THREADS)
spmv_tcoow_pair_kernel(TcooView T, const double* __restrict__ xpair, PairEpilogue epi) {
  extern __shared__ double tcoo_lds[];
  double acc = 0.0;
  tcoo_wg_sweep<PairEpilogue, 0, 1>(T, xpair, epi, acc, tcoo_lds, nullptr);
}

// xp[i] = (xr[i], xi[i]): the interleaved operand of the pair sweep
__global__ void __launch_bounds__(HIPEIG_BLOCK)
pair_pack_kernel(int64_t n, const double* __restrict__ xr, const double* __restrict__ xi, double2* __restrict__ xp) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) xp[i] = make_double2(xr[i], xi[i]);
}

__global__ void __launch_bounds__(HIPEIG_BLOCK) absmax_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ partials) {
  __shared__ double lds[16];
  double m = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = fabs(x[i]);
    m = (v > m || v != v) ? v : m;                     // NaN sticks
  }
  m = block_max_all(m, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = m;
}

// max_i sum_j |a_ij| over the local rows (one wavefront per row), as per-workgroup maxima
__global__ void __launch_bounds__(HIPEIG_BLOCK)
rowabs_max_kernel(const int32_t* __restrict__ rowptr, const double* __restrict__ val, int64_t nrows, double* __restrict__ partials) {
  __shared__ double lds[16];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double m = 0.0;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    double a = 0.0;
    for (int p = rowptr[r] + lane; p < rowptr[r + 1]; p += 64) a += fabs(val[p]);
    a = wave_reduce_sum(a);
    a = __shfl(a, 0, 64);
    m = fmax(m, a);
  }
  m = block_max_all(m, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = m;
}

// Operand maximum for the fixed-point sweep: per-workgroup maxima into the context's fx area.
#define HIPEIG_FX_AREA (4 * HIPEIG_WIDE_PARTIALS)      // behind the three partial-sum areas of the MINRES kernels
int hipeig_fixed_prepare(hipeig_ctx* c, const hipeig_csr* A, const double* xg, TcooView* t) {
  const int g = grid_for(A->gather_len, 8);
  double* area = c->d_partials + HIPEIG_FX_AREA;
  hipLaunchKernelGGL(absmax_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, xg, A->gather_len, area);
  t->fx_xmax = area; t->fx_count = g; t->fx_bound = A->absrow_max;
  return 0;
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
spmv_tcoow_combine_kernel(const double* __restrict__ parts, int nparts, int64_t stride, int64_t nrows, AxpyEpilogue epi) {
  double acc = 0.0;
  tcoow_combine_sweep(parts, nparts, stride, nrows, epi, acc);
}

TcooView hipeig_tcoow_view(const hipeig_csr* A) {
  TcooView t;
  t.idx = A->w_idx; t.val = A->w_val; t.off = A->w_off;
  t.nunits = A->w_nunits; t.nwin = A->w_nwin; t.wbits = A->w_wbits; t.rw = A->w_rw;
  t.unit_begin = 0;
  t.prefetch = 0;
  t.ablate = 0;
  if (const char* e = getenv("HIPEIG_TCOO_ABLATE")) t.ablate = atoi(e);   // timing experiments (wrong results)
  t.nrows = A->nrows;
  t.gather_len = A->gather_len;
  t.nrun = 0; t.yinit = nullptr; t.raw_out = nullptr;
  t.csplit = 1; t.part_base = 0; t.part_stride = 0;
  t.fx_xmax = nullptr; t.fx_count = 0; t.fx_bound = 0.0;
  return t;
}

TcooView hipeig_tcoo_view(const hipeig_csr* A) {
  TcooView t;
  t.idx = A->t_idx; t.val = A->t_val; t.off = A->t_off;
  t.nunits = A->t_nunits; t.nwin = A->t_nwin; t.wbits = A->t_wbits; t.rw = A->t_rw;
  t.unit_begin = 0;
  t.prefetch = A->t_prefetch;
  t.ablate = 0;
  if (const char* e = getenv("HIPEIG_TCOO_ABLATE")) t.ablate = atoi(e);   // timing experiments (wrong results)
  t.nrows = A->nrows;
  t.gather_len = A->gather_len;
  t.nrun = 0; t.yinit = nullptr; t.raw_out = nullptr;
  t.csplit = 1; t.part_base = 0; t.part_stride = 0;
  t.fx_xmax = nullptr; t.fx_count = 0; t.fx_bound = 0.0;
  return t;
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
  if (variant == 4) {
    g = A->w_wgs_per_sweep;                          // one unit per workgroup, one workgroup per CU
    if (g > A->w_nunits) g = A->w_nunits;
    if (A->w_csplit > 1) g = (int64_t)A->w_nunits * A->w_csplit;      // split mode: always a single launch
  } else if (variant == 3) {
    g = A->t_wgs_per_sweep;                          // workgroups of ONE sweep (4 units each)
    const int64_t need = (A->t_nunits + 3) / 4;
    if (g > need) g = need;
  } else if (variant == 1) {
    const int64_t groups_per_block = HIPEIG_BLOCK / A->lanes_per_row;
    g = (A->nrows + groups_per_block - 1) / groups_per_block;
  } else {
    g = A->n_row_blocks;
  }
  if (g < 1) g = 1;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return (int)g;
}

// Partial-sum slabs of the split / overlapped sweeps (context buffer, grown on demand).
static int tcoow_ensure_parts(hipeig_ctx* c, int64_t doubles) {
  if (c->ytmp_n >= doubles) return 0;
  if (c->ytmp) { if (hipFree(c->ytmp) != hipSuccess) return -1; }
  c->ytmp = nullptr; c->ytmp_n = 0;
  if (hipMalloc((void**)&c->ytmp, (size_t)doubles * sizeof(double)) != hipSuccess) {
    hipeig_set_error("out of device memory for the partial-sum slabs");
    return -1;
  }
  c->ytmp_n = doubles;
  return 0;
}

int64_t hipeig_tcoow_part_stride(const hipeig_csr* A) { return (A->nrows + 63) & ~(int64_t)63; }

void hipeig_phase_mark(hipeig_ctx* c, int k);
const double* hipeig_gathered(hipeig_ctx* c);

// The column windows of a partitioned TCOO-W operator by what they need:
// set 0 - windows that lie entirely inside this rank's own column ranges (one range per chunk of the gathered layout):
//         they can be swept as soon as the slice has been copied into the gathered buffer, while the exchange runs;
// set 1 + k - the other windows whose last column belongs to chunk k: complete once chunk k has arrived.
// Every set is at most two runs of consecutive windows (a chunk's windows minus the local run inside them).
struct WindowSets {
  int nset;                        // 1 + nchunks
  int nrun[1 + HIPEIG_GATHER_MAX_CHUNKS];
  int lo[1 + HIPEIG_GATHER_MAX_CHUNKS][4], hi[1 + HIPEIG_GATHER_MAX_CHUNKS][4];
};

static void tcoow_window_sets(const hipeig_ctx* c, const hipeig_csr* A, WindowSets* ws) {
  const GatherLayout& gl = A->gl;
  const int wb = A->w_wbits, nwin = A->w_nwin;
  const int64_t W = (int64_t)1 << wb;
  memset(ws, 0, sizeof(*ws));
  ws->nset = 1 + gl.nchunks;
  int prev_end = 0;                                   // first window not yet given to a chunk
  for (int k = 0; k < gl.nchunks; ++k) {
    // windows whose last position lies in chunk k: [prev_end, wend)
    int wend = (k == gl.nchunks - 1) ? nwin : (int)(gl.cbase[k + 1] >> wb);      // window holding cbase[k+1] straddles: it waits for k+1
    if (wend > nwin) wend = nwin;
    if (wend < prev_end) wend = prev_end;
    // own range inside chunk k, and the windows entirely inside it
    const int64_t rows_lo = (int64_t)k * gl.h;
    int64_t rows = A->nrows - rows_lo;
    if (rows > gl.h) rows = gl.h;
    int l0 = 0, l1 = 0;
    if (rows > 0) {
      const int64_t p0 = gl.cbase[k] + (int64_t)c->rank * gl.cstride(k), p1 = p0 + rows;
      l0 = (int)((p0 + W - 1) >> wb); l1 = (int)(p1 >> wb);
      if (l0 < prev_end) l0 = prev_end;
      if (l1 > wend) l1 = wend;
      if (l1 < l0) l1 = l0;
    }
    if (l1 > l0) {
      const int q = ws->nrun[0]++;
      ws->lo[0][q] = l0; ws->hi[0][q] = l1;
      if (l0 > prev_end) { const int r = ws->nrun[1 + k]++; ws->lo[1 + k][r] = prev_end; ws->hi[1 + k][r] = l0; }
      if (wend > l1) { const int r = ws->nrun[1 + k]++; ws->lo[1 + k][r] = l1; ws->hi[1 + k][r] = wend; }
    } else if (wend > prev_end) {
      const int r = ws->nrun[1 + k]++; ws->lo[1 + k][r] = prev_end; ws->hi[1 + k][r] = wend;
    }
    prev_end = wend;
  }
}

// Allocate the slabs one product of A can need, so that no allocation happens later (a hipGraph
// capture must not allocate).
int hipeig_tcoow_reserve(hipeig_ctx* c, const hipeig_csr* A) {
  if (!A->w_idx) return 0;
  const int64_t stride = hipeig_tcoow_part_stride(A);
  const int nl = c->collectives ? 1 + A->gl.nchunks : 1;
  const int64_t need = (A->w_csplit > 1) ? (int64_t)nl * A->w_csplit * stride : (c->collectives ? A->nrows : 0);
  return need > 0 ? tcoow_ensure_parts(c, need) : 0;
}

// Everything one TCOO-W product does before its epilogue-bearing launch, as a list of sweep launches:
//   launch i covers the windows of tv[i]; wait[i] >= 0: the compute stream must first wait for that chunk of the
//   operand exchange (hipeig_allgather_x_wait_chunk).
// Without a communicator (or without the overlap) that is ONE launch over all windows after a plain exchange.  With it,
// the exchange is started first and the launches are: this rank's own windows (under the exchange), then per chunk the
// windows that chunk completes.  Accumulators: with column splits every launch stores its own raw
// slabs (part_base = launch * csplit) and *ncombine > 0 tells the caller to add that many slabs in a combine kernel
// that carries the epilogue; without, the launches hand their sums on through ctx->ytmp (raw_out -> yinit) and the
// LAST launch carries the epilogue.
struct TcoowPlan {
  int nlaunch;
  TcooView tv[2 + HIPEIG_GATHER_MAX_CHUNKS];
  int wait[2 + HIPEIG_GATHER_MAX_CHUNKS];
  int ncombine;
  const double* xg;
  bool overlapped;
};

int hipeig_tcoow_plan(hipeig_ctx* c, hipeig_csr* A, const double* x_local, TcoowPlan* P) {
  const int cs = A->w_csplit;
  const int64_t stride = hipeig_tcoow_part_stride(A);
  P->nlaunch = 0; P->ncombine = 0; P->xg = nullptr; P->overlapped = false;
  WindowSets ws;
  bool split_sets = c->collectives && c->overlap && A->w_idx && A->col_stride > 0 && A->last_variant != 5;   // the fixed-point form needs max|x| of the whole operand first
  if (split_sets) {
    tcoow_window_sets(c, A, &ws);
    int nonempty = 0;
    for (int q = 0; q < ws.nset; ++q) nonempty += ws.nrun[q] > 0;
    if (ws.nrun[0] == 0 && A->gl.nchunks == 1) split_sets = false;        // nothing to hide the exchange behind
    if (nonempty <= 1 && ws.nrun[0] == 0) split_sets = false;
  }
  if (!split_sets) {
    if (hipeig_allgather_x(c, A->gl, x_local, A->nrows, &P->xg)) return 4;
    TcooView t = hipeig_tcoow_view(A);
    if (cs > 1) {
      if (tcoow_ensure_parts(c, cs * stride)) return 4;
      t.csplit = cs; t.part_base = 0; t.part_stride = stride; t.raw_out = c->ytmp;
      P->ncombine = cs;
    }
    P->tv[0] = t; P->wait[0] = -1; P->nlaunch = 1;
    return 0;
  }
  int nl = 0;
  for (int q = 0; q < ws.nset; ++q) nl += ws.nrun[q] > 0;
  if (tcoow_ensure_parts(c, cs > 1 ? (int64_t)nl * cs * stride : A->nrows)) return 4;
  if (hipeig_allgather_x_begin(c, A->gl, x_local, A->nrows)) return 4;
  P->overlapped = true;
  int li = 0;
  for (int q = 0; q < ws.nset; ++q) {
    if (ws.nrun[q] == 0) continue;
    TcooView t = hipeig_tcoow_view(A);
    t.nrun = ws.nrun[q];
    for (int r = 0; r < ws.nrun[q]; ++r) { t.run_lo[r] = ws.lo[q][r]; t.run_hi[r] = ws.hi[q][r]; }
    t.csplit = cs; t.part_stride = stride;
    if (cs > 1) { t.yinit = nullptr; t.raw_out = c->ytmp; t.part_base = li * cs; }
    else {
      t.part_base = 0;
      t.yinit = (li > 0) ? c->ytmp : nullptr;
      t.raw_out = (li + 1 < nl) ? c->ytmp : nullptr;
    }
    P->tv[li] = t;
    P->wait[li] = (q == 0) ? -1 : q - 1;
    ++li;
  }
  P->nlaunch = nl;
  P->ncombine = (cs > 1) ? nl * cs : 0;
  // a set without windows still has to be waited for by the launch that follows it; the last launch waits for the
  // last chunk in any case (it - or the combine kernel behind it - may read the scalar slots riding on that chunk)
  P->wait[nl - 1] = A->gl.nchunks - 1;
  P->xg = nullptr;                                           // set by the caller after the first wait: hipeig_gathered()
  return 0;
}

// Raw (epilogue-free) launch of plan entry i, all its sweeps.
static void tcoow_launch_raw(hipeig_ctx* c, hipeig_csr* A, TcooView t, const double* xg, int g, int fixed) {
  AxpyEpilogue none{0.0, 0.0, nullptr, nullptr};
  for (int ub = 0; ub < A->w_nunits * t.csplit; ub += g) {
    t.unit_begin = ub;
    if (fixed) hipLaunchKernelGGL(spmv_tcoow_kernel<1>, dim3(g), dim3(TCOOW_THREADS), hipeig_tcoow_lds_bytes(A), c->stream, t, xg, none);
    else hipLaunchKernelGGL(spmv_tcoow_kernel<0>, dim3(g), dim3(TCOOW_THREADS), hipeig_tcoow_lds_bytes(A), c->stream, t, xg, none);
  }
}

// Runs every launch of the plan that carries no epilogue; returns through *last the view the caller must launch with its
// own epilogue kernel (nullptr when a combine launch carries it: *ncombine slabs in ctx->ytmp), *xg the operand.
int hipeig_tcoow_run_plan(hipeig_ctx* c, hipeig_csr* A, const double* x_local, int fixed, TcooView* last, bool* has_last,
                          const double** xg, int* ncombine) {
  TcoowPlan P;
  if (hipeig_tcoow_plan(c, A, x_local, &P)) return 4;
  const int g = hipeig_spmv_grid(A, 4);
  const double* x = P.overlapped ? hipeig_gathered(c) : P.xg;
  const int nraw = P.ncombine ? P.nlaunch : P.nlaunch - 1;
  if (fixed) {                                              // never overlapped (hipeig_tcoow_plan): max|x| of the whole operand first
    HIPEIG_REQUIRE(!P.overlapped && P.nlaunch == 1, "the fixed-point sweep takes the plain exchange");
    if (hipeig_fixed_prepare(c, A, x, &P.tv[0])) return 4;
  }
  bool marked = false;
  for (int i = 0; i < P.nlaunch; ++i) {
    if (P.wait[i] >= 0) {
      if (hipeig_allgather_x_wait_chunk(c, A->gl, P.wait[i])) return 4;
      if (P.overlapped && !marked) { hipeig_phase_mark(c, 2); marked = true; }
    }
    if (i < nraw) {
      tcoow_launch_raw(c, A, P.tv[i], x, g, fixed);
      if (P.overlapped && i == 0) hipeig_phase_mark(c, 1);
    }
  }
  if (hipGetLastError() != hipSuccess) { hipeig_set_error("sweep launch failed"); return 4; }
  *has_last = (P.ncombine == 0);
  *last = P.tv[P.nlaunch - 1];
  *xg = x;
  *ncombine = P.ncombine;
  return 0;
}

static int launch_spmv(hipeig_ctx* c, hipeig_csr* A, double a_self, double a_sum,
                       const double* x, double* y) {
  if (A->nrows == 0) return 0;
  int variant = hipeig_csr_pick_variant(c, A);
  if (variant < 0) return 1;
  // shift term: x restricted to this operator's rows.  Partitioned run: x IS that slice; a row
  // slab applied to a full-length operand (single process): the slice starts at row_offset.
  AxpyEpilogue epi{a_self, a_sum, c->collectives ? x : x + A->row_offset, y};
  const double* xg = nullptr;
  const bool fixed = (variant == 5);                 // TCOO-W with fixed-point accumulators
  if (fixed) variant = 4;
  const int g = hipeig_spmv_grid(A, variant);
  if (variant == 4) {
    TcooView t;
    bool has_last = true;
    int ncombine = 0;
    if (hipeig_tcoow_run_plan(c, A, x, fixed ? 1 : 0, &t, &has_last, &xg, &ncombine)) return 4;
    if (has_last) {
      for (int ub = 0; ub < A->w_nunits * t.csplit; ub += g) {           // one launch per sweep
        t.unit_begin = ub;
        if (fixed) hipLaunchKernelGGL(spmv_tcoow_kernel<1>, dim3(g), dim3(TCOOW_THREADS), hipeig_tcoow_lds_bytes(A), c->stream, t, xg, epi);
        else hipLaunchKernelGGL(spmv_tcoow_kernel<0>, dim3(g), dim3(TCOOW_THREADS), hipeig_tcoow_lds_bytes(A), c->stream, t, xg, epi);
      }
    }
    if (ncombine)
      hipLaunchKernelGGL(spmv_tcoow_combine_kernel, dim3(grid_for(A->nrows, 2)), dim3(HIPEIG_BLOCK), 0, c->stream,
                         c->ytmp, ncombine, t.part_stride, A->nrows, epi);
    HIPEIG_CHECK(hipGetLastError());
    hipeig_phase_mark(c, 3);
    return 0;
  }
  if (hipeig_allgather_x(c, A->gl, x, A->nrows, &xg)) return 4;
  const CsrView v = hipeig_csr_view(A);
  if (variant == 3) {
    TcooView t = hipeig_tcoo_view(A);
    for (int ub = 0; ub < A->t_nunits; ub += g * 4) {            // one launch per sweep
      t.unit_begin = ub;
      hipLaunchKernelGGL(spmv_tcoo_kernel, dim3(g), dim3(HIPEIG_BLOCK), hipeig_tcoo_lds_bytes(A), c->stream, t, xg, epi);
    }
  } else if (variant == 1)
    hipLaunchKernelGGL(spmv_vector_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, v, xg, epi);
  else
    hipLaunchKernelGGL(spmv_stream_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, v, xg, epi);
  HIPEIG_CHECK(hipGetLastError());
  hipeig_phase_mark(c, 3);
  return 0;
}

// One sweep for both halves of a complex operand (see PairEpilogue).  Taken for operators the automatic choice
// gives the TCOO-W kernel to (variant 4) on an unpartitioned context; everything else - small operators, pinned or
// reproducible variants, row-partitioned runs - runs the two real sweeps and the two updates the pair replaces.
static int launch_spmv_pair(hipeig_ctx* c, hipeig_csr* A, double zr, double zi, double sign,
                            const double* xr, const double* xi, double* yr, double* yi) {
  if (A->nrows == 0) return 0;
  const double ar = sign * zr, ai = sign * zi, as = (sign == 0.0) ? 1.0 : -sign;
  int variant = hipeig_csr_pick_variant(c, A);
  if (variant < 0) return 1;
  A->last_pair_fused = 0;
  // Measured (tools/pair_bench.py, 64 non-zeros per row): N = 1e6 (32 per row) 0.227 -> 0.187 ms, 2e6 0.68 -> 0.50 ms,
  // 4e6 1.38 -> 1.49 ms, 6e6 2.46 -> 2.45 ms, 8e6 3.29 -> 3.55 ms, 1e7 4.2 -> 4.7 ms (window / bin sizes change
  // nothing): with two accumulators per row a unit holds half the rows, and once the rows no longer fit ONE launch
  // the 16-byte gathers of the sparser tiles cost more line fills than the second pass over the stream saves.
  const char* env = getenv("HIPEIG_PAIR_SWEEP");                     // 0 = always two sweeps, 1 = always the pair sweep
  const bool one_launch = A->nrows <= (int64_t)c->num_cu * (TCOOW_MAX_RW / 2);
  const bool want = variant == 4 && !c->collectives && (env ? atoi(env) != 0 : one_launch);
  if (want) {
    const int rc = build_tcoow_layout(c, A, 1);
    if (rc == 1) return 1;
    if (rc == 0 && !tcoow_ensure_parts(c, 2 * A->gather_len)) {
      double* xp = c->ytmp;
      hipLaunchKernelGGL(pair_pack_kernel, dim3(grid_stream(2 * A->gather_len)), dim3(HIPEIG_BLOCK), 0, c->stream,
                         A->gather_len, xr, xi, reinterpret_cast<double2*>(xp));
      TcooView t = hipeig_tcoow_view(A);
      t.idx = A->p_idx; t.val = A->p_val; t.off = A->p_off;
      t.nunits = A->p_nunits; t.nwin = A->p_nwin; t.wbits = A->p_wbits; t.rw = A->p_rw;
      t.nrun = 0;
      PairEpilogue epi{ar, ai, as, xr + A->row_offset, xi + A->row_offset, yr, yi};
      int g = A->p_wgs_per_sweep < A->p_nunits ? A->p_wgs_per_sweep : A->p_nunits;
      const size_t lds = (size_t)2 * A->p_rw * sizeof(double) + ((size_t)A->p_nwin + 2) * sizeof(uint32_t);
      for (int ub = 0; ub < A->p_nunits; ub += g) {                    // one launch per sweep
        t.unit_begin = ub;
        hipLaunchKernelGGL(spmv_tcoow_pair_kernel, dim3(g), dim3(TCOOW_THREADS), lds, c->stream, t, xp, epi);
      }
      HIPEIG_CHECK(hipGetLastError());
      A->last_pair_fused = 1;
      return 0;
    }
  }
  const double a_self = (sign == 0.0) ? 0.0 : ar;
  if (launch_spmv(c, A, a_self, as, xr, yr)) return 1;
  if (launch_spmv(c, A, a_self, as, xi, yi)) return 1;
  if (ai != 0.0) {
    const int64_t o = c->collectives ? 0 : A->row_offset;
    if (hipeig_axpby(c, A->nrows, -ai, xi + o, 1.0, yr)) return 1;
    if (hipeig_axpby(c, A->nrows, ai, xr + o, 1.0, yi)) return 1;
  }
  return 0;
}

extern "C" int hipeig_spmv_shift_pair(hipeig_ctx* c, hipeig_csr* A, double zr, double zi, double sign,
                                      const double* xr, const double* xi, double* yr, double* yi) {
  HIPEIG_REQUIRE(xr != yr && xr != yi && xi != yr && xi != yi && yr != yi, "in-place product is not supported");
  HIPEIG_REQUIRE(sign == 1.0 || sign == -1.0 || sign == 0.0, "sign must be +1, -1 or 0 (plain product)");
  return launch_spmv_pair(c, A, zr, zi, sign, xr, xi, yr, yi);
}

extern "C" int hipeig_csr_pair_info(hipeig_csr* A, int64_t out[2]) {
  out[0] = A->last_pair_fused;
  out[1] = 0;
  if (A->last_pair_fused && A->p_nunits > 0) {
    const int g = A->p_wgs_per_sweep < A->p_nunits ? A->p_wgs_per_sweep : A->p_nunits;
    out[1] = (A->p_nunits + g - 1) / g;
  }
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

// ---- TCOO construction -------------------------------------------------------------------
// One workgroup per unit (RW consecutive rows), rows handled in chunks of 256 (one row per
// thread).  count pass: non-zeros per (unit, window).  fill pass: a non-zero of row r and
// window c goes to  off[unit][c] + (#nnz of window c in earlier rows of the unit) + (#earlier
// nnz of window c in row r): row-major order inside every tile, independent of scheduling.

__global__ void __launch_bounds__(256)
tcoo_build_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                  const double* __restrict__ val, int64_t nrows, int rw, int nwin, int wbits,
                  uint32_t* __restrict__ counts /* [nunits][nwin], count pass */,
                  const uint32_t* __restrict__ off /* fill pass */, uint32_t* __restrict__ t_idx,
                  double* __restrict__ t_val, int fill) {
  extern __shared__ uint32_t tb_lds[];               // cnt[256][nwin] then base[nwin]
  uint32_t* cnt = tb_lds;
  uint32_t* base = tb_lds + 256 * nwin;
  const int u = blockIdx.x, t = threadIdx.x;
  for (int c = t; c < nwin; c += 256) base[c] = fill ? off[(size_t)u * nwin + c] : 0u;
  const int64_t row0 = (int64_t)u * rw;
  for (int chunk = 0; chunk < rw; chunk += 256) {
    const int64_t r = row0 + chunk + t;
    const bool live = (chunk + t < rw) && (r < nrows);
    for (int c = 0; c < nwin; ++c) cnt[t * nwin + c] = 0;
    int s = 0, e = 0;
    if (live) {
      s = rowptr[r]; e = rowptr[r + 1];
      for (int p = s; p < e; ++p) cnt[t * nwin + (col[p] >> wbits)] += 1;
    }
    __syncthreads();
    // column-wise exclusive scan over the 256 rows of the chunk (thread c scans window c)
    for (int c = t; c < nwin; c += 256) {
      uint32_t run = base[c];
      for (int k = 0; k < 256; ++k) {
        const uint32_t v = cnt[k * nwin + c];
        cnt[k * nwin + c] = run;
        run += v;
      }
      base[c] = run;
    }
    __syncthreads();
    if (fill && live) {
      const uint32_t rl = (uint32_t)(chunk + t);
      for (int p = s; p < e; ++p) {
        const int cc = col[p];
        const int c = cc >> wbits;
        const uint32_t dst = cnt[t * nwin + c]++;
        t_idx[dst] = (rl << wbits) | ((uint32_t)cc & ((1u << wbits) - 1u));
        t_val[dst] = val[p];
      }
    }
    __syncthreads();
  }
  if (!fill)
    for (int c = t; c < nwin; c += 256) counts[(size_t)u * nwin + c] = base[c];
}

size_t hipeig_tcoo_lds_bytes(const hipeig_csr* A) { return (size_t)4 * A->t_rw * sizeof(double); }

// Build the column-window blocked copy (idempotent).  Returns 0 on success, 1 on failure,
// 2 if the operator does not suit the layout (caller falls back to the CSR-stream kernel).
int hipeig_csr_build_tcoo(hipeig_ctx* c, hipeig_csr* A) {
  if (A->t_idx) return 0;
  if (A->nnz == 0 || A->nrows == 0) return 2;
  int wbits = TCOO_MAX_WBITS;
  if (const char* e = getenv("HIPEIG_TCOO_WBITS")) wbits = atoi(e);          // tuning knob
  HIPEIG_REQUIRE(wbits >= 10 && wbits <= 22, "HIPEIG_TCOO_WBITS out of range");
  while (wbits > 10 && ((int64_t)1 << (wbits - 1)) >= A->gather_len) --wbits;   // one window if x is short
  const int nwin = (int)((A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits);
  if (nwin > TCOO_MAX_WIN) return 2;
  // rows per wave: fill 2 workgroups x 4 waves per CU, at most TCOO_MAX_RW (20 KiB LDS per wave)
  int64_t rw = (A->nrows + (int64_t)c->num_cu * 8 - 1) / ((int64_t)c->num_cu * 8);
  rw = (rw + 63) / 64 * 64;
  if (rw < 64) rw = 64;
  if (rw > TCOO_MAX_RW) rw = TCOO_MAX_RW;
  if (const char* e = getenv("HIPEIG_TCOO_RW")) rw = (atoi(e) + 63) / 64 * 64;   // tuning knob
  HIPEIG_REQUIRE(rw >= 64 && rw * 32 <= 163840, "HIPEIG_TCOO_RW out of range");
  if (rw > ((int64_t)1 << (32 - wbits))) rw = (int64_t)1 << (32 - wbits);
  const int nunits = (int)((A->nrows + rw - 1) / rw);
  uint32_t* d_counts = nullptr;
  const size_t ntile = (size_t)nunits * nwin;
  HIPEIG_CHECK(hipMalloc((void**)&d_counts, ntile * sizeof(uint32_t)));
  const size_t lds = ((size_t)256 * nwin + nwin) * sizeof(uint32_t);
  HIPEIG_CHECK(hipFuncSetAttribute((const void*)tcoo_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(((size_t)256 * TCOO_MAX_WIN + TCOO_MAX_WIN) * sizeof(uint32_t))));
  hipLaunchKernelGGL(tcoo_build_kernel, dim3(nunits), dim3(256), lds, c->stream, A->d_rowptr, A->d_col, A->d_val,
                     A->nrows, (int)rw, nwin, wbits, d_counts, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                     (double*)nullptr, 0);
  HIPEIG_CHECK(hipGetLastError());
  std::vector<uint32_t> off(ntile + 1);
  HIPEIG_CHECK(hipMemcpyAsync(off.data() + 1, d_counts, ntile * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  off[0] = 0;
  uint64_t run = 0;
  for (size_t i = 1; i <= ntile; ++i) { run += off[i]; off[i] = (uint32_t)run; }
  HIPEIG_REQUIRE(run == (uint64_t)A->nnz, "TCOO count pass lost non-zeros");
  HIPEIG_CHECK(hipFree(d_counts));
  HIPEIG_CHECK(hipMalloc((void**)&A->t_off, (ntile + 1) * sizeof(uint32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->t_idx, (size_t)A->nnz * sizeof(uint32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->t_val, (size_t)A->nnz * sizeof(double)));
  HIPEIG_CHECK(hipMemcpyAsync(A->t_off, off.data(), (ntile + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(tcoo_build_kernel, dim3(nunits), dim3(256), lds, c->stream, A->d_rowptr, A->d_col, A->d_val,
                     A->nrows, (int)rw, nwin, wbits, (uint32_t*)nullptr, (const uint32_t*)A->t_off, A->t_idx, A->t_val, 1);
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  A->t_nunits = nunits; A->t_nwin = nwin; A->t_wbits = wbits; A->t_rw = (int)rw;
  {
    int per_cu = (int)(163840 / ((size_t)4 * rw * sizeof(double)));           // LDS-limited residency
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    if (const char* e = getenv("HIPEIG_TCOO_WG_PER_CU")) per_cu = atoi(e);    // tuning knob
    HIPEIG_REQUIRE(per_cu >= 1 && per_cu <= 8, "HIPEIG_TCOO_WG_PER_CU out of range");
    A->t_wgs_per_sweep = per_cu * c->num_cu;
    A->t_prefetch = 0;
    if (const char* e = getenv("HIPEIG_TCOO_PREFETCH")) A->t_prefetch = atoi(e) != 0;   // tuning knob
  }
  HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmv_tcoo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)HIPEIG_TCOO_LDS_MAX));   // the largest any operator may ask for
  A->bytes += (int64_t)A->nnz * 12 + (int64_t)(ntile + 1) * 4;
  return 0;
}

// ---- TCOO-W construction --------------------------------------------------------------------
// Bucketing by (unit, column bin) with global counters: count, scan on the host, scatter.
// Which slot of its bin a non-zero lands in depends on scheduling; the SET of non-zeros of
// every bin does not, and the kernel's accumulation order is unordered anyway.
__global__ void __launch_bounds__(256)
tcoow_bin_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                 const double* __restrict__ val, int64_t nrows, int rw, int binbits, int nbins, int wbits,
                 uint32_t* __restrict__ cursor, uint32_t* __restrict__ w_idx, double* __restrict__ w_val, int fill) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t wmask = (1u << wbits) - 1u;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t unit = r / rw;
    const uint32_t rl = (uint32_t)(r - unit * rw);
    uint32_t* cur = cursor + unit * nbins;
    const int s = rowptr[r], e = rowptr[r + 1];
    for (int p = s + lane; p < e; p += 64) {
      const uint32_t cc = (uint32_t)col[p];
      const uint32_t slot = atomicAdd(cur + (cc >> binbits), 1u);
      if (fill) {
        w_idx[slot] = (rl << wbits) | (cc & wmask);
        w_val[slot] = val[p];
      }
    }
  }
}

size_t hipeig_tcoow_lds_bytes(const hipeig_csr* A) {
  return (size_t)A->w_rw * sizeof(double) + ((size_t)A->w_nwin + 2) * sizeof(uint32_t);
}

// pair = 1 builds the copy of the pair sweep (two accumulators per row, so half the rows per unit; no column
// splits) into the p_* fields; everything else is the same construction.
static int build_tcoow_layout(hipeig_ctx* c, hipeig_csr* A, int pair) {
  if (pair ? (A->p_idx != nullptr) : (A->w_idx != nullptr)) return 0;
  if (A->nnz == 0 || A->nrows == 0) return 2;
  const int64_t max_rw = pair ? TCOOW_MAX_RW / 2 : TCOOW_MAX_RW;
  int wbits = 17;                                      // 15 bits are left for the row inside the unit
  if (const char* e = getenv(pair ? "HIPEIG_TCOOW_PAIR_WBITS" : "HIPEIG_TCOOW_WBITS")) wbits = atoi(e);         // tuning knob
  HIPEIG_REQUIRE(wbits >= 10 && wbits <= 17, "HIPEIG_TCOOW_WBITS out of range");
  while (wbits > 10 && ((int64_t)1 << (wbits - 1)) >= A->gather_len) --wbits;
  const int nwin = (int)((A->gather_len + ((int64_t)1 << wbits) - 1) >> wbits);
  if (nwin > TCOOW_MAX_WIN) return 2;
  int binbits = 4;                                     // 16 columns = ONE line of x per bin (with aligned bins: 4 / 5 / 6 / 7 -> 1.90 / 1.93 / 1.95 / 2.01 ms per product)
  if (const char* e = getenv(pair ? "HIPEIG_TCOOW_PAIR_BINBITS" : "HIPEIG_TCOOW_BINBITS")) binbits = atoi(e);     // tuning knob
  if (binbits > wbits) binbits = wbits;
  HIPEIG_REQUIRE(binbits >= 3, "HIPEIG_TCOOW_BINBITS out of range");
  const int bpw = 1 << (wbits - binbits);
  const int64_t nbins = (int64_t)nwin * bpw;
  // one unit per CU if it fits; otherwise the fewest sweeps that fit the LDS, with the rows spread
  // evenly over sweeps * CUs units so that the last sweep is as full as the first
  int64_t sweeps = (A->nrows + (int64_t)c->num_cu * max_rw - 1) / ((int64_t)c->num_cu * max_rw);
  if (sweeps < 1) sweeps = 1;
  int64_t rw = (A->nrows + sweeps * c->num_cu - 1) / (sweeps * c->num_cu);
  rw = (rw + 63) / 64 * 64;
  if (rw < 64) rw = 64;
  if (rw > max_rw) rw = max_rw;
  // Fewer full-size row blocks than half the CUs (a small operator, or the slab of a many-GPU run):
  // keep the row blocks as tall as the LDS allows and let `csplit` workgroups share each of them by
  // column ranges, so that the traffic of x through the L1s is (row blocks) x |x| instead of CUs x |x|.
  int csplit = 1;
  if (!pair) {
    const int64_t rb_min = (A->nrows + TCOOW_MAX_RW - 1) / TCOOW_MAX_RW;
    // Worth it only when x is much larger than the L2s (measured: slab of N = 1e7 on 1/8 of the rows
    // 0.53 -> 0.34 ms, on 1/4 0.90 -> 0.62 ms; at N = 1e6, x = 8 MB, the combine launch costs more than
    // the sweep gains: MINRES iteration 0.141 -> 0.163 ms).
    // Half of what the splits buy on these slabs is column locality per XCD (round 4, EXPERIMENTS.md R4-fold / R4-multigpu):
    // share = blockIdx % csplit, XCD = blockIdx % 8, so with 2 / 4 splits (the slabs of a 4- / 8-GPU run) an XCD's L2 sees
    // a half / a quarter of x, while 3 splits (6 ranks) leave every XCD all of x: 0.46 ms against 0.32 ideal.
    int want = (rb_min * 2 <= c->num_cu && A->gather_len >= 4000000) ? (int)(c->num_cu / rb_min) : 1;
    if (const char* e = getenv("HIPEIG_TCOOW_CSPLIT")) want = atoi(e);          // tuning knob (1 = off)
    if (want > 64) want = 64;
    if (want > 1 && rb_min * want <= HIPEIG_MAX_PARTIALS) {
      csplit = want;
      rw = ((A->nrows + rb_min - 1) / rb_min + 63) / 64 * 64;
      if (rw > TCOOW_MAX_RW) rw = TCOOW_MAX_RW;
    }
  }
  if (const char* e = getenv(pair ? "HIPEIG_TCOOW_PAIR_RW" : "HIPEIG_TCOOW_RW")) rw = (atoi(e) + 63) / 64 * 64;   // tuning knob
  HIPEIG_REQUIRE(rw >= 64 && rw <= max_rw && rw <= ((int64_t)1 << (32 - wbits)), "HIPEIG_TCOOW_RW out of range");
  const int64_t nunits = (A->nrows + rw - 1) / rw;
  const size_t ncnt = (size_t)(nunits * nbins);
  HIPEIG_REQUIRE(ncnt < ((size_t)1 << 30), "too many (unit, bin) counters");
  uint32_t* d_cur = nullptr;
  HIPEIG_CHECK(hipMalloc((void**)&d_cur, ncnt * sizeof(uint32_t)));
  HIPEIG_CHECK(hipMemsetAsync(d_cur, 0, ncnt * sizeof(uint32_t), c->stream));
  const int grid = 8 * c->num_cu;
  hipLaunchKernelGGL(tcoow_bin_kernel, dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->nrows,
                     (int)rw, binbits, (int)nbins, wbits, d_cur, (uint32_t*)nullptr, (double*)nullptr, 0);
  HIPEIG_CHECK(hipGetLastError());
  std::vector<uint32_t> cnt(ncnt);
  HIPEIG_CHECK(hipMemcpyAsync(cnt.data(), d_cur, ncnt * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  const size_t ntile = (size_t)nunits * nwin;
  std::vector<uint32_t> off(ntile + 1);
  // Bins never straddle a 64-element instruction group and tiles start on group boundaries (round 3; the gaps are
  // padding slots - index 0xFFFFFFFF, value 0 - that the sweep skips; ~2 % of the stream).  A line of x is then fetched
  // by exactly ONE gather instruction of one wave: L1 -> L2 read requests per launch 1.829e8 -> 1.704e8, i.e. gather fills
  // per non-zero 0.468 -> 0.431 against the ideal (1 - exp(-k))/k = 0.426, and 2.083 -> 1.90 ms per product at N = 1e7
  // (profiles/r03_spmv_aligned_bins.txt).  HIPEIG_TCOOW_ALIGN=0 restores the unaligned stream.
  const char* al_env = getenv(pair ? "HIPEIG_TCOOW_PAIR_ALIGN" : "HIPEIG_TCOOW_ALIGN");
  const bool align = !(al_env && atoi(al_env) == 0);
  uint64_t run = 0, counted = 0;
  for (size_t i = 0; i < ncnt; ++i) {
    if (i % bpw == 0) {
      if (align) run = (run + 63) & ~(uint64_t)63;
      off[i / bpw] = (uint32_t)run;
    }
    const uint32_t n = cnt[i];
    if (align && n > 0 && n <= 64 && (run & 63) + n > 64) run = (run + 63) & ~(uint64_t)63;
    cnt[i] = (uint32_t)run;                            // exclusive scan in place -> scatter cursors
    run += n;
    counted += n;
  }
  off[ntile] = (uint32_t)run;
  HIPEIG_REQUIRE(counted == (uint64_t)A->nnz, "TCOO-W count pass lost non-zeros");
  HIPEIG_REQUIRE(run < ((uint64_t)1 << 32), "blocked stream too long for 32-bit offsets");
  const size_t nslots = (size_t)run;                   // stream length incl. padding (== nnz without alignment)
  HIPEIG_CHECK(hipMemcpyAsync(d_cur, cnt.data(), ncnt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  uint32_t*& l_off = pair ? A->p_off : A->w_off;       // owned by the operator from the allocation on
  uint32_t*& l_idx = pair ? A->p_idx : A->w_idx;
  double*& l_val = pair ? A->p_val : A->w_val;
  HIPEIG_CHECK(hipMalloc((void**)&l_off, (ntile + 1) * sizeof(uint32_t)));
  {
    // experiment knob: place the once-read stream in uncached (MTYPE_UC) memory so that it does not
    // occupy L2 lines / tag bandwidth next to the x window
    const char* uc = getenv("HIPEIG_TCOOW_UNCACHED");
    if (uc && atoi(uc) != 0) {
      HIPEIG_CHECK(hipExtMallocWithFlags((void**)&l_idx, (size_t)A->nnz * sizeof(uint32_t), hipDeviceMallocUncached));
      HIPEIG_CHECK(hipExtMallocWithFlags((void**)&l_val, (size_t)A->nnz * sizeof(double), hipDeviceMallocUncached));
    } else {
      HIPEIG_CHECK(hipMalloc((void**)&l_idx, nslots * sizeof(uint32_t)));
      HIPEIG_CHECK(hipMalloc((void**)&l_val, nslots * sizeof(double)));
    }
    if (nslots != (size_t)A->nnz) {                      // padding slots: sentinel index, zero value
      HIPEIG_CHECK(hipMemsetAsync(l_idx, 0xFF, nslots * sizeof(uint32_t), c->stream));
      HIPEIG_CHECK(hipMemsetAsync(l_val, 0, nslots * sizeof(double), c->stream));
    }
  }
  HIPEIG_CHECK(hipMemcpyAsync(l_off, off.data(), (ntile + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(tcoow_bin_kernel, dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->nrows,
                     (int)rw, binbits, (int)nbins, wbits, d_cur, l_idx, l_val, 1);
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  HIPEIG_CHECK(hipFree(d_cur));
  int per_cu = (int)(163840 / ((pair ? 2 : 1) * rw * sizeof(double) + ((size_t)nwin + 2) * sizeof(uint32_t) + 256));
  if (per_cu > 2) per_cu = 2;                          // 1024-thread workgroups: at most 32 waves per CU
  if (per_cu < 1) per_cu = 1;
  if (const char* e = getenv("HIPEIG_TCOOW_WG_PER_LAUNCH_CU")) per_cu = atoi(e);   // tuning knob: workgroups per CU in ONE launch (2 with full-height units: both sweeps in one launch, the second workgroup of a CU starts when its first ends)
  HIPEIG_REQUIRE(per_cu >= 1 && per_cu <= 8, "HIPEIG_TCOOW_WG_PER_LAUNCH_CU out of range");
  if (pair) {
    A->p_nunits = (int)nunits; A->p_nwin = nwin; A->p_wbits = wbits; A->p_rw = (int)rw;
    A->p_wgs_per_sweep = per_cu * c->num_cu;
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmv_tcoow_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)HIPEIG_TCOOW_LDS_MAX));
    A->bytes += (int64_t)A->nnz * 12 + (int64_t)(ntile + 1) * 4;
    return 0;
  }
  A->w_nunits = (int)nunits; A->w_nwin = nwin; A->w_wbits = wbits; A->w_rw = (int)rw;
  A->w_csplit = csplit;
  A->w_binbits = binbits; A->w_align = align ? 1 : 0; A->w_slots = (int64_t)nslots;
  A->w_wgs_per_sweep = per_cu * c->num_cu;
  HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmv_tcoow_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)HIPEIG_TCOOW_LDS_MAX));
  HIPEIG_CHECK(hipFuncSetAttribute((const void*)spmv_tcoow_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)HIPEIG_TCOOW_LDS_MAX));
  {
    // bound of the fixed-point form (variant 5): max_i sum_j |a_ij| over the local rows
    const int gr = 8 * c->num_cu < HIPEIG_MAX_PARTIALS ? 8 * c->num_cu : HIPEIG_MAX_PARTIALS;
    double* area = c->d_partials + HIPEIG_FX_AREA;
    hipLaunchKernelGGL(rowabs_max_kernel, dim3(gr), dim3(HIPEIG_BLOCK), 0, c->stream, A->d_rowptr, A->d_val, A->nrows, area);
    std::vector<double> part((size_t)gr);
    HIPEIG_CHECK(hipMemcpyAsync(part.data(), area, sizeof(double) * gr, hipMemcpyDeviceToHost, c->stream));
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    double m = 0.0;
    for (double v : part) m = (v > m || v != v) ? v : m;
    A->absrow_max = m;
  }
  A->bytes += (int64_t)nslots * 12 + (int64_t)(ntile + 1) * 4;
  return 0;
}

int hipeig_csr_build_tcoow(hipeig_ctx* c, hipeig_csr* A) { return build_tcoow_layout(c, A, 0); }

// The variant a launch will use (building the TCOO copy on demand); -1 on failure.
int hipeig_csr_pick_variant(hipeig_ctx* c, hipeig_csr* A) {
  int variant = A->variant;
  if (variant == 0) {
    // the gathered operand does not fit one XCD's L2 -> window it; small problems stream
    variant = (A->gather_len * 8 > (int64_t)3 << 20 && A->nnz > (int64_t)1 << 20) ? (A->reproducible ? 5 : 4) : 2;
  }
  if (variant == 4 || variant == 5) {
    const int rc = hipeig_csr_build_tcoow(c, A);
    if (rc == 1) return -1;
    if (rc == 2) variant = 3;
  }
  if (variant == 3) {
    const int rc = hipeig_csr_build_tcoo(c, A);
    if (rc == 1) return -1;
    if (rc == 2) variant = 2;
  }
  A->last_variant = variant;
  A->last_launches = 1;
  if (variant == 4 || variant == 5) {
    int g = A->w_wgs_per_sweep < A->w_nunits ? A->w_wgs_per_sweep : A->w_nunits;
    A->last_launches = (A->w_csplit > 1) ? 1 : (A->w_nunits + g - 1) / g;     // sweep launches (split mode adds a combine launch)
  } else if (variant == 3) {
    int64_t g = A->t_wgs_per_sweep;
    const int64_t need = (A->t_nunits + 3) / 4;
    if (g > need) g = need;
    if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
    A->last_launches = (int)((A->t_nunits + g * 4 - 1) / (g * 4));
  }
  return variant;
}

// ---- layout preparation ------------------------------------------------------------------
__global__ void remap_cols_kernel(int32_t* __restrict__ col, int64_t nnz, const int64_t* __restrict__ offs,
                                  GatherLayout gl) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += step) {
    const int64_t j = col[p];
    int lo = 0, hi = gl.nranks - 1;                    // owner rank: offs[r] <= j < offs[r + 1] (empty slabs have equal offsets)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (j >= offs[mid]) lo = mid; else hi = mid - 1;
    }
    col[p] = (int32_t)gl.pos(lo, j - offs[lo]);
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

  A->gather_len = A->ncols;
  // distributed: gather row counts, size x_full and remap global columns to its layout
  if (hipeig_comm_setup_rows(c, A->nrows, &A->gl)) return 4;
  if (c->collectives) {
    HIPEIG_REQUIRE(c->nranks <= HIPEIG_MAX_RANKS, "too many ranks");
    std::vector<int64_t> offs(c->nranks + 1, 0);
    for (int k = 0; k < c->nranks; ++k) offs[k + 1] = offs[k] + c->row_counts[k];
    HIPEIG_REQUIRE(offs[c->nranks] == A->ncols, "row counts over ranks must add up to ncols");
    HIPEIG_REQUIRE(offs[c->rank] == A->row_offset, "row_offset does not match the rank order");
    HIPEIG_REQUIRE(A->gl.total() < (int64_t)1 << 31, "gathered operand too long for int32 columns");
    int64_t* d_offs = (int64_t*)(c->d_scalars + 1024);
    HIPEIG_CHECK(hipMemcpyAsync(d_offs, offs.data(), sizeof(int64_t) * (c->nranks + 1),
                                hipMemcpyHostToDevice, c->stream));
    if (A->nnz > 0) {
      hipLaunchKernelGGL(remap_cols_kernel, dim3(2048), dim3(HIPEIG_BLOCK), 0, c->stream,
                         A->d_col, A->nnz, d_offs, A->gl);
      HIPEIG_CHECK(hipGetLastError());
    }
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    A->col_stride = A->gl.h;
    A->gather_len = A->gl.total();
  }
  return 0;
}

static int csr_upload(hipeig_ctx* c, hipeig_csr* A, const int32_t* rp32, const int32_t* col, const double* val) {
  const int64_t nrows = A->nrows, nnz = A->nnz;
  HIPEIG_CHECK(hipMalloc((void**)&A->d_rowptr, (size_t)(nrows + 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_col, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_val, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double)));
  HIPEIG_CHECK(hipMemcpyAsync(A->d_rowptr, rp32, (size_t)(nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (nnz > 0) {
    HIPEIG_CHECK(hipMemcpyAsync(A->d_col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPEIG_CHECK(hipMemcpyAsync(A->d_val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
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
  int rc = csr_upload(c, A, rp32.data(), col, val);
  if (rc == 0) rc = hipeig_csr_finalize(c, A, rp32.data());
  if (rc) { hipeig_csr_destroy(c, A); return rc; }        // nothing allocated so far outlives a failure
  *out = A;
  return 0;
}

extern "C" int hipeig_csr_destroy(hipeig_ctx* c, hipeig_csr* A) {
  if (!A) return 0;
  hipStreamSynchronize(c->stream);
  if (c->mr_graph && c->mr_graph_key && *(const void**)c->mr_graph_key == (const void*)A) {
    hipGraphExecDestroy(c->mr_graph);      // the captured MINRES chunk points into this operator
    c->mr_graph = nullptr;
  }
  if (A->d_rowptr) hipFree(A->d_rowptr);
  if (A->d_col) hipFree(A->d_col);
  if (A->d_val) hipFree(A->d_val);
  if (A->d_row_blocks) hipFree(A->d_row_blocks);
  if (A->t_idx) hipFree(A->t_idx);
  if (A->t_val) hipFree(A->t_val);
  if (A->t_off) hipFree(A->t_off);
  if (A->w_idx) hipFree(A->w_idx);
  if (A->w_val) hipFree(A->w_val);
  if (A->w_off) hipFree(A->w_off);
  if (A->p_idx) hipFree(A->p_idx);
  if (A->p_val) hipFree(A->p_val);
  if (A->p_off) hipFree(A->p_off);
  for (int q = 0; q < 3; ++q) {
    if (A->bl[q].idx) hipFree(A->bl[q].idx);
    if (A->bl[q].val) hipFree(A->bl[q].val);
    if (A->bl[q].off) hipFree(A->bl[q].off);
  }
  free(A);
  return 0;
}

// Layout constants of the blocked copy the product runs on (what a committed counter profile is only valid for):
// out[0] = kernel variant of the last launch, [1] rows per row block, [2] window bits, [3] row blocks, [4] windows,
// [5] column splits, [6] workgroups per sweep launch, [7] threads per workgroup, [8] batch unroll, [9] chunks of the
// operand exchange, [10] rows per (rank, chunk) of the gathered layout (0: not partitioned)
extern "C" int hipeig_csr_layout_info(hipeig_csr* A, int64_t out[12]) {
  memset(out, 0, 12 * sizeof(int64_t));
  out[0] = A->last_variant;
  if (A->last_variant == 4 || A->last_variant == 5) {
    out[1] = A->w_rw; out[2] = A->w_wbits; out[3] = A->w_nunits; out[4] = A->w_nwin; out[5] = A->w_csplit;
    out[6] = A->w_wgs_per_sweep; out[7] = TCOOW_THREADS; out[8] = TCOO_UNROLL;
    out[11] = A->w_binbits + 100 * A->w_align;
  } else if (A->last_variant == 3) {
    out[1] = A->t_rw; out[2] = A->t_wbits; out[3] = A->t_nunits; out[4] = A->t_nwin; out[6] = A->t_wgs_per_sweep;
    out[7] = HIPEIG_BLOCK; out[8] = TCOO_UNROLL;
  }
  if (A->col_stride > 0) { out[9] = A->gl.nchunks; out[10] = A->gl.h; }
  return 0;
}

extern "C" int hipeig_csr_info(hipeig_csr* A, int64_t info[8]) {
  info[0] = A->nrows; info[1] = A->ncols; info[2] = A->nnz; info[3] = A->row_offset;
  info[4] = A->last_variant; info[5] = A->bytes; info[6] = A->n_row_blocks; info[7] = A->last_launches;
  return 0;
}

// Error-bound ingredients of the fixed-point variant: out[0] = max_i sum_j |a_ij| (local rows, set when the
// blocked layout is built; 0 before), out[1] = max |x| of the operand of the most recent variant-5 product.
extern "C" int hipeig_csr_fixed_info(hipeig_ctx* c, hipeig_csr* A, double out[2]) {
  out[0] = A->absrow_max;
  out[1] = 0.0;
  const int g = grid_for(A->gather_len, 8);
  std::vector<double> part((size_t)g);
  HIPEIG_CHECK(hipMemcpyAsync(part.data(), c->d_partials + HIPEIG_FX_AREA, sizeof(double) * g, hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  for (double v : part) out[1] = (v > out[1] || v != v) ? v : out[1];
  return 0;
}

extern "C" int hipeig_csr_set_variant(hipeig_csr* A, int variant) {
  HIPEIG_REQUIRE(variant >= 0 && variant <= 5, "unknown variant");
  A->variant = variant;
  return 0;
}

extern "C" int hipeig_csr_set_reproducible(hipeig_csr* A, int on) {
  A->reproducible = on ? 1 : 0;
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
