// BLAS-1 class and tall-skinny kernels of the Krylov loop (all HBM-bound, fp64).
//
// Every kernel is a grid-stride sweep with 16-byte (double2) accesses, at most
// HIPEIG_MAX_PARTIALS workgroups of 256 threads.  Reductions are two-stage with a fixed
// tree (wave shuffle -> LDS -> per-workgroup partial -> fixed-order final sum), never
// atomics, so the results of THESE kernels are bitwise reproducible.  (The operator sweeps are a
// separate matter: kernel variants 1-3 of spmv_device.h fix the order of the adds inside a row, the
// default variant 4 for large operators and the block kernel of spmm_device.h accumulate with LDS
// atomics and reproduce a row's sum only to rounding.)
#include "common.h"
#include <math.h>
#include <vector>

struct PtrTable {
  const double* p[HIPEIG_MAX_COLS];
};
struct CoefTable {
  double c[HIPEIG_MAX_COLS];
};

// ---- reductions of RECORDS that finish inside the kernel that produces them ------------------------------------------
// multi_dot (one value per column), the MGS pair of dots, a Gram block (up to 1024 elements): every workgroup leaves a
// record of nv values, and the totals are formed without a second launch (round 3 did this for single scalars, common.h):
//   * workgroup b stores its record at partials[b * nv ..] (agent-scope stores, waited for) and takes a ticket of its
//     GROUP of 64 workgroups;
//   * the workgroup that completes a group adds the group's <= 64 records in ascending order into a group record and takes
//     a ticket of the master counter; the one that completes the master adds the <= 32 group records in order and stores
//     the totals (to mapped host memory on one GPU: a stream wait, no copy).
// Two levels because one workgroup adding 2048 records of 1 KiB would take longer than the launch it replaces; the group
// sums run on as many CUs as there are groups.  Order and tree are fixed by (gridDim.x, nv): bitwise reproducible.
#define REC_GROUP 64
#define REC_GROUP_DOUBLES 32768            // size of ctx->d_group_partials: (groups) x (values per record) must fit

__device__ __forceinline__ double agent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store_nowait(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sum_{i = first, first + step, ... < count} p[i * nv]: loads in batches of 8 (all in flight), adds in index order
__device__ __forceinline__ double agent_sum_strided(const double* p, int first, int count, int step, size_t nv) {
  double a = 0.0;
  int i = first;
  for (; i + 7 * step < count; i += 8 * step) {
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = agent_load(p + (size_t)(i + q * step) * nv);
#pragma unroll
    for (int q = 0; q < 8; ++q) a += t[q];
  }
  for (; i < count; i += step) a += agent_load(p + (size_t)i * nv);
  return a;
}

// Totals of `count` records of nv values (record-major) by ONE 256-thread workgroup; store(e, total_e) is called once per
// value.  nv <= 128: 256 / nvp lanes share a value (nvp = nv rounded up to a power of two), their sums are added in lane
// order through LDS; larger records: one thread per value, 256 values at a time.  lds: 256 doubles.
template <class Store>
__device__ __forceinline__ void sum_records(const double* p, int count, int nv, double* lds, Store store) {
  const int t = threadIdx.x;
  if (nv <= 128) {
    int nvp = 1;
    while (nvp < nv) nvp <<= 1;
    const int kl = HIPEIG_BLOCK / nvp, v = t & (nvp - 1), k = t / nvp;
    lds[k * nvp + v] = (v < nv) ? agent_sum_strided(p + v, k, count, kl, (size_t)nv) : 0.0;
    __syncthreads();
    if (t < nv) {
      double s = lds[t];
      for (int q = 1; q < kl; ++q) s += lds[q * nvp + t];
      store(t, s);
    }
    __syncthreads();
  } else {
    for (int e = t; e < nv; e += HIPEIG_BLOCK) store(e, agent_sum_strided(p + e, 0, count, 1, (size_t)nv));
  }
}

// Call after every thread of the workgroup has issued the agent_store_nowait's of its share of the record
// partials[blockIdx.x * nv ..].  counters: 1 + ceil(gridDim.x / 64) words, zero before the kernel and after it.
__device__ __forceinline__ void finish_records(double* partials, double* group_partials, int nv, unsigned* counters,
                                               double* lds /* 256 doubles */, double* out) {
  __shared__ int sh_flag;
  const unsigned G = gridDim.x, ng = (G + REC_GROUP - 1) / REC_GROUP, g = blockIdx.x / REC_GROUP;
  const unsigned gsize = (g == ng - 1) ? G - g * REC_GROUP : REC_GROUP;
  wait_for_my_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(counters + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u;
    if (last) __hip_atomic_store(counters + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_flag = last;
  }
  __syncthreads();
  if (!sh_flag) return;                               // uniform
  __syncthreads();
  if (ng == 1) {
    sum_records(partials, (int)G, nv, lds, [&](int e, double s) { out[e] = s; });
    return;
  }
  double* mine = group_partials + (size_t)g * nv;
  sum_records(partials + (size_t)g * REC_GROUP * nv, (int)gsize, nv, lds, [&](int e, double s) { agent_store_nowait(mine + e, s); });
  wait_for_my_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u;
    if (last) __hip_atomic_store(counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_flag = last;
  }
  __syncthreads();
  if (!sh_flag) return;
  __syncthreads();
  sum_records(group_partials, (int)ng, nv, lds, [&](int e, double s) { out[e] = s; });
}

// Where a record kernel stores its totals: the pinned, mapped host buffer when the caller wants them on the host of a
// single-GPU context (the result then needs a stream wait but no copy command), else the device scalar area.
static double* record_target(hipeig_ctx* c, bool to_host) {
  return (to_host && !c->collectives && c->h_scalars_dev) ? c->h_scalars_dev : c->d_scalars;
}

// The totals are where record_target() said: sum them over the ranks if the context is partitioned and hand them over.
static int records_to_host(hipeig_ctx* c, int ncols, double* host_out) {
  HIPEIG_CHECK(hipGetLastError());
  if (host_out && !c->collectives && c->h_scalars_dev) {
    if (hipeig_sync_checked(c)) return 4;
    memcpy(host_out, c->h_scalars, sizeof(double) * ncols);
    return 0;
  }
  if (hipeig_allreduce_sum(c, c->d_scalars, ncols)) return 4;
  if (host_out) {
    HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(double) * ncols,
                                hipMemcpyDeviceToHost, c->stream));
    if (hipeig_sync_checked(c)) return 4;
    memcpy(host_out, c->h_scalars, sizeof(double) * ncols);
  }
  return 0;
}

// Grid of a streaming kernel that ends in finish_records: every workgroup pays a fixed price for its record (store, wait,
// ticket), so FEWER, fatter workgroups win as long as the chip stays busy.  Measured (tools/experiments/mgs_grid.sh,
// multidot_grid.sh; profiles/r04_mgs_fused_sweep_grid.txt, r04_multidot_grid.txt): MGS m = 16 at N = 1e7 with 2 / 8 / 32 /
// 64 / 128 elements per thread 1.71 / 1.16 / 0.93 / 0.97 / 1.44 ms, at N = 1e6 with 2 / 8 / 16 / 32: 0.42 / 0.277 / 0.275 /
// 0.35 ms; multi_dot m = 16 at N = 1e7 with 2 / 32 / 48: 0.273 / 0.245 / 0.267 ms, at N = 1e6 with 2 / 4 / 8 / 16: 0.063 /
// 0.053 / 0.054 / 0.072 ms.  Hence n / 8192 workgroups, but at least two per CU (as far as n / 1024 goes).
static int grid_records(const hipeig_ctx* c, int64_t n, const char* knob) {
  int64_t g = n / 8192;
  const int64_t floor2 = n / 1024 < 2 * (int64_t)c->num_cu ? n / 1024 : 2 * (int64_t)c->num_cu;
  if (g < floor2) g = floor2;
  if (const char* e = knob ? getenv(knob) : nullptr) g = atoi(e) > 0 ? n / (256 * (int64_t)atoi(e)) : g;      // tuning knob: elements per thread
  if (g < 1) g = 1;
  if (g > HIPEIG_MAX_PARTIALS) g = HIPEIG_MAX_PARTIALS;
  return (int)g;
}

// ---- dot / nrm2 ------------------------------------------------------------------------
// The total is formed by the kernel's last workgroup (common.h) and stored to `out` - on one GPU the pinned, mapped host
// word, so that a dot product is ONE launch and a stream wait (round 2: two launches).
__global__ void __launch_bounds__(HIPEIG_BLOCK)
dot_kernel(int64_t n, const double* __restrict__ x, const double* __restrict__ y,
           double* __restrict__ partials, unsigned* counters, double* __restrict__ out) {
  __shared__ double lds[4];
  const int64_t n2 = n >> 1;
  const double2* x2 = reinterpret_cast<const double2*>(x);
  const double2* y2 = reinterpret_cast<const double2*>(y);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (; i + stride < n2; i += 2 * stride) {        // two independent 16-byte loads in flight
    const double2 xa = x2[i], ya = y2[i];
    const double2 xb = x2[i + stride], yb = y2[i + stride];
    a0 = fma(xa.x, ya.x, a0); a0 = fma(xa.y, ya.y, a0);
    a1 = fma(xb.x, yb.x, a1); a1 = fma(xb.y, yb.y, a1);
  }
  for (; i < n2; i += stride) {
    const double2 xa = x2[i], ya = y2[i];
    a0 = fma(xa.x, ya.x, a0); a0 = fma(xa.y, ya.y, a0);
  }
  double a = a0 + a1;
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) a = fma(x[n - 1], y[n - 1], a);
  a = block_reduce_sum(a, lds);
  if (threadIdx.x == 0) store_partial(partials + blockIdx.x, a);
  if (last_block_ticket(counters, gridDim.x, blockIdx.x)) {
    const double t = sum_partials_agent(partials, gridDim.x, lds);
    if (threadIdx.x == 0) *out = t;
    release_ticket_counter(counters);
  }
}

extern "C" int hipeig_dot(hipeig_ctx* c, int64_t n, const double* x, const double* y, double* out) {
  HIPEIG_REQUIRE(out != nullptr, "null output");
  const int g = grid_records(c, n, "HIPEIG_DOT_PER_THREAD");
  unsigned* cnt = c->d_counters + 3 * HIPEIG_TICKET_WORDS;
  const bool direct = !c->collectives && c->h_scalars_dev;
  hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, x, y, c->d_partials, cnt,
                     direct ? c->h_scalars_dev : c->d_scalars);
  HIPEIG_CHECK(hipGetLastError());
  if (direct) {
    if (hipeig_sync_checked(c)) return 4;
    *out = c->h_scalars[0];
    return 0;
  }
  if (hipeig_allreduce_sum(c, c->d_scalars, 1)) return 4;
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  *out = c->h_scalars[0];
  return 0;
}

extern "C" int hipeig_nrm2(hipeig_ctx* c, int64_t n, const double* x, double* out) {
  double ss = 0.0;
  int rc = hipeig_dot(c, n, x, x, &ss);
  if (rc) return rc;
  *out = sqrt(ss);
  return 0;
}

// ---- scale / axpby ---------------------------------------------------------------------
// y = x * alpha (divide == 0) or y = x / alpha (divide == 1; the reference divides, e.g.
// array /= norm at numpyVector.py:77 and x/np.sqrt(innerprod) at :142)
__global__ void __launch_bounds__(HIPEIG_BLOCK)
scale_kernel(int64_t n, double alpha, int divide, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t n2 = n >> 1;
  const double2* x2 = reinterpret_cast<const double2*>(x);
  double2* y2 = reinterpret_cast<double2*>(y);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (divide) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
      double2 v = x2[i];
      v.x /= alpha; v.y /= alpha;
      y2[i] = v;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = x[n - 1] / alpha;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
      double2 v = x2[i];
      v.x *= alpha; v.y *= alpha;
      y2[i] = v;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = x[n - 1] * alpha;
  }
}

extern "C" int hipeig_scale(hipeig_ctx* c, int64_t n, double alpha, const double* x, double* y) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, alpha, 0, x, y);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hipeig_divide(hipeig_ctx* c, int64_t n, double alpha, const double* x, double* y) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, alpha, 1, x, y);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hipeig_normalize(hipeig_ctx* c, int64_t n, double* x, double* norm_out) {
  double nrm = 0.0;
  int rc = hipeig_nrm2(c, n, x, &nrm);
  if (rc) return rc;
  if (norm_out) *norm_out = nrm;
  // la.norm then array /= norm (numpyVector.py:76-78): a true division, not a
  // multiplication by the reciprocal, to stay on the reference's rounding.
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, nrm, 1, x, x);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
axpby_kernel(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  const int64_t n2 = n >> 1;
  const double2* x2 = reinterpret_cast<const double2*>(x);
  double2* y2 = reinterpret_cast<double2*>(y);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 xv = x2[i];
    double2 yv = y2[i];
    yv.x = a * xv.x + b * yv.x;
    yv.y = a * xv.y + b * yv.y;
    y2[i] = yv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = a * x[n - 1] + b * y[n - 1];
}

extern "C" int hipeig_axpby(hipeig_ctx* c, int64_t n, double a, const double* x, double b, double* y) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, x, b, y);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// ---- linear combination: out = (accumulate ? out : 0) + sum_j c[j]*v_j -----------------
template <int MB>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
lincomb_kernel(int64_t n, int m, PtrTable tab, CoefTable cf, const double* __restrict__ dcoef,
               double dscale, int accumulate, double* __restrict__ out) {
  double cj[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) cj[j] = (j < m) ? (dcoef ? dscale * dcoef[j] : cf.c[j]) : 0.0;
  const int64_t n2 = n >> 1;
  double2* o2 = reinterpret_cast<double2*>(out);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 acc = accumulate ? o2[i] : make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      if (j < m) {
        const double2 v = reinterpret_cast<const double2*>(tab.p[j])[i];
        acc.x = fma(cj[j], v.x, acc.x);
        acc.y = fma(cj[j], v.y, acc.y);
      }
    }
    o2[i] = acc;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double acc = accumulate ? out[n - 1] : 0.0;
    for (int j = 0; j < m; ++j) acc = fma(cj[j], tab.p[j][n - 1], acc);
    out[n - 1] = acc;
  }
}

// dcoef (device, scaled by dscale) overrides coeffs (host) when non-null; it holds k doubles.
static int lincomb_impl(hipeig_ctx* c, int64_t n, int k, const double* coeffs, const double* dcoef,
                        double dscale, const double* const* vecs, double* out, int accumulate_first) {
  if (n == 0) return 0;
  const int g = grid_stream(n);
  for (int j0 = 0; j0 < k; j0 += HIPEIG_MAX_COLS) {
    const int m = (k - j0 < HIPEIG_MAX_COLS) ? (k - j0) : HIPEIG_MAX_COLS;
    PtrTable tab;
    CoefTable cf;
    for (int j = 0; j < HIPEIG_MAX_COLS; ++j) {
      tab.p[j] = (j < m) ? vecs[j0 + j] : nullptr;
      cf.c[j] = (j < m && coeffs) ? coeffs[j0 + j] : 0.0;
    }
    const int acc = (j0 > 0) || accumulate_first;
    const double* dc = dcoef ? dcoef + j0 : nullptr;
    if (m <= 4)
      hipLaunchKernelGGL((lincomb_kernel<4>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, cf, dc, dscale, acc, out);
    else if (m <= 8)
      hipLaunchKernelGGL((lincomb_kernel<8>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, cf, dc, dscale, acc, out);
    else
      hipLaunchKernelGGL((lincomb_kernel<HIPEIG_MAX_COLS>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, cf, dc, dscale, acc, out);
    HIPEIG_CHECK(hipGetLastError());
  }
  return 0;
}

extern "C" int hipeig_lincomb(hipeig_ctx* c, int64_t n, int k, const double* coeffs,
                              const double* const* vecs, double* out) {
  HIPEIG_REQUIRE(k >= 1, "need at least one vector");
  for (int j = 0; j < k; ++j) HIPEIG_REQUIRE(vecs[j] != out, "out must not alias an input");
  return lincomb_impl(c, n, k, coeffs, nullptr, 1.0, vecs, out, 0);
}

// ---- block linear combination: outs[c] = sum_j C[j][c] * v_j, all k outputs in ONE pass ----
// (basisTransformation with a coefficient matrix, util_funcs.py:229-230: the tall-skinny product
// Y*C.)  Every thread keeps the KB output accumulators of its two rows in registers and streams the m
// inputs past them, so the inputs are read once instead of k times: (m + k) * 8N bytes.  Pointer
// table and coefficients are read from device memory with wave-uniform (scalar) loads.
struct OutTable {
  double* p[HIPEIG_MAX_COLS];
};

template <int KB>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
lincomb_block_kernel(int64_t n, int m, int k, const double* const* __restrict__ vecs,
                     const double* __restrict__ coef /* m x KB, row-major, zero-padded */, OutTable outs) {
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 acc[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) acc[c] = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int j = 0; j < m; ++j) {
      const double2 v = reinterpret_cast<const double2*>(vecs[j])[i];
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const double w = coef[j * KB + c];
        acc[c].x = fma(w, v.x, acc[c].x);
        acc[c].y = fma(w, v.y, acc[c].y);
      }
    }
#pragma unroll
    for (int c = 0; c < KB; ++c)
      if (c < k) reinterpret_cast<double2*>(outs.p[c])[i] = acc[c];
  }
  if ((n & 1) && blockIdx.x == 0 && (int)threadIdx.x < k) {
    const int c = threadIdx.x;
    double a = 0.0;
    for (int j = 0; j < m; ++j) a = fma(coef[j * KB + c], vecs[j][n - 1], a);
    outs.p[c][n - 1] = a;
  }
}

extern "C" int hipeig_lincomb_block(hipeig_ctx* c, int64_t n, int m, int k, const double* C, int ldc,
                                    const double* const* vecs, double* const* outs) {
  HIPEIG_REQUIRE(m >= 1 && k >= 1 && ldc >= k, "bad shape");
  HIPEIG_REQUIRE((size_t)m <= c->ptrs_count, "too many input vectors for one call");
  for (int j = 0; j < m; ++j)
    for (int q = 0; q < k; ++q) HIPEIG_REQUIRE(vecs[j] != outs[q], "an output must not alias an input");
  if (n == 0) return 0;
  // Sources of asynchronous copies are the context's PINNED staging buffers (h_ptrs, h_scalars), never
  // pageable memory that goes out of scope or is refilled while a copy may still be reading it.  Only this
  // function and hipeig_comm_setup_rows (which drains the stream) rewrite them from the host, so it is enough
  // to wait for the event recorded behind the previous call's last copy - not for the whole stream.
  HIPEIG_CHECK(hipEventSynchronize(c->ev_stage));
  for (int j = 0; j < m; ++j) c->h_ptrs[j] = vecs[j];
  HIPEIG_CHECK(hipMemcpyAsync((void*)c->d_ptrs, c->h_ptrs, sizeof(double*) * m, hipMemcpyHostToDevice, c->stream));
  const int g = grid_for(n, 2);                       // the per-workgroup coefficient prologue wants long-lived workgroups (measured: n/512 workgroups 8 % slower)
  double* dcoef = c->d_partials;                      // free here: this call has no reduction
  double* cf = c->h_scalars;
  size_t cf_used = 0;
  for (int c0 = 0; c0 < k; c0 += HIPEIG_MAX_COLS) {
    const int kk = (k - c0 < HIPEIG_MAX_COLS) ? (k - c0) : HIPEIG_MAX_COLS;
    const int KB = kk <= 4 ? 4 : kk <= 8 ? 8 : HIPEIG_MAX_COLS;
    if (cf_used + (size_t)m * KB > c->scalars_doubles) {            // staging area full: wait for the copies issued so far
      if (hipeig_sync_checked(c)) return 4;
      cf_used = 0;
    }
    HIPEIG_REQUIRE((size_t)m * KB <= c->scalars_doubles, "too many input vectors for the coefficient staging buffer");
    double* slot = cf + cf_used;
    for (int j = 0; j < m; ++j)
      for (int q = 0; q < KB; ++q) slot[(size_t)j * KB + q] = (q < kk) ? C[(size_t)j * ldc + c0 + q] : 0.0;
    HIPEIG_CHECK(hipMemcpyAsync(dcoef, slot, sizeof(double) * m * KB, hipMemcpyHostToDevice, c->stream));
    cf_used += (size_t)m * KB;
    OutTable ot;
    for (int q = 0; q < HIPEIG_MAX_COLS; ++q) ot.p[q] = (q < kk) ? outs[c0 + q] : nullptr;
    if (KB == 4)
      hipLaunchKernelGGL((lincomb_block_kernel<4>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, kk, c->d_ptrs, dcoef, ot);
    else if (KB == 8)
      hipLaunchKernelGGL((lincomb_block_kernel<8>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, kk, c->d_ptrs, dcoef, ot);
    else
      hipLaunchKernelGGL((lincomb_block_kernel<HIPEIG_MAX_COLS>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, kk, c->d_ptrs, dcoef, ot);
    HIPEIG_CHECK(hipGetLastError());
    dcoef += (size_t)m * KB;
  }
  HIPEIG_CHECK(hipEventRecord(c->ev_stage, c->stream));  // behind the last copy out of the staging buffers
  return 0;
}

extern "C" int hipeig_multi_axpy(hipeig_ctx* c, int64_t n, int m, const double* const* Y,
                                 const double* coef, double* x) {
  if (m == 0) return 0;
  return lincomb_impl(c, n, m, coef, nullptr, 1.0, Y, x, 1);
}

// ---- multi_dot: out[j] = <Y_j, x> -------------------------------------------------------
template <int MB>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
multi_dot_kernel(int64_t n, int m, PtrTable tab, const double* __restrict__ x,
                 double* __restrict__ partials, double* __restrict__ group_partials, unsigned* counters, double* __restrict__ out) {
  __shared__ double lds[HIPEIG_BLOCK];
  double acc[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) acc[j] = 0.0;
  const int64_t n2 = n >> 1;
  const double2* x2 = reinterpret_cast<const double2*>(x);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 xv = x2[i];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      if (j < m) {
        const double2 v = reinterpret_cast<const double2*>(tab.p[j])[i];
        acc[j] = fma(v.x, xv.x, acc[j]);
        acc[j] = fma(v.y, xv.y, acc[j]);
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < MB; ++j)
      if (j < m) acc[j] = fma(tab.p[j][n - 1], x[n - 1], acc[j]);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    const double r = wave_reduce_sum(acc[j]);
    if (lane == 0) lds[wid * MB + j] = r;
  }
  __syncthreads();
  if ((int)threadIdx.x < m) {
    const int j = threadIdx.x;
    agent_store_nowait(partials + (size_t)blockIdx.x * m + j, lds[j] + lds[MB + j] + lds[2 * MB + j] + lds[3 * MB + j]);
  }
  __syncthreads();                                   // lds is reused by the record sums
  finish_records(partials, group_partials, m, counters, lds, out);
}

// One launch for <= 16 columns: records [block][m] in the partial workspace (reused by the next launch: same stream),
// totals to out[0..m).
static int multi_dot_launch(hipeig_ctx* c, int64_t n, int m, const double* const* Y, const double* x, int g, double* out) {
  PtrTable tab;
  for (int j = 0; j < HIPEIG_MAX_COLS; ++j) tab.p[j] = (j < m) ? Y[j] : nullptr;
  unsigned* cnt = c->d_counters + 3 * HIPEIG_TICKET_WORDS;
  if (m <= 4)
    hipLaunchKernelGGL((multi_dot_kernel<4>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, x, c->d_partials, c->d_group_partials, cnt, out);
  else if (m <= 8)
    hipLaunchKernelGGL((multi_dot_kernel<8>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, x, c->d_partials, c->d_group_partials, cnt, out);
  else
    hipLaunchKernelGGL((multi_dot_kernel<HIPEIG_MAX_COLS>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, m, tab, x, c->d_partials, c->d_group_partials, cnt, out);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// Device-side result in ctx->d_scalars[0..m); optionally copied to the host.
static int multi_dot_impl(hipeig_ctx* c, int64_t n, int m, const double* const* Y, const double* x,
                          double* host_out) {
  HIPEIG_REQUIRE(m >= 1 && m <= HIPEIG_MAX_COLS * HIPEIG_MAX_COLS, "too many columns for one call");
  const int g = grid_records(c, n, "HIPEIG_MULTIDOT_PER_THREAD");
  double* target = record_target(c, host_out != nullptr);
  for (int j0 = 0; j0 < m; j0 += HIPEIG_MAX_COLS) {
    const int mm = (m - j0 < HIPEIG_MAX_COLS) ? (m - j0) : HIPEIG_MAX_COLS;
    if (multi_dot_launch(c, n, mm, Y + j0, x, g, target + j0)) return 1;
  }
  return records_to_host(c, m, host_out);
}

extern "C" int hipeig_multi_dot(hipeig_ctx* c, int64_t n, int m, const double* const* Y,
                                const double* x, double* out) {
  int done = 0;
  const int cap = HIPEIG_MAX_COLS * HIPEIG_MAX_COLS;
  while (done < m) {
    const int mm = (m - done < cap) ? (m - done) : cap;
    int rc = multi_dot_impl(c, n, mm, Y + done, x, out + done);
    if (rc) return rc;
    done += mm;
  }
  return 0;
}

// ---- Gram blocks on the matrix cores -------------------------------------------------------
// C[i][j] = sum_r A_i[r] * B_j[r]: v_mfma_f64_16x16x4_f64 with the ROW index r as the K dimension.
// One pass handles up to 32 columns of A and 32 of B (2 x 2 accumulator blocks of 16 x 16; the
// symmetric case A == B reads only its 32 columns and skips the lower block).  A workgroup stages a
// tile of 128 rows x all columns in LDS - every wave instruction is a coalesced 1 KiB read of ONE
// column (lane = row pair, 16 bytes) - then its four waves each take a quarter of the tile's rows
// as K-steps of 4.  LDS column stride 130 doubles (= 2 mod 32): the 32 lanes of an operand read
// (16 columns x 2 rows) hit 32 distinct 8-byte banks.  Two workgroups per CU overlap one's loads with
// the other's MFMAs.  Traffic: every column is read once per pass, (ma + mb) * 8N bytes against
// 8N * ma * (mb + 1) for mb multi_dot sweeps.  Per-workgroup partial blocks are summed in fixed order
// by the kernel's own last workgroups (finish_records above).
typedef double d4_t __attribute__((ext_vector_type(4)));
#define GRAM_ROWS 128
#define GRAM_LD 130
#define GRAM_MAX_WG 2048

struct PtrTable32 {
  const double* p[32];
};

template <int NA, int NB, bool SAME>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
gram_tile_kernel(int64_t n, int ma, int mb, PtrTable32 A, PtrTable32 B, double* __restrict__ partials,
                 double* __restrict__ group_partials, unsigned* counters, double* __restrict__ out) {
  extern __shared__ double gram_lds[];
  constexpr int CA = NA * 16, CB = SAME ? 0 : NB * 16;
  double* tA = gram_lds;
  double* tB = SAME ? gram_lds : gram_lds + CA * GRAM_LD;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // tell the compiler it is wave-uniform
  d4_t acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  const int64_t ntiles = (n + GRAM_ROWS - 1) / GRAM_ROWS;
  // The next tile is loaded into registers while the MFMAs of the current one run out of LDS.
  constexpr int NQ = (CA + CB) / 4;
  double2 pre[NQ];
  const double* srcs[NQ];                            // this wave's columns (wave-uniform: scalar registers)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = wid + 4 * q;
    srcs[q] = (col < CA) ? (col < ma ? A.p[col] : nullptr) : (col - CA < mb ? B.p[col - CA] : nullptr);
  }
#define GRAM_FETCH(T)                                                                                   \
  {                                                                                                     \
    const int64_t r = (T) * GRAM_ROWS + 2 * lane;                                                       \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
      const double* src = srcs[q];                                                                      \
      double2 v = make_double2(0.0, 0.0);                                                               \
      if (src && (T) < ntiles) {                                                                        \
        if (r + 1 < n) {                                                                                \
          typedef double gram_v2 __attribute__((ext_vector_type(2)));                                   \
          const gram_v2 w = __builtin_nontemporal_load(reinterpret_cast<const gram_v2*>(src + r));      \
          v = make_double2(w.x, w.y);                                                                   \
        } else if (r < n) v.x = src[r];                                                                 \
      }                                                                                                 \
      pre[q] = v;                                                                                       \
    }                                                                                                   \
  }
  GRAM_FETCH((int64_t)blockIdx.x)
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {           // uniform trip count per workgroup
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      *reinterpret_cast<double2*>(gram_lds + (wid + 4 * q) * GRAM_LD + 2 * lane) = pre[q];
    __syncthreads();
    GRAM_FETCH(t + gridDim.x)
#pragma unroll
    for (int s = 0; s < GRAM_ROWS / 16; ++s) {
      const int row = 4 * (wid * (GRAM_ROWS / 16) + s) + (lane >> 4);
      double a[NA], b[NB];
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = tA[(i * 16 + (lane & 15)) * GRAM_LD + row];
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = SAME ? a[j] : tB[(j * 16 + (lane & 15)) * GRAM_LD + row];
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (!SAME || j >= i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef GRAM_FETCH
  // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.  Sum the four waves'
  // blocks through LDS (the tile area is free now): red[wave][block][256].
  double* red = gram_lds;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        red[(wid * NA * NB + i * NB + j) * 256 + ((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[i][j][reg];
  __syncthreads();
  const int e = threadIdx.x;                       // 256 threads <-> 256 elements of a block
#pragma unroll
  for (int blk = 0; blk < NA * NB; ++blk) {
    const double v = red[(0 * NA * NB + blk) * 256 + e] + red[(1 * NA * NB + blk) * 256 + e] +
                     red[(2 * NA * NB + blk) * 256 + e] + red[(3 * NA * NB + blk) * 256 + e];
    agent_store_nowait(partials + (size_t)blockIdx.x * (NA * NB * 256) + blk * 256 + e, v);
  }
  __syncthreads();                                   // the LDS block area is reused by the record sums
  finish_records(partials, group_partials, NA * NB * 256, counters, gram_lds, out);
}

template <int NA, int NB, bool SAME>
static int gram_tile_launch(hipeig_ctx* c, int64_t n, int na, int nb, const PtrTable32& ta, const PtrTable32& tb, int& g) {
  constexpr size_t tile = (size_t)(NA * 16 + (SAME ? 0 : NB * 16)) * GRAM_LD * sizeof(double);
  constexpr size_t red = (size_t)4 * NA * NB * 256 * sizeof(double);
  constexpr size_t lds = tile > red ? tile : red;
  static bool configured = false;
  {
    // workgroups per CU that fit the LDS (at most 8), capped by the partial-block workspace
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t cap = (int64_t)c->num_cu * per_cu;
    const int64_t room = (int64_t)(c->partials_doubles / ((size_t)NA * NB * 256));
    if (cap > room) cap = room;
    if (g > cap) g = (int)cap;
  }
  if (!configured) {
    HIPEIG_CHECK(hipFuncSetAttribute((const void*)gram_tile_kernel<NA, NB, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  hipLaunchKernelGGL((gram_tile_kernel<NA, NB, SAME>), dim3(g), dim3(HIPEIG_BLOCK), lds, c->stream, n, na, nb, ta, tb, c->d_partials,
                     c->d_group_partials, c->d_counters + 3 * HIPEIG_TICKET_WORDS, record_target(c, true));
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hipeig_gram(hipeig_ctx* c, int64_t n, int ma, const double* const* A, int mb,
                           const double* const* B, double* out) {
  HIPEIG_REQUIRE(ma >= 1 && mb >= 1, "empty set");
  if (ma < 3 || mb < 3 || n < 64) {
    // tiny blocks: one multi_dot sweep per column of B
    double* col = (double*)malloc(sizeof(double) * ma);
    HIPEIG_REQUIRE(col != nullptr, "out of host memory");
    int rc = 0;
    for (int j = 0; j < mb && !rc; ++j) {
      rc = hipeig_multi_dot(c, n, ma, A, B[j], col);
      for (int i = 0; i < ma; ++i) out[(size_t)i * mb + j] = col[i];
    }
    free(col);
    return rc;
  }
  bool symmetric = (ma == mb);
  for (int i = 0; i < ma && symmetric; ++i) symmetric = (A[i] == B[i]);
  int64_t g64 = (n + GRAM_ROWS - 1) / GRAM_ROWS;
  if (g64 > GRAM_MAX_WG) g64 = GRAM_MAX_WG;         // upper bound; each launch lowers it to what fits a CU
  const int g = (int)g64;
  std::vector<double> blk(4 * 256);
  for (int ia = 0; ia < ma; ia += 32) {
    for (int ib = 0; ib < mb; ib += 32) {
      if (symmetric && ib < ia) continue;           // mirrored from the upper block below
      const int na = (ma - ia < 32) ? ma - ia : 32, nb = (mb - ib < 32) ? mb - ib : 32;
      const int NA = (na + 15) / 16, NB = (nb + 15) / 16;
      PtrTable32 ta, tb;
      bool same = (na == nb);
      for (int q = 0; q < 32; ++q) {
        ta.p[q] = (q < na) ? A[ia + q] : nullptr;
        tb.p[q] = (q < nb) ? B[ib + q] : nullptr;
        if (ta.p[q] != tb.p[q]) same = false;
      }
      int rc, gl = g;                              // the launch lowers gl to the grid it used
      if (same && NA == 1) rc = gram_tile_launch<1, 1, true>(c, n, na, nb, ta, tb, gl);
      else if (same) rc = gram_tile_launch<2, 2, true>(c, n, na, nb, ta, tb, gl);
      else if (NA == 1 && NB == 1) rc = gram_tile_launch<1, 1, false>(c, n, na, nb, ta, tb, gl);
      else if (NA == 1) rc = gram_tile_launch<1, 2, false>(c, n, na, nb, ta, tb, gl);
      else if (NB == 1) rc = gram_tile_launch<2, 1, false>(c, n, na, nb, ta, tb, gl);
      else rc = gram_tile_launch<2, 2, false>(c, n, na, nb, ta, tb, gl);
      if (rc) return rc;
      if (records_to_host(c, NA * NB * 256, blk.data())) return 4;
      for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
          const int bi = i >> 4, bj = j >> 4;
          double v;
          if (same && bj < bi) v = blk[(bj * NB + bi) * 256 + (j & 15) * 16 + (i & 15)];   // lower block of a symmetric pass
          else v = blk[(bi * NB + bj) * 256 + (i & 15) * 16 + (j & 15)];
          out[(size_t)(ia + i) * mb + (ib + j)] = v;
          if (symmetric && ib > ia) out[(size_t)(ib + j) * mb + (ia + i)] = v;
        }
    }
  }
  return 0;
}

// ---- Gram-Schmidt ------------------------------------------------------------------------
// The reference's sequential MGS (numpyVector.py:132-139: per q  t1 = x.q, t2 = q.q, x = x - q*(t1/t2)) as ONE kernel per
// basis vector plus one: kernel j applies the update of q_{j-1} with the totals kernel j-1 left and, on the updated x in
// the same pass, forms the two dots with q_j; the kernel behind the last vector applies the last update and forms x.x,
// the inner product the lindep test needs (:140).  Round 3 ran dots, a 2-workgroup sum and the update as three launches
// per vector and read x twice: 40N bytes per vector against 32N now.  Same element-wise roundings as the reference:
// x + (-(q * coef)), two roundings.
__global__ void __launch_bounds__(HIPEIG_BLOCK)
mgs_sweep_kernel(int64_t n, double* __restrict__ x, const double* __restrict__ q_prev, const double* __restrict__ t_prev,
                 const double* __restrict__ q_next, double* __restrict__ partials, double* __restrict__ group_partials,
                 unsigned* counters, double* __restrict__ out) {
  __shared__ double lds[HIPEIG_BLOCK];
  const double coef = q_prev ? t_prev[0] / t_prev[1] : 0.0;
  const int64_t n2 = n >> 1;
  double2* x2 = reinterpret_cast<double2*>(x);
  const double2* p2 = reinterpret_cast<const double2*>(q_prev);
  const double2* q2 = reinterpret_cast<const double2*>(q_next);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double a = 0.0, b = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 xv = x2[i];
    if (q_prev) {
      const double2 pv = p2[i];
      xv.x = add_rn(xv.x, -mul_rn(pv.x, coef));
      xv.y = add_rn(xv.y, -mul_rn(pv.y, coef));
      x2[i] = xv;
    }
    if (q_next) {
      const double2 qv = q2[i];
      a = fma(xv.x, qv.x, a); a = fma(xv.y, qv.y, a);
      b = fma(qv.x, qv.x, b); b = fma(qv.y, qv.y, b);
    } else {
      a = fma(xv.x, xv.x, a); a = fma(xv.y, xv.y, a);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double xv = x[n - 1];
    if (q_prev) { xv = add_rn(xv, -mul_rn(q_prev[n - 1], coef)); x[n - 1] = xv; }
    if (q_next) { a = fma(xv, q_next[n - 1], a); b = fma(q_next[n - 1], q_next[n - 1], b); }
    else a = fma(xv, xv, a);
  }
  a = block_reduce_sum(a, lds);
  b = block_reduce_sum(b, lds);
  if (threadIdx.x == 0) {
    agent_store_nowait(partials + 2 * blockIdx.x, a);
    agent_store_nowait(partials + 2 * blockIdx.x + 1, b);
  }
  finish_records(partials, group_partials, 2, counters, lds, out);
}

// ---- sequential MGS projection with the coefficients kept on the device -------------------
// w <- w - sum_j c_j V_j with c_j = <V_j, w_current> taken one column after the other (the
// Arnoldi orthogonalisation of GMRES-type solvers: scipy _fgmres, the loop the reference's
// gcrotmk runs).  Two launches per column and no host round trip: the update kernel sums the
// dot kernel's partials in its prologue (identical value in every workgroup), workgroup 0 stores
// the coefficient, and all m coefficients are copied back once at the end.
// `npart` = 1 means p[0..nval) already holds reduced (all-reduced) values.
__global__ void __launch_bounds__(HIPEIG_BLOCK)
mgsp_dot_kernel(int64_t n, const double* __restrict__ v, const double* __restrict__ w, double* __restrict__ partials) {
  __shared__ double lds[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a = fma(v[i], w[i], a);
  a = block_reduce_sum(a, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = a;
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
mgsp_update_kernel(int64_t n, const double* __restrict__ p, int npart, const double* __restrict__ v,
                   double* __restrict__ w, double* __restrict__ coef_out) {
  __shared__ double lds[4];
  const double cj = (npart == 1) ? p[0] : block_sum_partials(p, npart, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) *coef_out = cj;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] = fma(-cj, v[i], w[i]);
}

// complex vectors as (re, im) pairs: c = conj(v).w = (vr.wr + vi.wi) + i (vr.wi - vi.wr)
__global__ void __launch_bounds__(HIPEIG_BLOCK)
mgsp_pair_dot_kernel(int64_t n, const double* __restrict__ vr, const double* __restrict__ vi,
                     const double* __restrict__ wr, const double* __restrict__ wi, double* __restrict__ partials) {
  __shared__ double lds[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double re = 0.0, im = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double a = vr[i], b = vi[i], x = wr[i], y = wi[i];
    re = fma(a, x, re); re = fma(b, y, re);
    im = fma(a, y, im); im = fma(-b, x, im);
  }
  re = block_reduce_sum(re, lds);
  im = block_reduce_sum(im, lds);
  if (threadIdx.x == 0) { partials[blockIdx.x] = re; partials[gridDim.x + blockIdx.x] = im; }
}

__global__ void __launch_bounds__(HIPEIG_BLOCK)
mgsp_pair_update_kernel(int64_t n, const double* __restrict__ p, int npart, int pstride,
                        const double* __restrict__ vr, const double* __restrict__ vi,
                        double* __restrict__ wr, double* __restrict__ wi, double* __restrict__ coef_out) {
  __shared__ double lds[4];
  const double cr = (npart == 1) ? p[0] : block_sum_partials(p, npart, lds);
  const double ci = (npart == 1) ? p[1] : block_sum_partials(p + pstride, npart, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) { coef_out[0] = cr; coef_out[1] = ci; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double a = vr[i], b = vi[i];
    wr[i] = wr[i] - (cr * a - ci * b);           // w -= c * v
    wi[i] = wi[i] - (cr * b + ci * a);
  }
}

__global__ void mgsp_reduce_kernel(const double* __restrict__ p, int npart, int nval, int pstride, double* __restrict__ out) {
  __shared__ double lds[4];
  for (int k = 0; k < nval; ++k) {
    const double v = block_sum_partials(p + (size_t)k * pstride, npart, lds);
    if (threadIdx.x == 0) out[k] = v;
  }
}

extern "C" int hipeig_mgs_project(hipeig_ctx* c, int64_t n, int m, const double* const* V, double* w, double* coeffs) {
  HIPEIG_REQUIRE(m >= 0 && m <= 1024 && coeffs, "bad arguments");
  if (m == 0) return 0;
  const int g = grid_for(n, 4);
  double* dcoef = c->d_scalars + 2560;                 // m doubles
  double* red = c->d_scalars + 3600;
  for (int j = 0; j < m; ++j) {
    hipLaunchKernelGGL(mgsp_dot_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, V[j], w, c->d_partials);
    const double* p = c->d_partials;
    int npart = g;
    if (c->collectives) {
      hipLaunchKernelGGL(mgsp_reduce_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, c->d_partials, g, 1, g, red);
      if (hipeig_allreduce_sum(c, red, 1)) return 4;
      p = red; npart = 1;
    }
    hipLaunchKernelGGL(mgsp_update_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, p, npart, V[j], w, dcoef + j);
  }
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dcoef, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(coeffs, c->h_scalars, sizeof(double) * m);
  return 0;
}

extern "C" int hipeig_pair_mgs_project(hipeig_ctx* c, int64_t n, int m, const double* const* Vre,
                                       const double* const* Vim, double* wre, double* wim, double* coeffs) {
  HIPEIG_REQUIRE(m >= 0 && m <= 512 && coeffs, "bad arguments");
  if (m == 0) return 0;
  const int g = grid_for(n, 4);
  double* dcoef = c->d_scalars + 2560;                 // 2m doubles
  double* red = c->d_scalars + 3600;
  for (int j = 0; j < m; ++j) {
    hipLaunchKernelGGL(mgsp_pair_dot_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, Vre[j], Vim[j], wre, wim, c->d_partials);
    const double* p = c->d_partials;
    int npart = g;
    if (c->collectives) {
      hipLaunchKernelGGL(mgsp_reduce_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, c->d_partials, g, 2, g, red);
      if (hipeig_allreduce_sum(c, red, 2)) return 4;
      p = red; npart = 1;
    }
    hipLaunchKernelGGL(mgsp_pair_update_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, p, npart, g,
                       Vre[j], Vim[j], wre, wim, dcoef + 2 * j);
  }
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dcoef, sizeof(double) * 2 * m, hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(coeffs, c->h_scalars, sizeof(double) * 2 * m);
  return 0;
}

// ---- one Arnoldi step of a GMRES-type solver with a single host round trip ---------------------
// scipy _fgmres per inner iteration (the loop behind the reference's gcrotmk, numpyVector.py:161):
//   w_norm = ||w||;  for v in [C..., V...]: h = <v, w>, w -= h v;  h_last = ||w||;  w *= 1/h_last (if finite)
// Everything stays on the device; out = [ ||w||^2 before, h_0 .. h_{m-1}, ||w||^2 after ] comes back in one
// copy (for pairs: complex h as (re, im), so 2m + 2 doubles).  The host takes the square roots, checks the
// breakdown condition and updates its small QR factorisation.
__global__ void __launch_bounds__(HIPEIG_BLOCK)
sumsq_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ partials) {
  __shared__ double lds[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    s = fma(a[i], a[i], s);
    if (b) s = fma(b[i], b[i], s);
  }
  s = block_reduce_sum(s, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// w *= 1/sqrt(ss[0]) when that factor is finite (scipy: alpha = 1/h; if isfinite(alpha): w = scal(alpha, w))
__global__ void __launch_bounds__(HIPEIG_BLOCK)
scale_by_inv_norm_kernel(int64_t n, const double* __restrict__ ss, double* __restrict__ a, double* __restrict__ b) {
  const double alpha = 1.0 / sqrt(ss[0]);
  if (!isfinite(alpha)) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    a[i] *= alpha;
    if (b) b[i] *= alpha;
  }
}

// Single-GPU form of the step: the update with column j and the dot product with column j+1 share one
// pass (and one launch), the two norms ride on the first and the last pass: m + 2 launches and three vector
// passes per column instead of 2m + 5 launches and five passes.  (A partitioned run keeps the unfused path: it
// needs an all-reduce between a dot and its update.  The two paths assign elements to threads differently, so their
// coefficients agree to rounding, not bit for bit.)
// Workspace (doubles, g = grid): two phase buffers of 3g - [re | im | ||w||^2 after] - used alternately, and
// g for ||w||^2 before at offset 6g.
template <bool PAIR>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
arnoldi_first_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                     const double* __restrict__ wre, const double* __restrict__ wim, double* __restrict__ partials) {
  __shared__ double lds[4];
  const int g = gridDim.x;
  const int64_t stride = (int64_t)g * blockDim.x;
  double ss = 0.0, re = 0.0, im = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = wre[i], y = PAIR ? wim[i] : 0.0;
    ss = fma(x, x, ss);
    if (PAIR) ss = fma(y, y, ss);
    if (a) {
      const double p = a[i], q = PAIR ? b[i] : 0.0;
      re = fma(p, x, re);
      if (PAIR) { re = fma(q, y, re); im = fma(p, y, im); im = fma(-q, x, im); }
    }
  }
  ss = block_reduce_sum(ss, lds);
  re = block_reduce_sum(re, lds);
  if (PAIR) im = block_reduce_sum(im, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = re;
    if (PAIR) partials[g + blockIdx.x] = im;
    partials[6 * g + blockIdx.x] = ss;
  }
}

// The kernel GCROT's orthogonalisation lives in (63 % of the device time of a complex contour solve at N = 1e6,
// rocprofv3 of tools/experiments/gcrot_complex_solve.py).  16-byte accesses, two of them per stream in flight per
// thread and every load of a trip issued before its first store; the column being subtracted is read for the last
// time here (non-temporal), the next column stays cached for the next launch.
typedef double arn_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 arn_ld(const double* p, int64_t i2, bool nt) {
  if (nt) { const arn_d2 v = __builtin_nontemporal_load(reinterpret_cast<const arn_d2*>(p) + i2); return make_double2(v.x, v.y); }
  return reinterpret_cast<const double2*>(p)[i2];
}

template <bool PAIR>
struct ArnoldiLane {                  // one 16-byte slot (two consecutive elements) of every stream
  double2 p, q, a2p, a2q, x, y;
  __device__ __forceinline__ void load(int64_t i2, const double* a, const double* b, const double* a2, const double* b2,
                                       const double* wre, const double* wim, int last) {
    p = arn_ld(a, i2, true);
    if (PAIR) q = arn_ld(b, i2, true);
    if (!last) { a2p = arn_ld(a2, i2, false); if (PAIR) a2q = arn_ld(b2, i2, false); }
    x = arn_ld(wre, i2, false);
    if (PAIR) y = arn_ld(wim, i2, false);
  }
  __device__ __forceinline__ void update(double cr, double ci) {
    if (PAIR) {
      x.x = x.x - (cr * p.x - ci * q.x); y.x = y.x - (cr * q.x + ci * p.x);       // w -= c * v
      x.y = x.y - (cr * p.y - ci * q.y); y.y = y.y - (cr * q.y + ci * p.y);
    } else {
      x.x = fma(-cr, p.x, x.x); x.y = fma(-cr, p.y, x.y);
    }
  }
  __device__ __forceinline__ void store(int64_t i2, double* wre, double* wim) const {
    reinterpret_cast<double2*>(wre)[i2] = x;
    if (PAIR) reinterpret_cast<double2*>(wim)[i2] = y;
  }
  __device__ __forceinline__ void accumulate(int last, double& re, double& im, double& ss) const {
    if (last) {
      ss = fma(x.x, x.x, ss); ss = fma(x.y, x.y, ss);
      if (PAIR) { ss = fma(y.x, y.x, ss); ss = fma(y.y, y.y, ss); }
    } else {
      re = fma(a2p.x, x.x, re); re = fma(a2p.y, x.y, re);
      if (PAIR) {
        re = fma(a2q.x, y.x, re); re = fma(a2q.y, y.y, re);
        im = fma(a2p.x, y.x, im); im = fma(-a2q.x, x.x, im);
        im = fma(a2p.y, y.y, im); im = fma(-a2q.y, x.y, im);
      }
    }
  }
};

template <bool PAIR>
__global__ void __launch_bounds__(HIPEIG_BLOCK)
arnoldi_column_kernel(int64_t n, const double* __restrict__ pin, double* __restrict__ pout,
                      const double* __restrict__ a, const double* __restrict__ b,
                      const double* __restrict__ a2, const double* __restrict__ b2, int last,
                      double* __restrict__ wre, double* __restrict__ wim, double* __restrict__ coef_out) {
  __shared__ double lds[4];
  const int g = gridDim.x;
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)g * blockDim.x;
  // the first trip's loads do not depend on the coefficient: issue them BEFORE the prologue reduces the previous
  // launch's partial sums, so that the reduction (L2 reads + two barriers) hides behind their HBM latency
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  ArnoldiLane<PAIR> L0, L1;
  bool has0 = i < n2, has1 = i + stride < n2;
  if (has0) L0.load(i, a, b, a2, b2, wre, wim, last);
  if (has1) L1.load(i + stride, a, b, a2, b2, wre, wim, last);
  const double cr = block_sum_partials(pin, g, lds);
  const double ci = PAIR ? block_sum_partials(pin + g, g, lds) : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    coef_out[0] = cr;
    if (PAIR) coef_out[1] = ci;
  }
  double re = 0.0, im = 0.0, ss = 0.0;
  while (has0) {
    L0.update(cr, ci);
    L0.store(i, wre, wim);
    L0.accumulate(last, re, im, ss);
    if (has1) {
      L1.update(cr, ci);
      L1.store(i + stride, wre, wim);
      L1.accumulate(last, re, im, ss);
    }
    i += 2 * stride;
    has0 = i < n2; has1 = i + stride < n2;
    if (has0) L0.load(i, a, b, a2, b2, wre, wim, last);
    if (has1) L1.load(i + stride, a, b, a2, b2, wre, wim, last);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {           // odd length: the last element
    const int64_t i = n - 1;
    double x, y = 0.0;
    if (PAIR) {
      const double p = a[i], q = b[i];
      x = wre[i] - (cr * p - ci * q);
      y = wim[i] - (cr * q + ci * p);
      wre[i] = x; wim[i] = y;
    } else {
      x = fma(-cr, a[i], wre[i]);
      wre[i] = x;
    }
    if (last) {
      ss = fma(x, x, ss);
      if (PAIR) ss = fma(y, y, ss);
    } else {
      const double p = a2[i], q = PAIR ? b2[i] : 0.0;
      re = fma(p, x, re);
      if (PAIR) { re = fma(q, y, re); im = fma(p, y, im); im = fma(-q, x, im); }
    }
  }
  if (last) {
    ss = block_reduce_sum(ss, lds);
    if (threadIdx.x == 0) pout[2 * g + blockIdx.x] = ss;
  } else {
    re = block_reduce_sum(re, lds);
    if (PAIR) im = block_reduce_sum(im, lds);
    if (threadIdx.x == 0) {
      pout[blockIdx.x] = re;
      if (PAIR) pout[g + blockIdx.x] = im;
    }
  }
}

// ||w||^2 before / after to dres, then w *= 1/||w|| when that factor is finite
__global__ void __launch_bounds__(HIPEIG_BLOCK)
arnoldi_last_kernel(int64_t n, const double* __restrict__ p_before, const double* __restrict__ p_after,
                    double* __restrict__ wre, double* __restrict__ wim, double* __restrict__ d_before,
                    double* __restrict__ d_after) {
  __shared__ double lds[4];
  const int g = gridDim.x;
  const double sb = block_sum_partials(p_before, g, lds);
  const double sa = block_sum_partials(p_after, g, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) { *d_before = sb; *d_after = sa; }
  const double alpha = 1.0 / sqrt(sa);
  if (!isfinite(alpha)) return;
  const int64_t stride = (int64_t)g * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    wre[i] *= alpha;
    if (wim) wim[i] *= alpha;
  }
}

// Short vectors (the reference's own problem sizes: n = 100 ... 4000): the whole step - norm, the sequential sweep over
// all m columns, norm, scaling - in ONE workgroup and ONE launch, the vector being orthogonalised held in registers.
// At these lengths a launch per column is pure dispatch latency (~5 us each, 20-60 columns per step); a workgroup-wide
// reduction costs two barriers.  Same order of operations as the column kernels (sequential MGS, SciPy's _fgmres).
#define ARN_SMALL_THREADS 1024
#define ARN_SMALL_E 8                       // elements per thread: n <= 8192 (16 would spill the pair form)
#define ARN_SMALL_MAXCOLS 64
struct ArnSmallCols { const double* re[ARN_SMALL_MAXCOLS]; const double* im[ARN_SMALL_MAXCOLS]; };

// sums of (a, b) over the workgroup, returned to every thread; fixed tree.  lds: 2 x 16 doubles
__device__ __forceinline__ void arn_small_reduce2(double& a, double& b, double* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();                                   // the previous reduction's readers are done with lds
  if (lane == 0) { lds[wid] = a; lds[16 + wid] = b; }
  __syncthreads();
  a = lds[0]; b = lds[16];
  for (int w = 1; w < ARN_SMALL_THREADS / 64; ++w) { a += lds[w]; b += lds[16 + w]; }
}

// column tables of the step: in the kernel arguments (one step per launch) or in LDS (one step per WORKGROUP, below)
struct ArnColsArg {
  const ArnSmallCols& V;
  __device__ __forceinline__ const double* re(int j) const { return V.re[j]; }
  __device__ __forceinline__ const double* im(int j) const { return V.im[j]; }
};
struct ArnColsLds {
  const double* const* tab;                          // [2][ARN_SMALL_MAXCOLS]
  __device__ __forceinline__ const double* re(int j) const { return tab[j]; }
  __device__ __forceinline__ const double* im(int j) const { return tab[ARN_SMALL_MAXCOLS + j]; }
};

template <bool PAIR, class Cols>
__device__ __forceinline__ void arnoldi_small_body(int n, int m, const Cols& V, double* __restrict__ wre, double* __restrict__ wim,
                                                   double* __restrict__ dres, double* lds) {
  const int W = PAIR ? 2 : 1;
  double x[ARN_SMALL_E], y[ARN_SMALL_E];
  double ss = 0.0, zero = 0.0;
#pragma unroll
  for (int e = 0; e < ARN_SMALL_E; ++e) {
    const int i = threadIdx.x + e * ARN_SMALL_THREADS;
    x[e] = i < n ? wre[i] : 0.0;
    y[e] = (PAIR && i < n) ? wim[i] : 0.0;
    ss = fma(x[e], x[e], ss);
    if (PAIR) ss = fma(y[e], y[e], ss);
  }
  arn_small_reduce2(ss, zero, lds);
  if (threadIdx.x == 0) dres[0] = ss;
  for (int j = 0; j < m; ++j) {
    const double* __restrict__ vr = V.re(j);
    const double* __restrict__ vi = PAIR ? V.im(j) : nullptr;
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int e = 0; e < ARN_SMALL_E; ++e) {
      const int i = threadIdx.x + e * ARN_SMALL_THREADS;
      const double p = i < n ? vr[i] : 0.0;
      const double q = (PAIR && i < n) ? vi[i] : 0.0;
      re = fma(p, x[e], re);
      if (PAIR) { re = fma(q, y[e], re); im = fma(p, y[e], im); im = fma(-q, x[e], im); }
    }
    arn_small_reduce2(re, im, lds);                  // c = conj(v) . w, the same value in every thread
    if (threadIdx.x == 0) { dres[1 + W * j] = re; if (PAIR) dres[2 + W * j] = im; }
#pragma unroll
    for (int e = 0; e < ARN_SMALL_E; ++e) {          // the column again (L1 / L2 at these lengths): registers hold only w
      const int i = threadIdx.x + e * ARN_SMALL_THREADS;
      const double p = i < n ? vr[i] : 0.0;
      const double q = (PAIR && i < n) ? vi[i] : 0.0;
      if (PAIR) {
        const double nx = x[e] - (re * p - im * q);              // w -= c * v
        y[e] = y[e] - (re * q + im * p);
        x[e] = nx;
      } else {
        x[e] = fma(-re, p, x[e]);
      }
    }
  }
  double sa = 0.0;
  zero = 0.0;
#pragma unroll
  for (int e = 0; e < ARN_SMALL_E; ++e) { sa = fma(x[e], x[e], sa); if (PAIR) sa = fma(y[e], y[e], sa); }
  arn_small_reduce2(sa, zero, lds);
  if (threadIdx.x == 0) dres[1 + W * m] = sa;
  const double alpha = 1.0 / sqrt(sa);
  const bool scale = isfinite(alpha);
#pragma unroll
  for (int e = 0; e < ARN_SMALL_E; ++e) {
    const int i = threadIdx.x + e * ARN_SMALL_THREADS;
    if (i < n) {
      wre[i] = scale ? x[e] * alpha : x[e];
      if (PAIR) wim[i] = scale ? y[e] * alpha : y[e];
    }
  }
}

template <bool PAIR>
__global__ void __launch_bounds__(ARN_SMALL_THREADS)
arnoldi_small_kernel(int n, int m, ArnSmallCols V, double* __restrict__ wre, double* __restrict__ wim, double* __restrict__ dres) {
  __shared__ double lds[32];
  arnoldi_small_body<PAIR>(n, m, ArnColsArg{V}, wre, wim, dres, lds);
}

// Several independent steps in ONE launch, a workgroup each (the right-hand sides of a lock-step block solve at lengths
// where a step is a single workgroup: launched one after the other they would use one CU of 256 in turn).  The items sit
// in pinned host memory the device reads directly; each workgroup copies its column table to LDS and writes its scalars
// straight into its pinned result slot - no copy in either direction is enqueued.
struct ArnBatchItem {
  int m, pad;
  double* wre; double* wim;
  const double* re[ARN_SMALL_MAXCOLS];
  const double* im[ARN_SMALL_MAXCOLS];
};
#define ARN_SLOT_DOUBLES 128

__global__ void __launch_bounds__(ARN_SMALL_THREADS)
arnoldi_small_batch_kernel(int n, const ArnBatchItem* __restrict__ items, double* __restrict__ slots) {
  __shared__ double lds[32];
  __shared__ const double* tab[2 * ARN_SMALL_MAXCOLS];
  const ArnBatchItem* it = items + blockIdx.x;
  const int m = it->m;
  for (int j = threadIdx.x; j < m; j += ARN_SMALL_THREADS) { tab[j] = it->re[j]; tab[ARN_SMALL_MAXCOLS + j] = it->im[j]; }
  double* wre = it->wre;
  double* wim = it->wim;
  __syncthreads();
  arnoldi_small_body<true>(n, m, ArnColsLds{tab}, wre, wim, slots + (size_t)blockIdx.x * ARN_SLOT_DOUBLES, lds);
}

// the step is ONE workgroup (arnoldi_small_kernel): it only WRITES its scalars, so they can go straight to mapped host memory
static bool arnoldi_is_small(int64_t n, int m) {
  static const bool small_on = !(getenv("HIPEIG_ARNOLDI_SMALL") && atoi(getenv("HIPEIG_ARNOLDI_SMALL")) == 0);
  return small_on && n <= (int64_t)ARN_SMALL_THREADS * ARN_SMALL_E && m <= ARN_SMALL_MAXCOLS;
}

template <bool PAIR>
static int arnoldi_fused(hipeig_ctx* c, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                         double* wre, double* wim, double* dres) {
  if (arnoldi_is_small(n, m)) {
    ArnSmallCols V;
    for (int j = 0; j < ARN_SMALL_MAXCOLS; ++j) {
      V.re[j] = j < m ? Vre[j] : nullptr;
      V.im[j] = (PAIR && j < m) ? Vim[j] : nullptr;
    }
    hipLaunchKernelGGL((arnoldi_small_kernel<PAIR>), dim3(1), dim3(ARN_SMALL_THREADS), 0, c->stream, (int)n, m, V, wre, wim, dres);
    HIPEIG_CHECK(hipGetLastError());
    return 0;
  }
  const int g = grid_for(n, 4);
  const int W = PAIR ? 2 : 1;
  double* P = c->d_partials;
  hipLaunchKernelGGL((arnoldi_first_kernel<PAIR>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n,
                     m > 0 ? Vre[0] : nullptr, (PAIR && m > 0) ? Vim[0] : nullptr, wre, wim, P);
  for (int j = 0; j < m; ++j) {
    const int last = (j + 1 == m);
    hipLaunchKernelGGL((arnoldi_column_kernel<PAIR>), dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n,
                       P + (size_t)(j & 1) * 3 * g, P + (size_t)((j + 1) & 1) * 3 * g, Vre[j], PAIR ? Vim[j] : nullptr,
                       last ? nullptr : Vre[j + 1], (PAIR && !last) ? Vim[j + 1] : nullptr, last, wre, wim, dres + 1 + W * j);
  }
  const double* p_after = (m == 0) ? P + 6 * g : P + (size_t)(m & 1) * 3 * g + 2 * g;
  hipLaunchKernelGGL(arnoldi_last_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, P + 6 * g, p_after, wre, wim,
                     dres, dres + 1 + W * m);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// ---- several columns per pass (option; hipeig_arnoldi_step_p) ------------------------------------------------------
// The sweep above reads w once per column.  Here a pass applies the updates of a whole BLOCK of P = 4 columns and, in
// the same pass, forms everything the next block needs: its dot products with the updated w AND the Gram entries
// G_kl = <V_k, V_l> (l < k) of its own columns, from which the coefficients of the sequential sweep follow exactly,
//   h_0 = <V_0, w>,   h_k = <V_k, w - sum_{l<k} h_l V_l> = <V_k, w> - sum_{l<k} h_l G_kl .
// Algebraically this IS the modified Gram-Schmidt sweep of scipy's _fgmres, column after column; it differs in
// rounding (the G terms are accumulated sums instead of being folded into w element by element), which is why it is an
// option and the one-column form stays the default (iteration-count parity with scipy.sparse.linalg.gcrotmk).
// Traffic per column: (2P + 2) / P = 2.5 vector streams instead of 4, and a launch per FOUR columns.
// All sums finish in the kernel's last workgroup (common.h), so the next launch's prologue reads ~20 doubles.
#ifndef ARN_P
#define ARN_P 4
#endif
#ifndef ARN_BLOCK_THREADS
#define ARN_BLOCK_THREADS 256          // threads per workgroup of the blocked sweep (build-time knob)
#endif
struct ArnBlockCols { const double* re[ARN_P]; const double* im[ARN_P]; };
template <bool PAIR> struct ArnBlockShape {
  static constexpr int W = PAIR ? 2 : 1;
  static constexpr int NG = W * ARN_P;                         // dot products of a block with w
  static constexpr int NGRAM = W * ARN_P * (ARN_P - 1) / 2;    // lower triangle of the block's Gram matrix
  static constexpr int NV = NG + NGRAM + 1;                    // + one sum of squares
};

template <bool PAIR>
__global__ void __launch_bounds__(ARN_BLOCK_THREADS)
arnoldi_block_kernel(int64_t n, const double* __restrict__ tin, int nb_in, ArnBlockCols cur, int nb_next, ArnBlockCols nxt,
                     int want_ss, double* __restrict__ wre, double* __restrict__ wim, double* __restrict__ coef_out,
                     double* __restrict__ partials, unsigned* counters, double* __restrict__ tout, double* __restrict__ ss_out) {
  using Sh = ArnBlockShape<PAIR>;
  constexpr int W = Sh::W, P = ARN_P;
  __shared__ double lds[4];
  // coefficients of the block being applied, from the sums the previous launch left (identical in every thread)
  double hr[P], hi[P];
#pragma unroll
  for (int k = 0; k < P; ++k) { hr[k] = 0.0; hi[k] = 0.0; }
  if (nb_in > 0) {
    int gi = Sh::NG;                                           // Gram entries follow the dots: (k, l) for k = 1.., l < k
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_in) {
        double cr = tin[W * k], ci = PAIR ? tin[W * k + 1] : 0.0;
#pragma unroll
        for (int l = 0; l < k; ++l) {
          const double gr = tin[gi + W * l], gim = PAIR ? tin[gi + W * l + 1] : 0.0;
          cr -= hr[l] * gr - hi[l] * gim;                      // h_l * G_kl
          if (PAIR) ci -= hr[l] * gim + hi[l] * gr;
        }
        hr[k] = cr; hi[k] = ci;
      }
      gi += W * k;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (int k = 0; k < nb_in; ++k) { coef_out[W * k] = hr[k]; if (PAIR) coef_out[W * k + 1] = hi[k]; }
  }
  double acc[Sh::NV];
#pragma unroll
  for (int v = 0; v < Sh::NV; ++v) acc[v] = 0.0;
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 cp[P], cq[P], np_[P], nq[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_in) { cp[k] = arn_ld(cur.re[k], i, true); if (PAIR) cq[k] = arn_ld(cur.im[k], i, true); }
      if (k < nb_next) { np_[k] = arn_ld(nxt.re[k], i, false); if (PAIR) nq[k] = arn_ld(nxt.im[k], i, false); }
    }
    double2 x = arn_ld(wre, i, false), y = PAIR ? arn_ld(wim, i, false) : make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_in) {
        if (PAIR) {
          x.x -= hr[k] * cp[k].x - hi[k] * cq[k].x; y.x -= hr[k] * cq[k].x + hi[k] * cp[k].x;      // w -= h * v
          x.y -= hr[k] * cp[k].y - hi[k] * cq[k].y; y.y -= hr[k] * cq[k].y + hi[k] * cp[k].y;
        } else {
          x.x = fma(-hr[k], cp[k].x, x.x); x.y = fma(-hr[k], cp[k].y, x.y);
        }
      }
    }
    if (nb_in > 0) {
      reinterpret_cast<double2*>(wre)[i] = x;
      if (PAIR) reinterpret_cast<double2*>(wim)[i] = y;
    }
    int gi = Sh::NG;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_next) {
        // conj(v_k) . w
        acc[W * k] = fma(np_[k].x, x.x, acc[W * k]); acc[W * k] = fma(np_[k].y, x.y, acc[W * k]);
        if (PAIR) {
          acc[W * k] = fma(nq[k].x, y.x, acc[W * k]); acc[W * k] = fma(nq[k].y, y.y, acc[W * k]);
          acc[W * k + 1] = fma(np_[k].x, y.x, acc[W * k + 1]); acc[W * k + 1] = fma(-nq[k].x, x.x, acc[W * k + 1]);
          acc[W * k + 1] = fma(np_[k].y, y.y, acc[W * k + 1]); acc[W * k + 1] = fma(-nq[k].y, x.y, acc[W * k + 1]);
        }
#pragma unroll
        for (int l = 0; l < k; ++l) {                          // conj(v_k) . v_l
          double& gr = acc[gi + W * l];
          gr = fma(np_[k].x, np_[l].x, gr); gr = fma(np_[k].y, np_[l].y, gr);
          if (PAIR) {
            gr = fma(nq[k].x, nq[l].x, gr); gr = fma(nq[k].y, nq[l].y, gr);
            double& gm = acc[gi + W * l + 1];
            gm = fma(np_[k].x, nq[l].x, gm); gm = fma(-nq[k].x, np_[l].x, gm);
            gm = fma(np_[k].y, nq[l].y, gm); gm = fma(-nq[k].y, np_[l].y, gm);
          }
        }
      }
      gi += W * k;
    }
    if (want_ss) {
      double& ss = acc[Sh::NV - 1];
      ss = fma(x.x, x.x, ss); ss = fma(x.y, x.y, ss);
      if (PAIR) { ss = fma(y.x, y.x, ss); ss = fma(y.y, y.y, ss); }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {         // odd length: the last element (loops unrolled: acc stays in registers)
    const int64_t i = n - 1;
    double x = wre[i], y = PAIR ? wim[i] : 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_in) {
        const double p = cur.re[k][i], q = PAIR ? cur.im[k][i] : 0.0;
        const double nx = x - (hr[k] * p - hi[k] * q);
        y = y - (hr[k] * q + hi[k] * p);
        x = nx;
      }
    }
    if (nb_in > 0) { wre[i] = x; if (PAIR) wim[i] = y; }
    double pn[P], qn[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { pn[k] = k < nb_next ? nxt.re[k][i] : 0.0; qn[k] = (PAIR && k < nb_next) ? nxt.im[k][i] : 0.0; }
    int gi = Sh::NG;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k < nb_next) {
        acc[W * k] = fma(pn[k], x, acc[W * k]);
        if (PAIR) { acc[W * k] = fma(qn[k], y, acc[W * k]); acc[W * k + 1] = fma(pn[k], y, acc[W * k + 1]); acc[W * k + 1] = fma(-qn[k], x, acc[W * k + 1]); }
#pragma unroll
        for (int l = 0; l < k; ++l) {
          acc[gi + W * l] = fma(pn[k], pn[l], acc[gi + W * l]);
          if (PAIR) {
            acc[gi + W * l] = fma(qn[k], qn[l], acc[gi + W * l]);
            acc[gi + W * l + 1] = fma(pn[k], qn[l], acc[gi + W * l + 1]); acc[gi + W * l + 1] = fma(-qn[k], pn[l], acc[gi + W * l + 1]);
          }
        }
      }
      gi += W * k;
    }
    if (want_ss) { acc[Sh::NV - 1] = fma(x, x, acc[Sh::NV - 1]); if (PAIR) acc[Sh::NV - 1] = fma(y, y, acc[Sh::NV - 1]); }
  }
  // all NV sums of the workgroup through ONE LDS stage (wave shuffles, one barrier, thread v adds the four wave sums)
  const int G = gridDim.x;
  __shared__ double red[Sh::NV][ARN_BLOCK_THREADS / 64];
  {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < Sh::NV; ++v) {
      const double r = wave_reduce_sum(acc[v]);
      if (lane == 0) red[v][wid] = r;
    }
    __syncthreads();
    if ((int)threadIdx.x < Sh::NV) {
      double t = red[threadIdx.x][0];
      for (int w = 1; w < ARN_BLOCK_THREADS / 64; ++w) t += red[threadIdx.x][w];
      store_partial(partials + (size_t)threadIdx.x * G + blockIdx.x, t);
    }
  }
  if (last_block_ticket(counters, (unsigned)G, blockIdx.x)) {
    // the NV totals side by side: 8 lanes per value (8 x 21 = 168 <= 256 threads), lane k adds the partials k, k + 8, ...
    // in ascending order and the 8 lane sums are folded in a fixed tree
    const int v = threadIdx.x >> 3, k = threadIdx.x & 7;
    double a = 0.0;
    if (v < Sh::NV)
      for (int i = k; i < G; i += 8) a += __hip_atomic_load(partials + (size_t)v * G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 1, 64);
    if (v < Sh::NV && k == 0) {
      tout[v] = a;
      if (v == Sh::NV - 1 && ss_out) *ss_out = a;
    }
    release_ticket_counter(counters);
  }
}

// Workspace of one blocked Arnoldi step: a stream and what the step keeps per stream (so that the steps of different
// right-hand sides can run on different streams at once).
struct ArnSpace {
  hipStream_t stream;
  double* partials;          // Sh::NV areas of <= 8192 doubles
  double* tot;               // two total records of 32 doubles, used alternately
  unsigned* cnt;             // ticket counters (zero between kernels)
};

static ArnSpace arnoldi_main_space(hipeig_ctx* c) {
  return ArnSpace{c->stream, c->d_partials, c->d_scalars + 3200, c->d_counters + 3 * HIPEIG_TICKET_WORDS};
}

#define ARN_SIDE_PARTIALS (32 * 8192)      // doubles per side stream: >= Sh::NV (21) areas of 8192
#define ARN_SIDE_DOUBLES (ARN_SIDE_PARTIALS + 64 + 128)      // + two total records + the step's result record

// The side streams are created on first use: HIPEIG_ARNOLDI_STREAMS of them (1..16; 1 = none, everything on the compute stream).
// Measured (tools/experiments/gcrot_block_solve.py, the 16 solves of one contour point, 4-column sweeps): 1 / 2 / 4 / 8 / 16
// streams at N = 1e6 8.78 / 8.70 / 7.13 / 6.70 / 6.64 s (6: 7.26; with GPU_MAX_HW_QUEUES=8 instead of the runtime's 4: 7.1-7.3),
// at N = 4e6 22.8 / - / 21.9 s, at N = 1e7 37.9 / - / 36.3 / 36.2 s - a sweep of 160 MB does not fill the chip at N = 1e6 (4.3 TB/s:
// launch ramp, the ticket tail), several of them from different right-hand sides do; identical coefficients and iteration counts.
static int arnoldi_side_streams(hipeig_ctx* c) {
  if (c->arn_nstreams) return 0;
  int ns = 8;
  if (const char* e = getenv("HIPEIG_ARNOLDI_STREAMS")) ns = atoi(e);
  if (ns < 1) ns = 1;
  if (ns > 16) ns = 16;
  if (ns > 1) {
    HIPEIG_CHECK(hipMalloc((void**)&c->d_arn_ws, (size_t)ns * ARN_SIDE_DOUBLES * sizeof(double)));
    HIPEIG_CHECK(hipMalloc((void**)&c->d_arn_cnt, (size_t)ns * HIPEIG_TICKET_WORDS * sizeof(unsigned)));
    HIPEIG_CHECK(hipMemset(c->d_arn_cnt, 0, (size_t)ns * HIPEIG_TICKET_WORDS * sizeof(unsigned)));
    for (int k = 0; k < ns; ++k) HIPEIG_CHECK(hipStreamCreateWithFlags(&c->arn_stream[k], hipStreamNonBlocking));
    for (int k = 0; k < 16; ++k) HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_arn_in[k], hipEventDisableTiming));
  }
  c->arn_nstreams = ns;
  return 0;
}

template <bool PAIR>
static int arnoldi_blocked(hipeig_ctx* c, const ArnSpace& sp, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                           double* wre, double* wim, double* dres) {
  using Sh = ArnBlockShape<PAIR>;
  constexpr int W = Sh::W;
  // One workgroup per CU, whatever the length (measured, tools/experiments/arnoldi_bench.py, 28 complex columns: N = 1e7
  // 145 / 113 / 98 / 89 / 82 us per column at 4883 / 2441 / 1220 / 610 / 244 workgroups, N = 1e6 15.6 / 12.8 at 488 / 244; the
  // sequential sweep: 116 / 13.9).  A pass reads 18 streams at once; with one workgroup per CU the whole chip walks through
  // each of them as one narrow front.
  int g = c->num_cu;
  if (const char* e = getenv("HIPEIG_ARNOLDI_PER_THREAD")) g = grid_wide(n, atoi(e) > 0 ? atoi(e) : 16);      // tuning knob (elements per thread)
  if (const char* e = getenv("HIPEIG_ARNOLDI_WGS")) g = atoi(e) > 0 ? atoi(e) : g;                              // tuning knob (workgroups)
  if ((int64_t)g * ARN_BLOCK_THREADS * 2 > n) g = (int)((n / 2 + ARN_BLOCK_THREADS - 1) / ARN_BLOCK_THREADS);
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;                                      // Sh::NV partial areas of g doubles each
  double* P0 = sp.partials;
  double* tot = sp.tot;                                        // two total records of <= 32 doubles, used alternately
  unsigned* cnt = sp.cnt;
  auto cols = [&](int b0, ArnBlockCols* out) -> int {
    int nb = m - b0;
    if (nb > ARN_P) nb = ARN_P;
    if (nb < 0) nb = 0;
    for (int k = 0; k < ARN_P; ++k) {
      out->re[k] = k < nb ? Vre[b0 + k] : nullptr;
      out->im[k] = (PAIR && k < nb) ? Vim[b0 + k] : nullptr;
    }
    return nb;
  };
  ArnBlockCols none, cur, nxt;
  cols(m, &none);
  int nb_next = cols(0, &nxt);
  // first pass: ||w||^2 before and everything block 0 needs (no update); with m == 0 it is also ||w||^2 after
  hipLaunchKernelGGL((arnoldi_block_kernel<PAIR>), dim3(g), dim3(ARN_BLOCK_THREADS), 0, sp.stream, n, (const double*)nullptr, 0, none,
                     nb_next, nxt, 1, wre, wim, (double*)nullptr, P0, cnt, tot, dres);
  int flip = 0;
  for (int b0 = 0; b0 < m; b0 += ARN_P) {
    cur = nxt;
    const int nb_in = nb_next;
    nb_next = cols(b0 + ARN_P, &nxt);
    const int last = (nb_next == 0);
    hipLaunchKernelGGL((arnoldi_block_kernel<PAIR>), dim3(g), dim3(ARN_BLOCK_THREADS), 0, sp.stream, n, tot + 32 * flip, nb_in, cur,
                       nb_next, nxt, last, wre, wim, dres + 1 + W * b0, P0, cnt, tot + 32 * (flip ^ 1),
                       last ? dres + 1 + W * m : (double*)nullptr);
    flip ^= 1;
  }
  if (m == 0)
    HIPEIG_CHECK(hipMemcpyAsync(dres + 1, dres, sizeof(double), hipMemcpyDeviceToDevice, sp.stream));
  hipLaunchKernelGGL(scale_by_inv_norm_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, sp.stream, n, dres + 1 + W * m, wre, wim);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

static int arnoldi_sumsq(hipeig_ctx* c, int64_t n, const double* a, const double* b, int g, double* dst) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, a, b, c->d_partials);
  hipLaunchKernelGGL(mgsp_reduce_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, c->d_partials, g, 1, g, dst);
  return hipeig_allreduce_sum(c, dst, 1);
}

extern "C" int hipeig_arnoldi_step(hipeig_ctx* c, int64_t n, int m, const double* const* V, double* w, double* out) {
  HIPEIG_REQUIRE(m >= 0 && m <= 1000 && out, "bad arguments");
  const int g = grid_for(n, 4);
  double* dres = c->d_scalars + 2560;                  // m + 2 doubles
  double* red = c->d_scalars + 3600;
  if (!c->collectives) {
    const bool direct = c->h_scalars_dev && arnoldi_is_small(n, m);     // no copy to wait for: 28 -> 17 us per step
    if (arnoldi_fused<false>(c, n, m, V, nullptr, w, nullptr, direct ? c->h_scalars_dev : dres)) return 4;
    if (!direct) HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (m + 2), hipMemcpyDeviceToHost, c->stream));
    if (hipeig_sync_checked(c)) return 4;
    memcpy(out, c->h_scalars, sizeof(double) * (m + 2));
    return 0;
  }
  if (arnoldi_sumsq(c, n, w, nullptr, g, dres)) return 4;
  for (int j = 0; j < m; ++j) {
    hipLaunchKernelGGL(mgsp_dot_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, V[j], w, c->d_partials);
    const double* p = c->d_partials;
    int npart = g;
    if (c->collectives) {
      hipLaunchKernelGGL(mgsp_reduce_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, c->d_partials, g, 1, g, red);
      if (hipeig_allreduce_sum(c, red, 1)) return 4;
      p = red; npart = 1;
    }
    hipLaunchKernelGGL(mgsp_update_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, p, npart, V[j], w, dres + 1 + j);
  }
  if (arnoldi_sumsq(c, n, w, nullptr, g, dres + 1 + m)) return 4;
  hipLaunchKernelGGL(scale_by_inv_norm_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, dres + 1 + m, w, (double*)nullptr);
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (m + 2), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(out, c->h_scalars, sizeof(double) * (m + 2));
  return 0;
}

extern "C" int hipeig_pair_arnoldi_step(hipeig_ctx* c, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                                        double* wre, double* wim, double* out) {
  HIPEIG_REQUIRE(m >= 0 && m <= 500 && out, "bad arguments");
  const int g = grid_for(n, 4);
  double* dres = c->d_scalars + 2560;                  // 2m + 2 doubles
  double* red = c->d_scalars + 3600;
  if (!c->collectives) {
    const bool direct = c->h_scalars_dev && arnoldi_is_small(n, m);
    if (arnoldi_fused<true>(c, n, m, Vre, Vim, wre, wim, direct ? c->h_scalars_dev : dres)) return 4;
    if (!direct) HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (2 * m + 2), hipMemcpyDeviceToHost, c->stream));
    if (hipeig_sync_checked(c)) return 4;
    memcpy(out, c->h_scalars, sizeof(double) * (2 * m + 2));
    return 0;
  }
  if (arnoldi_sumsq(c, n, wre, wim, g, dres)) return 4;
  for (int j = 0; j < m; ++j) {
    hipLaunchKernelGGL(mgsp_pair_dot_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, Vre[j], Vim[j], wre, wim, c->d_partials);
    const double* p = c->d_partials;
    int npart = g;
    if (c->collectives) {
      hipLaunchKernelGGL(mgsp_reduce_kernel, dim3(1), dim3(HIPEIG_BLOCK), 0, c->stream, c->d_partials, g, 2, g, red);
      if (hipeig_allreduce_sum(c, red, 2)) return 4;
      p = red; npart = 1;
    }
    hipLaunchKernelGGL(mgsp_pair_update_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, p, npart, g,
                       Vre[j], Vim[j], wre, wim, dres + 1 + 2 * j);
  }
  if (arnoldi_sumsq(c, n, wre, wim, g, dres + 1 + 2 * m)) return 4;
  hipLaunchKernelGGL(scale_by_inv_norm_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, dres + 1 + 2 * m, wre, wim);
  HIPEIG_CHECK(hipGetLastError());
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (2 * m + 2), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(out, c->h_scalars, sizeof(double) * (2 * m + 2));
  return 0;
}

// The same Arnoldi step with `cols_per_pass` columns per pass over w: 1 = the sequential sweep above (scipy's order of
// rounding, the default everywhere), 4 = the blocked form (same algebra, 2.5 instead of 4 vector streams per column and
// a launch per four columns; coefficients agree with the sequential sweep to rounding).  One GPU, vectors beyond the
// one-workgroup size; anything else takes the sequential sweep.
extern "C" int hipeig_arnoldi_step_p(hipeig_ctx* c, int64_t n, int m, const double* const* V, double* w, double* out, int cols_per_pass) {
  HIPEIG_REQUIRE(cols_per_pass == 1 || cols_per_pass == 4, "cols_per_pass must be 1 or 4");
  if (cols_per_pass == 1 || c->collectives || n <= (int64_t)ARN_SMALL_THREADS * ARN_SMALL_E) return hipeig_arnoldi_step(c, n, m, V, w, out);
  HIPEIG_REQUIRE(m >= 0 && m <= 600 && out, "bad arguments");
  double* dres = c->d_scalars + 2560;                  // m + 2 doubles (< 640: the total records sit at 3200)
  if (arnoldi_blocked<false>(c, arnoldi_main_space(c), n, m, V, nullptr, w, nullptr, dres)) return 4;
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (m + 2), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(out, c->h_scalars, sizeof(double) * (m + 2));
  return 0;
}

extern "C" int hipeig_pair_arnoldi_step_p(hipeig_ctx* c, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                                          double* wre, double* wim, double* out, int cols_per_pass) {
  HIPEIG_REQUIRE(cols_per_pass == 1 || cols_per_pass == 4, "cols_per_pass must be 1 or 4");
  if (cols_per_pass == 1 || c->collectives || n <= (int64_t)ARN_SMALL_THREADS * ARN_SMALL_E)
    return hipeig_pair_arnoldi_step(c, n, m, Vre, Vim, wre, wim, out);
  HIPEIG_REQUIRE(m >= 0 && m <= 250 && out, "bad arguments");
  double* dres = c->d_scalars + 2560;                  // 2m + 2 doubles (< 640: the total records sit at 3200)
  if (arnoldi_blocked<true>(c, arnoldi_main_space(c), n, m, Vre, Vim, wre, wim, dres)) return 4;
  HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars, dres, sizeof(double) * (2 * m + 2), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  memcpy(out, c->h_scalars, sizeof(double) * (2 * m + 2));
  return 0;
}

// Split form for several independent steps in a row (the right-hand sides of a lock-step block solve each orthogonalise
// against their OWN basis): `begin` enqueues the step and an asynchronous copy of its scalars into pinned slot `slot`
// (0..15, up to 126 doubles each), `end` waits for the stream and hands them over - the host work of one right-hand side
// then overlaps the kernels of the next instead of the GPU idling at every step's round trip.
extern "C" int hipeig_pair_arnoldi_step_begin(hipeig_ctx* c, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                                              double* wre, double* wim, int cols_per_pass, int slot) {
  HIPEIG_REQUIRE(slot >= 0 && slot < 16 && m >= 0 && 2 * m + 2 <= ARN_SLOT_DOUBLES - 2, "bad slot / too many columns for the split form");
  HIPEIG_REQUIRE(!c->collectives, "the split form is for one GPU");
  HIPEIG_REQUIRE(cols_per_pass == 1 || cols_per_pass == 4, "cols_per_pass must be 1 or 4");
  double* dres = c->d_scalars + 2560;
  const bool blocked = cols_per_pass != 1 && n > (int64_t)ARN_SMALL_THREADS * ARN_SMALL_E;
  const bool direct = !blocked && c->h_scalars_dev && arnoldi_is_small(n, m);
  if (direct) dres = c->h_scalars_dev + 2048 + (size_t)slot * ARN_SLOT_DOUBLES;
  if (blocked) {
    if (arnoldi_side_streams(c)) return 4;
    if (c->arn_nstreams > 1) {
      // The steps of the right-hand sides are independent: slot s runs on side stream s % ns with that stream's own partial
      // areas, total records, ticket counters and result record, behind an event that says "the compute stream has produced
      // this slot's operands".  The caller collects the slot (hipeig_arnoldi_step_end waits for the slot's event) before it
      // enqueues anything that reads what the step wrote, so nothing on the compute stream has to wait for the side stream.
      const int k = slot % c->arn_nstreams;
      double* base = c->d_arn_ws + (size_t)k * ARN_SIDE_DOUBLES;
      const ArnSpace sp{c->arn_stream[k], base, base + ARN_SIDE_PARTIALS, c->d_arn_cnt + (size_t)k * HIPEIG_TICKET_WORDS};
      double* res = base + ARN_SIDE_PARTIALS + 64;
      HIPEIG_CHECK(hipEventRecord(c->ev_arn_in[slot], c->stream));
      HIPEIG_CHECK(hipStreamWaitEvent(sp.stream, c->ev_arn_in[slot], 0));
      if (arnoldi_blocked<true>(c, sp, n, m, Vre, Vim, wre, wim, res)) return 4;
      HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars + 2048 + (size_t)slot * ARN_SLOT_DOUBLES, res, sizeof(double) * (2 * m + 2),
                                  hipMemcpyDeviceToHost, sp.stream));
      HIPEIG_CHECK(hipEventRecord(c->ev_slot[slot], sp.stream));
      return 0;
    }
  }
  if (blocked ? arnoldi_blocked<true>(c, arnoldi_main_space(c), n, m, Vre, Vim, wre, wim, dres) : arnoldi_fused<true>(c, n, m, Vre, Vim, wre, wim, dres)) return 4;
  if (!direct)
    HIPEIG_CHECK(hipMemcpyAsync(c->h_scalars + 2048 + (size_t)slot * ARN_SLOT_DOUBLES, dres, sizeof(double) * (2 * m + 2),
                                hipMemcpyDeviceToHost, c->stream));
  HIPEIG_CHECK(hipEventRecord(c->ev_slot[slot], c->stream));
  return 0;
}

// `count` (<= 16) such steps of length n <= 8192 in one launch (arnoldi_small_batch_kernel): step i has m[i] columns
// Vre[i * 64 + j], Vim[i * 64 + j] (tables of 64 entries per step), works on (wre[i], wim[i]) and reports into pinned
// slot i; collect with hipeig_arnoldi_step_end(slot i).  Needs the mapped scalar area (returns 5 without it or for longer
// vectors: the caller then takes the step-by-step form).
extern "C" int hipeig_pair_arnoldi_step_batch_begin(hipeig_ctx* c, int64_t n, int count, const int* m, const double* const* Vre,
                                                    const double* const* Vim, double* const* wre, double* const* wim) {
  HIPEIG_REQUIRE(count >= 1 && count <= 16 && m && Vre && Vim && wre && wim, "bad arguments");
  HIPEIG_REQUIRE(!c->collectives, "the split form is for one GPU");
  if (n > (int64_t)ARN_SMALL_THREADS * ARN_SMALL_E || !c->h_scalars_dev) return 5;
  if (!c->h_arn_items) {
    HIPEIG_CHECK(hipHostMalloc(&c->h_arn_items, 16 * sizeof(ArnBatchItem), hipHostMallocMapped));
    if (hipHostGetDevicePointer(&c->d_arn_items, c->h_arn_items, 0) != hipSuccess) {
      (void)hipGetLastError();
      hipHostFree(c->h_arn_items);
      c->h_arn_items = nullptr;
      return 5;
    }
  }
  // the previous batch has been collected (every `end` waits for the batch's event) before its items are overwritten
  HIPEIG_CHECK(hipEventSynchronize(c->ev_slot[0]));
  ArnBatchItem* items = (ArnBatchItem*)c->h_arn_items;
  for (int i = 0; i < count; ++i) {
    HIPEIG_REQUIRE(m[i] >= 0 && m[i] <= ARN_SMALL_MAXCOLS && 2 * m[i] + 2 <= ARN_SLOT_DOUBLES - 2, "too many columns for the split form");
    items[i].m = m[i]; items[i].pad = 0;
    items[i].wre = wre[i]; items[i].wim = wim[i];
    for (int j = 0; j < m[i]; ++j) {
      items[i].re[j] = Vre[(size_t)i * ARN_SMALL_MAXCOLS + j];
      items[i].im[j] = Vim[(size_t)i * ARN_SMALL_MAXCOLS + j];
    }
  }
  hipLaunchKernelGGL(arnoldi_small_batch_kernel, dim3(count), dim3(ARN_SMALL_THREADS), 0, c->stream, (int)n,
                     (const ArnBatchItem*)c->d_arn_items, c->h_scalars_dev + 2048);
  HIPEIG_CHECK(hipGetLastError());
  for (int i = 0; i < count; ++i) HIPEIG_CHECK(hipEventRecord(c->ev_slot[i], c->stream));
  return 0;
}

extern "C" int hipeig_arnoldi_step_end(hipeig_ctx* c, int slot, int count, double* out) {
  HIPEIG_REQUIRE(slot >= 0 && slot < 16 && count >= 0 && count <= ARN_SLOT_DOUBLES && out, "bad arguments");
  HIPEIG_CHECK(hipEventSynchronize(c->ev_slot[slot]));        // this step only: the steps enqueued behind it keep running
  memcpy(out, c->h_scalars + 2048 + (size_t)slot * ARN_SLOT_DOUBLES, sizeof(double) * count);
  return 0;
}

extern "C" int hipeig_orthonormalize(hipeig_ctx* c, int64_t n, int m, const double* const* Y,
                                     double* x, double lindep, int method, double* innerprod,
                                     int* is_lindep) {
  HIPEIG_REQUIRE(innerprod && is_lindep, "null outputs");
  HIPEIG_REQUIRE(method == 0 || method == 1, "method must be 0 (MGS) or 1 (CGS2)");
  double ip = 0.0;
  if (method == 0) {
    // kernel j: update with q_{j-1}, dots with q_j; totals alternate between two slots of the scalar area (kernel j reads
    // slot j-1 while its last workgroup writes slot j); the last kernel's x.x goes to the host
    const int g = grid_records(c, n, "HIPEIG_MGS_PER_THREAD");
    unsigned* cnt = c->d_counters + 3 * HIPEIG_TICKET_WORDS;
    for (int j = 0; j <= m; ++j) {
      double* slot = (j == m) ? record_target(c, true) : c->d_scalars + 8 + 2 * (j & 1);
      const double* prev = c->d_scalars + 8 + 2 * ((j + 1) & 1);
      hipLaunchKernelGGL(mgs_sweep_kernel, dim3(g), dim3(HIPEIG_BLOCK), 0, c->stream, n, x, j > 0 ? Y[j - 1] : nullptr, prev,
                         j < m ? Y[j] : nullptr, c->d_partials, c->d_group_partials, cnt, slot);
      HIPEIG_CHECK(hipGetLastError());
      if (j < m && c->collectives && hipeig_allreduce_sum(c, slot, 2)) return 4;
    }
    if (records_to_host(c, 1, &ip)) return 4;
  } else {
    const int cap = HIPEIG_MAX_COLS * HIPEIG_MAX_COLS;
    for (int pass = 0; pass < 2; ++pass) {
      for (int j0 = 0; j0 < m; j0 += cap) {
        const int mm = (m - j0 < cap) ? (m - j0) : cap;
        if (multi_dot_impl(c, n, mm, Y + j0, x, nullptr)) return 4;       // coefficients stay on the device
        if (lincomb_impl(c, n, mm, nullptr, c->d_scalars, -1.0, Y + j0, x, 1)) return 4;
      }
    }
  }
  if (method != 0) {
    int rc = hipeig_dot(c, n, x, x, &ip);
    if (rc) return rc;
  }
  *innerprod = ip;
  if (ip > lindep) {
    *is_lindep = 0;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, sqrt(ip), 1, x, x);
    HIPEIG_CHECK(hipGetLastError());
  } else {
    *is_lindep = 1;
  }
  return 0;
}
