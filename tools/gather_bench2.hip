// Second round of micro-benchmarks: (a) L1-sized tables, (b) cache-policy variants of the
// gather from an 80 MB table (plain / nt / sc1 / sc0sc1), (c) LDS gather at 16 waves/CU,
// (d) a windowed gather whose indices sweep 2 MB windows in lockstep (the schedule of a
// column-window blocked SpMV).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ double ld(const double* p) {
  if (MODE == 0) return *p;
  if (MODE == 1) return __builtin_nontemporal_load(p);
  double v;
  if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ void gather8(const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ table, double* out) {
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = __builtin_nontemporal_load(idx + i);
    int32_t j1 = __builtin_nontemporal_load(idx + i + stride);
    int32_t j2 = __builtin_nontemporal_load(idx + i + 2 * stride);
    int32_t j3 = __builtin_nontemporal_load(idx + i + 3 * stride);
    a += ld<MODE>(table + j0) + ld<MODE>(table + j1) + ld<MODE>(table + j2) + ld<MODE>(table + j3);
  }
  if (a == 12345.678) out[0] = a;
}

// each block walks its own contiguous chunk of the index stream (like a row block's tile stream)
__global__ void gather_chunked(const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ table, double* out) {
  double a = 0;
  const int64_t per = n / gridDim.x;
  const int32_t* my = idx + per * blockIdx.x;
  for (int64_t i = threadIdx.x; i + 3 * blockDim.x < per; i += 4 * blockDim.x) {
    int32_t j0 = __builtin_nontemporal_load(my + i);
    int32_t j1 = __builtin_nontemporal_load(my + i + blockDim.x);
    int32_t j2 = __builtin_nontemporal_load(my + i + 2 * blockDim.x);
    int32_t j3 = __builtin_nontemporal_load(my + i + 3 * blockDim.x);
    a += table[j0] + table[j1] + table[j2] + table[j3];
  }
  if (a == 12345.678) out[0] = a;
}

template <int LDSN>
__global__ void lds_gather(const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ table, double* out) {
  __shared__ double t[LDSN];
  for (int k = threadIdx.x; k < LDSN; k += blockDim.x) t[k] = table[k];
  __syncthreads();
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = __builtin_nontemporal_load(idx + i) & (LDSN - 1);
    int32_t j1 = __builtin_nontemporal_load(idx + i + stride) & (LDSN - 1);
    int32_t j2 = __builtin_nontemporal_load(idx + i + 2 * stride) & (LDSN - 1);
    int32_t j3 = __builtin_nontemporal_load(idx + i + 3 * stride) & (LDSN - 1);
    a += t[j0] + t[j1] + t[j2] + t[j3];
  }
  if (a == 12345.678) out[0] = a;
}

template <int LDSN>
__global__ void lds_atomic(const int32_t* __restrict__ idx, int64_t n, double* out) {
  __shared__ double t[LDSN];
  for (int k = threadIdx.x; k < LDSN; k += blockDim.x) t[k] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = __builtin_nontemporal_load(idx + i) & (LDSN - 1);
    int32_t j1 = __builtin_nontemporal_load(idx + i + stride) & (LDSN - 1);
    int32_t j2 = __builtin_nontemporal_load(idx + i + 2 * stride) & (LDSN - 1);
    int32_t j3 = __builtin_nontemporal_load(idx + i + 3 * stride) & (LDSN - 1);
    atomicAdd(&t[j0], 1.0); atomicAdd(&t[j1], 1.0); atomicAdd(&t[j2], 1.0); atomicAdd(&t[j3], 1.0);
  }
  __syncthreads();
  double a = 0;
  for (int k = threadIdx.x; k < LDSN; k += blockDim.x) a += t[k];
  if (a == 12345.678) out[0] = a;
}

static uint64_t rs = 88172645463325252ULL;
static inline uint64_t xs() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  const int64_t nidx = 1LL << 27;   // 134M indices
  int32_t* d_idx; double* d_out; CK(hipMalloc(&d_idx, nidx * 4)); CK(hipMalloc(&d_out, 64));
  std::vector<int32_t> h(nidx);
  const int64_t table_n = 10000000;
  double* d_table; CK(hipMalloc(&d_table, table_n * 8));
  std::vector<double> ht(table_n); for (auto& v : ht) v = (double)(xs() % 1000) * 1e-3;
  CK(hipMemcpy(d_table, ht.data(), table_n * 8, hipMemcpyHostToDevice));
  const int grid = 2048, block = 256;
  for (int64_t tsize : {1LL << 10, 1LL << 11, 1LL << 12, 1LL << 14, 1LL << 16}) {
    for (int64_t i = 0; i < nidx; ++i) h[i] = (int32_t)(xs() % (uint64_t)tsize);
    CK(hipMemcpy(d_idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
    float ms = timeit([&] { hipLaunchKernelGGL((gather8<0>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("(a) gather 8B from %8.1f KB table: %.3f ms -> %.1f Ggather/s\n", tsize * 8 / 1e3, ms, nidx / ms / 1e6);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL((lds_gather<16384>), dim3(256), dim3(1024), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("(c) LDS gather 128 KiB table, 1024-thread WG (16 waves/CU): %.3f ms -> %.1f Ggather/s\n", ms, nidx / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((lds_gather<4096>), dim3(1024), dim3(512), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("(c) LDS gather 32 KiB table, 512-thread WG x4/CU (32 waves/CU): %.3f ms -> %.1f Ggather/s\n", ms, nidx / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((lds_atomic<16384>), dim3(256), dim3(1024), 0, 0, d_idx, nidx, d_out); }, 3);
    printf("(c) LDS atomicAdd f64 128 KiB, 1024-thread WG: %.3f ms -> %.1f Gatomic/s\n", ms, nidx / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((lds_atomic<4096>), dim3(1024), dim3(512), 0, 0, d_idx, nidx, d_out); }, 3);
    printf("(c) LDS atomicAdd f64 32 KiB, 512-thread WG x4/CU: %.3f ms -> %.1f Gatomic/s\n", ms, nidx / ms / 1e6);
  }
  for (int64_t i = 0; i < nidx; ++i) h[i] = (int32_t)(xs() % (uint64_t)table_n);
  CK(hipMemcpy(d_idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
  {
    float m0 = timeit([&] { hipLaunchKernelGGL((gather8<0>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    float m1 = timeit([&] { hipLaunchKernelGGL((gather8<1>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    float m2 = timeit([&] { hipLaunchKernelGGL((gather8<2>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    float m3 = timeit([&] { hipLaunchKernelGGL((gather8<3>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("(b) 80 MB table: plain %.1f | nt %.1f | sc1(serialised) %.1f | sc0sc1(serialised) %.1f Ggather/s\n",
           nidx / m0 / 1e6, nidx / m1 / 1e6, nidx / m2 / 1e6, nidx / m3 / 1e6);
  }
  // (d) windowed: block b's chunk of the stream is split into nwin equal phases; phase c draws from window c
  for (int64_t W : {131072LL, 262144LL, 524288LL}) {
    const int nwin = (int)((table_n + W - 1) / W);
    for (int g : {512, 1024, 2048}) {
      const int64_t per = nidx / g;
      for (int b = 0; b < g; ++b)
        for (int64_t k = 0; k < per; ++k) {
          const int c = (int)(k * nwin / per);
          int64_t lo = (int64_t)c * W, hi = lo + W; if (hi > table_n) hi = table_n;
          h[(int64_t)b * per + k] = (int32_t)(lo + xs() % (uint64_t)(hi - lo));
        }
      CK(hipMemcpy(d_idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
      float ms = timeit([&] { hipLaunchKernelGGL(gather_chunked, dim3(g), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
      printf("(d) windowed gather W=%ld cols (%.1f MB), %d windows, grid %d: %.3f ms -> %.1f Ggather/s\n", (long)W, W * 8 / 1e6, nwin, g, ms, nidx / ms / 1e6);
    }
  }
  return 0;
}
