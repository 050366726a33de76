// Micro-benchmarks that size the SpMV design on MI355X: streaming read bandwidth, random
// 8-byte gathers from tables of growing size (L2 / Infinity Cache / HBM resident), LDS
// random gather and LDS fp64 atomic-add rates.  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double dvec2 __attribute__((ext_vector_type(2)));
__global__ void stream_read(const dvec2* __restrict__ p, int64_t n2, double* out) {
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    dvec2 v = __builtin_nontemporal_load(p + i);
    a += v.x + v.y;
  }
  if (a == 12345.678) out[0] = a;
}

// idx: coalesced stream of 32-bit indices; table: doubles
template <int NT>
__global__ void gather8(const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ table, double* out) {
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = NT ? __builtin_nontemporal_load(idx + i) : idx[i];
    int32_t j1 = NT ? __builtin_nontemporal_load(idx + i + stride) : idx[i + stride];
    int32_t j2 = NT ? __builtin_nontemporal_load(idx + i + 2 * stride) : idx[i + 2 * stride];
    int32_t j3 = NT ? __builtin_nontemporal_load(idx + i + 3 * stride) : idx[i + 3 * stride];
    a += table[j0] + table[j1] + table[j2] + table[j3];
  }
  if (a == 12345.678) out[0] = a;
}

// LDS gather: table in LDS (16K doubles), indices from global stream
__global__ void lds_gather(const int32_t* __restrict__ idx, int64_t n, const double* __restrict__ table, double* out) {
  __shared__ double t[16384];
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) t[k] = table[k];
  __syncthreads();
  double a = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = __builtin_nontemporal_load(idx + i) & 16383;
    int32_t j1 = __builtin_nontemporal_load(idx + i + stride) & 16383;
    int32_t j2 = __builtin_nontemporal_load(idx + i + 2 * stride) & 16383;
    int32_t j3 = __builtin_nontemporal_load(idx + i + 3 * stride) & 16383;
    a += t[j0] + t[j1] + t[j2] + t[j3];
  }
  if (a == 12345.678) out[0] = a;
}

__global__ void lds_atomic(const int32_t* __restrict__ idx, int64_t n, double* out) {
  __shared__ double t[16384];
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) t[k] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t j0 = __builtin_nontemporal_load(idx + i) & 16383;
    int32_t j1 = __builtin_nontemporal_load(idx + i + stride) & 16383;
    int32_t j2 = __builtin_nontemporal_load(idx + i + 2 * stride) & 16383;
    int32_t j3 = __builtin_nontemporal_load(idx + i + 3 * stride) & 16383;
    atomicAdd(&t[j0], 1.0); atomicAdd(&t[j1], 1.0); atomicAdd(&t[j2], 1.0); atomicAdd(&t[j3], 1.0);
  }
  __syncthreads();
  double a = 0;
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) a += t[k];
  if (a == 12345.678) out[0] = a;
}

static uint64_t rng_state = 88172645463325252ULL;
static inline uint64_t xs() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  const int64_t nidx = 1LL << 28;   // 268M indices = 1 GiB of int32
  int32_t* d_idx; double* d_out; CK(hipMalloc(&d_idx, nidx * 4)); CK(hipMalloc(&d_out, 64));
  std::vector<int32_t> h(nidx);
  const int64_t table_max = 1LL << 27;  // 128M doubles = 1 GiB
  double* d_table; CK(hipMalloc(&d_table, table_max * 8)); CK(hipMemset(d_table, 0, table_max * 8));
  const int grid = 2048, block = 256;
  {
    float ms = timeit([&] { hipLaunchKernelGGL(stream_read, dim3(grid), dim3(block), 0, 0, (const dvec2*)d_table, table_max / 2, d_out); }, 5);
    printf("stream read 1 GiB (16 B/lane, nt): %.3f ms -> %.1f GB/s\n", ms, 1.073741824 / ms * 1e3);
  }
  for (int64_t tsize : {1LL << 16, 1LL << 18, 1LL << 19, 1LL << 20, 1LL << 21, 1LL << 23, 10000000LL, 1LL << 25, 1LL << 27}) {
    for (int64_t i = 0; i < nidx; ++i) h[i] = (int32_t)(xs() % (uint64_t)tsize);
    CK(hipMemcpy(d_idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
    float ms = timeit([&] { hipLaunchKernelGGL((gather8<1>), dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    float ms2 = timeit([&] { hipLaunchKernelGGL((gather8<1>), dim3(grid * 4), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("gather 8B from %8.2f MB table: %.3f ms -> %.1f Ggather/s (12 B/nnz equiv %.0f GB/s) | grid x4: %.1f Ggather/s\n",
           tsize * 8 / 1e6, ms, nidx / ms / 1e6, nidx * 12.0 / ms / 1e6, nidx / ms2 / 1e6);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(lds_gather, dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_table, d_out); }, 3);
    printf("LDS gather (128 KiB table/WG, 1 WG/CU): %.3f ms -> %.1f Ggather/s\n", ms, nidx / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(lds_atomic, dim3(grid), dim3(block), 0, 0, d_idx, nidx, d_out); }, 3);
    printf("LDS atomicAdd f64 (128 KiB/WG): %.3f ms -> %.1f Gatomic/s\n", ms, nidx / ms / 1e6);
  }
  return 0;
}
