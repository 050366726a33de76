// Seeded synthetic "gapped random-sparse Hermitian" operator, generated on the device.
//
// Specification (mirrored bit for bit by eigensolvers_amd/generators.py):
//   * K pseudo-random permutations pi_k of [0,N): 4-round Feistel networks on 2h bits
//     (2^(2h) >= N) with cycle walking, round function mix64(half ^ key[k][round]) & mask.
//   * row i holds, for every k, the forward edge (i, pi_k(i)) if keep(k,i) and the inverse
//     edge (i, pi_k^-1(i)) if keep(k, pi_k^-1(i)); keep(k,s) = (mix64(s ^ keepkey[k]) >> 40)
//     < thresh24.  Entry value of edge s -> pi_k(s): (sum of the four 16-bit fields of
//     mix64(s ^ valkey[k]) - 131070) * vscale, identical seen from both ends => symmetric.
//   * diagonal: target rows get targets[t]; the others +-(1 + 9u), u in [0,1).
//   * each row is sorted by (column, slot); duplicate columns are kept as separate entries.
// The keys are derived from the seed on the host (splitmix64) and passed in.
#include <vector>
#include "common.h"

int hipeig_csr_finalize(hipeig_ctx* c, hipeig_csr* A, const int32_t* rowptr32);

#define GEN_MAX_K 64

struct GenParams {
  int64_t N;
  int K;
  int half_bits;
  uint32_t thresh24;
  double vscale;
  uint64_t diagkey, signkey;
  int ntargets;
  int64_t target_stride, target_first;
};

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t feistel_fwd(uint64_t v, const uint64_t* key4, int hb, uint64_t mask, int64_t N) {
  do {
    uint64_t L = v >> hb, R = v & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint64_t t = L ^ (mix64(R ^ key4[r]) & mask);
      L = R; R = t;
    }
    v = (L << hb) | R;
  } while ((int64_t)v >= N);
  return v;
}

__device__ __forceinline__ uint64_t feistel_inv(uint64_t v, const uint64_t* key4, int hb, uint64_t mask, int64_t N) {
  do {
    uint64_t L = v >> hb, R = v & mask;
#pragma unroll
    for (int r = 3; r >= 0; --r) {
      const uint64_t t = R ^ (mix64(L ^ key4[r]) & mask);
      R = L; L = t;
    }
    v = (L << hb) | R;
  } while ((int64_t)v >= N);
  return v;
}

__device__ __forceinline__ bool gen_keep(uint64_t s, uint64_t keepkey, uint32_t thresh24) {
  return (uint32_t)(mix64(s ^ keepkey) >> 40) < thresh24;
}

__device__ __forceinline__ double gen_value(uint64_t s, uint64_t valkey, double vscale) {
  const uint64_t h = mix64(s ^ valkey);
  const int64_t sum = (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) +
                      (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
  return mul_rn((double)(sum - 131070), vscale);
}

__device__ __forceinline__ double gen_diag(int64_t i, const GenParams& P, const double* targets) {
  if (i >= P.target_first && (i - P.target_first) % P.target_stride == 0) {
    const int64_t t = (i - P.target_first) / P.target_stride;
    if (t < P.ntargets) return targets[t];
  }
  const double u = mul_rn((double)(mix64((uint64_t)i ^ P.diagkey) >> 11), 1.1102230246251565e-16);  // 2^-53
  const double mag = add_rn(1.0, mul_rn(9.0, u));
  return (mix64((uint64_t)i ^ P.signkey) & 1ULL) ? -mag : mag;
}

// keys layout: [k*4 + r] round keys, then [4K + k] keep keys, then [5K + k] value keys
__global__ void gen_count_kernel(GenParams P, const uint64_t* __restrict__ keys, int64_t row_begin,
                                 int64_t nrows, int32_t* __restrict__ counts) {
  const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= nrows) return;
  const uint64_t i = (uint64_t)(row_begin + li);
  const uint64_t mask = (1ULL << P.half_bits) - 1;
  int cnt = 1;
  for (int k = 0; k < P.K; ++k) {
    if (gen_keep(i, keys[4 * P.K + k], P.thresh24)) ++cnt;
    const uint64_t s = feistel_inv(i, keys + 4 * k, P.half_bits, mask, P.N);
    if (gen_keep(s, keys[4 * P.K + k], P.thresh24)) ++cnt;
  }
  counts[li] = cnt;
}

__global__ void gen_fill_kernel(GenParams P, const uint64_t* __restrict__ keys, const double* __restrict__ targets,
                                int64_t row_begin, int64_t nrows, const int32_t* __restrict__ rowptr,
                                int32_t* __restrict__ col, double* __restrict__ val) {
  const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= nrows) return;
  const uint64_t i = (uint64_t)(row_begin + li);
  const uint64_t mask = (1ULL << P.half_bits) - 1;
  uint64_t key[2 * GEN_MAX_K + 1];   // (col << 8) | slot
  double v[2 * GEN_MAX_K + 1];
  int n = 0;
  for (int k = 0; k < P.K; ++k) {
    if (gen_keep(i, keys[4 * P.K + k], P.thresh24)) {
      const uint64_t j = feistel_fwd(i, keys + 4 * k, P.half_bits, mask, P.N);
      key[n] = (j << 8) | (uint64_t)(2 * k);
      v[n++] = gen_value(i, keys[5 * P.K + k], P.vscale);
    }
    const uint64_t s = feistel_inv(i, keys + 4 * k, P.half_bits, mask, P.N);
    if (gen_keep(s, keys[4 * P.K + k], P.thresh24)) {
      key[n] = (s << 8) | (uint64_t)(2 * k + 1);
      v[n++] = gen_value(s, keys[5 * P.K + k], P.vscale);
    }
  }
  key[n] = (i << 8) | (uint64_t)(2 * P.K);
  v[n++] = gen_diag((int64_t)i, P, targets);
  // insertion sort by (column, slot)
  for (int a = 1; a < n; ++a) {
    const uint64_t ka = key[a];
    const double va = v[a];
    int b = a - 1;
    while (b >= 0 && key[b] > ka) { key[b + 1] = key[b]; v[b + 1] = v[b]; --b; }
    key[b + 1] = ka; v[b + 1] = va;
  }
  const int32_t p0 = rowptr[li];
  for (int a = 0; a < n; ++a) {
    col[p0 + a] = (int32_t)(key[a] >> 8);
    val[p0 + a] = v[a];
  }
}

extern "C" int hipeig_csr_generate(hipeig_ctx* c, int64_t N, int64_t row_begin, int64_t row_end,
                                   int K, uint64_t seed, double eps, uint32_t keep_thresh24,
                                   const double* targets, int ntargets, hipeig_csr** out) {
  HIPEIG_REQUIRE(out != nullptr, "null output");
  HIPEIG_REQUIRE(N >= 2 && N < ((int64_t)1 << 31), "N out of range");
  HIPEIG_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= N, "bad row range");
  HIPEIG_REQUIRE(K >= 1 && K <= GEN_MAX_K, "K out of range");
  HIPEIG_REQUIRE(ntargets >= 0 && ntargets <= 256 && (ntargets == 0 || targets), "bad targets");
  HIPEIG_REQUIRE(keep_thresh24 <= (1u << 24), "keep threshold is a 24-bit fraction");
  GenParams P;
  P.N = N; P.K = K; P.thresh24 = keep_thresh24;
  int bits = 1;
  while (((int64_t)1 << bits) < N) ++bits;
  P.half_bits = (bits + 1) / 2;
  // mean row length 2*K*q + 1 -> unit-variance values scaled to eps/sqrt(nnz_row)
  const double q = (double)keep_thresh24 / 16777216.0;
  const double nnz_row = 2.0 * K * q;
  P.vscale = eps * 1.7320508075688772 / (65535.0 * sqrt(nnz_row));
  std::vector<uint64_t> keys((size_t)6 * K);
  uint64_t st = seed;
  auto next = [&st]() { st += 0x9E3779B97F4A7C15ULL; return mix64(st); };
  for (int i = 0; i < 6 * K; ++i) keys[i] = next();
  P.diagkey = next();
  P.signkey = next();
  P.ntargets = ntargets;
  P.target_stride = ntargets ? N / ntargets : 1;
  if (P.target_stride < 1) P.target_stride = 1;
  P.target_first = P.target_stride / 2;

  const int64_t nrows = row_end - row_begin;
  hipeig_csr* A = (hipeig_csr*)calloc(1, sizeof(hipeig_csr));
  HIPEIG_REQUIRE(A != nullptr, "out of host memory");
  A->nrows = nrows; A->ncols = N; A->row_offset = row_begin;
  uint64_t* d_keys = nullptr;
  double* d_targets = nullptr;
  int32_t* d_counts = nullptr;
  HIPEIG_CHECK(hipMalloc((void**)&d_keys, keys.size() * sizeof(uint64_t)));
  HIPEIG_CHECK(hipMalloc((void**)&d_targets, sizeof(double) * (ntargets > 0 ? ntargets : 1)));
  HIPEIG_CHECK(hipMalloc((void**)&d_counts, sizeof(int32_t) * (size_t)(nrows > 0 ? nrows : 1)));
  HIPEIG_CHECK(hipMemcpyAsync(d_keys, keys.data(), keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  if (ntargets) HIPEIG_CHECK(hipMemcpyAsync(d_targets, targets, sizeof(double) * ntargets, hipMemcpyHostToDevice, c->stream));
  const int blocks = (int)((nrows + 255) / 256);
  std::vector<int32_t> rp((size_t)nrows + 1, 0);
  if (nrows > 0) {
    hipLaunchKernelGGL(gen_count_kernel, dim3(blocks), dim3(256), 0, c->stream, P, d_keys, row_begin, nrows, d_counts);
    HIPEIG_CHECK(hipGetLastError());
    HIPEIG_CHECK(hipMemcpyAsync(rp.data() + 1, d_counts, sizeof(int32_t) * (size_t)nrows, hipMemcpyDeviceToHost, c->stream));
    HIPEIG_CHECK(hipStreamSynchronize(c->stream));
    int64_t run = 0;
    for (int64_t i = 1; i <= nrows; ++i) {
      run += rp[i];
      HIPEIG_REQUIRE(run < ((int64_t)1 << 31), "local nnz must fit int32");
      rp[i] = (int32_t)run;
    }
  }
  A->nnz = rp[nrows];
  HIPEIG_CHECK(hipMalloc((void**)&A->d_rowptr, (size_t)(nrows + 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_col, (size_t)(A->nnz > 0 ? A->nnz : 1) * sizeof(int32_t)));
  HIPEIG_CHECK(hipMalloc((void**)&A->d_val, (size_t)(A->nnz > 0 ? A->nnz : 1) * sizeof(double)));
  HIPEIG_CHECK(hipMemcpyAsync(A->d_rowptr, rp.data(), (size_t)(nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (nrows > 0) {
    hipLaunchKernelGGL(gen_fill_kernel, dim3(blocks), dim3(256), 0, c->stream, P, d_keys, d_targets, row_begin, nrows,
                       A->d_rowptr, A->d_col, A->d_val);
    HIPEIG_CHECK(hipGetLastError());
  }
  HIPEIG_CHECK(hipStreamSynchronize(c->stream));
  hipFree(d_keys); hipFree(d_targets); hipFree(d_counts);
  int rc = hipeig_csr_finalize(c, A, rp.data());
  if (rc) { hipeig_csr_destroy(c, A); return rc; }
  *out = A;
  return 0;
}
