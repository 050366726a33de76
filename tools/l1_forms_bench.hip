// Round-2 micro-benchmark: what the vector L1 / LDS of a CU sustain for the access FORMS the
// column-window blocked products can choose between.  One 1024-thread workgroup per CU, all
// workgroups walk the column windows of the operand in lockstep (the schedule of the TCOO-W
// sweep); indices come from a hash so that nothing but the form under test is in the loop,
// optionally next to the 12 B / element (idx, val) stream of the real kernel.
//
//   gather forms (SpMV, 8-byte x, 2^17-column windows, 17152 elements per tile):
//     1  element gathers, elements bucketed by 32-column bins (the shipped layout)
//     6  element gathers, elements fully sorted by column (same-line lanes adjacent, ascending)
//     2  line-granular: every 128-byte line of the tile's column range loaded ONCE by
//        8 lanes x 16 B (30 lines per 64 elements)
//   gather forms (SpMM, k = 8 interleaved operand, 64 B per column, 2^14-column windows):
//     4  4 lanes x 16 B per non-zero (16 non-zeros per wave instruction)
//     5  8 lanes x  8 B per non-zero ( 8 non-zeros per wave instruction)
//   LDS forms:
//     1  ds_add_f64, one random row per lane (SpMV scatter)
//     2  ds_add_f64 x 2 per lane, 4 lanes cover the 8 accumulators of one random row (SpMM)
//     3  the same rows, non-atomic: ds_read_b128 + add + ds_write_b128
// Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hsh(uint32_t a) {
  a ^= a >> 15; a *= 2246822519u; a ^= a >> 13; a *= 3266489917u; a ^= a >> 16;
  return a;
}

struct P {
  const double* x;          // operand table
  int64_t table_doubles;
  int wshift;               // log2(doubles per window)
  int nwin;
  int tile_elems;           // elements per (workgroup, window) tile
  int rows;                 // LDS accumulator rows (SpMV: doubles; SpMM: rows of 8 doubles)
  const uint32_t* sidx;     // stream (nullptr = off)
  const double* sval;
  double* out;
};

template <int G, int L>
__global__ void __launch_bounds__(1024) forms_kernel(P p) {
  extern __shared__ double yacc[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nrow_d = (G == 4 || G == 5) ? p.rows * 8 : p.rows;
  for (int k = threadIdx.x; k < nrow_d; k += 1024) yacc[k] = 0.0;
  __syncthreads();
  double sink = 0.0;
  const int64_t wdoubles = (int64_t)1 << p.wshift;
  // per wave-iteration: EPI elements
  constexpr int EPI = (G == 4) ? 64 : (G == 5) ? 32 : 256;   // G4: 4 instr x 16 nnz, G5: 4 instr x 8 nnz
  const int nb = p.tile_elems / EPI;
  for (int c = 0; c < p.nwin; ++c) {
    int64_t w0 = (int64_t)c << p.wshift;
    if (w0 + wdoubles > p.table_doubles) w0 = p.table_doubles - wdoubles;
    const double* __restrict__ xw = p.x + w0;
    for (int b = wid; b < nb; b += 16) {
      uint32_t sid[4] = {0, 0, 0, 0};
      double sv[4] = {1.0, 1.0, 1.0, 1.0};
      if (p.sidx) {
        // 12 B per element, coalesced, read once (non-temporal) - like the real (idx, val) stream
        const int64_t base = (((int64_t)blockIdx.x * p.nwin + c) * p.tile_elems + (int64_t)b * EPI);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (EPI == 256 || (EPI == 64 && j == 0) || (EPI == 32 && j == 0 && lane < 32)) {
            const int64_t q = base + lane + 64 * j;
            sid[j] = __builtin_nontemporal_load(p.sidx + q);
            sv[j] = __builtin_nontemporal_load(p.sval + q);
          }
        }
      }
      double v[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t gi = (uint32_t)b * 4 + j;                 // gather-instruction number inside the tile
        const uint32_t h = hsh(lane * 2654435761u + gi * 40503u + c * 9176u + blockIdx.x * 7919u + sid[j]);
        v[j][0] = v[j][1] = 0.0;
        if (G == 1) {
          const uint32_t e = gi * 64 + lane;
          const uint32_t bin = (uint32_t)(((uint64_t)e << (p.wshift - 5)) / (uint32_t)p.tile_elems);
          v[j][0] = xw[bin * 32 + (h & 31)];
        } else if (G == 6) {
          const uint32_t e = gi * 64 + lane;
          uint32_t col = (uint32_t)(((uint64_t)e << p.wshift) / (uint32_t)p.tile_elems) + (h % 7u);
          if (col >= (1u << p.wshift)) col = (1u << p.wshift) - 1;
          v[j][0] = xw[col];
        } else if (G == 2) {
          // lines of this gather-instruction's share of the window: 30 of them, 8 lanes each
          const uint32_t nlines = 1u << (p.wshift - 4);
          const uint32_t ngi = (uint32_t)p.tile_elems / 64;
          const uint32_t l0 = (uint32_t)(((uint64_t)gi * nlines) / ngi);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t li = t * 8 + (lane >> 3);
            if (li < 30) {
              uint32_t line = l0 + li;
              if (line >= nlines) line = nlines - 1;
              const double2 d = *reinterpret_cast<const double2*>(xw + (size_t)line * 16 + (lane & 7) * 2);
              v[j][0] += d.x; v[j][1] += d.y;
            }
          }
        } else if (G == 4) {
          const uint32_t hq = hsh((lane >> 2) * 2654435761u + gi * 40503u + c * 9176u + blockIdx.x * 7919u + sid[0]);
          const uint32_t col = hq & ((1u << (p.wshift - 3)) - 1);
          const double2 d = *reinterpret_cast<const double2*>(xw + (size_t)col * 8 + (lane & 3) * 2);
          v[j][0] = d.x; v[j][1] = d.y;
        } else if (G == 5) {
          const uint32_t hq = hsh((lane >> 3) * 2654435761u + gi * 40503u + c * 9176u + blockIdx.x * 7919u + sid[0]);
          const uint32_t col = hq & ((1u << (p.wshift - 3)) - 1);
          v[j][0] = xw[(size_t)col * 8 + (lane & 7)];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t gi = (uint32_t)b * 4 + j;
        const uint32_t h = hsh(lane * 40503u + gi * 2654435761u + c * 7919u + blockIdx.x * 9176u);
        if (L == 0) {
          sink += (v[j][0] + v[j][1]) * sv[j];
        } else if (L == 1) {
          __hip_atomic_fetch_add(yacc + (h % (uint32_t)p.rows), v[j][0] * sv[j] + v[j][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (L == 2 || L == 3) {
          // rows chosen per group of 4 (G4) or 8 (G5) lanes
          const int lanes_per = (G == 5) ? 8 : 4;
          const uint32_t hr = hsh((lane / lanes_per) * 40503u + gi * 2654435761u + c * 7919u + blockIdx.x * 9176u);
          const uint32_t row = hr % (uint32_t)p.rows;
          if (G == 5) {
            double* a = yacc + (size_t)row * 8 + (lane & 7);
            if (L == 2) __hip_atomic_fetch_add(a, v[j][0] * sv[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *a += v[j][0] * sv[0];
          } else {
            double* a = yacc + (size_t)row * 8 + (lane & 3) * 2;
            if (L == 2) {
              __hip_atomic_fetch_add(a, v[j][0] * sv[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(a + 1, v[j][1] * sv[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              double2 t = *reinterpret_cast<double2*>(a);
              t.x += v[j][0] * sv[0]; t.y += v[j][1] * sv[0];
              *reinterpret_cast<double2*>(a) = t;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (L != 0) for (int k = threadIdx.x; k < nrow_d; k += 1024) sink += yacc[k];
  if (sink == 12345.6789) p.out[0] = sink;
}

template <int G, int L>
static float run(P p, int ncu, int reps) {
  const size_t lds = (size_t)((G == 4 || G == 5) ? p.rows * 8 : p.rows) * sizeof(double);
  CK(hipFuncSetAttribute((const void*)forms_kernel<G, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((forms_kernel<G, L>), dim3(ncu), dim3(1024), lds, 0, p);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((forms_kernel<G, L>), dim3(ncu), dim3(1024), lds, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  return best;
}

__global__ void fill(double* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = 1e-3 * (double)(i % 1000);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, ncu);
  double* out; CK(hipMalloc(&out, 64));
  // ---- SpMV shape: x = 1e7 doubles, windows of 2^17, 17152 elements per tile, 77 windows ----
  {
    P p; p.table_doubles = 10000000; p.wshift = 17; p.nwin = 77; p.tile_elems = 17152; p.rows = 20224; p.out = out;
    double* x; CK(hipMalloc(&x, p.table_doubles * 8)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, x, p.table_doubles);
    p.x = x;
    const int64_t nel = (int64_t)ncu * p.nwin * p.tile_elems;
    uint32_t* si; double* sv; CK(hipMalloc(&si, nel * 4)); CK(hipMalloc(&sv, nel * 8));
    CK(hipMemset(si, 0, nel * 4)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, sv, nel);
    CK(hipDeviceSynchronize());
    printf("SpMV shape: %lld elements per launch (%.2f per CU x1e6), x %.0f MB\n", (long long)nel, nel / 1e6 / ncu, p.table_doubles * 8 / 1e6);
    for (int st = 0; st < 2; ++st) {
      p.sidx = st ? si : nullptr; p.sval = st ? sv : nullptr;
      printf(" stream %d | G1 bins: L0 %.3f L1 %.3f | G6 sorted: L0 %.3f L1 %.3f | G2 lines: L0 %.3f L1 %.3f ms\n", st,
             run<1, 0>(p, ncu, 3), run<1, 1>(p, ncu, 3), run<6, 0>(p, ncu, 3), run<6, 1>(p, ncu, 3),
             run<2, 0>(p, ncu, 3), run<2, 1>(p, ncu, 3));
    }
    // LDS scatter alone (no gathers): G=0 is not a form; emulate with a tiny table (everything L1-resident)
    {
      P q = p; q.table_doubles = 1 << 17; q.nwin = 77; q.sidx = nullptr; q.sval = nullptr;
      printf(" x L2/L1-resident (128 Ki doubles, one window): G1 L0 %.3f L1 %.3f | G2 L0 %.3f ms\n",
             run<1, 0>(q, ncu, 3), run<1, 1>(q, ncu, 3), run<2, 0>(q, ncu, 3));
    }
    CK(hipFree(si)); CK(hipFree(sv)); CK(hipFree(x));
  }
  // ---- SpMM k = 8 shape: X = 1e7 x 8 doubles, windows of 2^14 columns (1 MiB), 611 windows ----
  {
    P p; p.table_doubles = 80000000; p.wshift = 17; p.nwin = 611; p.tile_elems = 4160; p.rows = 2528; p.out = out;
    double* x; CK(hipMalloc(&x, p.table_doubles * 8)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, x, p.table_doubles);
    p.x = x;
    const int64_t nel = (int64_t)ncu * p.nwin * p.tile_elems;
    uint32_t* si; double* sv; CK(hipMalloc(&si, nel * 4)); CK(hipMalloc(&sv, nel * 8));
    CK(hipMemset(si, 0, nel * 4)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, sv, nel);
    CK(hipDeviceSynchronize());
    printf("SpMM k=8 shape: %lld non-zeros per launch, X %.0f MB, %d windows of 1 MiB\n", (long long)nel, p.table_doubles * 8 / 1e6, p.nwin);
    for (int st = 0; st < 2; ++st) {
      p.sidx = st ? si : nullptr; p.sval = st ? sv : nullptr;
      printf(" stream %d | G4 (4 lanes x 16 B): L0 %.3f L2 %.3f L3 %.3f | G5 (8 lanes x 8 B): L0 %.3f L2 %.3f L3 %.3f ms\n", st,
             run<4, 0>(p, ncu, 3), run<4, 2>(p, ncu, 3), run<4, 3>(p, ncu, 3),
             run<5, 0>(p, ncu, 3), run<5, 2>(p, ncu, 3), run<5, 3>(p, ncu, 3));
    }
    // windows of 2^15 columns (2 MiB) and 2^13 (512 KiB)
    for (int ws : {16, 18}) {
      P q = p; q.wshift = ws; q.nwin = (int)((p.table_doubles + ((int64_t)1 << ws) - 1) >> ws);
      q.tile_elems = (int)(nel / ncu / q.nwin) / 64 * 64; q.sidx = si; q.sval = sv;
      printf(" window 2^%d doubles (%d windows, %d nnz per tile), stream 1: G4 L0 %.3f L3 %.3f ms\n", ws, q.nwin, q.tile_elems,
             run<4, 0>(q, ncu, 3), run<4, 3>(q, ncu, 3));
    }
    {
      P q = p; q.table_doubles = 1 << 17; q.nwin = 611; q.sidx = nullptr; q.sval = nullptr;
      printf(" X one window only (L2-resident, no HBM): G4 L0 %.3f L2 %.3f L3 %.3f ms\n", run<4, 0>(q, ncu, 3), run<4, 2>(q, ncu, 3), run<4, 3>(q, ncu, 3));
    }
    CK(hipFree(si)); CK(hipFree(sv)); CK(hipFree(x));
  }
  // ---- SpMM k = 8 at N = 1e6 (BASELINE config #3): X = 64 MB (Infinity-Cache resident), 33 nnz/row ----
  {
    P p; p.table_doubles = 8000000; p.wshift = 17; p.nwin = 62; p.tile_elems = 1088; p.rows = 1953; p.out = out;
    double* x; CK(hipMalloc(&x, p.table_doubles * 8)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, x, p.table_doubles);
    p.x = x;
    const int64_t nel = (int64_t)ncu * p.nwin * p.tile_elems;
    uint32_t* si; double* sv; CK(hipMalloc(&si, nel * 4)); CK(hipMalloc(&sv, nel * 8));
    CK(hipMemset(si, 0, nel * 4)); hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, sv, nel);
    CK(hipDeviceSynchronize());
    printf("SpMM k=8, N=1e6 shape: %lld non-zeros per launch (half a product), X %.0f MB\n", (long long)nel, p.table_doubles * 8 / 1e6);
    for (int ws : {15, 16, 17, 18, 19}) {
      P q = p; q.wshift = ws; q.nwin = (int)((p.table_doubles + ((int64_t)1 << ws) - 1) >> ws);
      q.tile_elems = (int)(nel / ncu / q.nwin) / 64 * 64; q.sidx = si; q.sval = sv;
      printf(" window 2^%d doubles (%d windows, %d nnz per tile), stream 1: G4 L0 %.4f L2 %.4f L3 %.4f | G5 L2 %.4f ms\n", ws, q.nwin, q.tile_elems,
             run<4, 0>(q, ncu, 5), run<4, 2>(q, ncu, 5), run<4, 3>(q, ncu, 5), run<5, 2>(q, ncu, 5));
    }
    {
      P q = p; q.wshift = 22; q.nwin = 1; q.tile_elems = (int)(nel / ncu) / 64 * 64; q.sidx = si; q.sval = sv;
      printf(" no windows (random over 32 MB of X, Infinity Cache): G4 L0 %.4f L2 %.4f ms\n", run<4, 0>(q, ncu, 5), run<4, 2>(q, ncu, 5));
    }
    CK(hipFree(si)); CK(hipFree(sv)); CK(hipFree(x));
  }
  return 0;
}
