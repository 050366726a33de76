// Minimal stand-alone reproduction attempt for "rocprofv3 --kernel-trace aborts inside a hipGraph
// capture" (round-1 commit 0336101 switched HIPEIG_GRAPH off by default because of it).  It mimics
// what hipeig_minres captures - a chain of kernels with > 64 KiB of dynamic LDS and 1024-thread
// workgroups plus small 256-thread kernels, thread-local capture mode, an optional device-to-pinned-host
// copy node - with nothing of the library in it, so that a crash here is the tool's and not ours.
//   graph_repro [copy=0|1] [lds_kb=0..158] [mode=0 global|1 threadlocal|2 relaxed] [iters] [replays]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); exit(2);} } while (0)

__global__ void __launch_bounds__(1024) big_kernel(double* x, int n) {
  extern __shared__ double lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  lds[threadIdx.x] = (double)i;
  __syncthreads();
  if (i < n) x[i] += lds[(threadIdx.x + 1) % blockDim.x] * 1e-9;
}
__global__ void small_kernel(double* x, double* rec, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= 1.0000001;
  if (i == 0) rec[0] += 1.0;
}

int main(int argc, char** argv) {
  const int copy = argc > 1 ? atoi(argv[1]) : 1;
  const int lds_kb = argc > 2 ? atoi(argv[2]) : 158;
  const int mode = argc > 3 ? atoi(argv[3]) : 1;
  const int iters = argc > 4 ? atoi(argv[4]) : 18;
  const int replays = argc > 5 ? atoi(argv[5]) : 20;
  const int n = 1 << 20;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *x, *rec, *hrec;
  CK(hipMalloc(&x, n * sizeof(double))); CK(hipMalloc(&rec, 256)); CK(hipHostMalloc(&hrec, 256, hipHostMallocDefault));
  CK(hipMemset(x, 0, n * sizeof(double))); CK(hipMemset(rec, 0, 256));
  const size_t lds = (size_t)(lds_kb > 8 ? lds_kb : 8) * 1024;
  CK(hipFuncSetAttribute((const void*)big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  const hipStreamCaptureMode m = mode == 0 ? hipStreamCaptureModeGlobal : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
  CK(hipStreamBeginCapture(s, m));
  for (int k = 0; k < iters; ++k) {
    hipLaunchKernelGGL(big_kernel, dim3(n / 1024), dim3(1024), lds, s, x, n);
    hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, s, x, rec, n);
    hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, s, x, rec, n);
  }
  if (copy) CK(hipMemcpyAsync(hrec, rec, 64, hipMemcpyDeviceToHost, s));
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  for (int r = 0; r < replays; ++r) {
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    if ((r + 1) % 50 == 0) { fprintf(stderr, "replay %d done\n", r + 1); fflush(stderr); }
  }
  if (!copy) CK(hipMemcpy(hrec, rec, 64, hipMemcpyDeviceToHost));
  printf("graph_repro copy=%d lds_kb=%d mode=%d iters=%d replays=%d: ok, record %.0f (expected %d)\n", copy, lds_kb, mode, iters, replays,
         hrec[0], replays * 2 * iters);
  CK(hipGraphExecDestroy(ge));
  return 0;
}
