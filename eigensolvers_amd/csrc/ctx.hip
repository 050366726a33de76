// Context, device memory and timing entry points of libhipeig.so.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[1024] = "";

void hipeig_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hipeig_last_error(void) { return g_err; }

extern "C" int hipeig_ctx_create(int device, hipeig_ctx** out) {
  HIPEIG_REQUIRE(out != nullptr, "null output");
  int count = 0;
  HIPEIG_CHECK(hipGetDeviceCount(&count));
  HIPEIG_REQUIRE(device >= 0 && device < count, "no such HIP device");
  HIPEIG_CHECK(hipSetDevice(device));
  hipeig_ctx* c = (hipeig_ctx*)calloc(1, sizeof(hipeig_ctx));
  HIPEIG_REQUIRE(c != nullptr, "out of host memory");
  c->device = device;
  hipDeviceProp_t prop;
  HIPEIG_CHECK(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount;
  HIPEIG_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPEIG_CHECK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  HIPEIG_CHECK(hipEventCreate(&c->ev0));
  HIPEIG_CHECK(hipEventCreate(&c->ev1));
  HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
  HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming));
  for (int k = 0; k < HIPEIG_GATHER_MAX_CHUNKS; ++k) HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_chunk[k], hipEventDisableTiming));
  for (int k = 0; k < 8; ++k) HIPEIG_CHECK(hipEventCreate(&c->ev_ph[k]));
  for (int k = 0; k < 16; ++k) HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_slot[k], hipEventDisableTiming));
  HIPEIG_CHECK(hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
  HIPEIG_CHECK(hipMalloc((void**)&c->d_counters, 4 * HIPEIG_TICKET_WORDS * sizeof(unsigned)));
  HIPEIG_CHECK(hipMemset(c->d_counters, 0, 4 * HIPEIG_TICKET_WORDS * sizeof(unsigned)));
  HIPEIG_CHECK(hipMalloc((void**)&c->d_group_partials, (size_t)32 * 1024 * sizeof(double)));
  c->partials_doubles = (size_t)HIPEIG_MAX_PARTIALS * HIPEIG_MAX_COLS * HIPEIG_MAX_COLS * 2;   // 8 MiB
  HIPEIG_CHECK(hipMalloc((void**)&c->d_partials, c->partials_doubles * sizeof(double)));
  c->scalars_doubles = 4096;
  HIPEIG_CHECK(hipMalloc((void**)&c->d_scalars, c->scalars_doubles * sizeof(double)));
  HIPEIG_CHECK(hipHostMalloc((void**)&c->h_scalars, c->scalars_doubles * sizeof(double),
                             hipHostMallocDefault));
  {
    void* dev = nullptr;
    const char* e = getenv("HIPEIG_MAPPED_SCALARS");            // 0: always copy results back (default 1)
    if (!(e && atoi(e) == 0) && hipHostGetDevicePointer(&dev, c->h_scalars, 0) == hipSuccess) c->h_scalars_dev = (double*)dev;
    else (void)hipGetLastError();
  }
  c->ptrs_count = 1024;
  HIPEIG_CHECK(hipMalloc((void**)&c->d_ptrs, c->ptrs_count * sizeof(double*)));
  HIPEIG_CHECK(hipHostMalloc((void**)&c->h_ptrs, c->ptrs_count * sizeof(double*),
                             hipHostMallocDefault));
  HIPEIG_CHECK(hipMalloc((void**)&c->d_mr_state, 4 * sizeof(MinresState)));
  HIPEIG_CHECK(hipHostMalloc((void**)&c->h_mr_state, 4 * sizeof(MinresState),
                             hipHostMallocDefault));
  c->nranks = 1;
  c->rank = 0;
  // hipGraph replay of the MINRES chunk is opt-in: it gains <= 7 % and only at N ~ 1e5..1e6, and
  // rocprofv3 --kernel-trace crashes inside the capture on this ROCm (7.2) when it is on.
  const char* g = getenv("HIPEIG_GRAPH");
  c->use_graph = g ? atoi(g) : 0;
  *out = c;
  return 0;
}

extern "C" int hipeig_ctx_destroy(hipeig_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->comm_stream);
  if (c->comm || c->direct) hipeig_comm_destroy(c);
  hipFree(c->d_partials);
  hipFree(c->d_scalars);
  hipHostFree(c->h_scalars);
  if (c->h_arn_items) hipHostFree(c->h_arn_items);
  for (int k = 0; k < 16; ++k) if (c->arn_stream[k]) { hipStreamSynchronize(c->arn_stream[k]); hipStreamDestroy(c->arn_stream[k]); }
  for (int k = 0; k < 16; ++k) if (c->ev_arn_in[k]) hipEventDestroy(c->ev_arn_in[k]);
  if (c->d_arn_ws) hipFree(c->d_arn_ws);
  if (c->d_arn_cnt) hipFree(c->d_arn_cnt);
  hipFree(c->d_ptrs);
  hipHostFree(c->h_ptrs);
  hipFree(c->d_mr_state);
  hipHostFree(c->h_mr_state);
  if (c->mr_graph) hipGraphExecDestroy(c->mr_graph);
  free(c->mr_graph_key);
  if (c->mr_ws) hipFree(c->mr_ws);
  if (c->x_full) hipFree(c->x_full);
  if (c->ytmp) hipFree(c->ytmp);
  if (c->blk_ws) hipFree(c->blk_ws);
  if (c->xb_full) hipFree(c->xb_full);
  if (c->mrb_ws) hipFree(c->mrb_ws);
  if (c->d_mrb_state) hipFree(c->d_mrb_state);
  if (c->h_mrb_state) hipHostFree(c->h_mrb_state);
  free(c->row_counts);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  hipEventDestroy(c->ev_comm);
  hipEventDestroy(c->ev_x);
  for (int k = 0; k < HIPEIG_GATHER_MAX_CHUNKS; ++k) hipEventDestroy(c->ev_chunk[k]);
  for (int k = 0; k < 8; ++k) hipEventDestroy(c->ev_ph[k]);
  for (int k = 0; k < 16; ++k) hipEventDestroy(c->ev_slot[k]);
  hipEventDestroy(c->ev_stage);
  hipFree(c->d_counters);
  hipFree(c->d_group_partials);
  hipStreamDestroy(c->stream);
  hipStreamDestroy(c->comm_stream);
  free(c);
  return 0;
}

extern "C" int hipeig_ctx_sync(hipeig_ctx* c) {
  for (int k = 0; k < 16; ++k)
    if (c->arn_stream[k]) HIPEIG_CHECK(hipStreamSynchronize(c->arn_stream[k]));
  if (hipeig_sync_checked(c)) return 4;
  return 0;
}

extern "C" int hipeig_device_info(hipeig_ctx* c, int64_t info[8], char* name, int name_len) {
  hipDeviceProp_t prop;
  HIPEIG_CHECK(hipGetDeviceProperties(&prop, c->device));
  size_t fr = 0, tot = 0;
  HIPEIG_CHECK(hipMemGetInfo(&fr, &tot));
  info[0] = prop.multiProcessorCount;
  info[1] = prop.warpSize;
  info[2] = (int64_t)tot;
  info[3] = (int64_t)fr;
  info[4] = prop.l2CacheSize;
  info[5] = prop.clockRate;
  info[6] = prop.memoryClockRate;
  info[7] = prop.memoryBusWidth;
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return 0;
}

// ---- vectors --------------------------------------------------------------------------
extern "C" int hipeig_vec_alloc(hipeig_ctx* c, int64_t n, double** out) {
  HIPEIG_REQUIRE(n >= 0 && out, "bad arguments");
  void* p = nullptr;
  // 256-byte granularity keeps every vector 16-byte aligned for the double2 loads.
  HIPEIG_CHECK(hipMalloc(&p, (size_t)(n > 0 ? n : 1) * sizeof(double)));
  *out = (double*)p;
  return 0;
}

extern "C" int hipeig_vec_free(hipeig_ctx* c, double* v) {
  if (!v) return 0;
  // pending kernels on the compute stream may still read v
  if (hipeig_sync_checked(c)) return 4;
  HIPEIG_CHECK(hipFree(v));
  return 0;
}

extern "C" int hipeig_vec_upload(hipeig_ctx* c, double* dst, const double* src, int64_t n) {
  HIPEIG_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  return 0;
}

extern "C" int hipeig_vec_download(hipeig_ctx* c, double* dst, const double* src, int64_t n) {
  HIPEIG_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (hipeig_sync_checked(c)) return 4;
  return 0;
}

// A copy kernel instead of hipMemcpyAsync: the runtime's device-to-device copy runs at 4.8-5.3 TB/s on MI355X, a
// plain 16-byte-per-thread kernel on n/512 workgroups at 6.0-6.6 (tools/stream_forms_bench.hip).
__global__ void __launch_bounds__(HIPEIG_BLOCK)
copy_kernel(int64_t n, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t n2 = n >> 1;
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) d2[i] = s2[i];
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}

extern "C" int hipeig_vec_copy(hipeig_ctx* c, double* dst, const double* src, int64_t n) {
  if (n == 0 || dst == src) return 0;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) != 0) {      // not 16-byte aligned (never the case for pool buffers)
    HIPEIG_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  hipLaunchKernelGGL(copy_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, n, src, dst);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

__global__ void fill_kernel(double* __restrict__ v, int64_t n, double value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = value;
}

extern "C" int hipeig_vec_fill(hipeig_ctx* c, double* v, int64_t n, double value) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_stream(n)), dim3(HIPEIG_BLOCK), 0, c->stream, v, n, value);
  HIPEIG_CHECK(hipGetLastError());
  return 0;
}

// ---- timing ---------------------------------------------------------------------------
extern "C" int hipeig_timer_start(hipeig_ctx* c) {
  HIPEIG_CHECK(hipEventRecord(c->ev0, c->stream));
  return 0;
}

extern "C" int hipeig_timer_stop(hipeig_ctx* c, float* ms) {
  HIPEIG_CHECK(hipEventRecord(c->ev1, c->stream));
  HIPEIG_CHECK(hipEventSynchronize(c->ev1));
  HIPEIG_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}
