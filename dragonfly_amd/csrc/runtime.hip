// Context, memory and error plumbing of libdfhip.so, plus the small elementwise / reduction
// kernels every stage shares.
#include "common.h"
#include <cstdlib>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <mutex>
#include <set>

static thread_local char g_err[1024] = "";

void dfh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dfh_last_error(void) { return g_err; }
extern "C" int dfh_abi_version(void) { return DFH_ABI_VERSION; }

extern "C" int dfh_device_count(int* count) {
  DFH_ARG(count != nullptr);
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c = 0;
  }
  *count = c;
  return DFH_OK;
}

// live contexts: a dfh_gp released after its context was destroyed must not touch the context
static std::mutex g_live_mu;
static std::set<const dfh_ctx*> g_live_ctx;
bool ctx_is_live(const dfh_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live_ctx.count(ctx) != 0;
}

extern "C" int dfh_ctx_create(int device, dfh_ctx** out) {
  DFH_ARG(out != nullptr);
  *out = nullptr;
  int count = 0;
  DFH_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count || device >= DFH_MAX_DEVICES) {
    dfh_set_error("dfh_ctx_create: device %d out of range (found %d HIP devices)", device, count);
    return DFH_ERR_BAD_ARG;
  }
  DFH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  DFH_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    dfh_set_error("libdfhip.so is built for gfx950 (MI355X) only; device %d is %s", device,
                  prop.gcnArchName);
    return DFH_ERR_HIP;
  }
  dfh_ctx* ctx = new dfh_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
           prop.multiProcessorCount);
  DFH_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->main_stream = ctx->stream;
  {
    int least = 0, greatest = 0;
    DFH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    DFH_HIP(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, greatest));
    DFH_HIP(hipStreamCreateWithPriority(&ctx->bulk, hipStreamNonBlocking, least));
    DFH_HIP(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    DFH_HIP(hipStreamCreateWithFlags(&ctx->bulk_normal, hipStreamNonBlocking));
  }
  DFH_HIP(hipEventCreate(&ctx->ev0));
  DFH_HIP(hipEventCreate(&ctx->ev1));
  for (int i = 0; i < DFH_T_COUNT; ++i) {
    DFH_HIP(hipEventCreate(&ctx->tev0[i]));
    DFH_HIP(hipEventCreate(&ctx->tev1[i]));
  }
  ctx->scratch.resize(SCR_COUNT);
  DFH_HIP(hipMalloc(&ctx->d_info, (CHOL_MAX_BATCH + 16) * sizeof(int64_t)));
  DFH_HIP(hipMemset(ctx->d_info, 0, (CHOL_MAX_BATCH + 16) * sizeof(int64_t)));
  DFH_HIP(hipHostMalloc(&ctx->h_info, (CHOL_MAX_BATCH + 16) * sizeof(int64_t)));
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ctx.insert(ctx);
  }
  {
    // idle blocks the cache may keep: 1/16 of the device (18 GB of the MI355X's 288), at least 2 GiB -- the
    // buffers of an n = 16384 fit (L alone is 2 GiB) have to fit, or every refit pays a 2 GiB hipFree + hipMalloc
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 16 > ctx->pool_idle_limit) ctx->pool_idle_limit = total_b / 16;
    else (void)hipGetLastError();
  }
  if (const char* e = getenv("DFH_POOL_MAX_MIB")) ctx->pool_idle_limit = size_t(strtoull(e, nullptr, 10)) << 20;
  *out = ctx;
  return DFH_OK;
}

extern "C" void dfh_ctx_destroy(dfh_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!g_live_ctx.erase(ctx)) return;          // not (or no longer) a live context
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->scratch)
    if (b.p) (void)hipFree(b.p);
  for (auto& b : ctx->pool_idle) (void)hipFree(b.p);     // blocks still held by live GPs stay theirs
  if (ctx->d_info) (void)hipFree(ctx->d_info);
  if (ctx->h_info) (void)hipHostFree(ctx->h_info);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  for (int i = 0; i < DFH_T_COUNT; ++i) {
    (void)hipEventDestroy(ctx->tev0[i]);
    (void)hipEventDestroy(ctx->tev1[i]);
  }
  for (auto e : ctx->evpool) (void)hipEventDestroy(e);
  for (auto& r : ctx->gemm_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->bulk) (void)hipStreamDestroy(ctx->bulk);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->bulk_normal) (void)hipStreamDestroy(ctx->bulk_normal);
  (void)hipStreamDestroy(ctx->main_stream);
  delete ctx;
}

extern "C" int dfh_mem_info(dfh_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
  DFH_ARG(ctx != nullptr);
  DFH_HIP(hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  DFH_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (uint64_t)f;
  if (total_bytes) *total_bytes = (uint64_t)t;
  return DFH_OK;
}

extern "C" int dfh_sync(dfh_ctx* ctx) {
  DFH_ARG(ctx != nullptr);
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->side));
  DFH_HIP(hipStreamSynchronize(ctx->bulk));
  DFH_HIP(hipStreamSynchronize(ctx->aux));
  return DFH_OK;
}

extern "C" int dfh_device_name(dfh_ctx* ctx, char* buf, size_t buflen) {
  DFH_ARG(ctx && buf && buflen > 0);
  snprintf(buf, buflen, "%s", ctx->name);
  return DFH_OK;
}

int pinned_get(dfh_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->h_stage_bytes) {
    if (ctx->h_stage) { DFH_HIP(hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
    size_t cap = 1 << 16;
    while (cap < bytes) cap <<= 1;
    // mapped + coherent: kernels of the small-call paths read descriptors from it and write results into it directly
    DFH_HIP(hipHostMalloc(&ctx->h_stage, cap, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable));
    void* dev_view = nullptr;
    DFH_HIP(hipHostGetDevicePointer(&dev_view, ctx->h_stage, 0));
    if (dev_view != ctx->h_stage) {        // (one address space on this platform; anything else is not supported here)
      dfh_set_error("pinned staging buffer: device view %p differs from host address %p", dev_view, ctx->h_stage);
      return DFH_ERR_HIP;
    }
    ctx->h_stage_bytes = cap;
  }
  *out = ctx->h_stage;
  return DFH_OK;
}

// Capacity classes: powers of two up to 4 MiB, multiples of 2 MiB beyond, so that the buffers of
// consecutive fits of similar size land in the same class.
static size_t pool_capacity(size_t bytes) {
  if (bytes <= 256) return 256;
  if (bytes <= (size_t(4) << 20)) {
    size_t cap = 256;
    while (cap < bytes) cap <<= 1;
    return cap;
  }
  const size_t step = size_t(2) << 20;
  return (bytes + step - 1) / step * step;
}

int dev_alloc(dfh_ctx* ctx, size_t bytes, void** out) {
  *out = nullptr;
  const size_t cap = pool_capacity(bytes);
  for (size_t i = 0; i < ctx->pool_idle.size(); ++i) {
    if (ctx->pool_idle[i].cap == cap) {
      *out = ctx->pool_idle[i].p;
      ctx->pool_idle_bytes -= cap;
      ctx->pool_idle[i] = ctx->pool_idle.back();
      ctx->pool_idle.pop_back();
      return DFH_OK;
    }
  }
  hipError_t e = hipMalloc(out, cap);
  if (e != hipSuccess && !ctx->pool_idle.empty()) {
    // out of memory with blocks parked in the cache: give them back and try once more
    (void)hipGetLastError();
    for (auto& b : ctx->pool_idle) { ctx->pool_caps.erase(b.p); (void)hipFree(b.p); }
    ctx->pool_idle.clear();
    ctx->pool_idle_bytes = 0;
    e = hipMalloc(out, cap);
  }
  if (e != hipSuccess) {
    dfh_set_error("%s:%d: hipMalloc(%zu) -> %s", __FILE__, __LINE__, cap, hipGetErrorString(e));
    return DFH_ERR_HIP;
  }
  ctx->pool_caps[*out] = cap;
  return DFH_OK;
}

void dev_release(dfh_ctx* ctx, void* p) {
  if (!p) return;
  if (ctx == nullptr || !ctx_is_live(ctx)) { (void)hipFree(p); return; }   // context gone: nothing to cache in
  auto it = ctx->pool_caps.find(p);
  if (it == ctx->pool_caps.end()) { (void)hipFree(p); return; }
  const size_t cap = it->second;
  if (ctx->pool_idle_bytes + cap <= ctx->pool_idle_limit) {
    ctx->pool_idle.push_back({p, cap});
    ctx->pool_idle_bytes += cap;
    return;
  }
  ctx->pool_caps.erase(it);
  (void)hipFree(p);
}

extern "C" int dfh_malloc(dfh_ctx* ctx, size_t bytes, void** dptr) {
  DFH_ARG(ctx && dptr);
  DFH_HIP(hipSetDevice(ctx->device));
  return dev_alloc(ctx, bytes ? bytes : 8, dptr);
}

extern "C" int dfh_free(dfh_ctx* ctx, void* dptr) {
  if (!dptr) return DFH_OK;
  const bool live = ctx && ctx_is_live(ctx);
  if (live) DFH_HIP(hipStreamSynchronize(ctx->stream));   // else: context already destroyed
  dev_release(live ? ctx : nullptr, dptr);
  return DFH_OK;
}

extern "C" int dfh_memcpy_h2d(dfh_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DFH_ARG(ctx && (bytes == 0 || (dst && src)));
  if (!bytes) return DFH_OK;
  DFH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_memcpy_d2h(dfh_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DFH_ARG(ctx && (bytes == 0 || (dst && src)));
  if (!bytes) return DFH_OK;
  DFH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_timer_begin(dfh_ctx* ctx) {
  DFH_ARG(ctx != nullptr);
  DFH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_timer_end(dfh_ctx* ctx, double* ms) {
  DFH_ARG(ctx && ms);
  DFH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  DFH_HIP(hipEventSynchronize(ctx->ev1));
  float f = 0.f;
  DFH_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
  *ms = (double)f;
  return DFH_OK;
}

extern "C" int dfh_ctx_counters(dfh_ctx* ctx, int64_t* out) {
  DFH_ARG(ctx && out);
  out[0] = ctx->chol_fallbacks;
  out[1] = ctx->chol_cooldown;
  out[2] = 0; out[3] = 0;
  return DFH_OK;
}

extern "C" int dfh_ctx_timings(dfh_ctx* ctx, int enable, double* ms_out) {
  DFH_ARG(ctx != nullptr);
  if (ms_out) memcpy(ms_out, ctx->t_ms, sizeof(ctx->t_ms));
  ctx->timing = enable != 0;
  memset(ctx->t_ms, 0, sizeof(ctx->t_ms));
  return DFH_OK;
}

SectionTimer::SectionTimer(dfh_ctx* c, int w) : ctx(c), which(w), on(c->timing) {
  if (on) (void)hipEventRecord(ctx->tev0[which], ctx->stream);
}
SectionTimer::~SectionTimer() {
  if (!on) return;
  (void)hipEventRecord(ctx->tev1[which], ctx->stream);
  (void)hipEventSynchronize(ctx->tev1[which]);
  float f = 0.f;
  if (hipEventElapsedTime(&f, ctx->tev0[which], ctx->tev1[which]) == hipSuccess) ctx->t_ms[which] += f;
}

int ctx_event(dfh_ctx* ctx, size_t idx, hipEvent_t* out) {
  while (ctx->evpool.size() <= idx) {
    hipEvent_t e;
    DFH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->evpool.push_back(e);
  }
  *out = ctx->evpool[idx];
  return DFH_OK;
}

int scratch_get(dfh_ctx* ctx, int slot, size_t bytes, void** out) {
  DevBuf& b = ctx->scratch[slot];
  if (b.bytes < bytes || !b.p) {
    if (b.p) {
      DFH_HIP(hipStreamSynchronize(ctx->main_stream));
      DFH_HIP(hipStreamSynchronize(ctx->side));
      DFH_HIP(hipStreamSynchronize(ctx->bulk));
      DFH_HIP(hipStreamSynchronize(ctx->aux));
      DFH_HIP(hipFree(b.p));
      b.p = nullptr; b.bytes = 0;
    }
    // growing slots get the pool's capacity classes (powers of two up to 4 MiB, multiples of 2 MiB
    // beyond): a sequence of slowly growing n / m does not pay a device-wide sync + hipMalloc per call
    size_t want = pool_capacity(bytes);
    DFH_HIP(hipMalloc(&b.p, want));
    b.bytes = want;
  }
  *out = b.p;
  return DFH_OK;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // plain host memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

int to_device(dfh_ctx* ctx, const void* p, size_t bytes, int slot, const double** out) {
  if (is_device_ptr(p)) {
    *out = static_cast<const double*>(p);
    return DFH_OK;
  }
  void* d = nullptr;
  DFH_TRY(scratch_get(ctx, slot, bytes, &d));
  if (bytes) {
    DFH_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable source: the copy is staged before return, but keep ordering explicit
    DFH_HIP(hipStreamSynchronize(ctx->stream));
  }
  *out = static_cast<const double*>(d);
  return DFH_OK;
}

int from_device(dfh_ctx* ctx, void* user_dst, const void* dev_src, size_t bytes) {
  if (!bytes || user_dst == dev_src) return DFH_OK;
  if (is_device_ptr(user_dst)) {
    DFH_HIP(hipMemcpyAsync(user_dst, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    DFH_HIP(hipMemcpyAsync(user_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void k_fill(double* p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
int fill_f64(dfh_ctx* ctx, double* p, int64_t n, double v) {
  if (n <= 0) return DFH_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, n, v);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// zero the strict upper triangle (numpy.linalg.cholesky returns zeros there)
__global__ void k_zero_upper(double* A, int64_t n, int64_t lda) {
  const int64_t row = blockIdx.y;
  int64_t col = (int64_t)blockIdx.x * blockDim.x * 2 + threadIdx.x * 2;
  // vectorised where the pair is fully above the diagonal
  if (col + 1 < n && col > row && (lda & 1) == 0) {
    *reinterpret_cast<double2_t*>(A + row * lda + col) = (double2_t){0.0, 0.0};
  } else {
    if (col < n && col > row) A[row * lda + col] = 0.0;
    if (col + 1 < n && col + 1 > row) A[row * lda + col + 1] = 0.0;
  }
}
int zero_upper(dfh_ctx* ctx, double* A, int64_t n, int64_t lda) {
  if (n <= 1) return DFH_OK;
  dim3 grid((unsigned)((n + 511) / 512), (unsigned)n);
  hipLaunchKernelGGL(k_zero_upper, grid, dim3(256), 0, ctx->stream, A, n, lda);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

__global__ void k_add_diag(double* A, int64_t n, int64_t lda, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * lda + i] += v;
}
int add_diag(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double v) {
  if (n <= 0) return DFH_OK;
  hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, A, n, lda, v);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// max over the diagonal (np.diag(M).max(), general_utils.py:184); NaN propagates like numpy
__global__ void k_diag_max(const double* A, int64_t n, int64_t lda, double* out) {
  __shared__ double sm[256];
  __shared__ int snan[256];
  double m = -INFINITY;
  int has_nan = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    double v = A[i * lda + i];
    if (v != v) has_nan = 1;
    m = v > m ? v : m;
  }
  sm[threadIdx.x] = m;
  snan[threadIdx.x] = has_nan;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
      snan[threadIdx.x] |= snan[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = snan[0] ? NAN : sm[0];
}
int diag_max(dfh_ctx* ctx, const double* A, int64_t n, int64_t lda, double* host_out) {
  double* d = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_RED, 256, (void**)&d));
  hipLaunchKernelGGL(k_diag_max, dim3(1), dim3(256), 0, ctx->stream, A, n, lda, d);
  DFH_LAUNCH_CHECK();
  DFH_HIP(hipMemcpyAsync(host_out, d, 8, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

__global__ void k_copy_matrix(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                              int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t r = blockIdx.y;
  int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (c + 1 < cols && ((lds | ldd) & 1) == 0 &&
      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    *reinterpret_cast<double2_t*>(dst + r * ldd + c) =
        *reinterpret_cast<const double2_t*>(src + r * lds + c);
  } else {
    if (c < cols) dst[r * ldd + c] = src[r * lds + c];
    if (c + 1 < cols) dst[r * ldd + c + 1] = src[r * lds + c + 1];
  }
}
int copy_matrix(dfh_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd,
                int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return DFH_OK;
  // grid.y is limited to 65535: loop over row slabs
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    int64_t rr = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((unsigned)((cols + 511) / 512), (unsigned)rr);
    hipLaunchKernelGGL(k_copy_matrix, grid, dim3(256), 0, ctx->stream, src + r0 * lds, lds,
                       dst + r0 * ldd, ldd, rr, cols);
    DFH_LAUNCH_CHECK();
  }
  return DFH_OK;
}

// dst[c][r] = src[r][c] through a padded LDS tile (coalesced both sides)
__global__ void k_transpose(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                            int64_t ldd, int64_t rows, int64_t cols) {
  __shared__ double tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  for (int i = ty; i < 32; i += 8) {
    int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * lds + c] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[c * ldd + r] = tile[tx][i];
  }
}
int transpose_matrix(dfh_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd,
                     int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return DFH_OK;
  for (int64_t r0 = 0; r0 < rows; r0 += 32 * 65535LL) {
    int64_t rr = rows - r0 < 32 * 65535LL ? rows - r0 : 32 * 65535LL;
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rr + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, src + r0 * lds, lds,
                       dst + r0, ldd, rr, cols);
    DFH_LAUNCH_CHECK();
  }
  return DFH_OK;
}

// yout[i] = beta*yin[i] + alpha * sum_j A[i][j] x[j] : one 256-thread block per row, fixed
// reduction tree (deterministic)
__global__ void k_gemv_rows(const double* __restrict__ A, int64_t n, int64_t lda,
                            const double* __restrict__ x, double alpha, const double* yin,
                            double beta, double* yout, int tri_lower) {
  __shared__ double sm[4];
  const int64_t row = blockIdx.x;
  const double* a = A + row * lda;
  if (tri_lower && row + 1 < n) n = row + 1;       // only columns j <= row
  double s0 = 0.0, s1 = 0.0;
  const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    const int64_t n2 = n >> 1;
    for (int64_t j = threadIdx.x; j < n2; j += blockDim.x) {
      double2_t av = reinterpret_cast<const double2_t*>(a)[j];
      double2_t xv = reinterpret_cast<const double2_t*>(x)[j];
      s0 = fma(av.x, xv.x, s0);
      s1 = fma(av.y, xv.y, s1);
    }
    if ((n & 1) && threadIdx.x == 0) s0 = fma(a[n - 1], x[n - 1], s0);
  } else {
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) s0 = fma(a[j], x[j], s0);
  }
  double s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = alpha * ((sm[0] + sm[1]) + (sm[2] + sm[3]));
    if (beta != 0.0) v += beta * yin[row];
    yout[row] = v;
  }
}
// short rows: one wave per row, four rows per workgroup
__global__ void k_gemv_rows_wave(const double* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                                 const double* __restrict__ x, double alpha, const double* yin,
                                 double beta, double* yout) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int lane = threadIdx.x & 63;
  const double* a = A + row * lda;
  double s0 = 0.0, s1 = 0.0;
  const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    const int64_t n2 = n >> 1;
    for (int64_t j = lane; j < n2; j += 64) {
      double2_t av = reinterpret_cast<const double2_t*>(a)[j];
      double2_t xv = reinterpret_cast<const double2_t*>(x)[j];
      s0 = fma(av.x, xv.x, s0);
      s1 = fma(av.y, xv.y, s1);
    }
    if ((n & 1) && lane == 0) s0 = fma(a[n - 1], x[n - 1], s0);
  } else {
    for (int64_t j = lane; j < n; j += 64) s0 = fma(a[j], x[j], s0);
  }
  double s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    double v = alpha * s;
    if (beta != 0.0) v += beta * yin[row];
    yout[row] = v;
  }
}

int gemv_rows(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, const double* x,
              double alpha, const double* yin, double beta, double* yout, bool tri_lower) {
  if (m <= 0) return DFH_OK;
  if (!tri_lower && n <= 1024) {
    hipLaunchKernelGGL(k_gemv_rows_wave, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, A, m,
                       n, lda, x, alpha, yin, beta, yout);
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  hipLaunchKernelGGL(k_gemv_rows, dim3((unsigned)m), dim3(256), 0, ctx->stream, A, n, lda, x,
                     alpha, yin, beta, yout, tri_lower ? 1 : 0);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// A^T x : stage 1, each block owns a slab of GC_ROWS rows and 512 columns (2 per thread) and
// writes one partial per column; stage 2 adds the slab partials in slab order (deterministic).
#define GC_ROWS 32
__global__ void k_gemv_cols_partial(const double* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                                    const double* __restrict__ x, double* __restrict__ part) {
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * GC_ROWS;
  const int64_t r1 = r0 + GC_ROWS < m ? r0 + GC_ROWS : m;
  double s0 = 0.0, s1 = 0.0;
  if (c + 1 < n && (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) {
    for (int64_t r = r0; r < r1; ++r) {
      const double xr = x[r];
      double2_t av = *reinterpret_cast<const double2_t*>(A + r * lda + c);
      s0 = fma(av.x, xr, s0);
      s1 = fma(av.y, xr, s1);
    }
  } else {
    for (int64_t r = r0; r < r1; ++r) {
      const double xr = x[r];
      if (c < n) s0 = fma(A[r * lda + c], xr, s0);
      if (c + 1 < n) s1 = fma(A[r * lda + c + 1], xr, s1);
    }
  }
  if (c < n) part[(int64_t)blockIdx.y * n + c] = s0;
  if (c + 1 < n) part[(int64_t)blockIdx.y * n + c + 1] = s1;
}
__global__ void k_gemv_cols_reduce(const double* __restrict__ part, int64_t nslab, int64_t n,
                                   double alpha, const double* yin, double beta, double* yout) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0;
  for (int64_t k = 0; k < nslab; ++k) s += part[k * n + c];
  double v = alpha * s;
  if (beta != 0.0) v += beta * yin[c];
  yout[c] = v;
}
int gemv_cols(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, const double* x,
              double alpha, const double* yin, double beta, double* yout) {
  if (n <= 0) return DFH_OK;
  const int64_t nslab = (m + GC_ROWS - 1) / GC_ROWS;
  double* part = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_RED, (size_t)(nslab > 0 ? nslab : 1) * n * 8, (void**)&part));
  if (nslab > 0) {
    dim3 grid((unsigned)((n + 511) / 512), (unsigned)nslab);
    hipLaunchKernelGGL(k_gemv_cols_partial, grid, dim3(256), 0, ctx->stream, A, m, n, lda, x, part);
    DFH_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_gemv_cols_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, part, nslab, n, alpha, yin, beta, yout);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

__global__ void k_row_sumsq(const double* __restrict__ A, int64_t n, int64_t lda,
                            double* __restrict__ out) {
  __shared__ double sm[4];
  const int64_t row = blockIdx.x;
  const double* a = A + row * lda;
  double s0 = 0.0, s1 = 0.0;
  const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  if (vec) {
    const int64_t n2 = n >> 1;
    for (int64_t j = threadIdx.x; j < n2; j += blockDim.x) {
      double2_t av = reinterpret_cast<const double2_t*>(a)[j];
      s0 = fma(av.x, av.x, s0);
      s1 = fma(av.y, av.y, s1);
    }
    if ((n & 1) && threadIdx.x == 0) s0 = fma(a[n - 1], a[n - 1], s0);
  } else {
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) s0 = fma(a[j], a[j], s0);
  }
  double s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[row] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
int row_sumsq(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, double* out) {
  if (m <= 0) return DFH_OK;
  hipLaunchKernelGGL(k_row_sumsq, dim3((unsigned)m), dim3(256), 0, ctx->stream, A, n, lda, out);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// sum(log(diag L)) and dot(a,b) in one single-block pass (n is at most a few 10^4)
__global__ void k_logdet_dot(const double* L, int64_t n, int64_t ldl, const double* a,
                             const double* b, double* out) {
  __shared__ double s1[256], s2[256];
  double ld = 0.0, dt = 0.0;
  // (every diagonal entry is a cache line of its own: eight loads in flight per thread instead of one memory latency
  //  per term; the terms are still added in the same order -- the sum's bits do not change)
  int64_t i = threadIdx.x;
  for (; i + 7 * (int64_t)blockDim.x < n; i += 8 * (int64_t)blockDim.x) {
    double lv[8], av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t k = i + u * (int64_t)blockDim.x;
      lv[u] = L[k * ldl + k]; av[u] = a[k]; bv[u] = b[k];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      ld += log(lv[u]);
      dt = fma(av[u], bv[u], dt);
    }
  }
  for (; i < n; i += blockDim.x) {
    ld += log(L[i * ldl + i]);
    dt = fma(a[i], b[i], dt);
  }
  s1[threadIdx.x] = ld;
  s2[threadIdx.x] = dt;
  __syncthreads();
  for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      s1[threadIdx.x] += s1[threadIdx.x + s];
      s2[threadIdx.x] += s2[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s1[0]; out[1] = s2[0]; }
}
// Large n (round 6): every diagonal entry of L sits in a page of its own (a row is 128 KB at n = 16384) and one CU's
// address translation was what the single-workgroup pass waited for (44 us at n = 16384 whatever the thread count).
// Stage 1: workgroup g sums the terms of indices [g * LD_CHUNK, (g + 1) * LD_CHUNK) as k_logdet_dot does; stage 2 adds
// the partial sums in workgroup order.  Deterministic.
constexpr int64_t LD_CHUNK = 256;
__global__ __launch_bounds__(256) void k_logdet_dot_part(const double* L, int64_t n, int64_t ldl, const double* a,
                                                         const double* b, double* part) {
  __shared__ double s1[256], s2[256];
  const int64_t i = (int64_t)blockIdx.x * LD_CHUNK + threadIdx.x;
  s1[threadIdx.x] = i < n ? log(L[i * ldl + i]) : 0.0;
  s2[threadIdx.x] = i < n ? a[i] * b[i] : 0.0;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      s1[threadIdx.x] += s1[threadIdx.x + s];
      s2[threadIdx.x] += s2[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
__global__ __launch_bounds__(64) void k_logdet_dot_sum(const double* part, int count, double* out) {
  if (threadIdx.x == 0) {
    double ld = 0.0, dt = 0.0;
    for (int g = 0; g < count; ++g) { ld += part[2 * g]; dt += part[2 * g + 1]; }
    out[0] = ld; out[1] = dt;
  }
}
int logdet_and_dot_device(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* a,
                          const double* b, double* d_out2) {
  if (n > 2048) {
    const int count = (int)((n + LD_CHUNK - 1) / LD_CHUNK);
    double* part = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_RED2, (size_t)count * 16, (void**)&part));
    hipLaunchKernelGGL(k_logdet_dot_part, dim3((unsigned)count), dim3(256), 0, ctx->stream, L, n, ldl, a, b, part);
    DFH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_logdet_dot_sum, dim3(1), dim3(64), 0, ctx->stream, part, count, d_out2);
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  hipLaunchKernelGGL(k_logdet_dot, dim3(1), dim3(256), 0, ctx->stream, L, n, ldl, a, b, d_out2);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}
int logdet_and_dot(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* a,
                   const double* b, double* host_logdet, double* host_dot) {
  double* d = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_RED, 256, (void**)&d));
  DFH_TRY(logdet_and_dot_device(ctx, L, n, ldl, a, b, d));
  double h[2];
  DFH_HIP(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  *host_logdet = h[0];
  *host_dot = h[1];
  return DFH_OK;
}
