// Context, batch descriptor, memory helpers and error reporting of libworld_hip.so.
#include <math.h>
#include <string.h>

#include "wh_device.h"
#include "wh_host.h"
#include "wh_math.h"

namespace {
thread_local std::string g_last_error;

__global__ void frame_utt_kernel(const int64_t* __restrict__ frame_off, int n_utt, int64_t total, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int lo = 0, hi = n_utt;  // largest u with frame_off[u] <= i
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (frame_off[mid] <= i) lo = mid; else hi = mid;
    }
    out[i] = lo;
  }
}

// 16 bytes per thread and step; the tail (bytes % 16 == 8) by one thread
__global__ void copy_mapped_kernel(double2* __restrict__ dst, const double2* __restrict__ src, size_t n16, double* dst_tail,
                                   const double* src_tail) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
  if (dst_tail && blockIdx.x == 0 && threadIdx.x == 0) *dst_tail = *src_tail;
}
}  // namespace

namespace wh {
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(const char* where, hipError_t e) {
  g_last_error = std::string(where) + ": " + hipGetErrorString(e);
  // The runtime keeps the error of a failed call (an allocation that did not fit, say) as its "last error" until somebody
  // reads it — and every launch check of this library reads it (hipGetLastError behind a launch): left standing, the NEXT,
  // perfectly ordinary call reported it as its own ("wh_batch_create: out of memory" after a workspace that could not be
  // allocated; tools/oom_probe.py).  Reported once, here, and cleared.
  (void)hipGetLastError();
  return (int)e == 0 ? -1 : (int)e;
}
int fail_msg(const char* where, const char* msg) {
  g_last_error = std::string(where) + ": " + msg;
  return -1;
}
int ws_reserve(wh_ctx* ctx, size_t bytes) {
  // Every stage lays the arena out afresh through this call, so whatever time base an earlier wh_synthesis_timebase left
  // in it is gone from here on — also when the arena does not move (wh_synthesis_timebase sets the mark again after
  // its own reservation; wh_synthesis_render reserves nothing).
  ctx->timebase.valid = false;
  if (bytes <= ctx->ws_bytes) return 0;
  if (ctx->ws) {
    WH_CHECK(hipDeviceSynchronize());
    WH_CHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  size_t want = bytes + bytes / 4;
  WH_CHECK(hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return 0;
}
int const_table(wh_ctx* ctx, const std::string& key, const std::vector<double>& host, const double** out) {
  auto it = ctx->tables.find(key);
  if (it != ctx->tables.end()) {
    *out = it->second;
    return 0;
  }
  double* d = nullptr;
  WH_CHECK(hipMalloc((void**)&d, (host.size() ? host.size() : 1) * sizeof(double)));
  WH_CHECK(hipMemcpy(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
  ctx->tables[key] = d;
  ctx->table_bytes += (host.size() ? host.size() : 1) * sizeof(double);
  *out = d;
  return 0;
}
int tables_make_room(wh_ctx* ctx) {
  if (ctx->table_bytes <= kTableCacheBytes) return 0;
  WH_CHECK(hipDeviceSynchronize());  // kernels of earlier calls may still read them
  for (auto& kv : ctx->tables)
    if (kv.second) WH_CHECK(hipFree(kv.second));
  ctx->tables.clear();
  ctx->table_bytes = 0;
  return 0;
}
int persistent_upload(wh_ctx* ctx, hipStream_t st, const std::string& slot, const void* host, size_t bytes, void** dptr) {
  wh_ctx::Persist& e = ctx->persist[slot];
  if (e.d && e.host.size() == bytes && (bytes == 0 || memcmp(e.host.data(), host, bytes) == 0)) {
    *dptr = e.d;
    return 0;
  }
  if (e.cap < bytes || !e.d) {
    if (e.d) {
      WH_CHECK(hipDeviceSynchronize());  // kernels of earlier calls may still read the buffer that is about to go
      WH_CHECK(hipFree(e.d));
      e.d = nullptr;
    }
    const size_t want = bytes < 256 ? 256 : bytes + bytes / 4;
    WH_CHECK(hipMalloc(&e.d, want));
    e.cap = want;
  }
  if (bytes) {
    const int k = e.next_stage;
    e.next_stage = (k + 1) % wh_ctx::Persist::kStages;
    if (e.stage_done[k]) WH_CHECK(hipEventSynchronize(e.stage_done[k]));  // the copy that last used this staging buffer
    if (e.stage_cap[k] < bytes) {
      if (e.stage[k]) WH_CHECK(hipHostFree(e.stage[k]));
      e.stage[k] = nullptr;
      const size_t want = bytes < 256 ? 256 : bytes + bytes / 4;
      WH_CHECK(hipHostMalloc(&e.stage[k], want, hipHostMallocDefault));
      e.stage_cap[k] = want;
    }
    if (!e.stage_done[k]) WH_CHECK(hipEventCreateWithFlags(&e.stage_done[k], hipEventDisableTiming));
    memcpy(e.stage[k], host, bytes);
    WH_CHECK(hipMemcpyAsync(e.d, e.stage[k], bytes, hipMemcpyHostToDevice, st));
    WH_CHECK(hipEventRecord(e.stage_done[k], st));
  }
  e.host.assign(reinterpret_cast<const char*>(host), reinterpret_cast<const char*>(host) + bytes);
  *dptr = e.d;
  return 0;
}
int persistent_scratch(wh_ctx* ctx, const std::string& slot, size_t bytes, void** dptr) {
  wh_ctx::Persist& e = ctx->persist[slot];
  if (e.cap < bytes || !e.d) {
    if (e.d) {
      WH_CHECK(hipDeviceSynchronize());  // kernels of earlier calls may still use the buffer that is about to go
      WH_CHECK(hipFree(e.d));
      e.d = nullptr;
    }
    const size_t want = bytes < 256 ? 256 : bytes + bytes / 4;
    WH_CHECK(hipMalloc(&e.d, want));
    e.cap = want;
  }
  *dptr = e.d;
  return 0;
}
}  // namespace wh

// bounds build: the per-translation-unit readers of the kernels' out-of-range records, and the last record taken
static std::vector<int (*)(unsigned long long*)>& bounds_readers() {
  static std::vector<int (*)(unsigned long long*)> v;
  return v;
}
int wh::bounds_register(int (*reader)(unsigned long long*)) {
  bounds_readers().push_back(reader);
  return (int)bounds_readers().size();
}
static thread_local int64_t g_bounds_last[4] = {0, 0, 0, 0};

extern "C" {

int wh_version(void) { return 106; }
const char* wh_last_error(void) { return g_last_error.c_str(); }

int wh_device_count(int* count) {
  WH_CHECK(hipGetDeviceCount(count));
  return 0;
}

int wh_ctx_create(int device, wh_ctx** out) {
  if (!out) return wh::fail_msg("wh_ctx_create", "null out pointer");
  WH_CHECK(hipSetDevice(device));
  wh_ctx* c = new wh_ctx();
  c->device = device;
  // twiddle tables: for N = 2,4,..,WH_MAX_TWIDDLE the table exp(-2*pi*i*k/N), k<N, lives at [N, 2N)
  std::vector<double2> tw(2 * WH_MAX_TWIDDLE);
  tw[0] = tw[1] = make_double2(1.0, 0.0);
  for (int n = 2; n <= WH_MAX_TWIDDLE; n <<= 1) {
    for (int k = 0; k < n; ++k) {
      long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
      tw[n + k] = make_double2((double)cosl(a), (double)sinl(a));
    }
    // exact values on the axes
    tw[n] = make_double2(1.0, 0.0);
    if (n >= 2) tw[n + n / 2] = make_double2(-1.0, 0.0);
    if (n >= 4) {
      tw[n + n / 4] = make_double2(0.0, -1.0);
      tw[n + 3 * n / 4] = make_double2(0.0, 1.0);
    }
  }
  hipError_t e = hipMalloc((void**)&c->d_twiddle, tw.size() * sizeof(double2));
  if (e == hipSuccess) e = hipMemcpy(c->d_twiddle, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_flags, 16 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(c->d_flags, 0, 16 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_flag_cum, 16 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(c->d_flag_cum, 0, 16 * sizeof(int32_t));
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_flag_cum, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) memset((void*)c->h_flag_cum, 0, 16 * sizeof(int32_t));
  if (e != hipSuccess) {
    delete c;
    return wh::fail("wh_ctx_create", e);
  }
  *out = c;
  return 0;
}

int wh_ctx_destroy(wh_ctx* ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  if (ctx->d_flag_cum) (void)hipFree(ctx->d_flag_cum);
  if (ctx->h_flag_cum) (void)hipHostFree((void*)ctx->h_flag_cum);
  for (auto& kv : ctx->tables) (void)hipFree(kv.second);
  for (auto e : ctx->prof_events) (void)hipEventDestroy(e);
  for (auto& kv : ctx->persist) {
    if (kv.second.d) (void)hipFree(kv.second.d);
    for (int k = 0; k < wh_ctx::Persist::kStages; ++k) {
      if (kv.second.stage[k]) (void)hipHostFree(kv.second.stage[k]);
      if (kv.second.stage_done[k]) (void)hipEventDestroy(kv.second.stage_done[k]);
    }
  }
  delete ctx;
  return 0;
}

// Give the context's scratch back to the device: the workspace arena and the per-call buffers (utterance tables, overlap-
// add rows ...).  They only ever grow — a context that has served a 1024-utterance Harvest batch keeps ~100 GB — so a
// process that moves on to other work calls this between phases.  Constant tables, twiddles and flags stay; the next
// call allocates what it needs again.  Synchronises the device (kernels of earlier calls may still use the buffers).
int wh_ctx_trim(wh_ctx* ctx) {
  if (!ctx) return wh::fail_msg("wh_ctx_trim", "null ctx");
  WH_ENTER(ctx);
  WH_CHECK(hipDeviceSynchronize());
  if (ctx->ws) WH_CHECK(hipFree(ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  ctx->timebase.valid = false;
  for (auto& kv : ctx->persist) {
    wh_ctx::Persist& e = kv.second;
    if (e.d) WH_CHECK(hipFree(e.d));
    e.d = nullptr;
    e.cap = 0;
    e.host.clear();  // (nothing is resident any more: the next identical call uploads again)
  }
  return 0;
}

// what wh_flags_post has published and no poll / take has reported yet
static void drain_posted(wh_ctx* ctx, int32_t* h_flags16, bool accumulate) {
  for (int i = 0; i < 16; ++i) {
    const int32_t cum = ctx->h_flag_cum[i];
    const int32_t fresh = cum - ctx->h_flag_seen[i];
    ctx->h_flag_seen[i] = cum;
    h_flags16[i] = (accumulate ? h_flags16[i] : 0) | (fresh != 0 ? 1 : 0);
  }
}


int wh_bounds_build(void) {
#if defined(WH_BOUNDS) && WH_BOUNDS
  return 1;
#else
  return 0;
#endif
}

#if defined(WH_BOUNDS) && WH_BOUNDS
// positive control of the bounds build: four threads store into a four-element checked buffer, thread 3 one past its end
static __global__ void bounds_selftest_kernel(double* buf) {
  const wh::ckp<double> b = wh::ck_make(buf, 4, wh::WH_CK_TABLE);
  b[threadIdx.x == 3 ? 4 : threadIdx.x] = 1.0;
}
#endif
int wh_bounds_selftest(wh_ctx* ctx, void* stream) {
  if (!ctx) return wh::fail_msg("wh_bounds_selftest", "null ctx");
#if defined(WH_BOUNDS) && WH_BOUNDS
  WH_ENTER(ctx);
  if (int rc = wh::ws_reserve(ctx, 64)) return rc;
  hipLaunchKernelGGL(bounds_selftest_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, reinterpret_cast<double*>(ctx->ws));
  WH_LAUNCH_CHECK("bounds_selftest_kernel");
  return 0;
#else
  return wh::fail_msg("wh_bounds_selftest", "not a bounds build");
#endif
}

// the spectral kernels' own log / exp / sincospi (wh_math.h) on a device array: which = 0 log, 1 exp, 2 sincospi (out[2i], out[2i+1])
static __global__ void math_probe_kernel(int which, const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  if (which == 0) out[i] = wh::flog(x);
  else if (which == 1) out[i] = wh::fexp(x);
  else {
    const double2 sc = wh::fsincospi(x);
    out[2 * i] = sc.x;
    out[2 * i + 1] = sc.y;
  }
}
int wh_math_probe(wh_ctx* ctx, void* stream, int which, const double* in, double* out, int64_t n) {
  if (!ctx || !in || !out || n < 0 || which < 0 || which > 2) return wh::fail_msg("wh_math_probe", "bad argument");
  WH_ENTER(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, in, out, (long long)n);
  WH_LAUNCH_CHECK("math_probe_kernel");
  return 0;
}

int wh_bounds_last(int64_t* out4) {
  if (!out4) return wh::fail_msg("wh_bounds_last", "null argument");
  for (int i = 0; i < 4; ++i) out4[i] = g_bounds_last[i];
  return 0;
}

int wh_take_flags(wh_ctx* ctx, void* stream, int32_t* h_flags16) {
  if (!ctx || !h_flags16) return wh::fail_msg("wh_take_flags", "null argument");
  WH_ENTER(ctx);
  hipStream_t st = (hipStream_t)stream;
  WH_CHECK(hipMemcpyAsync(h_flags16, ctx->d_flags, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  WH_CHECK(hipMemsetAsync(ctx->d_flags, 0, 16 * sizeof(int32_t), st));
  WH_CHECK(hipStreamSynchronize(st));
  drain_posted(ctx, h_flags16, true);  // conditions an earlier wh_flags_post moved out of d_flags
  // (bounds build: the records of every translation unit's kernels; the device-wide wait orders them behind kernels of
  // other streams too — this is a test vehicle)
  for (int i = 0; i < 4; ++i) g_bounds_last[i] = 0;
  if (!bounds_readers().empty()) {
    WH_CHECK(hipDeviceSynchronize());
    for (auto rd : bounds_readers()) {
      unsigned long long rec[4] = {0, 0, 0, 0};
      if (rd(rec)) return wh::fail_msg("wh_take_flags", "reading a bounds record failed");
      if (rec[0]) {
        if (!g_bounds_last[0]) for (int i = 1; i < 4; ++i) g_bounds_last[i] = (int64_t)rec[i];
        g_bounds_last[0] += (int64_t)rec[0];
      }
    }
    if (g_bounds_last[0]) h_flags16[WH_FLAG_OOB] = 1;
  }
  return 0;
}

// One thread: flags that are set bump their cumulative counter, the counters go to the host mirror, the flags are cleared.
static __global__ void flags_post_kernel(int32_t* __restrict__ flags, int32_t* __restrict__ cum,
                                         volatile int32_t* __restrict__ mirror) {
  const int i = threadIdx.x;
  if (i >= 16) return;
  const int32_t v = flags[i];
  if (v) {
    flags[i] = 0;
    const int32_t c = cum[i] + 1;
    cum[i] = c;
    mirror[i] = c;
    __threadfence_system();
  }
}

int wh_flags_post(wh_ctx* ctx, void* stream, int discard) {
  if (!ctx) return wh::fail_msg("wh_flags_post", "null ctx");
  WH_ENTER(ctx);
  hipStream_t st = (hipStream_t)stream;
  if (discard) {
    WH_CHECK(hipMemsetAsync(ctx->d_flags, 0, 16 * sizeof(int32_t), st));
    return 0;
  }
  int32_t* d_mirror = nullptr;
  WH_CHECK(hipHostGetDevicePointer((void**)&d_mirror, (void*)ctx->h_flag_cum, 0));
  hipLaunchKernelGGL(flags_post_kernel, dim3(1), dim3(64), 0, st, ctx->d_flags, ctx->d_flag_cum, d_mirror);
  WH_LAUNCH_CHECK("flags_post_kernel");
  return 0;
}

int wh_flags_poll(wh_ctx* ctx, int32_t* h_flags16) {
  if (!ctx || !h_flags16) return wh::fail_msg("wh_flags_poll", "null argument");
  drain_posted(ctx, h_flags16, false);
  return 0;
}

int wh_profile_enable(wh_ctx* ctx, int on) {
  if (!ctx) return wh::fail_msg("wh_profile_enable", "null ctx");
  ctx->prof = on != 0;
  ctx->prof_names.clear();
  return 0;
}

int wh_profile_collect(wh_ctx* ctx, char* names, size_t names_bytes, float* ms, int max_records, int* n_records) {
  if (!ctx || !names || !ms || !n_records) return wh::fail_msg("wh_profile_collect", "null argument");
  WH_ENTER(ctx);
  WH_CHECK(hipDeviceSynchronize());
  const int n = (int)ctx->prof_names.size();
  std::string joined;
  int out = 0;
  for (int i = 0; i < n && out < max_records; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]) != hipSuccess) t = -1.f;
    ms[out++] = t;
    joined += ctx->prof_names[i];
    joined += '\n';
  }
  if (joined.size() + 1 > names_bytes) return wh::fail_msg("wh_profile_collect", "names buffer too small");
  memcpy(names, joined.c_str(), joined.size() + 1);
  *n_records = out;
  ctx->prof_names.clear();
  return 0;
}

int wh_malloc(void** dptr, size_t bytes) {
  WH_CHECK(hipMalloc(dptr, bytes ? bytes : 8));
  return 0;
}
int wh_free(void* dptr) {
  if (dptr) WH_CHECK(hipFree(dptr));
  return 0;
}
int wh_memcpy_h2d(void* dst, const void* h_src, size_t bytes, void* stream) {
  WH_CHECK(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int wh_memcpy_d2h(void* h_dst, const void* src, size_t bytes, void* stream) {
  WH_CHECK(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  WH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int wh_memset(void* dst, int value, size_t bytes, void* stream) {
  WH_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return 0;
}
int wh_stream_sync(void* stream) {
  WH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int wh_host_alloc(void** h_ptr, size_t bytes) {
  if (!h_ptr) return wh::fail_msg("wh_host_alloc", "null argument");
  WH_CHECK(hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault));
  return 0;
}
int wh_host_free(void* h_ptr) {
  if (h_ptr) WH_CHECK(hipHostFree(h_ptr));
  return 0;
}
int wh_copy_mapped(wh_ctx* ctx, void* stream, void* dst, const void* src, size_t bytes, int max_blocks) {
  if (!ctx || (bytes && (!dst || !src))) return wh::fail_msg("wh_copy_mapped", "null argument");
  if (bytes % 8 || ((uintptr_t)dst % 16) || ((uintptr_t)src % 16))
    return wh::fail_msg("wh_copy_mapped", "bytes must be a multiple of 8 and both pointers 16-byte aligned");
  if (!bytes) return 0;
  WH_ENTER(ctx);
  hipStream_t st = (hipStream_t)stream;
  const size_t n16 = bytes / 16;
  const bool tail = bytes % 16 != 0;
  int blocks = max_blocks > 0 ? max_blocks : 64;
  const size_t need = (n16 + 255) / 256;
  if ((size_t)blocks > need) blocks = need ? (int)need : 1;
  wh::KernelTimer t(ctx, st, "copy_mapped_kernel");
  copy_mapped_kernel<<<blocks, 256, 0, st>>>((double2*)dst, (const double2*)src, n16,
                                            tail ? (double*)dst + 2 * n16 : nullptr,
                                            tail ? (const double*)src + 2 * n16 : nullptr);
  WH_LAUNCH_CHECK("copy_mapped_kernel");
  return 0;
}

int64_t wh_num_frames(int64_t n_samples, double fs, double frame_period_ms) {
  return (int64_t)(1000.0 * (double)n_samples / fs / frame_period_ms + 1.0);
}

int wh_batch_create(wh_ctx* ctx, int n_utt, const int64_t* h_x_off, const int64_t* h_frame_off, wh_batch** out) {
  if (!ctx || !out || !h_x_off || !h_frame_off) return wh::fail_msg("wh_batch_create", "null argument");
  if (n_utt < 1) return wh::fail_msg("wh_batch_create", "n_utt must be >= 1");
  for (int u = 0; u < n_utt; ++u) {
    if (h_x_off[u + 1] < h_x_off[u] || h_frame_off[u + 1] < h_frame_off[u])
      return wh::fail_msg("wh_batch_create", "offsets must be non-decreasing");
  }
  WH_CHECK(hipSetDevice(ctx->device));
  wh_batch* b = new wh_batch();
  b->ctx = ctx;
  b->n_utt = n_utt;
  b->h_x_off.assign(h_x_off, h_x_off + n_utt + 1);
  b->h_frame_off.assign(h_frame_off, h_frame_off + n_utt + 1);
  b->total_samples = h_x_off[n_utt] - h_x_off[0];
  b->total_frames = h_frame_off[n_utt] - h_frame_off[0];
  size_t ob = (size_t)(n_utt + 1) * sizeof(int64_t);
  hipError_t e = hipMalloc((void**)&b->d_x_off, ob);
  if (e == hipSuccess) e = hipMalloc((void**)&b->d_frame_off, ob);
  if (e == hipSuccess) e = hipMalloc((void**)&b->d_frame_utt, (size_t)(b->total_frames > 0 ? b->total_frames : 1) * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(b->d_x_off, h_x_off, ob, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(b->d_frame_off, h_frame_off, ob, hipMemcpyHostToDevice);
  if (e == hipSuccess && b->total_frames > 0) {
    int blocks = (int)((b->total_frames + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    frame_utt_kernel<<<blocks, 256>>>(b->d_frame_off, n_utt, b->total_frames, b->d_frame_utt);
    e = hipGetLastError();
    // The descriptor is complete when this call returns (it has no stream argument: whichever stream uses the batch next
    // finds the table written).  Only the null stream is waited for — a device-wide wait here made every batch created
    // while another context's kernels were running (a second pipeline on its own non-blocking stream) wait for them.
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (e != hipSuccess) {
    wh_batch_destroy(b);
    return wh::fail("wh_batch_create", e);
  }
  *out = b;
  return 0;
}

int wh_batch_destroy(wh_batch* b) {
  if (!b) return 0;
  if (b->d_x_off) (void)hipFree(b->d_x_off);
  if (b->d_frame_off) (void)hipFree(b->d_frame_off);
  if (b->d_frame_utt) (void)hipFree(b->d_frame_utt);
  delete b;
  return 0;
}

}  // extern "C"
