// Host-side plumbing shared by the C-ABI translation units: context, batch descriptor,
// error reporting, workspace.  Not part of the public ABI (see include/world_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/world_hip.h"

struct wh_ctx {
  int device = 0;
  double2* d_twiddle = nullptr;  // tables for N = 2 .. WH_MAX_TWIDDLE, table of size N at offset N
  void* ws = nullptr;            // growable scratch
  size_t ws_bytes = 0;
  std::map<std::string, double*> tables;  // small constant tables resident on the device (windows, taps)
  size_t table_bytes = 0;                 // their total; tables_make_room() empties the cache beyond kTableCacheBytes
  int32_t* d_flags = nullptr;             // [16] sticky device-side condition flags (see wh_take_flags)
  // deferred reading of the flags (wh_flags_post / wh_flags_poll): d_flag_cum[i] counts the posts that found flag i set;
  // h_flag_cum is its mirror in pinned, device-mapped host memory (the post kernel is its only writer, the host only
  // reads it), h_flag_seen what the host has reported so far
  int32_t* d_flag_cum = nullptr;          // [16]
  volatile int32_t* h_flag_cum = nullptr; // [16] pinned
  int32_t h_flag_seen[16] = {0};
  // small per-call tables (utterance metadata, filter taps) kept in their own device buffers together with
  // the host bytes they were filled from: an identical call re-uses them without any copy or stream sync
  struct Persist {
    void* d = nullptr;
    size_t cap = 0;
    std::vector<char> host;
    // changed content is staged in pinned host memory and copied with hipMemcpyAsync on the call's stream: the copy is
    // ordered behind every kernel of earlier calls (which may still read the old content) without a device sync
    static constexpr int kStages = 4;
    void* stage[kStages] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[kStages] = {0, 0, 0, 0};
    hipEvent_t stage_done[kStages] = {nullptr, nullptr, nullptr, nullptr};
    int next_stage = 0;
  };
  std::map<std::string, Persist> persist;
  // what the last wh_synthesis_timebase left in this context's workspace (wh_synthesis_render of ANY context reads it)
  struct TimeBase {
    bool valid = false;
    int n_utt = 0;
    int64_t pulse_cap = 0, ny_tot = 0, frames = 0;
    size_t o_vuv = 0, o_pt = 0, o_pi = 0, o_ps = 0, o_pn = 0, o_pc = 0, o_pb = 0, o_rec = 0;
  } timebase;
  // Harvest's zero-crossing lists (wh_harvest_set_event_caps / wh_harvest_event_counts): the capacities the NEXT
  // wh_harvest takes instead of its estimate (one-shot; empty: estimate, hv_caps_worst: ylen/2 + 2 for every list), and
  // where the last call left its per-(utterance, channel, train) counts — a buffer of its own, not the shared scratch,
  // so that they can still be read after the stages behind Harvest have run
  std::vector<int64_t> hv_caps_next;
  bool hv_caps_worst = false;
  int32_t* hv_last_cnt = nullptr;
  int64_t hv_last_cnt_lists = 0;  // utterances x channels of that call
  // optional per-kernel timing (HIP events on the launch stream), see wh_profile_*
  bool prof = false;
  std::vector<hipEvent_t> prof_events;    // pool, two per record
  std::vector<const char*> prof_names;    // one per record
};

struct wh_batch {
  wh_ctx* ctx = nullptr;
  int n_utt = 0;
  int64_t total_samples = 0;
  int64_t total_frames = 0;
  std::vector<int64_t> h_x_off, h_frame_off;
  int64_t* d_x_off = nullptr;      // [n_utt+1]
  int64_t* d_frame_off = nullptr;  // [n_utt+1]
  int32_t* d_frame_utt = nullptr;  // [total_frames]
};

#define WH_MAX_FFT 8192  // longest in-LDS transform (CheapTrick / synthesis stop at 4096, D4C / love-train / SWIPE' at 8192)
// The twiddle tables go further: StoneMask and the Harvest refinement evaluate a handful of bins of a transform of
// 2^(2 + floor(log2(window))) points directly (stonemask.py:33-35) — no transform is run, only exp(-2 pi i k / n) is
// looked up — and a 3-period window at a low floor and a high rate asks for 16384 or 32768 (96 kHz below 70 Hz, 48 kHz
// below 35 Hz: the fft_size override).  1 MB per context.
#define WH_MAX_TWIDDLE 32768

namespace wh {
// bounds build: every translation unit registers a reader of its kernels' out-of-range record (wh_device.h)
int bounds_register(int (*reader)(unsigned long long*));
void set_error(const std::string& msg);
int fail(const char* where, hipError_t e);
int fail_msg(const char* where, const char* msg);
int ws_reserve(wh_ctx* ctx, size_t bytes);  // grows ctx->ws (hipFree + hipMalloc => implicit sync)
inline const double2* twiddle(const wh_ctx* ctx, int n) { return ctx->d_twiddle + n; }
// Upload-once constant table keyed by name (synchronous on first use, cached afterwards).
int const_table(wh_ctx* ctx, const std::string& key, const std::vector<double>& host, const double** out);
// The cache is keyed by what the tables depend on — rate, window bound, F0 floor — and a long-running caller that varies
// those (StoneMask on contours with ever new minima: a window table of up to 2 MB per distinct bound) would grow it without
// limit.  Called at the top of the entry points that cache large tables, BEFORE any table pointer is taken: beyond
// kTableCacheBytes everything is freed (one device synchronisation) and rebuilt on demand.
constexpr size_t kTableCacheBytes = (size_t)256 << 20;
int tables_make_room(wh_ctx* ctx);
// Upload `bytes` from host memory into the persistent device buffer named `slot` unless it already holds
// exactly these bytes.  Changed content goes through a ring of pinned staging buffers and hipMemcpyAsync on `st`
// (no device synchronisation; a context is driven from one stream at a time, so stream order protects readers of
// the old content).  Only growing the buffer synchronises (hipFree).
int persistent_upload(wh_ctx* ctx, hipStream_t st, const std::string& slot, const void* host, size_t bytes, void** dptr);
// A persistent device buffer of at least `bytes` under the name `slot`, contents unspecified (per-call scratch whose size
// follows the batch shape).  Growing it synchronises the device once (earlier kernels may still use the old buffer).
int persistent_scratch(wh_ctx* ctx, const std::string& slot, size_t bytes, void** dptr);
template <typename T>
inline int persistent_upload(wh_ctx* ctx, hipStream_t st, const std::string& slot, const std::vector<T>& v, T** dptr) {
  void* p = nullptr;
  const int rc = persistent_upload(ctx, st, slot, v.data(), v.size() * sizeof(T), &p);
  *dptr = reinterpret_cast<T*>(p);
  return rc;
}
// Opt a kernel into more than 64 KiB of dynamic LDS (gfx950: up to 160 KiB per workgroup).
template <typename K>
inline int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 65536) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : fail("hipFuncSetAttribute", e);
}
// RAII bracket around one kernel launch: records a start/stop event pair on the launch stream when
// profiling is enabled (zero cost otherwise).
// WH_TRACE_LAUNCH=1 (environment, debugging): every launch is announced on stderr and waited for — the last name printed
// before an abort (a device-side sanitizer finding, a memory fault) is the kernel that caused it.
inline bool trace_launches() {
  static const bool on = getenv("WH_TRACE_LAUNCH") && getenv("WH_TRACE_LAUNCH")[0] == '1';
  return on;
}
struct KernelTimer {
  wh_ctx* c;
  hipStream_t st;
  int slot = -1;
  KernelTimer(wh_ctx* ctx, hipStream_t s, const char* name) : c(ctx), st(s) {
    if (trace_launches()) {
      fprintf(stderr, "[wh] launch %s\n", name);
      fflush(stderr);
    }
    if (!c->prof) return;
    const size_t rec = c->prof_names.size();
    while (c->prof_events.size() < 2 * (rec + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      c->prof_events.push_back(e);
    }
    c->prof_names.push_back(name);
    slot = (int)rec;
    (void)hipEventRecord(c->prof_events[2 * slot], st);
  }
  ~KernelTimer() {
    if (slot >= 0) (void)hipEventRecord(c->prof_events[2 * slot + 1], st);
    if (trace_launches()) (void)hipStreamSynchronize(st);
  }
};
}  // namespace wh

#define WH_CHECK(expr)                                   \
  do {                                                   \
    hipError_t _e = (expr);                              \
    if (_e != hipSuccess) return wh::fail(#expr, _e);    \
  } while (0)

// First statement of every C-ABI entry that allocates or launches: make the context's device current on the calling
// thread (a process may hold contexts for several GPUs; allocations and launches follow the current device).
#define WH_ENTER(ctx) WH_CHECK(hipSetDevice((ctx)->device))

#define WH_LAUNCH_CHECK(name)                            \
  do {                                                   \
    hipError_t _e = hipGetLastError();                   \
    if (_e != hipSuccess) return wh::fail(name, _e);     \
  } while (0)
