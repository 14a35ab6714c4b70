// Harvest back end: contour tracking (FixF0Contour), smoothing (SmoothF0) and the pick onto the output
// frame grid.  Included by wh_harvest.hip inside its anonymous namespace (needs HvUtt, kRows).
// Reference: world/harvest.py:301-559.  The reference walks one utterance at a time in Python; here
// everything that is independent runs in parallel (frames, voiced sections, merge scoring) and only the
// genuinely sequential bookkeeping (ordering/merging of a few dozen sections) is done by one lane.
#pragma once

struct HcUtt {  // per-utterance slices of the contour workspace (element offsets)
  int64_t f_base;    // 6 rows of nf1 doubles: base, s1, s2, s3, s4, smoothed
  int64_t run_base;  // run lists: 2 x rcap int32 (starts, ends)
  int64_t rcap;
  int64_t sec_base;  // section records (HcSec), rcap of them
  int64_t ch_base;   // channel windows, 32*nf1 + 1024 doubles
  int64_t ch_cap;
};

struct HcSec {
  int32_t st, ed;    // voiced run in s2
  int32_t w0, wlen;  // channel window covers frames [w0, w0+wlen)
  int64_t ch_off;    // offset of the window inside the utterance's channel area
  int32_t r0, r1;    // extended range
  int32_t kept;
  int32_t pad_;
};

// The refined candidates of a 1 ms frame are a LIST in the reference's row order (hv_refine_kernel): lst[frame] =
// (first slot in the pf0 / psc pools << 8) | count; bit k of the frame's 128-bit keep mask says entry k survived
// hv_prune_kernel.  The contour kernels read the lists through the mask instead of pruned copies.  Only the ORDER of a
// frame's candidates enters the reference's rules (np.argmax: first maximum; SelectBestF0: last minimum), and a pruned
// or absent candidate is a zero there, which can win neither.
struct HcList {
  const double* f0;
  const double* sc;
  int n;
};
__device__ __forceinline__ HcList hc_list(const int64_t* __restrict__ lst, const double* __restrict__ pf0,
                                          const double* __restrict__ psc, int64_t fr) {
  const int64_t ent = lst[fr];
  HcList l;
  l.f0 = pf0 + (ent >> 8);
  l.sc = psc + (ent >> 8);
  l.n = (int)(ent & 255);
  return l;
}

inline size_t contour_workspace_bytes(int64_t f1_tot, int n_utt) {
  // rows + runs + sections + channels, generously aligned
  return (size_t)f1_tot * 6 * 8 + (size_t)(f1_tot + 16 * n_utt) * 2 * 4 + (size_t)(f1_tot / 2 + 16 * n_utt) * sizeof(HcSec) +
         (size_t)(f1_tot * 32 + 1024 * n_utt) * 8 + sizeof(HcUtt) * n_utt + 4096;
}

__global__ __launch_bounds__(256) void hc_base_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                      const double* __restrict__ pf0, const double* __restrict__ psc,
                                                      const int64_t* __restrict__ lst,
                                                      const uint32_t* __restrict__ keep, double* __restrict__ rows) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m.nf1) return;
  const HcList l = hc_list(lst, pf0, psc, m.f1_off + j);
  uint32_t kw[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) kw[w] = keep[(m.f1_off + j) * 4 + w];
  // np.argmax of the pruned scores (first maximum).  A surviving candidate's score is >= 2.5 (hv_refine_row), every
  // other row of the reference's map holds 0: the first maximum is the first largest surviving score, or a zero row
  // (f0 = 0) when nothing survives.  Scores are fetched four at a time (independent loads).
  int best = -1;
  double bs = 0.0;
  for (int e0 = 0; e0 < l.n; e0 += 4) {
    double sv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = l.sc[e0 + q < l.n ? e0 + q : l.n - 1];  // clamped (a conditional load is a branch: four of them, a chain of waits)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q;
      const double se = (e < l.n && ((kw[e >> 5] >> (e & 31)) & 1u)) ? sv[q] : 0.0;
      if (se > bs) {
        bs = se;
        best = e;
      }
    }
  }
  rows[hc[blockIdx.y].f_base + j] = best >= 0 ? l.f0[best] : 0.0;
}

__global__ __launch_bounds__(256) void hc_step1_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                       double* __restrict__ rows) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m.nf1) return;
  const double* base = rows + hc[blockIdx.y].f_base;
  double* s1 = rows + hc[blockIdx.y].f_base + m.nf1;
  const double EPS = 2.220446049250313e-16;
  double v = base[j];
  if (j < 2) v = 0.0;
  else if (v != 0.0) {
    const double ref = base[j - 1] * 2 - base[j - 2];
    if (fabs((v - ref) / (ref + EPS)) > 0.008 && fabs((v - base[j - 1]) / (base[j - 1] + EPS)) > 0.008) v = 0.0;
  }
  s1[j] = v;
}

// Ordered list of voiced runs of a[0..n): starts[k]..ends[k] inclusive.  With force_ends the first and
// last frame count as unvoiced (GetBoundaryList, harvest.py:572-580).  Block-wide; returns the run count
// (clamped to cap) to every thread.  Contains barriers.
__device__ __forceinline__ int find_runs(const double* __restrict__ a, int64_t n, bool force_ends,
                                         int32_t* __restrict__ starts, int32_t* __restrict__ ends, int cap,
                                         int* sh /* >= 16 ints of LDS */) {
  int base_s = 0, base_e = 0;
  auto voiced = [&](int64_t j) -> bool {
    if (j < 0 || j >= n) return false;
    if (force_ends && (j == 0 || j == n - 1)) return false;
    return a[j] != 0.0;
  };
  for (int64_t t0 = 0; t0 < n; t0 += 256) {
    const int64_t j = t0 + threadIdx.x;
    const bool v = voiced(j);
    const bool is_s = v && !voiced(j - 1);
    const bool is_e = v && !voiced(j + 1);
    const unsigned long long ms = __ballot(is_s), me = __ballot(is_e);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
      sh[w] = __popcll(ms);
      sh[4 + w] = __popcll(me);
    }
    __syncthreads();
    int off_s = base_s, off_e = base_e, tot_s = 0, tot_e = 0;
    for (int i = 0; i < 4; ++i) {
      if (i < w) {
        off_s += sh[i];
        off_e += sh[4 + i];
      }
      tot_s += sh[i];
      tot_e += sh[4 + i];
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (is_s) {
      const int p = off_s + __popcll(ms & below);
      if (p < cap) starts[p] = (int32_t)j;
    }
    if (is_e) {
      const int p = off_e + __popcll(me & below);
      if (p < cap) ends[p] = (int32_t)j;
    }
    base_s += tot_s;
    base_e += tot_e;
  }
  __threadfence_block();
  __syncthreads();
  return base_s < cap ? base_s : cap;
}

// step 2 (drop runs shorter than 6 frames) and the section table for step 3
__global__ __launch_bounds__(256) void hc_sections_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                          double* __restrict__ rows, int32_t* __restrict__ runs,
                                                          HcSec* __restrict__ secs, int32_t* __restrict__ nsec_out) {
  __shared__ int sh[16];
  const HvUtt m = meta[blockIdx.x];
  const HcUtt h = hc[blockIdx.x];
  const int64_t n = m.nf1;
  const double* s1 = rows + h.f_base + n;
  double* s2 = rows + h.f_base + 2 * n;
  int32_t* starts = runs + h.run_base;
  int32_t* ends = starts + h.rcap;
  for (int64_t j = threadIdx.x; j < n; j += 256) s2[j] = s1[j];
  __threadfence_block();
  __syncthreads();
  int nr = find_runs(s1, n, true, starts, ends, (int)h.rcap, sh);
  for (int k = threadIdx.x; k < nr; k += 256) {
    if (ends[k] - starts[k] < 6)
      for (int j = starts[k]; j <= ends[k]; ++j) s2[j] = 0.0;
  }
  __threadfence_block();
  __syncthreads();
  nr = find_runs(s2, n, true, starts, ends, (int)h.rcap, sh);
  HcSec* sc = secs + h.sec_base;
  if (threadIdx.x == 0) {
    int64_t off = 0;
    int kept_n = 0;
    for (int k = 0; k < nr; ++k) {
      HcSec s;
      s.st = starts[k];
      s.ed = ends[k];
      int w0 = s.st - 101;
      if (w0 < 0) w0 = 0;
      int w1 = s.ed + 101;
      if (w1 > n - 1) w1 = (int)n - 1;
      s.w0 = w0;
      s.wlen = w1 - w0 + 1;
      s.ch_off = off;
      s.r0 = s.st;
      s.r1 = s.ed;
      s.kept = 0;
      s.pad_ = 0;
      if (off + s.wlen > h.ch_cap) break;  // cannot happen with runs >= 6 frames (32x head-room)
      off += s.wlen;
      sc[k] = s;
      ++kept_n;
    }
    nsec_out[blockIdx.x] = kept_n;
  }
}

// SelectBestF0 (harvest.py:238-248) over one frame's kRows candidates, wave-parallel: c0 / c1 are this lane's two
// candidates (rows lane and lane + 64, 0 beyond kRows).  A frame holds a handful of non-zero candidates and at most a
// few within the allowed range, so the cross-lane stage walks the lanes that hold one (ballot + readlane, scalar) instead
// of a six-step butterfly on three values.  Smallest error wins, the later row on ties.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double select_best_regs(double ref, double c0, double c1, double allowed) {
  const int lane = threadIdx.x & 63;
  double best_err = INFINITY, best_val = 0.0;
  int best_idx = -1;
  {
    const double err = fabs(ref - c0) / ref;
    if (!(err > allowed)) {
      best_err = err;
      best_val = c0;
      best_idx = lane;
    }
  }
  if (lane + 64 < kRows) {
    const double err = fabs(ref - c1) / ref;
    if (!(err > allowed) && !(err > best_err)) {  // later entries win ties
      best_err = err;
      best_val = c1;
      best_idx = lane + 64;
    }
  }
  unsigned long long mk = __ballot(best_idx >= 0);
  double be = INFINITY, bv = 0.0;
  int bi = -1;
  while (mk) {
    const int l = __ffsll((long long)mk) - 1;
    mk &= mk - 1;
    const double oe = readlane_f64(best_err, l), ov = readlane_f64(best_val, l);
    const int oi = __builtin_amdgcn_readlane(best_idx, l);
    if (bi < 0 || oe < be || (oe == be && oi > bi)) {
      be = oe;
      bv = ov;
      bi = oi;
    }
  }
  return bi >= 0 ? bv : 0.0;
}

// FixStep3, first half: extend every section forward then backward through the candidate map
// (ExtendF0, harvest.py:408-429) — one wave per section.
__global__ __launch_bounds__(64) void hc_extend_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                       const double* __restrict__ rows, const double* __restrict__ pf0,
                                                       const int64_t* __restrict__ lst,
                                                       const uint32_t* __restrict__ keep,
                                                       HcSec* __restrict__ secs, const int32_t* __restrict__ nsec,
                                                       double* __restrict__ chan) {
  const int u = blockIdx.y;
  if ((int)blockIdx.x >= nsec[u]) return;
  const HvUtt m = meta[u];
  const HcUtt h = hc[u];
  const int64_t n = m.nf1;
  HcSec* sp = secs + h.sec_base + blockIdx.x;
  HcSec s = *sp;
  const double* s2 = rows + h.f_base + 2 * n;
  double* ch = chan + h.ch_base + s.ch_off;  // ch[j - w0]
  const int lane = threadIdx.x;
  for (int i = lane; i < s.wlen; i += 64) {
    const int j = s.w0 + i;
    ch[i] = (j >= s.st && j <= s.ed) ? s2[j] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // The walk is a chain (the next step's reference value is this step's pick), but which rows it will read is known:
  // the candidate rows of the next kAhead frames are fetched together, so a step waits for global memory once in
  // kAhead instead of every time.
  constexpr int kAhead = 4;
  auto load_rows = [&](int64_t first, int dir, int count, double (&c0)[kAhead], double (&c1)[kAhead]) {
    int64_t ent[kAhead];  // the list heads of the kAhead frames first (one round trip), then lists and masks side by side
#pragma unroll
    for (int q = 0; q < kAhead; ++q) {
      const int64_t fr = first + (int64_t)dir * q;
      const bool ok = q < count && fr >= 0 && fr < n;
      ent[q] = ok ? lst[m.f1_off + fr] : 0;
    }
#pragma unroll
    for (int q = 0; q < kAhead; ++q) {
      const int64_t fr = first + (int64_t)dir * q;
      const bool ok = q < count && fr >= 0 && fr < n;
      const double* col = pf0 + (ent[q] >> 8);
      const int cnt = (int)(ent[q] & 255);
      const double v0 = lane < cnt ? col[lane] : 0.0;  // entry k of the list sits on lane k & 63
      const double v1 = lane + 64 < cnt ? col[lane + 64] : 0.0;
      const uint32_t w0 = ok ? keep[(m.f1_off + fr) * 4 + (lane >> 5)] : 0u;
      const uint32_t w1 = ok ? keep[(m.f1_off + fr) * 4 + 2 + (lane >> 5)] : 0u;
      c0[q] = ((w0 >> (lane & 31)) & 1u) ? v0 : 0.0;
      c1[q] = ((w1 >> (lane & 31)) & 1u) ? v1 : 0.0;
    }
  };
  // forward
  int r1 = s.ed;
  {
    double cur = ch[s.ed - s.w0];
    int miss = 0;
    int last = (int)(n - 2 < (int64_t)s.ed + 100 ? n - 2 : (int64_t)s.ed + 100) + 1;
    bool stop = false;
    for (int i0 = s.ed; i0 < last && !stop; i0 += kAhead) {
      double c0[kAhead], c1[kAhead];
      load_rows((int64_t)i0 + 1, +1, last - i0, c0, c1);
#pragma unroll
      for (int q = 0; q < kAhead; ++q) {
        const int i = i0 + q;
        if (i >= last || stop) break;
        const double b = select_best_regs(cur, c0[q], c1[q], 0.18);
        if (lane == 0) ch[i + 1 - s.w0] = b;
        if (b != 0.0) {
          cur = b;
          miss = 0;
          r1 = i + 1;
        } else if (++miss == 4) stop = true;
      }
    }
  }
  // backward
  int r0 = s.st;
  {
    double cur = ch[s.st - s.w0];
    int miss = 0;
    int last = (s.st - 100 > 1 ? s.st - 100 : 1) - 1;
    bool stop = false;
    for (int i0 = s.st; i0 > last && !stop; i0 -= kAhead) {
      double c0[kAhead], c1[kAhead];
      load_rows((int64_t)i0 - 1, -1, i0 - last, c0, c1);
#pragma unroll
      for (int q = 0; q < kAhead; ++q) {
        const int i = i0 - q;
        if (i <= last || stop) break;
        const double b = select_best_regs(cur, c0[q], c1[q], 0.18);
        if (lane == 0) ch[i - 1 - s.w0] = b;
        if (b != 0.0) {
          cur = b;
          miss = 0;
          r0 = i - 1;
        } else if (++miss == 4) stop = true;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double sum = 0.0;
  for (int j = r0 + lane; j <= r1; j += 64) sum += ch[j - s.w0];
  sum = wh::wave_sum(sum);
  const double mean = sum / (double)(r1 - r0 + 1);
  if (lane == 0) {
    s.r0 = r0;
    s.r1 = r1;
    s.kept = (2200.0 / mean < (double)(r1 - r0)) ? 1 : 0;
    *sp = s;
  }
}

__device__ __forceinline__ double chan_at(const double* __restrict__ chan_u, const HcSec& s, int64_t j) {
  return (j >= s.w0 && j < (int64_t)s.w0 + s.wlen) ? chan_u[s.ch_off + (j - s.w0)] : 0.0;
}

// SerachScore (harvest.py:490-495)
__device__ __forceinline__ double search_score(double f0, const HcList& l, const uint32_t* __restrict__ kw) {
  const double* __restrict__ cf = l.f0;
  const double* __restrict__ cs = l.sc;
  double sc = 0.0;
  if (f0 == 0.0) return sc;  // only surviving (non-zero) candidates can match from here on
  // the candidates are fetched eight at a time (one wait per block instead of one per entry); a score is only read for
  // the entry or two that match
  for (int e0 = 0; e0 < l.n; e0 += 8) {
    double c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = cf[e0 + q < l.n ? e0 + q : l.n - 1];  // clamped: see hc_base_kernel (surplus slots are tested out below)
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (e0 + q < l.n && f0 == c[q] && ((kw[(e0 + q) >> 5] >> ((e0 + q) & 31)) & 1u)) {
        const double v = cs[e0 + q];
        if (sc < v) sc = v;
      }
  }
  return sc;
}

// FixStep3 second half (MergeF0, harvest.py:442-486), FixStep4 (harvest.py:388-404), vuv.
__global__ __launch_bounds__(256) void hc_merge_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                       double* __restrict__ rows, const double* __restrict__ pf0,
                                                       const double* __restrict__ psc, const int64_t* __restrict__ lst,
                                                       const uint32_t* __restrict__ keep,
                                                       const HcSec* __restrict__ secs,
                                                       const int32_t* __restrict__ nsec, const double* __restrict__ chan,
                                                       int32_t* __restrict__ runs) {
  __shared__ int sh[16];
  __shared__ double red[16];
  __shared__ int order_n;
  const HvUtt m = meta[blockIdx.x];
  const HcUtt h = hc[blockIdx.x];
  const int64_t n = m.nf1;
  const double* s2 = rows + h.f_base + 2 * n;
  double* s3 = rows + h.f_base + 3 * n;
  double* s4 = rows + h.f_base + 4 * n;
  const HcSec* sc = secs + h.sec_base;
  const double* chan_u = chan + h.ch_base;
  int32_t* order = runs + h.run_base;  // reuse the run list area for the kept-section order
  const int ns = nsec[blockIdx.x];
  if (threadIdx.x == 0) {
    int k = 0;
    for (int i = 0; i < ns; ++i)
      if (sc[i].kept) {  // stable insertion by extended start (np.argsort of the range starts)
        int p = k++;
        while (p > 0 && sc[order[p - 1]].r0 > sc[i].r0) {
          order[p] = order[p - 1];
          --p;
        }
        order[p] = i;
      }
    order_n = k;
  }
  __syncthreads();
  const int nk = order_n;
  if (nk == 0) {
    for (int64_t j = threadIdx.x; j < n; j += 256) s3[j] = s2[j];
  } else {
    const HcSec first = sc[order[0]];
    for (int64_t j = threadIdx.x; j < n; j += 256) s3[j] = chan_at(chan_u, first, j);
    int R0 = first.r0, R1 = first.r1;
    for (int q = 1; q < nk; ++q) {
      __threadfence_block();
      __syncthreads();
      const HcSec s = sc[order[q]];
      if (s.r0 - R1 > 0) {
        for (int64_t j = s.r0 + threadIdx.x; j <= s.r1; j += 256) s3[j] = chan_at(chan_u, s, j);
        R0 = s.r0;
        R1 = s.r1;
      } else if (R0 <= s.r0 && R1 >= s.r1) {
        // completely covered: nothing changes
      } else {
        double a = 0.0, b = 0.0;
        for (int64_t j = s.r0 + threadIdx.x; j <= R1; j += 256) {
          const HcList l = hc_list(lst, pf0, psc, m.f1_off + j);
          const uint32_t* kw = keep + (m.f1_off + j) * 4;
          a += search_score(s3[j], l, kw);
          b += search_score(chan_at(chan_u, s, j), l, kw);
        }
        wh::block_sum2(a, b, red);
        const int64_t from = (a > b) ? R1 : s.r0;
        for (int64_t j = from + threadIdx.x; j <= s.r1; j += 256) s3[j] = chan_at(chan_u, s, j);
        R1 = s.r1;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- step 4: bridge unvoiced gaps shorter than 9 frames ------------------------------------------------
  for (int64_t j = threadIdx.x; j < n; j += 256) s4[j] = s3[j];
  int32_t* starts = runs + h.run_base;
  int32_t* ends = starts + h.rcap;
  const int nr = find_runs(s3, n, true, starts, ends, (int)h.rcap, sh);
  for (int k = 1 + threadIdx.x; k < nr; k += 256) {
    const int e0 = ends[k - 1], s1 = starts[k];
    const int dist = s1 - e0 - 1;
    if (dist >= 9) continue;
    const double t0 = s3[e0] + 1;
    const double t1 = s3[s1] - 1;
    const double c = (t1 - t0) / (double)(dist + 1);
    int cnt = 1;
    for (int j = e0 + 1; j < s1; ++j, ++cnt) s4[j] = t0 + c * (double)cnt;
  }
}

// SmoothF0 (harvest.py:533-559): per voiced run, edge-held signal through a 2nd-order Butterworth
// forward and backward.  The reference filters the whole zero-padded contour per run; the state of a
// stable IIR forgets its start within the 300-sample hold (|pole|^300 < 1e-17), so each run is filtered
// from 300 samples before to 300 samples after it.
__global__ __launch_bounds__(256) void hc_smooth_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                        double* __restrict__ rows, int32_t* __restrict__ runs,
                                                        double* __restrict__ chan) {
  __shared__ int sh[16];
  __shared__ long long offs[2];
  const HvUtt m = meta[blockIdx.x];
  const HcUtt h = hc[blockIdx.x];
  const int64_t n = m.nf1;
  const double* s4 = rows + h.f_base + 4 * n;
  double* sm = rows + h.f_base + 5 * n;
  int32_t* starts = runs + h.run_base;
  int32_t* ends = starts + h.rcap;
  for (int64_t j = threadIdx.x; j < n; j += 256) sm[j] = s4[j];
  const int nr = find_runs(s4, n, false, starts, ends, (int)h.rcap, sh);  // padded contour: ends are not forced
  const double b0 = 0.0078202080334971724, b1 = 0.015640416066994345, b2 = 0.0078202080334971724;
  const double a1 = -1.7347257688092754, a2 = 0.76600660094326412;
  double* scratch = chan + h.ch_base;
  // runs are processed 256 at a time; each thread owns a scratch slice of (len + 300) doubles
  for (int k0 = 0; k0 < nr; k0 += 256) {
    __syncthreads();
    if (threadIdx.x == 0) {
      offs[0] = 0;
    }
    __syncthreads();
    const int k = k0 + threadIdx.x;
    long long my_off = 0;
    int st = 0, ed = -1;
    if (k < nr) {
      st = starts[k];
      ed = ends[k];
    }
    // exclusive prefix of slice lengths, serial per block tile (few runs in practice)
    __shared__ long long lens[256];
    lens[threadIdx.x] = k < nr ? (long long)(ed - st + 1 + 300) : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      long long run = 0;
      for (int i = 0; i < 256; ++i) {
        const long long l = lens[i];
        lens[i] = run;
        run += l;
      }
    }
    __syncthreads();
    my_off = lens[threadIdx.x];
    if (k < nr && my_off + (ed - st + 1 + 300) <= h.ch_cap) {
      double* fw = scratch + my_off;  // forward outputs for frames st .. ed+300
      const double c0 = s4[st], c1 = s4[ed];
      double z0 = 0.0, z1 = 0.0, yv = 0.0;
#define BW_STEP(X)                \
  {                               \
    const double xin = (X);       \
    yv = z0 + b0 * xin;           \
    z0 = z1 + xin * b1 - yv * a1; \
    z1 = xin * b2 - yv * a2;      \
  }
      // the recurrences are chains of ~40 cycles per sample; their operands are fetched 16 at a time, a block ahead
      // (wh::serial_run), instead of one dependent global load per step
      auto always = [](int64_t) { return true; };
      for (int i = 0; i < 300; ++i) BW_STEP(c0);
      wh::serial_run<16>(
          st, (int64_t)ed + 1, always, [&](int64_t j) { return s4[j]; }, [&](int64_t j) { return s4[j]; },
          [&](int64_t j, double v) {
            BW_STEP(v);
            fw[j - st] = yv;
          });
      for (int i = 0; i < 300; ++i) {
        BW_STEP(c1);
        fw[ed - st + 1 + i] = yv;
      }
      z0 = 0.0;
      z1 = 0.0;
      const int64_t top = (int64_t)ed - st + 300;  // last forward output; the backward pass walks fw[top - i]
      wh::serial_run<16>(
          0, top + 1, always, [&](int64_t i) { return fw[top - i]; }, [&](int64_t i) { return fw[top - i]; },
          [&](int64_t i, double v) {
            BW_STEP(v);
            if (i >= 300) sm[ed - (i - 300)] = yv;
          });
#undef BW_STEP
    }
  }
}

__global__ __launch_bounds__(256) void hc_pick_kernel(const HvUtt* __restrict__ meta, const HcUtt* __restrict__ hc,
                                                      const double* __restrict__ rows, const double* __restrict__ tp,
                                                      double* __restrict__ f0_out, double* __restrict__ vuv_out) {
  const HvUtt m = meta[blockIdx.y];
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= m.nf) return;
  const int64_t n = m.nf1;
  const double* s4 = rows + hc[blockIdx.y].f_base + 4 * n;
  const double* sm = rows + hc[blockIdx.y].f_base + 5 * n;
  const double v = tp[m.f_off + f] * 1000;
  double r = v > 0 ? v + 0.5 : v - 0.5;  // round_matlab, truncated by the int cast (harvest.py:48-49)
  r = fmin((double)(n - 1), r);
  const int64_t idx = (int64_t)r;
  f0_out[m.f_off + f] = sm[idx];
  vuv_out[m.f_off + f] = s4[idx] != 0.0 ? 1.0 : 0.0;
}

inline int harvest_contour(wh_ctx* ctx, hipStream_t st, int B, const HvUtt* d_meta, const std::vector<HvUtt>& meta,
                           int64_t f1_tot, int64_t max_nf1, int64_t max_nf, const double* d_pf0, const double* d_psc,
                           const int64_t* d_lst, const uint32_t* d_keep,
                           char* d_ws, const double* tp, double* f0_out, double* vuv_out, double* dbg_f0_1ms) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  std::vector<HcUtt> hc(B);
  int64_t run_tot = 0, sec_tot = 0, ch_tot = 0, max_secs = 0;
  for (int u = 0; u < B; ++u) {
    HcUtt& h = hc[u];
    h.f_base = meta[u].f1_off * 6;
    h.rcap = meta[u].nf1 / 2 + 8;
    h.run_base = run_tot;
    run_tot += 2 * h.rcap;
    h.sec_base = sec_tot;
    sec_tot += h.rcap;
    h.ch_base = ch_tot;
    h.ch_cap = meta[u].nf1 * 32 + 1024;
    ch_tot += h.ch_cap;
    max_secs = std::max(max_secs, meta[u].nf1 / 7 + 2);
  }
  size_t off = 0;
  const size_t o_rows = off; off += al(sizeof(double) * f1_tot * 6);
  const size_t o_runs = off; off += al(sizeof(int32_t) * run_tot);
  const size_t o_secs = off; off += al(sizeof(HcSec) * sec_tot);
  const size_t o_ns = off; off += al(sizeof(int32_t) * B);
  const size_t o_ch = off; off += al(sizeof(double) * ch_tot);
  (void)off;
  HcUtt* d_hc = nullptr;
  double* d_rows = reinterpret_cast<double*>(d_ws + o_rows);
  int32_t* d_runs = reinterpret_cast<int32_t*>(d_ws + o_runs);
  HcSec* d_secs = reinterpret_cast<HcSec*>(d_ws + o_secs);
  int32_t* d_ns = reinterpret_cast<int32_t*>(d_ws + o_ns);
  double* d_ch = reinterpret_cast<double*>(d_ws + o_ch);
  if (int rc = wh::persistent_upload(ctx, st, "hv.contour", hc, &d_hc)) return rc;
  const dim3 gf((unsigned)((max_nf1 + 255) / 256), B);
  { wh::KernelTimer _kt(ctx, st, "hc_base_kernel"); hipLaunchKernelGGL(hc_base_kernel, gf, dim3(256), 0, st, d_meta, d_hc, d_pf0, d_psc, d_lst, d_keep, d_rows); }
  WH_LAUNCH_CHECK("hc_base_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_step1_kernel"); hipLaunchKernelGGL(hc_step1_kernel, gf, dim3(256), 0, st, d_meta, d_hc, d_rows); }
  WH_LAUNCH_CHECK("hc_step1_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_sections_kernel"); hipLaunchKernelGGL(hc_sections_kernel, dim3(B), dim3(256), 0, st, d_meta, d_hc, d_rows, d_runs, d_secs, d_ns); }
  WH_LAUNCH_CHECK("hc_sections_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_extend_kernel"); hipLaunchKernelGGL(hc_extend_kernel, dim3((unsigned)max_secs, B), dim3(64), 0, st, d_meta, d_hc, d_rows, d_pf0, d_lst, d_keep, d_secs, d_ns, d_ch); }
  WH_LAUNCH_CHECK("hc_extend_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_merge_kernel"); hipLaunchKernelGGL(hc_merge_kernel, dim3(B), dim3(256), 0, st, d_meta, d_hc, d_rows, d_pf0, d_psc, d_lst, d_keep, d_secs, d_ns, d_ch, d_runs); }
  WH_LAUNCH_CHECK("hc_merge_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_smooth_kernel"); hipLaunchKernelGGL(hc_smooth_kernel, dim3(B), dim3(256), 0, st, d_meta, d_hc, d_rows, d_runs, d_ch); }
  WH_LAUNCH_CHECK("hc_smooth_kernel");
  { wh::KernelTimer _kt(ctx, st, "hc_pick_kernel"); hipLaunchKernelGGL(hc_pick_kernel, dim3((unsigned)((max_nf + 255) / 256), B), dim3(256), 0, st, d_meta, d_hc, d_rows, tp, f0_out, vuv_out); }
  WH_LAUNCH_CHECK("hc_pick_kernel");
  if (dbg_f0_1ms) {
    for (int u = 0; u < B; ++u)
      WH_CHECK(hipMemcpyAsync(dbg_f0_1ms + meta[u].f1_off, d_rows + hc[u].f_base + 4 * meta[u].nf1,
                              sizeof(double) * meta[u].nf1, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}
