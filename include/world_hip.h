/* world_hip.h — C ABI of libworld_hip.so: the WORLD vocoder analysis/synthesis hot path as
 * hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * The reference (tuanad121/Python-WORLD) has no FFI: its seam is the set of module-level stage
 * functions imported by world/main.py:14-23.  Each entry point below is the batched, device-side
 * replacement of ONE of those functions (cited per function); the Python mirror in
 * python-world_amd/world/ binds them with ctypes and keeps the reference's names, arguments and
 * result-dict keys.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; wh_last_error() gives the text
 *    of the last failure on the calling thread.  No C++ exception crosses the boundary.
 *  - pointers named h_* are HOST pointers; all other data pointers are DEVICE pointers
 *    (hipMalloc / torch.cuda memory).  `stream` is a hipStream_t passed as void* (NULL = default).
 *    Calls are asynchronous with respect to the host unless stated.
 *  - dtype is IEEE float64 everywhere (the reference is float64 end to end).
 *  - a *batch* is a set of utterances concatenated sample-wise (x, offsets h_x_off[n_utt+1]) and
 *    frame-wise (per-frame arrays, offsets h_frame_off[n_utt+1]).  Utterances are independent.
 *  - dense per-frame outputs are FRAME-MAJOR on the device: spectrogram[frame][bin] (the
 *    reference's NumPy arrays are (bins, frames); the Python mirror transposes at the boundary).
 *  - nothing is retained past return except inside wh_ctx / wh_batch objects.
 */
#ifndef WORLD_HIP_H
#define WORLD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wh_ctx wh_ctx;     /* one per device per host thread: twiddle tables + scratch */
typedef struct wh_batch wh_batch; /* utterance/frame offsets of one batch, resident on the device */

int wh_version(void);
const char* wh_last_error(void);

/* ---- device / memory helpers (so a host without torch can drive the library) ------------- */
int wh_device_count(int* count);
int wh_ctx_create(int device, wh_ctx** out);
int wh_ctx_destroy(wh_ctx* ctx);
/* Free the context's scratch (workspace arena, per-call buffers: they only grow with the largest batch served); tables and
 * flags stay, the next call allocates again.  Synchronises the device.  A time base held by the context is dropped, and a
 * hipGraph captured over calls of this context holds the freed addresses: capture again instead of replaying it. */
int wh_ctx_trim(wh_ctx* ctx);
int wh_malloc(void** dptr, size_t bytes);
int wh_free(void* dptr);
int wh_memcpy_h2d(void* dst, const void* h_src, size_t bytes, void* stream);
int wh_memcpy_d2h(void* h_dst, const void* src, size_t bytes, void* stream);
int wh_memset(void* dst, int value, size_t bytes, void* stream);
int wh_stream_sync(void* stream);
/* Pinned, device-mapped host memory (hipHostMalloc): the GPU can read and write it through the same address. */
int wh_host_alloc(void** h_ptr, size_t bytes);
int wh_host_free(void* h_ptr);
/* Copy `bytes` (a multiple of 8) between two device-ACCESSIBLE pointers with a kernel on `stream`: either side may be
 * device memory or pinned host memory from wh_host_alloc / torch's pin_memory().  Unlike wh_memcpy_* this never
 * touches a DMA engine queue: an upload issued this way cannot be serialised behind a long download that another
 * stream has queued on the same engine (bench.py with_transfers_pipelined).  `max_blocks` <= 0 picks 64 workgroups
 * (enough to saturate PCIe, a quarter of the CUs at most). */
int wh_copy_mapped(wh_ctx* ctx, void* stream, void* dst, const void* src, size_t bytes, int max_blocks);

/* Sticky device-side condition flags raised by kernels instead of failing silently; reading them
 * synchronises the stream and clears them.  h_flags16[WH_FLAG_*] != 0 means the condition occurred. */
#define WH_FLAG_STONEMASK_WINDOW 0 /* a frame's f0 needed a longer window than kmax: left unrefined */
#define WH_FLAG_EVENT_OVERFLOW 1   /* a zero-crossing list exceeded its capacity (DIO: cannot happen, cap = len/2+2; Harvest: see wh_harvest_set_event_caps) */
#define WH_FLAG_NOISE_SHORT 2      /* synthesis ran out of host-supplied noise samples */
#define WH_FLAG_NO_PULSE 3         /* an utterance produced no pulse (reference asserts, synthesis.py:131) */
#define WH_FLAG_PULSE_OVERFLOW 4   /* more pulses than the pulse capacity, or more overlap-add rows than its row region holds */
#define WH_FLAG_OOB 5              /* bounds build only (wh_bounds_build() == 1): a kernel indexed outside a named buffer */
int wh_take_flags(wh_ctx* ctx, void* stream, int32_t* h_flags16);
/* The bounds build of the library (tools/build_variants.py ...:-DWH_BOUNDS=1; a test vehicle, never the shipped file):
 * the covered kernels index their buffers through checked pointers (csrc/wh_device.h, wh::ckp); an access outside a
 * buffer is redirected to its first element, counted, and the first one recorded.  wh_take_flags then reports
 * WH_FLAG_OOB, and wh_bounds_last returns that record: out4 = {accesses out of range since the previous take, buffer
 * tag (WH_CK_* of wh_device.h), element index, elements in the buffer}.  In a normal build wh_bounds_build() is 0, the
 * flag is never set and the record is all zeros. */
int wh_bounds_build(void);
int wh_bounds_last(int64_t* out4);
/* Positive control (bounds build; fails elsewhere): a kernel that stores one element past a four-element checked
 * buffer — the next wh_take_flags must report WH_FLAG_OOB with the record {1, WH_CK_TABLE (7), 4, 4}. */
int wh_bounds_selftest(wh_ctx* ctx, void* stream);
/* Test hook: the spectral kernels' own log / exp / sincospi (csrc/wh_math.h: a few ulp, a third of the device library's
 * instructions) applied to a DEVICE array: which = 0 log -> out[n], 1 exp -> out[n], 2 (sin, cos)(pi x) -> out[2n]. */
int wh_math_probe(wh_ctx* ctx, void* stream, int which, const double* in, double* out, int64_t n);
/* The same flags without a host wait, for callers that keep a pipeline of batches in flight:
 *   wh_flags_post — enqueue (one 16-lane kernel on `stream`) the publication of the flags raised by everything before
 *     it on the stream to a pinned host word per flag, and clear them on the device (discard != 0: clear them
 *     without publishing — for work whose results have been abandoned);
 *   wh_flags_poll — read, WITHOUT synchronising, the conditions published since the last poll / take (h_flags16[i] != 0).
 * A condition is therefore reported at the first poll after its post has executed — late, never lost; wh_take_flags
 * (which synchronises) also reports whatever has been posted and not yet polled. */
int wh_flags_post(wh_ctx* ctx, void* stream, int discard);
int wh_flags_poll(wh_ctx* ctx, int32_t* h_flags16);

/* Per-kernel timing: while enabled every kernel launch of this ctx is bracketed by a HIP event pair on
 * its launch stream.  wh_profile_collect synchronises the device and returns one record per launch in
 * launch order: names = '\n'-joined kernel names, ms[i] = elapsed milliseconds. Clears the record list. */
int wh_profile_enable(wh_ctx* ctx, int on);
int wh_profile_collect(wh_ctx* ctx, char* names, size_t names_bytes, float* ms, int max_records, int* n_records);

/* ---- batch descriptor --------------------------------------------------------------------- */
/* h_x_off[n_utt+1]: sample offsets into the concatenated waveform; h_frame_off[n_utt+1]: frame
 * offsets into the concatenated per-frame arrays.  Synchronous: the descriptor is complete on return (small H2D
 * copies and one kernel on the NULL stream, which is waited for — work in flight on non-blocking streams is not). */
int wh_batch_create(wh_ctx* ctx, int n_utt, const int64_t* h_x_off, const int64_t* h_frame_off, wh_batch** out);
int wh_batch_destroy(wh_batch* b);
/* Frame count of one utterance: int(1000*n/fs/frame_period + 1) — world/dio.py:28, world/harvest.py:46. */
int64_t wh_num_frames(int64_t n_samples, double fs, double frame_period_ms);

/* ---- CheapTrick: replaces cheaptrick()  (world/cheaptrick.py:9-39) ------------------------ */
/* x[total_samples]; tp/vuv[total_frames]; f0[total_frames] is IN/OUT: unvoiced frames and frames
 * below 3*fs/(fft_size-3) are overwritten with 500 Hz exactly as the reference mutates
 * source_object['f0'] (cheaptrick.py:26-27,32-33).  spectrogram[total_frames][fft_size/2+1];
 * ps_spectrogram (optional, may be NULL) [total_frames][fft_size] interleaved (re,im) =
 * the reference's 'ps spectrogram'.  fft_size must be a power of two in [256, 4096]. */
int wh_cheaptrick(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double* f0,
                  const double* vuv, double fs, int fft_size, double q1, double* spectrogram, double* ps_spectrogram);

/* ---- D4C: replaces d4c()  (world/d4c.py:10-64) ------------------------------------------------ */
/* f0 is IN/OUT: frames with vuv==0 are zeroed like the reference does (d4c.py:32).  The D4C FFT
 * size 2^ceil(log2(4fs/47+1)), the love-train FFT size and the band layout follow the reference.
 * aperiodicity[total_frames][fft_size_for_spectrum/2+1] (amplitude, 1-1e-12 on unvoiced frames);
 * coarse_ap (optional, may be NULL) [total_frames][wh_d4c_bands(fs,0)] = the reference's
 * 'coarse_ap' debug rows (negative dB). */
int wh_d4c(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double* f0,
           const double* vuv, double fs, double threshold, int fft_size_for_spectrum, double* aperiodicity,
           double* coarse_ap);
/* Number of aperiodicity bands: floor(min(15000, fs/2-interval)/interval) (d4c.py:34, d4cRequiem.py:19). */
int wh_d4c_bands(double fs, int requiem);

/* ---- D4C-Requiem: replaces d4cRequiem()  (world/d4cRequiem.py:9-44) ------------------------- */
/* fft_size <= 0 selects the reference default 2^ceil(log2(3fs/47+1)).
 * band_aperiodicity[total_frames][wh_d4c_bands(fs,1)+2] in dB (row 0 = -60, last = -1e-12). */
int wh_d4c_requiem(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double* f0,
                   const double* vuv, double fs, double threshold, int fft_size, double* band_aperiodicity);

/* ---- DIO: replaces dio()  (world/dio.py:10-55) -------------------------------------------------- */
/* tp[total_frames]: frame times (s) = arange(nf)*frame_period/1000, nf = wh_num_frames(n, fs, frame_period).
 * The band filters are DATA supplied by the host (so that the even-length Nuttall argmax tie that fixes
 * each band's delay, dio.py:130-131, is decided by the host's NumPy exactly like the reference):
 *   h_band_f0[n_bands]   boundary f0 of each band: f0_floor*2^((i+1)/channels_in_octave)   (dio.py:32-34)
 *   h_band_len[n_bands]  tap count 4*int(target_fs/f/2+0.5);  h_band_taps: the Nuttall windows, concatenated
 *   h_band_bias[n_bands] argmax of each window (the reference's index_bias)
 *   h_lowcut[2*lowcut_half+1]  the zero-phase low-cut FIR of dio.py:80-83, lowcut_half = int(target_fs/50+0.5)
 * Outputs: f0_out, vuv_out [total_frames]; optional cand_out / raw_out: for utterance u a row-major
 * [n_bands][nf_u] block at offset frame_off[u]*n_bands (the reference's 'f0_candidates' / 'raw_f0_candidates').
 * Decimation ratio is int(fs/target_fs) and the decimated rate is taken as target_fs, like the reference. */
int wh_dio(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double fs, double f0_floor,
           double f0_ceil, double target_fs, double frame_period_ms, double allowed_range, int n_bands,
           const double* h_band_f0, const int32_t* h_band_bias, const int32_t* h_band_len, const double* h_band_taps,
           const double* h_lowcut, int lowcut_half, double* f0_out, double* vuv_out, double* cand_out, double* raw_out);

/* ---- Harvest: replaces harvest()  (world/harvest.py:17-54) ---------------------------------------- */
/* tp[total_frames]: output frame times (s).  Host-supplied filter DATA (the reference obtains them from
 * SciPy / NumPy at run time, so the host language evaluates the same expressions):
 *   decimation_ratio r = int(fs/8000 + 0.5) (1 for fs <= 8000); if fs > 8000 — ALSO where r rounds to 1, 8 kHz < fs <
 *   12 kHz: harvest.py:60 branches on the rate and still low-pass filters — h_ba[8] = (b0..b3, a0..a3) of
 *   scipy.signal.cheby1(3, 0.05, 0.8/r) and h_zi[3] = scipy.signal.lfilter_zi(b, a)   (harvest.py:599-603); else
 *   h_ba = NULL or all zeros (a0 == 0: no filter);
 *   h_band_f0[n_bands]: channel centre frequencies (harvest.py:22-29, 152 for the default range);
 *   h_band_half[n_bands]: h = round-half-up(2*fs_d/f) (harvest.py:253);
 *   h_band_taps: concatenated band-pass FIRs nuttall(2h+1)*cos(2*pi*f*k/fs_d), k=-h..h (harvest.py:254-256).
 * Outputs f0_out / vuv_out [total_frames].  Optional debug outputs (DEVICE, may be NULL): dbg_y — the
 * decimated, mean-removed signals concatenated; dbg_raw — per utterance [n_bands][nf1] raw channel candidates
 * on the 1 ms grid; dbg_f0_1ms — the 1 ms contour before smoothing. */
int wh_harvest(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, double fs,
               double f0_floor, double f0_ceil, double frame_period_ms, int decimation_ratio, const double* h_ba,
               const double* h_zi, int n_bands, const double* h_band_f0, const int32_t* h_band_half,
               const double* h_band_taps, double* f0_out, double* vuv_out, double* dbg_y, double* dbg_raw,
               double* dbg_f0_1ms);
/* Capacities of Harvest's zero-crossing lists.  The reference keeps the four crossing trains of a channel as NumPy arrays
 * of whatever length they turn out to have (ZeroCrossingEngine, world/harvest.py:283-297); wh_harvest sizes its lists
 * from an estimate (three times the channel's centre frequency per second) before anything has run.  The estimate fails
 * where a stretch of the filtered signal is constant up to rounding — digital silence next to signal: harvest.py:69
 * removes the mean, the silence becomes a DC level, and the first difference of its filtered image changes sign at
 * random, as it does in the reference's own arithmetic.  Such a call raises WH_FLAG_EVENT_OVERFLOW and its results
 * are not to be used; its COUNTS are exact (counting goes on past a full list):
 *   wh_harvest_event_counts — h_caps_out[u * n_bands + i] (HOST) = the longest of the four trains of channel i of
 *     utterance u in this context's last wh_harvest; n = utterances x channels of that call.  Waits for `stream`.
 *   wh_harvest_set_event_caps — h_caps != NULL: these capacities for the NEXT wh_harvest of this context (one call);
 *     NULL, n == -1: the bound no signal exceeds (ylen/2 + 2 per train: 4.3 x the estimate's memory for the default
 *     range) for every later call; NULL, n == 0: back to the estimate.
 * A repeat with the counted capacities fits by construction (same arithmetic, same counts). */
int wh_harvest_set_event_caps(wh_ctx* ctx, const int64_t* h_caps, int64_t n);
int wh_harvest_event_counts(wh_ctx* ctx, void* stream, int64_t* h_caps_out, int64_t n);

/* ---- StoneMask: replaces stonemask()  (world/stonemask.py:8-27) -------------------------------- */
/* f0[total_frames] in, refined_f0[total_frames] out (a different buffer: the reference returns a new
 * array).  h_qtime[2*kmax+1] (HOST): h_qtime[k+kmax] = float("%.4f" % (k/fs)) — the reference
 * quantises its window time base through string formatting (stonemask.py:38); the table is built by
 * the host language so that the decimal rounding is exactly Python's.  kmax >= ceil(1.5*fs/min f0). */
int wh_stonemask(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, const double* tp, const double* f0,
                 double fs, const double* h_qtime, int kmax, double* refined_f0);

/* ---- Synthesis: replaces synthesis()  (world/synthesis.py:21-82) ------------------------------- */
/* Inputs per frame (batch frame layout): tp, f0, vuv [total_frames]; spectrogram and aperiodicity
 * [total_frames][fft_size/2+1] frame-major (aperiodicity = amplitude as produced by wh_d4c).
 * Output y: concatenated waveforms, offsets h_y_off[n_utt+1] (HOST).  The length of utterance u must be
 * len(np.arange(tp[0], tp[-1]+1/fs, 1/fs)) and its time axis is t_i = h_t0[u] + i*h_dt[u] with
 * h_dt = (t0 + 1/fs) - t0: NumPy's float arange semantics are evaluated by the host (synthesis.py:39; the
 * 48 kHz case yields 480002 samples, not 480001).
 * pulse_cap: pulse slots per utterance (>= number of pulses; ny/8+64 suffices for mean f0 < fs/8);
 *   overflow raises WH_FLAG_PULSE_OVERFLOW.
 * noise: optional DEVICE array of standard-normal samples, utterance u reads noise[h_noise_off[u] ..
 *   h_noise_off[u+1]); pulse i consumes max(3, noise_size_i) consecutive samples in pulse order — exactly
 *   the reference's np.random.randn call sequence (synthesis.py:93), so feeding np.random.randn(total)
 *   reproduces the reference bit-for-bit in the noise it uses.  noise == NULL: a counter-based Philox
 *   generator seeded by `seed` is used on the device instead (statistically equivalent, not the same samples).
 * pulse_count_out: optional DEVICE int32[n_utt]. */
int wh_synthesis(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0, const double* vuv,
                 const double* spectrogram, const double* aperiodicity, double fs, int fft_size, const int64_t* h_y_off,
                 const double* h_t0, const double* h_dt, int64_t pulse_cap, const double* noise,
                 const int64_t* h_noise_off, uint64_t seed, double* y, int32_t* pulse_count_out);
/* wh_synthesis in two halves, so that the half that depends on the time base alone can run early, on another stream:
 *   wh_synthesis_timebase — phase increments, the exact cumulative phase (np.cumsum, bit for bit), pulse positions and
 *     fractional shifts, noise offsets, per-pulse frame pairs (synthesis.py:118-152).  Inputs tp / f0 / vuv only.  The
 *     results stay in ctx's workspace until another call of that context uses the workspace.  f0_low_limit > 0: f0 is
 *     the F0 stage's output and is read as World.encode leaves it after CheapTrick and D4C (vuv == 0 -> 0, f0 <
 *     f0_low_limit = 3 fs / (fft_size - 3) -> 500 Hz), i.e. the call can be issued before those two stages have run.
 *   wh_synthesis_render — the spectral half: one response per pulse of the time base held by timebase_ctx (the same
 *     context or another one of the same device), overlap-added into y.  The caller orders it behind the time base
 *     (same stream, or an event wait).  Fails if timebase_ctx holds no time base for this batch / pulse_cap.
 * wh_synthesis(ctx, ...) == wh_synthesis_timebase(ctx, ..., 0) followed by wh_synthesis_render(ctx, ..., ctx, ...). */
int wh_synthesis_timebase(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                          const double* vuv, double fs, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                          int64_t pulse_cap, double f0_low_limit);
int wh_synthesis_render(wh_ctx* ctx, void* stream, const wh_batch* b, const wh_ctx* timebase_ctx, const double* tp,
                        const double* spectrogram, const double* aperiodicity, double fs, int fft_size,
                        const int64_t* h_y_off, const double* h_t0, const double* h_dt, int64_t pulse_cap,
                        const double* noise, const int64_t* h_noise_off, uint64_t seed, double* y,
                        int32_t* pulse_count_out);
/* Pulse bookkeeping only (synchronous): per-utterance pulse count and the exact number of normal
 * samples the reference would draw, sum_i max(3, noise_size_i).  HOST outputs. */
int wh_synthesis_plan(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                      const double* vuv, double fs, const int64_t* h_y_off, const double* h_t0, const double* h_dt,
                      int64_t pulse_cap, int32_t* h_pulse_count, int64_t* h_noise_total);

/* The device noise of wh_synthesis / wh_synthesis_render (noise == NULL), exposed for sample-exact checks: out[i]
 * (DEVICE, n doubles) = sample q0 + i of the standard-normal stream that utterance `utt` of a batch reads under
 * `seed` — the stand-in for the reference's np.random.randn draws (world/synthesis.py:93): pulse i consumes samples
 * noff_i .. noff_i + max(3, noise_size_i) of it, exactly as it would consume a host-supplied `noise` stream, so a
 * decode with noise = this dump equals the decode with noise == NULL and the same seed. */
int wh_philox_normals(wh_ctx* ctx, void* stream, uint64_t seed, int utt, int64_t q0, int64_t n, double* out);

/* decode()'s peak normalisation (world/main.py:209-212): per utterance u, y[h_y_off[u] .. h_y_off[u+1]) is divided
 * by max|y| when that exceeds 1.  In place, on the stream; the workspace of ctx is used (call it after wh_synthesis*
 * has finished enqueueing, as the Python mirror does). */
int wh_peak_normalise(wh_ctx* ctx, void* stream, double* y, const int64_t* h_y_off, int n_utt);

/* The phase accumulator of synthesis() is np.cumsum over the per-sample phase increments (world/synthesis.py:128):
 * a sequential float64 sum whose rounding decides the pulse positions.  This entry exposes the routine that
 * reproduces it bit for bit (in place, n_seg independent segments of NON-NEGATIVE doubles, h_off[n_seg + 1] element
 * offsets into d_data) so that the agreement can be tested directly. */
int wh_cumsum_exact(wh_ctx* ctx, void* stream, double* d_data, const int64_t* h_off, int n_seg);

/* ---- Requiem synthesis: replaces synthesisRequiem()  (world/synthesisRequiem.py:12-25) ------------ */
/* band_aperiodicity[total_frames][n_bands] in dB as produced by wh_d4c_requiem (n_bands = bands+2 <= 8).
 * Seed tables (DEVICE, row-major like the reference's NumPy arrays): pulse_seed[pulse_fft][n_bands],
 * noise_seed[noise_len][n_bands] — built by the host (get_seeds_signals, world/get_seeds_signals.py:8).
 * h_hop[u] = int((tp[1]-tp[0])*fs) of utterance u (synthesisRequiem.py:78, truncating).
 * h_cursor[n_utt][n_bands]: read position in the circular noise seed at which utterance u starts — the
 * reference keeps this in a function attribute that persists across calls (synthesisRequiem.py:131-141);
 * after an utterance of ny samples the position is (cursor + ny - 1) mod noise_len.
 * y / h_y_off / h_t0 / h_dt / pulse_cap as for wh_synthesis. */
int wh_synthesis_requiem(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp, const double* f0,
                         const double* vuv, const double* spectrogram, const double* band_aperiodicity, double fs,
                         int fft_size, const int64_t* h_y_off, const double* h_t0, const double* h_dt, const int64_t* h_hop,
                         int64_t pulse_cap, const double* pulse_seed, int pulse_fft, const double* noise_seed,
                         int64_t noise_len, int n_bands, const int64_t* h_cursor, double* y);

/* ---- Spectral feature heads: replace World.encode_lfbank / encode_mcep / decode_mcep  (world/main.py:305-358) ---- */
/* One dense product per frame with an elementwise prologue and epilogue, on the FP64 matrix cores:
 *     out[f][n] = epi( sum_{k<ka} pro(a[f*lda + k], k) * h_w[k*nw + n] ),   f < n_rows, n < nw;  out row stride ldo
 *   prologue 0: pro(v) = v;  1: pro(v, k) = pscale * (v * h_p[k])^2  (encode_lfbank: pre-emphasis |H(k)| and
 *   1/nfft power, main.py:313-316);  2: pro(v) = log(v)  (encode_mcep, main.py:333)
 *   epilogue 0: none;  1: log with 0 -> DBL_EPSILON (main.py:321-322);  2: exp (decode_mcep, main.py:358);
 *   3: sqrt(max(0, v)) (SWIPE' loudness, swipe.py:42-44)
 * a / out: DEVICE, frame-major (wh_cheaptrick's spectrogram layout).  h_w[ka][nw] and h_p[ka] are HOST tables: the mel
 * filterbank transposed (get_filterbanks, main.py:275-303), or the cosine rows of the inverse / forward real FFT with
 * the mel warp of main.py:335-337 / 351-356 folded in — built by the host with the reference's own expressions. */
int wh_feature_matmul(wh_ctx* ctx, void* stream, const double* a, int64_t n_rows, int ka, int64_t lda, int prologue,
                      const double* h_p, double pscale, const double* h_w, int nw, int epilogue, double* out, int64_t ldo);
/* The same product with the weight table identified by the caller: `table_tag` != 0 promises that every call with this
 * tag (and the same ka x nw) passes the same h_w content.  The padded matrix is then built and uploaded on the first
 * call only and every later call is a look-up — no re-padding, no comparison of the host bytes (7 MB per SWIPE' call
 * at 16 kHz otherwise).  Tables of different content MUST carry different tags; table_tag == 0 is wh_feature_matmul. */
int wh_feature_matmul_tagged(wh_ctx* ctx, void* stream, const double* a, int64_t n_rows, int ka, int64_t lda, int prologue,
                             const double* h_p, double pscale, const double* h_w, int nw, int epilogue, double* out,
                             int64_t ldo, uint64_t table_tag);
/* get_context (main.py:360-365): out[i][j*d + c] = x[clamp(i + j - w, 0, n_rows-1)][c], j = 0..2w.  DEVICE pointers. */
int wh_context_frames(wh_ctx* ctx, void* stream, const double* x, int64_t n_rows, int d, int w, double* out);

/* ---- SWIPE': replaces swipe()  (world/swipe.py:9-105; f0_method='swipe', world/main.py:45-46,134-135) ---------- */
/* One entry of the window-size table (HOST): candidates [j0, j0+n_c) of the candidate set use window size ws with
 * hop `hop` (= ws - noverlap of the reference's specgram call, swipe.py:35-38). */
typedef struct wh_swipe_window {
  int32_t ws, hop;          /* window length (power of two in [64, 16384]), hop in samples */
  int32_t j0, n_c;
  const double* h_window;   /* [ws]               np.hanning(ws + 2)[1:-1] */
  const double* h_interp;   /* [ws/2+1][n_erb]    magnitude bins -> ERB grid: interp1d(kind='cubic') as a matrix, k-major */
  const double* h_kernels;  /* [n_erb][n_c]       candidate kernels (pitchStrengthOneCandidate, swipe.py:127-146), k-major */
  const double* h_mu;       /* [n_c]              window-size membership weights (swipe.py:62-66) */
  uint64_t table_tag;       /* != 0: identity of (h_interp, h_kernels) for wh_feature_matmul_tagged (content-stable per tag) */
} wh_swipe_window;
/* x: concatenated waveforms (DEVICE); the batch's frame grid is the output grid t = arange(nf) * dt with
 * nf = int(1000*n/fs/(dt*1000) + 1).  HOST tables: h_pc[n_cand] candidate pitches, h_win[n_win], and for the parabolic
 * refinement of a maximum at candidate j (swipe.py:83-100): h_ntc[j][3] = normalised periods of candidates j-1..j+1,
 * h_fine[j][fine_stride] / h_n_fine[j] = the 1/768-octave evaluation grid (fine_stride must be 20).
 * s_thr: pitch-strength threshold (frames below it are unvoiced, f0 = 0).  Outputs f0_out / vuv_out [total_frames]. */
int wh_swipe(wh_ctx* ctx, void* stream, const wh_batch* b, const double* x, double fs, double dt, double s_thr, int n_cand,
             const double* h_pc, int n_erb, int n_win, const wh_swipe_window* h_win, const double* h_ntc,
             const double* h_fine, const int32_t* h_n_fine, int fine_stride, double* f0_out, double* vuv_out);

/* ---- Requiem seed signals on the device: replaces get_seeds_signals()  (world/get_seeds_signals.py:8-73) --------- */
/* pulse_seed[fft_size][n_bands] (DEVICE): the band pulses, deterministic, equal to the reference's up to rounding.
 * noise_seed[noise_length][n_bands] (DEVICE): modified velvet noise (segments of the reference's three short periods
 * chosen at random, one +-2 impulse per 4-sample cell at a random offset, signs balanced per segment in random
 * order) circularly convolved with each band pulse.  The randomness comes from a Philox-4x32-10 stream keyed by
 * `seed` instead of Python's `random` / NumPy's global generator: same construction and statistics, not the same
 * samples.  velvet_out (optional, may be NULL): the velvet noise itself [noise_length].
 * n_bands = wh_d4c_bands(fs, 1) + 2; fft_size / noise_length: the reference defaults are
 * 1024 * 2^ceil(log2(fs/48000)) and 2^ceil(log2(fs/2)). */
int wh_requiem_seeds(wh_ctx* ctx, void* stream, double fs, int fft_size, int64_t noise_length, int n_bands, uint64_t seed,
                     double* pulse_seed, double* noise_seed, double* velvet_out);

/* ---- Modifiers on a resident encoding: replace World.warp_spectrum / modify_duration  (world/main.py:180-196) ---- */
/* warp_spectrum: spectrogram[n_frames][k_bins] (DEVICE, in place): frame <- np.interp((k/K)^factor, k/K, frame).  The
 * query points are the same for every frame, so the host runs NumPy's search once per bin and passes (HOST tables)
 * h_src[k] = interval index j, h_dx[k] = x_k - xp[j], h_den[k] = xp[j+1] - xp[j], with h_den[k] = 0 where np.interp
 * returns fp[j] itself (exact knot / last knot / beyond it); the kernel applies NumPy's
 * (fp[j+1]-fp[j]) / den * dx + fp[j]. */
int wh_warp_spectrum(wh_ctx* ctx, void* stream, double* spectrogram, int64_t n_frames, int k_bins, const int32_t* h_src,
                     const double* h_dx, const double* h_den);
/* modify_duration: tp_out[f] = np.interp(tp_in[f], xp_u, fp_u) for the utterance u of frame f; h_xp / h_fp
 * [n_utt][n_anchor] (HOST): xp_u = [0, from_time..., last frame time of u], fp_u = to_time with a trailing -1 replaced
 * by that last frame time (main.py:186-189).  tp_out must not alias tp_in (the reference installs a new array). */
int wh_modify_duration(wh_ctx* ctx, void* stream, const wh_batch* b, const double* tp_in, double* tp_out,
                       const double* h_xp, const double* h_fp, int n_anchor);

/* ---- 16-bit PCM at the batch boundary (the reference's WAV usage: example/prosody.py:12-13,57) ---------------------- */
/* x[i] = pcm[i] / (2^15 - 1); pcm[i] = int16(trunc(y[i] * 2^15)) (low 16 bits, like NumPy's astype on the reference's
 * platform).  DEVICE pointers: the 2-byte samples cross PCIe instead of the 8-byte ones. */
int wh_pcm16_to_f64(wh_ctx* ctx, void* stream, const int16_t* pcm, int64_t n, double* x);
int wh_f64_to_pcm16(wh_ctx* ctx, void* stream, const double* y, int64_t n, int16_t* pcm);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_HIP_H */
