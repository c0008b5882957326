/*
 * ddsp_hip.h -- C ABI of libddsp_hip.so: the MI355X (gfx950) DDSP synthesis hot path.
 *
 * The reference (kureta/ddsp-pytorch) has no FFI of its own: its hot path is two
 * Python nn.Modules made of stock torch ops.  These entry points are what a binding
 * for that path binds instead of those op sequences:
 *
 *   ddsp_osc_forward      replaces OscillatorBank.forward / .live
 *                         (model/ddsp/harmonic_oscillator.py:57-62 and :64-75, i.e.
 *                          prepare_harmonics :24-37, generate_phases :39-43, generate_signal :45-50)
 *   ddsp_noise_forward    replaces FilteredNoise.forward
 *                         (model/ddsp/filtered_noise.py:40-53, i.e. amp_to_impulse_response :7-22,
 *                          fft_convolve :25-32, and the torch.rand draw :44-48)
 *
 * Conventions: plain pointers and sizes only (no torch types); every pointer is DEVICE memory
 * unless said otherwise; tensors are dense row-major fp32 with the reference's shapes; nothing
 * is allocated or synchronised inside (the caller passes scratch; launches are asynchronous on
 * `stream`, a hipStream_t passed as void*; NULL = the default stream).  Return value: 0 on
 * success, a hipError_t (> 0) if a launch failed, or a negative DDSP_E* code for bad
 * arguments.  No exceptions cross the boundary.
 */
#ifndef DDSP_HIP_H
#define DDSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDSP_HIP_ABI_VERSION 4

#define DDSP_EINVAL (-1)   /* null pointer / non-positive size */
#define DDSP_ERANGE (-2)   /* shape outside what the kernels are built for */
#define DDSP_EPERM  (-3)   /* a test / tuning hook called in a process that did not opt in (see ddsp_test_hooks_enabled) */

/* ABI version of the loaded library (== DDSP_HIP_ABI_VERSION it was built with). */
int ddsp_hip_abi_version(void);

/*
 * The process-global test / tuning hooks below (ddsp_osc_set_tiling, ddsp_noise_set_generic, ddsp_gru_set_mode,
 * ddsp_gru_set_fault_step) change every later launch of the process.  They only work when the environment variable
 * DDSP_TEST_HOOKS=1 was set at the moment the library was loaded (tests/conftest.py and the tools/ scripts do that);
 * in any other process they return DDSP_EPERM for a non-default value and change nothing, so a production process cannot
 * flip them by accident.  Returns 1 when the hooks are enabled.
 */
int ddsp_test_hooks_enabled(void);

/*
 * Bytes of device scratch ddsp_osc_forward needs for a [B,T,H] problem (any hop): frame-rate increments fp32 [B,T,H] +
 * normalised amplitudes fp32 [B,T,H] + an fp64 region of [B,T,H] (chunk totals, or frame-start phases + their superblock
 * totals) + a few int32 arrays of at most B*T entries (16*B*T*H bytes + ~2*B*(T+3)/4*H + 12*B*T + 4*T*(B+16) + flag words;
 * 256-byte aligned parts).
 */
size_t ddsp_osc_scratch_bytes(int B, int T, int H);

/*
 * Harmonic oscillator bank (harmonic_oscillator.py:57-62; .live :64-75 when live_in != NULL).
 *   f0 [B,T,1] Hz, c [B,T,H] harmonic amplitudes, a [B,T,1] loudness  ->  y [B, T*hop]
 *   scratch        >= ddsp_osc_scratch_bytes(B,T,H) bytes, 256-byte aligned
 *   live_in  [H]   nullable: phase offsets added to the first increment row of batch row 0 (:70)
 *   live_out [H]   nullable: receives the last phase row of batch row 0 (:72); must not alias live_in
 *   dbg_phi  [B,T*hop,H] nullable (tests only): the wrapped phases, bit-exact w.r.t. torch CPU
 * Inputs are not modified.  Requires T*hop < 2^24 (exact fp32 sample indices).
 */
int ddsp_osc_forward(const float *f0, const float *c, const float *a, float *y, void *scratch,
                     const float *live_in, float *live_out, float *dbg_phi,
                     int B, int T, int H, int hop, int sample_rate, void *stream);
/*
 * The same with flags.  DDSP_OSC_KEEP_FRAME_SCRATCH: the caller will hand `scratch` to ddsp_osc_backward, which re-walks
 * the frame-rate layout (start phase of every frame); without it the forward is free to take the chunked form
 * (power-of-two hops >= 64), whose scratch the backward refuses (it then returns NaN gradients).
 */
#define DDSP_OSC_KEEP_FRAME_SCRATCH 1u
int ddsp_osc_forward_ex(const float *f0, const float *c, const float *a, float *y, void *scratch,
                        const float *live_in, float *live_out, float *dbg_phi,
                        int B, int T, int H, int hop, int sample_rate, unsigned flags, void *stream);

/*
 * Filtered noise (filtered_noise.py:40-53).
 *   Hmag [B,T,F] filter magnitudes -> y [B, T*hop]; per frame: zero-phase IR (irfft, length 2(F-1)),
 *   periodic-Hann window, re-wrapped to hop samples (cropped when hop < 2(F-1)), then the first hop
 *   samples of the linear convolution with uniform noise in [-1,1); frames are concatenated.
 *   uniform [B,T,hop] nullable: the U[0,1) draw (what torch.rand returned, filtered_noise.py:44-48).
 *                      NULL => drawn on the device with Philox4x32-10 from (seed, offset); that stream is
 *                      NOT the torch CPU generator's (documented in DESIGN.md).
 *   accumulate != 0: y += noise instead of y = noise (fuses decoder.py:132 `harmonics + noise`).
 */
int ddsp_noise_forward(const float *Hmag, const float *uniform, float *y,
                       int B, int T, int F, int hop, uint64_t seed, uint64_t offset,
                       int accumulate, void *stream);
/* Same with the in-kernel draw starting at *counter_dev (a device uint64 the caller advances between calls, e.g. by a
 * node of the same hipGraph: a replayed graph then draws fresh noise every time).  counter_dev is only read. */
int ddsp_noise_forward_counter(const float *Hmag, float *y, int B, int T, int F, int hop, uint64_t seed,
                               const uint64_t *counter_dev, int accumulate, void *stream);
/* The same launch with a caller-provided workspace (device memory, 16-byte aligned, contents undefined before and after).
 * ddsp_noise_workspace_bytes is 0 for the shapes that have no use for one; for the reference's default shape (195 bands at
 * hop 512, config/default.py:15,19: 2(F-1) = 388 has no radix-2 transform) at >= 512 frames it holds the cosine operand and the
 * impulse responses of the whole batch, which are then ONE split-bf16 matrix-core product (csrc/ddsp_noise_ir.hip) instead of
 * F x S/4 cosine sums per frame pair: 2x faster end to end at large batches (the forward takes it from 4 096 frames on, the
 * backward from 512: below, the extra launches cost more than the sums).  A NULL / too small workspace takes the ddsp_noise_forward path
 * (same results within rounding).  uniform and counter_dev exclude each other (both NULL: the draw starts at `offset`). */
size_t ddsp_noise_workspace_bytes(int B, int T, int F, int hop);
int ddsp_noise_forward_ws(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop, uint64_t seed,
                          uint64_t offset, const uint64_t *counter_dev, int accumulate, void *workspace, size_t workspace_bytes,
                          void *stream);
/* Backward of the same (ddsp_noise_backward / _counter with a workspace of ddsp_noise_workspace_bytes: for the default shape the
 * correlation runs in the in-LDS FFT form and dH = dz C^T is one matrix-core product, instead of the direct kernels' F x S/2
 * cosine sums per frame).  The workspace need not be the forward's. */
int ddsp_noise_backward_ws(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop, uint64_t seed,
                           uint64_t offset, const uint64_t *counter_dev, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward of ddsp_osc_forward w.r.t. c and a (autograd of harmonic_oscillator.py:24-62; f0 carries no gradient,
 * decoder.py:105).  `fwd_scratch` is the scratch buffer the matching ddsp_osc_forward call filled (same B,T,H,hop,
 * sample_rate, same tiling); `bwd_scratch` >= ddsp_osc_backward_scratch_bytes(B,T,H).
 *   grad_y [B,T*hop] -> grad_c [B,T,H], grad_a [B,T,1]
 */
size_t ddsp_osc_backward_scratch_bytes(int B, int T, int H);
int ddsp_osc_backward(const float *grad_y, const float *f0, const float *c, const float *a, const void *fwd_scratch,
                      void *bwd_scratch, float *grad_c, float *grad_a, int B, int T, int H, int hop, int sample_rate,
                      void *stream);

/*
 * Backward of ddsp_noise_forward w.r.t. Hmag (autograd of filtered_noise.py:40-53; the noise draw is a constant).
 * `uniform`/`seed`/`offset` must be the forward call's.   grad_y [B,T*hop] -> grad_H [B,T,F]
 */
int ddsp_noise_backward(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop,
                        uint64_t seed, uint64_t offset, void *stream);
/* The backward of a ddsp_noise_forward_counter call: the same draw, read from the same device counter (which the caller
 * advances only after both). */
int ddsp_noise_backward_counter(const float *grad_y, float *grad_H, int B, int T, int F, int hop, uint64_t seed,
                                const uint64_t *counter_dev, void *stream);

/*
 * Tuning hook (benchmarks only): force the number of harmonics each lane keeps in registers
 * (one of 4,8,12,13,15,16,20,23,25); 0 restores the automatic choice.  Process-global, not
 * thread-safe; results are identical for every setting.
 */
int ddsp_osc_set_tiling(int harmonics_per_lane);
/* Test / tuning hook: 1 = frame kernels for every shape (the round 1-3 decomposition, one lane group per frame), 0 = automatic
 * (chunked form where it applies and the batch fills the row blocks to >= 88 %), 2 = chunked form for every eligible shape
 * whatever the batch (tests of small batches).  Same results within rounding. */
int ddsp_osc_set_path(int path);
/* What ddsp_osc_forward would launch for this shape on the current device (HOST array of >= 8 ints): out[0] harmonics per
 * lane, [1] lanes per row group, [2] 1 = chunked form, then its [3] chunk length in samples, [4] chunks per row,
 * [5] row blocks, [6] compute units and [7] resident workgroups per unit the chunk length was sized for. */
int ddsp_osc_plan(int B, int T, int H, int hop, int sample_rate, int *out, int cap);
/* Diagnostic, SYNCHRONISES `stream`: the shader clock (GHz) one wavefront of the synth kernel of the LAST ddsp_osc_forward
 * on `scratch` (same B, T, H, hop, sample_rate, same hooks) ran at: in-kernel shader-clock ticks over 100 MHz wall-clock
 * ticks between that wavefront's start and end.  ghz is a HOST pointer; 0.0 if the kernel did not run.  Not for launch paths. */
int ddsp_osc_clock(const void *scratch, int B, int T, int H, int hop, int sample_rate, double *ghz, void *stream);

/* Wavefronts per CU (1..8; 0 = the default, 4) of the hop-128 / 65-band noise kernel's persistent grid.  A production knob, not a
 * test hook: at the eight that fit, the kernel's power density makes an MI355X drop its shader clock for the ~25 ms that follow,
 * which costs the kernels around it more than the noise kernel gains; where the clock gives way differs from box to box (between
 * 4 and 8).  The default is safe on every box measured; a caller may measure its own box (the Python package's
 * calibrate_noise_residency) and set more.  Results are bit-identical whatever the value.  Process-global, read once per launch. */
int ddsp_noise_set_residency(int waves_per_cu);
int ddsp_noise_get_residency(void);

/* Test / tuning hook (process-global, read once per launch): bit 0 forces the generic one-frame-per-workgroup noise kernels
 * (any hop) instead of the batched ones (hop % 8 == 0, tile fits LDS); bit 1 keeps the direct (time-domain) forms where the
 * in-LDS FFT form would run (hop 512 with 2(F-1) <= hop); bit 2 takes the FFT form for hop 256 too (correct, not faster);
 * bit 3 keeps the batched kernel where the wavefront-private form would run (hop 128, 65 bands); bit 4 keeps the cosine sums
 * where ddsp_noise_forward_ws would take the matrix product;
 * (l + 1) << 8 forces 64 >> l frames per workgroup in the batched forward kernel (l = 0..3); 0 restores the defaults.
 * Same results within rounding. */
int ddsp_noise_set_generic(int on);

/*
 * Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 *   ddsp_profile_enable(capacity)  capacity > 0: pre-create that many event pairs and start recording one
 *                                  pair around every kernel launch; capacity <= 0: stop and free them.
 *   ddsp_profile_read(ids, ms, cap) HOST arrays; waits for the recorded events, returns how many records were
 *                                  written (kernel id: 1 frame totals, 2 superblock scan, 3 synth, 4 noise;
 *                                  5 the 195-band noise product; elapsed milliseconds) and resets the pool.  Never called
 *                                  from a launch path.
 *   ddsp_profile_select(mask)      record only the kernels whose id bit is set (0: all, the default).  An event pair costs the
 *                                  stream ~4 us: bench.py records only the dominant kernel inside its timed region.
 */
int ddsp_profile_enable(int capacity);
int ddsp_profile_select(unsigned kernel_mask);
int ddsp_profile_read(int *kernel_ids, float *ms, int cap);

/*
 * Recurrence of the control network's GRU (model/autoencoder/decoder.py:60-65 builds
 * nn.GRU(2*width, units, layers, batch_first=True); :91 runs it).  One persistent launch
 * per direction replaces the ~20 library launches per time step of the stock path; the input projection
 * gi = x W_ih^T + b_ih (and the weight-gradient GEMMs) stay library GEMMs on the caller's side.
 * Single layer, unidirectional, gate order r|z|n as in torch; fp32; Hd <= 512.
 *
 *   ddsp_gru_scratch_bytes(B, Hd)   device scratch for either direction (status word + hand-off granules)
 *   ddsp_gru_max_batch(Hd, backward) rows one launch accepts on the current device (callers split larger
 *                                   batches: rows are independent); 0 if Hd is unsupported
 *   ddsp_gru_forward   gi [B,T,3Hd], w_hh [3Hd,Hd], b_hh [3Hd] (nullable), h0 [B,Hd] (nullable = zeros)
 *                      -> y [B,T,Hd] (all h_t), hT [B,Hd]; gates [B,T,3Hd] and hn [B,T,Hd] (both nullable
 *                      together) receive r|z|n and W_hn h_{t-1} + b_hn for the backward
 *   ddsp_gru_backward  dy [B,T,Hd], dhT [B,Hd] (nullable) + the forward's tensors
 *                      -> d_gi [B,T,3Hd] (gradient of gi), d_gh [B,T,3Hd] (gradient of W_hh h + b_hh), dh0 [B,Hd]
 *   ddsp_gru_status    HOST int*: 0, or 1 if a workgroup gave up waiting for its peers (outputs are then NaN);
 *                      synchronous copy, tests / diagnostics only
 * The grid is sized to be co-resident (one workgroup per CU) and checked against the runtime's occupancy answer
 * (DDSP_ERANGE if it could not be); launches of one process on one device are ordered behind each other on the device
 * (event wait on the launching stream; not inside a stream capture); every wait inside is bounded (2 s): on a time-out
 * the status word is raised and every output of the unfinished steps is NaN.
 */
size_t ddsp_gru_scratch_bytes(int B, int Hd);
int ddsp_gru_max_batch(int Hd, int backward);
int ddsp_gru_forward(const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *y, float *hT,
                     float *gates, float *hn, void *scratch, int B, int T, int Hd, void *stream);
int ddsp_gru_backward(const float *dy, const float *dhT, const float *w_hh, const float *h0, const float *y,
                      const float *gates, const float *hn, float *d_gi, float *d_gh, float *dh0, void *scratch,
                      int B, int T, int Hd, void *stream);
int ddsp_gru_status(const void *scratch, int *status_host);
/* The same recurrences for torch.autocast callers (train/train.py:50 `precision=16`): the products h W_hh^T (forward) and
 * (dr, dz, dhn) W_hh (backward) run on the matrix cores in bf16 with fp32 accumulation (v_mfma_f32_16x16x32_bf16), every
 * tensor at the boundary, the gate arithmetic and the hand-off stay fp32.  At most 16 rows per group: ddsp_gru_max_batch(Hd, 2)
 * rows per launch.  Results differ from the fp32 entry points at bf16 precision (~1e-3).
 * Backward io_type (ABI 3): 0 = fp32 arrays as above; DDSP_IO_BF16 (= 1) = `d_gi` / `d_gh` are written as bf16 arrays -- what the
 * autocast GEMMs behind the recurrence consume, so that no cast pass runs on them. */
int ddsp_gru_forward_bf16(const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *y, float *hT,
                          float *gates, float *hn, void *scratch, int B, int T, int Hd, void *stream);
int ddsp_gru_backward_bf16(const float *dy, const float *dhT, const float *w_hh, const float *h0, const float *y,
                           const float *gates, const float *hn, void *d_gi, void *d_gh, float *dh0, void *scratch,
                           int B, int T, int Hd, int io_type, void *stream);
/* Test hooks (process-global bit mask; 0 restores the default).  Bit 0: deal every group's workgroups over all XCDs (odd
 * blockIdx modulus) instead of keeping a group on one XCD -- bitwise the same results either way (placement only changes
 * speed).  Bit 1: fault injection -- workgroup 0 withholds its publishes from step ddsp_gru_set_fault_step() on and the
 * spin bound drops to 20 ms, so that the time-out path (status word, NaN in every unfinished output) can be tested. */
int ddsp_gru_set_mode(int mode);
int ddsp_gru_set_fault_step(int step);

/*
 * The FIRST block of the f0 / loudness stacks (model/autoencoder/decoder.py:9-39 with n_input = 1, :43-44):
 * Linear(1 -> D) -> LayerNorm -> LeakyReLU as one pass each way.  x [rows] fp32 (the one input feature), w [D] (the Linear's
 * [D, 1] weight), bias [D]; y [rows, D] in io_type (0 fp32, DDSP_IO_BF16, DDSP_IO_F16); mean / rstd [rows] kept for the backward.
 * The backward returns the four parameter gradients (fp32, deterministically summed) and NO input gradient -- callers whose
 * x requires one use the separate Linear and ddsp_ln_lrelu_*.  D = 256 or 512 (DDSP_ERANGE otherwise).
 * scratch >= ddsp_outer_ln_lrelu_scratch_bytes(D).
 */
size_t ddsp_outer_ln_lrelu_scratch_bytes(int D);
int ddsp_outer_ln_lrelu_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, void *y,
                                float *mean, float *rstd, long rows, int D, float eps, float slope, int io_type, void *stream);
int ddsp_outer_ln_lrelu_backward(const void *grad_y, const float *x, const float *w, const float *bias, const void *y,
                                 const float *gamma, const float *mean, const float *rstd, float *grad_w, float *grad_bias,
                                 float *grad_gamma, float *grad_beta, void *scratch, long rows, int D, float slope,
                                 int io_type, void *stream);

/*
 * Column sums of a row-major [M, N] matrix (fp32 io_type 0, bf16 DDSP_IO_BF16, fp16 DDSP_IO_F16) -> out [N] fp32: the bias
 * gradient of the control network's dense layers (decoder.py:9-39, :60-72), deterministic.  scratch: ddsp_colsum_scratch_bytes(N).
 */
size_t ddsp_colsum_scratch_bytes(int N);
int ddsp_colsum(const void *x, float *out, void *scratch, long M, int N, int io_type, void *stream);

/*
 * Framing of the multi-scale spectral loss (loss/mss_loss.py:11-33 on torch.stft semantics: center=True, reflect padding,
 * window of n_fft taps, frames = 1 + N / hop): everything around the batched library FFT of one scale.
 *   ddsp_stft_frames           x [B,N] -> frames [B, frames, n_fft] = x[reflect(f*hop + j - n_fft/2)] * window[j], contiguous
 *   ddsp_stft_frames_backward  grad_frames -> grad_x [B,N]: overlap-add and the padding's adjoint as a gather (deterministic;
 *                              accumulate != 0 adds to grad_x)
 * n_fft % 4 == 0, n_fft <= 8192, N > n_fft / 2.
 */
int ddsp_stft_frames(const float *x, const float *window, float *frames, long B, long N, int n_fft, int hop, void *stream);
int ddsp_stft_frames_backward(const float *grad_frames, const float *window, float *grad_x, long B, long N, int n_fft, int hop,
                              int accumulate, void *stream);

/*
 * One whole scale of the multi-scale spectral loss (loss/mss_loss.py:17-31: spectrogram of both signals, mean |P - Q| +
 * alpha * mean |log2(Q + eps) - log2(P + eps)|) in one kernel, transforms included (in-LDS FFTs; n_fft a power of two in
 * [64, 2048], L > n_fft / 2, torch.stft center / reflect semantics with the given window and hop):
 *   out3        {loss, linear term, log term} of this scale
 *   grad_frames nullable; [B * (1 + L / hop), n_fft] = d loss / d (windowed frame of x_pred), to be folded back onto the
 *               waveform with ddsp_stft_frames_backward (which applies the window)
 *   scratch     ddsp_mss_scale_scratch_bytes() bytes
 * eps must be a normal positive float (>= 1.1754944e-38; the kernels take the hardware log2 / reciprocal of P + eps), else
 * DDSP_EINVAL.  The window is read as [n_fft] floats (8-byte aligned for n_fft = 2048).
 */
size_t ddsp_mss_scale_scratch_bytes(void);
int ddsp_mss_scale_supported(int n_fft);
int ddsp_mss_scale(const float *x_pred, const float *x_true, const float *window, float *grad_frames, void *scratch, float *out3,
                   long B, long L, int n_fft, int hop, float alpha, float eps, void *stream);

/*
 * Reverb (model/ddsp/reverb.py:8-49).  noise [length], t [length] (seconds), decay / wet: one device
 * float each (the module's parameters, read on the device: nothing is synchronised).
 *
 * ddsp_reverb_impulse          build_impulse (:24-29) written straight into the buffer the convolution wants: taps
 *                              [0, min(length, n_out)) = noise * exp(-softplus(-decay) * t * 500) * sigmoid(wet) with tap 0
 *                              forced to 1 (:28), zeros up to n_out -- i.e. `F.pad(impulse, (0, n_out - length))` of :34,
 *                              which CROPS when n_out < length.
 * ddsp_reverb_impulse_backward gradient of the first n_used (<= length) taps w.r.t. noise [length] (zero where cropped and
 *                              at tap 0), decay and wet (one float each); deterministic.
 * ddsp_spectral_mul            y[r,f] = x[r,f] * k[f] on interleaved re/im spectra ([rows, bins] x [bins]): the product of
 *                              fft_convolve (filtered_noise.py:28-30) with one kernel shared by the rows.  The 2N-point real
 *                              transforms themselves stay library FFTs on the caller's side.
 * ddsp_spectral_mul_backward   gk[r,f] = g[r,f] conj(k[f]) (-> irfft -> grad of the signal; nullable) and
 *                              s[f] = sum_r g[r,f] conj(x[r,f]) (-> irfft -> grad of the kernel; nullable, rows summed in order).
 * ddsp_reverb_live             live_forward (:40-49): history_out = [history_in[n:], x]; y[j] = the last n samples of the
 *                              causal convolution of that window with the impulse, computed directly (n x length
 *                              multiply-adds, no FFT).  x [n], y [n], history_* [length], history_out must not alias
 *                              history_in; scratch >= ddsp_reverb_live_scratch_bytes(length, n); n <= length.
 */
int ddsp_reverb_impulse(const float *noise, const float *decay, const float *wet, const float *t, float *impulse,
                        int length, int n_out, void *stream);
int ddsp_reverb_impulse_backward(const float *grad_impulse, const float *noise, const float *decay, const float *wet,
                                 const float *t, float *grad_noise, float *grad_decay, float *grad_wet, int length,
                                 int n_used, void *stream);
int ddsp_spectral_mul(const float *x_ri, const float *k_ri, float *y_ri, long rows, long bins, void *stream);
int ddsp_spectral_mul_backward(const float *g_ri, const float *x_ri, const float *k_ri, float *gk_ri, float *s_ri,
                               long rows, long bins, void *stream);
size_t ddsp_reverb_live_scratch_bytes(int length, int n);
int ddsp_reverb_live(const float *x, const float *history_in, float *history_out, const float *noise, const float *decay,
                     const float *wet, const float *t, float *y, void *scratch, int length, int n, void *stream);

/*
 * One scale of the multi-scale spectral loss (loss/mss_loss.py:11-33: L1 of the power spectrograms + alpha * L1 of their
 * log2) fused into one pass over the two complex STFTs (interleaved re/im, n_bins complex bins each), with the
 * gradient w.r.t. the predicted STFT produced in the same pass (grad_ri nullable).  out3 (device) = {loss, linear term,
 * log term}; scratch >= ddsp_spectral_loss_scratch_bytes().  Deterministic summation.  The STFTs stay library FFTs.
 */
size_t ddsp_spectral_loss_scratch_bytes(void);
int ddsp_spectral_loss(const float *pred_ri, const float *true_ri, float *grad_ri, void *scratch, float *out3,
                       long n_bins, float alpha, float eps, void *stream);

/*
 * The control heads' output non-linearity (model/autoencoder/decoder.py:110-116): y = 2 * sigmoid(x)^2.3026 + 1e-7 over n
 * fp32 elements, and its backward grad_x = grad_y * dy/dx (x is the forward's input), one elementwise pass each.
 */
int ddsp_scaled_sigmoid_forward(const float *x, float *y, long n, void *stream);
int ddsp_scaled_sigmoid_backward(const float *x, const float *grad_y, float *grad_x, long n, void *stream);
/* The controller's three heads (decoder.py:96-100) after ONE GEMM on their concatenated weights: modified_sigmoid of x
 * [rows, n0+n1+n2] (io_type 0 fp32, 1 bf16, 2 fp16: the GEMM's output type) written as three dense fp32 tensors [rows,n0],
 * [rows,n1], [rows,n2]; the backward takes their three fp32 gradients and writes grad_x in x's type. */
int ddsp_heads_sigmoid_forward(const void *x, float *out0, float *out1, float *out2, long rows, int n0, int n1, int n2,
                               int io_type, void *stream);
int ddsp_heads_sigmoid_backward(const void *x, const float *g0, const float *g1, const float *g2, void *grad_x, long rows,
                                int n0, int n1, int n2, int io_type, void *stream);

/*
 * LayerNorm followed by LeakyReLU over rows of D = 256, 512, 768 or 1024 fp32 elements (the MLP blocks of
 * model/autoencoder/decoder.py:9-39 after their Linear): y = lrelu(gamma * (x - mean) * rstd + beta); the forward keeps
 * mean / rstd per row for the backward, which returns grad_x and (deterministically summed) grad_gamma / grad_beta -- and, when
 * grad_xsum is not null (ABI 3), the column sums of grad_x [D]: the bias gradient of the Linear in front of the block, which then
 * needs no pass of its own.  scratch >= ddsp_ln_lrelu_scratch_bytes(D).
 */
size_t ddsp_ln_lrelu_scratch_bytes(int D);
int ddsp_ln_lrelu_forward(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                          long rows, int D, float eps, float slope, void *stream);
int ddsp_ln_lrelu_backward(const float *grad_y, const float *x, const float *y, const float *gamma, const float *mean,
                           const float *rstd, float *grad_x, float *grad_gamma, float *grad_beta, float *grad_xsum, void *scratch,
                           long rows, int D, float slope, void *stream);
/* The same passes on 16-bit activations (io_type: DDSP_IO_BF16 / DDSP_IO_F16) for torch.autocast callers: x, y and their
 * gradients are bf16 / fp16 arrays (read and written as such: no cast pass on either side), gamma / beta, the row
 * statistics, the parameter gradients and all arithmetic stay fp32.  Reference: train/train.py:50 (`precision=16`). */
#define DDSP_IO_BF16 1
#define DDSP_IO_F16 2
int ddsp_ln_lrelu_forward_16(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                             long rows, int D, float eps, float slope, int io_type, void *stream);
int ddsp_ln_lrelu_backward_16(const void *grad_y, const void *x, const void *y, const float *gamma, const float *mean,
                              const float *rstd, void *grad_x, float *grad_gamma, float *grad_beta, float *grad_xsum, void *scratch,
                              long rows, int D, float slope, int io_type, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* DDSP_HIP_H */
