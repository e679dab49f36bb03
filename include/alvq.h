/*
 * alvq.h -- C ABI of libalvq.so: the MI355X (gfx950) kernels behind the VQ-VAE
 * train-step hot path of guy3540/Acoustic_Locating_VQ-VAE.
 *
 * The reference has no FFI layer: its operator API for this path is the
 * torch.nn.Module surface of src/acoustic_locating_vq_vae/vq_vae/ (SURVEY 8b), and
 * the ATen ops those modules call.  Each entry point below replaces one ATen op
 * call site (cited as file:line into /root/reference/src/acoustic_locating_vq_vae/).
 *
 * Conventions (all entry points):
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless noted;
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library
 *     never allocates, frees or retains device memory;
 *   - launches are asynchronous on the caller's `stream` (a hipStream_t passed as
 *     void*); no host synchronisation inside, so calls are graph-capturable;
 *   - return 0 on success; a negative ALVQ_E* for arguments rejected before any
 *     launch; a positive value is the hipError_t of a failed launch.
 *     alvq_last_error() returns a thread-local message for the last failure;
 *   - tensors are dense row-major fp32 unless the name says otherwise
 *     ("_bf16": storage is bfloat16, accumulation is always fp32).
 */
#ifndef ALVQ_H
#define ALVQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALVQ_OK 0
#define ALVQ_EINVAL (-1)      /* bad pointer / dimension                        */
#define ALVQ_EUNSUPPORTED (-2) /* shape or flag combination not implemented      */

/* weight layouts for the conv family */
#define ALVQ_W_OIK 0 /* w[M][C][KW]: nn.Conv1d weight used forward, or ConvTranspose1d weight used for its data-grad */
#define ALVQ_W_IOK 1 /* w[C][M][KW] read flipped+transposed: nn.ConvTranspose1d forward, or nn.Conv1d data-grad       */

/* storage dtypes */
#define ALVQ_F32 0
#define ALVQ_BF16 1

const char* alvq_version(void);
const char* alvq_last_error(void);

/* Dispatch options: which kernel variant serves a launch (never what it computes).  Each starts from the environment
 * variable ALVQ_<NAME> (read once, at the first use of the library) or its default, and can be changed at run time here;
 * the launch path reads an atomic, never the environment.  Names / defaults:
 *   wide_min_tiles 192   bf16: 256 x 256-tile conv kernels only for problems with at least this many such tiles
 *   conv_v2 1, conv_k3 1 bf16: the 256 x 256 kernels at all / the shared-slab width-3 form
 *   wgrad_v3 3           bf16 weight gradient without bias: v3 kernels for width 1 (bit 0) / width 3 (bit 1)
 *   fx_rows 0            f16mx / bf16x3 conv row tile: 0 automatic, 128 or 256 forced
 *   fx_narrow 1          f16mx: 128-channel m-tile for fp32-NCL outputs of <= 128 channels; bf16x3: the 128-channel m-tiles
 *                        (outputs of <= 128 channels, problems with fewer than 192 tiles of 256 x 256)
 *   vq_reg 1             quantiser argmin, D <= 256: x rows in registers + double-buffered codebook tile (0: the
 *                        LDS-stationary kernel; bit-identical results)
 * alvq_set_option returns ALVQ_EINVAL for an unknown name; alvq_get_option returns INT64_MIN. */
int alvq_set_option(const char* name, int64_t value);
int64_t alvq_get_option(const char* name);

/* ------------------------------------------------------------------------------------------------
 * 1-D convolution, stride 1, "same" padding, KW in {1,3}:  im2col-free implicit GEMM on MFMA.
 *
 *   acc[b,m,l] = sum_c sum_t A_t[m,c] * x[b,c,l+t-(KW-1)/2]           (zero outside [0,L))
 *   A_t[m,c]   = w[m][c][t]            (ALVQ_W_OIK)
 *              = w[c][m][KW-1-t]       (ALVQ_W_IOK)
 *   v = acc (+ bias[m]) (+ skip1[b,m,l]) (+ skip2[b,m,l]);  if relu: v = max(v,0);
 *   if mask: v = mask[b,m,l] > 0 ? v : 0;   y = v;   if y2: y2 = v + post[b,m,l]
 *
 * x is (B,C,L), y/skip1/skip2/mask/post/y2 are (B,M,L), all contiguous.  Optional pointers may be NULL.
 * Replaces: nn.Conv1d forward (vq_vae/convolutional_encoder.py:17-23,40; convolutional_vq_vae.py:32-37,95;
 * deconvolutional_decoder.py:19-25,69; modules/residual.py:37-54), nn.ConvTranspose1d forward
 * (deconvolutional_decoder.py:35-59,73-77), F.relu / nn.ReLU(True) (residual.py:36,47;
 * residual_stack.py:46), the residual and encoder skip adds (residual.py:66; convolutional_encoder.py:42),
 * and -- through autograd -- the data-gradient of each of those with the ReLU-backward mask fused.
 * ---------------------------------------------------------------------------------------------- */
int alvq_conv1d_f32(const float* x, const float* w, const float* bias,
                    const float* skip1, const float* skip2, const float* mask, const float* post,
                    float* y, float* y2,
                    int B, int C, int M, int L, int KW, int w_layout, int relu, void* stream);

/* Weight gradient of the same convolution, split over the batch*length reduction:
 *   dw_t[m,c] = sum_b sum_l dy[b,m,l] * x[b,c,l+t-(KW-1)/2]
 * written as dw[m][c][t] (ALVQ_W_OIK) or dw[c][m][KW-1-t] (ALVQ_W_IOK, with dy/x roles as the caller passes them).
 * `workspace` must hold alvq_conv1d_wgrad_workspace_bytes(...) bytes; partial sums are reduced in a fixed
 * order (bitwise reproducible).  accumulate!=0 adds into dw (shared residual weights, residual_stack.py:40-41).
 * If dbias != NULL also dbias[m] (+)= sum_b sum_l dy[b,m,l].
 * Replaces: autograd's convolution_backward weight/bias grads for every call site listed above. */
int64_t alvq_conv1d_wgrad_workspace_bytes(int B, int C, int M, int L, int KW);
int alvq_conv1d_wgrad_f32(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                          int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vector quantiser (vq_vae/vector_quantizer.py:29-58).  x is the contiguous (B,D,L) buffer viewed as
 * (N,D) rows in memory order (no permute, :32); codebook is (K,D).
 * ---------------------------------------------------------------------------------------------- */
/* idx[n] = argmin_k fl(fl(|x_n|^2 + |e_k|^2) - 2 x_n.e_k), lowest k on ties (:34-38).
 * min_dist may be NULL.  workspace: alvq_vq_argmin_workspace_bytes(N,K) bytes. */
int64_t alvq_vq_argmin_workspace_bytes(int64_t N, int K, int D);
int alvq_vq_argmin_f32(const float* x, const float* codebook, int64_t* idx, float* min_dist, void* workspace,
                       int64_t N, int K, int D, void* stream);

/* q_st = x + (E[idx] - x) (:43,54);  sq_partials[ALVQ_VQ_PARTIALS] = per-workgroup partial sums of
 * (E[idx]-x)^2 (fixed order, reproducible);  hist[k] += #rows with idx==k (int32[K], caller zeroes; may be NULL). */
#define ALVQ_VQ_PARTIALS 1024
int alvq_vq_gather_loss_f32(const float* x, const float* codebook, const int64_t* idx, float* q_st,
                            float* sq_partials, int32_t* hist, int64_t N, int K, int D, void* stream);

/* m = sum(sq_partials)/(N*D); loss = m + beta*m (:46-52); perplexity = exp(-sum p log(p+1e-10)), p = hist/N (:55-56).
 * out[0] = loss, out[1] = perplexity. */
int alvq_vq_finalize_f32(const float* sq_partials, const int32_t* hist, float* out, int64_t N, int K, int D,
                         float beta, void* stream);

/* Backward:  dx = g + gl*(2*beta/(N*D))*(x - E[idx]);
 * dE[k] += gl*(2/(N*D)) * sum_{n: idx_n = k} (E[k] - x_n)  (skipped when dE == NULL, i.e. _train_vq False);
 * g = grad wrt q_st (may be NULL = 0), gl = *grad_loss (device scalar, may be NULL = 1).  dE is accumulated into
 * (zero it first for a plain gradient).  No atomics: workgroups (code, row segment) gather their rows in row order
 * and sum them in a fixed pattern, partials are added in segment order, so the result is bitwise reproducible.
 * `workspace`: alvq_vq_backward_workspace_bytes(K,D) bytes, required when dE != NULL.  D % 4 == 0 and D <= 512,
 * or D <= 256. */
int64_t alvq_vq_backward_workspace_bytes(int K, int D);
int alvq_vq_backward_f32(const float* g, const float* grad_loss, const float* x, const float* codebook,
                         const int64_t* idx, float* dx, float* dE, void* workspace, int64_t N, int K, int D, float beta,
                         void* stream);

/* Dense one-hot encodings (N,K) fp32 (:39-40) -- only materialised for get_latent_representation callers. */
int alvq_onehot_f32(const int64_t* idx, float* encodings, int64_t N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Jitter (vq_vae/modules/jitter.py:42-70): y[b,c,l] = x[b,c,src[l]];  backward: dx[b,c,l] = src[l]==l ? dy : 0.
 * src is an int32[L] DEVICE array produced on the host with the reference's numpy call order.
 * ---------------------------------------------------------------------------------------------- */
int alvq_jitter_gather_f32(const float* x, const int32_t* src, float* y, int64_t rows, int L, int backward, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Callers' per-step arithmetic (scripts/train_speech.py:63-64,74; train_rir.py:42-49).
 * ---------------------------------------------------------------------------------------------- */
/* y[b,c,l] = (x' - mean_c x') / (std_c x' + 1e-8), x' = |x| if take_abs else x; unbiased std over dim=1;
 * x, y are (B,C,L). */
int alvq_standardise_f32(const float* x, float* y, int B, int C, int L, int take_abs, void* stream);

/* loss[0] = mean((a-b)^2) over n elements (F.mse_loss, train_speech.py:74); workspace: ALVQ_EW_PARTIALS floats. */
#define ALVQ_EW_PARTIALS 1024
int alvq_mse_f32(const float* a, const float* b, float* loss, void* workspace, int64_t n, void* stream);
/* grad = grad_loss[0] * (2/n) * (a - b)   (grad_loss: device scalar, NULL = 1). */
int alvq_mse_backward_f32(const float* a, const float* b, const float* grad_loss, float* grad, int64_t n, void* stream);

/* out[0..n) = value (out 16-byte aligned): the flat gradient buffer's zeroing at the top of a step
 * (optimizer.zero_grad(), train_speech.py:88) as a kernel node -- a memset node recorded during stream capture
 * was observed not to be ordered before the kernel behind it on replay. */
int alvq_fill_f32(float* out, float value, int64_t n, void* stream);

/* out = a + b (elementwise), used where a gradient has two consumers. */
int alvq_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* y[r] = mean over L of x[r][0..L) and its backward dx[r][l] = dy[r] / L: the optional average pooling of the latent,
 * torch.mean(z, dim=2, keepdim=True) (convolutional_vq_vae.py:96-97), with rows = B * D. */
int alvq_row_mean_f32(const float* x, float* y, int64_t rows, int L, void* stream);
int alvq_row_mean_backward_f32(const float* dy, float* dx, int64_t rows, int L, void* stream);

/* out = t > 0 ? dy : 0: ReLU backward from the saved post-ReLU activation (F.relu, residual_stack.py:46). */
int alvq_relu_mask_f32(const float* dy, const float* t, float* out, int64_t n, void* stream);

/* (B,R,C) -> (B,C,R) dense transpose (materialises permute(0,2,1), train_rir.py:45). */
int alvq_transpose_f32(const float* x, float* y, int B, int R, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam on a flat parameter buffer (torch.optim.Adam(lr, betas, eps, amsgrad=False), train_speech.py:154).
 * step is 1-based. grad_scale multiplies the gradient first (1/world after the all-reduce).
 * ---------------------------------------------------------------------------------------------- */
int alvq_adam_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Graph-replayable form: the step-dependent scalars come from device memory,
 * scalars = {lr / (1 - beta1^step), sqrt(1 - beta2^step), grad_scale}. */
/* skip (nullable, here and in alvq_adam_pack_batch / alvq_adam_segments_f32): a device float, the SKIP SLOT of the flat
 * gradient buffer.  Non-zero = this step saturated an fp16-range format on some rank (alvq_range_flag_to_slot wrote it
 * before the step's all-reduce, which sums it over the ranks): the launch leaves parameters, moments and packed images
 * bit for bit as they were -- the dynamic-loss-scaling "skip" pattern, decided on the device, no host sync. */
int alvq_adam_dev_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* scalars, float beta1, float beta2, float eps, const float* skip, void* stream);

/* Device-side step counter for the form above: scalars is EIGHT floats, {.., .., .., step, skipped, -, -, -} (both counters
 * start at 0).  Each call does step += 1 and rewrites the first three for that step, in stream order -- no host staging
 * buffer, so a host that queues steps ahead of the device cannot overwrite a step's scalars before its Adam launch reads
 * them (hyper-parameters and powers in double, as torch.optim.Adam computes them; exact step count up to 2^24).
 * prev_skip (nullable): the skip slot, still holding the PREVIOUS step's verdict (the flat gradient buffer is zeroed inside
 * the step): if non-zero that step was not applied -- step stays, skipped += 1 (a GradScaler-skipped step does not advance
 * torch.optim.Adam's step count either).  With prev_skip the call also folds the range flag's bits into its sticky word, so
 * the step about to run sees only its own. */
int alvq_adam_advance_f32(float* scalars, double lr, double beta1, double beta2, double grad_scale,
                          const float* prev_skip, void* stream);

/* *slot = 1.0f if the fp16-range formats' flag holds any bit raised since the step's guarded alvq_adam_advance_f32, else
 * 0.0f.  The last launch of a step's backward; slot is element 0 of the flat gradient buffer. */
int alvq_range_flag_to_slot(float* slot, void* stream);

/* ------------------------------------------------------------------------------------------------
 * STFT power spectrogram (scripts/genereate_dataset.py:90-91,37,39,47-49; torchaudio Spectrogram semantics:
 * center=True reflect pad, periodic Hann(n_fft), one-sided, window-normalised, |.|^2).
 * wave (B,S) -> power (B, n_fft/2+1, 1+S/hop).
 * ---------------------------------------------------------------------------------------------- */
int alvq_stft_power_f32(const float* wave, float* power, int B, int S, int n_fft, int hop, void* stream);
/* float64 form: the reference's echoed signal is float64 (scipy convolve output, genereate_dataset.py:38-39). */
int alvq_stft_power_f64(const double* wave, double* power, int B, int S, int n_fft, int hop, void* stream);

/* The same transform keeping the phase: spec (B, n_fft/2+1, T) complex, stored as interleaved (re, im) pairs of the
 * waveform's precision, divided by sqrt(sum(window^2)) (power=None, normalized=True). */
int alvq_stft_complex_f32(const float* wave, float* spec, int B, int S, int n_fft, int hop, void* stream);
int alvq_stft_complex_f64(const double* wave, double* spec, int B, int S, int n_fft, int hop, void* stream);

/* scipy.signal.convolve(wave, h, mode="same") in float64 for a float32 waveform (B,S) and float64 impulse responses
 * (genereate_dataset.py:38): out[b][i] = sum_j h[b][j] * wave[b][i + (Nh-1)/2 - j].  h_batch_stride: elements between
 * consecutive impulse responses (0: one response shared by the batch).  Nh <= S. */
int alvq_fir_same_f64(const float* wave, const double* h, double* out, int B, int S, int Nh, int h_batch_stride, void* stream);

/* Dataset-generator arithmetic on the two complex STFTs (genereate_dataset.py:41-49), per batch item:
 *   r = S / (E + 1e-8);  rir_pow = |r / max|r||^2;  wiener[f] = |sum_t E conj(S) / (sum_t S conj(S) + 1e-8)|^2;
 *   speech_pow = |S|^2 (fp32);  echoed_pow = |E|^2 (fp64).
 * speech_spec: complex64 (B,F,T) interleaved; echoed_spec: complex128 (B,F,T) interleaved; workspace: B*F doubles.
 * All sums and the max run in a fixed order. */
int alvq_spec_rir_wiener_f64(const float* speech_spec, const double* echoed_spec, float* speech_pow, double* echoed_pow,
                             double* rir_pow, double* wiener, double* workspace, int B, int F, int T, void* stream);

/* ================================================================================================
 * bf16 throughput path (BASELINE configs[1]: "batch=64 bf16").  Storage bf16, accumulation fp32.
 *
 * Activations live in the "NLC-padded" layout: a bf16 matrix act[rows][Cp] with channels contiguous,
 *   row(b,l) = 1 + b*(L+1) + l,   rows = alvq_nlc_rows(B,L) (rounded up to 128),   Cp = alvq_nlc_channels(C),
 * row 0, the row after each sample, rows past the batch and channels >= C are zero.  The caller allocates
 * alvq_nlc_guard_rows() readable rows before row 0 and after the last row; pointers passed below point at row 0.
 * ============================================================================================== */
int64_t alvq_nlc_rows(int B, int L);
int alvq_nlc_channels(int C);
int alvq_nlc_guard_rows(void);

/* Re-pack an fp32 weight into the K-contiguous bf16 image the kernels read: wp[tap][Mp128][Cp] = A_t[m][c]
 * (A_t as defined for alvq_conv1d_f32; zero padded).  wp holds alvq_packed_weight_elems(M,C,KW) bf16 values. */
int64_t alvq_packed_weight_elems(int M, int C, int KW);
int alvq_pack_weight_bf16(const float* w, void* wp, int M, int C, int KW, int w_layout, void* stream);

/* The same for n weights in one launch (a train step re-packs every conv weight, in the forward and in the
 * data-gradient layout, after each optimiser update: one launch instead of ~20).  `descs` is a HOST array; it is
 * copied into the kernel argument, so the call can be captured into a hipGraph.  planes = 1: bf16 images;
 * planes = 2: the hi + lo images of the split-bf16 path (as alvq_pack_weight_bf16x3); planes = 3: the H + Q images of
 * the f16mx path (two images of alvq_packed_weight_elems 2-byte units each, weight-class scale). */
typedef struct alvq_pack_desc {
  const float* w;   /* fp32 weight, (M,C,KW) for ALVQ_W_OIK or (C,M,KW) for ALVQ_W_IOK */
  void* wp;         /* packed image(s): planes * alvq_packed_weight_elems(M,C,KW) bf16 values */
  int32_t M, C, KW, w_layout;
} alvq_pack_desc;
int alvq_pack_weights_bf16_batch(const alvq_pack_desc* descs, int n, int planes, void* stream);

/* Adam + packing in one pass (scripts/train_speech.py:154 + the re-pack above): for each conv weight (fp32, native
 * layout (dim0, dim1, KW), with its gradient and Adam moments) apply torch.optim.Adam's update -- the arithmetic of
 * alvq_adam_dev_f32, bit for bit, scalars = {lr/bias_correction1, sqrt(bias_correction2), grad_scale} on the device -- and
 * write the updated weight's packed image read as OIK (wp_oik: M = dim0, C = dim1) and / or as IOK (wp_iok: C = dim0,
 * M = dim1, taps flipped) while the new values are in registers.  Images must have been packed in full once before
 * (their padding is not rewritten).  planes as in alvq_pack_weights_bf16_batch.  descs: HOST array. */
typedef struct alvq_adam_pack_desc {
  float* w; const float* g; float* m; float* v;
  void* wp_oik; void* wp_iok;      /* either may be NULL */
  int32_t dim0, dim1, KW;
} alvq_adam_pack_desc;
int alvq_adam_pack_batch(const alvq_adam_pack_desc* descs, int n, int planes, const float* scalars,
                         float beta1, float beta2, float eps, const float* skip, void* stream);
/* alvq_adam_dev_f32 over the nseg element ranges [lo[i], hi[i]) of flat buffers, one launch (lo / hi: HOST arrays): the
 * parameters alvq_adam_pack_batch does not own (biases, the codebook). */
int alvq_adam_segments_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                           const int64_t* lo, const int64_t* hi, int nseg, const float* scalars,
                           float beta1, float beta2, float eps, const float* skip, void* stream);

/* Input boundary for a source whose CHANNEL axis is already contiguous (round 4; SURVEY 8(b) "element strides of x"):
 * x is (B, L, C) fp32 contiguous -- i.e. the model input (B, C, L) handed over as `t.permute(0, 2, 1)` of a contiguous t,
 * which is what scripts/train_rir.py:45 and scripts/train_echoed_speech.py:66 do -- and y the NLC-padded activation in format
 * fmt (1 bf16, 2 bf16x3 hi + lo planes, 3 f16mx H + Q planes): y[row(b,l)][c] = x[b][l][c], no transposition in either
 * direction.  standardise != 0 fuses train_rir.py:43-44 into the same pass: per (b, c) the mean and the UNBIASED std over l,
 * (v - mean) / (std + 1e-8), in alvq_standardise_f32's arithmetic and summation order (bit-identical to standardise ->
 * transpose -> convert); requires 2 <= L <= alvq_rows_to_nlc_max_std_rows().  take_abs applies |.| first. */
int alvq_rows_to_nlc(const float* x, void* y, int B, int C, int L, int fmt, int standardise, int take_abs, void* stream);
int alvq_rows_to_nlc_max_std_rows(void);

/* (B,C,L) fp32 -> NLC-padded bf16 (the boundary conversion for x, quantized and incoming gradients). */
int alvq_ncl_to_nlc_bf16(const float* x, void* y, int B, int C, int L, void* stream);

/* NLC-padded bf16 -> (B,C,L) fp32. */
int alvq_nlc_to_ncl_f32(const void* x, float* y, int B, int C, int L, void* stream);

/* out = t > 0 ? dy : 0 over n bf16 elements (n % 8 == 0). */
int alvq_relu_mask_bf16(const void* dy, const void* t, void* out, int64_t n, void* stream);

/* Same fused convolution as alvq_conv1d_f32 on NLC-padded bf16 operands and a packed weight.  Exactly one of
 * y (NLC bf16, row stride alvq_nlc_channels(M)) and y_ncl ((B,M,L) fp32, bias-only epilogue) is non-NULL.
 * Sign bits (both optional, NLC output only): a byte per 8 consecutive channels of a row,
 * bits[row * alvq_nlc_channels(M) / 8 + m / 8], bit (m % 8) = (value > 0); alvq_nlc_rows(B,L) * alvq_nlc_channels(M) / 8
 * bytes.  relu_bits_out receives the signs of the stored y, so that the data-gradient launch that later needs
 * "* (y > 0)" can pass them as mask_bits (instead of `mask` = the tensor itself) and read 1/16 of the bytes. */
int alvq_conv1d_bf16(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                     const void* mask, const void* post, void* y, void* y2, float* y_ncl,
                     int B, int C, int M, int L, int KW, int relu, const void* mask_bits, void* relu_bits_out,
                     void* stream);

/* Weight (and bias) gradient from NLC-padded bf16 dy [rows][Mp] and x [rows][Cp]; fp32 result in the weight's
 * native layout, same contract as alvq_conv1d_wgrad_f32. */
int64_t alvq_conv1d_wgrad_bf16_workspace_bytes(int B, int C, int M, int L, int KW);
int alvq_conv1d_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace,
                           int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream);

/* The same weight gradient summed over nseg (1..4) operand pairs of identical shape in ONE launch:
 * dw (+)= sum_i wgrad(dy[i], x[i]).  This is how the R uses of a shared residual weight (residual_stack.py:40-41)
 * are accumulated: one longer contraction and one split reduction instead of R.  dy / x are HOST arrays of device
 * pointers; no bias gradient (those layers have none).  Workspace: alvq_conv1d_wgrad_bf16_workspace_bytes. */
int alvq_conv1d_wgrad_bf16_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                 int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream);
/* Host-only: the number of split partials (each KW*M*C floats at the start of `workspace`) a launch over nseg segments
 * writes; never more than alvq_conv1d_wgrad_*_workspace_bytes sizes for any nseg in 1..4 (the launch re-checks it).
 * with_bias: the single-segment launch that also produces dbias.  -1 for unsupported arguments. */
int alvq_conv1d_wgrad_bf16_splits(int B, int C, int M, int L, int KW, int nseg, int with_bias);

/* Deferred, batched split reduction (bf16 / fp16 family).  With accumulate = ALVQ_WGRAD_DEFER the four weight-gradient
 * entry points of the family launch the contraction only: dw / dbias are not touched (dw may be NULL), the
 * alvq_conv1d_wgrad_bf16_splits(...) partials stay at the start of `workspace` ([split][KW][M][C] fp32) and, for the
 * single-segment launch with dbias != NULL, the bias partials ([split][pad64(M)] fp32) at
 * workspace + alvq_conv1d_wgrad_bf16_bias_offset(...).  The caller gives every deferred launch its OWN workspace and later
 * sums all of them with ONE alvq_wgrad_reduce_batch launch: per descriptor  dst (+)= scale * sum_s partial[s]  in split
 * order -- the same sums in the same order as the immediate reduction (bitwise identical), as one launch that fills the
 * chip instead of one small bandwidth-bound launch per layer.  A bias reduction is the descriptor
 * {KW = 1, M = 1, C = channels, w_layout = ALVQ_W_OIK, stride = pad64(channels)}.  descs: HOST array. */
#define ALVQ_WGRAD_DEFER 2
int64_t alvq_conv1d_wgrad_bf16_bias_offset(int B, int C, int M, int L, int KW);
typedef struct alvq_reduce_desc {
  const float* partial;   /* [splits][stride] */
  float* dst;             /* weight gradient in its native layout (w_layout), or a bias gradient */
  const float* scale;     /* device scalar multiplied into the sum (undoes a loss scale), or NULL */
  int32_t splits, KW, M, C, w_layout, accumulate;
  int64_t stride;         /* elements between consecutive partials: KW * M * C, or the padded length of bias partials */
} alvq_reduce_desc;
int alvq_wgrad_reduce_batch(const alvq_reduce_desc* descs, int n, void* stream);

/* ================================================================================================
 * Split-bf16 ("bf16x3") path: fp32-grade results on the bf16 matrix cores (gfx950 has no TF32/xf32 and its
 * exact-fp32 MFMA runs at 1/16 of the bf16 rate).  Every value is two bf16 planes, hi = bf16(v) and
 * lo = bf16(v - hi); products are hi*hi + hi*lo + lo*hi with fp32 accumulation (~1e-5 relative).
 * An NLC-padded operand's lo plane follows its hi plane (guard rows included) at +alvq_nlc_plane_bytes(B,L,C);
 * a packed weight's lo image follows its hi image at +2*alvq_packed_weight_elems(M,C,KW) bytes.  Pointers passed
 * below point at row 0 of the hi plane.  Same contracts as the bf16 entry points of the same name.
 * ============================================================================================== */
int64_t alvq_nlc_plane_bytes(int B, int L, int C);
int alvq_pack_weight_bf16x3(const float* w, void* wp, int M, int C, int KW, int w_layout, void* stream);
int alvq_ncl_to_nlc_bf16x3(const float* x, void* y, int B, int C, int L, void* stream);
int alvq_nlc_to_ncl_bf16x3(const void* x, float* y, int B, int C, int L, void* stream);
int alvq_relu_mask_bf16x3(const void* dy, const void* t, void* out, int B, int C, int L, void* stream);
int alvq_conv1d_bf16x3(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                       const void* mask, const void* post, void* y, void* y2, float* y_ncl,
                       int B, int C, int M, int L, int KW, int relu, void* stream);
int64_t alvq_conv1d_wgrad_bf16x3_workspace_bytes(int B, int C, int M, int L, int KW);
int alvq_conv1d_wgrad_bf16x3(const void* dy, const void* x, float* dw, float* dbias, void* workspace,
                             int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream);
/* dw (+)= sum_i wgrad(dy[i], x[i]) over nseg (1..4) pairs of one shape in ONE launch (the R uses of a shared residual
 * weight), as alvq_conv1d_wgrad_bf16_multi. */
int alvq_conv1d_wgrad_bf16x3_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                   int B, int C, int M, int L, int KW, int w_layout, int accumulate, void* stream);
int alvq_conv1d_wgrad_bf16x3_splits(int B, int C, int M, int L, int KW, int nseg);   /* as alvq_conv1d_wgrad_bf16_splits */

/* ================================================================================================
 * "f16mx" split path: fp32-grade results at TWO matrix-pipe units per product.  Every value v is H = fp16(v) plus
 * hi8 = e4m3(v/S), lo8 = e4m3((v-H)/(S*2^-11)); a product is H*H (one fp16 MFMA) + hi8*lo8 + lo8*hi8 (one
 * block-scaled fp8 MFMA at twice the fp16 rate per K), ~1.5e-5 rms per product.  S is a power of two per tensor class
 * (activations and loss-scaled gradients 1, weights 2^-8; csrc/f16mx_common.h).  An NLC-padded operand is two planes of
 * a bf16 plane's geometry: H as [rows][Cp] fp16, then at +alvq_nlc_plane_bytes(B,L,C) Q as [rows][Cp/32][hi8 x32|lo8 x32];
 * packed weights likewise (alvq_pack_weights_bf16_batch with planes = 3).  Same contracts as the bf16x3 entry points of
 * the same name, plus device scalars that carry a loss scale in and out of a backward chain:
 *   alvq_grad_scale_f32    state[0] = S, state[1] = 1/S with S the power of two that puts amax|x| in [2^7, 2^8)
 *                          (state: 4 floats, zero-initialised by the caller once; re-armed by every call)
 *   alvq_ncl_to_nlc_f16mx  multiplies by *scale (NULL = 1) while converting; alvq_nlc_to_ncl_f16mx likewise on the way out
 *   alvq_conv1d_f16mx      y_ncl output multiplied by *out_scale (NULL = 1); mask_bits / relu_bits_out as in
 *                          alvq_conv1d_bf16 (one bit per element, [rows][Mp/8] bytes, bit = stored H > 0)
 *   alvq_conv1d_wgrad_f16mx  dw / dbias multiplied by *inv_scale (NULL = 1)
 *   alvq_f16mx_range_flag  *out (device int) = the format's sticky range flag of the current device, optionally cleared:
 *                          bit 0 = a value of magnitude >= 65504 entered the format (stored saturated: fp16 has no
 *                          larger finite value), bit 1 = a NaN entered it (set by alvq_ncl_to_nlc_f16mx: model inputs
 *                          and gradients entering a backward chain, after the loss scale); bit 2 = a value PRODUCED
 *                          inside a chain reached the limit or is a NaN (an activation above 65504, a loss-scaled gradient
 *                          that outgrew its 2^8 headroom) -- watched by the epilogues of alvq_conv1d_f16mx and
 *                          alvq_conv1d_f16.  A set flag means "this step's results are fp16-saturated somewhere".
 * ============================================================================================== */
int alvq_grad_scale_f32(const float* x, int64_t n, float* state, void* stream);
int alvq_f16mx_range_flag(int* out, int reset, void* stream);
int alvq_ncl_to_nlc_f16mx(const float* x, void* y, int B, int C, int L, const float* scale, void* stream);
int alvq_nlc_to_ncl_f16mx(const void* x, float* y, int B, int C, int L, const float* scale, void* stream);
int alvq_relu_mask_f16mx(const void* dy, const void* t, void* out, int B, int C, int L, void* stream);
int alvq_conv1d_f16mx(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                      const void* mask, const void* post, void* y, void* y2, float* y_ncl,
                      int B, int C, int M, int L, int KW, int relu, const void* mask_bits, void* relu_bits_out,
                      const float* out_scale, void* stream);
int64_t alvq_conv1d_wgrad_f16mx_workspace_bytes(int B, int C, int M, int L, int KW);
int alvq_conv1d_wgrad_f16mx(const void* dy, const void* x, float* dw, float* dbias, void* workspace,
                            int B, int C, int M, int L, int KW, int w_layout, int accumulate,
                            const float* inv_scale, void* stream);
/* dw (+)= inv_scale * sum_i wgrad(dy[i], x[i]) over nseg (1..4) pairs of one shape in ONE launch (the R uses of a shared
 * residual weight), as alvq_conv1d_wgrad_bf16_multi; all segments carry the same loss scale. */
int alvq_conv1d_wgrad_f16mx_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                  int B, int C, int M, int L, int KW, int w_layout, int accumulate,
                                  const float* inv_scale, void* stream);
int alvq_conv1d_wgrad_f16mx_splits(int B, int C, int M, int L, int KW, int nseg);    /* as alvq_conv1d_wgrad_bf16_splits */

/* ================================================================================================
 * fp16 element type of the 16-bit pipeline: the same kernels, layouts (NLC-padded [rows][Cp], packed weights
 * [tap][Mp][Cp], sign bits) and contracts as the "_bf16" entry points of the same name, with fp16 elements, fp16 MFMAs
 * (v_mfma_f32_16x16x32_f16 / 32x32x16_f16) and saturating conversions.  Used by the BACKWARD pass of the "f16mx_hb" mode
 * (f16mx forward: fp32-grade outputs; fp16 backward: gradients under the device-chosen loss scale of alvq_grad_scale_f32,
 * products of fp16 operands with fp32 accumulation, ~5e-4 per product): an fp16 operand may be the H plane of an f16mx
 * activation or of an f16mx packed weight (same geometry: pass the plane's pointer), a mask its sign bits.
 *   alvq_ncl_to_nlc_f16 / alvq_nlc_to_ncl_f16   multiply by *scale (NULL = 1) while converting
 *   alvq_conv1d_f16        y_ncl output multiplied by *out_scale (NULL = 1)
 *   alvq_conv1d_wgrad_f16  dw / dbias multiplied by *inv_scale (NULL = 1); workspace = alvq_conv1d_wgrad_bf16_workspace_bytes
 *   alvq_relu_mask_bf16    serves fp16 buffers unchanged (the test is "element > 0" on the int16 pattern)
 * ============================================================================================== */
int alvq_ncl_to_nlc_f16(const float* x, void* y, int B, int C, int L, const float* scale, void* stream);
int alvq_nlc_to_ncl_f16(const void* x, float* y, int B, int C, int L, const float* scale, void* stream);
int alvq_conv1d_f16(const void* x, const void* wp, const float* bias, const void* skip1, const void* skip2,
                    const void* mask, const void* post, void* y, void* y2, float* y_ncl,
                    int B, int C, int M, int L, int KW, int relu, const void* mask_bits, void* relu_bits_out,
                    const float* out_scale, void* stream);
int alvq_conv1d_wgrad_f16(const void* dy, const void* x, float* dw, float* dbias, void* workspace,
                          int B, int C, int M, int L, int KW, int w_layout, int accumulate,
                          const float* inv_scale, void* stream);
int alvq_conv1d_wgrad_f16_multi(const void* const* dy, const void* const* x, int nseg, float* dw, void* workspace,
                                int B, int C, int M, int L, int KW, int w_layout, int accumulate,
                                const float* inv_scale, void* stream);

/* ================================================================================================
 * Location head: LocationModule.fc_1 = nn.Linear(L*K, M) on the flattened one-hot codes of a
 * spectrogram (vq_vae/location_model/location_model.py:10,21; scripts/train_location.py:69-77 feeds it
 * encodings.reshape(B, 201, 1024)).  On one-hot input the dense product is an embedding bag over the indices the
 * quantiser produced: B*L*M gathered weights instead of a (B, L*K) x (L*K, M) GEMM against a 99.9 %-zero operand.
 * ============================================================================================== */
/* idx[row] = position of the single 1.0 in row `row` of encodings (rows, K); *not_onehot (device int, caller zeroes
 * it) is OR-ed with 1 if any row is not exactly one-hot (then the caller must fall back to the dense product).
 * Inverse of alvq_onehot_f32 for the value get_latent_representation returns (vector_quantizer.py:39-40,58). */
int alvq_onehot_to_index_f32(const float* encodings, int32_t* idx, int* not_onehot, int64_t rows, int K, void* stream);
/* out[i] = (int32) idx[i] for n int64 indices; a value outside [0, K) becomes -1 and ORs 1 into *bad_index (device int,
 * caller-zeroed, sticky) -- a plain narrowing cast would wrap 2^32 + 5 to 5. */
int alvq_indices_to_i32(const int64_t* idx, int32_t* out, int* bad_index, int64_t n, int K, void* stream);
/* out[b][m] = bias[m] + sum_l W[m][l*K + idx[b][l]]      W: (M, L*K) row-major (nn.Linear.weight), idx: (B, L),
 * out: (B, M); sums in a fixed order.  Replaces F.linear at location_model.py:21 for one-hot x.  B*L <= 16384 per call
 * (callers chunk over B).  An index outside [0, K) contributes nothing and ORs 1 into *bad_index (optional device int,
 * sticky, caller-zeroed): no out-of-bounds access; the caller raises, as torch's embedding_bag does. */
int alvq_embedding_bag_fwd_f32(const float* W, const float* bias, const int32_t* idx, float* out,
                               int B, int L, int K, int M, int* bad_index, void* stream);
/* dW[m][l*K + idx[b][l]] += dz[b][m] for every (b, l) -- the caller zero-fills dW (M, L*K) first (alvq_fill_f32) --
 * and dbias[m] (+)= sum_b dz[b][m].  One wave owns a weight row: no atomics, fixed order.  Out-of-range indices are
 * skipped and flagged as in the forward. */
int alvq_embedding_bag_bwd_f32(const float* dz, const int32_t* idx, float* dW, float* dbias,
                               int B, int L, int K, int M, int accumulate_bias, int* bad_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALVQ_H */
