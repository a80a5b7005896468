/* libaddvisor_hip -- C ABI of the MI355X (gfx950) ADDvisor explanation hot path.
 *
 * The reference (davidcombei/xAI-Audio-Deepfakes) has no FFI: its boundary is a set of Python
 * modules whose arithmetic lives in torch / transformers.  Each entry point below replaces the
 * device work behind one of those calls; the citation says which (paths relative to the
 * reference repo, `transformers/...` = the HF package it imports).  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (hipMalloc / torch allocator),
 *     except `advh_gemm_desc` and the small parameter structs, which are host memory read
 *     during the call;
 *   - `stream` is a hipStream_t (passed as void*; NULL = the default stream); calls only
 *     enqueue work, they never synchronise, allocate or free, so they may be graph-captured;
 *   - return value: 0 = ok, <0 = ADVH_E* error (no exceptions cross the ABI, nothing is printed);
 *   - no global state except immutable tables (FFT twiddles) built by advh_init();
 *   - thread-compatible: concurrent calls must use different streams and buffers.
 */
#ifndef ADDVISOR_HIP_H
#define ADDVISOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* advh_stream_t;

enum {
    ADVH_OK = 0,
    ADVH_EINVAL = -1,   /* bad argument (shape, alignment, null pointer)            */
    ADVH_ELAUNCH = -2,  /* hipLaunchKernel / hipGetLastError reported a failure     */
    ADVH_ENOTINIT = -3, /* advh_init() has not been called on this device          */
    ADVH_EUNSUPPORTED = -4
};

/* Library / build identification: returns a static NUL-terminated string ("gfx950 ..."). */
const char* advh_version(void);

/* Build the immutable device tables (FFT twiddles) on the current device and raise the dynamic-LDS
 * limits of the kernels that need more than 64 KiB.  Call once per process and device, outside
 * any stream capture.  Idempotent. */
int advh_init(void);

/* Tuning knobs (process-wide, not thread-safe; set before launching work):
 *   "stft_frames_per_workgroup" = 8 (default) | 16 : STFT / ISTFT frames per workgroup.
 *   "attention_bwd_mfma_f32" = 0 (default) | 1 : advh_attention_bwd_split on the fp32-input matrix instruction for every
 *       head dim (1) instead of the split-arithmetic kernel for head dims <= 64 (0); same results to 1e-6 (A/B runs). */
int advh_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------------------------
 * STFT  -- replaces AudioProcessor.compute_stft (audioprocessor.py:82-112):
 *   pad/crop to L samples, torch.stft(n_fft=1024, hop, win, window=None|window, center=True,
 *   pad_mode="reflect", onesided) -> X, |X|, angle(X).
 * wave   [B][wave_stride] fp32, n_in valid samples per clip (n_in < L: zero-padded tail,
 *        n_in > L: cropped);  window: NULL = rectangular `win`-long window, else `win` floats.
 * X      [B][513][T][2] fp32 (complex64, t fastest) or NULL;  mag, phase [B][513][T] or NULL.
 * T must equal 1 + L / hop.  n_fft is fixed at 1024 (the only size the reference uses).        */
int advh_stft_forward(const float* wave, int64_t wave_stride, int n_in, int B, int L, int hop, int win,
                      const float* window, float* X, float* mag, float* phase, int T, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Masked ISTFT -- replaces, fused in one kernel, the mask application + polar recombination of
 *   loss_function.py:36-45 (mode ADVH_MASK_LINEAR) / LMAC_metrics.py:136-153 (ADVH_MASK_LOG1P)
 *   and AudioProcessor.compute_invert_stft (audioprocessor.py:117-131: torch.istft, length=L).
 * mag, phase [B][513][T] fp32;  mask [B][Fm][Tm] fp32 (the U-Net output; bins outside the
 *   Fm x Tm crop count as mask = 0) or NULL with ADVH_MASK_NONE.
 * wave_in  <- istft( g(mask)   * e^{j phase} ),  wave_out <- istft( g(1-mask) * e^{j phase} );
 *   either may be NULL.  [B][wave_stride], L samples written per clip.                         */
enum { ADVH_MASK_NONE = 0, ADVH_MASK_LINEAR = 1, ADVH_MASK_LOG1P = 2 };
int advh_istft_masked(const float* mag, const float* phase, const float* mask, int Fm, int Tm, int mode,
                      float* wave_in, float* wave_out, int64_t wave_stride, int B, int T, int L, int hop,
                      int win, const float* window, advh_stream_t stream);

/* Same resynthesis from the COMPLEX spectrogram X [B][513][T][2] (advh_stft_forward's X) instead of (|X|, angle X):
 * X' = X * g(mask, |X|) / |X|, which equals g e^{j angle X} of loss_function.py:36-45 / LMAC_metrics.py:136-153 without the
 * atan2 / sincos round trip (more accurate, and no transcendental per bin in the linear domain).  mode: LINEAR or LOG1P. */
int advh_istft_masked_c64(const float* spec, const float* mask, int Fm, int Tm, int mode, float* wave_in, float* wave_out,
                          int64_t wave_stride, int B, int T, int L, int hop, int win, const float* window,
                          advh_stream_t stream);

/* Plain ISTFT of a complex64 spectrogram [B][513][T][2] (audioprocessor.py:117-131). */
int advh_istft_c64(const float* spec, float* wave, int64_t wave_stride, int B, int T, int L, int hop, int win,
                   const float* window, advh_stream_t stream);

/* Band-swap resynthesis (the data generator of hifigan.py:196-228 and train_logReg_swapping.py:64-92; SURVEY.md §8(f)
 * rank 2): for band z in [0, nbands) the bins [k0 + z*kw, k0 + (z+1)*kw) of the complex64 spectrogram spec_a
 * [B][513][T] are replaced by those of spec_b and the result is inverted (torch.istft semantics as advh_istft_c64);
 * waves [nbands][B][L], band stride band_stride >= B*wave_stride elements.  One launch, grid z = band.            */
int advh_istft_bandswap(const float* spec_a, const float* spec_b, int k0, int kw, int nbands, float* waves,
                        int64_t wave_stride, int64_t band_stride, int B, int T, int L, int hop, int win,
                        const float* window, advh_stream_t stream);

/* Backward of advh_istft_masked from ONE resynthesised waveform to the mask (LMACLoss backward,
 * loss_function.py:36-47 / SURVEY.md §8(f) rank 1): g_wave = dL/d wave [B][L] (row stride g_stride), which = 0 for
 * the mask-in branch (a = m M), 1 for mask-out (a = (1 - m) M); dmask [B][Fm][Tm] is overwritten.  `mask` is only
 * read in ADVH_MASK_LOG1P mode.  Adjoint of torch.istft: divide by the window envelope, zero-extend, frame with the
 * synthesis window, rfft, scale by c_k / n_fft.                                                                    */
int advh_istft_masked_bwd(const float* g_wave, int64_t g_stride, const float* mag, const float* phase, const float* mask,
                          int Fm, int Tm, int mode, int which, float* dmask, int B, int T, int L, int hop, int win,
                          const float* window, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Implicit GEMM on the matrix cores (fp16 operands, fp32 accumulate):
 *
 *     out[row(m) + col(n)] = act( sum_k A(m,k) * W[n][k] + bias[n] ) + resid[row(m) + col(n)]
 *
 * replaces the cuDNN / cuBLAS calls behind nn.Conv1d, nn.Linear (transformers/models/wav2vec2/
 * modeling_wav2vec2.py:254-572) and nn.Conv2d / nn.ConvTranspose2d (addvisor.py:12-84).
 * Activations are channels-last fp16; a convolution tap is a constant offset from the row's base
 * address, listed per 16-byte K-chunk in `ktab`.
 *
 * Row enumeration: m = (b*Hg + h)*Wg + w, 0 <= m < M.  Rows with (h,w) outside the window
 * [h0,h1) x [w0,w1) are "halo" rows: they read a safe in-window row and, if `halo_zero`, have
 * zeros written (so a zero-haloed NHWC output map is produced complete by one launch).
 * A row base, in 16-byte chunks (8 halfs):  b*a_sB[s] + h*a_sH[s] + w*a_sW[s] + a_c0[s]  (source s).
 * K-chunk c (0 <= c < Ktot/8) reads 8 halfs at  A_s + 8*(rowbase_s + (ktab[c] & 0x7fffffff)),
 * s = ktab[c] >> 31.  Ktot % 64 == 0; padding chunks must pair zero weights with any readable chunk.
 * W is [w_rows][Ktot] fp16, w_rows >= N rounded up to the tile's BN (extra rows are read, never used).
 * Output element offset: b*o_sB + h*o_sH + w*o_sW + o_c0 + (n / n_div)*o_sNhi + n % n_div + z*o_sZ
 * (n_div % 4 == 0; every term a multiple of 4 elements).  `nz` batches (grid z) advance the
 * operands by a_sZ (chunks), w_sZ (elements), bias_sZ, o_sZ.                                     */
enum { ADVH_ACT_NONE = 0, ADVH_ACT_GELU = 1, ADVH_ACT_LEAKY = 2 };
enum { ADVH_TILE_AUTO = 0, ADVH_TILE_128x128 = 1, ADVH_TILE_256x64 = 2, ADVH_TILE_256x32 = 3,
       ADVH_TILE_256x128_W8 = 9, ADVH_TILE_128x256_W8 = 10 /* 512-thread instances of the same single-buffer kernel, 64x64 wave tiles, 2 workgroups
                                                               per CU: tuner candidates (they win one K = 4096 shape); fp16 operands only */ };

typedef struct advh_gemm_desc {
    const void* A0;       /* fp16 source 0                                   */
    const void* A1;       /* fp16 source 1 (skip-concat by pointer) or NULL  */
    const void* W;        /* fp16 [nz][w_rows][Ktot]                         */
    const int32_t* ktab;  /* [Ktot/8] chunk offsets, bit 31 selects A1       */
    const float* bias;    /* fp32 [nz][N] or NULL                            */
    const void* resid;    /* residual, addressed like out, or NULL           */
    void* out_h;          /* fp16 output or NULL                             */
    void* out_f;          /* fp32 output or NULL                             */
    int32_t M, N, Ktot, w_rows;
    int32_t Hg, Wg, h0, h1, w0, w1, halo_zero;
    int64_t a_sB[2], a_sH[2], a_sW[2], a_c0[2], a_sZ[2];
    int64_t w_sZ, bias_sZ;
    int64_t o_sB, o_sH, o_sW, o_c0, o_sNhi, o_sZ;
    int32_t n_div, nz;
    int32_t act;          /* ADVH_ACT_*                                      */
    float slope;          /* LeakyReLU slope                                 */
    int32_t resid_f32;    /* 1: resid is fp32, 0: fp16                       */
    int32_t ktab_identity;/* 1: ktab[c] == c for all c (plain GEMM rows): kernels may skip the lookup */
    void* out_h2;         /* optional second fp16 output = LeakyReLU_{slope2}(value written to out_h):
                             the pre-activated copy the next HiFi-GAN conv consumes (ResBlock1)          */
    float slope2;
    /* ConvTranspose1d by phase decomposition (stride r = ph_r > 0): column block n / n_div is the output phase,
       row w the input position; the element is written only if 0 <= w*ph_r + n/n_div - ph_pad < ph_T.   */
    int32_t ph_r, ph_pad, ph_T;
    /* backward support (input-gradient chain of the frozen embedder, captum_saliency.py:131-135):
       out_pre: fp16 copy of the value BEFORE `act` (saved for the activation derivative), or NULL;
       dact_src: fp16 tensor addressed like out; if set, the value is multiplied by GELU'(dact_src[o])
       (after bias, before resid): turns a dgrad GEMM's output into the gradient w.r.t. the previous
       layer's pre-activation.                                                                          */
    void* out_pre;
    const void* dact_src;
    /* wide = 1: the rows of W are packed permuted inside every 32-row block -- packed row R holds output channel
       32 (R>>5) + 8 ((R>>2)&3) + 4 ((R>>4)&1) + (R&3) -- so that a lane's accumulators are 8 consecutive channels and
       the epilogue moves 16 bytes of fp16 per lane.  Requires N, n_div, o_c0 and every o_s* stride % 8 == 0
       (checked).  wide = 0: packed row R is channel R.                                                     */
    int32_t wide;
    /* row pitch of W in elements; 0 = Ktot.  Lets a launch (or grid-z batch) reduce over a K-slice of a wider
       K-major matrix: the split-K weight-gradient GEMMs of the U-Net training step.                       */
    int64_t w_ld;
    /* super-column width in N tiles for the 256-thread kernels' tile order (0 = plain row-major order): all M tiles of
       `sc` N-tiles are walked before the next `sc`, keeping that weight slice L2-resident.                     */
    int32_t sc;
    /* second level of the column -> address split (ConvTranspose2d with all kh*kw sub-pixels in ONE launch): with
       q = n / n_div, the offset is (q / n_sub)*o_sNhh + (q % n_sub)*o_sNhi instead of q*o_sNhi; n_sub <= 1 = off. */
    int32_t n_sub;
    int64_t o_sNhh;
    /* two-level grid-z batch (256-thread kernels only; the other tiles return ADVH_EUNSUPPORTED): with nz_lo > 1 the
       batch index z in [0, nz) splits into zh = z / nz_lo, zw = z % nz_lo and the operands advance by
       a_sZ*zh + a_sZ2*zw (chunks) and o_sZ*zh + o_sZ2*zw (elements); W and bias still advance by w_sZ*z, bias_sZ*z.
       The four output-parity classes of a fused ConvTranspose2d(2,2)+Conv2d launch are such a batch.
       z_inner = 1: the nz batches of one tile are dispatched next to each other (z fastest in the workgroup order)
       instead of batch after batch, so batches that read the same operand rows meet in L2.                  */
    int32_t nz_lo, z_inner;
    int64_t a_sZ2[2];
    int64_t o_sZ2;
    /* plain = 1 (requires ktab_identity and one source): every row m in [0, M) -- valid or not -- may be read at chunk
       a_c0[0] + m * a_sW[0] (+ z * a_sZ[0]), K contiguous chunks; a_sB / a_sH / the window are then used for the OUTPUT
       addressing only.  Lets the 128x128 tile run its affine-row loader (Linear layers, feature-encoder Conv1d: the
       caller guarantees that the last row's K chunks are inside the allocation).                                  */
    int32_t plain;
    /* plain_out = 1 (with plain): every row is valid (the window is the whole grid), output row m starts at
       o_c0 + m * o_sW + the batch offset and the columns are one block (n_div >= N, no phases, no sub-pixel split):
       the epilogue skips the row decomposition and the column divisions.                                          */
    int32_t plain_out;
    /* split = 1: fp32-class mode.  Every fp16 operand and fp16 output is a PAIR of planes (hi, lo) in the split format
       x = hi + lo * 2^-11 (csrc/device_math.h: hi = fp16(x), lo = fp16((x - hi) * 2^11)); the lo plane of source s starts
       a_lo[s] chunks behind A_s, that of W w_lo elements behind W, that of out_h / out_h2 / a fp16 resid o_lo elements
       behind the hi plane, all addressed like the hi plane.  The kernel issues three MFMAs per fragment pair
       (Wh*Ah + (Wh*Al + Wl*Ah) * 2^-11, fp32 accumulate): the arithmetic class of the reference's fp32 layers.
       Tiles: 128x128, 256x64, 256x32 (AUTO picks as for fp16); out_pre / dact_src are not supported.              */
    int32_t split;
    int64_t a_lo[2];
    int64_t w_lo;
    int64_t o_lo;
} advh_gemm_desc;

int advh_gemm_f16(const advh_gemm_desc* desc, int tile, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * wav2vec2 waveform front end -- replaces zero_mean_unit_var_norm (classifier_embedder.py:59-63)
 * and feature-encoder layer 0 (Conv1d(1,C0,10,stride 5) [+ GroupNorm(C0,C0) + GELU],
 * transformers/models/wav2vec2/modeling_wav2vec2.py:254-323) behind AudioProcessor.extract_features
 * (audioprocessor.py:69-77).
 * wave [B][wave_stride] fp32 (n_in valid samples, padded / cropped to L);  w0 [C0][10] fp32.
 * mode 0 ("group"): out = GELU(GroupNorm(conv)), gamma/beta [C0];  mode 1 ("layer"): out = conv + bias0.
 * out: fp16 channels-last [B][P0][C0], rows t in [T0,P0) written as zeros; T0 = (L-10)/5 + 1.
 * normalize: 1 = apply zero_mean_unit_var_norm first; 0 = wave is already normalised (the raw
 *   `wav2vec2(input_values)` call of audioprocessor.py:76).
 * stats_ws: [B][2] fp32 workspace (clip mean, 1/(std+1e-7));  norm_ws: [B][C0][2] fp32 (mode 0);
 * mr_ws: NULL, or [B][C0][2] fp32 <- per-channel (mean, rstd) of the GroupNorm, kept for the backward.  */
int advh_w2v2_frontend(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                       const float* bias0, const float* gamma, const float* beta, int mode, int normalize, float* stats_ws,
                       float* norm_ws, float* mr_ws, void* out, int T0, int P0, int C0, advh_stream_t stream);

/* LayerNorm over the last dimension (+ optional GELU): nn.LayerNorm call sites of
 * modeling_wav2vec2.py:275-299, 422-434, 575-654, 689-802.  in: [M][in_ld] fp32 (in_is_f32) or fp16;
 * out_f (fp32) and/or out_h (fp16), row stride out_ld.  C % 4 == 0, C <= 2048. */
int advh_layernorm(const void* in, int in_is_f32, int64_t in_ld, const float* gamma, const float* beta,
                   float* out_f, void* out_h, int64_t out_ld, int M, int C, float eps, int gelu,
                   advh_stream_t stream);
/* Same with an fp16 addend row: y = LayerNorm(in + add_h) -- the residual add of a post-LN encoder layer
 * (modeling_wav2vec2.py:689-726: hidden = attn_residual + attention(...); layer_norm(hidden)) fused into the LayerNorm, so
 * the projection before it stores only its fp16 result instead of reading and re-writing the fp32 stream.  add_h NULL =
 * advh_layernorm. */
int advh_layernorm_add(const void* in, int in_is_f32, int64_t in_ld, const void* add_h, int64_t add_ld, const float* gamma,
                       const float* beta, float* out_f, void* out_h, int64_t out_ld, int M, int C, float eps, int gelu,
                       advh_stream_t stream);

/* Operand gather of the grouped positional Conv1d (modeling_wav2vec2.py:326-379):
 * h [B][T][H] fp32 -> xg [G][B][T+K][H/G] fp16, data in rows [pad_left, pad_left+T), zeros elsewhere
 * (forward: pad_left = K/2; input-gradient pass: K/2 - 1 and h is multiplied by GELU'(dact_src), fp16 [B][T][H]). */
int advh_posconv_gather(const float* h, void* xg, int B, int T, int H, int G, int K, int pad_left, const void* dact_src,
                        advh_stream_t stream);

/* The grouped positional convolution itself as an LDS line-tile launch (csrc/posconv_tile.hip): one workgroup per
 * (group, clip) stages the clip's gathered rows once and streams the group's weights through an LDS ring.
 *   xg    [G][B][T+K][H/G] fp16 from advh_posconv_gather (pad_left = K/2)
 *   W     fp16 [G][K*(H/G)/32 k-steps][H/G rows][32]: weight-norm-folded conv weight [n][tap][ci] of each group, cut
 *         into 32-deep k-steps in (tap, ci) order
 *   out[b][t][c] = resid[b][t][c] + GELU(conv + bias[c])   fp32 [B][T][H] (out may alias resid).
 * Supported: K = 128, H/G in {48, 64}, T <= 256 (ADVH_EUNSUPPORTED otherwise: use the implicit GEMM).         */
typedef struct advh_posconv_desc {
    const void* xg;
    const void* W;
    const float* bias;
    const float* resid;
    float* out;
    int B, T, H, G, K;
} advh_posconv_desc;
int advh_posconv_tile_f16(const advh_posconv_desc* d, advh_stream_t stream);
int advh_posconv_tile_lds_bytes(int Cg, int T);   /* -1 = unsupported geometry */

/* softmax(Q K^T / sqrt(d)) V without mask (modeling_wav2vec2.py:438-548), T <= 256, d in {32, 64}.
 * qkv [B*T][3H] fp16 (q | k | v), ctx [B*T][H] fp16. */
int advh_attention_f16(const void* qkv, void* ctx, int B, int T, int H, int heads, advh_stream_t stream);

/* Time mean-pool + logistic regression head: LMAC_metrics.py:130 pooling + TorchLogReg.forward
 * (classifier_embedder.py:34-38).  h [B][T][H] fp32 -> logit[B], prob[B] (and pooled [B][H] if not NULL). */
int advh_pool_logreg(const float* h, const float* coef, float intercept, float* logit, float* prob,
                     float* pooled, int B, int T, int H, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * U-Net layers that are not GEMM-shaped (addvisor.py:27-84).  mag is torch's [B][Fq][Tq] fp32; the
 * H x W crop is taken by indexing.  NHWC maps are fp16 with a zero halo (PH, PW).
 * advh_unet_stem : e1.block.0 Conv2d(1,32,(5,3),stride (2,1),pad (2,1)) + folded BN + LeakyReLU;
 *                  wgt [32][15], out [B][H/2+2PH][W+2PW][32] (interior written).
 * advh_unet_pack_x: channels [c0,c0+8) of the C-channel d1 concat map <- (mag, 1, 0 x6)  (addvisor.py:79); the 1 is
 *                  the in-image indicator the fused ConvTranspose2d+Conv2d launch multiplies up1's bias with.
 * advh_unet_head : mask_head Conv2d(32,1,1) + Sigmoid -> mask (and pre-sigmoid logits) [B][H][W] fp32. */
int advh_unet_stem(const float* mag, int Fq, int Tq, int B, int H, int W, const float* wgt, const float* bias,
                   void* out, int PH, int PW, float slope, advh_stream_t stream);
int advh_unet_pack_x(const float* mag, int Fq, int Tq, int B, int H, int W, void* cat, int C, int c0, int PH, int PW,
                     advh_stream_t stream);
int advh_unet_head(const void* y, int B, int H, int W, int PH, int PW, const float* wgt, float bias, float* mask,
                   float* logits, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LMAC faithfulness metrics -- replaces compute_faithfulness / compute_fidelity / compute_AD /
 * compute_AI / compute_AG and their dataset means (LMAC_metrics.py:31-73, 164-172).
 * predictions, theta_out, masked_predictions: [n] fp32 probabilities (clean, mask-in, mask-out).
 * sums6 <- {sum faithfulness, sum fidelity, sum AD, sum AI, sum AG, n} in fp64, fixed summation
 * order; per_clip (or NULL) <- [5][n] fp32 per-clip values. */
int advh_lmac_metrics_accumulate(const float* predictions, const float* theta_out, const float* masked_predictions,
                                 int n, double* sums6, float* per_clip, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * HiFi-GAN V1 generator -- replaces hifi_gan.decode_batch (hifigan.py:106-110, 180; SpeechBrain HIFIGAN,
 * architecture of Kong et al. 2020 config V1) and the mel front end (hifigan.py:163-178).
 * The Conv1d / ConvTranspose1d layers are advh_gemm_f16 launches (plans in addvisor_hip/gemm.py); these are
 * the remaining pieces.  Maps are zero-haloed channels-last fp16 [B][T+2*halo][C].
 * advh_hifigan_pack_mel : mel [B][C][T] fp32 -> map interior.
 * advh_hifigan_mrf_mix  : y = LeakyReLU_slope((a+b+c)/3) over `numel` fp16 elements (the MRF average + the
 *                         activation in front of the next upsampler / conv_post).
 * advh_hifigan_conv_post: Conv1d(C,1,k,"same") + tanh -> wav [B][1][T] fp32; w is [k][C] fp32.
 * advh_mel_log          : out [B][n_mels][T] = log(clamp(fb^T |X|, 1e-5)), fb [F][n_mels], mag [B][F][T].     */
int advh_hifigan_pack_mel(const float* mel, void* out, int B, int C, int T, int halo, advh_stream_t stream);
/* Fidelity options of the SpeechBrain wrapper behind hifigan.py:106-110, 180 (not verifiable offline, hence options):
 * advh_hifigan_pack_mel_pad : as pack_mel with `pad` replicated frames on both sides (the generator's `inference_padding`:
 *                             F.pad(mel, (pad, pad), "replicate")); the map holds T + 2*pad interior rows.
 * advh_halo_fill_f16        : fill the halo of a channels-last fp16 map [B][T+2*halo][C] with zeros (mode 0) or with the
 *                             reflection of the interior about its first / last sample (mode 1: torch "reflect" padding, the
 *                             default padding_mode of SpeechBrain's Conv1d), so "same" convolutions read it in place.      */
int advh_hifigan_pack_mel_pad(const float* mel, void* out, int B, int C, int T, int pad, int halo, advh_stream_t stream);
int advh_halo_fill_f16(void* x, int B, int T, int C, int halo, int mode, advh_stream_t stream);
int advh_hifigan_mrf_mix(const void* a, const void* b, const void* c, void* y, float slope, int64_t numel, advh_stream_t stream);
int advh_hifigan_conv_post(const void* x, const float* w, float bias, float* wav, int B, int C, int T, int halo, int k,
                           advh_stream_t stream);
int advh_mel_log(const float* mag, const float* fb, float* out, int B, int F, int T, int n_mels, advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input-gradient chain of the frozen embedder -- what captum.attr.{Saliency, InputXGradient,
 * IntegratedGradients}(Wav2vec2LogReg) obtain from autograd (captum_saliency.py:84-100, 116-135).
 * Dense products are advh_gemm_f16 launches with transposed weights (dact_src / out_pre above); these are
 * the remaining pieces.  Gradients are fp16 between GEMMs, multiplied by a power-of-two loss scale.
 *
 * advh_layernorm_bwd : dx of y = LN(x)*gamma+beta [then GELU if gelu_fwd] given dy; optionally times
 *                      GELU'(dact_src) and plus `add`; out_f (fp32) and/or out_h (fp16), all [M][C]; with
 *                      remap_P > 0 output row b*remap_T + t is written at row b*remap_P + t.
 * advh_attention_bwd_f16 : dqkv [B*T][3H] from qkv and dctx [B*T][H] (fp16), T <= 256, head_dim 32 / 64.
 * advh_pool_logreg_bwd   : dh[b][t][:] = coef * dlogit[b] / T  (fp32 and/or fp16).
 * advh_w2v2_frontend_bwd_group : dz0 [B][P0][C0] fp16 from dy0 for the GroupNorm front end (z0 recomputed).
 * advh_wave_bwd      : g [B][P0][16] fp32 (= dz0 . w0, columns 0..9) -> dx [B][dx_stride] through the
 *                      stride-5 overlap and the normaliser's Jacobian; multiplied by out_scale (1/loss scale).
 * advh_scale_rows    : y[r][:] = alpha[r] * x[r % x_rows][:] (+ y): IG path points and step accumulation.   */
int advh_layernorm_bwd(const void* x, int x_is_f32, const void* dy, int dy_is_f32, const float* gamma, const float* beta,
                       int gelu_fwd, const float* add, const void* dact_src, float* out_f, void* out_h, int M, int C,
                       float eps, int remap_T, int remap_P, advh_stream_t stream);
int advh_attention_bwd_f16(const void* qkv, const void* dctx, void* dqkv, int B, int T, int H, int heads, advh_stream_t stream);
int advh_pool_logreg_bwd(const float* coef, const float* dlogit, float* dh, void* dh16, int B, int T, int H, advh_stream_t stream);
int advh_w2v2_frontend_bwd_group(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                 const float* gamma, const float* stats_ws, const float* norm_ws, const float* mr_ws,
                                 const void* dy0, float* part_ws, float* sums_ws, void* dz0, int T0, int P0, int C0,
                                 advh_stream_t stream);
int advh_wave_bwd(const float* g, const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* stats_ws,
                  float* dxhat_ws, float* part_ws, int normalize, float out_scale, float* dx, int64_t dx_stride, int T0, int P0,
                  advh_stream_t stream);
int advh_scale_rows(const float* x, int x_rows, const float* alpha, float* y, int rows, int64_t n, int accumulate,
                    advh_stream_t stream);
/* advh_attr_finalize : out = |g| (mode 0, captum Saliency) or x * g (mode 1, InputXGradient / IG with zero baseline).
 * advh_time_mask     : mask[b] = |attr[b]| / (max|attr[b]| + 1e-8) (captum_saliency.py:136-139) and, if wave is
 *                      given, wave_in = wave*mask, wave_out = wave*(1-mask) (:141-143); all [B][n] fp32.          */
int advh_attr_finalize(const float* g, const float* x, float* out, int mode, int64_t total, advh_stream_t stream);
int advh_time_mask(const float* attr, float* mask, float* wave_in, float* wave_out, const float* wave, int B, int64_t n,
                   advh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp32-class ("split") mode.  The reference computes the whole path in fp32 (addvisor.py:12-84,
 * transformers/models/wav2vec2/modeling_wav2vec2.py:254-802 under audioprocessor.py:69-77).  In this mode every tensor
 * that the fp16 path stores as fp16 is a PAIR of fp16 planes (hi, lo) with x = hi + lo * 2^-11 (hi = fp16(x),
 * lo = fp16((x - hi) * 2^11); csrc/device_math.h) -- ~22 significand bits -- and every matrix product runs as three fp16
 * MFMAs with fp32 accumulation (advh_gemm_desc.split).  The entry points below are the split-format variants of the row /
 * direct kernels above: same arithmetic, `*_lo` = distance in ELEMENTS from a tensor's hi plane to its lo plane (both planes
 * share one addressing).  fp32 tensors (residual stream, masks, waveforms, statistics) are unchanged.                     */
/* RANGE of the split format: hi is an fp16, so a value must satisfy |x| <= 65504 (the reference's fp32: 3.4e38).  Every kernel
 * that WRITES a split tensor saturates a larger value (hi = +-65504, lo = the clamped remainder) instead of producing inf /
 * NaN planes, and raises a sticky process-wide flag in host-mapped memory.  advh_split_overflow returns the flag (1 = some
 * kernel that has already run met an out-of-range value, +-inf included, since the last reset; a NaN passes through as NaN planes) and clears it when reset != 0; it reads
 * host memory only -- no synchronisation -- so a kernel still in flight is seen by a later call.  Weights are range-checked
 * on the host when they are packed.                                                                                     */
int advh_split_overflow(int reset);
int advh_w2v2_frontend_split(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                             const float* bias0, const float* gamma, const float* beta, int mode, int normalize, float* stats_ws,
                             float* norm_ws, float* mr_ws, void* out, int64_t out_lo, int T0, int P0, int C0, advh_stream_t stream);
/* in: fp32 rows (in_is_f32, in_lo ignored) or a split pair; add_h: NULL or a split pair; out_h: NULL or a split pair. */
int advh_layernorm_split(const void* in, int in_is_f32, int64_t in_ld, int64_t in_lo, const void* add_h, int64_t add_ld,
                         int64_t add_lo, const float* gamma, const float* beta, float* out_f, void* out_h, int64_t out_ld,
                         int64_t out_lo, int M, int C, float eps, int gelu, advh_stream_t stream);
int advh_posconv_gather_split(const float* h, void* xg, int64_t xg_lo, int B, int T, int H, int G, int K, int pad_left,
                              advh_stream_t stream);
/* softmax(Q K^T / sqrt(d)) V on split q | k | v -> split ctx; T <= 256, head dim a multiple of 8 up to 128 (above 64 the keys
 * stream through LDS in blocks with an online softmax: XLS-R's head dim 120). */
int advh_attention_split(const void* qkv, int64_t qkv_lo, void* ctx, int64_t ctx_lo, int B, int T, int H, int heads,
                         advh_stream_t stream);
int advh_unet_stem_split(const float* mag, int Fq, int Tq, int B, int H, int W, const float* wgt, const float* bias,
                         void* out, int64_t out_lo, int PH, int PW, float slope, advh_stream_t stream);
int advh_unet_pack_x_split(const float* mag, int Fq, int Tq, int B, int H, int W, void* cat, int64_t cat_lo, int C, int c0,
                           int PH, int PW, advh_stream_t stream);
int advh_unet_head_split(const void* y, int64_t y_lo, int B, int H, int W, int PH, int PW, const float* wgt, float bias,
                         float* mask, float* logits, advh_stream_t stream);
/* fp32-class input-gradient chain (captum_saliency.py:116-135: fp32 autograd through the embedder; loss_function.py:46-53 +
 * train_addvisor.py:376: loss.backward() through the frozen embedder).  Dense dgrad products are advh_gemm_f16 launches with
 * desc.split = 1 (transposed split weights, split gradients; dact_src / out_pre are plane pairs sharing o_lo); these are the
 * split-format forms of the row kernels above -- same arithmetic, fp16 tensors replaced by plane pairs -- and the attention
 * backward (head dim a multiple of 8 up to 128, T <= 256): head dims <= 64 in split arithmetic (three fp16 MFMAs per product,
 * csrc/attention_bwd_x3.hip), larger ones on the fp32-input matrix instruction v_mfma_f32_16x16x4_f32 (csrc/attention_bwd_f32.hip;
 * advh_set_option("attention_bwd_mfma_f32", 1) selects it for every head dim).  */
int advh_layernorm_bwd_split(const void* x, int x_is_f32, int64_t x_lo, const void* dy, int dy_is_f32, int64_t dy_lo,
                             const float* gamma, const float* beta, int gelu_fwd, const float* add, const void* dact_src,
                             int64_t dact_lo, float* out_f, void* out_h, int64_t out_lo, int M, int C, float eps, int remap_T,
                             int remap_P, advh_stream_t stream);
int advh_attention_bwd_split(const void* qkv, int64_t qkv_lo, const void* dctx, int64_t dctx_lo, void* dqkv, int64_t dqkv_lo,
                             int B, int T, int H, int heads, advh_stream_t stream);
int advh_pool_logreg_bwd_split(const float* coef, const float* dlogit, float* dh, void* dh16, int64_t dh16_lo, int B, int T, int H,
                               advh_stream_t stream);
/* fp32 -> split format on the device: dst[i] = hi, dst[dst_lo + i] = lo of src[i], i < n (csrc/device_math.h split_f32: saturates
 * and raises the sticky range flag above 65 504).  The per-step weight refresh of the training path (train_addvisor.py:376-378
 * steps the fp32 parameters with Adam; addvisor_hip/gemm.py GemmPlan.load_weights re-packs them).  src 16-byte aligned. */
int advh_split_f32(const float* src, void* dst, int64_t dst_lo, int64_t n, advh_stream_t stream);
/* dh [B][T][H] fp32 times GELU'(dact_src) (split pre-activation of the positional conv) -> split xg, rows [pad_left, pad_left+T) */
int advh_posconv_gather_bwd_split(const float* dh, void* xg, int64_t xg_lo, int B, int T, int H, int G, int K, int pad_left,
                                  const void* dact_src, int64_t dact_lo, advh_stream_t stream);
int advh_w2v2_frontend_bwd_group_split(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                       const float* gamma, const float* stats_ws, const float* norm_ws, const float* mr_ws,
                                       const void* dy0, int64_t dy_lo, float* part_ws, float* sums_ws, void* dz0, int64_t dz_lo,
                                       int T0, int P0, int C0, advh_stream_t stream);
/* fp32-class form of advh_resblock_pair_f16 for the 32-channel stage (csrc/resblock_pair_x3.hip): X / out_h are split-format maps
 * [2][M][32] (lo plane x_lo / o_lo elements behind), W1 / W2 [2][k][C_out][C_in] fp16 planes (w_lo elements apart), three MFMAs
 * per product in the K order of the x3 implicit GEMM.  C = 32 only; advh_resblock_pair_x3_lds_bytes < 0: unsupported.       */
typedef struct advh_resblock_x3_desc {
    const void* X;
    const void* W1;
    const float* b1;
    const void* W2;
    const float* b2;
    void* out_h;
    int M, Wg, w0, w1, k, dil;
    float slope;
    int64_t x_lo, o_lo, w_lo;
} advh_resblock_x3_desc;
int advh_resblock_pair_x3_lds_bytes(int C, int k, int dil);
int advh_resblock_pair_x3(const advh_resblock_x3_desc* d, int C, advh_stream_t stream);
/* HiFi-GAN generator pieces on split-format maps [2][B][T+2*halo][C] (the Conv1d / ConvTranspose1d layers are advh_gemm_f16
 * launches with desc.split = 1; advh_halo_fill_f16 serves both planes when called with 2*B maps).  `pad` = inference padding. */
int advh_hifigan_pack_mel_split(const float* mel, void* out, int64_t out_lo, int B, int C, int T, int pad, int halo,
                                advh_stream_t stream);
int advh_hifigan_mrf_mix_split(const void* a, const void* b, const void* c, void* y, float slope, int64_t numel, int64_t lo,
                               advh_stream_t stream);
int advh_hifigan_conv_post_split(const void* x, int64_t x_lo, const float* w, float bias, float* wav, int B, int C, int T, int halo,
                                 int k, advh_stream_t stream);

/* ---- LDS line-tile convolution for narrow layers (C_in = C_out = C in {32, 64}) -------------------------------
 * out[m][co] = act(bias[co] + sum_t sum_ci W[t][co][ci] * X[m + toff[t]][ci]) (+ resid[m][co]) over the M rows of a
 * zero-haloed channels-last fp16 map; rows outside the window (h0..h1, w0..w1 of the (Hg, Wg) row grid) are
 * written as zeros, so the output map keeps a zero halo.  X, resid, out_h, out_h2 share one geometry [M][C].
 * Replaces the HiFi-GAN ResBlock1 Conv1d layers of the 64- and 32-channel stages (speechbrain HifiganGenerator via
 * hifigan.py:106-110, 180): toff[t] = (t - (k-1)/2) * dilation.  Weights stay resident in LDS (<= 160 KiB with the
 * double-buffered line buffer: advh_conv_taps_lds_bytes), HBM traffic is input once + output once.               */
typedef struct advh_taps_desc {
    const void* X;        /* fp16 [M][C]                                                          */
    const void* W;        /* fp16 [ntap][C_out][C_in]                                             */
    const float* bias;    /* [C] or NULL                                                          */
    const void* resid;    /* fp16 [M][C] or NULL (added after the activation)                     */
    void* out_h;          /* fp16 [M][C]                                                          */
    void* out_h2;         /* fp16 [M][C] or NULL: LeakyReLU(slope2) of the value stored in out_h   */
    int M, Hg, Wg, h0, h1, w0, w1;
    int ntap;
    int toff[16];
    int act;              /* ADVH_ACT_NONE | ADVH_ACT_LEAKY */
    float slope, slope2;
    int pre_act;          /* 1: LeakyReLU(pre_slope) is applied to X inside the line buffer (X is the raw map) */
    float pre_slope;
} advh_taps_desc;
/* The same layer in the fp32-class mode, C = 64: X, W, resid, out_h, out_h2 are split-format plane pairs (lo plane x_lo / w_lo / r_lo /
 * o_lo elements behind the hi plane; out_h and out_h2 share o_lo), three MFMAs per fragment pair (the arithmetic of the x3 GEMM).  The weights
 * stream tap by tap through a two-slot LDS ring (two planes of an 11-tap tensor do not fit), the line buffer is double-buffered;
 * advh_conv_taps_split_tile(C, ntap, span) = positions per tile, 0 if the layer does not fit.  pre_act is not supported (ADVH_EUNSUPPORTED).
 * Replaces the x3 implicit GEMM for the k = 7 / 11 ResBlock convolutions of HiFi-GAN's 64-channel stage (hifigan.py:106-110, 180).  */
int advh_conv_taps_split_tile(int C, int ntap, int span);
int advh_conv_taps_split(const advh_taps_desc* d, int C, int64_t x_lo, int64_t w_lo, int64_t r_lo, int64_t o_lo, advh_stream_t stream);
/* 2-D variant: 3x3 stride-1 "same" Conv2d, C_in = C_out = C in {32, 64}, on zero-haloed NHWC fp16 maps of ONE geometry
 * [B][H+2PH][W_+2PW][C] (PH, PW >= 1) -- the second convolution of the U-Net's 32- / 64-channel ConvBlocks
 * (addvisor.py:20-24 with BatchNorm folded).  16 x 16 output tiles with an 18 x 18 line-buffer patch; W [9][C_out][C_in]
 * (tap = kh*3 + kw); only interior positions of out_h are written (its halo must already be zero).                */
typedef struct advh_taps2d_desc {
    const void* X;
    const void* W;
    const float* bias;
    void* out_h;
    int B, H, W_, PH, PW;
    int act;              /* ADVH_ACT_NONE | ADVH_ACT_LEAKY */
    float slope;
} advh_taps2d_desc;
int advh_conv_taps2d_f16(const advh_taps2d_desc* d, int C, advh_stream_t stream);
/* Last decoder stage of the U-Net as ONE line-tile launch: up1 = ConvTranspose2d(64,32,(2,1),stride (2,1)) folded into
 * d1.block.0 = Conv2d(33,32,3,padding 1) + BatchNorm + LeakyReLU (addvisor.py:53-54,78-80).
 *   Xc  coarse map  [B][Hc+2PHc][W_+2PWc][64]  fp16, zero halo (PHc, PWc >= 1)              (y2)
 *   Xs  skip map    [B][2Hc+2PHs][W_+2PWs][8]  fp16 = (spectrogram, in-image indicator, 0 x6), zero halo (advh_unet_pack_x)
 *   W   fp16 [2 row parities][15 k-steps][32 rows][32]: per parity the composed weights in the K order
 *       (coarse tap t = 3 ti + tj: 64 channels) x 6, (fine tap kh*3 + kw: 8 channels) x 9, zero-padded 456 -> 480;
 *       row R of a k-step carries output channel 8 ((R>>2)&3) + 4 ((R>>4)&1) + (R&3)
 *   out_h [B][2Hc+2PHo][W_+2PWo][32] fp16, interior written (halo must already be zero); Hc % 8 == 0.            */
typedef struct advh_upconv_desc {
    const void* Xc;
    const void* Xs;
    const void* W;
    const float* bias;    /* [32] or NULL */
    void* out_h;
    int B, Hc, W_, PHc, PWc, PHs, PWs, PHo, PWo;
    int act;              /* ADVH_ACT_NONE | ADVH_ACT_LEAKY */
    float slope;
} advh_upconv_desc;
int advh_upconv21_tile_f16(const advh_upconv_desc* d, advh_stream_t stream);
int advh_upconv21_tile_lds_bytes(void);
/* e2.block.0 of the U-Net as a line-tile launch: Conv2d(32, 64, (5,3), stride (2,1), padding (2,1)) + folded BatchNorm +
 * LeakyReLU (addvisor.py:32).  X [B][2Ho+2PHi][W_+2PWi][32] fp16 zero-haloed (PHi >= 2, PWi >= 1);
 * W fp16 [15 taps = kh*3+kw][64 rows][32 ci], row R of a tap = output channel 32 (R>>5) + 8 ((R>>2)&3) + 4 ((R>>4)&1) + (R&3);
 * out_h [B][Ho+2PHo][W_+2PWo][64], interior written (halo must already be zero).                                      */
typedef struct advh_convs21_desc {
    const void* X;
    const void* W;
    const float* bias;    /* [64] or NULL */
    void* out_h;
    int B, Ho, W_, PHi, PWi, PHo, PWo;
    int act;              /* ADVH_ACT_NONE | ADVH_ACT_LEAKY */
    float slope;
} advh_convs21_desc;
int advh_conv53s21_tile_f16(const advh_convs21_desc* d, advh_stream_t stream);
int advh_conv53s21_tile_lds_bytes(void);
int advh_conv_taps_tile(int C, int ntap, int span);       /* positions per workgroup tile (128/192/256); 0 = does not fit */
int advh_conv_taps_lds_bytes(int C, int ntap, int span);  /* weights + two line buffers; -1 = does not fit           */
int advh_conv_taps_f16(const advh_taps_desc* d, int C, advh_stream_t stream);

/* ---- training step of the U-Net mask decoder (addvisor.py:12-84 under train_addvisor.py:364-378)
 * Maps are zero-haloed channels-last fp16 [B][H+2PH][W+2PW][C]; only interiors are read for statistics / written.
 * BatchNorm2d in training mode (batch statistics over B*H*W, eps 1e-5) around LeakyReLU(slope):
 *   advh_bn_stats     sums[0..C) = sum z, sums[C..2C) = sum z^2 (deterministic two-stage; partial: advh_bn_partial_count()*2*C floats)
 *   advh_bn_apply     a = lrelu(coef[c]*z + coef[C+c])                        coef = [scale | shift | mean | invstd], 4*C floats
 *   advh_bn_bwd_sums  with dy^ = g_a * lrelu'(scale*z+shift), z^ = (z-mean)*invstd: sums = [sum dy^ | sum dy^ z^]
 *   advh_bn_bwd_apply dz = coef_b[c]*(dy^ - coef_b[C+c] - z^*coef_b[2C+c]) written at dz + d_c0 + b*d_sB + h*d_sH + w*d_sW
 *                     (interior coordinates; a dense map or the zero-upsampled grid of a strided layer), coef_b = [k1 | m1 | m2].
 *   g_a (same geometry as z) is fp16 or, with g_f32 = 1, fp32: the backward subtracts g_a's per-channel mean, so the
 *   dgrad GEMM that produces it stores its fp32 accumulators (out_f).                                            */
typedef struct advh_map_geom { int B, H, W, C, PH, PW; } advh_map_geom;
int advh_bn_partial_count(void);
/* per-channel coefficients from the reduced sums: forward coef (and the nn.BatchNorm2d train-mode update of the running
 * buffers: momentum, unbiased variance, num_batches_tracked += 1; all three NULL to skip), backward coef_b and the
 * affine gradients dgamma = sum dy^ z^ * inv_scale, dbeta = sum dy^ * inv_scale (n = B*H*W).                       */
int advh_bn_coef(const float* sums, const float* gamma, const float* beta, int C, float n, float eps, float momentum,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float* coef, advh_stream_t stream);
int advh_bn_bwd_coef(const float* sums, const float* coef, int C, float n, float inv_scale, float* coef_b, float* dgamma,
                     float* dbeta, advh_stream_t stream);
int advh_bn_stats(const void* z, const advh_map_geom* g, float* partial, float* sums, advh_stream_t stream);
int advh_bn_apply(const void* z, const advh_map_geom* g, const float* coef, float slope, void* a, advh_stream_t stream);
int advh_bn_bwd_sums(const void* z, const void* g_a, int g_f32, const advh_map_geom* g, const float* coef, float slope,
                     float* partial, float* sums, advh_stream_t stream);
int advh_bn_bwd_apply(const void* z, const void* g_a, int g_f32, const advh_map_geom* g, const float* coef, const float* coef_b, float slope,
                      void* dz, int64_t d_sB, int64_t d_sH, int64_t d_sW, int64_t d_c0, advh_stream_t stream);
/* Operand transpose for the weight-gradient GEMM (reduction over positions needs position-major operands):
 * dst[(t*rpt + r0 + c)*ld + col0 + p] = src[b][PHs + y][PWs + x][c0 + c], p = (b*Hg + hg)*Wg + wg over the layer's common
 * grid, (y, x) = (sy*(hg-GH) + oy[t], sx*(wg-GW) + ox[t]); zero where (y, x) is outside the source interior.  fp16.  */
typedef struct advh_transpose_desc {
    int B, Hg, Wg, GH, GW, H, W;
    int Hs, Ws, PHs, PWs, Cs, c0, nC;
    int sy, sx, ntap, oy[16], ox[16];
    int64_t ld, col0;
    int rpt, r0;          /* dst row of (tap t, channel c) = t*rpt + r0 + c (r0 > 0: second source of a skip concatenation) */
} advh_transpose_desc;
int advh_transpose_gather(const void* src, void* dst, const advh_transpose_desc* d, advh_stream_t stream);
/* mask head backward (addvisor.py:57-60): dlogit = dmask*m*(1-m) (fp32 out), dy1[i][c] = scale*dlogit[i]*w32[c]
 * ([total][32], fp32 if dy_f32 else fp16). */
int advh_unet_head_bwd(const float* dmask, const float* mask, const float* w32, float scale, int64_t total, float* dlogit,
                       void* dy1, int dy_f32, advh_stream_t stream);
/* mask head weight / bias gradient: dw33[c] = sum_i dlogit[i]*y1[i][c] for c < 32, dw33[32] = sum_i dlogit[i]; y1 fp16
 * [total][32]; partial: advh_bn_partial_count()*64 floats; dw33: 64 floats (33 used).                               */
int advh_unet_head_wgrad(const float* dlogit, const void* y1, int64_t total, float* partial, float* dw33, advh_stream_t stream);
/* weight gradient of the 1-channel stem e1.block.0: dw[co][kh*3+kw] = sum dz[b][ho][w][co] * mag[b][2ho+kh-2][w+kw-1];
 * dz = dense fp16 map [B][H/2+2PH][W+2PW][32]; partial: advh_bn_partial_count()*480 floats; dw [32][15] fp32.     */
int advh_unet_stem_wgrad(const void* dz, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                         float* partial, float* dw, advh_stream_t stream);

/* fp32-class training step (train_addvisor.py:363-378 runs the decoder's forward and loss.backward() in fp32): the same
 * kernels on split-format maps [2][B][H+2PH][W+2PW][C] (hi + lo fp16 planes, csrc/device_math.h; `*_lo` = distance in
 * elements between the planes), so LeakyReLU takes the branch the fp32 reference takes and the weight / activation
 * gradients keep ~22 bits.  The convolutions, dgrads and split-K wgrads are advh_gemm_f16 launches with desc.split = 1;
 * advh_transpose_gather is called once per plane.  g_a is a split map here.                                              */
int advh_bn_stats_split(const void* z, int64_t z_lo, const advh_map_geom* g, float* partial, float* sums, advh_stream_t stream);
int advh_bn_apply_split(const void* z, int64_t z_lo, const advh_map_geom* g, const float* coef, float slope, void* a, int64_t a_lo,
                        advh_stream_t stream);
int advh_bn_bwd_sums_split(const void* z, int64_t z_lo, const void* g_a, int64_t g_lo, const advh_map_geom* g, const float* coef,
                           float slope, float* partial, float* sums, advh_stream_t stream);
int advh_bn_bwd_apply_split(const void* z, int64_t z_lo, const void* g_a, int64_t g_lo, const advh_map_geom* g, const float* coef,
                            const float* coef_b, float slope, void* dz, int64_t dz_lo, int64_t d_sB, int64_t d_sH, int64_t d_sW,
                            int64_t d_c0, advh_stream_t stream);
int advh_unet_head_bwd_split(const float* dmask, const float* mask, const float* w32, float scale, int64_t total, float* dlogit,
                             void* dy1, int64_t dy_lo, advh_stream_t stream);
int advh_unet_head_wgrad_split(const float* dlogit, const void* y1, int64_t y_lo, int64_t total, float* partial, float* dw33,
                               advh_stream_t stream);
int advh_unet_stem_wgrad_split(const void* dz, int64_t dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                               float* partial, float* dw, advh_stream_t stream);
/* Weight gradient of the one skip (magnitude) channel of d1.block.0 -- Conv2d(33, 32, 3, padding 1) on torch.cat([up1(y2), x], 1),
 * addvisor.py:57-60, 79: dw[co][kh*3+kw] = sum_p dz[p][co] * mag[b][h+kh-1][w+kw-1] over the H x W crop of mag [B][Fq][Tq]; dz [B][H+2PH][W+2PW][32]
 * (fp16, or a split pair with the lo plane dz_lo elements behind); partial: advh_bn_partial_count() * 288 floats; dw: 288 floats [32][9]. */
int advh_unet_skip_wgrad(const void* dz, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW, float* partial, float* dw,
                         advh_stream_t stream);
int advh_unet_skip_wgrad_split(const void* dz, int64_t dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                               float* partial, float* dw, advh_stream_t stream);

/* Weight gradient of a 3x3 stride-1 "same" Conv2d, C_in = C_out = C in {32, 64}, without transposed copies in HBM:
 * dw[kh*3+kw][co][ci] = sum_p dz[p][co] * x[p + (kh-1, kw-1)][ci].  X and DZ are zero-haloed channels-last fp16 maps of
 * the same interior [B][H][W_] with their own halos (>= 1); both MFMA operands are read from LDS tiles with
 * ds_read_b64_tr_b16 (k = position).  partial: advh_conv_wgrad2d_parts(C,B,H,W) * 9*C*C floats of scratch; dw: 9*C*C.  */
typedef struct advh_wgrad2d_desc {
    const void* X;
    const void* DZ;
    float* partial;
    int B, H, W_, PHx, PWx, PHz, PWz;
} advh_wgrad2d_desc;
int advh_conv_wgrad2d_parts(int C, int B, int H, int W);
int advh_conv_wgrad2d_f16(const advh_wgrad2d_desc* d, int C, float* dw, advh_stream_t stream);
/* The same weight gradient in the fp32-class mode, per channel-slice pair: X and DZ are split-format maps (lo plane x_lo / dz_lo elements behind
 * the hi plane) with Cx / Cz channels; the launch computes dw[kh*3+kw][co][ci] for co in [cz0, cz0 + CO), ci in [cx0, cx0 + CI), CI, CO in
 * {32, 64} -- three MFMAs per fragment pair (the arithmetic of the x3 GEMM), fp32 partials (advh_conv_wgrad2d_split_parts(...) * 9*CO*CI floats)
 * and the same fixed-order reduction; dw: 9*CO*CI floats.  A wider layer or one with concatenated sources (addvisor.py:63-75 torch.cat) is covered
 * slice pair by slice pair.  Replaces four operand transposes + a split-K GEMM per layer (train_addvisor.py:376 loss.backward() through
 * addvisor.py:20-24).  */
int advh_conv_wgrad2d_split_parts(int CI, int CO, int B, int H, int W);
int advh_conv_wgrad2d_split(const advh_wgrad2d_desc* d, int CI, int CO, int Cx, int cx0, int Cz, int cz0, int64_t x_lo, int64_t dz_lo,
                            float* dw, advh_stream_t stream);

/* One fused HiFi-GAN ResBlock1 step for the 32- / 64-channel stages (speechbrain HifiganGenerator via hifigan.py:106-110,
 * 180): out = x + conv2(lrelu(conv1(lrelu(x)))) on a zero-haloed channels-last fp16 map [M][C] (rows m with
 * w0 <= m % Wg < w1 are real samples; the others are written as zeros); conv1: k taps, dilation dil, conv2: k taps,
 * dilation 1, both "same"; W1 / W2 [k][C_out][C_in] fp16, b1 / b2 [C] fp32.  The intermediate map stays in LDS.  X and
 * out_h must be different maps.  ADVH_EUNSUPPORTED when both weight tensors + line buffers exceed 160 KiB
 * (advh_resblock_pair_lds_bytes): the caller then launches the two convolutions separately.                         */
typedef struct advh_resblock_desc {
    const void* X;
    const void* W1;
    const float* b1;
    const void* W2;
    const float* b2;
    void* out_h;
    int M, Wg, w0, w1, k, dil;
    float slope;
} advh_resblock_desc;
int advh_resblock_pair_lds_bytes(int C, int k, int dil);
int advh_resblock_pair_f16(const advh_resblock_desc* d, int C, advh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADDVISOR_HIP_H */
