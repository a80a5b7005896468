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

/* Plain ISTFT of a complex64 spectrogram [B][513][T][2] (audioprocessor.py:117-131). */
int advh_istft_c64(const float* spec, float* wave, int64_t wave_stride, int B, int T, int L, int hop, int win,
                   const float* window, advh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADDVISOR_HIP_H */
