// Framed real FFT (STFT) and masked inverse (ISTFT + overlap-add) for gfx950.
//
// Reference arithmetic: AudioProcessor.compute_stft / compute_invert_stft
// (audioprocessor.py:82-131), mask application loss_function.py:36-45 and LMAC_metrics.py:136-153.
//
// Both kernels are HBM-bound (1.89 MB / clip forward, 1.47 MB / clip per
// resynthesis).  The spectrogram layout is torch's [B][513][T] with t fastest, so a workgroup
// owns FB = 16 consecutive frames of one clip: every global access of the spectrogram is then a
// 64-byte (fp32) or 128-byte (complex64) run per frequency bin, and the 16 transforms live in
// LDS rows (fft512.h) where each of the 4 wavefronts runs whole 512-point transforms in place.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>
#include "addvisor_hip.h"
#include "common.h"
#include "fft512.h"

namespace advh {

__device__ cf g_twiddle[1024];   // e^{+2 pi i k / 1024}
#ifdef ADVH_STAMPS                // diagnostic build only (make EXTRA=-DADVH_STAMPS): s_memtime at the phase boundaries of one workgroup
__device__ long long g_stamps[16];
#define STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 5 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
static bool g_init_done[64] = {false};     // per device: twiddle table + dynamic-LDS limits live in each device's code object
static bool init_done_here() { int dev = 0; return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && g_init_done[dev]; }

constexpr int NFFT = 1024;
constexpr int NBIN = 513;
// frames per workgroup: template parameter FB (16: 64/128-byte runs per bin, 90 KB LDS = 1 workgroup per CU;
// 8: half the run length, 45 KB = 3 workgroups per CU).  g_stft_fb selects it (advh_set_option).
static int g_stft_fb = 8;
constexpr int THREADS = 512;           // 8 wavefronts: one frame each at FB = 8, so a workgroup's transforms run side by side

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Per-lane constants of a wavefront's transforms, fetched once per wavefront (not per frame): the twiddles of the two
// twiddled passes and of the real-transform glue (bins lane + 1 + 64 q).
struct LaneTw { cf p8[7], p64[7], glue[4]; };
__device__ __forceinline__ void load_lane_twiddles(int lane, LaneTw& t) {
    fft512_lane_twiddles<8>(g_twiddle, lane, t.p8);
    fft512_lane_twiddles<64>(g_twiddle, lane, t.p64);
#pragma unroll
    for (int q = 0; q < 4; ++q) t.glue[q] = g_twiddle[lane + 1 + 64 * q];
}

// FROM_REGS: v already holds the lane's 8 inputs z[lane + 64 r] (the forward builds them straight from global memory),
// so the first pass needs no LDS round trip; TO_REGS: the outputs y[lane + 64 r] stay in v (the inverse overlap-adds
// them from registers), so the last pass stores nothing.  Each saves 16 ds_write + 16 ds_read per lane and frame.
template <int DIR, bool FROM_REGS, bool TO_REGS>
__device__ __forceinline__ void fft512_wave(float* re, float* im, int lane, const LaneTw& t, cf (&v)[8]) {
    if (FROM_REGS) dft8<DIR>(v);
    else { fft512_pass_load_tw<DIR, 1>(re, im, lane, t.p8, v); wave_fence(); }     // pass 1 has no twiddles (the argument is ignored)
    fft512_pass_store<1>(re, im, lane, v);
    wave_fence();
    fft512_pass_load_tw<DIR, 8>(re, im, lane, t.p8, v);
    wave_fence();
    fft512_pass_store<8>(re, im, lane, v);
    wave_fence();
    fft512_pass_load_tw<DIR, 64>(re, im, lane, t.p64, v);
    if (!TO_REGS) {
        wave_fence();
        fft512_pass_store<64>(re, im, lane, v);
        wave_fence();
    }
}

// Row pitch of the LDS tile (words).  The spectrogram side of both kernels walks the tile TRANSPOSED -- lane -> (frame
// tl = idx % FB, bin k = idx / FB) -- so 32 consecutive lanes touch FB rows x 32/FB neighbouring bins: conflict-free iff
// pitch % 32 == 32 / FB (rows land 32/FB banks apart, the bins fill the gaps).  The plain FFT_ROW (530, % 32 = 18) gave
// 2-way conflicts on a third of those accesses (round-1 PMC: 32.7 % of the forward kernel's LDS cycles).
template <int FB> struct RowPitch { static constexpr int value = FFT_ROW + ((32 / FB - FFT_ROW % 32) + 32) % 32; };
static_assert(RowPitch<8>::value % 32 == 4 && RowPitch<16>::value % 32 == 2, "row pitch");

// LDS carve: re rows | im rows | scratch (samples for the forward, overlap-add accumulator for the inverse)
template <int FB>
__device__ __forceinline__ void carve(float* smem, float*& re, float*& im, float*& scratch) {
    re = smem;
    im = smem + FB * RowPitch<FB>::value;
    scratch = smem + 2 * FB * RowPitch<FB>::value;
}

// log1p / expm1 on the hardware log2 / exp2 units with Kahan's correction (log1p(x) = log(u) x / (u - 1), u = fl(1 + x);
// expm1(y) = (u - 1) y / log(u), u = fl(e^y)): relative error <= ~4 ulp (5e-7) on the range the mask application uses
// (x = |X| >= 0, y = m log1p|X| in [0, ~8]), against ~1 ulp at 3-4x the instructions for libm's log1pf / expm1f.
__device__ __forceinline__ float fast_log1p(float x) {
    const float u = 1.f + x, d = u - 1.f;
    return d == 0.f ? x : __logf(u) * (x * __builtin_amdgcn_rcpf(d));
}
__device__ __forceinline__ float fast_expm1(float y) {
    const float u = __expf(y), d = u - 1.f;
    if (d == 0.f) return y;
    return d * (y * __builtin_amdgcn_rcpf(__logf(u)));
}
// |X| of the mask application a' = g(m, |X|) as a FACTOR on X itself: X' = X * g / |X|  (= g e^{i angle X} without the
// atan2 / sincos round trip).  linear: g / |X| = m;  log1p: expm1(m log1p M) / M -> m as M -> 0.
__device__ __forceinline__ float mask_factor(float m, float M, int mode) {
    if (mode == ADVH_MASK_LINEAR) return m;
    if (M < 1e-12f) return m;
    // g / M = ((1 + M)^m - 1) / M on the hardware log2 / exp2 units: three transcendentals per bin (v_log, v_exp, v_rcp) instead
    // of the seven of expm1(m * log1p(M)) with Kahan's corrections.  The cancellation in (1 + M)^m - 1 costs RELATIVE accuracy of
    // the factor only where the factor is small, i.e. where the masked bin contributes little: the absolute error of the masked
    // bin X * g / M is <= ~1e-7 (1 + M) (one ulp of (1 + M)^m), against a waveform tolerance of 5e-6.
    return (__builtin_amdgcn_exp2f(m * __builtin_amdgcn_logf(1.f + M)) - 1.f) * __builtin_amdgcn_rcpf(M);
}

// ------------------------------------------------------------------------------------------ forward
// ADJ = 1 turns the kernel into the ADJOINT of the masked ISTFT (the backward of loss_function.py:36-47 from the
// resynthesised waveform to the mask): `wave` is then dL/d(resynthesised wave) [B][L]; it is divided by the
// overlap-add window envelope, zero-extended (not reflected) to the padded signal, framed with the synthesis
// window and transformed; with G = rfft of a frame, dL/dRe X_k = (c_k/N) Re G_k and dL/dIm X_k = (c_k/N) Im G_k
// (c_k = 1 for k = 0, N/2 where the C2R transform ignores the imaginary part, else 2).  The epilogue chains to the
// mask: X = a e^{i phi}, a = m M (mask-in) or (1-m) M (mask-out) [linear] / expm1(m log1p M) [log1p]; it reads
// `mag`, `phase` (and `X` = the mask for log1p) and writes dmask[b][k][t] for k < Fm, t < Tm.
struct AdjArgs { const float* mask; float* dmask; int Fm, Tm, mode, which; };

template <int FB, int ADJ>
__global__ __launch_bounds__(THREADS, 6) void stft_fwd_kernel(
    const float* __restrict__ wave, long wave_stride, int n_in, int L, int hop, int win,
    const float* __restrict__ window, float* __restrict__ X, float* __restrict__ mag,
    float* __restrict__ phase, int T, AdjArgs adj) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWP = RowPitch<FB>::value;
    float *re, *im, *smp;
    carve<FB>(smem, re, im, smp);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, tA = blockIdx.x * FB;
    const int left = (NFFT - win) / 2;
    const int span = (FB - 1) * hop + win;

    // 1. (ADJ only) stage the samples the frames touch, divided by the overlap-add envelope.  The plain forward has no
    // staging phase: every wavefront loads its frame's samples straight into the registers of the first FFT pass
    // (step 2) -- 16 independent loads per lane in flight at once; the staged version serialised them behind one
    // s_waitcnt per loop iteration and a workgroup barrier (5.3 of a workgroup's 22 kcycles, in-kernel stamps).
    STAMP(0);
    const float* w = wave + (long)b * wave_stride;
    if (ADJ)
    for (int i = tid; i < span; i += THREADS) {
        int src = tA * hop + left + i - NFFT / 2;
        float v = 0.f;
        if (ADJ) {
            if (src >= 0 && src < L && src < n_in) {
                const int pp = tA * hop + left + i;       // padded-signal coordinate
                const int R = (win + hop - 1) / hop, thi = (pp - left) / hop;
                float env = 0.f;
                for (int r = 0; r < R; ++r) {
                    int t = thi - r, j = pp - left - t * hop;
                    if (t >= 0 && t < T && j >= 0 && j < win) { float ww = window ? window[j] : 1.f; env += ww * ww; }
                }
                v = env > 1e-11f ? w[src] / env : 0.f;
            }
        } else {
            if (src < 0) src = -src;
            if (src >= L) src = 2 * (L - 1) - src;
            if (src >= 0 && src < n_in && src < L) v = w[src];
        }
        smp[i] = v;
    }
    if (ADJ) __syncthreads();
    STAMP(1);

    // 2. one wavefront per frame: build z[n] = x[2n] + i x[2n+1], FFT512, real-transform glue
    LaneTw ltw;
    load_lane_twiddles(lane, ltw);
    for (int f = wv; f < FB; f += THREADS / 64) {
        if (tA + f >= T) break;                       // wave-uniform
        float* rr = re + f * ROWP;
        float* ii = im + f * ROWP;
        const float* fs = smp + f * hop;
        cf zv[8];                                     // z[n] = x[2n] + i x[2n+1], n = lane + 64 q: the first pass's inputs
        if (ADJ) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = lane + 64 * q;
                const int j0 = 2 * n - left, j1 = j0 + 1;
                float a = 0.f, c = 0.f;
                if (j0 >= 0 && j0 < win) a = window ? fs[j0] * window[j0] : fs[j0];
                if (j1 >= 0 && j1 < win) c = window ? fs[j1] * window[j1] : fs[j1];
                zv[q] = cf{a, c};
            }
        } else {
            const int s0 = (tA + f) * hop + left - NFFT / 2;      // clip sample under window position 0 (before reflection)
            const int nmax = min(n_in, L);
            float xa[8], xc[8], wa[8], wc[8];
            // interior frames (all but the first two / last two of a clip): no reflection, no tail -- the window's samples are the
            // contiguous run w[s0 .. s0 + win), one aligned 8-byte load per (j0, j0 + 1) pair and no index arithmetic beyond the
            // window test (wave-uniform branch; ~200 of the kernel's ~800 VALU instructions per frame were reflect / clamp logic)
            const bool interior = s0 >= 0 && s0 + win <= nmax && !window && !(left & 1) && !(s0 & 1) && !(wave_stride & 1);
            if (interior) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j0 = 2 * (lane + 64 * q) - left;
                    const bool ok = j0 >= 0 && j0 < win;          // win even: j0 + 1 < win too
                    const float2 v = ok ? *reinterpret_cast<const float2*>(w + s0 + j0) : make_float2(0.f, 0.f);
                    zv[q] = cf{v.x, v.y};
                }
            } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {             // all loads first (predicated), arithmetic afterwards
                const int n = lane + 64 * q;
                const int j0 = 2 * n - left, j1 = j0 + 1;
                int a0 = s0 + j0, a1 = s0 + j1;
                if (a0 < 0) a0 = -a0;
                if (a0 >= L) a0 = 2 * (L - 1) - a0;
                if (a1 < 0) a1 = -a1;
                if (a1 >= L) a1 = 2 * (L - 1) - a1;
                const bool ok0 = j0 >= 0 && j0 < win && a0 >= 0 && a0 < nmax, ok1 = j1 >= 0 && j1 < win && a1 >= 0 && a1 < nmax;
                xa[q] = ok0 ? w[a0] : 0.f;
                xc[q] = ok1 ? w[a1] : 0.f;
                wa[q] = (window && ok0) ? window[j0] : 1.f;
                wc[q] = (window && ok1) ? window[j1] : 1.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) zv[q] = cf{xa[q] * wa[q], xc[q] * wc[q]};
            }
        }
        fft512_wave<-1, true, false>(rr, ii, lane, ltw, zv);
        cf A[4], Bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = lane + 1 + 64 * q;
            A[q] = cf{rr[fidx(k)], ii[fidx(k)]};
            Bv[q] = cf{rr[fidx(512 - k)], ii[fidx(512 - k)]};
        }
        cf Z0 = cf{rr[0], ii[0]};
        wave_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = lane + 1 + 64 * q;
            cf xk, xm;
            rfft_post_pair(A[q], Bv[q], ltw.glue[q], xk, xm);
            rr[fidx(k)] = xk.x; ii[fidx(k)] = xk.y;
            rr[fidx(512 - k)] = xm.x; ii[fidx(512 - k)] = xm.y;
        }
        if (lane == 0) {
            rr[0] = Z0.x + Z0.y; ii[0] = 0.f;
            rr[fidx(512)] = Z0.x - Z0.y; ii[fidx(512)] = 0.f;
        }
    }
    __syncthreads();
    STAMP(2);

    // 3. coalesced epilogue: t fastest (16 frames = one 64 B / 128 B run per bin)
    const int nvalid = min(FB, T - tA);
    for (int idx = tid; idx < NBIN * FB; idx += THREADS) {
        int tl = idx & (FB - 1), k = idx / FB;
        if (tl >= nvalid) continue;
        float xr = re[tl * ROWP + fidx(k)], xi = im[tl * ROWP + fidx(k)];
        long o = ((long)b * NBIN + k) * T + tA + tl;
        if (ADJ) {
            const int t = tA + tl;
            if (k >= adj.Fm || t >= adj.Tm) continue;
            const bool edge = k == 0 || k == NFFT / 2;
            const float sc = (edge ? 1.f : 2.f) / NFFT;
            float sn, cs;
            sincosf(phase[o], &sn, &cs);
            const float da = sc * (xr * cs + (edge ? 0.f : xi * sn));       // dL/da, a = |X| after masking
            const long om = ((long)b * adj.Fm + k) * adj.Tm + t;
            const float M = mag[o];
            float dm;
            if (adj.mode == ADVH_MASK_LINEAR) {
                dm = adj.which ? -M * da : M * da;
            } else {                                       // a = expm1(m' log1p M), m' = m or 1 - m
                float m = adj.mask[om];
                if (adj.which) m = 1.f - m;
                const float lg = log1pf(M);
                dm = lg * expf(m * lg) * da;
                if (adj.which) dm = -dm;
            }
            adj.dmask[om] = dm;
            continue;
        }
        if (X) reinterpret_cast<float2*>(X)[o] = make_float2(xr, xi);
        if (mag) mag[o] = __builtin_sqrtf(fmaf(xr, xr, xi * xi));      // |X| <= 1024 max|x|: no overflow to guard against (hypotf's job)
        if (phase) phase[o] = atan2f(xi, xr);
    }
    STAMP(3);
}

// ------------------------------------------------------------------------------------------ inverse
// SRC 3: complex64 spectrogram X (in `mag`) + mask: X' = X * g(m, |X|) / |X| -- the mask application of SRC 0 without the
// polar round trip (no atan2 in the forward, no sincos here); what the explanation pipeline runs.
// SRC 0 / 3 with both outputs requested: grid z = 2, one workgroup per branch (the second branch's tile reads hit L2).
// SRC 0: mag/phase (+ optional mask, mode), SRC 1: complex64 spectrogram, SRC 2: band swap of two complex64
// spectrograms (hifigan.py:208-222, train_logReg_swapping.py:70-81): grid z = band, bins [Fm + z*Tm, Fm + (z+1)*Tm)
// come from `phase` (the vocoded signal), all others from `mag` (the original); output z at out0 + z * out1_stride.
template <int SRC, int FB>
__global__ __launch_bounds__(THREADS, 4) void istft_kernel(
    const float* __restrict__ mag, const float* __restrict__ phase, const float* __restrict__ mask,
    int Fm, int Tm, int mode, int which0, float* __restrict__ out0, float* __restrict__ out1,
    long wave_stride, int T, int L, int hop, int win, int R, const float* __restrict__ window, long zstride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWP = RowPitch<FB>::value;
    float *re, *im, *unused_scratch;
    carve<FB>(smem, re, im, unused_scratch);
    (void)unused_scratch;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, g = blockIdx.x;
    const int left = (NFFT - win) / 2;
    const int S = FB - R + 1;                         // complete hop-segments this workgroup emits
    const int tA = g * S - (R - 1);                   // first frame it transforms (may be < 0)
    // mask-in / mask-out are two workgroups (grid z): round 2 first ran them as two passes of one workgroup over tile values
    // kept in registers (one HBM read of the spectrogram), but the 27 live tile registers pushed the kernel to 191 VGPRs = ONE
    // workgroup per CU; as separate workgroups (the second read comes from L2) it fits 128 VGPRs = two per CU and is faster.
    const int pass = blockIdx.z;
    // tile loads: element idx = tid + it * THREADS -> (frame tl = idx % FB, bin k = idx / FB); SRC 0: (|X|, angle X),
    // SRC 1 / 2 / 3: complex X; ldm = the mask value (0 outside the Fm x Tm crop: SURVEY.md D2/D3)
    constexpr int NIT = (NBIN * FB + THREADS - 1) / THREADS;
    float2 ld0[NIT];
    float ldm[NIT];
    bool valid[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * THREADS;
        const int tl = idx & (FB - 1), k = idx / FB, t = tA + tl;
        valid[it] = idx < NBIN * FB && t >= 0 && t < T;
        ld0[it] = make_float2(0.f, 0.f);
        ldm[it] = 0.f;
        if (valid[it]) {
            const long o = ((long)b * NBIN + k) * T + t;
            if (SRC == 0) ld0[it] = make_float2(mag[o], phase[o]);
            else if (SRC == 2) {
                const int lo = Fm + (int)blockIdx.z * Tm;
                ld0[it] = reinterpret_cast<const float2*>((k >= lo && k < lo + Tm) ? phase : mag)[o];
            } else ld0[it] = reinterpret_cast<const float2*>(mag)[o];
            if ((SRC == 0 || SRC == 3) && mode != ADVH_MASK_NONE && k < Fm && t < Tm) ldm[it] = mask[((long)b * Fm + k) * Tm + t];
        }
    }
    const int which = which0 + (SRC == 2 ? 0 : pass);  // 0: mask-in, 1: mask-out
    float* out = (SRC == 2 ? out0 + (long)blockIdx.z * zstride : (pass == 0 ? out0 : out1)) + (long)b * wave_stride;
    if (pass == 0) STAMP(4);

    // 1. the 513 x FB tile: mask application, polar -> cartesian, into the LDS rows.  The global loads were issued before
    // the pass loop (all of a thread's elements in flight at once) and serve both passes from registers.
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * THREADS;
        if (idx >= NBIN * FB) break;
        const int tl = idx & (FB - 1), k = idx / FB;
        float xr = ld0[it].x, xi = ld0[it].y;
        if (SRC == 3) {
            float m = ldm[it];
            if (which == 1) m = 1.f - m;
            const float f = valid[it] ? mask_factor(m, __builtin_sqrtf(fmaf(xr, xr, xi * xi)), mode) : 0.f;
            xr *= f; xi *= f;
        } else if (SRC == 0) {
            float a = ld0[it].x;
            const float ph = ld0[it].y;
            if (mode != ADVH_MASK_NONE) {
                float m = ldm[it];
                if (which == 1) m = 1.f - m;
                a = (mode == ADVH_MASK_LINEAR) ? m * a : fast_expm1(m * fast_log1p(a));
            }
            float sn, cs;
            sincosf(ph, &sn, &cs);
            xr = valid[it] ? a * cs : 0.f; xi = valid[it] ? a * sn : 0.f;
        }
        re[tl * ROWP + fidx(k)] = xr;
        im[tl * ROWP + fidx(k)] = xi;
    }
    __syncthreads();
    if (pass == 0) STAMP(5);
    LaneTw ltw;                                       // fetched after the tile registers have died
    load_lane_twiddles(lane, ltw);

    // 2. one wavefront per frame: Hermitian glue, inverse FFT512, then the frame's windowed time samples go back into the frame's
    // OWN LDS rows with plain 8-byte stores (sample j at re-row word j for j < ROWP, else im-row word j - ROWP: the rows are dead once
    // the last pass has loaded them, and a wavefront's DS operations complete in order).  Round 2 overlap-added with LDS float
    // atomics (two addends per slot): the PMC pass of round 3 (profiles/r03_stft_pmc.txt) showed 114 LDS-array cycles per atomic
    // instruction -- 127 us of LDS time per launch, the whole kernel.  The sum over the R overlapping frames now happens in step 3.
    static_assert(2 * RowPitch<FB>::value >= NFFT, "a frame's time samples must fit its two LDS rows");
    for (int f = wv; f < FB; f += THREADS / 64) {
        int t = tA + f;
        if (t < 0 || t >= T) continue;                // wave-uniform
        float* rr = re + f * ROWP;
        float* ii = im + f * ROWP;
        cf A[4], Bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = lane + 1 + 64 * q;
            A[q] = cf{rr[fidx(k)], ii[fidx(k)]};
            Bv[q] = cf{rr[fidx(512 - k)], ii[fidx(512 - k)]};
        }
        float x0 = rr[0], xn = rr[fidx(512)];        // C2R semantics: imaginary parts of DC / Nyquist ignored
        wave_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = lane + 1 + 64 * q;
            cf zk, zm;
            irfft_pre_pair(A[q], Bv[q], ltw.glue[q], zk, zm);
            rr[fidx(k)] = zk.x; ii[fidx(k)] = zk.y;
            rr[fidx(512 - k)] = zm.x; ii[fidx(512 - k)] = zm.y;
        }
        if (lane == 0) { rr[0] = 0.5f * (x0 + xn); ii[0] = 0.5f * (x0 - xn); }
        wave_fence();
        cf yv[8];
        fft512_wave<+1, false, true>(rr, ii, lane, ltw, yv);          // y[lane + 64 q] stays in registers
        wave_fence();                                                  // every lane's last-pass loads precede the stores below
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = lane + 64 * q;
            const int j0 = 2 * n - left, j1 = j0 + 1;
            float a = yv[q].x * (1.f / 512.f), c = yv[q].y * (1.f / 512.f);
            if (!(left & 1)) {                        // j0 even (win % 4 == 0, the reference's 644 and Hann-1024): one aligned 8-byte store per pair
                if (j0 < 0 || j0 >= win) continue;    // win is even: j1 < win too
                if (window) { a *= window[j0]; c *= window[j1]; }
                *reinterpret_cast<float2*>(j0 < ROWP ? rr + j0 : ii + (j0 - ROWP)) = make_float2(a, c);
            } else {
                if (j0 >= 0 && j0 < win) *(j0 < ROWP ? rr + j0 : ii + (j0 - ROWP)) = window ? a * window[j0] : a;
                if (j1 >= 0 && j1 < win) *(j1 < ROWP ? rr + j1 : ii + (j1 - ROWP)) = window ? c * window[j1] : c;
            }
        }
    }
    __syncthreads();
    if (pass == 0) STAMP(6);

    // 3. emit the S complete hop-segments: sum of the R overlapping frames' samples, divided by the window envelope, trimmed to [0, L)
    // Segment s (s = 0 .. S-1) holds the hop samples u = 0 .. hop-1 at padded coordinate p = (tA + R - 1 + s) * hop + left + u; exactly
    // the frames t = tA + R - 1 + s - r, r = 0 .. R-1, cover it, each with its sample j = u + r * hop: no division per sample.
    for (int sgm = 0; sgm < S; ++sgm) {
        const int thi = tA + R - 1 + sgm;
        const int n0 = thi * hop + left - NFFT / 2;   // clip sample of u = 0
        for (int u = tid; u < hop; u += THREADS) {
            const int n = n0 + u;
            if (n < 0 || n >= L) continue;
            float env = 0.f, sum = 0.f;
            for (int r = 0; r < R; ++r) {
                const int t = thi - r, j = u + r * hop;
                if (t >= 0 && t < T && j < win) {
                    const float ww = window ? window[j] : 1.f;
                    env += ww * ww;
                    const int f = t - tA;             // 0 <= f < FB
                    sum += j < ROWP ? re[f * ROWP + j] : im[f * ROWP + j - ROWP];
                }
            }
            out[n] = env > 1e-11f ? sum / env : 0.f;
        }
    }
    if (pass == 0) STAMP(7);
}

// the inverse needs the FFT rows only (the overlap-add reuses them): 35 KB at FB = 8 -> four workgroups per CU
static size_t lds_rows_bytes(int FB) { return sizeof(float) * 2 * FB * (FB == 8 ? RowPitch<8>::value : RowPitch<16>::value); }
static size_t lds_bytes(int FB, int hop, int win) {
    const int pitch = FB == 8 ? RowPitch<8>::value : RowPitch<16>::value;
    return sizeof(float) * (2 * FB * pitch + (FB - 1) * hop + win);
}

}  // namespace advh

using namespace advh;

extern "C" const char* advh_version(void) { return "addvisor_hip 0.1 (gfx950, wave64)"; }

#ifdef ADVH_STAMPS
extern "C" int advh_debug_stamps(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 16) == hipSuccess ? ADVH_OK : ADVH_ELAUNCH; }
#endif

static int* g_split_flag_host = nullptr;       // hipHostMallocMapped: device-written, host-read without a synchronisation

extern "C" int advh_set_option(const char* name, int value) {
    if (!name) return ADVH_EINVAL;
    if (!strcmp(name, "stft_frames_per_workgroup")) {
        if (value != 8 && value != 16) return ADVH_EINVAL;
        g_stft_fb = value;
        return ADVH_OK;
    }
    if (!strcmp(name, "attention_bwd_mfma_f32")) {          // 1: head dims <= 64 on the fp32 matrix instruction too (the pre-x3 kernel; A/B runs)
        g_att_bwd_force_f32 = value != 0;
        return ADVH_OK;
    }
    return ADVH_EINVAL;
}

extern "C" int advh_init(void) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ADVH_ELAUNCH;
    if (g_init_done[dev]) return ADVH_OK;
    cf host[1024];
    for (int k = 0; k < 1024; ++k) {
        host[k].x = (float)cos(2.0 * M_PI * k / 1024.0);
        host[k].y = (float)sin(2.0 * M_PI * k / 1024.0);
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_twiddle), host, sizeof(host)) != hipSuccess) return ADVH_ELAUNCH;
    const int maxlds = 160 * 1024;
    const void* big[] = {(const void*)stft_fwd_kernel<16, 0>, (const void*)stft_fwd_kernel<8, 0>, (const void*)stft_fwd_kernel<16, 1>,
                         (const void*)stft_fwd_kernel<8, 1>, (const void*)istft_kernel<0, 16>,
                         (const void*)istft_kernel<2, 16>, (const void*)istft_kernel<2, 8>,
                         (const void*)istft_kernel<1, 16>, (const void*)istft_kernel<0, 8>, (const void*)istft_kernel<1, 8>,
                         (const void*)istft_kernel<3, 16>, (const void*)istft_kernel<3, 8>};
    for (const void* f : big)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess) return ADVH_ELAUNCH;
    int rc = advh_init_rest();
    if (rc != ADVH_OK) return rc;
    rc = advh_init_attention();
    if (rc != ADVH_OK) return rc;
    // the split format's sticky range flag: ONE host-mapped word per process, written by the kernels of every device
    if (!g_split_flag_host) {
        if (hipHostMalloc((void**)&g_split_flag_host, sizeof(int), hipHostMallocMapped) != hipSuccess) return ADVH_ELAUNCH;
        *g_split_flag_host = 0;
    }
    int* dflag = nullptr;
    if (hipHostGetDevicePointer((void**)&dflag, g_split_flag_host, 0) != hipSuccess) return ADVH_ELAUNCH;
    int (*const setters[])(int*) = {advh_split_flag_attention, advh_split_flag_attention_bwd_f32, advh_split_flag_attention_bwd_x3, advh_split_flag_backward, advh_split_flag_conv_taps, advh_split_flag_frontend, advh_split_flag_frontend_bwd, advh_split_flag_gemm, advh_split_flag_hifigan, advh_split_flag_rowops, advh_split_flag_unet_misc, advh_split_flag_unet_train, advh_split_flag_resblock_pair_x3};
    for (auto set : setters)
        if ((rc = set(dflag)) != ADVH_OK) return rc;
    g_init_done[dev] = true;
    return ADVH_OK;
}

extern "C" int advh_split_overflow(int reset) {
    if (!g_split_flag_host) return 0;
    const int v = *(volatile int*)g_split_flag_host;
    if (reset) *(volatile int*)g_split_flag_host = 0;
    return v != 0;
}

static int check_frame_args(int B, int T, int L, int hop, int win) {
    if (!init_done_here()) return ADVH_ENOTINIT;
    if (B <= 0 || L <= 0 || hop <= 0 || win <= 0 || win > NFFT || (win & 1) || T != 1 + L / hop) return ADVH_EINVAL;
    if (L <= NFFT / 2) return ADVH_EINVAL;           // reflect padding needs L > n_fft/2
    if (lds_bytes(g_stft_fb, hop, win) > 160 * 1024) return ADVH_EUNSUPPORTED;
    return ADVH_OK;
}

extern "C" int advh_stft_forward(const float* wave, int64_t wave_stride, int n_in, int B, int L, int hop, int win,
                                 const float* window, float* X, float* mag, float* phase, int T,
                                 advh_stream_t stream) {
    int rc = check_frame_args(B, T, L, hop, win);
    if (rc) return rc;
    if (!wave || n_in <= 0 || wave_stride < (n_in < L ? n_in : L)) return ADVH_EINVAL;
    const int FB = g_stft_fb;
    dim3 grid((T + FB - 1) / FB, B);
    const AdjArgs none = {nullptr, nullptr, 0, 0, 0, 0};
    if (FB == 8)
        hipLaunchKernelGGL((stft_fwd_kernel<8, 0>), grid, dim3(THREADS), lds_rows_bytes(8), (hipStream_t)stream, wave,     // the plain forward stages no samples: rows only
                           (long)wave_stride, n_in, L, hop, win, window, X, mag, phase, T, none);
    else
        hipLaunchKernelGGL((stft_fwd_kernel<16, 0>), grid, dim3(THREADS), lds_rows_bytes(16), (hipStream_t)stream, wave,
                           (long)wave_stride, n_in, L, hop, win, window, X, mag, phase, T, none);
    return hipGetLastError() == hipSuccess ? ADVH_OK : ADVH_ELAUNCH;
}

extern "C" int advh_istft_masked_bwd(const float* g_wave, int64_t g_stride, const float* mag, const float* phase,
                                     const float* mask, int Fm, int Tm, int mode, int which, float* dmask, int B, int T,
                                     int L, int hop, int win, const float* window, advh_stream_t stream) {
    int rc = check_frame_args(B, T, L, hop, win);
    if (rc) return rc;
    if (!g_wave || !mag || !phase || !dmask || g_stride < L || Fm <= 0 || Tm <= 0 || Fm > NBIN || Tm > T) return ADVH_EINVAL;
    if (mode != ADVH_MASK_LINEAR && mode != ADVH_MASK_LOG1P) return ADVH_EINVAL;
    if (mode == ADVH_MASK_LOG1P && !mask) return ADVH_EINVAL;
    if (which != 0 && which != 1) return ADVH_EINVAL;
    const int FB = g_stft_fb;
    dim3 grid((Tm + FB - 1) / FB, B);
    const AdjArgs adj = {mask, dmask, Fm, Tm, mode, which};
    float* magp = const_cast<float*>(mag);
    float* php = const_cast<float*>(phase);
    if (FB == 8)
        hipLaunchKernelGGL((stft_fwd_kernel<8, 1>), grid, dim3(THREADS), lds_bytes(8, hop, win), (hipStream_t)stream, g_wave,
                           (long)g_stride, L, L, hop, win, window, (float*)nullptr, magp, php, T, adj);
    else
        hipLaunchKernelGGL((stft_fwd_kernel<16, 1>), grid, dim3(THREADS), lds_bytes(16, hop, win), (hipStream_t)stream, g_wave,
                           (long)g_stride, L, L, hop, win, window, (float*)nullptr, magp, php, T, adj);
    return hipGetLastError() == hipSuccess ? ADVH_OK : ADVH_ELAUNCH;
}

static int launch_istft(int src, const float* a, const float* ph, const float* mask, int Fm, int Tm, int mode,
                        float* o0, float* o1, int64_t ws, int B, int T, int L, int hop, int win,
                        const float* window, advh_stream_t stream, int nbands = 0, int64_t zstride = 0) {
    int rc = check_frame_args(B, T, L, hop, win);
    if (rc) return rc;
    const int R = (win + hop - 1) / hop;
    const int FB = g_stft_fb;
    if (R > FB / 2) return ADVH_EUNSUPPORTED;
    const int S = FB - R + 1, left = (NFFT - win) / 2;
    int which0 = 0, nz = 2;
    float *p0 = o0, *p1 = o1;
    if (!o0 && !o1) return ADVH_EINVAL;
    if (!o0) { which0 = 1; p0 = o1; nz = 1; }
    else if (!o1) { nz = 1; }
    if (src == 1) p1 = nullptr;
    if (src == 2) nz = nbands;
    else if (nz == 1) p1 = nullptr;
    const int nG = (NFFT / 2 + L - left + S * hop - 1) / (S * hop);
    dim3 grid(nG, B, nz);
#define ISTFT_LAUNCH(SRC_, FB_)                                                                                          \
    hipLaunchKernelGGL((istft_kernel<SRC_, FB_>), grid, dim3(THREADS), lds_rows_bytes(FB_), (hipStream_t)stream, a, ph,  \
                       mask, Fm, Tm, mode, which0, p0, p1, (long)ws, T, L, hop, win, R, window, (long)zstride)
    if (src == 0) { if (FB == 8) ISTFT_LAUNCH(0, 8); else ISTFT_LAUNCH(0, 16); }
    else if (src == 1) { if (FB == 8) ISTFT_LAUNCH(1, 8); else ISTFT_LAUNCH(1, 16); }
    else if (src == 3) { if (FB == 8) ISTFT_LAUNCH(3, 8); else ISTFT_LAUNCH(3, 16); }
    else { if (FB == 8) ISTFT_LAUNCH(2, 8); else ISTFT_LAUNCH(2, 16); }
#undef ISTFT_LAUNCH
    return hipGetLastError() == hipSuccess ? ADVH_OK : ADVH_ELAUNCH;
}

extern "C" int advh_istft_masked(const float* mag, const float* phase, const float* mask, int Fm, int Tm, int mode,
                                 float* wave_in, float* wave_out, int64_t wave_stride, int B, int T, int L, int hop,
                                 int win, const float* window, advh_stream_t stream) {
    if (!mag || !phase || wave_stride < L) return ADVH_EINVAL;
    if (mode != ADVH_MASK_NONE && (!mask || Fm <= 0 || Tm <= 0 || Fm > NBIN || Tm > T)) return ADVH_EINVAL;
    if (mode < 0 || mode > ADVH_MASK_LOG1P) return ADVH_EINVAL;
    if (mode == ADVH_MASK_NONE && wave_out) return ADVH_EINVAL;
    return launch_istft(0, mag, phase, mask, Fm, Tm, mode, wave_in, wave_out, wave_stride, B, T, L, hop, win, window, stream);
}

extern "C" int advh_istft_masked_c64(const float* spec, const float* mask, int Fm, int Tm, int mode, float* wave_in, float* wave_out,
                                     int64_t wave_stride, int B, int T, int L, int hop, int win, const float* window,
                                     advh_stream_t stream) {
    if (!spec || !mask || wave_stride < L || Fm <= 0 || Tm <= 0 || Fm > NBIN || Tm > T) return ADVH_EINVAL;
    if (mode != ADVH_MASK_LINEAR && mode != ADVH_MASK_LOG1P) return ADVH_EINVAL;
    return launch_istft(3, spec, nullptr, mask, Fm, Tm, mode, wave_in, wave_out, wave_stride, B, T, L, hop, win, window, stream);
}

extern "C" int advh_istft_bandswap(const float* spec_a, const float* spec_b, int k0, int kw, int nbands, float* waves,
                                   int64_t wave_stride, int64_t band_stride, int B, int T, int L, int hop, int win,
                                   const float* window, advh_stream_t stream) {
    if (!spec_a || !spec_b || !waves || wave_stride < L || nbands <= 0 || nbands > 65535 || k0 < 0 || kw <= 0 ||
        k0 + (long)nbands * kw > NBIN || band_stride < (int64_t)B * wave_stride)
        return ADVH_EINVAL;
    return launch_istft(2, spec_a, spec_b, nullptr, k0, kw, ADVH_MASK_NONE, waves, nullptr, wave_stride, B, T, L, hop, win, window,
                        stream, nbands, band_stride);
}

extern "C" int advh_istft_c64(const float* spec, float* wave, int64_t wave_stride, int B, int T, int L, int hop, int win,
                              const float* window, advh_stream_t stream) {
    if (!spec || !wave || wave_stride < L) return ADVH_EINVAL;
    return launch_istft(1, spec, nullptr, nullptr, 0, 0, ADVH_MASK_NONE, wave, nullptr, wave_stride, B, T, L, hop, win, window, stream);
}
