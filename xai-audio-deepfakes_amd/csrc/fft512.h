// 512-point complex FFT held in an LDS row, one wavefront (64 lanes) per transform.
//
// Stockham auto-sort, three radix-8 passes (512 = 8*8*8).  In every pass lane j loads its 8
// points (stride 64: conflict-free), applies the pass twiddles, does a radix-8 butterfly in
// registers and stores to the auto-sort positions.  All 64 lanes' loads are issued before any
// lane's stores (one wavefront executes its DS instructions in order), so ONE row is enough:
// the transform is in place.  Rows are planar (re[], im[]) with one pad word every 32 words so the
// stride-8 / stride-64 stores of the passes spread over the 32 LDS banks.
//
// The 1024-point real transforms of the STFT / ISTFT (audioprocessor.py:102-108, 123-129 in the
// reference: torch.stft / torch.istft with n_fft=1024) are built on it with the usual
// even/odd split: z[n] = x[2n] + i x[2n+1].
//
// The functions are HD so tests/test_host_fft.cpp can run them lane by lane on the CPU.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define ADVH_HD __host__ __device__ __forceinline__
#else
#define ADVH_HD inline
#endif

namespace advh {

// A complex number is a 2-lane fp32 vector: add / sub are ONE packed instruction (v_pk_add_f32), a complex product two
// (v_pk_mul_f32 + v_pk_fma_f32: a.x * (b.x, b.y) + a.y * (-b.y, b.x); the broadcasts, the swap and the sign are operand
// modifiers of the packed encoding) -- gfx950 issues packed fp32 at the scalar-fp32 instruction rate, and the STFT / ISTFT
// kernels are VALU-issue-bound.
typedef float cf __attribute__((ext_vector_type(2)));

ADVH_HD cf cmul(cf a, cf b) { return cf{a.x, a.x} * b + cf{a.y, a.y} * cf{-b.y, b.x}; }
ADVH_HD cf cadd(cf a, cf b) { return a + b; }
ADVH_HD cf csub(cf a, cf b) { return a - b; }
ADVH_HD cf cconj(cf a) { return cf{a.x, -a.y}; }

// physical index of logical element i in a padded row
ADVH_HD int fidx(int i) { return i + (i >> 5); }
constexpr int FFT_ROW = 513 + 17;   // 513 logical slots (bin 512 included) + pads, rounded to even

// multiply by e^{DIR * i*pi/2}: DIR=-1 -> -i, DIR=+1 -> +i
template <int DIR> ADVH_HD cf rot90(cf a) { return DIR < 0 ? cf{a.y, -a.x} : cf{-a.y, a.x}; }

template <int DIR> ADVH_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = rot90<DIR>(csub(a1, a3));
    a0 = cadd(s0, s1); a2 = csub(s0, s1); a1 = cadd(d0, d1); a3 = csub(d0, d1);
}

// in-register DFT of size 8: v[r] <- sum_q v[q] e^{DIR*2*pi*i*q*r/8}
template <int DIR> ADVH_HD void dft8(cf (&v)[8]) {
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<DIR>(e0, e1, e2, e3);
    dft4<DIR>(o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    // w8^r, r = 1..3
    cf w1 = cf{h, DIR * h}, w3 = cf{-h, DIR * h};
    o1 = cmul(o1, w1);
    o2 = rot90<DIR>(o2);
    o3 = cmul(o3, w3);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// tw[k] = (cos(2*pi*k/1024), sin(2*pi*k/1024)), k in [0,1024).
// The pass twiddles of a lane depend on the lane only, not on the data: fft512_lane_twiddles fetches the 7 of pass NS
// once (a wavefront transforms several frames; per-frame table loads from global memory were the longest dependent
// chain of the STFT kernels), fft512_pass_load_tw consumes them.
template <int NS> ADVH_HD void fft512_lane_twiddles(const cf* tw, int lane, cf (&twr)[7]) {
    const int k = lane % NS;
#pragma unroll
    for (int r = 1; r < 8; ++r) twr[r - 1] = tw[(2 * r * k * (64 / NS)) & 1023];
}

template <int DIR, int NS> ADVH_HD void fft512_pass_load_tw(const float* re, const float* im, int lane, const cf (&twr)[7],
                                                            cf (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) { int p = fidx(lane + 64 * r); v[r] = cf{re[p], im[p]}; }
    if (NS > 1) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            cf w = twr[r - 1];
            if (DIR < 0) w.y = -w.y;
            v[r] = cmul(v[r], w);
        }
    }
    dft8<DIR>(v);
}

template <int DIR, int NS> ADVH_HD void fft512_pass_load(const float* re, const float* im, int lane,
                                                         const cf* tw, cf (&v)[8]) {
    cf twr[7];
    fft512_lane_twiddles<NS>(tw, lane, twr);
    fft512_pass_load_tw<DIR, NS>(re, im, lane, twr, v);
}

template <int NS> ADVH_HD void fft512_pass_store(float* re, float* im, int lane, const cf (&v)[8]) {
    const int k = lane % NS, j0 = (lane / NS) * NS * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) { int p = fidx(j0 + r * NS); re[p] = v[r].x; im[p] = v[r].y; }
}

// ---- real-transform glue, one conjugate pair (k, 512-k), k in [1,256] ------------------------
// forward: Z = FFT512(z) -> X[k], X[512-k] of the 1024-point real transform
ADVH_HD void rfft_post_pair(cf A, cf B, cf wk /* tw[k] */, cf& Xk, cf& Xm) {
    // w = e^{-2 pi i k/1024} = conj(tw[k]);  X[k] = (A + B*)/2 + w (A - B*)/(2i)
    cf Bc = cconj(B), Ac = cconj(A);
    cf e = cf{0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y)};
    cf d = csub(A, Bc);                       // (A - B*)
    cf o = cf{0.5f * d.y, -0.5f * d.x};       // d / (2i)
    cf w = cconj(wk);
    Xk = cadd(e, cmul(w, o));
    cf e2 = cf{0.5f * (B.x + Ac.x), 0.5f * (B.y + Ac.y)};
    cf d2 = csub(B, Ac);
    cf o2 = cf{0.5f * d2.y, -0.5f * d2.x};
    cf wm = cf{-wk.x, -wk.y};                 // e^{-2 pi i (512-k)/1024} = -conj(w) = -tw[k]
    Xm = cadd(e2, cmul(wm, o2));
}

// inverse: X (Hermitian half) -> Z[k], Z[512-k] such that IFFT512(Z)/512 = x[2n] + i x[2n+1]
ADVH_HD void irfft_pre_pair(cf A, cf B, cf wk /* tw[k] = e^{+2 pi i k/1024} */, cf& Zk, cf& Zm) {
    cf Bc = cconj(B), Ac = cconj(A);
    cf e = cf{0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y)};
    cf d = cf{0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y)};
    cf o = cmul(wk, d);
    Zk = cf{e.x - o.y, e.y + o.x};            // e + i*o
    cf e2 = cf{0.5f * (B.x + Ac.x), 0.5f * (B.y + Ac.y)};
    cf d2 = cf{0.5f * (B.x - Ac.x), 0.5f * (B.y - Ac.y)};
    cf wm = cf{-wk.x, wk.y};                  // e^{2 pi i (512-k)/1024} = -conj(tw[k])
    cf o2 = cmul(wm, d2);
    Zm = cf{e2.x - o2.y, e2.y + o2.x};
}

}  // namespace advh
