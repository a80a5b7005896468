// Experiment: does a FIFTH co-resident workgroup per CU (160 KB of LDS landing zone instead of 128 KB) raise the rate of
// the 128x128x64 single-buffered DMA GEMM?  Stripped plain-GEMM copy of gemm_f16_kernel's main loop (affine rows, no
// gather table, fp16 store) built for 4 and for 5 workgroups per CU.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/gemm_occ5.hip -o /tmp/gemm_occ5 && /tmp/gemm_occ5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int WPE>
__global__ __launch_bounds__(256, WPE) void gemm_plain(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                       _Float16* __restrict__ O, int M, int N, int K, int lda) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    char* ldsA = smem;
    char* ldsB = smem + 16384;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int tilesN = N / 128, nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q8 = nwg / 8, r8 = nwg % 8, xcd = id % 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + id / 8;
    }
    const int tile_m = id / tilesN, tile_n = id - tile_m * tilesN;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int ldrow = tid >> 3, q = (tid & 7) ^ (ldrow & 7);
    const _Float16* ap = A + (long)(m0 + ldrow) * lda + q * 8;
    const _Float16* wp = W + (long)(n0 + ldrow) * K + q * 8;
    const long astep = 32L * lda, wstep = 32L * K;
    const int fr = lane & 15, fq = lane >> 4;
    const int c0 = fq ^ (fr & 7), c1 = (4 + fq) ^ (fr & 7);
    const int oa = ((wm * 64 + fr) * 8) * 16, ob = ((wn * 64 + fr) * 8) * 16;
    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / 64;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ap + i * astep + kt * 64), LDS_PTR(ldsA + (wv * 64 + 256 * i) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wp + i * wstep + kt * 64), LDS_PTR(ldsB + (wv * 64 + 256 * i) * 16), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk ? c1 : c0;
            f16x8 b[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *(const f16x8*)(ldsB + ob + c * 16 + ni * 16 * 128);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f16x8 a = *(const f16x8*)(ldsA + oa + c * 16 + mi * 16 * 128);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a, acc[ni][mi], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + fq * 4;
            f16x4 h = {(_Float16)acc[ni][mi][0], (_Float16)acc[ni][mi][1], (_Float16)acc[ni][mi][2], (_Float16)acc[ni][mi][3]};
            *(f16x4*)(O + (long)m * N + n) = h;
        }
    }
}

template <int WPE>
static float run(const _Float16* A, const _Float16* W, _Float16* O, int M, int N, int K, int lda, int iters) {
    const int tiles = ((M + 127) / 128) * (N / 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_plain<WPE>, dim3(tiles), dim3(256), 0, 0, A, W, O, M, N, K, lda);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_plain<WPE>, dim3(tiles), dim3(256), 0, 0, A, W, O, M, N, K, lda);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main() {
    struct Shape { const char* name; int M, N, K, lda; };
    const Shape shapes[] = {{"qkv 3B", 38208, 2304, 768, 768},   {"ffn1 3B", 38208, 3072, 768, 768},  {"ffn2 3B", 38208, 768, 3072, 3072},
                            {"out 3B", 38208, 768, 768, 768},    {"fe1 (k3 s2)", 1228800, 512, 1536, 1024}, {"fe2", 614400, 512, 1536, 1024},
                            {"8192^3", 8192, 8192, 8192, 8192}};
    for (int w = 4; w <= 5; ++w) {
        int nb = 0;
        if (w == 4) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_plain<4>, 256, 0);
        else hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_plain<5>, 256, 0);
        printf("build for %d workgroups per CU: runtime reports %d resident\n", w, nb);
    }
    for (const Shape& s : shapes) {
        const size_t na = (size_t)(s.M + 256) * s.lda + 4096, nw = (size_t)s.N * s.K, no = (size_t)s.M * s.N;
        _Float16 *A, *W, *O;
        hipMalloc(&A, na * 2);
        hipMalloc(&W, nw * 2);
        hipMalloc(&O, no * 2);
        std::vector<_Float16> h(na > nw ? na : nw);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 255 - 127) / 256.0f);
        hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
        const float t4 = run<4>(A, W, O, s.M, s.N, s.K, s.lda, 20);
        std::vector<_Float16> o4(1024), o5(1024);
        hipMemcpy(o4.data(), O + no / 2, 2048, hipMemcpyDeviceToHost);
        const float t5 = run<5>(A, W, O, s.M, s.N, s.K, s.lda, 20);
        hipMemcpy(o5.data(), O + no / 2, 2048, hipMemcpyDeviceToHost);
        bool same = true;
        for (int i = 0; i < 1024; ++i) same = same && (float)o4[i] == (float)o5[i];
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("%-12s M=%8d N=%5d K=%5d | 4/CU %8.1f us %7.1f TF | 5/CU %8.1f us %7.1f TF | %s\n", s.name, s.M, s.N, s.K, t4 * 1e3,
               fl / t4 / 1e9, t5 * 1e3, fl / t5 / 1e9, same ? "same" : "DIFFERENT");
        hipFree(A); hipFree(W); hipFree(O);
    }
    return 0;
}
