// CPU check of csrc/fft512.h: runs the per-lane pass functions for all 64 lanes, loads before stores,
// exactly as one wavefront does, and compares with a direct O(N^2) double-precision DFT.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <complex>
#include "fft512.h"
using namespace advh;
static cf tw[1024];
template <int DIR> void fft512(float* re, float* im) {
    cf v[64][8];
    for (int l = 0; l < 64; ++l) fft512_pass_load<DIR, 1>(re, im, l, tw, v[l]);
    for (int l = 0; l < 64; ++l) fft512_pass_store<1>(re, im, l, v[l]);
    for (int l = 0; l < 64; ++l) fft512_pass_load<DIR, 8>(re, im, l, tw, v[l]);
    for (int l = 0; l < 64; ++l) fft512_pass_store<8>(re, im, l, v[l]);
    for (int l = 0; l < 64; ++l) fft512_pass_load<DIR, 64>(re, im, l, tw, v[l]);
    for (int l = 0; l < 64; ++l) fft512_pass_store<64>(re, im, l, v[l]);
}
int main() {
    for (int k = 0; k < 1024; ++k) { tw[k].x = (float)cos(2 * M_PI * k / 1024); tw[k].y = (float)sin(2 * M_PI * k / 1024); }
    std::vector<double> x(1024);
    srand(1); for (auto& v : x) v = rand() / (double)RAND_MAX * 2 - 1;
    float re[FFT_ROW] = {0}, im[FFT_ROW] = {0};
    for (int n = 0; n < 512; ++n) { re[fidx(n)] = (float)x[2 * n]; im[fidx(n)] = (float)x[2 * n + 1]; }
    fft512<-1>(re, im);
    // post-process to the real transform
    std::vector<std::complex<double>> X(513), R(513);
    cf Z0{re[fidx(0)], im[fidx(0)]};
    X[0] = Z0.x + Z0.y; X[512] = Z0.x - Z0.y;
    for (int k = 1; k <= 256; ++k) {
        cf A{re[fidx(k)], im[fidx(k)]}, B{re[fidx(512 - k)], im[fidx(512 - k)]}, xk, xm;
        rfft_post_pair(A, B, tw[k], xk, xm);
        X[k] = {xk.x, xk.y}; X[512 - k] = {xm.x, xm.y};
    }
    double err = 0, mx = 0;
    for (int k = 0; k <= 512; ++k) {
        std::complex<double> s = 0;
        for (int n = 0; n < 1024; ++n) s += x[n] * std::polar(1.0, -2 * M_PI * k * n / 1024);
        R[k] = s; err = fmax(err, std::abs(s - X[k])); mx = fmax(mx, std::abs(s));
    }
    printf("rfft max err %.3e (max |X| %.3f)\n", err, mx);
    if (err > 2e-5 * mx) return 1;
    // inverse from the exact spectrum
    cf z0{(float)(0.5 * (R[0].real() + R[512].real())), (float)(0.5 * (R[0].real() - R[512].real()))};
    re[fidx(0)] = z0.x; im[fidx(0)] = z0.y;
    for (int k = 1; k <= 256; ++k) {
        cf A{(float)R[k].real(), (float)R[k].imag()}, B{(float)R[512 - k].real(), (float)R[512 - k].imag()}, zk, zm;
        irfft_pre_pair(A, B, tw[k], zk, zm);
        re[fidx(k)] = zk.x; im[fidx(k)] = zk.y; re[fidx(512 - k)] = zm.x; im[fidx(512 - k)] = zm.y;
    }
    fft512<+1>(re, im);
    double ierr = 0;
    for (int n = 0; n < 512; ++n) {
        ierr = fmax(ierr, fabs(re[fidx(n)] / 512.0 - x[2 * n]));
        ierr = fmax(ierr, fabs(im[fidx(n)] / 512.0 - x[2 * n + 1]));
    }
    printf("irfft max err %.3e\n", ierr);
    return ierr > 2e-6 ? 2 : 0;
}
