// Lane mapping of ds_read_b64_tr_b16 (gfx950), checked empirically: prints, for a few lanes, which (row, col) elements
// of a [32][64] fp16 LDS tile (value = row * 64 + col) arrive in the 4 output halfs when lane 4q+p of each 16-lane
// group supplies the address of (row r0 + q, col c0 + 4p).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 t[32 * 64];
    for (int i = threadIdx.x; i < 2048; i += 64) t[i] = (_Float16)i;
    __syncthreads();
    int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    int row = 4 * g + q, col = 16 + 4 * p;                       // r0 = 4g, c0 = 16
    unsigned addr = (unsigned)(unsigned long)((__attribute__((address_space(3))) void*)(t + row * 64 + col));
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int lane : {0, 1, 5, 15, 16, 17, 33, 63}) {
        printf("lane %2d:", lane);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", (int)h[lane * 4 + j] / 64, (int)h[lane * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
