// Where does the dispatcher put workgroup k?  Dumps (XCC, SE, CU) of every workgroup of a 1-D grid of 256-thread blocks
// that each hold 32 KB of LDS (4 per CU), in launch order.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    __shared__ char pad[32 * 1024];
    pad[threadIdx.x] = 0;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (pad[threadIdx.x] == 77) out[0] = 0;
}
int main() {
    const int N = 4096;
    unsigned* d; hipMalloc(&d, N * 8);
    k<<<N, 256>>>(d, 200000);
    static unsigned h[2 * N]; hipMemcpy(h, d, N * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 72; ++i) {
        unsigned hw = h[2 * i], x = h[2 * i + 1];
        printf("wg %4d: xcc %u se %u sh %u cu %2u simd %u wave %u\n", i, x & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
    }
    // per (xcc, se, cu) list of the first workgroups that landed there
    printf("first 1024 workgroups: cu slot -> workgroup ids\n");
    for (int x = 0; x < 1; ++x)
        for (int key = 0; key < 128; ++key) {
            int cnt = 0;
            char buf[256]; int n = 0;
            for (int i = 0; i < 1024; ++i) {
                unsigned hw = h[2 * i], xc = h[2 * i + 1] & 0xf;
                int k2 = (int)(((hw >> 13) & 7) << 4 | ((hw >> 8) & 15));
                if ((int)xc == x && k2 == key) { n += snprintf(buf + n, sizeof(buf) - n, " %d", i); ++cnt; }
            }
            if (cnt) printf("xcc %d se/cu %02x:%s\n", x, key, buf);
        }
    return 0;
}
