// probes DPP wave_shl:1 semantics on gfx950: which lane does lane i read, what do boundary lanes get
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned v = 100 + threadIdx.x;
    unsigned a = __builtin_amdgcn_update_dpp(7777u, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    unsigned b = __builtin_amdgcn_update_dpp(7777u, v, 0x101, 0xf, 0xf, false);  // row_shl:1
    unsigned c = __builtin_amdgcn_update_dpp(7777u, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c;
}
int main() {
    unsigned* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); unsigned h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; ++r) { printf("%s:", r == 0 ? "wave_shl1" : r == 1 ? "row_shl1" : "wave_rol1"); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
