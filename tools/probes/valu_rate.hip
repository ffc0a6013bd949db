// measures VALU throughput (cycles per wave-instruction per SIMD) for the integer ops the scan uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int OP>
__global__ void k(unsigned* out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    unsigned c = blockIdx.x | 1;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                                   \
    if (OP == 0) x = (x & 0x06060606u) + c;                                                        \
    else if (OP == 1) x = __umul24(x, 0x41041u);                                                   \
    else if (OP == 2) x = __builtin_amdgcn_alignbit(x, c, 8);                                      \
    else if (OP == 3) x = (x >> 19) & 0xFFu;                                                       \
    else if (OP == 4) x = x * 0x9E3779B1u;                                                         \
    else if (OP == 5) x = __umul24(x, 0x41041u) + c;                                               \
    else if (OP == 6) x = (x << 3) | c;                                                            \
    else if (OP == 7) x = __builtin_amdgcn_update_dpp(c, x, 0x130, 0xf, 0xf, false);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(const char* name, unsigned* d, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<256, threads>>>(d, 100);
    CHECK(hipEventRecord(e0));
    k<OP><<<256, threads>>>(d, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double insts_per_simd = (double)iters * 8 * (threads / 64) / 4.0;  // wave-instructions per SIMD
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, threads / 256, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    unsigned* d; CHECK(hipMalloc(&d, 256 * 1024 * 4));
    for (int threads : {256, 512, 1024}) {
        run<0>("and+add (2 ops)", d, threads); run<1>("mul_u32_u24", d, threads); run<2>("alignbit", d, threads); run<3>("bfe", d, threads);
        run<4>("mul_lo_u32", d, threads); run<5>("mad_u32_u24", d, threads); run<6>("lshl_or", d, threads); run<7>("mov_dpp wave_shl", d, threads);
    }
    return 0;
}
