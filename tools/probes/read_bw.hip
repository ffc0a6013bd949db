// read-bandwidth probe: how fast can one MI355X stream a 15 GB buffer with different access shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A: classic grid-stride, one dwordx4 per thread per iteration
template <bool NT>
__global__ void k_gridstride(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// B: per-wave tiles of TILE KiB dealt round-robin, UNROLL loads in flight per wave
template <bool NT, int UNROLL>
__global__ __launch_bounds__(1024) void k_tiles(const u32x4* __restrict__ p, size_t n_tiles, int tile_chunks, unsigned* out) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * (blockDim.x / 64);
    unsigned acc = 0;
    for (size_t t = wave; t < n_tiles; t += n_waves) {
        const u32x4* base = p + t * (size_t)tile_chunks * 64 + lane;
        for (int c = 0; c < tile_chunks; c += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(base + (size_t)(c + u) * 64) : base[(size_t)(c + u) * 64];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <class F>
void timeit(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 15ull << 30;
    u32x4* d; unsigned* out; CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(d, 1, bytes));
    const size_t n16 = bytes / 16;
    timeit("gridstride 2048x256", bytes, [&] { k_gridstride<false><<<2048, 256>>>(d, n16, out); });
    timeit("gridstride 2048x256 nt", bytes, [&] { k_gridstride<true><<<2048, 256>>>(d, n16, out); });
    timeit("gridstride 8192x256 nt", bytes, [&] { k_gridstride<true><<<8192, 256>>>(d, n16, out); });
    timeit("gridstride 256x1024 nt", bytes, [&] { k_gridstride<true><<<256, 1024>>>(d, n16, out); });
    for (int tc : {16, 32, 64, 128}) {
        char nm[96];
        size_t nt = bytes / ((size_t)tc * 1024);
        snprintf(nm, sizeof(nm), "tiles %3d KiB, 256x1024, 4 in flight, nt", tc);
        timeit(nm, nt * tc * 1024, [&] { k_tiles<true, 4><<<256, 1024>>>(d, nt, tc, out); });
        snprintf(nm, sizeof(nm), "tiles %3d KiB, 256x1024, 8 in flight, nt", tc);
        timeit(nm, nt * tc * 1024, [&] { k_tiles<true, 8><<<256, 1024>>>(d, nt, tc, out); });
        snprintf(nm, sizeof(nm), "tiles %3d KiB, 256x1024, 4 in flight", tc);
        timeit(nm, nt * tc * 1024, [&] { k_tiles<false, 4><<<256, 1024>>>(d, nt, tc, out); });
        snprintf(nm, sizeof(nm), "tiles %3d KiB, 512x1024(2/CU), 4 in flight, nt", tc);
        timeit(nm, nt * tc * 1024, [&] { k_tiles<true, 4><<<512, 1024>>>(d, nt, tc, out); });
    }
    return 0;
}
