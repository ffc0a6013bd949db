// stream-policy probe: a wave streams 31 KiB tiles (4 x 1 KiB loads in flight) AND gathers G random
// 8-byte blocks per lane and chunk from a table of T MiB -- the shape of the global-filter scan kernel
// without its arithmetic.  Question: which cache policy on the STREAM loads lets the table stay in the
// 4 MiB L2 of an XCD?  plain | nt (builtin) | buffer loads with aux = sc1 / sc0+sc1 / nt+sc1 / nt+sc0+sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ u32x4 sload(const unsigned char* tile_base, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, int soff) {
    if constexpr (POLICY == 0) return *reinterpret_cast<const u32x4*>(tile_base + voff + soff);
    else if constexpr (POLICY == 1) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tile_base + voff + soff));
    else return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, POLICY);  // aux bits: sc0 = 1, nt = 2, sc1 = 16
}

template <int POLICY, int G>
__global__ __launch_bounds__(1024) void k(const unsigned char* __restrict__ p, size_t n_tiles, const uint2* __restrict__ tab, unsigned tab_blocks,
                                          unsigned* out) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 16;
    unsigned acc = 0, s = (unsigned)(wave * 64 + lane) * 2654435761u + 1u;
    for (size_t t = wave; t < n_tiles; t += n_waves) {
        const unsigned char* base = p + t * (size_t)(32 * 1024);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, 32 * 1024, 0x00020000);
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = sload<POLICY>(base, lane * 16, rsrc, u * 1024);
        for (int c = 4; c < 32 + 4; c += 4) {
            unsigned x = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) x ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            if (c < 32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = sload<POLICY>(base, lane * 16, rsrc, (c + u) * 1024);
            }
            // the gathers of this group: 4 chunks x G per lane, addresses depend on the streamed data a little
            s ^= x;
#pragma unroll
            for (int g = 0; g < 4 * G; ++g) {
                s = s * 1664525u + 1013904223u;
                const uint2 b = tab[(unsigned)(((unsigned long long)s * tab_blocks) >> 32)];
                acc += b.x ^ b.y;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int POLICY, int G>
void run(const char* name, const unsigned char* d, size_t bytes, const uint2* tab, unsigned tab_blocks, unsigned* out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n_tiles = bytes / (32 * 1024);
    k<POLICY, G><<<256, 1024>>>(d, n_tiles, tab, tab_blocks, out); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) { CHECK(hipEventRecord(e0)); k<POLICY, G><<<256, 1024>>>(d, n_tiles, tab, tab_blocks, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    const double gathers = (double)n_tiles * 32 * 64 * G;
    printf("  %-22s G=%d  %.3f ms  stream %.2f TB/s  gathers %.1f G/s\n", name, G, best, bytes / (best * 1e-3) / 1e12, gathers / (best * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = 4ull << 30;
    unsigned char* d; unsigned* out; CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(d, 1, bytes));
    for (unsigned mib : {2u, 3u, 4u, 8u}) {
        const unsigned blocks = mib * 131072u;
        uint2* tab; CHECK(hipMalloc(&tab, (size_t)blocks * 8)); CHECK(hipMemset(tab, 3, (size_t)blocks * 8));
        printf("table %u MiB\n", mib);
        run<0, 0>("plain, no gathers", d, bytes, tab, blocks, out);
        run<1, 0>("nt, no gathers", d, bytes, tab, blocks, out);
        run<19, 0>("buf nt+sc0+sc1, none", d, bytes, tab, blocks, out);
        run<0, 2>("plain", d, bytes, tab, blocks, out);
        run<1, 2>("nt", d, bytes, tab, blocks, out);
        run<2, 2>("buf nt", d, bytes, tab, blocks, out);
        run<16, 2>("buf sc1", d, bytes, tab, blocks, out);
        run<17, 2>("buf sc0+sc1", d, bytes, tab, blocks, out);
        run<18, 2>("buf nt+sc1", d, bytes, tab, blocks, out);
        run<19, 2>("buf nt+sc0+sc1", d, bytes, tab, blocks, out);
        CHECK(hipFree(tab));
    }
    return 0;
}
