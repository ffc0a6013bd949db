// random-gather probe: how many independent 8-byte (or 4-byte) reads per second can one MI355X
// serve from a table of a given size, by instruction flavour?  Sizes the global-memory filter
// of the large-pattern-set path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { PLAIN8 = 0, NT8 = 1, SC8 = 2, ATOM8 = 3, PLAIN4 = 4, SC4 = 5, PLAIN16 = 6 };

__device__ __forceinline__ uint32_t next_idx(uint32_t &s) {  // cheap per-lane LCG
    s = s * 1664525u + 1013904223u;
    return s >> 4;
}

template <int V, int INFLIGHT>
__global__ __launch_bounds__(1024) void k_gather(const uint64_t *__restrict__ tab, uint32_t mask8, int iters, unsigned *out) {
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const uint32_t i = next_idx(s) & mask8;  // index of an 8-byte block
            if constexpr (V == PLAIN8) {
                v[u] = tab[i];
            } else if constexpr (V == NT8) {
                v[u] = __builtin_nontemporal_load(tab + i);
            } else if constexpr (V == SC8) {
                uint64_t r;
                asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(r) : "v"(tab + i) : "memory");
                v[u] = r;
            } else if constexpr (V == ATOM8) {
                v[u] = __hip_atomic_fetch_or(const_cast<uint64_t *>(tab) + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if constexpr (V == PLAIN4) {
                v[u] = reinterpret_cast<const uint32_t *>(tab)[2 * i];
            } else if constexpr (V == SC4) {
                uint32_t r;
                asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(r) : "v"(reinterpret_cast<const uint32_t *>(tab) + 2 * i) : "memory");
                v[u] = r;
            } else if constexpr (V == PLAIN16) {
                const uint4 q = reinterpret_cast<const uint4 *>(tab)[i >> 1];
                v[u] = q.x ^ q.w;
            }
        }
        if constexpr (V == SC8 || V == SC4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) acc ^= v[u];
    }
    if (acc == 0x123456789ull) out[0] = 1;
}

template <int V, int INFLIGHT>
void run(const char *name, const uint64_t *tab, size_t bytes, unsigned *out) {
    const uint32_t mask8 = (uint32_t)(bytes / 8 - 1);
    const int iters = 4096 / INFLIGHT;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_gather<V, INFLIGHT><<<256, 1024>>>(tab, mask8, iters, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        k_gather<V, INFLIGHT><<<256, 1024>>>(tab, mask8, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double probes = 256.0 * 1024 * iters * INFLIGHT;
    printf("  %-28s %8.3f ms  %7.1f G probes/s\n", name, best, probes / (best * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    unsigned *out;
    CHECK(hipMalloc(&out, 4));
    for (size_t mb : {1, 2, 4, 8, 32, 256}) {
        const size_t bytes = mb << 20;
        uint64_t *tab;
        CHECK(hipMalloc(&tab, bytes));
        CHECK(hipMemset(tab, 0x5a, bytes));
        printf("table %zu MiB\n", mb);
        run<PLAIN8, 4>("load 8 B, 4 in flight", tab, bytes, out);
        run<PLAIN8, 8>("load 8 B, 8 in flight", tab, bytes, out);
        run<NT8, 4>("nt load 8 B, 4 in flight", tab, bytes, out);
        run<SC8, 4>("sc0 sc1 load 8 B, 4 in flight", tab, bytes, out);
        run<PLAIN4, 4>("load 4 B, 4 in flight", tab, bytes, out);
        run<SC4, 4>("sc0 sc1 load 4 B, 4 in flight", tab, bytes, out);
        run<PLAIN16, 4>("load 16 B, 4 in flight", tab, bytes, out);
        run<ATOM8, 4>("atomic or 0 (8 B), 4 in flight", tab, bytes, out);
        CHECK(hipFree(tab));
    }
    return 0;
}
