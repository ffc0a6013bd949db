// minimizer-bound probe (config 5, r04 review item 6): the LOWER BOUND of a scan that indexes each 21-mer pattern by ONE anchor q-gram
// -- its minimizer (q = 13: the smallest hash among its 9 q-grams) -- instead of by the 8 q-grams a stride-8 sample can land on.
// 500 k patterns then give 500 k level-1 entries instead of 4 M, and level 1 fits LDS: a 1 Mbit (128 KiB) bitmap, 39 % full.  The price
// is on the text side: every position's q-gram must be hashed, the minimum of every window of 9 found, and each NEW minimizer probed
// (density ~ 2 / 10 of the positions against 1 / 8 for the stride; DESIGN.md 5.2 estimated this and declined -- this measures it).
//   per lane and 1 KiB chunk: 16 bases + 20 bases of halo -> 24 q-gram hashes -> 16 window minima -> an LDS bit probe per new minimum
//   -> a bitmap positive costs one random 8-byte read of a 4 MiB table (the level-2 look-up that follows; L2-resident)
// No verification, no flags, no tuples: what this cannot do faster than today's 2.9 ms kernel (bar: 1.6 ms) no real kernel can.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack4_top(unsigned d) { return (d & 0x06060606u) * 0x00820820u; }
__device__ __forceinline__ unsigned pack16(u32x4 v) {
    const unsigned u0 = pack4_top(v.x), u1 = pack4_top(v.y), u2 = pack4_top(v.z), u3 = pack4_top(v.w);
    return __builtin_amdgcn_perm(u1, u0, 0x0c0c0703u) | __builtin_amdgcn_perm(u3, u2, 0x07030c0cu);
}
__device__ __forceinline__ unsigned hash13(unsigned key26) {  // 26 key bits -> 32 mixed bits (two 24-bit multiplies, as filter.hpp)
    return __umul24(key26, 0x9E3779u) + __umul24(key26 >> 24, 0x85EBCBu) * 0x10001u;
}

template <int FILL_SHIFT>
__global__ __launch_bounds__(1024) void probe(const unsigned char* __restrict__ p, size_t n_tiles, const unsigned* __restrict__ bitmap_img,
                                              const uint2* __restrict__ table, unsigned table_mask, unsigned long long* __restrict__ out) {
    extern __shared__ unsigned bitmap[];  // 32 Ki words = 1 Mbit
    for (unsigned i = threadIdx.x; i < 32768; i += 1024) bitmap[i] = bitmap_img[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 16;
    unsigned probes = 0, positives = 0, acc = 0;
    for (size_t t = wave; t < n_tiles; t += n_waves) {
        const unsigned char* base = p + t * (size_t)(31 * 1024) + lane * 16;
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + u * 1024));
        unsigned last_min = 0;
        for (int c = 0; c < 32; c += 4) {
            unsigned pk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pk[u] = pack16(v[u]);
            asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]) : "memory");
            if (c + 4 < 32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (c + 4 + u) * 1024));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned w0 = pk[u];
                // the 32 bases behind the lane's own 16: the next two lanes' words (wave_shl 1 and 2; the last lanes wrap: a probe)
                const unsigned w1 = __builtin_amdgcn_update_dpp(0, w0, 0x130, 0xf, 0xf, false);
                const unsigned w2 = __builtin_amdgcn_update_dpp(0, w1, 0x130, 0xf, 0xf, false);
                // hashes of the q-grams at positions 0..23 (13 bases = 26 bits each)
                unsigned h[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) {
                    unsigned key;
                    if (k < 16) key = (k == 0 ? w0 : __builtin_amdgcn_alignbit(w1, w0, 2 * k)) & 0x3FFFFFFu;
                    else key = (k == 16 ? w1 : __builtin_amdgcn_alignbit(w2, w1, 2 * (k - 16))) & 0x3FFFFFFu;
                    h[k] = hash13(key);
                }
                // minima of the 16 windows of 9: blocks of 8 with suffix / prefix minima (van Herk), 3 ops per element
                unsigned suf[24], pre[24];
#pragma unroll
                for (int b = 0; b < 24; b += 8) {
                    pre[b] = h[b];
#pragma unroll
                    for (int k = 1; k < 8; ++k) pre[b + k] = min(pre[b + k - 1], h[b + k]);
                    suf[b + 7] = h[b + 7];
#pragma unroll
                    for (int k = 6; k >= 0; --k) suf[b + k] = min(suf[b + k + 1], h[b + k]);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    // window [k, k + 8]: suffix of k's block from k, prefix of the next block up to k + 8 (two blocks when k % 8 == 0: + h[k + 8])
                    const unsigned m = (k % 8 == 0) ? min(suf[k], h[k + 8]) : min(suf[k], pre[k + 8]);
                    if (m != last_min) {  // a new minimizer: level 1 in LDS
                        ++probes;
                        const unsigned idx = m >> 12;  // 20 bits
                        const unsigned bit = (bitmap[idx >> 5] >> (idx & 31)) & 1u;
                        if (bit) {  // level 2: one random 8-byte read (L2)
                            ++positives;
                            const uint2 e = table[(m * 0x9E3779B1u >> 8) & table_mask];
                            acc += e.x ^ e.y;
                        }
                        last_min = m;
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) probes += __shfl_down(probes, o), positives += __shfl_down(positives, o), acc += __shfl_down(acc, o);
    if (lane == 0) {
        atomicAdd(&out[0], (unsigned long long)probes);
        atomicAdd(&out[1], (unsigned long long)positives);
        atomicAdd(&out[2], (unsigned long long)acc);
    }
}

int main() {
    const size_t n_bytes = 12500000ull * 250;  // config 5's shard
    const size_t n_tiles = n_bytes / (31 * 1024);
    unsigned char* d_text;
    CHECK(hipMalloc(&d_text, n_bytes + (1 << 20)));
    {  // uniform ACGT
        unsigned char* h = (unsigned char*)malloc(64 << 20);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < (64u << 20); ++i) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            h[i] = "ACGT"[(s >> 33) & 3];
        }
        for (size_t at = 0; at < n_bytes; at += 64u << 20) CHECK(hipMemcpy(d_text + at, h, n_bytes - at < (64u << 20) ? n_bytes - at : (64u << 20), hipMemcpyHostToDevice));
        free(h);
    }
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_out, 32));
    uint2* d_table;
    const unsigned table_entries = 1u << 19;  // 4 MiB of 8-byte entries
    CHECK(hipMalloc(&d_table, (size_t)table_entries * 8));
    CHECK(hipMemset(d_table, 1, (size_t)table_entries * 8));
    CHECK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int fill_pct : {39, 22, 63}) {  // 500 k keys in 1 Mbit with one hash bit: 39 %; with half / twice the keys for scale
        unsigned* h_bits = (unsigned*)calloc(32768, 4);
        unsigned long long s = 0x9E3779B97F4A7C15ull;
        const size_t n_keys = fill_pct == 39 ? 500000 : fill_pct == 22 ? 250000 : 1000000;
        for (size_t k = 0; k < n_keys; ++k) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            const unsigned idx = (unsigned)(s >> 20) & 0xFFFFFu;
            h_bits[idx >> 5] |= 1u << (idx & 31);
        }
        unsigned* d_bits;
        CHECK(hipMalloc(&d_bits, 131072));
        CHECK(hipMemcpy(d_bits, h_bits, 131072, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        unsigned long long res[3] = {0, 0, 0};
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipMemset(d_out, 0, 32));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 131072, 0, d_text, n_tiles, d_bits, d_table, table_entries - 1, d_out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
            CHECK(hipMemcpy(res, d_out, 24, hipMemcpyDeviceToHost));
        }
        printf("bitmap %2d %% full (%zu keys): %.3f ms per shard; %.1f M LDS probes (%.3f per base), %.1f M level-2 reads (%.3f per base)\n", fill_pct, n_keys, best,
               res[0] / 1e6, (double)res[0] / n_bytes, res[1] / 1e6, (double)res[1] / n_bytes);
        CHECK(hipFree(d_bits));
        free(h_bits);
    }
    return 0;
}
