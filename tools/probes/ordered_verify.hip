// ordered-verify probe (r04 review item 4): what would a SECOND kernel cost that verifies the level-2-confirmed
// (position, pattern) candidates in position order, instead of level 3 inside the streaming kernel?
//
// Shape: 100 M x 150 bp of text (15 GB), 10 000 31-mers, one candidate in every 1/d-th read (d = 1 %, 10 %, 100 % of the
// reads).  The candidate list is given in three orders:
//   sorted   every candidate in ascending text position (what per-2-MiB-page binning would give);
//   flushes  the order the EMIT scan kernels would produce with their per-wave staging: chunks of ~1000 candidates,
//            each chunk ascending inside one wave's tiles (runs of four 31 KiB tiles, then a jump of n_waves runs),
//            chunks in arbitrary order;
//   random   no locality at all (the bound the streaming kernel's level 3 is blamed for).
// One lane per candidate: two overlapping 16-byte loads of the text window and of the pattern (resolve_one's WIDE form),
// record = position / 150, flag byte stored on a match.  Reported: kernel time per list.  Against it: level 3 inside the
// scan kernel costs +0.6 ms per 10^7 occurrences at 10 % and +2.9 ms per 10^8 at 100 %, and the
// candidate list itself must be written (8 B each) by pass 1 and read back here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill(unsigned* p, size_t n_words, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
        const unsigned lut = 'A' | ('C' << 8) | ('G' << 16) | ((unsigned)'T' << 24);
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) w |= ((lut >> (8 * ((x >> (2 * b)) & 3))) & 0xFFu) << (8 * b);
        p[i] = w;
    }
}

__global__ __launch_bounds__(256) void verify(const unsigned char* __restrict__ text, const unsigned char* __restrict__ pats, const uint2* __restrict__ list,
                                              size_t n, unsigned char* __restrict__ flags, unsigned long long* __restrict__ n_true) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned hit = 0;
    if (i < n) {
        const uint2 e = list[i];  // x = position / 2 (fits 32 bits for 15 GB), y = pattern
        const size_t p = (size_t)e.x * 2;
        const unsigned char* tx = text + p;
        const unsigned char* pt = pats + (size_t)e.y * 31;
        unsigned long long x[2], y[2], u[2], v[2];
        __builtin_memcpy(x, tx, 16); __builtin_memcpy(y, pt, 16); __builtin_memcpy(u, tx + 15, 16); __builtin_memcpy(v, pt + 15, 16);
        const unsigned long long diff = (x[0] ^ y[0]) | (x[1] ^ y[1]) | (u[0] ^ v[0]) | (u[1] ^ v[1]);
        if (diff == 0) {
            flags[(size_t)((double)p * (1.0 / 150.0))] = 1;
            hit = 1;
        }
    }
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(n_true, (unsigned long long)__popcll(m));
}

int main() {
    const size_t n_rec = 100000000ull, L = 150, bytes = n_rec * L;
    unsigned char* d; CHECK(hipMalloc(&d, bytes + 65536));
    fill<<<4096, 256>>>(reinterpret_cast<unsigned*>(d), (bytes + 65536) / 4, 777u);
    unsigned char* pats; CHECK(hipMalloc(&pats, 10000 * 31 + 64));
    fill<<<64, 256>>>(reinterpret_cast<unsigned*>(pats), (10000 * 31 + 64) / 4, 99u);
    unsigned char* flags; CHECK(hipMalloc(&flags, n_rec)); CHECK(hipMemset(flags, 0, n_rec));
    unsigned long long* n_true; CHECK(hipMalloc(&n_true, 8)); CHECK(hipMemset(n_true, 0, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::mt19937_64 rng(5);
    const size_t tile = 31 * 1024, n_waves = 4096, run = 4;
    for (int every : {100, 10, 1}) {
        const size_t n = n_rec / every;
        std::vector<uint2> base(n);
        for (size_t k = 0; k < n; ++k) {  // one candidate in every `every`-th read, at an even offset that keeps the 31-mer inside
            const size_t r = k * every;
            const size_t p = r * L + 2 * (rng() % 60);
            base[k] = make_uint2((unsigned)(p / 2), (unsigned)(rng() % 10000));
        }
        uint2* dl; CHECK(hipMalloc(&dl, n * 8));
        for (int order = 0; order < 3; ++order) {
            std::vector<uint2> l = base;  // already ascending
            if (order == 1) {
                // wave of a position: tile t belongs to run t / 4, run r to wave r % n_waves; a wave visits its runs in
                // ascending order and flushes ~1000 candidates at a time; flushes of different waves interleave
                std::vector<std::vector<uint2>> per_wave(n_waves);
                for (auto& c : l) per_wave[((size_t)c.x * 2 / tile / run) % n_waves].push_back(c);
                std::vector<std::pair<unsigned long long, std::pair<size_t, size_t>>> chunks;  // (random key, (wave, first))
                for (size_t w = 0; w < n_waves; ++w)
                    for (size_t f = 0; f < per_wave[w].size(); f += 1000) chunks.push_back({rng(), {w, f}});
                std::sort(chunks.begin(), chunks.end());
                size_t o = 0;
                for (auto& ch : chunks) {
                    auto& v = per_wave[ch.second.first];
                    for (size_t f = ch.second.second; f < std::min(v.size(), ch.second.second + 1000); ++f) l[o++] = v[f];
                }
            } else if (order == 2) {
                std::shuffle(l.begin(), l.end(), rng);
            }
            CHECK(hipMemcpy(dl, l.data(), n * 8, hipMemcpyHostToDevice));
            float best = 1e9;
            for (int r = 0; r < 4; ++r) {
                CHECK(hipEventRecord(e0));
                verify<<<(unsigned)((n + 255) / 256), 256>>>(d, pats, dl, n, flags, n_true);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms < best) best = ms;
            }
            printf("1 read in %3d with a candidate (%9zu candidates), list %-8s: verify kernel %.3f ms  (%.1f ns... %.2f G candidates/s; list itself: %.2f GB written + read)\n",
                   every, n, order == 0 ? "sorted" : order == 1 ? "flushes" : "random", best, best * 1e6 / n, n / (best * 1e-3) / 1e9, 2 * n * 8 / 1e9);
        }
        CHECK(hipFree(dl));
    }
    return 0;
}
