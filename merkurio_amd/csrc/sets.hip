// sets.hip -- what the reference's record loops derive from the matcher's hits, computed on the device from the
// ordered tuples instead of by one host thread walking them:
//   * tag: the SET of matched patterns of every record, ascending and distinct -- kmers_found after sort_unstable +
//     dedup (src/cmd_tag.rs:392-442,484-485) -- as a CSR (found_off[n_rec + 1], found_pat);
//   * BNDMq pattern_hit_counts: += 1 per (record, pattern) with at least one hit (src/cmd_extract.rs:380-383,
//     src/cmd_tag.rs:431-433) = one count per entry of that CSR;
//   * log rows: logger.log_fields arguments (src/logger.rs:41), one mk_row per tuple in emission order.
// Input: the tuples in "set order" (record, pattern, position ascending -- the BNDMq emission order, which
// order_hits.hip produces for any matcher).  A tuple is a HEAD if its (record, pattern) differs from its
// predecessor's; heads are the distinct patterns of the records.  Three passes of hand-written kernels:
//   mk_sets_count_kernel   heads per 4096-tuple tile
//   mk_sets_scan_kernel    exclusive prefix sum over the tile counts (one workgroup) and the total
//   mk_sets_emit_kernel    found_pat[rank of a head] = its pattern; the LAST tuple of a record's run stores the
//                          number of heads up to and including itself at found_off[record + 1]
// and, because records without hits store nothing, a prefix MAXIMUM over found_off (values only grow along the
// records): mk_prefmax_tile_kernel / mk_prefmax_scan_kernel / mk_prefmax_apply_kernel, the same three-pass shape.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "scan_kernel.h"

namespace mk {

namespace {

constexpr int kSetsThreads = 1024;
constexpr int kSetsPer = 4;  // tuples (or offsets) per lane and tile
constexpr uint32_t kSetsTile = kSetsThreads * kSetsPer;

__device__ __forceinline__ bool is_head(const uint4 *__restrict__ hv, uint64_t i) {
    if (i == 0) return true;
    const uint4 a = hv[i - 1], b = hv[i];
    return a.x != b.x || a.y != b.y || a.z != b.z;  // record (64 bits) or pattern differs
}

// workgroup-wide sum of a per-lane count; the result is valid in every lane
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds /* 16 words */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kSetsThreads / 64; ++w) t += lds[w];
    return t;
}

__global__ __launch_bounds__(kSetsThreads) void mk_sets_count_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t lds[16];
    const uint4 *__restrict__ hv = reinterpret_cast<const uint4 *>(hits);
    const uint64_t base = (uint64_t)blockIdx.x * kSetsTile;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k) {
        const uint64_t i = base + (uint64_t)threadIdx.x * kSetsPer + k;  // consecutive tuples per lane
        if (i < n && is_head(hv, i)) ++c;
    }
    const uint32_t t = block_sum(c, lds);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = t;
}

// exclusive prefix sum over n_tiles counts, in place (one workgroup, any n_tiles); total -> *total
__global__ __launch_bounds__(kSetsThreads) void mk_sets_scan_kernel(uint32_t *__restrict__ tile_cnt, uint32_t n_tiles, unsigned long long *__restrict__ total) {
    __shared__ uint32_t part[kSetsThreads];
    const uint32_t per = (n_tiles + kSetsThreads - 1) / kSetsThreads;
    const uint32_t lo = std::min(n_tiles, threadIdx.x * per), hi = std::min(n_tiles, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_cnt[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < kSetsThreads; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = tile_cnt[i];
        tile_cnt[i] = run;
        run += c;
    }
    if (threadIdx.x == kSetsThreads - 1) *total = part[kSetsThreads - 1];
}

__global__ __launch_bounds__(kSetsThreads) void mk_sets_emit_kernel(const mk_hit *__restrict__ hits, uint64_t n, const uint32_t *__restrict__ tile_base,
                                                                     uint32_t *__restrict__ found_pat, unsigned long long *__restrict__ found_off,
                                                                     uint64_t n_rec) {
    __shared__ uint32_t wsum[kSetsThreads / 64];
    const uint4 *__restrict__ hv = reinterpret_cast<const uint4 *>(hits);
    const uint64_t base = (uint64_t)blockIdx.x * kSetsTile;
    const uint64_t i0 = base + (uint64_t)threadIdx.x * kSetsPer;
    bool head[kSetsPer];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k) {
        head[k] = i0 + k < n && is_head(hv, i0 + k);
        c += head[k];
    }
    // exclusive prefix of c over the workgroup: inside the wave by shuffles, across waves through LDS
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((threadIdx.x & 63) >= (uint32_t)o) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    uint32_t rank = tile_base[blockIdx.x] + before + incl - c;  // heads in front of this lane's first tuple
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k) {
        const uint64_t i = i0 + k;
        if (i >= n) break;
        const uint4 h = hv[i];
        if (head[k]) found_pat[rank++] = h.z;
        // the last tuple of its record: heads up to here = where the NEXT record's patterns start
        bool last = i + 1 == n;
        if (!last) {
            const uint4 nx = hv[i + 1];
            last = nx.x != h.x || nx.y != h.y;
        }
        const uint64_t rec = ((uint64_t)h.y << 32) | h.x;
        if (last && rec < n_rec) found_off[rec + 1] = rank;
    }
}

// ---- prefix maximum over v[0, n) (u64), in place ------------------------------------------------------------
__global__ __launch_bounds__(kSetsThreads) void mk_prefmax_tile_kernel(const unsigned long long *__restrict__ v, uint64_t n,
                                                                        unsigned long long *__restrict__ tile_max) {
    __shared__ unsigned long long lds[kSetsThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSetsTile;
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k) {
        const uint64_t i = base + (uint64_t)k * kSetsThreads + threadIdx.x;  // coalesced; the order does not matter for a maximum
        if (i < n) m = std::max(m, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = std::max<unsigned long long>(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kSetsThreads / 64; ++w) t = std::max(t, lds[w]);
        tile_max[blockIdx.x] = t;
    }
}

// exclusive prefix maximum over the tile maxima, in place (one workgroup)
__global__ __launch_bounds__(kSetsThreads) void mk_prefmax_scan_kernel(unsigned long long *__restrict__ tile_max, uint32_t n_tiles) {
    __shared__ unsigned long long part[kSetsThreads];
    const uint32_t per = (n_tiles + kSetsThreads - 1) / kSetsThreads;
    const uint32_t lo = std::min(n_tiles, threadIdx.x * per), hi = std::min(n_tiles, lo + per);
    unsigned long long m = 0;
    for (uint32_t i = lo; i < hi; ++i) m = std::max(m, tile_max[i]);
    part[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t o = 1; o < kSetsThreads; o <<= 1) {
        const unsigned long long v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] = std::max(part[threadIdx.x], v);
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;  // maximum of everything in front of this lane's tiles
    for (uint32_t i = lo; i < hi; ++i) {
        const unsigned long long c = tile_max[i];
        tile_max[i] = run;
        run = std::max(run, c);
    }
}

__global__ __launch_bounds__(kSetsThreads) void mk_prefmax_apply_kernel(unsigned long long *__restrict__ v, uint64_t n,
                                                                         const unsigned long long *__restrict__ tile_before) {
    __shared__ unsigned long long wmax[kSetsThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSetsTile;
    const uint64_t i0 = base + (uint64_t)threadIdx.x * kSetsPer;  // consecutive entries per lane: the order matters here
    unsigned long long x[kSetsPer];
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k) {
        x[k] = i0 + k < n ? v[i0 + k] : 0;
        m = std::max(m, x[k]);
        x[k] = m;  // running maximum inside the lane
    }
    unsigned long long incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(incl, o);
        if ((threadIdx.x & 63) >= (uint32_t)o) incl = std::max(incl, u);
    }
    if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = tile_before[blockIdx.x];
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before = std::max(before, wmax[w]);
    const unsigned long long left = __shfl_up(incl, 1);  // inclusive maximum of the lanes in front, inside the wave
    if ((threadIdx.x & 63) != 0) before = std::max(before, left);
#pragma unroll
    for (int k = 0; k < kSetsPer; ++k)
        if (i0 + k < n) v[i0 + k] = std::max(x[k], before);
}

// ---- counts[entry] += 1 over a u32 list (BNDMq pattern_hit_counts from found_pat) -----------------------------
__global__ void mk_count_u32_kernel(const uint32_t *__restrict__ list, uint64_t n, uint32_t *__restrict__ counts, uint32_t n_bins) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = list[i];
        if (b < n_bins) atomicAdd(&counts[b], 1u);
    }
}
// ---- log rows: mk_row {rec, pat, pos, file, 0} per tuple ------------------------------------------------------
__global__ void mk_rows_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t file, mk_row *__restrict__ rows) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const mk_hit h = hits[i];
        mk_row r;
        r.rec = h.rec;
        r.pat = h.pat;
        r.pos = h.pos;
        r.file = file;
        r._pad = 0;
        rows[i] = r;
    }
}

// ---- paired extract: the tuples of both mates in ONE list, the mate carried inside a key field ------------------
// Aho-Corasick pair order (src/cmd_extract.rs:479-537): per pair all mate-1 matches, then all mate-2 matches ->
// record' = 2 * record + mate, then the AC order.  BNDMq pair order (:542-587): per pair and pattern the mate-1
// positions, then the mate-2 positions -> position' = mate << 31 | position (mates shorter than 2 GiB), then the
// BNDMq order.
__global__ void mk_pair_mark_kernel(mk_hit *__restrict__ hits, uint64_t n, uint32_t mate, uint32_t ac) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (ac)
            hits[i].rec = 2 * hits[i].rec + mate;
        else
            hits[i].pos |= mate << 31;
    }
}

__global__ void mk_rows_pair_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t ac, mk_row *__restrict__ rows) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const mk_hit h = hits[i];
        mk_row r;
        r.rec = ac ? h.rec >> 1 : h.rec;
        r.pat = h.pat;
        r.pos = ac ? h.pos : (h.pos & 0x7FFFFFFFu);
        r.file = ac ? (uint32_t)(h.rec & 1) : (h.pos >> 31);
        r._pad = 0;
        rows[i] = r;
    }
}

// BNDMq pattern_hit_counts of a pair list in BNDMq pair order: += 1 per (pair, pattern, mate) with a hit
// (src/cmd_extract.rs:575-584: once for mate 1, once more for mate 2)
__global__ void mk_count_pair_heads_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t *__restrict__ counts, uint32_t n_bins) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const mk_hit h = hits[i];
        bool head = i == 0;
        if (!head) {
            const mk_hit p = hits[i - 1];
            head = p.rec != h.rec || p.pat != h.pat || (p.pos >> 31) != (h.pos >> 31);
        }
        if (head && h.pat < n_bins) atomicAdd(&counts[h.pat], 1u);
    }
}

}  // namespace

void launch_pair_mark(mk_hit *d_hits, uint64_t n, uint32_t mate, bool ac, hipStream_t st) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(mk_pair_mark_kernel, dim3(blocks), dim3(256), 0, st, d_hits, n, mate, ac ? 1u : 0u);
}

void launch_rows_pair(const mk_hit *d_hits, uint64_t n, bool ac, mk_row *d_rows, hipStream_t st) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(mk_rows_pair_kernel, dim3(blocks), dim3(256), 0, st, d_hits, n, ac ? 1u : 0u, d_rows);
}

void launch_count_pair_heads(const mk_hit *d_hits, uint64_t n, uint32_t *d_counts, uint32_t n_bins, hipStream_t st) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(mk_count_pair_heads_kernel, dim3(blocks), dim3(256), 0, st, d_hits, n, d_counts, n_bins);
}

// tuples in set order -> d_found_pat (distinct patterns per record, ascending), d_found_off[n_rec + 1], *d_total.
// d_tile: scratch of max(ceil(n / 4096), ceil((n_rec + 1) / 4096)) * 8 bytes.
void launch_pattern_sets(const mk_hit *d_hits, uint64_t n, uint64_t n_rec, uint32_t *d_found_pat, unsigned long long *d_found_off,
                         unsigned long long *d_total, void *d_tile, hipStream_t st) {
    (void)hipMemsetAsync(d_found_off, 0, (n_rec + 1) * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st);
    if (n) {
        const uint32_t tiles = (uint32_t)((n + kSetsTile - 1) / kSetsTile);
        uint32_t *cnt = (uint32_t *)d_tile;
        hipLaunchKernelGGL(mk_sets_count_kernel, dim3(tiles), dim3(kSetsThreads), 0, st, d_hits, n, cnt);
        hipLaunchKernelGGL(mk_sets_scan_kernel, dim3(1), dim3(kSetsThreads), 0, st, cnt, tiles, d_total);
        hipLaunchKernelGGL(mk_sets_emit_kernel, dim3(tiles), dim3(kSetsThreads), 0, st, d_hits, n, cnt, d_found_pat, d_found_off, n_rec);
    }
    const uint64_t m = n_rec + 1;
    const uint32_t tiles = (uint32_t)((m + kSetsTile - 1) / kSetsTile);
    unsigned long long *tmax = (unsigned long long *)d_tile;
    hipLaunchKernelGGL(mk_prefmax_tile_kernel, dim3(tiles), dim3(kSetsThreads), 0, st, d_found_off, m, tmax);
    hipLaunchKernelGGL(mk_prefmax_scan_kernel, dim3(1), dim3(kSetsThreads), 0, st, tmax, tiles);
    hipLaunchKernelGGL(mk_prefmax_apply_kernel, dim3(tiles), dim3(kSetsThreads), 0, st, d_found_off, m, tmax);
}

void launch_count_u32(const uint32_t *d_list, uint64_t n, uint32_t *d_counts, uint32_t n_bins, hipStream_t st) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(mk_count_u32_kernel, dim3(blocks), dim3(256), 0, st, d_list, n, d_counts, n_bins);
}

void launch_rows(const mk_hit *d_hits, uint64_t n, uint32_t file, mk_row *d_rows, hipStream_t st) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(mk_rows_kernel, dim3(blocks), dim3(256), 0, st, d_hits, n, file, d_rows);
}

}  // namespace mk
