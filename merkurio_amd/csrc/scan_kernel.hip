// scan_kernel.hip -- host-side dispatch over the gfx950 scan-kernel variants (the kernel itself is
// scan_kernel_impl.hpp, instantiated by scan_variants.hip), the flag-count kernel and the
// synthetic-workload kernels.
#include <algorithm>

#include "scan_kernel.h"

namespace mk {

// number of records with rec_flags != 0 -> counters[n_pat + MK_SUM_RECORDS_HIT]
// (flag bytes are 0 or 1; 16-byte loads over the 16-byte aligned middle, bytes at both ends)
__global__ __launch_bounds__(1024) void mk_count_flags_kernel(const uint8_t *__restrict__ flags, uint64_t n_rec,
                                                             unsigned long long *__restrict__ out) {
    const uint64_t head = std::min<uint64_t>(n_rec, (16 - ((uintptr_t)flags & 15)) & 15);
    const uint64_t n16 = (n_rec - head) / 16;
    const uint4 *__restrict__ v = reinterpret_cast<const uint4 *>(flags + head);
    unsigned long long c = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {  // four loads in flight per lane
        const uint4 a = v[i], b = v[i + stride], d = v[i + 2 * stride], e = v[i + 3 * stride];
        c += __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w) +
             __popc(d.x) + __popc(d.y) + __popc(d.z) + __popc(d.w) + __popc(e.x) + __popc(e.y) + __popc(e.z) + __popc(e.w);
    }
    for (; i < n16; i += stride) {
        const uint4 a = v[i];
        c += __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (uint64_t r = 0; r < head; ++r) c += flags[r] != 0;
        for (uint64_t r = head + n16 * 16; r < n_rec; ++r) c += flags[r] != 0;
    }
    // one atomic per block, one block per CU (a single address takes ~11 ns per atomic: 8192 wave
    // atomics cost 90 us, 1024 block atomics 11 of this kernel's 26 us)
    __shared__ unsigned long long part[16];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];
        if (t) atomicAdd(out, t);
    }
}

void launch_count_flags(const ScanParams &p, hipStream_t st) {
    const uint64_t n16 = p.n_rec / 16;
    const int blocks = (int)std::min<uint64_t>(256, std::max<uint64_t>(1, (n16 + 4095) / 4096));
    hipLaunchKernelGGL(mk_count_flags_kernel, dim3(blocks), dim3(1024), 0, st, reinterpret_cast<const uint8_t *>(p.rec_flags32),
                       p.n_rec, p.counters + p.n_pat + MK_SUM_RECORDS_HIT);
}

// ---- start of a scan: flags[0, n_words * 4) = 0 and *n_hits = 0 in one launch -------------------------
// (two hipMemsetAsync took 22 + 2 us per 100 M records; this takes ~15)
__global__ __launch_bounds__(1024) void mk_clear_kernel(uint32_t *__restrict__ flags32, uint64_t n_words,
                                                       unsigned long long *__restrict__ n_hits) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_hits = 0;
    const uint64_t head = std::min<uint64_t>(n_words, ((16 - ((uintptr_t)flags32 & 15)) & 15) / 4);  // words before 16-byte alignment
    const uint64_t n16 = (n_words - head) / 4;
    uint4 *__restrict__ v = reinterpret_cast<uint4 *>(flags32 + head);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) v[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        if (threadIdx.x < head) flags32[threadIdx.x] = 0;
        const uint64_t t = head + n16 * 4 + threadIdx.x;  // < 4 words behind the 16-byte body
        if (threadIdx.x < 4 && t < n_words) flags32[t] = 0;
    }
}

void launch_clear(uint32_t *flags32, uint64_t n_words, unsigned long long *n_hits, hipStream_t st) {
    const int blocks = (int)std::min<uint64_t>(512, std::max<uint64_t>(1, (n_words / 4 + 1023) / 1024));
    hipLaunchKernelGGL(mk_clear_kernel, dim3(blocks), dim3(1024), 0, st, flags32, n_words, n_hits);
}

// ---- end of a scan by a sparse-hit kernel: the records the scan waves listed get their flag bytes -----
// One wave here per scan wave's list; nothing waits for these scattered byte stores.
__global__ __launch_bounds__(1024) void mk_flag_scatter_kernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ counts,
                                                              uint32_t cap, uint32_t n_waves, uint8_t *__restrict__ flags) {
    const uint32_t w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (w >= n_waves) return;
    const uint32_t n = counts[w] < cap ? counts[w] : cap;
    const uint32_t *__restrict__ l = list + (uint64_t)w * cap;
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) flags[l[i]] = 1;
}

void launch_flag_scatter(const uint32_t *flag_list, const uint32_t *flag_counts, uint32_t flag_cap, uint32_t n_waves, uint8_t *rec_flags,
                         hipStream_t st) {
    hipLaunchKernelGGL(mk_flag_scatter_kernel, dim3((n_waves + 15) / 16), dim3(1024), 0, st, flag_list, flag_counts, flag_cap, n_waves,
                       rec_flags);
}

// ---- occurrences per pattern from the emitted tuples (MK_MODE_HITS with a counter vector) ------------
// counters[pat] += 1 for every stored tuple.  Workgroups take slabs of >= 64 Ki tuples; a slab is
// histogrammed in LDS (pattern sets up to 36 Ki patterns: 144 KiB of u32 bins) and flushed with one
// atomic per non-empty bin, so a launch with few tuples costs a few microseconds and a launch where
// every read hits (100 M tuples) reads 1.6 GB once.  Larger pattern sets go straight to global atomics
// (their counters spread over megabytes: no hot line).
constexpr uint32_t kHistLdsBins = 36864;
__global__ __launch_bounds__(1024) void mk_hist_hits_kernel(const mk_hit *__restrict__ hits, const unsigned long long *__restrict__ n_hits,
                                                            uint64_t cap, unsigned long long *__restrict__ counters, uint32_t n_pat) {
    __shared__ uint32_t bins[kHistLdsBins];
    const uint64_t n = std::min<uint64_t>(*n_hits, cap);
    const uint64_t per = std::max<uint64_t>(65536, (n + gridDim.x - 1) / gridDim.x);
    const uint64_t lo = (uint64_t)blockIdx.x * per;
    if (lo >= n) return;  // block-uniform
    const uint64_t hi = std::min(n, lo + per);
    const bool in_lds = n_pat <= kHistLdsBins;
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < n_pat; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t pat = hits[i].pat;
        if (pat >= n_pat) continue;
        if (in_lds)
            atomicAdd(&bins[pat], 1u);
        else
            atomicAdd(&counters[pat], 1ull);
    }
    if (in_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_pat; i += blockDim.x)
            if (bins[i]) atomicAdd(&counters[i], (unsigned long long)bins[i]);
    }
}

void launch_hist_hits(const ScanParams &p, int grid_blocks, hipStream_t st) {
    hipLaunchKernelGGL(mk_hist_hits_kernel, dim3(grid_blocks), dim3(1024), 0, st, p.hits, p.n_hits, p.hits_cap, p.counters, p.n_pat);
}

// ---- coarse record index for batches whose records differ in length -------------------------------------
__global__ void mk_rec_index_kernel(const uint64_t *__restrict__ rec_off, uint64_t n_rec, uint64_t n_entries, uint32_t *__restrict__ out) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_entries; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = k << kRecIndexShift;
        uint64_t lo = 0, hi = n_rec;  // invariant: rec_off[lo] <= p (rec_off[0] = 0), and (hi == n_rec or rec_off[hi] > p)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rec_off[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        out[k] = (uint32_t)lo;
    }
}

void launch_rec_index(const uint64_t *rec_off, uint64_t n_rec, uint64_t n_bytes, uint32_t *rec_index, hipStream_t stream) {
    const uint64_t n_entries = (n_bytes >> kRecIndexShift) + 2;
    const int blocks = (int)std::min<uint64_t>(2048, (n_entries + 255) / 256);
    hipLaunchKernelGGL(mk_rec_index_kernel, dim3(blocks), dim3(256), 0, stream, rec_off, n_rec, n_entries, rec_index);
}

// dst[i] += src[i]: counter vectors of two handles that share a device (mk_reduce_counters)
__global__ void mk_add_u64_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, size_t len) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

void launch_add_u64(unsigned long long *dst, const unsigned long long *src, size_t len, hipStream_t stream) {
    const int blocks = (int)std::min<size_t>(1024, (len + 255) / 256);
    hipLaunchKernelGGL(mk_add_u64_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, dst, src, len);
}

// the kernel variants are instantiated in groups by scan_variants.hip (one translation unit per
// group, compiled in parallel); launch_variant<...> is the host-side launcher of one of them
template <int S, int QC, bool EMIT, bool GF, int FL = 1, int MC = 0>
static const char *launch_one(const ScanParams &p, int grid, hipStream_t st, const char *name) {
    launch_variant<S, QC, EMIT, GF, FL, MC>(p, grid, st);
    return name;
}

#define MK_VARIANT(S_, QC_, GF_)                                                                                   \
    return emit ? launch_one<S_, QC_, true, GF_>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true," #GF_ ">") \
                : launch_one<S_, QC_, false, GF_>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false," #GF_ ">")

#define MK_VARIANT_PLAIN(S_, QC_)                                                                                       \
    return emit ? launch_one<S_, QC_, true, false, 0>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,false,plain>") \
                : launch_one<S_, QC_, false, false, 0>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,false,plain>")
#define MK_VARIANT_MC(S_, QC_)                                                                                              \
    return emit ? launch_one<S_, QC_, true, false, 1, 1>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,false,2-class>") \
                : launch_one<S_, QC_, false, false, 1, 1>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,false,2-class>")
// the k-mer families with a byte-table short class of stride 2 / 4 / 8: that stride compiled in
#define MK_VARIANT_MCS(S_, QC_)                                                                                                         \
    do {                                                                                                                                \
        if (p.short_bytes && p.s2 == 2)                                                                                                 \
            return emit ? launch_one<S_, QC_, true, false, 1, 2>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,false,s2=2,2-class>") \
                        : launch_one<S_, QC_, false, false, 1, 2>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,false,s2=2,2-class>"); \
        if (p.short_bytes && p.s2 == 4)                                                                                                 \
            return emit ? launch_one<S_, QC_, true, false, 1, 4>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,false,s2=4,2-class>") \
                        : launch_one<S_, QC_, false, false, 1, 4>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,false,s2=4,2-class>"); \
        if (p.short_bytes && p.s2 == 8)                                                                                                 \
            return emit ? launch_one<S_, QC_, true, false, 1, 8>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,false,s2=8,2-class>") \
                        : launch_one<S_, QC_, false, false, 1, 8>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,false,s2=8,2-class>"); \
        MK_VARIANT_MC(S_, QC_);                                                                                                         \
    } while (0)

const char *launch_scan(const ScanParams &p, int S, bool wide, bool emit, bool global_filter, int flavour, int grid_blocks,
                        hipStream_t stream) {
    const bool plain_loads = flavour == 0;
#define MK_VARIANT_MC_GF(S_, QC_)                                                                                          \
    return emit ? launch_one<S_, QC_, true, true, 1, 1>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",true,true,2-class>") \
                : launch_one<S_, QC_, false, true, 1, 1>(p, grid_blocks, stream, "mk_scan_kernel<" #S_ "," #QC_ ",false,true,2-class>")
    if (p.s2 && global_filter) {  // two length classes, main filter in global memory: the runtime-q kernels (no context fingerprints)
        if (wide) switch (S) {
                case 2: MK_VARIANT_MC_GF(2, -1);
                case 4: MK_VARIANT_MC_GF(4, -1);
                case 8: MK_VARIANT_MC_GF(8, -1);
                case 16: MK_VARIANT_MC_GF(16, -1);
                default: return nullptr;
            }
        switch (S) {
            case 2: MK_VARIANT_MC_GF(2, 0);
            case 4: MK_VARIANT_MC_GF(4, 0);
            case 8: MK_VARIANT_MC_GF(8, 0);
            case 16: MK_VARIANT_MC_GF(16, 0);
            default: return nullptr;
        }
    }
#undef MK_VARIANT_MC_GF
    if (p.s2) {  // two length classes (filter.hpp): main filter in LDS + the short class's table; one load flavour
        if (S == 16 && p.q == 16) MK_VARIANT_MCS(16, 16);
        if (S == 8 && p.q == 24) MK_VARIANT_MCS(8, 24);
        if (S == 4 && p.q == 28) MK_VARIANT_MCS(4, 28);
        if (S == 4 && p.q == 18) MK_VARIANT_MCS(4, 18);
        if (wide) switch (S) {
                case 2: MK_VARIANT_MC(2, -1);
                case 4: MK_VARIANT_MC(4, -1);
                case 8: MK_VARIANT_MC(8, -1);
                case 16: MK_VARIANT_MC(16, -1);
                default: return nullptr;
            }
        switch (S) {
            case 2: MK_VARIANT_MC(2, 0);
            case 4: MK_VARIANT_MC(4, 0);
            case 8: MK_VARIANT_MC(8, 0);
            case 16: MK_VARIANT_MC(16, 0);
            default: return nullptr;
        }
    }
    if (global_filter) {  // large pattern sets: filter blocks in global memory
        if (S == 8 && p.q == 14) MK_VARIANT(8, 14, true);  // 21-mers
        if (S == 4 && p.q == 18) MK_VARIANT(4, 18, true);
        if (S == 8 && p.q == 24) MK_VARIANT(8, 24, true);  // 31-mers
        if (wide) switch (S) {
                case 1: MK_VARIANT(1, -1, true);
                case 2: MK_VARIANT(2, -1, true);
                case 4: MK_VARIANT(4, -1, true);
                case 8: MK_VARIANT(8, -1, true);
                case 16: MK_VARIANT(16, -1, true);
                default: return nullptr;
            }
        switch (S) {
            case 1: MK_VARIANT(1, 0, true);
            case 2: MK_VARIANT(2, 0, true);
            case 4: MK_VARIANT(4, 0, true);
            case 8: MK_VARIANT(8, 0, true);
            case 16: MK_VARIANT(16, 0, true);
            default: return nullptr;
        }
    }
    // k-mer sizes with their own kernels (q fixed at compile time): the 31-mer family
    // (q = 32 - S) and the 21-mer family (q = 22 - S, S <= 4)
    if (plain_loads) {  // hit-dense text: cacheable stream loads, 16-byte loads at level 3
        if (S == 16 && p.q == 16) MK_VARIANT_PLAIN(16, 16);
        if (S == 8 && p.q == 24) MK_VARIANT_PLAIN(8, 24);
        if (S == 4 && p.q == 28) MK_VARIANT_PLAIN(4, 28);
        if (S == 4 && p.q == 18) MK_VARIANT_PLAIN(4, 18);
        if (wide) switch (S) {
                case 1: MK_VARIANT_PLAIN(1, -1);
                case 2: MK_VARIANT_PLAIN(2, -1);
                case 4: MK_VARIANT_PLAIN(4, -1);
                case 8: MK_VARIANT_PLAIN(8, -1);
                case 16: MK_VARIANT_PLAIN(16, -1);
                default: return nullptr;
            }
        switch (S) {
            case 1: MK_VARIANT_PLAIN(1, 0);
            case 2: MK_VARIANT_PLAIN(2, 0);
            case 4: MK_VARIANT_PLAIN(4, 0);
            case 8: MK_VARIANT_PLAIN(8, 0);
            case 16: MK_VARIANT_PLAIN(16, 0);
            default: return nullptr;
        }
    }
    if (S == 16 && p.q == 16) MK_VARIANT(16, 16, false);
    if (S == 8 && p.q == 24) MK_VARIANT(8, 24, false);
    if (S == 4 && p.q == 28) MK_VARIANT(4, 28, false);
    if (S == 4 && p.q == 18) MK_VARIANT(4, 18, false);
    // everything else: runtime q, narrow (q <= 16) or wide keys
    if (wide) switch (S) {
            case 1: MK_VARIANT(1, -1, false);
            case 2: MK_VARIANT(2, -1, false);
            case 4: MK_VARIANT(4, -1, false);
            case 8: MK_VARIANT(8, -1, false);
            case 16: MK_VARIANT(16, -1, false);
            default: return nullptr;
        }
    switch (S) {
        case 1: MK_VARIANT(1, 0, false);
        case 2: MK_VARIANT(2, 0, false);
        case 4: MK_VARIANT(4, 0, false);
        case 8: MK_VARIANT(8, 0, false);
        case 16: MK_VARIANT(16, 0, false);
        default: return nullptr;
    }
}
#undef MK_VARIANT
#undef MK_VARIANT_PLAIN
#undef MK_VARIANT_MC
#undef MK_VARIANT_MCS

// ---- synthetic reads (bench / full-size parity tests) ----------------------------------
// byte0 = global position of seq[0] in the synthetic stream (a multiple of 32)
__global__ void mk_synth_fill_kernel(uint64_t seed, uint64_t byte0, uint64_t n_bytes, uint8_t *__restrict__ seq) {
    const uint64_t n16 = (n_bytes + 15) / 16;
    const uint64_t blk0 = byte0 >> 5;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t bits = synth_block(seed, blk0 + (i >> 1)) >> (32 * (i & 1));
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)synth_base(bits, 4 * k + b) << (8 * b);
            w[k] = v;
        }
        const uint64_t pos = i * 16;
        if (pos + 16 <= n_bytes) {
            *reinterpret_cast<uint4 *>(seq + pos) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint64_t k = 0; pos + k < n_bytes; ++k) seq[pos + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        }
    }
}

__global__ void mk_synth_off_plant_kernel(uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len, uint32_t plant_every,
                                          const uint8_t *__restrict__ pat_bytes, const uint32_t *__restrict__ pat_off,
                                          uint32_t n_pat, uint8_t *__restrict__ seq, uint64_t *__restrict__ seq_off) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rec; r += (uint64_t)gridDim.x * blockDim.x) {
        seq_off[r] = r * read_len;
        if (r == n_rec || plant_every == 0) continue;
        const uint64_t gr = rec0 + r;  // global record index: a shard of a job equals that slice of the whole job
        const uint64_t h = synth_rec_hash(seed, gr);
        if (h % plant_every != 0) continue;
        const uint32_t pat = (uint32_t)((gr * 2654435761ull) % n_pat);
        const uint32_t a = pat_off[pat], len = pat_off[pat + 1] - a;
        if (len > read_len) continue;
        const uint32_t o = (uint32_t)((h >> 32) % (read_len - len + 1));
        for (uint32_t i = 0; i < len; ++i) seq[r * read_len + o + i] = pat_bytes[a + i];
    }
}

void launch_synth(uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len, uint32_t plant_every, const uint8_t *d_pat_bytes,
                  const uint32_t *d_pat_off, uint32_t n_pat, uint8_t *d_seq, uint64_t *d_seq_off, hipStream_t stream) {
    const uint64_t n_bytes = n_rec * read_len;
    hipLaunchKernelGGL(mk_synth_fill_kernel, dim3(2048), dim3(256), 0, stream, seed, rec0 * read_len, n_bytes, d_seq);
    hipLaunchKernelGGL(mk_synth_off_plant_kernel, dim3(2048), dim3(256), 0, stream, seed, rec0, n_rec, read_len, plant_every,
                       d_pat_bytes, d_pat_off, n_pat, d_seq, d_seq_off);
}

}  // namespace mk
