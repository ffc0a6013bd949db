// order_hits.hip -- the reference's emission order, restored on the device by hand-written gfx950 kernels.
//
// The scan kernel writes (record, pattern, position) tuples as its waves find them.  The reference
// emits them in the order of its search loops:
//   Aho-Corasick  find_overlapping_iter per record (src/cmd_extract.rs:332-351, src/cmd_tag.rs:393-414):
//                 end ascending; at one end the longer pattern (smaller start) first, then pattern id
//   BNDMq         one pattern after the other per record (src/cmd_extract.rs:365-384): pattern-major,
//                 positions ascending
// A batch where every read hits carries 10^7..10^8 tuples.  A comparison sort over 16-byte tuples makes
// log2(n / block) ~ 17 passes over them (rocPRIM's merge sort: 41 ms per 10^8 tuples, six times the scan
// that produced them).  The order is instead restored in TWO passes over the tuples plus one over 8-byte
// keys:
//
//   every tuple is a triple (record, A, B) compared field by field --
//       AC:    A = end = pos + len(pattern),  B = rank of the pattern in (length descending, index
//              ascending) order (identity for a k-mer set: all lengths equal) -- "longer pattern first,
//              then pattern id" at one end IS that rank;
//       BNDMq: A = pattern, B = pos;
//   G = (record - base) << bits(A) | A;  bin = G >> shift;  key = (G mod 2^shift) << bits(B) | B   (8 bytes)
//   (second attempt on skewed batches: the bins are the top bits of the WHOLE triple -- with shift = 0 the top b_hi
//   bits of B go into the bin index as well, key = the low bits of B).
//
//   1. mk_order_hist_kernel     reads the tuples once: tuples per bin (LDS histogram per workgroup, one
//                               global atomic per non-empty bin) and the maxima of record, A and B;
//      mk_order_scan_kernel     bin starts (one workgroup) and the largest bin; the host reads 48 bytes
//                               back and fixes the field widths and the leaf geometry -- the only host
//                               round trip (the caller has just read the tuple count the same way);
//   2. mk_order_scatter_kernel  reads the tuples again, packs each into its 8-byte key and stores it in
//                               its bin's range of the scratch array: per 4096-tuple tile the ranks inside
//                               a bin come from LDS atomics and ONE global atomic per (tile, bin touched)
//                               reserves the range -- the scan kernel emits tuples in runs of ~1000 per
//                               wave, a tile touches a handful of bins;
//   3. mk_order_leaf_kernel     one workgroup per bin: keys -> LDS, bitonic network with 16 keys per
//                               lane in registers (four compare-exchange stages per LDS round trip, the
//                               padded layout keeps every round bank-conflict-free), keys decoded back
//                               into tuples and written to their final place in the caller's array.
//
// Bins are 2^k consecutive records with ~2048 tuples on average and at most 16384 (128 KiB of keys in
// LDS).  Skew: when a bin overflows (few huge records: a genome FASTA; hits clustered in one stretch of
// the batch; one pattern all over one chromosome) the histogram is taken again on the top bits of the whole
// (record, A, B) triple instead of the record alone;
// if that overflows too, or the three fields do not fit 64 bits, the caller falls back to the library
// merge sort (order_hits_fallback.hip) -- correctness never depends on the distribution.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "scan_kernel.h"

namespace mk {

namespace {

constexpr int kOrderThreads = 1024;
constexpr int kLogKpt = 4, kKpt = 1 << kLogKpt;  // keys per lane in the leaf network

__device__ __forceinline__ uint32_t pattern_len(const OrderKey &L, uint32_t pat) {
    return L.uniform_len ? L.uniform_len : L.pat_off[pat + 1] - L.pat_off[pat];
}

// (record, A, B) of a tuple
__device__ __forceinline__ void tuple_fields(const OrderKey &L, const uint4 h, uint64_t &rec, uint64_t &a, uint64_t &b) {
    rec = ((uint64_t)h.y << 32) | h.x;
    const uint32_t pat = h.z, pos = h.w;
    if (L.ac) {
        a = (uint64_t)pos + pattern_len(L, pat);
        b = L.rank ? L.rank[pat] : pat;
    } else {
        a = pat;
        b = pos;
    }
}

__device__ __forceinline__ uint64_t wave_max(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// ---- pass 1: tuples per bin + field maxima ---------------------------------------------------------------
// bits_a == 0: bin = record >> shift (first attempt: the width of A is not known yet)
__global__ __launch_bounds__(kOrderThreads) void mk_order_hist_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t *__restrict__ g_cnt,
                                                                       unsigned long long *__restrict__ stats, const OrderKey L) {
    extern __shared__ uint32_t lds_cnt[];
    __shared__ unsigned long long red[4];
    for (uint32_t i = threadIdx.x; i < L.n_bins; i += kOrderThreads) lds_cnt[i] = 0;
    if (threadIdx.x < 4) red[threadIdx.x] = 0;
    __syncthreads();
    // a contiguous slab per workgroup: consecutive tuples share bins, few bins to flush
    uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + kOrderThreads - 1) / kOrderThreads * kOrderThreads;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = std::min<uint64_t>(n, lo + per);
    uint64_t mr = 0, ma = 0, mb = 0, mn = 0;  // mn = max of ~record: the smallest record, complemented
    const uint4 *__restrict__ hv = reinterpret_cast<const uint4 *>(hits);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kOrderThreads) {
        uint64_t rec, a, b;
        tuple_fields(L, hv[i], rec, a, b);
        mr = rec > mr ? rec : mr;
        mn = ~rec > mn ? ~rec : mn;
        rec -= L.rec_base;
        ma = a > ma ? a : ma;
        mb = b > mb ? b : mb;
        const uint64_t g = L.bits_a ? ((rec << L.bits_a) | a) : rec;
        const uint64_t d = ((g >> L.shift) << L.b_hi) | (b >> L.b_lo);  // b_hi == 0: b_lo >= the width of B, the second term is 0
        atomicAdd(&lds_cnt[d < L.n_bins ? (uint32_t)d : L.n_bins - 1], 1u);
    }
    mr = wave_max(mr);
    ma = wave_max(ma);
    mb = wave_max(mb);
    mn = wave_max(mn);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&red[3], (unsigned long long)mn);
        atomicMax(&red[0], (unsigned long long)mr);
        atomicMax(&red[1], (unsigned long long)ma);
        atomicMax(&red[2], (unsigned long long)mb);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < L.n_bins; i += kOrderThreads) {
        const uint32_t c = lds_cnt[i];
        if (c) atomicAdd(&g_cnt[i], c);
    }
    if (threadIdx.x < 3 && red[threadIdx.x]) atomicMax(&stats[threadIdx.x], red[threadIdx.x]);
    if (threadIdx.x == 3 && red[3]) atomicMax(&stats[4], red[3]);  // stats[3] is the largest bin (scan kernel)
}

// ---- bin starts (exclusive prefix sum over <= 32768 bins, one workgroup) and the largest bin -------------
// (also lists the bins above kOrderLeafSmall tuples -- big_list, their number in stats[5] -- for the second leaf launch)
__global__ __launch_bounds__(kOrderThreads) void mk_order_scan_kernel(const uint32_t *__restrict__ g_cnt, uint32_t n_bins, uint32_t *__restrict__ bin_start,
                                                                       uint32_t *__restrict__ cursor, unsigned long long *__restrict__ stats,
                                                                       uint32_t *__restrict__ big_list) {
    __shared__ uint32_t part[kOrderThreads / 64];
    __shared__ uint32_t wmax[kOrderThreads / 64];
    const uint32_t per = (n_bins + kOrderThreads - 1) / kOrderThreads;
    const uint32_t lo = std::min(n_bins, threadIdx.x * per), hi = std::min(n_bins, lo + per);
    uint32_t sum = 0, mx = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = g_cnt[i];
        sum += c;
        mx = c > mx ? c : mx;
    }
    // exclusive prefix of the per-lane sums: inside a wave by shuffles, across the 16 waves through LDS
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((threadIdx.x & 63) >= (uint32_t)o) incl += v;
    }
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kOrderThreads / 64; ++w) {
        const uint32_t t = part[w];
        if (w < (threadIdx.x >> 6)) before += t;
        total += t;
    }
    uint32_t run = before + incl - sum;  // exclusive
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = g_cnt[i];
        bin_start[i] = run;
        cursor[i] = run;
        run += c;
        if (c > kOrderLeafSmall) big_list[atomicAdd(&stats[5], 1ull)] = i;  // rare
    }
    if (threadIdx.x == 0) bin_start[n_bins] = total;
    mx = (uint32_t)wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (int w = 0; w < kOrderThreads / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
        stats[3] = m;
    }
}

// ---- pass 2: tuples -> 8-byte keys in their bin's range --------------------------------------------------
constexpr int kScatterPer = 4;  // tuples per lane and tile
__global__ __launch_bounds__(kOrderThreads) void mk_order_scatter_kernel(const mk_hit *__restrict__ hits, uint64_t n, uint32_t *__restrict__ cursor,
                                                                          uint64_t *__restrict__ keys, const OrderKey L) {
    extern __shared__ uint32_t lds_cnt[];
    for (uint32_t i = threadIdx.x; i < L.n_bins; i += kOrderThreads) lds_cnt[i] = 0;
    __syncthreads();
    const uint4 *__restrict__ hv = reinterpret_cast<const uint4 *>(hits);
    const uint64_t low_mask = (1ull << L.shift) - 1ull;  // shift <= 63
    const uint64_t b_lo_mask = (1ull << L.b_lo) - 1ull;  // b_lo = bits of B kept in the key (<= 63)
    constexpr uint64_t kTile = (uint64_t)kOrderThreads * kScatterPer;
    for (uint64_t base = (uint64_t)blockIdx.x * kTile; base < n; base += (uint64_t)gridDim.x * kTile) {
        uint64_t key[kScatterPer];
        uint32_t bin[kScatterPer], rk[kScatterPer];
#pragma unroll
        for (int k = 0; k < kScatterPer; ++k) {
            const uint64_t i = base + (uint64_t)k * kOrderThreads + threadIdx.x;
            bin[k] = 0xFFFFFFFFu;
            if (i < n) {
                uint64_t rec, a, b;
                tuple_fields(L, hv[i], rec, a, b);
                const uint64_t g = ((rec - L.rec_base) << L.bits_a) | a;
                const uint64_t d = ((g >> L.shift) << L.b_hi) | (b >> L.b_lo);
                bin[k] = d < L.n_bins ? (uint32_t)d : L.n_bins - 1;
                key[k] = ((g & low_mask) << L.b_lo) | (b & b_lo_mask);
                rk[k] = atomicAdd(&lds_cnt[bin[k]], 1u);  // rank inside (tile, bin)
            }
        }
        __syncthreads();
        // the first arrival of a bin reserves the tile's range in that bin: one global atomic per (tile, bin)
#pragma unroll
        for (int k = 0; k < kScatterPer; ++k)
            if (bin[k] != 0xFFFFFFFFu && rk[k] == 0) lds_cnt[bin[k]] = atomicAdd(&cursor[bin[k]], lds_cnt[bin[k]]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScatterPer; ++k)
            if (bin[k] != 0xFFFFFFFFu) keys[(uint64_t)lds_cnt[bin[k]] + rk[k]] = key[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScatterPer; ++k)
            if (bin[k] != 0xFFFFFFFFu && rk[k] == 0) lds_cnt[bin[k]] = 0;
        __syncthreads();
    }
}

// ---- pass 3: one workgroup sorts one bin in LDS -----------------------------------------------------------
// Bitonic network over m = 2^logm >= 16 keys (the bin padded with all-ones keys), 16 keys per lane.  A round
// loads the 16 keys whose indices differ in four consecutive bits [be, be+4), runs up to four
// compare-exchange stages on them in registers and stores them back: ceil(p / 4) LDS round trips for phase p
// instead of p.  Element i lives at LDS slot i + (i >> 4): with that padding the 32 lanes of a half-wave hit
// 32 different bank pairs in every round (simulated for every be; at most one 2-way conflict).
__device__ __forceinline__ uint32_t padi(uint32_t i) { return i + (i >> 4); }

template <int JR>
__device__ __forceinline__ void leaf_stage(uint64_t (&v)[kKpt], uint32_t dir_t, uint32_t dir_rmask) {
#pragma unroll
    for (int r = 0; r < kKpt; ++r) {
        if (r & (1 << JR)) continue;
        const uint64_t a = v[r], b = v[r | (1 << JR)];
        const bool desc = (dir_t | (((uint32_t)r & dir_rmask) ? 1u : 0u)) != 0;  // bit p of the element index
        const bool sw = (a > b) != desc;
        v[r] = sw ? b : a;
        v[r | (1 << JR)] = sw ? a : b;
    }
}

// One launch serves the bins with min_cnt < tuples <= max_cnt (its geometry is sized for max_cnt): a batch whose
// largest bin is just past a power of two does not make every bin pay for the larger workgroup.
__global__ __launch_bounds__(kOrderThreads) void mk_order_leaf_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ bin_start,
                                                                       mk_hit *__restrict__ out, const OrderKey L, uint32_t min_cnt, uint32_t max_cnt,
                                                                       const uint32_t *__restrict__ bin_list) {
    extern __shared__ uint64_t sk[];
    const uint32_t bin = bin_list ? bin_list[blockIdx.x] : blockIdx.x;
    const uint32_t lo = bin_start[bin], cnt = bin_start[bin + 1] - lo;
    if (cnt <= min_cnt || cnt > max_cnt) return;  // workgroup-uniform (empty bins: min_cnt >= 0)
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    uint32_t logm = cnt > 1 ? 32u - (uint32_t)__builtin_clz(cnt - 1) : 0u;
    if (logm < kLogKpt) logm = kLogKpt;
    const uint32_t m = 1u << logm;  // <= T * 16 (the host sizes T from the largest bin)
    for (uint32_t i = tid; i < m; i += T) sk[padi(i)] = i < cnt ? keys[(uint64_t)lo + i] : ~0ull;
    __syncthreads();
    const uint32_t nact = m >> kLogKpt;
    uint64_t v[kKpt];
    if (tid < nact) {  // phases 1..4 on 16 consecutive keys: one round trip
        const uint32_t base = tid * kKpt + tid;  // padi(16 tid + r) = 17 tid + r
#pragma unroll
        for (int r = 0; r < kKpt; ++r) v[r] = sk[base + r];
        leaf_stage<0>(v, 0, 2);
        leaf_stage<1>(v, 0, 4);
        leaf_stage<0>(v, 0, 4);
        leaf_stage<2>(v, 0, 8);
        leaf_stage<1>(v, 0, 8);
        leaf_stage<0>(v, 0, 8);
        const uint32_t d4 = logm > kLogKpt ? (tid & 1u) : 0u;
        leaf_stage<3>(v, d4, 0);
        leaf_stage<2>(v, d4, 0);
        leaf_stage<1>(v, d4, 0);
        leaf_stage<0>(v, d4, 0);
#pragma unroll
        for (int r = 0; r < kKpt; ++r) sk[base + r] = v[r];
    }
    __syncthreads();
    for (uint32_t p = kLogKpt + 1; p <= logm; ++p) {
        for (int jhi = (int)p - 1; jhi >= 0;) {
            const int nb = ((jhi + 1) & 3) ? ((jhi + 1) & 3) : 4;  // the rounds below this one take four stages each
            const int jlo = jhi - nb + 1;
            const uint32_t be = std::min<uint32_t>((uint32_t)jlo, logm - kLogKpt);  // register index = index bits [be, be+4)
            const int jr_hi = jhi - (int)be, jr_lo = jlo - (int)be;
            if (tid < nact) {
                const uint32_t p_rel = p - be;
                const uint32_t dir_t = (p < logm && p_rel >= (uint32_t)kLogKpt) ? ((tid >> (p - kLogKpt)) & 1u) : 0u;
                const uint32_t dir_rmask = (p < logm && p_rel < (uint32_t)kLogKpt) ? (1u << p_rel) : 0u;
                const uint32_t i0 = ((tid >> be) << (be + kLogKpt)) | (tid & ((1u << be) - 1u));
#pragma unroll
                for (int r = 0; r < kKpt; ++r) v[r] = sk[padi(i0 | ((uint32_t)r << be))];
                if (jr_hi >= 3 && jr_lo <= 3) leaf_stage<3>(v, dir_t, dir_rmask);
                if (jr_hi >= 2 && jr_lo <= 2) leaf_stage<2>(v, dir_t, dir_rmask);
                if (jr_hi >= 1 && jr_lo <= 1) leaf_stage<1>(v, dir_t, dir_rmask);
                if (jr_lo <= 0) leaf_stage<0>(v, dir_t, dir_rmask);
#pragma unroll
                for (int r = 0; r < kKpt; ++r) sk[padi(i0 | ((uint32_t)r << be))] = v[r];
            }
            __syncthreads();
            jhi = jlo - 1;
        }
    }
    // keys -> tuples, in their final place
    const uint64_t mask_blo = (1ull << L.b_lo) - 1ull, mask_a = (1ull << L.bits_a) - 1ull;  // widths <= 63
    const uint64_t bin_g = (uint64_t)bin >> L.b_hi, bin_b = (uint64_t)bin & ((1ull << L.b_hi) - 1ull);
    uint4 *__restrict__ ov = reinterpret_cast<uint4 *>(out);
    for (uint32_t i = tid; i < cnt; i += T) {
        const uint64_t k = sk[padi(i)];
        const uint64_t g = (bin_g << L.shift) | (k >> L.b_lo);
        const uint64_t b = (bin_b << L.b_lo) | (k & mask_blo), a = g & mask_a, rec = (g >> L.bits_a) + L.rec_base;
        uint32_t pat, pos;
        if (L.ac) {
            pat = L.unrank ? L.unrank[(uint32_t)b] : (uint32_t)b;
            pos = (uint32_t)a - pattern_len(L, pat);
        } else {
            pat = (uint32_t)a;
            pos = (uint32_t)b;
        }
        ov[(uint64_t)lo + i] = make_uint4((uint32_t)rec, (uint32_t)(rec >> 32), pat, pos);
    }
}

}  // namespace

hipError_t order_kernels_prepare() {
    // dynamic LDS beyond 64 KiB must be requested per kernel
    constexpr int kMax = 160 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mk_order_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMax - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(mk_order_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(mk_order_leaf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
}

void launch_order_hist(const mk_hit *d_hits, uint64_t n, const OrderKey &L, const OrderScratch &S, int num_cus, hipStream_t st) {
    const int blocks = (int)std::min<uint64_t>((uint64_t)num_cus, (n + 4095) / 4096);
    hipLaunchKernelGGL(mk_order_hist_kernel, dim3(blocks), dim3(kOrderThreads), L.n_bins * sizeof(uint32_t), st, d_hits, n, S.g_cnt, S.stats, L);
    hipLaunchKernelGGL(mk_order_scan_kernel, dim3(1), dim3(kOrderThreads), 0, st, S.g_cnt, L.n_bins, S.bin_start, S.cursor, S.stats, S.big_list);
}

void launch_order_scatter_leaf(mk_hit *d_hits, uint64_t n, const OrderKey &L, const OrderScratch &S, uint32_t max_bin, uint32_t n_big,
                               int num_cus, hipStream_t st) {
    const int blocks = (int)std::min<uint64_t>((uint64_t)num_cus, (n + 4095) / 4096);
    hipLaunchKernelGGL(mk_order_scatter_kernel, dim3(blocks), dim3(kOrderThreads), L.n_bins * sizeof(uint32_t), st, d_hits, n, S.cursor, S.keys, L);
    // leaf geometry: 16 keys per lane, 64..1024 lanes.  Bins of up to 4096 tuples (the common size: ~2048-3000 on
    // average) get 256-lane workgroups of their own launch; larger ones a second launch sized for the largest bin.
    auto leaf = [&](uint32_t min_cnt, uint32_t max_cnt, uint32_t grid, const uint32_t *bin_list) {
        uint32_t m = kKpt * 64;
        while (m < max_cnt) m <<= 1;
        const size_t lds = ((size_t)m + m / 16) * sizeof(uint64_t);
        hipLaunchKernelGGL(mk_order_leaf_kernel, dim3(grid), dim3(m / kKpt), lds, st, S.keys, S.bin_start, d_hits, L, min_cnt, max_cnt, bin_list);
    };
    if (max_bin <= kOrderLeafSmall) {
        leaf(0, max_bin, L.n_bins, nullptr);
    } else {
        leaf(0, kOrderLeafSmall, L.n_bins, nullptr);
        if (n_big) leaf(kOrderLeafSmall, max_bin, n_big, S.big_list);  // one workgroup per listed bin
    }
}

}  // namespace mk
