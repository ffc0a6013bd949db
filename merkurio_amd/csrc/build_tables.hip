// build_tables.hip -- compiles a pattern set into its filter images and exact table ON THE DEVICE (SURVEY.md §8 f-4:
// pattern-set compilation at scale; the reference builds its matcher at src/cmd_extract.rs:259-277).
//
// Until r04 one host thread walked patterns x strides: pack the q-gram, hash, set the filter bits, linear-probe the
// 32-byte buckets of a table that is far larger than any host cache (128 MiB for 500 k 21-mers) -- 0.43 s for 4 M
// entries, 150x the shard scan it feeds, once per handle (per device of --gpus N).  Here one lane does that for one
// (pattern, offset) entry: filter bits with atomicOr, a table slot claimed with atomicCAS on its pattern word.  The
// layout of a bucket then depends on the order the lanes arrive in, the SET of entries a lookup sees does not:
//   * an entry lives in the first bucket from its home bucket on that had a free slot when it arrived;
//   * every bucket it passed was full at that moment and carries the overflow flag from then on (set by the passing
//     entry itself before it moves on), so a lookup that reaches a full bucket always walks on -- the invariant the
//     scan kernel's probe_chain relies on (scan_kernel_impl.hpp), whatever the arrival order.
#include <algorithm>

#include "scan_kernel.h"

namespace mk {

__device__ __forceinline__ uint64_t pack_qgram_dev(const uint8_t *__restrict__ p, uint32_t q) {
    uint64_t k = 0;
    for (uint32_t i = 0; i < q; ++i) k |= (uint64_t)code2(p[i]) << (2 * i);
    return k;
}

// table[0, n_slots) = {fp 0, empty}
__global__ void mk_table_clear_kernel(uint2 *__restrict__ table, uint64_t n_slots) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x)
        table[i] = make_uint2(0u, kEmptyPat);
}

__global__ __launch_bounds__(256) void mk_build_tables_kernel(const BuildParams B) {
    // one lane per (pattern, offset): 16 offsets per pattern, those at or beyond the pattern's stride idle
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t pi64 = idx >> 4;
    const uint32_t o = (uint32_t)idx & 15u;
    if (pi64 >= B.n_pat) return;
    const uint32_t pi = (uint32_t)pi64;
    const uint32_t a = B.pat_off[pi], len = B.pat_off[pi + 1] - a;
    const bool two = B.split != 0;
    const bool is_short = two && len < B.split;
    const uint32_t S = is_short ? B.S2 : B.S, q = is_short ? B.q2 : B.q;
    if (o >= S) return;
    const uint8_t *p = B.pat_bytes + a;
    const uint64_t key = pack_qgram_dev(p + o, q);
    uint32_t h = bloom_hash((uint32_t)key, (uint32_t)(key >> 32));
    // level-2 fingerprint: the filter hash, or (context kernels) that hash mixed with the pattern bases around the
    // q-gram; two classes: bit 0 says which class's samples may verify the entry (filter.hpp)
    uint32_t fp = B.gf_ctx ? ctx_fp(h, ctx_of_pattern(p, o, q, S) & ctx_mask(o, S)) : h;
    if (two) fp = is_short ? short_fp((uint32_t)key) : main_fp(h);
    if (is_short) {  // level 1 of the short class: the table over its packed keys
        if (B.q2 <= kShortByteMaxQ)
            atomicOr(&B.short_table[key >> 2], 1u << (8 * ((uint32_t)key & 3u)));
        else
            atomicOr(&B.short_table[key >> 5], 1u << ((uint32_t)key & 31u));
    } else if (B.gbloom_blocks) {
        const size_t blk = (size_t)gbloom_block(h, B.gbloom_blocks) * 2;
        const uint32_t hb = gbloom_bits(h);
        atomicOr(&B.bloom[blk], (1u << bloom_bit_a(hb)) | (1u << bloom_bit_d(hb)));
        atomicOr(&B.bloom[blk + 1], (1u << bloom_bit_b(hb)) | (1u << bloom_bit_c(hb)));
    } else {
        const uint32_t blk = bloom_block_byte(h) >> 2;  // index of the block's low word
        atomicOr(&B.bloom[blk], 1u << bloom_bit_a(h));
        atomicOr(&B.bloom[blk + 1], (1u << bloom_bit_b(h)) | (1u << bloom_bit_c(h)));
    }
    const uint32_t val = (pi << 4) | o;
    uint32_t b = table_bucket(two ? fp : h, B.bucket_mask);
    for (;;) {  // first bucket from the home bucket on with a free slot
        TableEntry *e = B.table + (size_t)b * kBucketEntries;
        bool placed = false;
        for (uint32_t k = 0; k < kBucketEntries && !placed; ++k) {
            if (atomicCAS(&e[k].pat_off, kEmptyPat, val) == kEmptyPat) {
                e[k].fp = fp;
                placed = true;
            }
        }
        if (placed) break;
        atomicOr(&e[0].pat_off, kBucketOverflow);  // a lookup that reaches this bucket must look further
        b = (b + 1) & B.bucket_mask;
    }
}

void launch_build_tables(const BuildParams &B, uint64_t n_slots, hipStream_t st) {
    const int clear_blocks = (int)std::min<uint64_t>(8192, (n_slots + 255) / 256);
    hipLaunchKernelGGL(mk_table_clear_kernel, dim3(clear_blocks ? clear_blocks : 1), dim3(256), 0, st, reinterpret_cast<uint2 *>(B.table), n_slots);
    const uint64_t lanes = (uint64_t)B.n_pat * 16;
    hipLaunchKernelGGL(mk_build_tables_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, B);
}

}  // namespace mk
