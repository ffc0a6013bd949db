// filter.hpp -- definitions shared by the host-side pattern compiler and the gfx950 scan kernel.
//
// The device scan does not run BNDMq or a DFA.  Both reference matchers report exactly "every
// occurrence of every pattern" (SURVEY.md §0.4), so the kernel computes that result set with a
// streaming q-gram filter + exact verification:
//
//   * 2-bit code of a byte: code(c) = (c >> 1) & 3  (A,a->0  C,c->1  T,t->2  G,g->3; any other
//     byte maps somewhere -- the filter only needs  bytes equal => codes equal,  and that
//     holds for every byte value, and for ASCII case folding too).
//   * every pattern contributes the S q-grams starting at offsets 0..S-1 of its first
//     L' = q + S - 1 bytes.  The text is sampled at positions t = 0 mod S.  An occurrence of
//     pattern P at text position p covers exactly one sampled position t in [p, p+S-1], and
//     the text q-gram at t equals P's q-gram at offset t-p, so: no false negatives, and every
//     occurrence is discovered exactly once (t-p is unique).
//   * level 1 (LDS): blocked Bloom filter over the packed q-gram keys, 3 bits per key inside
//     one 64-bit block;   level 2 (L2/HBM): open-addressing table key -> (pattern, offset);
//     level 3: byte-exact (or ASCII-case-folded) comparison against the pattern text, then
//     the record-boundary check.  Only level 3 decides; levels 1-2 may only over-approximate.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MK_HD __host__ __device__ __forceinline__
#else
#define MK_HD inline
#endif

namespace mk {

constexpr uint32_t kBloomLog2Blocks = 14;                  // 16384 blocks of 64 bits
constexpr uint32_t kBloomBlocks = 1u << kBloomLog2Blocks;
constexpr uint32_t kBloomWords = 2 * kBloomBlocks;         // 32-bit words: 128 KiB of LDS
constexpr uint32_t kBloomBytes = kBloomWords * 4;
constexpr uint32_t kEmptyPat = 0xFFFFFFFFu;
constexpr int kChunkBytes = 1024;                          // one wave-iteration: 64 lanes x 16 B
#ifndef MK_TILE_CHUNKS
#define MK_TILE_CHUNKS 31  // (A/B builds: 4 k - 1 for any k, e.g. 127 = the halo chunk carried across a run of four tiles)
#endif
constexpr int kTileChunks = MK_TILE_CHUNKS;                // scanned chunks per wave tile (+1 halo chunk = 32 loads)
static_assert((kTileChunks + 1) % 4 == 0, "a tile is a whole number of four-chunk load groups");
constexpr int kBlockThreads = 1024;                        // 16 waves, one workgroup per CU

// one exact-table slot (8 B): the 32-bit filter hash h of the q-gram key (level 2 is keyed by the
// hash level 1 has already computed, so a queued candidate is just {h, position}) and
// (pattern << 4 | offset of the q-gram inside the pattern).  Two keys with the same h only
// cost a wasted byte comparison in level 3.
struct alignas(8) TableEntry {
    uint32_t fp;       // = bloom_hash(key)
    uint32_t pat_off;  // kEmptyPat = empty slot
};
constexpr uint32_t kMaxPatterns = (1u << 27) - 2;
// The table is an array of 32-byte buckets of 4 entries, filled front to back; a lookup reads
// one whole bucket (2 x global_load_dwordx4, one memory round trip) and moves on to the next
// bucket only if this one has OVERFLOWED while the table was built (some key that hashes here or
// before was pushed past it): bit 31 of the pat_off word of the bucket's entry 0.  "Bucket full"
// alone would send 3.6 % of the lookups -- 9 of 10 wave-wide probe rounds -- into a second,
// synchronous round trip at load 0.3; the overflow flag does so for 0.8 %.
constexpr uint32_t kBucketEntries = 4;
constexpr uint32_t kBucketOverflow = 0x80000000u;
MK_HD uint32_t table_bucket(uint32_t h, uint32_t bucket_mask) { return ((h * 0x9E3779B1u) >> 6) & bucket_mask; }

MK_HD uint32_t code2(uint8_t c) { return (c >> 1) & 3u; }

// Hash for the LDS Bloom filter, from the 2q-bit packed key (bits at or above 2q are zero):
//     lo = key[0:32)   t = key[24:56)
//     h  = lo[0:24) * C1  +  t[0:24) * C2  +  (t & 0xFF000000)          (mod 2^32)
// i.e. two 24x24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24) over key bits 0..47 plus key
// bits 48..55 added in place; bits >= 56 (q > 28) do not take part.  The device variants
// specialised for one q skip whatever masking that q makes redundant (scan_kernel.hip).
MK_HD uint32_t bloom_hash(uint32_t lo, uint32_t hi) {
    const uint32_t t = (lo >> 24) | (hi << 8);
    return (lo & 0xFFFFFFu) * 0x9E3779u + (t & 0xFFFFFFu) * 0x85EBCBu + (t & 0xFF000000u);
}
// Blocked Bloom filter, 3 bits per key inside one 64-bit block (one ds_read_b64 per probe):
// block = bits [3,17) of h (byte offset h & 0x1FFF8), bit a in the block's low word, bits b
// and c in its high word.  a, b, c are the top three 5-bit groups of h: a shift instruction
// uses only the low 5 bits of its count, so h >> 27, h >> 22, h >> 17 need no masking.
MK_HD uint32_t bloom_block_byte(uint32_t h) { return h & (((kBloomBlocks - 1) << 3)); }
MK_HD uint32_t bloom_bit_a(uint32_t h) { return h >> 27; }
MK_HD uint32_t bloom_bit_b(uint32_t h) { return (h >> 22) & 31u; }
MK_HD uint32_t bloom_bit_c(uint32_t h) { return (h >> 17) & 31u; }

// Global-memory variant of the filter for pattern sets too large for a useful 128 KiB LDS
// image (more than ~96 k filter entries even at stride 1): the same 3-bit / 64-bit blocks,
// but 2^g_log2 of them in HBM (resident in L2 / Infinity Cache), sized ~32 bits per entry.
// Block index = scaled high bits of h, bit positions from a second multiply so that they do not
// correlate with the block index.
// block = floor(h * n_blocks / 2^32): one v_mul_hi, and the filter need not be a power of two in size
// (the fastest size is "as large as still stays in the 4 MiB L2 next to the text stream": r02_c5_*)
MK_HD uint32_t gbloom_block(uint32_t h, uint32_t n_blocks) { return (uint32_t)(((uint64_t)h * n_blocks) >> 32); }
MK_HD uint32_t gbloom_bits(uint32_t h) { return h * 0x9E3779B1u; }  // a,b,c,d = top four 5-bit groups
// the global filter sets FOUR bits per key (a, d in the block's low word, b, c in its high word): at the
// 6-8 bits per entry an L2-resident filter for millions of entries affords, that is a quarter fewer
// false positives than three, and a false positive there is a random HBM read
MK_HD uint32_t bloom_bit_d(uint32_t h) { return (h >> 12) & 31u; }

// ---- context fingerprints (global-filter kernels with a compile-time q: gf_has_ctx) ---------------
// With hundreds of thousands of patterns the sampled q-grams get short (500 k 21-mers: q = 14) and
// 1.5 % of all text samples are REAL q-gram matches of some pattern: each would cost level 3 a byte
// compare, i.e. three random HBM reads.  A pattern occurrence that puts its offset-o q-gram on the
// sample at t also fixes the o text bases before t and the S-1-o bases after the q-gram (every pattern
// has at least q + S - 1 bases).  The candidate carries up to 7 + 7 of them, 2-bit packed (ctx:
// bits [0,14) = bases t-7..t-1, bits [14,28) = bases t+q..t+q+6), and the level-2 fingerprint of
// entry (pattern, o) is the q-gram hash mixed with the pattern's own bases under the mask of o.
// For 21-mers at S = 8 the fingerprint covers the whole pattern: level 3 sees real occurrences only.
MK_HD bool gf_has_ctx(uint32_t S, uint32_t q) { return (S == 8 && q == 14) || (S == 4 && q == 18) || (S == 8 && q == 24); }
MK_HD uint32_t ctx_mask(uint32_t o, uint32_t S) {
    const uint32_t npre = o < 7 ? o : 7;
    const uint32_t rem = S - 1 - o;
    const uint32_t nsuf = rem < 7 ? rem : 7;
    return ((0x3FFFu << (14 - 2 * npre)) & 0x3FFFu) | (((1u << (2 * nsuf)) - 1u) << 14);
}
MK_HD uint32_t ctx_fp(uint32_t h, uint32_t ctx_masked) { return h ^ (ctx_masked * 0x85EBCA6Bu); }
// the context word of pattern p (its bytes) for the q-gram at offset o: the same layout as the text side
MK_HD uint32_t ctx_of_pattern(const uint8_t *p, uint32_t o, uint32_t q, uint32_t S) {
    const uint32_t npre = o < 7 ? o : 7;
    const uint32_t rem = S - 1 - o;
    const uint32_t nsuf = rem < 7 ? rem : 7;
    uint32_t c = 0;
    for (uint32_t i = 0; i < npre; ++i) c |= code2(p[o - npre + i]) << (14 - 2 * npre + 2 * i);
    for (uint32_t i = 0; i < nsuf; ++i) c |= code2(p[o + q + i]) << (14 + 2 * i);
    return c;
}

// ---- two length classes (matcher.cpp: plan_classes; scan_kernel_impl.hpp: MC kernels) ---------------------------
// A pattern set whose shortest pattern is much shorter than the rest would drag the whole set down to the stride
// and q-gram length that pattern admits (10 000 31-mers + one 8-mer: S = 1, q = 8 for everybody).  Such a set is split
// by length: the long patterns keep the hashed q-gram filter above, the short ones get their own stride S2 and
// q-grams of q2 <= 8 bases looked up in a plain bitmap over the 4^q2 packed keys (8 KiB of LDS).  Both classes'
// entries live in one exact table; bit 0 of an entry's fingerprint is its class, so that a sample of one class can
// never verify an entry of the other (every occurrence is still discovered exactly once).
constexpr uint32_t kShortMaxQ = 8;
constexpr uint32_t kShortByteMaxQ = 6;  // up to here the table holds a byte per key (4 KiB), above a bit per key
constexpr uint32_t kShortBitmapWords = (1u << (2 * kShortMaxQ)) / 32;  // 2048 words
// (keys have at most 16 bits: a 24-bit multiply -- full rate on the device -- is the same product)
MK_HD uint32_t short_fp(uint32_t key) { return ((key & 0xFFFFFFu) * 0x9E3779u) | 1u; }
MK_HD uint32_t main_fp(uint32_t h) { return h & ~1u; }

MK_HD uint8_t fold_ascii(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c | 0x20) : c; }

// counter-based synthetic read generator (bench / full-size parity tests)
MK_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// the 32 bases of block b (bytes 32b .. 32b+31), 2 bits each, base j at bits 2j
MK_HD uint64_t synth_block(uint64_t seed, uint64_t b) { return splitmix64(seed ^ (b * 0xD1B54A32D192ED03ull)); }
MK_HD uint8_t synth_base(uint64_t bits, uint32_t j) {
    const uint32_t lut = ('A') | ('C' << 8) | ('G' << 16) | ((uint32_t)'T' << 24);
    return (uint8_t)(lut >> (8 * ((bits >> (2 * j)) & 3u)));
}
MK_HD uint64_t synth_rec_hash(uint64_t seed, uint64_t rec) { return splitmix64((seed + 0x5851F42D4C957F2Dull) ^ (rec * 0x2545F4914F6CDD1Dull)); }

}  // namespace mk
