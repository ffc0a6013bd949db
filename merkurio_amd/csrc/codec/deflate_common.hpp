// deflate_common.hpp -- the serial pieces of RFC 1951 (DEFLATE) shared by the BGZF kernels of the
// device codec (bgzf_deflate.hip, bgzf_inflate.hip) and by their host test harness
// (tests/helpers/codec_harness.cpp compiles this header with g++ and checks it against zlib).
//
// What the reference gets from its crates: BGZF blocks inflated / deflated by `bam 0.1.4` (flate2) on
// `tag`'s reader and writer threads (src/cmd_tag.rs:254-271,503-615) and gzip'ed FASTA/FASTQ by needletail
// (src/cmd_extract.rs:281).  DEFLATE is specified by RFC 1951, BGZF by the SAM specification §4.1; the
// functions below restate those documents, every wave-level step lives in the .hip files.
//
// Everything here is branchy scalar code over small arrays (<= 288 symbols): on the device ONE lane of a
// wave runs it, between the parallel phases of a block.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define MKZ_HD __host__ __device__ inline
#else
#define MKZ_HD inline
#endif

namespace mkz {

constexpr int kLitLen = 286;   // literal / length alphabet (RFC 1951 §3.2.5)
constexpr int kDist = 30;      // distance alphabet
constexpr int kCl = 19;        // code-length alphabet (§3.2.7)
constexpr int kMaxBits = 15;   // longest literal / length / distance codeword
constexpr int kMaxClBits = 7;  // longest code-length codeword
constexpr int kMinMatch = 3, kMaxMatch = 258, kWindow = 32768;

MKZ_HD int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }

// length 3..258 -> index of its length symbol (symbol = 257 + idx), number of extra bits, their value
MKZ_HD void length_symbol(uint32_t len, uint32_t &idx, uint32_t &nbits, uint32_t &bits) {
    const uint32_t l = len - 3;
    if (len == 258) {
        idx = 28, nbits = 0, bits = 0;
    } else if (l < 8) {
        idx = l, nbits = 0, bits = 0;
    } else {
        nbits = (uint32_t)floor_log2(l) - 2;
        idx = 4 * nbits + 4 + ((l >> nbits) & 3);
        bits = l & ((1u << nbits) - 1);
    }
}
// distance 1..32768 -> distance symbol, number of extra bits, their value
MKZ_HD void distance_symbol(uint32_t dist, uint32_t &sym, uint32_t &nbits, uint32_t &bits) {
    const uint32_t d = dist - 1;
    if (d < 4) {
        sym = d, nbits = 0, bits = 0;
    } else {
        nbits = (uint32_t)floor_log2(d) - 1;
        sym = 2 * nbits + 2 + ((d >> nbits) & 1);
        bits = d & ((1u << nbits) - 1);
    }
}
// the inverse maps (decoder): base value and extra-bit count of a length symbol index / a distance symbol
MKZ_HD uint32_t length_extra_bits(uint32_t idx) { return idx < 8 || idx == 28 ? 0 : (idx - 4) >> 2; }
MKZ_HD uint32_t length_base(uint32_t idx) {
    if (idx < 8) return 3 + idx;
    if (idx == 28) return 258;
    const uint32_t e = (idx - 4) >> 2;
    return 3 + ((4 + (idx & 3)) << e);
}
MKZ_HD uint32_t distance_extra_bits(uint32_t sym) { return sym < 4 ? 0 : (sym - 2) >> 1; }
MKZ_HD uint32_t distance_base(uint32_t sym) {
    if (sym < 4) return 1 + sym;
    const uint32_t e = (sym - 2) >> 1;
    return 1 + ((2 + (sym & 1)) << e);
}

MKZ_HD uint32_t reverse_bits(uint32_t code, uint32_t len) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// ---- code lengths of a length-limited prefix code ------------------------------------------------
// skey[0..m): the used symbols, (frequency << 9 | symbol) in ascending order, m >= 2.  Writes lens[symbol] for
// those symbols.  Unrestricted Huffman lengths by the two-queue merge over the sorted leaves; where the tree is
// deeper than `maxbits` the count of codewords per length is repaired until the Kraft sum is exactly one (every
// step takes one codeword off the longest length and splits one shorter codeword into two children: the sum
// falls by one unit of 2^-maxbits), and the lengths are handed out again by frequency rank.
struct HuffScratch {
    uint32_t weight[288];    // internal nodes, in creation (= ascending weight) order
    uint16_t parent_leaf[288];
    uint16_t parent_node[288];
    uint8_t depth[288];
    uint32_t count[kMaxBits + 2];
};

MKZ_HD void code_lengths_from_sorted(const uint32_t *skey, int m, int maxbits, uint8_t *lens, HuffScratch &s) {
    int leaf = 0, head = 0;  // next unmerged leaf / internal node
    for (int k = 0; k < m - 1; ++k) {
        uint32_t w = 0;
        for (int side = 0; side < 2; ++side) {
            const bool take_leaf = leaf < m && (head >= k || (skey[leaf] >> 9) <= s.weight[head]);
            if (take_leaf) {
                w += skey[leaf] >> 9;
                s.parent_leaf[leaf++] = (uint16_t)k;
            } else {
                w += s.weight[head];
                s.parent_node[head++] = (uint16_t)k;
            }
        }
        s.weight[k] = w;
    }
    s.depth[m - 2] = 0;
    for (int k = m - 3; k >= 0; --k) s.depth[k] = (uint8_t)(s.depth[s.parent_node[k]] + 1);
    for (int l = 0; l <= maxbits; ++l) s.count[l] = 0;
    bool over = false;
    for (int i = 0; i < m; ++i) {
        int d = s.depth[s.parent_leaf[i]] + 1;
        if (d > maxbits) d = maxbits, over = true;
        s.count[d]++;
    }
    if (over) {
        uint32_t kraft = 0;
        for (int l = 1; l <= maxbits; ++l) kraft += s.count[l] << (maxbits - l);
        while (kraft > (1u << maxbits)) {
            s.count[maxbits]--;
            for (int l = maxbits - 1; l > 0; --l)
                if (s.count[l]) {
                    s.count[l]--;
                    s.count[l + 1] += 2;
                    break;
                }
            kraft--;
        }
    }
    int i = 0;  // the rarest symbols take the longest codewords
    for (int l = maxbits; l >= 1; --l)
        for (uint32_t c = 0; c < s.count[l]; ++c) lens[skey[i++] & 511u] = (uint8_t)l;
}

// canonical codewords (§3.2.2) of lens[0..n), stored bit-reversed: DEFLATE packs Huffman codes starting with
// their most significant bit into a stream that is otherwise filled from the least significant bit
// (count[] and next[] -- indexed by a code length -- live in the caller's HuffScratch, LDS on the device: a local array indexed by
// a run-time value would be private memory)
MKZ_HD void canonical_codes(const uint8_t *lens, int n, uint16_t *codes, HuffScratch &s) {
    uint32_t *const count = s.count, *const next = s.weight;  // (weight[]: free once the lengths are known)
    for (int l = 0; l <= kMaxBits; ++l) count[l] = 0;
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= kMaxBits; ++l) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
    }
    for (int i = 0; i < n; ++i) codes[i] = lens[i] ? (uint16_t)reverse_bits(next[lens[i]]++, lens[i]) : 0;
}

// ---- the two codes of a block from its symbol frequencies -------------------------------------------
// keys of the used symbols of freq[0..n) (frequency << 9 | symbol), UNSORTED; a code needs two codewords to be
// complete, so an alphabet with fewer than two used symbols is topped up with symbols of frequency 1 that are
// never emitted (what zlib's build_tree does: every inflater accepts the result).  Returns the number of keys.
MKZ_HD int symbol_keys(const uint32_t *freq, int n, uint32_t *key) {
    int m = 0;
    for (int s = 0; s < n; ++s)
        if (freq[s]) key[m++] = freq[s] << 9 | (uint32_t)s;
    for (uint32_t s = 0; m < 2; ++s) {
        bool used = false;
        for (int k = 0; k < m; ++k) used |= (key[k] & 511u) == s;
        if (!used) key[m++] = 1u << 9 | s;
    }
    return m;
}

struct BlockCodes {
    uint8_t ll_len[288];
    uint8_t d_len[32];
    uint16_t ll_code[288];
    uint16_t d_code[32];
};
// sorted keys of both alphabets -> lengths and codewords
MKZ_HD void block_codes_from_sorted(const uint32_t *ll_key, int ll_m, const uint32_t *d_key, int d_m, BlockCodes &c, HuffScratch &s) {
    for (int i = 0; i < 288; ++i) c.ll_len[i] = 0;
    for (int i = 0; i < 32; ++i) c.d_len[i] = 0;
    code_lengths_from_sorted(ll_key, ll_m, kMaxBits, c.ll_len, s);
    code_lengths_from_sorted(d_key, d_m, kMaxBits, c.d_len, s);
    canonical_codes(c.ll_len, kLitLen, c.ll_code, s);
    canonical_codes(c.d_len, kDist, c.d_code, s);
}

// ---- bit stream, filled from the least significant bit of 32-bit words (words zeroed by the caller) ----
struct BitSink {
    uint32_t *words;
    uint32_t bitpos;
};
MKZ_HD void put_bits(BitSink &b, uint32_t v, uint32_t n) {  // n <= 16
    const uint32_t w = b.bitpos >> 5, sh = b.bitpos & 31;
    b.words[w] |= v << sh;
    if (sh + n > 32) b.words[w + 1] |= v >> (32 - sh);
    b.bitpos += n;
}

// ---- header of a dynamic block (§3.2.7) ------------------------------------------------------------
// run-length form of the hlit + hdist code lengths: entries (code-length symbol | extra value << 8)
struct HeaderScratch {
    uint16_t rle[kLitLen + kDist];
    uint32_t n_rle;
    uint32_t cl_freq[kCl];
    uint8_t cl_len[kCl];
    uint16_t cl_code[kCl];
    uint32_t hlit, hdist, hclen;
    uint32_t cl_key[kCl + 2];  // the code-length code's (frequency << 9 | symbol) keys, sorted
    HuffScratch huff;
};

// the order in which the lengths of the code-length code are sent (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 -- 5 bits
// each in two constants, so that position i needs a shift, not an array
MKZ_HD uint32_t cl_order_at(uint32_t i) {
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 |
                        11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return (uint32_t)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31u);
}

// Builds the run-length form and the code-length code; returns the header's size in bits (BFINAL/BTYPE included).
MKZ_HD uint32_t plan_dynamic_header(const uint8_t *ll_len, const uint8_t *d_len, HeaderScratch &h) {
    uint32_t hlit = kLitLen, hdist = kDist;
    while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
    while (hdist > 1 && d_len[hdist - 1] == 0) --hdist;
    h.hlit = hlit, h.hdist = hdist;
    for (int i = 0; i < kCl; ++i) h.cl_freq[i] = 0, h.cl_len[i] = 0;
    const uint32_t total = hlit + hdist;
    uint32_t n = 0, i = 0;
    while (i < total) {
        const uint32_t v = i < hlit ? ll_len[i] : d_len[i - hlit];
        uint32_t run = 1;
        while (i + run < total && (i + run < hlit ? ll_len[i + run] : d_len[i + run - hlit]) == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11) {
                const uint32_t k = run < 138 ? run : 138;
                h.rle[n++] = (uint16_t)(18 | (k - 11) << 8), h.cl_freq[18]++, run -= k;
            }
            if (run >= 3) h.rle[n++] = (uint16_t)(17 | (run - 3) << 8), h.cl_freq[17]++, run = 0;
        } else {
            h.rle[n++] = (uint16_t)v, h.cl_freq[v]++, --run;
            while (run >= 3) {
                const uint32_t k = run < 6 ? run : 6;
                h.rle[n++] = (uint16_t)(16 | (k - 3) << 8), h.cl_freq[16]++, run -= k;
            }
        }
        for (; run; --run) h.rle[n++] = (uint16_t)v, h.cl_freq[v]++;
    }
    h.n_rle = n;
    // code-length code: <= 19 symbols, insertion sort of their keys
    uint32_t *const key = h.cl_key;
    int m = 0;
    for (int s = 0; s < kCl; ++s)
        if (h.cl_freq[s]) {
            const uint32_t k = h.cl_freq[s] << 9 | (uint32_t)s;
            int j = m++;
            for (; j > 0 && key[j - 1] > k; --j) key[j] = key[j - 1];
            key[j] = k;
        }
    if (m == 1) {  // a code of one symbol still takes one bit; give it a partner so that the code is complete
        const uint32_t other = (key[0] & 511u) == 0 ? 1u : 0u;
        const uint32_t k = 1u << 9 | other;  // frequency 1 (never emitted)
        if (k < key[0]) key[1] = key[0], key[0] = k; else key[1] = k;
        m = 2;
    }
    code_lengths_from_sorted(key, m, kMaxClBits, h.cl_len, h.huff);
    canonical_codes(h.cl_len, kCl, h.cl_code, h.huff);
    uint32_t hclen = kCl;
    while (hclen > 4 && h.cl_len[cl_order_at(hclen - 1)] == 0) --hclen;
    h.hclen = hclen;
    uint32_t bits = 3 + 5 + 5 + 4 + 3 * hclen;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t s = h.rle[k] & 255u;
        bits += h.cl_len[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0);
    }
    return bits;
}

MKZ_HD void write_dynamic_header(BitSink &b, const HeaderScratch &h, bool final_block) {
    put_bits(b, final_block ? 1 : 0, 1);
    put_bits(b, 2, 2);  // BTYPE = 10: dynamic Huffman codes
    put_bits(b, h.hlit - 257, 5);
    put_bits(b, h.hdist - 1, 5);
    put_bits(b, h.hclen - 4, 4);
    for (uint32_t i = 0; i < h.hclen; ++i) put_bits(b, h.cl_len[cl_order_at(i)], 3);
    for (uint32_t k = 0; k < h.n_rle; ++k) {
        const uint32_t s = h.rle[k] & 255u, x = h.rle[k] >> 8;
        put_bits(b, h.cl_code[s], h.cl_len[s]);
        if (s == 16) put_bits(b, x, 2);
        if (s == 17) put_bits(b, x, 3);
        if (s == 18) put_bits(b, x, 7);
    }
}

// the fixed code of §3.2.6
MKZ_HD void fixed_code_lengths(uint8_t *ll_len /*288*/, uint8_t *d_len /*32*/) {
    for (int i = 0; i < 144; ++i) ll_len[i] = 8;
    for (int i = 144; i < 256; ++i) ll_len[i] = 9;
    for (int i = 256; i < 280; ++i) ll_len[i] = 7;
    for (int i = 280; i < 288; ++i) ll_len[i] = 8;
    for (int i = 0; i < 32; ++i) d_len[i] = 5;
}

// ---- CRC-32 (gzip: reflected polynomial 0xedb88320) in pieces ------------------------------------------
// The register after a piece that starts from register r:  state(B, r) = state(B, 0) ^ shift(r, |B|), shift = the
// register clocked through |B| zero bytes = r * x^(8|B|) mod P.  The lanes of a wave take one piece each (from
// register 0, the first one from 0xffffffff) and fold.
constexpr uint32_t kCrcPoly = 0xedb88320u;
MKZ_HD uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
    return c;
}
// a * b mod P, polynomials in the reflected representation (x^0 = bit 31)
MKZ_HD uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
// x^(8 n) mod P
MKZ_HD uint32_t crc_x_pow_bytes(uint64_t n) {
    uint32_t r = 0x80000000u;       // x^0
    uint32_t sq = 0x00800000u;      // x^8
    for (; n; n >>= 1) {
        if (n & 1) r = crc_mulmod(r, sq);
        sq = crc_mulmod(sq, sq);
    }
    return r;
}

}  // namespace mkz
