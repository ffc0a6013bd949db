// inflate_serial.hpp -- one DEFLATE stream (RFC 1951) decoded by ONE lane: the body of mk_bgzf_inflate_kernel
// (bgzf_inflate.hip: 64 BGZF members per wave, one per lane) and, compiled with g++, of the host test harness
// (tests/helpers/codec_harness.cpp), which checks it against streams zlib wrote.
//
// Replaces, for BGZF input, the inflate the reference gets from flate2 inside `bam 0.1.4` (src/cmd_tag.rs:506)
// and needletail's gzip reader (src/cmd_extract.rs:281).  A BGZF member is at most 64 KiB of text, members are
// independent: the parallelism is across members, the decode of one member is the serial loop below.
#pragma once
#include "deflate_common.hpp"

namespace mkz {

constexpr int kLlFastBits = 9;  // literal / length codewords of up to 9 bits decode with one table read
constexpr int kDFastBits = 6;   // distance codewords of up to 6 bits
constexpr int kLlFastSize = 1 << kLlFastBits, kDFastSize = 1 << kDFastBits;

// error codes of inflate_stream (negative; 0 = the stream ended cleanly with exactly n_out bytes)
constexpr int kInfTruncated = -1, kInfBadBlockType = -2, kInfBadStored = -3, kInfBadLengths = -4, kInfBadSymbol = -5,
              kInfBadDistance = -6, kInfOutputOverrun = -7, kInfOutputShort = -8;

// the slow half of a decoder's tables (per lane: private memory on the device)
struct InflateScratch {
    DecodeCounts ll_count, d_count;
    uint16_t ll_sorted[288], d_sorted[32];
    uint8_t lens[288 + 32];
};

MKZ_HD uint64_t load_le64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// every entry of a fast table in symbol order (one lane owns the table)
template <class Table>
MKZ_HD void fill_decode_table_serial(const uint8_t *lens, int n, int pbits, Table table) {
    uint32_t count[kMaxBits + 1], next[kMaxBits + 2];
    for (int l = 0; l <= kMaxBits; ++l) count[l] = 0;
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= kMaxBits; ++l) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
    }
    for (int k = 0; k < (1 << pbits); ++k) table[k] = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t l = lens[i];
        if (l == 0) continue;
        const uint32_t c = next[l]++;
        if ((int)l > pbits) continue;
        const uint32_t r = reverse_bits(c, l);
        for (uint32_t k = r; k < (1u << pbits); k += 1u << l) table[k] = (uint16_t)((uint32_t)i << 4 | l);
    }
}

// in[0, n_in): the raw DEFLATE stream (readable up to in + n_in + 16); out[0, n_out): exactly what it must inflate to
// (ISIZE of the BGZF member).  ll_fast / d_fast: this decoder's fast tables (kLlFastSize / kDFastSize entries; LDS on
// the device).
template <class Table>
MKZ_HD int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, Table ll_fast, Table d_fast,
                          InflateScratch &s) {
    const uint8_t *p = in, *const in_end = in + n_in;
    uint8_t *op = out, *const out_end = out + n_out;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
#define MKZ_REFILL()                                     \
    do {                                                 \
        bitbuf |= load_le64(p) << bitcnt;                \
        p += (63 - bitcnt) >> 3;                         \
        bitcnt |= 56;                                    \
    } while (0)
#define MKZ_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
    for (;;) {
        MKZ_REFILL();
        // bytes the buffer has taken past the end of the stream are zeros of the padding: a stream that needs them is truncated
        if (p > in_end + 8) return kInfTruncated;
        const uint32_t final_block = (uint32_t)bitbuf & 1u, type = ((uint32_t)bitbuf >> 1) & 3u;
        MKZ_TAKE(3);
        if (type == 0) {  // stored: skip to the byte boundary, LEN, ~LEN, bytes
            MKZ_TAKE(bitcnt & 7);
            const uint32_t len = (uint32_t)bitbuf & 0xffffu, nlen = ((uint32_t)(bitbuf >> 16)) & 0xffffu;
            MKZ_TAKE(32);
            if ((len ^ nlen) != 0xffffu) return kInfBadStored;
            p -= bitcnt >> 3;  // whole bytes still in the buffer go back
            bitbuf = 0, bitcnt = 0;
            if (p + len > in_end) return kInfTruncated;
            if (len > (uint32_t)(out_end - op)) return kInfOutputOverrun;
            for (uint32_t i = 0; i < len; ++i) op[i] = p[i];
            op += len, p += len;
            if (final_block) break;
            continue;
        }
        if (type == 3) return kInfBadBlockType;
        if (type == 1) {
            fixed_code_lengths(s.lens, s.lens + 288);
            canonical_decode_order(s.lens, 288, s.ll_count, s.ll_sorted);
            canonical_decode_order(s.lens + 288, 30, s.d_count, s.d_sorted);
            fill_decode_table_serial(s.lens, 288, kLlFastBits, ll_fast);
            fill_decode_table_serial(s.lens + 288, 30, kDFastBits, d_fast);
        } else {
            const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = ((uint32_t)(bitbuf >> 5) & 31u) + 1,
                           hclen = ((uint32_t)(bitbuf >> 10) & 15u) + 4;
            MKZ_TAKE(14);
            if (hlit > 286 || hdist > 30) return kInfBadLengths;
            uint8_t order[kCl], cl_len[kCl];
            cl_order(order);
            for (int i = 0; i < kCl; ++i) cl_len[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                if (bitcnt < 3) MKZ_REFILL();
                cl_len[order[i]] = (uint8_t)((uint32_t)bitbuf & 7u);
                MKZ_TAKE(3);
            }
            // the code-length code borrows the literal table (128 of its 512 entries) and the distance scratch
            if (canonical_decode_order(cl_len, kCl, s.d_count, s.d_sorted, false)) return kInfBadLengths;
            fill_decode_table_serial(cl_len, kCl, kMaxClBits, ll_fast);
            uint32_t i = 0;
            while (i < hlit + hdist) {
                MKZ_REFILL();
                if (p > in_end + 8) return kInfTruncated;
                const uint32_t e = ll_fast[(uint32_t)bitbuf & ((1u << kMaxClBits) - 1)];
                if (e == 0) return kInfBadLengths;
                MKZ_TAKE(e & 15u);
                const uint32_t sym = e >> 4;
                if (sym < 16) {
                    s.lens[i++] = (uint8_t)sym;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return kInfBadLengths;
                    val = s.lens[i - 1];
                    rep = 3 + ((uint32_t)bitbuf & 3u);
                    MKZ_TAKE(2);
                } else if (sym == 17) {
                    rep = 3 + ((uint32_t)bitbuf & 7u);
                    MKZ_TAKE(3);
                } else {
                    rep = 11 + ((uint32_t)bitbuf & 127u);
                    MKZ_TAKE(7);
                }
                if (i + rep > hlit + hdist) return kInfBadLengths;
                for (; rep; --rep) s.lens[i++] = (uint8_t)val;
            }
            if (p > in_end + 8) return kInfTruncated;
            if (s.lens[256] == 0) return kInfBadLengths;  // no end-of-block codeword
            // distance lengths behind the literal ones, each alphabet padded with zeros to its full size
            uint8_t *const dl = s.lens + 288;
            for (int k = (int)hdist - 1; k >= 0; --k) dl[k] = s.lens[hlit + (uint32_t)k];
            for (uint32_t k = hdist; k < 32; ++k) dl[k] = 0;
            for (uint32_t k = hlit; k < 288; ++k) s.lens[k] = 0;
            if (canonical_decode_order(s.lens, 288, s.ll_count, s.ll_sorted)) return kInfBadLengths;
            if (canonical_decode_order(dl, 30, s.d_count, s.d_sorted)) return kInfBadLengths;
            fill_decode_table_serial(s.lens, 288, kLlFastBits, ll_fast);
            fill_decode_table_serial(dl, 30, kDFastBits, d_fast);
        }
        // symbols of this block
        for (;;) {
            MKZ_REFILL();
            if (p > in_end + 8) return kInfTruncated;  // (also what keeps a corrupt stream's reads inside the buffer's padding)
            uint32_t e = ll_fast[(uint32_t)bitbuf & (kLlFastSize - 1)];
            if (e == 0) {
                e = decode_slow((uint32_t)bitbuf, s.ll_count, s.ll_sorted);
                if (e == 0) return kInfBadSymbol;
            }
            MKZ_TAKE(e & 15u);
            const uint32_t sym = e >> 4;
            if (sym < 256) {
                if (op == out_end) return kInfOutputOverrun;
                *op++ = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            const uint32_t idx = sym - 257;
            if (idx > 28) return kInfBadSymbol;
            const uint32_t leb = length_extra_bits(idx);
            const uint32_t len = length_base(idx) + ((uint32_t)bitbuf & ((1u << leb) - 1));
            MKZ_TAKE(leb);
            uint32_t d = d_fast[(uint32_t)bitbuf & (kDFastSize - 1)];
            if (d == 0) {
                d = decode_slow((uint32_t)bitbuf, s.d_count, s.d_sorted);
                if (d == 0) return kInfBadSymbol;
            }
            MKZ_TAKE(d & 15u);
            const uint32_t dsym = d >> 4;
            if (dsym > 29) return kInfBadSymbol;
            const uint32_t deb = distance_extra_bits(dsym);
            const uint32_t dist = distance_base(dsym) + ((uint32_t)bitbuf & ((1u << deb) - 1));
            MKZ_TAKE(deb);
            if (dist > (uint32_t)(op - out)) return kInfBadDistance;
            if (len > (uint32_t)(out_end - op)) return kInfOutputOverrun;
            const uint8_t *src = op - dist;
            for (uint32_t i = 0; i < len; ++i) op[i] = src[i];
            op += len;
        }
        if (p > in_end + 8) return kInfTruncated;
        if (final_block) break;
    }
#undef MKZ_REFILL
#undef MKZ_TAKE
    // bits consumed must lie inside the stream
    if ((int64_t)(p - in) - (int64_t)(bitcnt >> 3) > (int64_t)n_in) return kInfTruncated;
    return op == out_end ? 0 : kInfOutputShort;
}

}  // namespace mkz
