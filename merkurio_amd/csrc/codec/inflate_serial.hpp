// inflate_serial.hpp -- one DEFLATE stream (RFC 1951) decoded by ONE lane: the body of mk_bgzf_inflate_kernel
// (bgzf_inflate.hip: 64 BGZF members per wave, one per lane) and, compiled with g++, of the host test harness
// (tests/helpers/codec_harness.cpp), which checks it against streams zlib wrote.
//
// Replaces, for BGZF input, the inflate the reference gets from flate2 inside `bam 0.1.4` (src/cmd_tag.rs:506)
// and needletail's gzip reader (src/cmd_extract.rs:281).  A BGZF member is at most 64 KiB of text, members are
// independent: the parallelism is across members, the decode of one member is the serial loop below.
//
// What shapes the loop is latency, not arithmetic: 64 lanes run it in lockstep, so whatever ONE lane waits for, the
// wave waits for -- a fast path that most lanes take most of the time is worth nothing when some lane is always on
// the slow one.  Hence (r04, second version; the first one had a fast table in LDS, the rare-codeword tables in
// private memory and read the stream with one global load per token: 5-7 us per token):
//   * ONE way to decode a codeword, of fixed cost: the canonical description of a code as left-aligned limits --
//     the codeword's length is the number of limits the next 15 stream bits (bit-reversed) reach, found without a
//     branch from 8 dword reads of the lane's LDS block, then base[length] and the symbol: three LDS round trips;
//   * the compressed stream passes through a 64-byte LDS ring per lane, topped up by ALL lanes together on every
//     8th token turn: one global round trip per 8 turns ("reload when empty" waits on almost every turn: some lane
//     of the 64 is always empty);
//   * matches are copied in rounds of up to 32 bytes (a register pattern for distances below 8), at most one round
//     per token turn, and a round's loads are issued one turn before its stores: the L2 round trip runs beside the
//     next decode.  A long match occupies its own lane for several turns instead of stalling all 64.
// 836 bytes of LDS per lane (an odd number of dwords: the lanes' blocks start in different banks), 52 KiB per wave.
// Nothing sits in private memory (r05: a block header is decoded twice instead of keeping its code lengths in an array).
#pragma once
#include "deflate_common.hpp"

namespace mkz {

constexpr int kWinWords = 16;         // the stream window: 64 bytes
constexpr uint32_t kStreamPad = 128;  // readable bytes every stream needs behind its last byte

// error codes of inflate_stream (negative; 0 = the stream ended cleanly with exactly n_out bytes)
constexpr int kInfTruncated = -1, kInfBadBlockType = -2, kInfBadStored = -3, kInfBadLengths = -4, kInfBadSymbol = -5,
              kInfBadDistance = -6, kInfOutputOverrun = -7, kInfOutputShort = -8;

// a lane's decoder memory, in 16-bit units (LDS on the device):
//   ll_limit[16] ll_base[16] d_limit[16] d_base[16] win[32 = 16 dwords] ll_sorted[288] d_sorted[32], one unit of padding
constexpr int kOffLlLimit = 0, kOffLlBase = 16, kOffDLimit = 32, kOffDBase = 48, kOffWin = 64, kOffLlSorted = kOffWin + 2 * kWinWords,
              kOffDSorted = kOffLlSorted + 288, kLaneTableU16 = kOffDSorted + 32 + 2;
static_assert(kOffWin % 2 == 0 && kLaneTableU16 % 2 == 0 && (kLaneTableU16 / 2) % 2 == 1, "dword-aligned blocks of an odd number of dwords");

MKZ_HD uint64_t load_le64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
MKZ_HD void store_le64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
MKZ_HD uint32_t load_le32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// The canonical description of a prefix code, as the decoders use it: sorted[] = the symbols in (length, symbol) order; limit[l] =
// the first left-aligned 15-bit value behind the codewords of length l (ascending in l); base[l] = (rank of the first symbol of
// length l) - (its first codeword), mod 2^16.  Built by tables_from_counts below (one lane, no arrays) and by wave_build_tables
// (bgzf_inflate_wave.hip: a whole wave).  A code is refused when it is over-subscribed, or incomplete and not one of the two shapes
// zlib's inflate_table accepts: no codeword at all, or -- literal / length and distance codes -- exactly one codeword of length 1.

MKZ_HD uint32_t bit_reverse32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v);
#else
    v = (v >> 16) | (v << 16);
    v = ((v & 0xff00ff00u) >> 8) | ((v & 0x00ff00ffu) << 8);
    v = ((v & 0xf0f0f0f0u) >> 4) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v & 0xccccccccu) >> 2) | ((v & 0x33333333u) << 2);
    return ((v & 0xaaaaaaaau) >> 1) | ((v & 0x55555555u) << 1);
#endif
}
// the codeword at the low end of `bits`: symbol << 4 | length, or 0 if no codeword matches
MKZ_HD uint32_t decode_codeword(uint32_t bits, const uint16_t *sorted, const uint16_t *limit, const uint16_t *base) {
    const uint32_t w = bit_reverse32(bits) >> (32 - kMaxBits);  // the next 15 stream bits, first bit on top
    // (the limits are written as 16-bit values and read here two at a time: a type that may alias them, or the compiler is free to
    // move these loads across the stores that built the table -- which it did once both sat in one function, r05's block search)
    typedef uint32_t __attribute__((may_alias)) pair_u16;
    const pair_u16 *lw = reinterpret_cast<const pair_u16 *>(limit);
    uint32_t l = 1;
    for (int k = 0; k < 8; ++k) {  // limits ascend with the length: count those w has reached (limit[0] = 0 stands for the start of l, limit[15] is left out)
        const uint32_t pair = lw[k];
        l += (k ? w >= (pair & 0xffffu) : 0u) + (k < 7 ? w >= (pair >> 16) : 0u);
    }
    if (w >= limit[l]) return 0;
    const uint32_t idx = (base[l] + (w >> (kMaxBits - l))) & 0xffffu;
    return (uint32_t)sorted[idx] << 4 | l;
}

// ---- a block header without a code-length array (r05) -----------------------------------------------------------------------
// The 288 + 32 code lengths of a dynamic block used to sit in a per-lane array: 336 bytes of private (scratch) memory, the only
// ones of the kernel.  They are not needed as an array: the header is decoded TWICE.  Pass 1 only counts the codewords of each
// length (fifteen 10-bit counters in three 64-bit registers per alphabet); from the counts follow the limits, the bases and the rank
// of the first symbol of every length; the bit reader goes back to where the lengths began and pass 2 drops each symbol into its
// place in sorted[].  The code-length code itself (19 symbols of at most 7 bits) lives in four 64-bit registers.  No array is
// indexed by a run-time value anywhere in the header any more: 0 bytes of scratch.
// (three separate scalars, not an array and not a struct whose members a select could pick between by address: what a run-time
// value selects here is a VALUE, so nothing has to live in memory)
// ("Pack15": fifteen 10-bit fields, l = 1..15: 1..6 in a, 7..12 in b, 13..15 in c)
MKZ_HD uint32_t p15_get(uint64_t a, uint64_t b, uint64_t c, uint32_t l) {
    const uint32_t k = l - 1;
    const uint32_t f = k < 6 ? k : k < 12 ? k - 6 : k - 12;
    const uint64_t w = (k < 6 ? a : 0ull) | ((k >= 6 && k < 12) ? b : 0ull) | (k >= 12 ? c : 0ull);
    return (uint32_t)(w >> (10 * f)) & 1023u;
}
MKZ_HD void p15_add(uint64_t &a, uint64_t &b, uint64_t &c, uint32_t l, uint32_t v) {
    const uint32_t k = l - 1;
    const uint32_t f = k < 6 ? k : k < 12 ? k - 6 : k - 12;
    const uint64_t inc = (uint64_t)v << (10 * f);
    a += k < 6 ? inc : 0ull;
    b += (k >= 6 && k < 12) ? inc : 0ull;
    c += k >= 12 ? inc : 0ull;
}
MKZ_HD void p15_set(uint64_t &a, uint64_t &b, uint64_t &c, uint32_t l, uint32_t v) {
    const uint32_t k = l - 1;
    const uint32_t f = k < 6 ? k : k < 12 ? k - 6 : k - 12;
    const uint64_t m = ~(1023ull << (10 * f)), x = (uint64_t)v << (10 * f);
    a = k < 6 ? (a & m) | x : a;
    b = (k >= 6 && k < 12) ? (b & m) | x : b;
    c = k >= 12 ? (c & m) | x : c;
}
// limit[] / base[] (as defined above) and offs = rank of the first symbol of every length, from the counts per
// length; `used` = codewords in all.  Returns 0 / 1 over-subscribed / 2 incomplete and not a shape zlib's inflate_table accepts.
// (in place: `p` holds the counts on entry and the ranks on return -- the two are never needed together)
MKZ_HD int tables_from_counts(uint64_t &pa, uint64_t &pb, uint64_t &pc, uint32_t used, uint16_t *limit, uint16_t *base, bool allow_single) {
    int left = 1;
    uint32_t code = 0, off = 0, prev = 0;
    limit[0] = 0, base[0] = 0;
    const uint32_t c1 = p15_get(pa, pb, pc, 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
        const uint32_t c = p15_get(pa, pb, pc, l);
        left = (left << 1) - (int)c;
        if (left < 0) return 1;
        code = (code + prev) << 1;
        limit[l] = (uint16_t)((code + c) << (kMaxBits - l));
        base[l] = (uint16_t)(off - code);
        p15_set(pa, pb, pc, l, off);  // count -> rank of the first symbol of this length
        off += c;
        prev = c;
    }
    if (left > 0 && !(used == 0 || (allow_single && used == 1 && c1 == 1))) return 2;
    return 0;
}

// the code-length code in registers: limit / base of the lengths 1..7 (8 bits each, field l), the symbols in (length, symbol)
// order (5 bits each: ranks 0..11 in sorted_lo, 12..18 in sorted_hi)
struct ClCode {
    uint64_t limit, base, sorted_lo, sorted_hi;
};
// cl = the 19 lengths, 3 bits each, indexed by symbol.  Returns 0 or 1 (over-subscribed, or incomplete with any codeword at all)
MKZ_HD int cl_build(uint64_t cl, ClCode &t) {
    uint64_t cnt = 0;  // eight 5-bit counters
    uint32_t used = 0;
    for (int s = 0; s < kCl; ++s) {
        const uint32_t len = (uint32_t)(cl >> (3 * s)) & 7u;
        cnt += len ? 1ull << (5 * len) : 0ull;
        used += len != 0;
    }
    int left = 1;
    uint32_t code = 0, off = 0, prev = 0;
    uint64_t offs = 0;
    t.limit = t.base = t.sorted_lo = t.sorted_hi = 0;
    for (uint32_t l = 1; l <= (uint32_t)kMaxClBits; ++l) {
        const uint32_t c = (uint32_t)(cnt >> (5 * l)) & 31u;
        left = (left << 1) - (int)c;
        if (left < 0) return 1;
        code = (code + prev) << 1;
        t.limit |= (uint64_t)(((code + c) << (kMaxClBits - l)) & 0xffu) << (8 * l);
        t.base |= (uint64_t)((off - code) & 0xffu) << (8 * l);
        offs |= (uint64_t)off << (5 * l);
        off += c;
        prev = c;
    }
    if (left > 0 && used != 0) return 1;
    for (int s = 0; s < kCl; ++s) {
        const uint32_t len = (uint32_t)(cl >> (3 * s)) & 7u;
        if (!len) continue;
        const uint32_t pos = (uint32_t)(offs >> (5 * len)) & 31u;
        offs += 1ull << (5 * len);
        if (pos < 12) t.sorted_lo |= (uint64_t)s << (5 * pos);
        else t.sorted_hi |= (uint64_t)s << (5 * (pos - 12));
    }
    return 0;
}
// the code-length codeword at the low end of `bits`: symbol << 4 | length, or 0 (decode_codeword's rule on 7 bits)
MKZ_HD uint32_t cl_decode(uint32_t bits, const ClCode &t) {
    const uint32_t w = bit_reverse32(bits) >> (32 - kMaxClBits);
    uint32_t l = 1;
    for (int k = 1; k < kMaxClBits; ++k) l += w >= ((uint32_t)(t.limit >> (8 * k)) & 0xffu);  // (a limit of 128 is never reached: w < 128)
    // limit[l] == 128 means "every 7-bit value is in front of it"; as an 8-bit field 128 survives, as the general code's 2^15 does in 16 bits
    if (w >= ((uint32_t)(t.limit >> (8 * l)) & 0xffu)) return 0;
    const uint32_t idx = ((uint32_t)(t.base >> (8 * l)) + (w >> (kMaxClBits - l))) & 0xffu;
    const uint32_t sym = idx < 12 ? (uint32_t)(t.sorted_lo >> (5 * idx)) & 31u : (uint32_t)(t.sorted_hi >> (5 * (idx - 12))) & 31u;
    return sym << 4 | l;
}

// in[0, n_in): the raw DEFLATE stream (readable up to in + n_in + kStreamPad); out[0, n_out): exactly what it must
// inflate to (ISIZE of the BGZF member), readable up to out + n_out + 8 (match copies load whole 8-byte words; nothing is
// written outside out[0, n_out)).  t: this decoder's kLaneTableU16 16-bit words (4-byte aligned).
MKZ_HD int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint16_t *t) {
    uint16_t *const ll_sorted = t + kOffLlSorted, *const d_sorted = t + kOffDSorted, *const ll_limit = t + kOffLlLimit, *const ll_base = t + kOffLlBase,
                    *const d_limit = t + kOffDLimit, *const d_base = t + kOffDBase;
    uint32_t *const win = reinterpret_cast<uint32_t *>(t + kOffWin);
    uint8_t *op = out, *const out_end = out + n_out;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    // the window: a ring of kWinWords dwords of the stream.  wpos = stream offset of the next dword to load, wr / rd =
    // dwords loaded / handed to the bit buffer.  Dwords behind n_in + 8 are not loaded (zeros stand for them): a stream
    // that takes one of those has run out (MKZ_RAN_OUT), a valid one never does.
    uint32_t wpos = 0, wr = 0, rd = 0;
    uint32_t turn = 0;  // token turns of this stream: the window is topped up by all lanes together every 8th
#define MKZ_TOPUP()                                                                                   \
    do {                                                                                              \
        for (int k_ = 0; k_ < kWinWords; ++k_)                                                        \
            if (wr - rd < (uint32_t)kWinWords) {                                                      \
                win[wr & (kWinWords - 1)] = wpos <= n_in + 8 ? load_le32(in + wpos) : 0u;             \
                ++wr, wpos += 4;                                                                      \
            }                                                                                         \
    } while (0)
// >= 32 valid bits in the buffer (the top-up inside is the exception: the cadence of the token loop keeps the ring filled)
#define MKZ_NEED32()                                            \
    do {                                                        \
        if (bitcnt < 32) {                                      \
            if (rd == wr) MKZ_TOPUP();                          \
            bitbuf |= (uint64_t)win[rd & (kWinWords - 1)] << bitcnt; \
            ++rd, bitcnt += 32;                                 \
        }                                                       \
    } while (0)
#define MKZ_TAKEN() (wpos - 4 * (wr - rd))  // stream bytes handed to the bit buffer
#define MKZ_RAN_OUT() (MKZ_TAKEN() > n_in + 8)
#define MKZ_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
    for (;;) {
        MKZ_NEED32();
        if (MKZ_RAN_OUT()) return kInfTruncated;
        const uint32_t final_block = (uint32_t)bitbuf & 1u, type = ((uint32_t)bitbuf >> 1) & 3u;
        MKZ_TAKE(3);
        if (type == 0) {  // stored: skip to the byte boundary, LEN, ~LEN, bytes
            MKZ_TAKE(bitcnt & 7);
            MKZ_NEED32();
            const uint32_t len = (uint32_t)bitbuf & 0xffffu, nlen = ((uint32_t)(bitbuf >> 16)) & 0xffffu;
            MKZ_TAKE(32);
            if ((len ^ nlen) != 0xffffu) return kInfBadStored;
            // the bytes follow in the stream: what the bit buffer was handed, minus what it still holds
            const uint32_t pos = MKZ_TAKEN() - (bitcnt >> 3);
            if (MKZ_RAN_OUT() || pos > n_in || len > n_in - pos) return kInfTruncated;
            if (len > (uint32_t)(out_end - op)) return kInfOutputOverrun;
            const uint8_t *p = in + pos;
            uint32_t i = 0;
            for (; i + 8 <= len; i += 8) store_le64(op + i, load_le64(p + i));
            for (; i < len; ++i) op[i] = p[i];
            op += len;
            bitbuf = 0, bitcnt = 0, wpos = pos + len, wr = 0, rd = 0;  // the next block starts behind them
            if (final_block) break;
            continue;
        }
        if (type == 3) return kInfBadBlockType;
        if (type == 1) {
            // the fixed code of RFC 1951 3.2.6: lengths 7 (256..279), 8 (0..143, 280..287), 9 (144..255); 32 distance codewords of
            // 5 bits (symbols 30 and 31 complete the code and are refused when met)
            uint64_t ca = 0, cb = 0, cc = 0;
            p15_add(ca, cb, cc, 7, 24), p15_add(ca, cb, cc, 8, 152), p15_add(ca, cb, cc, 9, 112);
            (void)tables_from_counts(ca, cb, cc, 288, ll_limit, ll_base, true);
            // rank k -> symbol: 256..279, then 0..143, then 280..287, then 144..255
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
            for (uint32_t k = 0; k < 288; ++k) ll_sorted[k] = (uint16_t)(k < 24 ? 256 + k : k < 168 ? k - 24 : k < 176 ? k + 112 : k - 32);
            ca = cb = cc = 0;
            p15_add(ca, cb, cc, 5, 32);
            (void)tables_from_counts(ca, cb, cc, 32, d_limit, d_base, true);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
            for (uint32_t k = 0; k < 32; ++k) d_sorted[k] = (uint16_t)k;
        } else {
            const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = ((uint32_t)(bitbuf >> 5) & 31u) + 1,
                           hclen = ((uint32_t)(bitbuf >> 10) & 15u) + 4;
            MKZ_TAKE(14);
            if (hlit > 286 || hdist > 30) return kInfBadLengths;
            // the code-length code: 19 lengths of 3 bits, sent in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            uint64_t cl = 0;
            {
                const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                                          5ull << 45 | 11ull << 50 | 4ull << 55;
                const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
                for (uint32_t i = 0; i < hclen; ++i) {
                    MKZ_NEED32();
                    const uint32_t sym = (uint32_t)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
                    cl |= (uint64_t)((uint32_t)bitbuf & 7u) << (3 * sym);
                    MKZ_TAKE(3);
                }
            }
            ClCode clc;
            if (cl_build(cl, clc)) return kInfBadLengths;
            // where the code lengths begin, in bits of the stream: pass 2 starts here again
            const uint64_t lens_at = (uint64_t)MKZ_TAKEN() * 8 - bitcnt;
            uint64_t la = 0, lb = 0, lc = 0, da = 0, db = 0, dc = 0;  // (Pack15 each) pass 1: codewords per length; pass 2: the next free rank of every length
            uint32_t ll_used = 0, d_used = 0;
            bool has_eob = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    if (!has_eob) return kInfBadLengths;  // no end-of-block codeword
                    if (tables_from_counts(la, lb, lc, ll_used, ll_limit, ll_base, true)) return kInfBadLengths;
                    if (tables_from_counts(da, db, dc, d_used, d_limit, d_base, true)) return kInfBadLengths;
                    // back to the first code length
                    const uint32_t byte = (uint32_t)(lens_at >> 3);
                    bitbuf = 0, bitcnt = 0, wpos = byte, wr = 0, rd = 0;
                    MKZ_NEED32();
                    MKZ_TAKE((uint32_t)lens_at & 7u);
                }
                uint32_t i = 0, prev = 0;
                while (i < hlit + hdist) {
                    MKZ_NEED32();
                    if (MKZ_RAN_OUT()) return kInfTruncated;
                    const uint32_t e = cl_decode((uint32_t)bitbuf, clc);
                    if (e == 0) return kInfBadLengths;
                    MKZ_TAKE(e & 15u);
                    const uint32_t sym = e >> 4;
                    uint32_t rep = 1, val = sym;
                    if (sym == 16) {
                        if (i == 0) return kInfBadLengths;
                        val = prev;
                        rep = 3 + ((uint32_t)bitbuf & 3u);
                        MKZ_TAKE(2);
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + ((uint32_t)bitbuf & 7u);
                        MKZ_TAKE(3);
                    } else if (sym == 18) {
                        val = 0;
                        rep = 11 + ((uint32_t)bitbuf & 127u);
                        MKZ_TAKE(7);
                    }
                    if (i + rep > hlit + hdist) return kInfBadLengths;
                    if (val) {  // (rep <= 6 here: runs of a non-zero length come from symbol 16 or are single)
                        for (uint32_t j = i; j < i + rep; ++j) {
                            const bool lit = j < hlit;
                            const uint32_t symbol = lit ? j : j - hlit;
                            if (pass == 0) {
                                if (lit) ++ll_used, has_eob = has_eob || symbol == 256;
                                else ++d_used;
                            } else if (lit) {
                                ll_sorted[p15_get(la, lb, lc, val)] = (uint16_t)symbol;
                            } else {
                                d_sorted[p15_get(da, db, dc, val)] = (uint16_t)symbol;
                            }
                            if (lit) p15_add(la, lb, lc, val, 1);  // (pass 1 counts, pass 2 moves on to the next rank: the same step)
                            else p15_add(da, db, dc, val, 1);
                        }
                    }
                    prev = val, i += rep;
                }
            }
        }
        // Symbols of this block.  One token per turn of the loop, and a turn never waits for memory it has just asked
        // for (the 64 lanes of a wave turn together: whatever one of them waits for, all wait for):
        //  * a match is copied in rounds of <= 32 bytes, at most one round per turn.  The loads of a round are issued at
        //    the END of a turn; its stores happen in the next turn, AFTER that turn's decode -- the L2 round trip runs
        //    beside the decode's LDS work instead of in front of it.  A long match keeps its own lane copying for several
        //    turns while the others decode (a copy loop inside the turn made every lane wait for the longest match of the
        //    64, one round trip per 8 bytes);
        //  * the window is topped up by all lanes together on every 8th turn (a turn takes at most 48 bits, the ring
        //    holds 512): one round trip per 8 turns, where "reload when empty" put one on almost every turn -- some
        //    lane of the 64 is always empty.
        uint32_t pend = 0, pdist = 0;  // bytes of the current match not yet asked for, its distance
        uint32_t cn = 0, cperiod = 0;  // the round in flight: bytes to store at cdst (0 = none), pattern period (0 = plain)
        uint8_t *cdst = op;
        uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        int status = 1;  // 1 = in the block, 0 = end of block, < 0 = error
        while (status == 1) {
            if ((turn++ & 7u) == 0) MKZ_TOPUP();
            bool literal = false;
            if (pend == 0) {
                MKZ_NEED32();
                const uint32_t e = decode_codeword((uint32_t)bitbuf, ll_sorted, ll_limit, ll_base);
                MKZ_TAKE(e & 15u);
                const uint32_t sym = e >> 4;
                if (e == 0 || MKZ_RAN_OUT()) {
                    status = e == 0 ? kInfBadSymbol : kInfTruncated;
                } else if (sym < 256) {
                    if (op == out_end) status = kInfOutputOverrun;
                    else *op++ = (uint8_t)sym, literal = true;
                } else if (sym == 256) {
                    status = 0;
                } else if (sym > 285) {
                    status = kInfBadSymbol;
                } else {
                    const uint32_t idx = sym - 257;
                    const uint32_t leb = length_extra_bits(idx);
                    const uint32_t len = length_base(idx) + ((uint32_t)bitbuf & ((1u << leb) - 1));
                    MKZ_TAKE(leb);
                    MKZ_NEED32();
                    const uint32_t d = decode_codeword((uint32_t)bitbuf, d_sorted, d_limit, d_base);
                    MKZ_TAKE(d & 15u);
                    const uint32_t dsym = d >> 4;
                    const uint32_t deb = dsym < 30 ? distance_extra_bits(dsym) : 0;
                    const uint32_t dist = dsym < 30 ? distance_base(dsym) + ((uint32_t)bitbuf & ((1u << deb) - 1)) : 0;
                    MKZ_TAKE(deb);
                    if (d == 0 || dsym > 29) status = kInfBadSymbol;
                    else if (dist > (uint32_t)(op - out)) status = kInfBadDistance;
                    else if (len > (uint32_t)(out_end - op)) status = kInfOutputOverrun;
                    else pend = len, pdist = dist;
                }
            }
            // the round asked for in the previous turn: its bytes have had this turn's decode to arrive
            if (cn) {
                if (cperiod == 0) {
                    uint32_t k = 0;
                    if (cn >= 8) store_le64(cdst, v0), k = 8;
                    if (cn >= 16) store_le64(cdst + 8, v1), k = 16;
                    if (cn >= 24) store_le64(cdst + 16, v2), k = 24;
                    if (cn >= 32) store_le64(cdst + 24, v3), k = 32;
                    uint64_t tail = k == 0 ? v0 : k == 8 ? v1 : k == 16 ? v2 : v3;
                    for (; k < cn; ++k, tail >>= 8) cdst[k] = (uint8_t)tail;
                } else {  // a period of 1..7 bytes: the pattern, spread over a register, is stored a whole number of periods at a time
                    uint64_t pat = 0;
                    for (uint32_t j = 0; j < 8; ++j) pat |= ((v0 >> (8 * (j % cperiod))) & 0xffull) << (8 * j);
                    const uint32_t step = 8 - 8 % cperiod;
                    uint32_t k = 0;
                    for (; k + 8 <= cn; k += step) store_le64(cdst + k, pat);
                    for (uint32_t j = 0; k < cn; ++k, ++j) cdst[k] = (uint8_t)(pat >> (8 * j));
                }
                cn = 0;
            }
            // the next round of the match in hand: ask for its bytes (their source may be what was just stored)
            if (pend && !literal && status == 1) {
                const uint8_t *src = op - pdist;
                cdst = op;
                v0 = load_le64(src);  // (loads may reach up to 7 bytes past what is used: inside the output buffer or its 8 bytes of slack)
                if (pdist >= 8) {
                    // n <= distance: the round reads nothing it writes
                    const uint32_t n = pend < 32 ? (pend < pdist ? pend : pdist) : (pdist < 32 ? pdist : 32);
                    if (n > 8) v1 = load_le64(src + 8);
                    if (n > 16) v2 = load_le64(src + 16);
                    if (n > 24) v3 = load_le64(src + 24);
                    cn = n, cperiod = 0;
                } else {
                    cn = pend, cperiod = pdist;  // the whole run from one load: its first `distance` bytes are the pattern
                }
                op += cn, pend -= cn;
            }
        }
        if (status < 0) return status;
        if (MKZ_RAN_OUT()) return kInfTruncated;
        if (final_block) break;
    }
    // bits consumed must lie inside the stream
    if (MKZ_RAN_OUT() || (uint64_t)MKZ_TAKEN() - (bitcnt >> 3) > (uint64_t)n_in) return kInfTruncated;
#undef MKZ_TOPUP
#undef MKZ_NEED32
#undef MKZ_TAKEN
#undef MKZ_RAN_OUT
#undef MKZ_TAKE
    return op == out_end ? 0 : kInfOutputShort;
}

}  // namespace mkz
