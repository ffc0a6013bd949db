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
// Only the code lengths while a block header is read sit in private memory (once per block).
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

// Tables of one code from lens[0..n): sorted[] (symbols in (length, symbol) order), limit[l] = the first left-aligned
// 15-bit value behind the codewords of length l (ascending in l), base[l] = (rank of the first symbol of length l) -
// (its codeword), mod 2^16.  Returns 0 / 1 over-subscribed / 2 incomplete (usable only as zlib's inflate_table
// accepts it: no codeword at all, or -- allow_single: literal / length and distance codes -- exactly one codeword of
// length 1).
MKZ_HD int build_decode_tables(const uint8_t *lens, int n, uint16_t *sorted, uint16_t *limit, uint16_t *base, bool allow_single) {
    uint32_t count[kMaxBits + 1], offs[kMaxBits + 2];
    for (int l = 0; l <= kMaxBits; ++l) count[l] = 0;
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    const int used = n - (int)count[0];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= kMaxBits; ++l) {
        left <<= 1;
        left -= (int)count[l];
        if (left < 0) return 1;
    }
    uint32_t code = 0;
    offs[1] = 0;
    limit[0] = 0, base[0] = 0;
    for (int l = 1; l <= kMaxBits; ++l) {
        code = (code + count[l - 1]) << 1;
        offs[l + 1] = offs[l] + count[l];
        limit[l] = (uint16_t)((code + count[l]) << (kMaxBits - l));
        base[l] = (uint16_t)(offs[l] - code);
    }
    for (int i = 0; i < n; ++i)
        if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    if (left > 0 && !(used == 0 || (allow_single && used == 1 && count[1] == 1))) return 2;
    return 0;
}

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
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(limit);
    uint32_t l = 1;
    for (int k = 0; k < 8; ++k) {  // limits ascend with the length: count those w has reached (limit[0] = 0 stands for the start of l, limit[15] is left out)
        const uint32_t pair = lw[k];
        l += (k ? w >= (pair & 0xffffu) : 0u) + (k < 7 ? w >= (pair >> 16) : 0u);
    }
    if (w >= limit[l]) return 0;
    const uint32_t idx = (base[l] + (w >> (kMaxBits - l))) & 0xffffu;
    return (uint32_t)sorted[idx] << 4 | l;
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
        {
            uint8_t lens[288 + 32];
            if (type == 1) {
                fixed_code_lengths(lens, lens + 288);
            } else {
                const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = ((uint32_t)(bitbuf >> 5) & 31u) + 1,
                               hclen = ((uint32_t)(bitbuf >> 10) & 15u) + 4;
                MKZ_TAKE(14);
                if (hlit > 286 || hdist > 30) return kInfBadLengths;
                uint8_t order[kCl], cl_len[kCl];
                cl_order(order);
                for (int i = 0; i < kCl; ++i) cl_len[i] = 0;
                for (uint32_t i = 0; i < hclen; ++i) {
                    MKZ_NEED32();
                    cl_len[order[i]] = (uint8_t)((uint32_t)bitbuf & 7u);
                    MKZ_TAKE(3);
                }
                // the code-length code borrows the literal tables
                if (build_decode_tables(cl_len, kCl, ll_sorted, ll_limit, ll_base, false)) return kInfBadLengths;
                uint32_t i = 0;
                while (i < hlit + hdist) {
                    MKZ_NEED32();
                    if (MKZ_RAN_OUT()) return kInfTruncated;
                    const uint32_t e = decode_codeword((uint32_t)bitbuf, ll_sorted, ll_limit, ll_base);
                    if (e == 0 || (e & 15u) > (uint32_t)kMaxClBits) return kInfBadLengths;
                    MKZ_TAKE(e & 15u);
                    const uint32_t sym = e >> 4;
                    if (sym < 16) {
                        lens[i++] = (uint8_t)sym;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (i == 0) return kInfBadLengths;
                        val = lens[i - 1];
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
                    for (; rep; --rep) lens[i++] = (uint8_t)val;
                }
                if (lens[256] == 0) return kInfBadLengths;  // no end-of-block codeword
                // distance lengths behind the literal ones, each alphabet padded with zeros to its full size
                uint8_t *const dl = lens + 288;
                for (int k = (int)hdist - 1; k >= 0; --k) dl[k] = lens[hlit + (uint32_t)k];
                for (uint32_t k = hdist; k < 32; ++k) dl[k] = 0;
                for (uint32_t k = hlit; k < 288; ++k) lens[k] = 0;
            }
            if (build_decode_tables(lens, 288, ll_sorted, ll_limit, ll_base, true)) return kInfBadLengths;
            // (the fixed distance code is 32 codewords of 5 bits: symbols 30 and 31 complete it and are refused when met)
            if (build_decode_tables(lens + 288, type == 1 ? 32 : 30, d_sorted, d_limit, d_base, true)) return kInfBadLengths;
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
