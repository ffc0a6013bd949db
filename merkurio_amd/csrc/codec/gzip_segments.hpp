// gzip_segments.hpp -- ONE DEFLATE stream decoded in parallel pieces (r05): the serial parts, shared by the kernels of
// gzip_inflate.hip and by the host harness (tests/helpers/gzip_harness.cpp) that checks them against zlib.
//
// What this replaces: needletail's gzip reader under `merkurio extract` (src/cmd_extract.rs:281-282) on the input it meets most --
// a FASTQ / FASTA compressed by plain gzip: ONE member, one DEFLATE stream of thousands of blocks, which zlib can only walk from the
// front (0.5 GB/s of text on one host thread, 2.6 s per 1.27 GB end to end in r04).  A BGZF file is cut into independent members; a gzip
// stream is not -- but its blocks can be FOUND, and decoded without knowing what came before (the scheme of pugz / rapidgzip,
// restated for a GPU):
//   1. block starts.  A dynamic block opens with a header that almost no bit position satisfies: HLIT / HDIST in range, a COMPLETE
//      code-length code, code lengths that decode to a complete literal / length code with an end-of-block codeword and a usable
//      distance code.  seg_header_plausible() tests one bit position with registers only.  Over 5.7 G bit positions of gzip -1 / -6
//      / -9 FASTQ streams and of random bytes it passed at the 22 000 block starts and at 3 other places; those give codes whose
//      first codewords are no symbols, or a block that ends after a handful of symbols in front of something that is no header -- so a
//      position that passes is looked at by decoding the first 2 048 symbols of its block dry (gzip_segments_wave.hpp; the first
//      version decoded the whole block, seg_confirm_block_start: a second decode of a quarter to all of the stream).  What PROVES
//      a start is step 2 -- the piece in front of it ends exactly there -- and a start that a piece runs over is dropped and the
//      pieces are decoded again.  The kernel tries 64 positions per wave and step, one wave per nominal chunk of the stream.
//   2. segments.  The stream is cut at the confirmed starts; inflate_segment() decodes [start, next start) into 16-bit symbols: a
//      value below 256 is a byte; a match that reaches back beyond the segment's start copies from a 32 768-element prefix that the
//      caller has filled with place-holders 0x8000 | k ("byte k of the 32 KiB in front of this segment") -- so every match is a plain
//      copy inside the segment's buffer, and place-holders travel through later matches like bytes do.
//   3. resolution.  Segment by segment, the last 32 KiB of resolved text become the next segment's context (sequential, 32 KiB per
//      step); then every symbol of every segment is translated in parallel and the text is checked against the member's CRC-32 / ISIZE.
// Anything that does not add up -- no block start found where one is needed, a segment that does not end exactly on the next start,
// a buffer that is too small for an unusually compressible stream, a wrong CRC -- makes the whole call report "not taken": the
// caller inflates that file with zlib, as before.  RFC 1951 / RFC 1952.
#pragma once
#include "inflate_serial.hpp"

namespace mkz {

constexpr uint32_t kSegPrefix = 32768;     // place-holder elements in front of a segment's output
constexpr uint16_t kSegUnknown = 0x8000u;  // | index into the 32 KiB in front of the segment
constexpr uint32_t kSegSlack = 16;         // elements a segment buffer holds behind its capacity (8-byte copy rounds overshoot)

// error codes (negative; >= 0 are fine)
constexpr int kSegDesync = -20, kSegOverflow = -21, kSegNoStart = -22;

// ---- bit reader: straight from global memory, one dword ahead --------------------------------------------------------------------
struct SegReader {
    const uint8_t *in;
    uint64_t n_in;  // bytes of the stream; readable up to n_in + kStreamPad
    uint64_t pos;   // byte offset of the dword behind `ahead`
    uint64_t bitbuf;
    uint32_t bitcnt, ahead;
};
MKZ_HD uint32_t sr_load(const SegReader &r, uint64_t byte) { return byte <= r.n_in + 8 ? load_le32(r.in + byte) : 0u; }
MKZ_HD void sr_need32(SegReader &r) {
    if (r.bitcnt < 32) {
        r.bitbuf |= (uint64_t)r.ahead << r.bitcnt;
        r.bitcnt += 32;
        r.ahead = sr_load(r, r.pos);  // (asked for one refill before it is needed)
        r.pos += 4;
    }
}
MKZ_HD void sr_take(SegReader &r, uint32_t n) { r.bitbuf >>= n, r.bitcnt -= n; }
MKZ_HD uint64_t sr_bitpos(const SegReader &r) { return (r.pos - 4) * 8 - r.bitcnt; }  // bits consumed so far
MKZ_HD bool sr_ran_out(const SegReader &r) { return sr_bitpos(r) > r.n_in * 8; }
MKZ_HD void sr_seek(SegReader &r, uint64_t bit) {
    r.pos = bit >> 3;
    r.bitbuf = 0, r.bitcnt = 0;
    r.ahead = sr_load(r, r.pos);
    r.pos += 4;
    sr_need32(r);
    sr_take(r, (uint32_t)bit & 7u);
}

// completeness of a code from its counts per length (Pack15 fields): 0 complete, 1 over-subscribed, 2 incomplete
MKZ_HD int p15_kraft(uint64_t a, uint64_t b, uint64_t c) {
    int left = 1;
    for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
        left = (left << 1) - (int)p15_get(a, b, c, l);
        if (left < 0) return 1;
    }
    return left > 0 ? 2 : 0;
}

// A dynamic block's header at the reader's position (behind BFINAL / BTYPE).  t == nullptr: only checked (registers only) -- strict:
// what a block start must look like to be believed (complete codes).  t != nullptr: the decode tables are built into t (layout of
// inflate_serial.hpp) with the decoder's own rules (zlib's).  Returns 0 or a negative kInf* code.
MKZ_HD int seg_dynamic_header(SegReader &r, uint16_t *t) {
    sr_need32(r);
    const uint32_t hlit = ((uint32_t)r.bitbuf & 31u) + 257, hdist = ((uint32_t)(r.bitbuf >> 5) & 31u) + 1, hclen = ((uint32_t)(r.bitbuf >> 10) & 15u) + 4;
    sr_take(r, 14);
    if (hlit > 286 || hdist > 30) return kInfBadLengths;
    uint64_t cl = 0;
    for (uint32_t i = 0; i < hclen; ++i) {
        sr_need32(r);
        cl |= (uint64_t)((uint32_t)r.bitbuf & 7u) << (3 * cl_order_at(i));
        sr_take(r, 3);
    }
    ClCode clc;
    if (cl_build(cl, clc)) return kInfBadLengths;
    if (!t && cl == 0) return kInfBadLengths;  // (a block start without any code-length codeword is not one)
    const uint64_t lens_at = sr_bitpos(r);
    uint64_t la = 0, lb = 0, lc = 0, da = 0, db = 0, dc = 0;
    uint32_t ll_used = 0, d_used = 0;
    bool has_eob = false;
    uint16_t *ll_sorted = t ? t + kOffLlSorted : nullptr, *d_sorted = t ? t + kOffDSorted : nullptr;
    for (int pass = 0; pass < (t ? 2 : 1); ++pass) {
        if (pass == 1) {
            if (tables_from_counts(la, lb, lc, ll_used, t + kOffLlLimit, t + kOffLlBase, true)) return kInfBadLengths;
            if (tables_from_counts(da, db, dc, d_used, t + kOffDLimit, t + kOffDBase, true)) return kInfBadLengths;
            sr_seek(r, lens_at);
        }
        uint32_t i = 0, prev = 0;
        while (i < hlit + hdist) {
            sr_need32(r);
            if (sr_ran_out(r)) return kInfTruncated;
            const uint32_t e = cl_decode((uint32_t)r.bitbuf, clc);
            if (e == 0) return kInfBadLengths;
            sr_take(r, e & 15u);
            const uint32_t sym = e >> 4;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (i == 0) return kInfBadLengths;
                val = prev;
                rep = 3 + ((uint32_t)r.bitbuf & 3u);
                sr_take(r, 2);
            } else if (sym == 17) {
                val = 0;
                rep = 3 + ((uint32_t)r.bitbuf & 7u);
                sr_take(r, 3);
            } else if (sym == 18) {
                val = 0;
                rep = 11 + ((uint32_t)r.bitbuf & 127u);
                sr_take(r, 7);
            }
            if (i + rep > hlit + hdist) return kInfBadLengths;
            if (val) {
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
                    if (lit) p15_add(la, lb, lc, val, 1);
                    else p15_add(da, db, dc, val, 1);
                }
            }
            prev = val, i += rep;
        }
        if (pass == 0) {
            if (!has_eob) return kInfBadLengths;
            if (!t) {  // a block start: both codes as a compressor writes them
                if (p15_kraft(la, lb, lc) != 0) return kInfBadLengths;
                if (d_used > 1 && p15_kraft(da, db, dc) != 0) return kInfBadLengths;
                if (d_used == 1 && p15_get(da, db, dc, 1) != 1) return kInfBadLengths;
            }
        }
    }
    return 0;
}

// the fixed code's tables (RFC 1951 3.2.6) into t
MKZ_HD void seg_fixed_tables(uint16_t *t) {
    uint64_t ca = 0, cb = 0, cc = 0;
    p15_add(ca, cb, cc, 7, 24), p15_add(ca, cb, cc, 8, 152), p15_add(ca, cb, cc, 9, 112);
    (void)tables_from_counts(ca, cb, cc, 288, t + kOffLlLimit, t + kOffLlBase, true);
    for (uint32_t k = 0; k < 288; ++k) t[kOffLlSorted + k] = (uint16_t)(k < 24 ? 256 + k : k < 168 ? k - 24 : k < 176 ? k + 112 : k - 32);
    ca = cb = cc = 0;
    p15_add(ca, cb, cc, 5, 32);
    (void)tables_from_counts(ca, cb, cc, 32, t + kOffDLimit, t + kOffDBase, true);
    for (uint32_t k = 0; k < 32; ++k) t[kOffDSorted + k] = (uint16_t)k;
}

// The first two levels of that test on their own (the search kernel runs them on all positions / on the survivors, so that the lanes of
// a wave do the same work at the same time).  Level 1: BFINAL = 0, BTYPE = dynamic, HLIT / HDIST in range -- the 13 bits `v` at the
// position (1 in 9 passes).  Level 2: the code-length code is complete and not empty (~1 in 100 of those).
MKZ_HD bool seg_header_bits_plausible(uint32_t v) { return (v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u; }
MKZ_HD bool seg_header_cl_plausible(const uint8_t *in, uint64_t n_in, uint64_t bit) {
    if (bit + 64 > n_in * 8) return false;
    SegReader r;
    r.in = in, r.n_in = n_in;
    sr_seek(r, bit);
    if (!seg_header_bits_plausible((uint32_t)r.bitbuf)) return false;
    const uint32_t hclen = ((uint32_t)(r.bitbuf >> 13) & 15u) + 4;
    sr_take(r, 17);
    uint64_t cl = 0;
    for (uint32_t i = 0; i < hclen; ++i) {
        sr_need32(r);
        cl |= (uint64_t)((uint32_t)r.bitbuf & 7u) << (3 * cl_order_at(i));
        sr_take(r, 3);
    }
    ClCode clc;
    return cl != 0 && cl_build(cl, clc) == 0;
}

// is `bit` the start of a non-final dynamic block, as far as its header says?  (registers only)
MKZ_HD bool seg_header_plausible(const uint8_t *in, uint64_t n_in, uint64_t bit) {
    if (bit + 64 > n_in * 8) return false;
    SegReader r;
    r.in = in, r.n_in = n_in;
    sr_seek(r, bit);
    if (((uint32_t)r.bitbuf & 7u) != 4u) return false;  // BFINAL = 0, BTYPE = 10 (dynamic): bits 0 | 0 1 -> value 4
    sr_take(r, 3);
    return seg_dynamic_header(r, nullptr) == 0;
}

// Decodes from bit0 (a block start).  Stops in front of the block that would start at bit_end (reaching it exactly is the proof
// that both ends are block starts), or behind the final block (*final_seen), or -- max_blocks > 0 -- after that many blocks.
// out: 16-bit symbols, out[-kSegPrefix .. -1] are the caller's place-holders; cap elements (+ kSegSlack writable).  out == nullptr:
// dry run (nothing is stored or copied: the bit stream alone is checked).  t: kLaneTableU16 16-bit words of table memory.
// Returns 0 or a negative code; *n_out = elements produced, *bit_stop = the bit position behind the last block decoded.
MKZ_HD int inflate_segment(const uint8_t *in, uint64_t n_in, uint64_t bit0, uint64_t bit_end, uint32_t max_blocks, uint16_t *out, uint64_t cap, uint16_t *t,
                           uint64_t *n_out, uint64_t *bit_stop, bool *final_seen) {
    SegReader r;
    r.in = in, r.n_in = n_in;
    sr_seek(r, bit0);
    uint64_t op = 0;
    *final_seen = false;
    uint32_t blocks = 0;
    for (;;) {
        const uint64_t at = sr_bitpos(r);
        if (at == bit_end || (max_blocks && blocks == max_blocks)) break;
        if (at > bit_end) {  // passed over bit_end: it is not a block start (*bit_stop: the block boundary behind it)
            *n_out = op, *bit_stop = at;
            return kSegDesync;
        }
        sr_need32(r);
        if (sr_ran_out(r)) return kInfTruncated;
        const uint32_t final_block = (uint32_t)r.bitbuf & 1u, type = ((uint32_t)r.bitbuf >> 1) & 3u;
        sr_take(r, 3);
        ++blocks;
        if (type == 3) return kInfBadBlockType;
        if (type == 0) {
            sr_take(r, r.bitcnt & 7);
            sr_need32(r);
            const uint32_t len = (uint32_t)r.bitbuf & 0xffffu, nlen = ((uint32_t)(r.bitbuf >> 16)) & 0xffffu;
            sr_take(r, 32);
            if ((len ^ nlen) != 0xffffu) return kInfBadStored;
            const uint64_t pos = sr_bitpos(r) >> 3;  // (byte-aligned here)
            if (sr_ran_out(r) || pos > n_in || len > n_in - pos) return kInfTruncated;
            if (len > cap - op) return kSegOverflow;
            if (out)
                for (uint32_t i = 0; i < len; ++i) out[op + i] = in[pos + i];
            op += len;
            sr_seek(r, (pos + len) * 8);
        } else {
            if (type == 1) {
                seg_fixed_tables(t);
            } else {
                const int rc = seg_dynamic_header(r, t);
                if (rc) return rc;
            }
            const uint16_t *ll_sorted = t + kOffLlSorted, *d_sorted = t + kOffDSorted, *ll_limit = t + kOffLlLimit, *ll_base = t + kOffLlBase,
                           *d_limit = t + kOffDLimit, *d_base = t + kOffDBase;
            for (;;) {
                sr_need32(r);
                const uint32_t e = decode_codeword((uint32_t)r.bitbuf, ll_sorted, ll_limit, ll_base);
                sr_take(r, e & 15u);
                const uint32_t sym = e >> 4;
                if (e == 0) return kInfBadSymbol;
                if (sr_ran_out(r)) return kInfTruncated;
                if (sym < 256) {
                    if (op >= cap) return kSegOverflow;
                    if (out) out[op] = (uint16_t)sym;
                    ++op;
                    continue;
                }
                if (sym == 256) break;
                if (sym > 285) return kInfBadSymbol;
                const uint32_t idx = sym - 257;
                const uint32_t leb = length_extra_bits(idx);
                const uint32_t len = length_base(idx) + ((uint32_t)r.bitbuf & ((1u << leb) - 1));
                sr_take(r, leb);
                sr_need32(r);
                const uint32_t d = decode_codeword((uint32_t)r.bitbuf, d_sorted, d_limit, d_base);
                sr_take(r, d & 15u);
                const uint32_t dsym = d >> 4;
                if (d == 0 || dsym > 29) return kInfBadSymbol;
                const uint32_t deb = distance_extra_bits(dsym);
                const uint32_t dist = distance_base(dsym) + ((uint32_t)r.bitbuf & ((1u << deb) - 1));
                sr_take(r, deb);
                if (len > cap - op) return kSegOverflow;
                if (out) {
                    // the match as bytes of the 16-bit buffer: 2 len bytes from 2 dist bytes back (the place-holder prefix is
                    // ordinary memory: a match that reaches in front of the segment copies place-holders)
                    uint8_t *dst = reinterpret_cast<uint8_t *>(out + op);
                    const uint8_t *src = dst - 2 * (uint64_t)dist;
                    const uint32_t nb = 2 * len;
                    if (dist >= 4) {  // 8 bytes a round: a round never reads what it writes
                        for (uint32_t k = 0; k < nb; k += 8) store_le64(dst + k, load_le64(src + k));
                    } else {  // a period of 2, 4 or 6 bytes, spread over a register and stored a whole number of periods at a time
                        const uint64_t v0 = load_le64(src);
                        const uint32_t period = 2 * dist;
                        uint64_t pat = 0;
                        for (uint32_t j = 0; j < 8; ++j) pat |= ((v0 >> (8 * (j % period))) & 0xffull) << (8 * j);
                        const uint32_t step = 8 - 8 % period;
                        for (uint32_t k = 0; k < nb; k += step) store_le64(dst + k, pat);
                    }
                }
                op += len;
            }
            if (sr_ran_out(r)) return kInfTruncated;
        }
        if (final_block) {
            *final_seen = true;
            break;
        }
    }
    *n_out = op;
    *bit_stop = sr_bitpos(r);
    return 0;
}

// a plausible header at `bit`: is it a block start?  Its block decodes cleanly (dry) and another block header follows at once --
// a stored block or a plausible dynamic one.  The whole-block form: the host harness cuts its streams with it (and the search
// kernel's first version did); the kernel now decodes the first symbols only (wave_confirm_block_start, gzip_segments_wave.hpp).
MKZ_HD bool seg_confirm_block_start(const uint8_t *in, uint64_t n_in, uint64_t bit, uint16_t *t) {
    uint64_t n_out = 0, stop = 0;
    bool fin = false;
    if (inflate_segment(in, n_in, bit, ~0ull, 1, nullptr, ~0ull, t, &n_out, &stop, &fin) != 0) return false;
    if (n_out == 0 || fin) return false;  // (the block at a nominal cut is neither empty nor the last one)
    if (stop + 3 > n_in * 8) return false;
    SegReader r;
    r.in = in, r.n_in = n_in;
    sr_seek(r, stop);
    const uint32_t type = ((uint32_t)r.bitbuf >> 1) & 3u;
    if (type == 3) return false;
    if (type == 2) {
        sr_take(r, 3);
        return seg_dynamic_header(r, nullptr) == 0;
    }
    if (type == 0) {  // stored: LEN and its complement must agree
        sr_take(r, 3);
        sr_take(r, r.bitcnt & 7);
        sr_need32(r);
        const uint32_t len = (uint32_t)r.bitbuf & 0xffffu, nlen = ((uint32_t)(r.bitbuf >> 16)) & 0xffffu;
        return (len ^ nlen) == 0xffffu;
    }
    return false;  // (a fixed block behind it proves nothing: such a start is passed over, the chunk joins its predecessor)
}

}  // namespace mkz
