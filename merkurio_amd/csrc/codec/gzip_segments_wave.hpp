// gzip_segments_wave.hpp -- a piece of ONE gzip stream (gzip_segments.hpp says what the pieces are) decoded by a WAVE (r05): what
// inflate_segment() does on a lane, symbol for symbol, the way bgzf_inflate_wave.hip decodes a BGZF member -- a wave-uniform bit buffer
// fed by v_readlane from 256 bytes of stream held a dword per lane, codewords through direct tables in LDS that carry bases and extra-bit
// counts, matches copied by the whole wave.  What differs from a member: the output is 16-bit symbols (a byte, or a place-holder for
// text in front of the piece); the piece starts at any BIT of the stream and ends in front of the block that starts at bit_end (or
// behind the final block, or -- max_blocks -- after that many blocks); a match may reach up to 32 768 symbols in front of the piece,
// where the caller has laid place-holders; and only the most recent 2 048 symbols live in LDS (4 KiB: 9.3 KiB per wave, 17 waves per
// CU) -- a match that reaches further reads symbols back from global memory, where they went with a flush (or where the place-holders
// lie).  kDry: nothing is stored or copied, the bit stream alone is checked, max_symbols of it at most (the block-start search looks at
// a candidate with it).  Users: gzip_segments_wave.hip (the pieces), gzip_inflate.hip (the search).  RFC 1951.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gzip_segments.hpp"
#include "inflate_wave_common.hpp"

namespace mkz {
namespace {

// MK_SEG_PAIRS=1 (a measurement build): table entries of 32 bits that hold TWO literals where the entry's bits do, taken in one turn.
// Measured on gzip -6 FASTQ: 14 instead of 17 waves per CU, pieces 31 -> 43 ms.  DNA text is not literals to DEFLATE: three tokens of
// four are matches of ~12 bases found thousands of positions back (any 8-mer has occurred in the last 32 KiB), and what a wave waits
// for is those matches' sources coming back from device memory -- hidden by the other waves of the CU, so waves per CU count.
#ifndef MK_SEG_PAIRS
#define MK_SEG_PAIRS 0
#endif
constexpr bool kSegPairs = MK_SEG_PAIRS != 0;
#ifndef MK_SEG_RING
#define MK_SEG_RING 2048  // (a measurement build may take another power of two >= 1024; 1024 = 21 waves per CU: pieces 31.2 -> 32.7 ms, small streams 15.5 -> 16.6)
#endif
constexpr uint32_t kSegRing = MK_SEG_RING, kSegRingMask = kSegRing - 1, kSegFlush = kSegRing / 2;  // (16-bit elements)
constexpr uint64_t kSegCapMax = 0xfff00000ull;  // (the decoder counts symbols in 32 bits)
constexpr int kSegCutShort = 1;                // (dry) max_symbols were decoded without an error: not an error
constexpr uint64_t kSegConfirmSymbols = 2048;  // what the search decodes of a candidate's block (a multiple of kSegFlush)

struct SegWaveTables {
    // ll_fast as bgzf_inflate_wave.hip has it (one codeword); kSegPairs: that is the low half, and the high half, if not 0, says that
    // the stream bits of this entry hold TWO literals -- second literal << 4 | length of both codewords
    std::conditional_t<kSegPairs, uint32_t, uint16_t> ll_fast[1 << kFastLl];
    uint32_t d_fast[1 << kFastD];
    uint16_t ll_limit[16], ll_base[16], d_limit[16], d_base[16];
    uint16_t ll_sorted[288], d_sorted[32];
    uint8_t lens[320];
};
static_assert(kSegRing != 2048 || sizeof(SegWaveTables) + 2 * kSegRing <= (kSegPairs ? 11648 : 9728), "fourteen (without pairs: seventeen) waves per CU");
static_assert(kSegRing >= 1024 && (kSegRing & (kSegRing - 1)) == 0, "the flush writes 512 symbols a step");

// All lanes call it with the same arguments and get the same results.  ring: kSegRing elements of LDS, 16-byte aligned (kDry: unused).
// out[-kSegPrefix .. -1]: the caller's place-holders; cap elements may be written (what counts is min(cap, kSegCapMax): a piece of more
// symbols than that is reported as one that does not fit).  Returns 0 or a negative code (kInf*, kSeg*);
// *n_out = symbols produced, *bit_stop = the bit position behind the last block decoded, *final_seen: that block was the final one.
template <bool kDry>
__device__ __forceinline__ int wave_inflate_segment(const uint8_t *__restrict__ in, uint64_t n_in, uint64_t bit0, uint64_t bit_end, uint32_t max_blocks,
                                    uint16_t *__restrict__ out, uint64_t cap, uint16_t *ring, SegWaveTables &S, uint64_t max_symbols, uint64_t *n_out,
                                    uint64_t *bit_stop, bool *final_seen) {
    const uint32_t lane = lane_id();
    // ---- the compressed stream: lane k holds dword (sbase + k) of the stream that starts at byte s0; `nxt` the 64 dwords behind
    uint64_t s0 = 0;
    uint32_t sbase = 0, rd = 0;  // rd = dwords handed to the bit buffer (relative to s0)
    uint32_t look = 0, nxt = 0;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    auto load_dword = [&](uint32_t dw) -> uint32_t {  // dword dw of the stream at s0; zeros behind n_in + 8 (a valid stream never takes them)
        const uint64_t byte = s0 + 4ull * dw;
        return byte <= n_in + 8 ? load_le32(in + byte) : 0u;
    };
    auto restart = [&](uint64_t byte_pos) {
        s0 = byte_pos, sbase = 0, rd = 0, bitbuf = 0, bitcnt = 0;
        look = load_dword(lane);
        nxt = load_dword(64 + lane);
    };
#define MKS_NEED32()                                                                              \
    do {                                                                                          \
        if (bitcnt < 32) {                                                                        \
            uint32_t k_ = uni(rd - sbase);                                                        \
            if (k_ == 64) {                                                                       \
                look = nxt, sbase += 64, k_ = 0;                                                  \
                nxt = load_dword(sbase + 64 + lane);                                              \
            }                                                                                     \
            bitbuf |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)look, (int)k_) << bitcnt; \
            bitcnt += 32, ++rd;                                                                   \
        }                                                                                         \
    } while (0)
#define MKS_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
#define MKS_BITPOS() ((s0 + 4ull * rd) * 8 - bitcnt)  // bits of the stream consumed so far
#define MKS_RAN_OUT() (MKS_BITPOS() > n_in * 8)
    restart(bit0 >> 3);
    MKS_NEED32();
    MKS_TAKE((uint32_t)bit0 & 7u);

    uint32_t op = 0, flushed = 0;  // symbols produced / symbols that have left for global memory
    const uint32_t cap32 = (uint32_t)(cap < kSegCapMax ? cap : kSegCapMax);
    int status = 0;
    bool fin = false;
    auto flush_blocks = [&]() {  // whole pieces of kSegFlush symbols: 8 symbols (16 bytes) per lane and step
        while (op - flushed >= kSegFlush) {
            if constexpr (!kDry) {
#pragma unroll
                for (uint32_t k = 0; k < kSegFlush; k += 512) {
                    const uint32_t at = flushed + k + 8 * lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(&ring[at & kSegRingMask]);
                    __builtin_memcpy(out + at, &v, 16);
                }
            }
            flushed += kSegFlush;
        }
        // (far matches read flushed symbols back: the stores above must have arrived before such a load is issued)
        // (this wave alone reads them, past the CU's L1: they need to have reached the L2, not to be written back from it -- an agent-scope
        // release is a write-back of the whole L2, `buffer_wbl2`, per flush)
        if constexpr (!kDry) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
    };

    for (uint32_t blocks = 0;; ++blocks) {
        const uint64_t at = MKS_BITPOS();
        if (at == bit_end || (max_blocks && blocks == max_blocks)) break;
        if (at > bit_end) {
            status = kSegDesync;
            break;
        }
        MKS_NEED32();
        if (MKS_RAN_OUT()) {
            status = kInfTruncated;
            break;
        }
        const bool final_block = ((uint32_t)bitbuf & 1u) != 0;
        const uint32_t type = ((uint32_t)bitbuf >> 1) & 3u;
        MKS_TAKE(3);
        if (type == 3) {
            status = kInfBadBlockType;
            break;
        }
        if (type == 0) {  // stored: skip to the byte boundary, LEN, ~LEN, bytes -- copied stream -> ring by all lanes
            MKS_TAKE(bitcnt & 7);
            MKS_NEED32();
            const uint32_t len = (uint32_t)bitbuf & 0xffffu, nlen = ((uint32_t)(bitbuf >> 16)) & 0xffffu;
            MKS_TAKE(32);
            const uint64_t pos = MKS_BITPOS() >> 3;  // (byte-aligned here)
            if ((len ^ nlen) != 0xffffu) status = kInfBadStored;
            else if (MKS_RAN_OUT() || pos > n_in || len > n_in - pos) status = kInfTruncated;
            else if (len > cap32 - op) status = kSegOverflow;
            if (status) break;
            for (uint32_t done = 0; done < len;) {
                const uint32_t piece = min(len - done, kSegFlush);
                if constexpr (!kDry)
                    for (uint32_t i = lane; i < piece; i += 64) ring[(op + i) & kSegRingMask] = in[pos + done + i];
                op += piece, done += piece;
                flush_blocks();
            }
            restart(pos + len);
        } else {
            // ---- code lengths of the block -> S.lens
            if (type == 1) {
                for (uint32_t i = lane; i < 320; i += 64) S.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
            } else {
                const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = ((uint32_t)(bitbuf >> 5) & 31u) + 1, hclen = ((uint32_t)(bitbuf >> 10) & 15u) + 4;
                MKS_TAKE(14);
                if (hlit > 286 || hdist > 30) {
                    status = kInfBadLengths;
                    break;
                }
                if (lane < (uint32_t)kCl) S.lens[lane] = 0;
                constexpr uint64_t kOrderLo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                                              5ull << 45 | 11ull << 50 | 4ull << 55;
                constexpr uint64_t kOrderHi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
                for (uint32_t i = 0; i < hclen; ++i) {
                    MKS_NEED32();
                    const uint32_t s_ = (uint32_t)((i < 12 ? kOrderLo >> (5 * i) : kOrderHi >> (5 * (i - 12))) & 31u);
                    if (lane == 0) S.lens[s_] = (uint8_t)((uint32_t)bitbuf & 7u);
                    MKS_TAKE(3);
                }
                // (the code-length code borrows the literal tables)
                if (wave_build_tables(S.lens, kCl, S.ll_sorted, S.ll_limit, S.ll_base, false)) {
                    status = kInfBadLengths;
                    break;
                }
                uint32_t i = 0, prev_len = 0;
                while (i < hlit + hdist && status == 0) {
                    MKS_NEED32();
                    if (MKS_RAN_OUT()) {
                        status = kInfTruncated;
                        break;
                    }
                    const uint32_t e = decode_codeword((uint32_t)bitbuf, S.ll_sorted, S.ll_limit, S.ll_base);
                    if (e == 0 || (e & 15u) > (uint32_t)kMaxClBits) {
                        status = kInfBadLengths;
                        break;
                    }
                    MKS_TAKE(e & 15u);
                    const uint32_t cs = e >> 4;
                    if (cs < 16) {
                        if (lane == 0) S.lens[i < hlit ? i : 288 + (i - hlit)] = (uint8_t)cs;
                        prev_len = cs, ++i;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (cs == 16) {
                        if (i == 0) {
                            status = kInfBadLengths;
                            break;
                        }
                        val = prev_len;
                        rep = 3 + ((uint32_t)bitbuf & 3u);
                        MKS_TAKE(2);
                    } else if (cs == 17) {
                        rep = 3 + ((uint32_t)bitbuf & 7u);
                        MKS_TAKE(3);
                    } else {
                        rep = 11 + ((uint32_t)bitbuf & 127u);
                        MKS_TAKE(7);
                    }
                    if (i + rep > hlit + hdist) {
                        status = kInfBadLengths;
                        break;
                    }
                    for (uint32_t q = lane; q < rep; q += 64) {
                        const uint32_t x = i + q;
                        S.lens[x < hlit ? x : 288 + (x - hlit)] = (uint8_t)val;
                    }
                    prev_len = val, i += rep;
                }
                if (status) break;
                for (uint32_t k = hlit + lane; k < 288; k += 64) S.lens[k] = 0;
                if (lane + hdist < 32) S.lens[288 + hdist + lane] = 0;
                __builtin_amdgcn_wave_barrier();
                if (S.lens[256] == 0) {  // no end-of-block codeword
                    status = kInfBadLengths;
                    break;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (wave_build_tables(S.lens, 288, S.ll_sorted, S.ll_limit, S.ll_base, true) ||
                wave_build_tables(S.lens + 288, type == 1 ? 32 : 30, S.d_sorted, S.d_limit, S.d_base, true)) {
                status = kInfBadLengths;
                break;
            }
            __builtin_amdgcn_wave_barrier();
            // one codeword per entry ...
            for (uint32_t e = lane; e < (1u << kFastLl); e += 64) {
                const uint32_t r = decode_codeword(e, S.ll_sorted, S.ll_limit, S.ll_base);
                S.ll_fast[e] = (r & 15u) <= (uint32_t)kFastLl ? pack_ll(r) : 0u;
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (kSegPairs) {
                // ... and where that is a literal and what follows it in the entry's bits is a whole literal too, the two of them (only
                // the high halves are written here, only the low halves read)
                for (uint32_t e = lane; e < (1u << kFastLl); e += 64) {
                    const uint32_t a = reinterpret_cast<const uint16_t *>(S.ll_fast)[2 * e];
                    const uint32_t la = a & 15u;
                    if (a != 0 && a < 0x1000u && la < (uint32_t)kFastLl) {
                        const uint32_t b = reinterpret_cast<const uint16_t *>(S.ll_fast)[2 * (e >> la)];  // (the bits behind the first, zeros above)
                        if (b != 0 && b < 0x1000u && la + (b & 15u) <= (uint32_t)kFastLl) reinterpret_cast<uint16_t *>(S.ll_fast)[2 * e + 1] = (uint16_t)((b & 0xff0u) | (la + (b & 15u)));
                    }
                }
            }
            wave_fill_fast_d(S.d_fast, S.d_sorted, S.d_limit, S.d_base);
            __builtin_amdgcn_wave_barrier();

            // ---- the symbols of the block (the turn of bgzf_inflate_wave.hip; capacity and stream end are asked at the flush points)
            uint32_t flush_at = flushed + kSegFlush;
            for (;;) {
                MKS_NEED32();
                uint32_t e = S.ll_fast[(uint32_t)bitbuf & ((1u << kFastLl) - 1)];
                if (kSegPairs && e > 0xffffu) {  // two literals
                    if constexpr (!kDry) {
                        ring[op & kSegRingMask] = (uint16_t)((e >> 4) & 0xffu);
                        ring[(op + 1) & kSegRingMask] = (uint16_t)(e >> 20);
                    }
                    MKS_TAKE((e >> 16) & 15u);
                    op += 2;
                    if (op < flush_at) continue;
                    e = 0x10000u;  // (the flush point's questions, then on)
                } else if (e == 0) {
                    e = pack_ll(decode_codeword((uint32_t)bitbuf, S.ll_sorted, S.ll_limit, S.ll_base));
                    if (e == 0) {
                        status = kInfBadSymbol;
                        break;
                    }
                }
                if (e != 0x10000u) {
                    MKS_TAKE(e & 15u);
                    if (e < 0x1000u) {
                        if constexpr (!kDry) ring[op & kSegRingMask] = (uint16_t)(e >> 4);
                        ++op;
                        if (op < flush_at) continue;
                    } else if (e < 0x8000u) {
                        break;
                    } else {
                        const uint32_t leb = (e >> 4) & 7u;
                        const uint32_t len = ((e >> 7) & 255u) + 3u + ((uint32_t)bitbuf & ((1u << leb) - 1));
                        MKS_TAKE(leb);
                        MKS_NEED32();
                        uint32_t d = S.d_fast[(uint32_t)bitbuf & ((1u << kFastD) - 1)];
                        if (d == 0) {
                            d = pack_d(decode_codeword((uint32_t)bitbuf, S.d_sorted, S.d_limit, S.d_base));
                            if (d == 0) {
                                status = kInfBadSymbol;
                                break;
                            }
                        }
                        MKS_TAKE(d & 15u);
                        const uint32_t deb = (d >> 4) & 15u;
                        const uint32_t dist = (d >> 8) + ((uint32_t)bitbuf & ((1u << deb) - 1));
                        MKS_TAKE(deb);
                        if constexpr (!kDry) {
                            // symbol i of the match = symbol (i mod distance) of the `distance` symbols in front of it.  A distance is at most
                            // 32 768 = kSegPrefix: a source in front of the piece is a place-holder the caller has laid there
                            const uint32_t src0 = op - dist;  // (wraps where the source lies in front of the piece: the far path's case)
                            if (dist + len > kSegRing || dist > op) {
                                // the source has (partly) left the ring, or never was in it: symbols below `flushed` come from global memory
                                // (past the CU's L1: a line may have been cached before its last symbols were stored), the rest from the ring
                                for (uint32_t i = lane; i < len; i += 64) {
                                    const long long p = (long long)op - (long long)dist + (dist >= len ? i : i % dist);
                                    const uint16_t v = p < (long long)flushed ? __hip_atomic_load(out + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                                               : ring[(uint64_t)p & kSegRingMask];
                                    ring[(op + i) & kSegRingMask] = v;
                                }
                            } else if (dist >= len) {
                                for (uint32_t i = lane; i < len; i += 64) ring[(op + i) & kSegRingMask] = ring[(src0 + i) & kSegRingMask];
                            } else {
                                for (uint32_t i = lane; i < len; i += 64) ring[(op + i) & kSegRingMask] = ring[(src0 + i % dist) & kSegRingMask];
                            }
                        }
                        op += len;
                        if (op < flush_at) continue;
                    }
                }
                if (op > cap32) {
                    status = kSegOverflow;
                    break;
                }
                if (MKS_RAN_OUT()) {
                    status = kInfTruncated;
                    break;
                }
                if constexpr (kDry) {
                    if (max_symbols && op >= max_symbols) {
                        status = kSegCutShort;
                        break;
                    }
                }
                flush_blocks();
                flush_at = flushed + kSegFlush;
            }
            if (status == 0 && op > cap32) status = kSegOverflow;
            if (status == 0 && MKS_RAN_OUT()) status = kInfTruncated;
            if (status) break;
        }
        if (final_block) {
            fin = true;
            break;
        }
    }
    if constexpr (!kDry) {
        if (status == 0) {
            flush_blocks();
            for (uint32_t i = flushed + lane; i < op; i += 64) out[i] = ring[i & kSegRingMask];
        }
    }
    *n_out = op;
    *bit_stop = MKS_BITPOS();
    *final_seen = fin;
    return status;
#undef MKS_NEED32
#undef MKS_TAKE
#undef MKS_BITPOS
#undef MKS_RAN_OUT
}

// a plausible header at `bit` (the same on every lane): is it a block start?  The block is decoded dry by the whole wave, kSegConfirmSymbols
// symbols of it at most: the few headers that pass seg_header_plausible() without being one (3 in 5.7 G bit positions of gzip-written
// FASTQ) give codes whose first codewords are no symbols, or whose block ends after a handful of symbols in front of something that
// is no header.  A block that ends within the bound is asked what seg_confirm_block_start() of gzip_segments.hpp asks.  (The first
// version decoded every candidate's WHOLE block: a second decode of a quarter to all of the stream, 8-16 ms of the search.)
__device__ __forceinline__ bool wave_confirm_block_start(const uint8_t *__restrict__ in, uint64_t n_in, uint64_t bit, SegWaveTables &S) {
    uint64_t n_out = 0, stop = 0;
    bool fin = false;
    const int rc = wave_inflate_segment<true>(in, n_in, bit, ~0ull, 1, nullptr, ~0ull, nullptr, S, kSegConfirmSymbols, &n_out, &stop, &fin);
    if (rc == kSegCutShort) return true;
    if (rc != 0) return false;
    if (n_out == 0 || fin) return false;  // (the block at a nominal cut is neither empty nor the last one)
    if (stop + 3 > n_in * 8) return false;
    SegReader r;
    r.in = in, r.n_in = n_in;
    sr_seek(r, stop);
    const uint32_t type = ((uint32_t)r.bitbuf >> 1) & 3u;
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
    return false;  // (a fixed block behind it proves nothing, type 3 is none)
}

}  // namespace
}  // namespace mkz
