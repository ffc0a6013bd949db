// bgzf_inflate_wave.hip -- BGZF members inflated on the device, ONE WAVE per member, the member's recent text in LDS (r05).
//
// The lane-per-member decoder (bgzf_inflate.hip) hides the latency of its serial token loop only by the number of members in
// flight: a call of a few thousand members keeps one lane per wave busy and lasts as long as that lane's ~10 000 token turns, each
// of which waits for the L2 round trip of its match copy (15-23 ms per launch whatever it holds; 64 MB of zlib level-6 members:
// 3-4 GB/s, profiles/r05_codec_real_before.txt).  Here a wave owns a member:
//   * the last 32 KiB of the member's text -- all a DEFLATE distance can reach -- live in an LDS ring; a match is copied by the
//     whole wave, LDS to LDS, in one or two steps whatever its length (an overlapping match is its first `distance` bytes repeated:
//     lane i takes byte i mod distance); literals are one LDS byte store; the text leaves for global memory 4 KiB at a time,
//     coalesced 16-byte stores, and is never read back;
//   * every lane runs the same token loop on the same (wave-uniform) bit buffer: no divergence, no cross-lane hand-off -- the lane
//     id only matters where bytes are spread over lanes.  The compressed stream is read 256 bytes at a time, one dword per lane,
//     and handed to the bit buffer by v_readlane; the next 256 bytes are already on their way;
//   * codewords are decoded with ONE LDS read: a 10-bit (literal / length) and a 9-bit (distance) direct table, built per block by
//     all lanes from the canonical description (limits / bases / symbols by length), which also decodes the rare longer codewords;
//   * the tables are built in parallel -- counts per length by lane, symbol ranks by ballot -- so that a block header costs
//     microseconds; nothing of the decoder lives in private memory (0 bytes of scratch).
// With the whole 32 KiB in LDS a wave takes 36.1 KiB: 4 waves per CU, one per SIMD -- nothing fills the waits of a token turn's
// ~150 (literal) to ~500 (match) cycles of dependent LDS latency.  The kernel is a template on the ring's size: with the most
// recent 8 KiB (12.4 KiB per wave, 12 waves per CU, 3 072 members in flight on an MI355X) the same member takes as long but three
// times as many run beside it (4 KiB: 9.3 KiB per wave, 17 per CU, 4 352 in flight); matches that reach behind the ring come back
// from global memory (see WaveLds).  launch_inflate (bgzf_inflate.hip) takes the 4 KiB ring for calls of up to ~2 000 members, the 2 KiB ring
// (7.3 KiB per wave, 25 per CU) up to ~24 000; above that the lane-per-member kernel's sheer parallelism wins
// (profiles/r05_codec_real_rings2.txt).
// RFC 1951; the reference's inflate is flate2 under `bam 0.1.4` (src/cmd_tag.rs:503-506) and needletail (src/cmd_extract.rs:281).
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "inflate_wave_common.hpp"

namespace mkz {

namespace {

// kRing bytes of the member's most recent text.  32 KiB hold everything a DEFLATE distance can reach (4 waves per CU).  With a
// SHORTER ring (r05, second step) the waves per CU go up -- 8 KiB: 12, 16 KiB: 7 -- which is what hides a token turn's chain of
// dependent LDS accesses; a match that reaches behind the ring reads its source from global memory, where those bytes went with a
// flush (text that is still in flight, [flushed, op), is always in the ring: kRing >= 2 * kFlush).  In BAM / FASTQ text most
// matches point at the previous record or two, a few hundred bytes back.
template <uint32_t kRing>
struct WaveLds {
    uint8_t ring[kRing];
    // direct tables for codewords of <= 10 / 9 bits (else 0), everything a token turn needs in ONE read (r05: the kernel is bound by
    // its scalar instructions -- 107 per token, 1 M per member, one scalar unit per CU --, so what can be looked up is not computed):
    // ll_fast: a literal or end-of-block = symbol << 4 | codeword length (< 0x8000); a length symbol = 0x8000 | (base - 3) << 7 |
    // extra bits << 4 | codeword length.  d_fast: distance base << 8 | extra bits << 4 | codeword length.
    uint16_t ll_fast[1 << kFastLl];
    uint32_t d_fast[1 << kFastD];
    uint16_t ll_limit[16], ll_base[16], d_limit[16], d_base[16];
    uint16_t ll_sorted[288], d_sorted[32];
    uint8_t lens[320];  // code lengths of the block being set up: literal / length [0, 288), distance [288, 320)
};
static_assert(sizeof(WaveLds<32768>) <= 40 * 1024, "four waves per CU");
static_assert(sizeof(WaveLds<8192>) <= 14 * 1024, "eleven waves per CU");
static_assert(sizeof(WaveLds<4096>) <= 9728, "seventeen waves per CU");

}  // namespace

template <uint32_t kRing>
__global__ __launch_bounds__(64) void mk_bgzf_inflate_wave_kernel(const uint8_t *__restrict__ in_all, uint64_t n_in_all, const Member *__restrict__ members,
                                                                  uint32_t n_members, uint8_t *__restrict__ out_all, int32_t *__restrict__ status_out) {
    constexpr uint32_t kFlush = kRing >= 8192 ? 4096u : kRing / 2;  // (the text leaves in pieces of this size; the ring holds two of them)
    static_assert(kRing >= 2 * kFlush && kFlush >= 1024 && (kRing & (kRing - 1)) == 0, "ring: a power of two that holds two flush pieces");
    constexpr uint32_t kRingMask = kRing - 1;
    extern __shared__ uint8_t lds_raw[];
    WaveLds<kRing> &S = *reinterpret_cast<WaveLds<kRing> *>(lds_raw);
    const uint32_t lane = lane_id();
    const uint32_t mi = blockIdx.x;
    if (mi >= n_members) return;
    const Member m = members[mi];
    const uint8_t *in = in_all + m.data_off;
    const uint32_t n_in = m.data_len, n_out = m.isize;
    uint8_t *const out = out_all + m.out_off;

    // ---- the compressed stream: lane k holds dword (sbase + k) of the stream that starts at byte s0; `nxt` the 64 dwords behind
    uint32_t s0 = 0, sbase = 0, rd = 0;  // rd = dwords handed to the bit buffer (relative to s0)
    uint32_t look = 0, nxt = 0;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    auto load_dword = [&](uint32_t dw) -> uint32_t {  // dword dw of the stream at s0; zeros behind n_in + 8 (a valid stream never takes them)
        const uint32_t byte = s0 + 4 * dw;
        return byte <= n_in + 8 ? load_le32(in + byte) : 0u;
    };
    auto restart = [&](uint32_t byte_pos) {
        s0 = byte_pos, sbase = 0, rd = 0, bitbuf = 0, bitcnt = 0;
        look = load_dword(lane);
        nxt = load_dword(64 + lane);
    };
#define MKW_NEED32()                                                                              \
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
#define MKW_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
#define MKW_TAKEN() (s0 + 4 * rd)  // stream bytes handed to the bit buffer
#define MKW_RAN_OUT() (MKW_TAKEN() > n_in + 8)
    restart(0);

    uint32_t op = 0, flushed = 0;  // bytes produced / bytes that have left for global memory
    int status = 0;
    auto flush_blocks = [&]() {  // whole 4 KiB pieces: 16 bytes per lane and step, LDS reads and global stores both in a row
        while (op - flushed >= kFlush) {
#pragma unroll
            for (uint32_t k = 0; k < kFlush; k += 1024) {
                const uint32_t at = flushed + k + 16 * lane;
                const uint4 v = *reinterpret_cast<const uint4 *>(&S.ring[at & kRingMask]);
                __builtin_memcpy(out + at, &v, 16);
            }
            flushed += kFlush;
        }
        // (a shorter ring reads flushed text back: the stores above must have arrived before such a load is issued)
        // (this wave alone reads them, past the CU's L1: they need to have reached the L2, not to be written back from it -- an agent-scope
        // release is a write-back of the whole L2, `buffer_wbl2`, per flush)
        if (kRing < 32768) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
    };

    for (bool last_block = false; !last_block && status == 0;) {
        MKW_NEED32();
        if (MKW_RAN_OUT()) {
            status = kInfTruncated;
            break;
        }
        last_block = ((uint32_t)bitbuf & 1u) != 0;
        const uint32_t type = ((uint32_t)bitbuf >> 1) & 3u;
        MKW_TAKE(3);
        if (type == 3) {
            status = kInfBadBlockType;
            break;
        }
        if (type == 0) {  // stored: skip to the byte boundary, LEN, ~LEN, bytes -- copied stream -> ring by all lanes
            MKW_TAKE(bitcnt & 7);
            MKW_NEED32();
            const uint32_t len = (uint32_t)bitbuf & 0xffffu, nlen = ((uint32_t)(bitbuf >> 16)) & 0xffffu;
            MKW_TAKE(32);
            const uint32_t pos = MKW_TAKEN() - (bitcnt >> 3);
            if ((len ^ nlen) != 0xffffu) status = kInfBadStored;
            else if (MKW_RAN_OUT() || pos > n_in || len > n_in - pos) status = kInfTruncated;
            else if (len > n_out - op) status = kInfOutputOverrun;
            if (status) break;
            for (uint32_t done = 0; done < len;) {
                const uint32_t piece = min(len - done, kFlush);
                for (uint32_t i = lane; i < piece; i += 64) S.ring[(op + i) & kRingMask] = in[pos + done + i];
                op += piece, done += piece;
                flush_blocks();
            }
            restart(pos + len);
            continue;
        }
        // ---- code lengths of the block -> S.lens
        if (type == 1) {
            for (uint32_t i = lane; i < 320; i += 64) S.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
        } else {
            const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = ((uint32_t)(bitbuf >> 5) & 31u) + 1, hclen = ((uint32_t)(bitbuf >> 10) & 15u) + 4;
            MKW_TAKE(14);
            if (hlit > 286 || hdist > 30) {
                status = kInfBadLengths;
                break;
            }
            // the code-length code: 19 lengths of 3 bits in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (5 bits each, packed)
            if (lane < (uint32_t)kCl) S.lens[lane] = 0;
            constexpr uint64_t kOrderLo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                                          5ull << 45 | 11ull << 50 | 4ull << 55;
            constexpr uint64_t kOrderHi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
            for (uint32_t i = 0; i < hclen; ++i) {
                MKW_NEED32();
                const uint32_t sym = (uint32_t)((i < 12 ? kOrderLo >> (5 * i) : kOrderHi >> (5 * (i - 12))) & 31u);
                if (lane == 0) S.lens[sym] = (uint8_t)((uint32_t)bitbuf & 7u);
                MKW_TAKE(3);
            }
            // (the code-length code borrows the literal tables)
            if (wave_build_tables(S.lens, kCl, S.ll_sorted, S.ll_limit, S.ll_base, false)) {
                status = kInfBadLengths;
                break;
            }
            uint32_t i = 0, prev_len = 0;
            while (i < hlit + hdist && status == 0) {
                MKW_NEED32();
                if (MKW_RAN_OUT()) {
                    status = kInfTruncated;
                    break;
                }
                const uint32_t e = decode_codeword((uint32_t)bitbuf, S.ll_sorted, S.ll_limit, S.ll_base);
                if (e == 0 || (e & 15u) > (uint32_t)kMaxClBits) {
                    status = kInfBadLengths;
                    break;
                }
                MKW_TAKE(e & 15u);
                const uint32_t sym = e >> 4;
                if (sym < 16) {
                    // (lengths of the distance alphabet are stored where they belong at once: [288, 288 + hdist))
                    if (lane == 0) S.lens[i < hlit ? i : 288 + (i - hlit)] = (uint8_t)sym;
                    prev_len = sym, ++i;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) {
                        status = kInfBadLengths;
                        break;
                    }
                    val = prev_len;
                    rep = 3 + ((uint32_t)bitbuf & 3u);
                    MKW_TAKE(2);
                } else if (sym == 17) {
                    rep = 3 + ((uint32_t)bitbuf & 7u);
                    MKW_TAKE(3);
                } else {
                    rep = 11 + ((uint32_t)bitbuf & 127u);
                    MKW_TAKE(7);
                }
                if (i + rep > hlit + hdist) {
                    status = kInfBadLengths;
                    break;
                }
                if (lane < rep) {
                    const uint32_t j = i + lane;  // (rep <= 138: three rounds of lanes at most)
                    S.lens[j < hlit ? j : 288 + (j - hlit)] = (uint8_t)val;
                }
                if (lane + 64 < rep) {
                    const uint32_t j = i + lane + 64;
                    S.lens[j < hlit ? j : 288 + (j - hlit)] = (uint8_t)val;
                }
                if (lane + 128 < rep) {
                    const uint32_t j = i + lane + 128;
                    S.lens[j < hlit ? j : 288 + (j - hlit)] = (uint8_t)val;
                }
                prev_len = val, i += rep;
            }
            if (status) break;
            // each alphabet padded with zeros to its full size
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
        wave_fill_fast_ll(S.ll_fast, S.ll_sorted, S.ll_limit, S.ll_base);
        wave_fill_fast_d(S.d_fast, S.d_sorted, S.d_limit, S.d_base);
        __builtin_amdgcn_wave_barrier();

        // ---- the symbols of the block: one token per turn, the same turn on every lane.  The turn is kept short (the scalar unit is
        // the bound): bases and extra bits come out of the tables, and what need not be known per token -- has the stream run out,
        // has the text outgrown ISIZE -- is asked where the text is about to leave for global memory and at the block's end (until
        // then the ring's index mask keeps every write in place, and bytes are only flushed below ISIZE).
        uint32_t flush_at = flushed + kFlush;
        for (;;) {
            MKW_NEED32();
            uint32_t e = S.ll_fast[(uint32_t)bitbuf & ((1u << kFastLl) - 1)];
            if (e == 0) {  // a codeword longer than the table's reach (or none)
                e = pack_ll(decode_codeword((uint32_t)bitbuf, S.ll_sorted, S.ll_limit, S.ll_base));
                if (e == 0) {
                    status = kInfBadSymbol;
                    break;
                }
            }
            MKW_TAKE(e & 15u);
            if (e < 0x1000u) {  // a literal (every lane stores the same byte to the same place)
                S.ring[op & kRingMask] = (uint8_t)(e >> 4);
                ++op;
                if (op < flush_at) continue;
            } else if (e < 0x8000u) {  // 256: the end of the block
                break;
            } else {
                const uint32_t leb = (e >> 4) & 7u;
                const uint32_t len = ((e >> 7) & 255u) + 3u + ((uint32_t)bitbuf & ((1u << leb) - 1));
                MKW_TAKE(leb);
                MKW_NEED32();
                uint32_t d = S.d_fast[(uint32_t)bitbuf & ((1u << kFastD) - 1)];
                if (d == 0) {
                    d = pack_d(decode_codeword((uint32_t)bitbuf, S.d_sorted, S.d_limit, S.d_base));
                    if (d == 0) {
                        status = kInfBadSymbol;
                        break;
                    }
                }
                MKW_TAKE(d & 15u);
                const uint32_t deb = (d >> 4) & 15u;
                const uint32_t dist = (d >> 8) + ((uint32_t)bitbuf & ((1u << deb) - 1));
                MKW_TAKE(deb);
                if (dist > op) {
                    status = kInfBadDistance;
                    break;
                }
                // byte i of the match = byte (i mod distance) of the `distance` bytes in front of it: every source byte exists already
                const uint32_t src0 = op - dist;
                if (kRing < 32768 && dist + len > kRing) {
                    // the source has (partly) left the ring: bytes below `flushed` come back from global memory (past the CU's L1:
                    // a line may have been cached before its last bytes were stored), the rest is still in the ring
                    for (uint32_t i = lane; i < len; i += 64) {
                        const uint32_t p = src0 + i;  // (dist > kRing - 258 > len: no wrap inside the match)
                        const uint8_t b = p < flushed ? __hip_atomic_load(out + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S.ring[p & kRingMask];
                        S.ring[(op + i) & kRingMask] = b;
                    }
                } else if (dist >= len) {
                    for (uint32_t i = lane; i < len; i += 64) S.ring[(op + i) & kRingMask] = S.ring[(src0 + i) & kRingMask];
                } else {
                    for (uint32_t i = lane; i < len; i += 64) S.ring[(op + i) & kRingMask] = S.ring[(src0 + i % dist) & kRingMask];
                }
                op += len;
                if (op < flush_at) continue;
            }
            // a flush point: the questions that were put off
            if (op > n_out) {
                status = kInfOutputOverrun;
                break;
            }
            if (MKW_RAN_OUT()) {
                status = kInfTruncated;
                break;
            }
            flush_blocks();
            flush_at = flushed + kFlush;
        }
        if (status == 0 && op > n_out) status = kInfOutputOverrun;
        if (status == 0 && MKW_RAN_OUT()) status = kInfTruncated;
    }
    // bits consumed must lie inside the stream; the text must be exactly ISIZE bytes
    if (status == 0 && (MKW_RAN_OUT() || (uint64_t)MKW_TAKEN() - (bitcnt >> 3) > (uint64_t)n_in)) status = kInfTruncated;
    if (status == 0 && op != n_out) status = kInfOutputShort;
    if (status == 0) {
        flush_blocks();
        for (uint32_t i = flushed + lane; i < op; i += 64) out[i] = S.ring[i & kRingMask];
    }
    if (lane == 0) status_out[mi] = status;
#undef MKW_NEED32
#undef MKW_TAKE
#undef MKW_TAKEN
#undef MKW_RAN_OUT
}

template <uint32_t kRing>
static void launch_ring(const uint8_t *in, uint64_t n_in, const Member *members, uint32_t n_members, uint8_t *out, int32_t *status, hipStream_t s) {
    static bool raised = false;  // (up to 36 KiB of dynamic LDS)
    if (!raised) {
        (void)hipFuncSetAttribute((const void *)mk_bgzf_inflate_wave_kernel<kRing>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WaveLds<kRing>));
        raised = true;
    }
    hipLaunchKernelGGL(mk_bgzf_inflate_wave_kernel<kRing>, dim3(n_members), dim3(64), sizeof(WaveLds<kRing>), s, in, n_in, members, n_members, out, status);
}

void launch_inflate_wave(const uint8_t *in, uint64_t n_in, const Member *members, uint32_t n_members, uint8_t *out, int32_t *status, hipStream_t s,
                         uint32_t ring_bytes) {
    if (!n_members) return;
    if (ring_bytes == 2048) launch_ring<2048>(in, n_in, members, n_members, out, status, s);
    else if (ring_bytes == 4096) launch_ring<4096>(in, n_in, members, n_members, out, status, s);
    else if (ring_bytes == 8192) launch_ring<8192>(in, n_in, members, n_members, out, status, s);
    else if (ring_bytes == 16384) launch_ring<16384>(in, n_in, members, n_members, out, status, s);
    else launch_ring<32768>(in, n_in, members, n_members, out, status, s);
}

}  // namespace mkz
