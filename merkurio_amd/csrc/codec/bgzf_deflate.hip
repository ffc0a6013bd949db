// bgzf_deflate.hip -- BGZF members written on the device: CRC-32, LZ77 parse, Huffman codes, bit packing.
//
// Replaces, behind mk_bgzf_deflate (include/merkurio_hip.h), the deflate the reference gets from flate2 inside
// `bam 0.1.4`'s writer threads (src/cmd_tag.rs:254-271,577-615): 2.6 of the 3.9-4.1 s of `merkurio tag` with BAM
// output were zlib on 16 host threads (profiles/r04_e2e_tag.txt).  The members inflate to the same bytes; their
// compressed form is this file's own parse (any RFC 1951 stream is a valid BGZF payload: the reference's readers,
// htslib and zlib accept it; tests inflate every member with zlib).
//
// One WAVE per member (<= 65280 bytes of input), one wave per workgroup, a resident grid (16 waves per CU) whose waves
// take the next member from a counter when they are done with one:
//   1. parse, 64 positions (one per lane) at a time: 4-byte hash -> most recent earlier position (u16 table in LDS),
//      match length by 8-byte compares, a distance-1 candidate for runs; a scalar walk over the window's lanes
//      (ballot + readlane, one step per MATCH, literals in between are taken in one go; one-step lazy evaluation)
//      picks the tokens; the picked lanes write them, compacted, to the wave's token scratch and count their symbols
//      in LDS.  A window a match has jumped over is skipped altogether.
//   2. the two Huffman codes from the counts: keys rank-sorted by the wave, then ONE lane runs the serial builder
//      and the header planner of deflate_common.hpp (what the host harness checks against zlib).
//   3. the size is known before a bit is written (counts x code lengths): a member that would not shrink is stored.
//   4. bit packing, 64 tokens at a time: wave scan of the code sizes, lanes OR their bits into an LDS staging
//      area, whole words go out coalesced; the member's header and BSIZE lead the same word stream.
// Bound: none of HBM / MFMA -- latency of dependent LDS and L2 accesses per window (59 % of the wave cycles are waits,
// profiles/r04_codec_kernels.txt); what matters is resident waves (hence a 1 Ki-entry table: 9.8 KiB of LDS per wave,
// see kHashBits) and skipping work (covered windows, literal runs).
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "deflate_common.hpp"

namespace mkz {

// The hash table's size is an occupancy decision, not a compression one: the kernel waits ~60 % of its cycles (dependent
// L2 / LDS accesses per window), and only other resident waves can fill them.  On BAM-shaped records (2.1 GB, one MI355X;
// profiles/r04_codec_hash_bits.txt): 14 bits = 3 waves per CU 14.5 GB/s, 13 = 6: 22, 12 = 9: 29.5, 11 = 12: 33,
// 10 = 14: 38.5, 9 = 16: 42.5 GB/s -- with the ratio unchanged down to 10 bits (3.03; FASTQ text 2.97 -> 2.90, zlib level 1:
// 2.88) because the table keeps the MOST RECENT position of a hash, which is where BAM / FASTQ repeats come from; at 9
// bits FASTQ text falls below zlib level 1 (2.86).
constexpr int kHashBits = 10;
constexpr uint32_t kMinLen = 4;
constexpr uint32_t kTokMatch = 0x80000000u;  // token: literal / end-of-block symbol, or kTokMatch | (len-3) << 16 | (dist-1)

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }
__device__ __forceinline__ uint32_t below(uint64_t mask) {  // set bits of mask in lanes below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ uint64_t bits_range(uint32_t a, uint32_t b) {  // bits a .. b-1, a <= b <= 64
    const uint64_t hi = b >= 64 ? ~0ull : (1ull << b) - 1, lo = a >= 64 ? ~0ull : (1ull << a) - 1;
    return hi & ~lo;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= (uint32_t)o) v += u;
    }
    return v;
}

// bytes a[0..) and b[0..) have in common, at most maxlen (reads up to 7 bytes past a + maxlen)
__device__ __forceinline__ uint32_t match_len(const uint8_t *a, const uint8_t *b, uint32_t maxlen) {
    uint32_t l = 0;
    while (l < maxlen) {
        const uint64_t x = ld64(a + l) ^ ld64(b + l);
        if (x) {
            l += (uint32_t)__builtin_ctzll(x) >> 3;
            break;
        }
        l += 8;
    }
    return l < maxlen ? l : maxlen;
}

struct DeflateLds {
    uint16_t head[1 << kHashBits];
    uint32_t freq_ll[288], freq_d[32];
    uint32_t skey_ll[288], skey_d[32];  // (count << 9 | symbol) of the used symbols in ascending order
    BlockCodes codes;
    union {  // the unsorted keys are dead once they are ranked, before the code builder's scratch comes to life
        struct {
            uint32_t key_ll[288], key_d[32];
        } k;
        HeaderScratch hdr;
    };
    uint32_t words[192];  // the dynamic header's bits, then the staging area of 64 tokens
};
static_assert(sizeof(DeflateLds) <= 10240, "16 waves per CU (160 KiB of LDS)");

// keys of the used symbols of freq[0, n), compacted by the wave; returns their number (>= 2, see symbol_keys)
__device__ uint32_t wave_symbol_keys(const uint32_t *freq, uint32_t n, uint32_t *key, uint32_t lane) {
    uint32_t m = 0;
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t f = s < n ? freq[s] : 0;
        const uint64_t used = __ballot(f != 0);
        if (f) key[m + below(used)] = f << 9 | s;
        m += (uint32_t)__popcll(used);
    }
    __syncthreads();
    if (m < 2) {
        if (lane == 0) {
            for (uint32_t s = 0; m < 2; ++s) {
                bool used = false;
                for (uint32_t k = 0; k < m; ++k) used |= (key[k] & 511u) == s;
                if (!used) key[m++] = 1u << 9 | s;
            }
        }
        m = 2;
        __syncthreads();
    }
    return m;
}
// keys are distinct (the symbol is part of them): the rank of a key is the number of smaller keys
__device__ void wave_rank_sort(const uint32_t *key, uint32_t m, uint32_t *sorted, uint32_t lane) {
    for (uint32_t k = lane; k < m; k += 64) {
        const uint32_t mine = key[k];
        uint32_t r = 0;
        for (uint32_t j = 0; j < m; ++j) r += key[j] < mine;
        sorted[r] = mine;
    }
    __syncthreads();
}

__device__ __forceinline__ void token_code(uint32_t t, const BlockCodes &c, uint64_t &bits, uint32_t &nb) {
    if (!(t & kTokMatch)) {
        bits = c.ll_code[t], nb = c.ll_len[t];
        return;
    }
    uint32_t li, lnb, lxb, ds, dnb, dxb;
    length_symbol(((t >> 16) & 255u) + 3, li, lnb, lxb);
    distance_symbol((t & 0xffffu) + 1, ds, dnb, dxb);
    bits = c.ll_code[257 + li], nb = c.ll_len[257 + li];
    bits |= (uint64_t)lxb << nb, nb += lnb;
    bits |= (uint64_t)c.d_code[ds] << nb, nb += c.d_len[ds];
    bits |= (uint64_t)dxb << nb, nb += dnb;
}

__global__ __launch_bounds__(64, 4) void mk_bgzf_deflate_kernel(const uint8_t *__restrict__ in, uint64_t n_bytes, uint32_t block_bytes,
                                                             uint32_t n_blocks, const uint32_t *__restrict__ crc,
                                                             uint32_t *__restrict__ tokens, uint8_t *__restrict__ slots,
                                                             uint32_t *__restrict__ slot_len, uint32_t *__restrict__ next_block) {
    __shared__ DeflateLds L;
    const uint32_t lane = lane_id();
    uint32_t *const tok = tokens + (uint64_t)blockIdx.x * kTokensPerWave;

    // members are handed out one at a time (they take unequal time: a wave that is done asks for the next one)
    for (;;) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(next_block, 1u);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        if (b >= n_blocks) break;
        const uint64_t at = (uint64_t)b * block_bytes;
        const uint8_t *const src = in + at;
        const uint32_t n = (uint32_t)(n_bytes - at < block_bytes ? n_bytes - at : block_bytes);
        uint8_t *const slot = slots + (uint64_t)b * kSlotBytes;
        uint32_t *const slot_w = reinterpret_cast<uint32_t *>(slot);

        for (uint32_t k = lane; k < (1u << kHashBits) / 2; k += 64) reinterpret_cast<uint32_t *>(L.head)[k] = 0xffffffffu;
        for (uint32_t k = lane; k < 288; k += 64) L.freq_ll[k] = 0;
        if (lane < 32) L.freq_d[lane] = 0;
        __syncthreads();

        // ---- 1. parse -----------------------------------------------------------------------------------
        uint32_t n_tok = 0;  // tokens written
        uint32_t p = 0;      // the next position the parse takes
        for (uint32_t base = 0; base < n; base += 64) {
            if (p >= base + 64) continue;  // inside a match
            const uint32_t i = base + lane;
            const bool can = i + kMinLen <= n;
            const uint32_t w = i < n ? ld32(src + i) : 0;  // (bytes behind n: the next block or the padding, never part of a match)
            uint32_t cand = 0xffffu, h = 0;
            if (can) {
                h = (w * 0x9E3779B1u) >> (32 - kHashBits);
                cand = L.head[h];
            }
            if (can) L.head[h] = (uint16_t)i;  // lanes with one hash: one of them stays, all are earlier positions for what follows
            uint32_t len = 0, dist = 0;
            if (can && i >= p) {
                const uint32_t maxlen = n - i < (uint32_t)kMaxMatch ? n - i : (uint32_t)kMaxMatch;
                if (cand != 0xffffu && i - cand <= (uint32_t)kWindow) {
                    const uint32_t l = match_len(src + i, src + cand, maxlen);
                    if (l >= kMinLen) len = l, dist = i - cand;
                }
                if (i > 0 && len < maxlen && ld32(src + i - 1) == w) {  // a run: the nearest candidate the table cannot hold yet
                    const uint32_t l = match_len(src + i, src + i - 1, maxlen);
                    if (l >= len) len = l, dist = 1;
                }
            }
            // the walk: wave-uniform, one step per match
            const uint32_t lim = n - base < 64 ? n - base : 64;
            const uint64_t has = __ballot(len >= kMinLen);
            uint64_t starts = 0, mstarts = 0;
            uint32_t q = p - base;
            while (q < lim) {
                const uint64_t rest = has & bits_range(q, 64);
                if (!rest) {
                    starts |= bits_range(q, lim);
                    q = lim;
                    break;
                }
                const uint32_t m = (uint32_t)__builtin_ctzll(rest);
                starts |= bits_range(q, m + 1);
                const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
                if (m + 1 < lim && ((has >> (m + 1)) & 1) && (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(m + 1)) > l0) {
                    q = m + 1;  // the next position starts a longer match: this one goes out as a literal
                    continue;
                }
                mstarts |= 1ull << m;
                q = m + l0;
            }
            p = base + q;
            if ((starts >> lane) & 1) {
                const uint32_t idx = n_tok + below(starts);
                if ((mstarts >> lane) & 1) {
                    tok[idx] = kTokMatch | (len - 3) << 16 | (dist - 1);
                    uint32_t li, ds, nb, xb;
                    length_symbol(len, li, nb, xb);
                    distance_symbol(dist, ds, nb, xb);
                    atomicAdd(&L.freq_ll[257 + li], 1u);
                    atomicAdd(&L.freq_d[ds], 1u);
                } else {
                    tok[idx] = w & 255u;
                    atomicAdd(&L.freq_ll[w & 255u], 1u);
                }
            }
            n_tok += (uint32_t)__popcll(starts);
        }
        if (lane == 0) {
            tok[n_tok] = 256;  // end of block
            L.freq_ll[256] = 1;
        }
        n_tok += 1;
        __syncthreads();

        // ---- 2. the codes -------------------------------------------------------------------------------
        const uint32_t m_ll = wave_symbol_keys(L.freq_ll, kLitLen, L.k.key_ll, lane);
        const uint32_t m_d = wave_symbol_keys(L.freq_d, kDist, L.k.key_d, lane);
        wave_rank_sort(L.k.key_ll, m_ll, L.skey_ll, lane);
        wave_rank_sort(L.k.key_d, m_d, L.skey_d, lane);
        if (lane == 0) {
            block_codes_from_sorted(L.skey_ll, (int)m_ll, L.skey_d, (int)m_d, L.codes, L.hdr.huff);
            L.words[191] = plan_dynamic_header(L.codes.ll_len, L.codes.d_len, L.hdr);
        }
        __syncthreads();
        const uint32_t hdr_bits = L.words[191];

        // ---- 3. size --------------------------------------------------------------------------------------
        uint32_t body = 0;
        for (uint32_t s = lane; s < (uint32_t)kLitLen; s += 64)
            body += L.freq_ll[s] * (L.codes.ll_len[s] + (s > 256 ? length_extra_bits(s - 257) : 0));
        if (lane < (uint32_t)kDist) body += L.freq_d[lane] * (L.codes.d_len[lane] + distance_extra_bits(lane));
        const uint32_t stream_bits = hdr_bits + wave_sum(body);
        const uint32_t stream_bytes = (stream_bits + 7) >> 3;
        const bool stored = stream_bytes >= n + 5;
        const uint32_t payload = stored ? n + 5 : stream_bytes;
        const uint32_t member = 18 + payload + 8;
        __syncthreads();

        // gzip header with the BC subfield (bytes 0-15), BSIZE = member size - 1 (bytes 16-17)
        if (lane == 0) slot_w[0] = 0x04088b1fu, slot_w[1] = 0, slot_w[2] = 0x0006ff00u, slot_w[3] = 0x00024342u;
        if (stored) {
            if (lane == 0) {
                slot_w[4] = (member - 1) | 0x01u << 16 | (n & 255u) << 24;  // BSIZE, BFINAL = 1 / BTYPE = 00, LEN
                slot[20] = (uint8_t)(n >> 8), slot[21] = (uint8_t)(~n & 255u), slot[22] = (uint8_t)((~n >> 8) & 255u);
            }
            for (uint32_t k = lane; k < n; k += 64) slot[23 + k] = src[k];
        } else {
            // ---- 4. bits: the word stream starts at word 4 = BSIZE in its low half, DEFLATE from bit 16 -----------
            for (uint32_t k = lane; k < 192; k += 64) L.words[k] = 0;
            __syncthreads();
            if (lane == 0) {
                L.words[0] = member - 1;
                BitSink bs{L.words, 16};
                write_dynamic_header(bs, L.hdr, true);
            }
            __syncthreads();
            uint32_t bitpos = 16 + hdr_bits;  // bits of the word stream so far
            uint32_t w0 = 4;                  // slot word that staging word 0 stands for
            {
                const uint32_t cw = bitpos >> 5;
                for (uint32_t k = lane; k < cw; k += 64) slot_w[w0 + k] = L.words[k];
                const uint32_t carry = L.words[cw];
                w0 += cw;
                __syncthreads();
                for (uint32_t k = lane; k < 192; k += 64) L.words[k] = 0;
                __syncthreads();
                if (lane == 0) L.words[0] = carry;
                __syncthreads();
            }
            for (uint32_t t0 = 0; t0 < n_tok; t0 += 64) {
                uint64_t bits = 0;
                uint32_t nb = 0;
                if (t0 + lane < n_tok) token_code(tok[t0 + lane], L.codes, bits, nb);
                const uint32_t incl = wave_incl_scan(nb, lane);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (nb) {
                    const uint32_t off = (bitpos & 31u) + incl - nb, wi = off >> 5, sh = off & 31u;
                    const uint64_t lo = bits << sh;
                    atomicOr(&L.words[wi], (uint32_t)lo);
                    if ((uint32_t)(lo >> 32)) atomicOr(&L.words[wi + 1], (uint32_t)(lo >> 32));
                    const uint32_t top = sh ? (uint32_t)(bits >> (64 - sh)) : 0;
                    if (top) atomicOr(&L.words[wi + 2], top);
                }
                __syncthreads();
                const uint32_t cw = ((bitpos & 31u) + total) >> 5;  // <= 97
                for (uint32_t k = lane; k < cw; k += 64) slot_w[w0 + k] = L.words[k];
                const uint32_t carry = L.words[cw];
                w0 += cw, bitpos += total;
                __syncthreads();
                L.words[lane] = 0, L.words[lane + 64] = 0;
                __syncthreads();
                if (lane == 0) L.words[0] = carry;
                __syncthreads();
            }
            // the bytes of the last, partial word
            const uint32_t used = ((bitpos & 31u) + 7) >> 3;
            if (lane < used) slot[(uint64_t)w0 * 4 + lane] = (uint8_t)(L.words[0] >> (8 * lane));
            __syncthreads();
        }
        if (lane < 8) {
            const uint32_t v = lane < 4 ? crc[b] : n;
            slot[18 + payload + lane] = (uint8_t)(v >> (8 * (lane & 3)));
        }
        if (lane == 0) slot_len[b] = member;
        __syncthreads();
    }
}

// ---- CRC-32 of each block: a piece per lane (slice-by-4), folded by the wave ------------------------------------
__device__ __forceinline__ void build_crc_tables(uint32_t (*t)[256]) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) t[0][i] = crc_table_entry(i);
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) t[k][i] = t[0][t[k - 1][i] & 255u] ^ (t[k - 1][i] >> 8);
        __syncthreads();
    }
}
// the CRC-32 of d[0, n), computed by one wave: a contiguous piece per lane, read 16 bytes at a time (a lane's loads walk
// its own piece, so a wave-wide load touches 64 lines: with dword loads every line came up from L2 sixteen times)
__device__ uint32_t wave_crc32(const uint8_t *d, uint32_t n, const uint32_t (*t)[256], uint32_t lane) {
    const uint32_t piece = (((n + 63) >> 6) + 15) & ~15u;
    const uint32_t b0 = lane * piece < n ? lane * piece : n, e0 = b0 + piece < n ? b0 + piece : n;
    uint32_t r = lane == 0 ? 0xffffffffu : 0u;
    uint32_t i = b0;
    for (; i + 16 <= e0; i += 16) {
        uint32_t w[4];
        __builtin_memcpy(w, d + i, 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r ^= w[k];
            r = t[3][r & 255u] ^ t[2][(r >> 8) & 255u] ^ t[1][(r >> 16) & 255u] ^ t[0][r >> 24];
        }
    }
    for (; i + 4 <= e0; i += 4) {
        r ^= ld32(d + i);
        r = t[3][r & 255u] ^ t[2][(r >> 8) & 255u] ^ t[1][(r >> 16) & 255u] ^ t[0][r >> 24];
    }
    for (; i < e0; ++i) r = t[0][(r ^ d[i]) & 255u] ^ (r >> 8);
    uint32_t term = crc_mulmod(r, crc_x_pow_bytes(n - e0));  // the register clocked through the bytes behind this piece
    for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o);
    return ~term;
}

__global__ __launch_bounds__(256) void mk_bgzf_crc_kernel(const uint8_t *__restrict__ in, uint64_t n_bytes, uint32_t block_bytes,
                                                          uint32_t n_blocks, uint32_t *__restrict__ crc) {
    __shared__ uint32_t t[4][256];
    build_crc_tables(t);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const uint64_t at = (uint64_t)b * block_bytes;
    const uint32_t n = (uint32_t)(n_bytes - at < block_bytes ? n_bytes - at : block_bytes);
    const uint32_t c = wave_crc32(in + at, n, t, lane);
    if (lane == 0) crc[b] = c;
}

__global__ __launch_bounds__(256) void mk_bgzf_crc_check_kernel(const uint8_t *__restrict__ out, const Member *__restrict__ members,
                                                                uint32_t n_members, int32_t *__restrict__ status) {
    __shared__ uint32_t t[4][256];
    build_crc_tables(t);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_members) return;
    const Member m = members[b];
    const uint32_t c = m.isize ? wave_crc32(out + m.out_off, m.isize, t, lane) : 0u;
    if (lane == 0 && c != m.crc) status[b] |= 0x100;
}

// ---- members back to back ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mk_bgzf_offsets_kernel(const uint32_t *__restrict__ len, uint64_t *__restrict__ off,
                                                               uint64_t *__restrict__ total, uint32_t n) {
    __shared__ uint64_t part[1024];
    const uint32_t per = (n + 1023) / 1024, b0 = threadIdx.x * per, e0 = b0 + per < n ? b0 + per : n;
    uint64_t s = 0;
    for (uint32_t i = b0; i < e0; ++i) s += len[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; ++k) {
            const uint64_t v = part[k];
            part[k] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = b0; i < e0; ++i) {
        off[i] = run;
        run += len[i];
    }
}

__global__ __launch_bounds__(256) void mk_bgzf_pack_kernel(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ len,
                                                           const uint64_t *__restrict__ off, uint8_t *__restrict__ packed) {
    const uint32_t b = blockIdx.x, n = len[b];
    const uint8_t *s = slots + (uint64_t)b * kSlotBytes;
    uint8_t *d = packed + off[b];
    // whole dwords where source and destination allow it: the destination's alignment decides
    const uint32_t head = (uint32_t)((4 - ((uintptr_t)d & 3)) & 3);
    const uint32_t h = head < n ? head : n;
    if (threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
    const uint32_t words = (n - h) >> 2;
    for (uint32_t k = threadIdx.x; k < words; k += 256) reinterpret_cast<uint32_t *>(d + h)[k] = ld32(s + h + 4 * k);
    const uint32_t done = h + 4 * words;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
}

// ---- launchers ----------------------------------------------------------------------------------------------------
uint32_t deflate_grid(uint32_t n_blocks, int num_cus) {
    const uint32_t resident = (uint32_t)num_cus * 16;  // 10 KiB of LDS and < 128 VGPRs per wave: 4 waves per SIMD
    return n_blocks < resident ? n_blocks : resident;
}

void launch_crc(const uint8_t *in, uint64_t n, uint32_t block_bytes, uint32_t n_blocks, uint32_t *crc, hipStream_t s) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(mk_bgzf_crc_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, s, in, n, block_bytes, n_blocks, crc);
}
void launch_crc_check(const uint8_t *out, const Member *members, uint32_t n_members, int32_t *status, hipStream_t s) {
    if (!n_members) return;
    hipLaunchKernelGGL(mk_bgzf_crc_check_kernel, dim3((n_members + 3) / 4), dim3(256), 0, s, out, members, n_members, status);
}
void launch_deflate(const uint8_t *in, uint64_t n, uint32_t block_bytes, uint32_t n_blocks, const uint32_t *crc, uint32_t *tokens,
                    uint8_t *slots, uint32_t *slot_len, uint32_t *next_block, uint32_t grid, hipStream_t s) {
    if (!n_blocks) return;
    (void)hipMemsetAsync(next_block, 0, 4, s);
    hipLaunchKernelGGL(mk_bgzf_deflate_kernel, dim3(grid), dim3(64), 0, s, in, n, block_bytes, n_blocks, crc, tokens, slots, slot_len, next_block);
}
void launch_pack(const uint8_t *slots, const uint32_t *slot_len, uint64_t *slot_off, uint64_t *total, uint32_t n_blocks, uint8_t *packed,
                 hipStream_t s) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(mk_bgzf_offsets_kernel, dim3(1), dim3(1024), 0, s, slot_len, slot_off, total, n_blocks);
    hipLaunchKernelGGL(mk_bgzf_pack_kernel, dim3(n_blocks), dim3(256), 0, s, slots, slot_len, slot_off, packed);
}

}  // namespace mkz
