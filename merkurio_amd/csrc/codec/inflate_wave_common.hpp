// inflate_wave_common.hpp -- what the wave-uniform DEFLATE decoders share (bgzf_inflate_wave.hip: a wave per BGZF member;
// gzip_segments_wave.hip: a wave per piece of a gzip stream): canonical tables built by all lanes, the direct tables whose entries
// carry everything a token turn needs.  RFC 1951.
#pragma once
#include <hip/hip_runtime.h>

#include "inflate_serial.hpp"

namespace mkz {
namespace {

constexpr int kFastLl = 10, kFastD = 9;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Canonical tables of one code from lens[0, n) (n <= 320, in LDS), by the whole wave: sorted[] = symbols in (length, symbol)
// order, limit[l] / base[l] as inflate_serial.hpp defines them.  Returns 0 / 1 over-subscribed /
// 2 incomplete and not one of the shapes zlib accepts.
__device__ int wave_build_tables(const uint8_t *lens, uint32_t n, uint16_t *sorted, uint16_t *limit, uint16_t *base, bool allow_single) {
    const uint32_t lane = lane_id();
    // lane l (1..15) counts the codewords of length l
    uint32_t cnt = 0, used = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v = lens[i];
        cnt += (v == lane);
        used += (v != 0);
    }
    if (lane == 0 || lane > (uint32_t)kMaxBits) cnt = 0;
    // the recurrences over the lengths, the same on every lane; lane l keeps offs[l] (rank of the first symbol of length l)
    int left = 1;
    uint32_t code = 0, off = 0, my_off = 0, prev = 0;
    bool over = false;
    if (lane == 0) limit[0] = 0, base[0] = 0;
    for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)l);
        left = (left << 1) - (int)c;
        over = over || left < 0;
        code = (code + prev) << 1;
        if (lane == l) {
            my_off = off;
            limit[l] = (uint16_t)((code + c) << (kMaxBits - l));
            base[l] = (uint16_t)(off - code);
        }
        off += c;
        prev = c;
    }
    if (over) return 1;
    // symbol i goes to offs[its length] + (symbols of that length in front of it): ranks by ballot, 64 symbols at a time
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + lane;
        const uint32_t mine = i < n ? lens[i] : 0u;
        for (uint32_t l = 1; l <= (uint32_t)kMaxBits; ++l) {
            const uint64_t m = __ballot(mine == l);
            if (m == 0) continue;
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)l);
            if (mine == l) sorted[o + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
            if (lane == l) my_off += (uint32_t)__popcll(m);
        }
    }
    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 1);
    if (left > 0 && !(used == 0 || (allow_single && used == 1 && c1 == 1))) return 2;
    return 0;
}

// what decode_codeword makes of a literal / length codeword (symbol << 4 | length, 0 = none) in the form of ll_fast; 0: not a symbol
// of the alphabet (286, 287)
__device__ __forceinline__ uint32_t pack_ll(uint32_t r) {
    const uint32_t sym = r >> 4;
    if (sym <= 256) return r;
    if (sym > 285) return 0;
    const uint32_t idx = sym - 257;
    return 0x8000u | (length_base(idx) - 3) << 7 | length_extra_bits(idx) << 4 | (r & 15u);
}
__device__ __forceinline__ uint32_t pack_d(uint32_t r) {
    const uint32_t sym = r >> 4;
    if (r == 0 || sym > 29) return 0;
    return distance_base(sym) << 8 | distance_extra_bits(sym) << 4 | (r & 15u);
}
// direct tables from the canonical descriptions: entry e = what decode_codeword makes of the stream bits e, if that codeword is at
// most `bits` long
__device__ void wave_fill_fast_ll(uint16_t *fast, const uint16_t *sorted, const uint16_t *limit, const uint16_t *base) {
    for (uint32_t e = lane_id(); e < (1u << kFastLl); e += 64) {
        const uint32_t r = decode_codeword(e, sorted, limit, base);
        fast[e] = (uint16_t)((r & 15u) <= (uint32_t)kFastLl ? pack_ll(r) : 0u);
    }
}
__device__ void wave_fill_fast_d(uint32_t *fast, const uint16_t *sorted, const uint16_t *limit, const uint16_t *base) {
    for (uint32_t e = lane_id(); e < (1u << kFastD); e += 64) {
        const uint32_t r = decode_codeword(e, sorted, limit, base);
        fast[e] = (r & 15u) <= (uint32_t)kFastD ? pack_d(r) : 0u;
    }
}


}  // namespace
}  // namespace mkz
