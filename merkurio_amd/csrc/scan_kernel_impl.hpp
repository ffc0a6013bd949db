#pragma once
// scan_kernel_impl.hpp -- hand-written gfx950 (CDNA4, wave64) multi-pattern scan kernel (template body;
// instantiated per variant group by scan_variants.hip, dispatched by scan_kernel.hip).
//
// Replaces, for a whole batch of records at once, the reference's per-record matcher calls:
//   BNDMq::find_match / find_iter        src/pattern_matching.rs:82-209
//   AhoCorasick::find_overlapping_iter   src/cmd_extract.rs:332,480,507; src/cmd_tag.rs:393-396
// Result set = every (record, pattern, start) occurrence + per-record any-hit flags
// (bit-exact vs the oracle); emission ORDER is restored afterwards (order_hits.hip, matcher.cpp).
//
// Shape of the work (integer, HBM-streaming; no MFMA -- DESIGN.md §3/§4):
//   * one persistent 1024-thread workgroup per CU (16 waves); level 1, the 128 KiB blocked
//     q-gram Bloom filter of the pattern set, lives in LDS for the life of the workgroup
//     (pattern sets too large for it use the same filter layout in global memory / L2: GF);
//   * the concatenated text is cut into 31 KiB tiles dealt to the 4096 waves in short runs (1, 2 or 4
//     consecutive tiles, then a jump of n_waves runs); a wave walks a tile in 1 KiB chunks: each lane issues ONE non-temporal global_load_dwordx4
//     (64 lanes x 16 B, fully coalesced) per chunk, four chunks (one group) are in flight while
//     the previous group is filtered;
//   * a lane 2-bit-packs its 16 bytes into one dword (11 VALU ops), gets the 32-base halo from
//     lanes +1/+2 by DPP wave_shl (no LDS traffic), forms the 16/S sampled q-gram keys with
//     v_alignbit, hashes with two 24-bit multiplies and probes the filter with one ds_read_b64
//     per sample;
//   * filter positives are compacted (ballot + mbcnt) into a per-wave LDS ring; when the ring
//     fills, 64 candidates at a time go through level 2 (bucketised exact table in L2, loads
//     issued at the end of one group and consumed at the top of the next) and level 3
//     (byte-exact compare, record lookup, boundary check) with all 64 lanes busy;
//   * results: the flag byte of a hit record (stored directly by the kernels for hit-dense text, via a
//     per-wave list of flagged records + a small kernel otherwise), optional (record, pattern, position) tuples
//     staged through a per-wave buffer so the output cursor sees one atomic per ~1000 hits, summary
//     counters summed per workgroup (one global atomic per workgroup and counter); occurrences per
//     pattern are counted from the tuples afterwards (mk_hist_hits_kernel, scan_kernel.hip);
//   * global-filter kernels with a compile-time q also carry 14 context bases with each candidate
//     and fingerprint level 2 with them (filter.hpp: context fingerprints).
#include <algorithm>

#include "scan_kernel.h"

namespace mk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16 ASCII bytes -> 32 bits, base i at bits 2i..2i+1, code = (c >> 1) & 3.
// Per dword: x = d & 0x06060606 holds the four codes at bits 1,9,17,25; x * (2^23+2^17+2^11+2^5)
// lines them up in the TOP BYTE of the product (bits 24..31; the other partial products fall
// on distinct 2-bit slots below bit 24, so nothing carries).  Three v_perm_b32 then gather the
// four top bytes: 4 x (v_and + v_mul_lo) + 3 = 11 VALU ops per 16 bases.
__device__ __forceinline__ uint32_t pack4_top(uint32_t d) { return (d & 0x06060606u) * 0x00820820u; }
__device__ __forceinline__ uint32_t pack16(uint4 v) {
    const uint32_t u0 = pack4_top(v.x), u1 = pack4_top(v.y), u2 = pack4_top(v.z), u3 = pack4_top(v.w);
    // v_perm_b32(src0, src1, sel): bytes 0-3 of the 8-byte pool come from src1, 4-7 from src0
    const uint32_t lo16 = __builtin_amdgcn_perm(u1, u0, 0x0c0c0703u);  // [u0.b3, u1.b3, 0, 0]
    const uint32_t hi16 = __builtin_amdgcn_perm(u3, u2, 0x07030c0cu);  // [0, 0, u2.b3, u3.b3]
    return lo16 | hi16;
}

// 16 bytes at text position pos (pos % 16 == 0); bytes at or beyond n read as 0
__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ seq, uint64_t pos, uint64_t n) {
    if (pos + 16 <= n) return *reinterpret_cast<const uint4 *>(seq + pos);
    uint32_t w[4] = {0, 0, 0, 0};
    if (pos < n) {
        uint32_t rem = (uint32_t)(n - pos);
        for (uint32_t i = 0; i < rem; ++i) w[i >> 2] |= (uint32_t)seq[pos + i] << (8 * (i & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// one filter positive waiting for level 2: the filter hash of its q-gram (level 2 is keyed by it)
// and the low 32 bits of its text position (a wave's queued positions span far less than 4 GiB;
// the high bits are restored from the wave's current position when the entry is taken)
struct alignas(8) CandEntry {
    uint32_t h;
    uint32_t t_lo;
};
constexpr uint32_t kRingEntries = 128;  // per wave; <= 64 pending before an append round of <= 64
constexpr uint32_t kHitSlots = 64;  // per wave: verified-q-gram hits waiting for level 3 (one per lane)
constexpr uint32_t kLdsSumWords = 4;  // per-workgroup sums of the summary counters (candidates, occurrences): they take
                                      // the place of wave 0's candidate ring once every wave has drained its own
constexpr uint32_t kLdsBytes = kBloomBytes + (kBlockThreads / 64) * kRingEntries * sizeof(CandEntry) +
                               (kBlockThreads / 64) * kHitSlots * 8;  // 152 KiB
// two length classes (MC kernels): the short class's q-gram bitmap, one bit per 2-bit-packed key of up to 8 bases
constexpr uint32_t kShortBitmapBytes = kShortBitmapWords * 4;  // 8 KiB
static_assert(kLdsBytes + kShortBitmapBytes <= 160 * 1024, "one workgroup per CU");

// compile-time ablation switches for profiling builds (hipcc -DMK_ABLATE=<bits>):
//   1 = drop every filter positive, 2 = no LDS probe, 4 = loads + pack only,
//   8 = filter positives are queued in the LDS ring but never probed (level 1 -> 2 hand-off cost),
//   16 = level 3 dropped (q-gram hits are queued, never resolved), 32 = level 3 without its stores / atomics,
//   64 = q-gram hits are not even queued, 256 = no record-flag stores
#ifndef MK_ABLATE
#define MK_ABLATE 0
#endif

// ---- level 3: one q-gram hit (pattern `pat` would start at text position p) --------------
// byte-exact (or ASCII-case-folded) comparison of the whole pattern, record lookup, boundary
// check, then flag / counters; returns whether it is a true occurrence and, for EMIT kernels,
// its tuple in `out`.
// FL = flavour of the kernel variant: 1 sparse hits (non-temporal stream, 8-byte compare loads, flagged records
// listed), 0 hit-dense text (cacheable stream, 16-byte compare loads, flags stored directly).  (2 = the sparse
// flavour with 16-byte compare loads still compiles; it is not instantiated: no gain, profiles/r03_cmp16_mid.txt)
template <bool EMIT, int FL>
__device__ __forceinline__ bool resolve_one(const ScanParams &P, uint32_t pat, uint64_t p, uint32_t &n_true, mk_hit &out) {
    // uniform-length pattern sets (every k-mer list): no pat_off lookup, one dependent trip fewer
    if constexpr ((MK_ABLATE & 16) != 0) return false;
    const uint32_t a = P.uniform_len ? pat * P.uniform_len : P.pat_off[pat];
    const uint32_t len = P.uniform_len ? P.uniform_len : P.pat_off[pat + 1] - a;
    if (p + len > P.n_bytes) return false;
    const uint8_t *__restrict__ tx = P.seq + p;
    const uint8_t *__restrict__ pt = P.pat_bytes + a;
    // Everything below that touches memory is independent of everything else, so it is issued
    // together and costs ONE round trip: the record-offset pair at the interpolated record
    // index (reads are mostly of similar length, so the guess is usually right) and the text /
    // pattern loads of the comparison.
    constexpr bool WIDE = FL != 1;  // 16-byte loads
    const uint64_t n = P.n_rec;
    uint64_t rstart, rend, lo;
    if (P.rec_len) {
        // records of one length: record = floor(p / L), no memory access.  The double product is within one of the
        // quotient (p < 2^53, 1 / L rounded once); one compare-and-step makes it exact.
        lo = (uint64_t)((double)p * P.inv_rec_len);
        rstart = lo * P.rec_len;
        if (rstart > p) {
            --lo;
            rstart -= P.rec_len;
        } else if (p - rstart >= P.rec_len) {
            ++lo;
            rstart += P.rec_len;
        }
        rend = rstart + P.rec_len;
    } else {
        // + 1e-6: with equal-length records the quotient of a record's FIRST byte is an integer, and the rounded
        // factor n_rec / n_bytes can land it a hair below -- the guess is then one record short and the wave
        // walks the gallop / bisect path (4-5 dependent memory round trips) for it: 0.8 % of the occurrences,
        // i.e. 4 of 10 drains of 64.  The nudge is far below 1 / record length and above the rounding error
        // for up to ~10^10 records.
        lo = (uint64_t)((double)p * P.rec_per_byte + 1e-6);
        if (P.rec_index) {
            // records of unequal length: the quotient drifts by thousands of records over a 15 GB batch and the
            // gallop below would take ~20 dependent memory round trips (+8 % kernel time at 1 % of the reads
            // hitting, r02_ragged_sweep).  A coarse index -- the record at every 64 Ki-th byte, L2-resident --
            // and interpolation inside its 64 KiB window land within a record or two.
            const uint32_t k = (uint32_t)(p >> kRecIndexShift);
            const uint32_t r0 = P.rec_index[k], r1 = P.rec_index[k + 1];
            lo = r0 + ((((uint32_t)p & ((1u << kRecIndexShift) - 1u)) * (uint64_t)(r1 - r0 + 1)) >> kRecIndexShift);
            if (lo > r1) lo = r1;
        }
        if (lo >= n) lo = n - 1;
        // Two forms of the loads below, chosen with the kernel variant (WIDE = every flavour but the one for
        // sparse hits).  Once more than a few reads in a hundred hit, the kernel is bound by the number of
        // memory REQUESTS per occurrence (the L2 serves ~200 G of them a second next to the stream), so the
        // record-offset pair is one 16-byte request and a pattern of 16..32 bytes -- every k-mer -- is compared
        // with two overlapping 16-byte loads a side, [0, 16) and [len - 16, len): 7.7 -> 5.9 ms per 100 M reads
        // when every read hits.  In the kernels for sparse hits the same code costs the scan loop around it
        // 1.4 % (register allocation; r02_cmp16_ab) and gains nothing (as their own instantiation for 2-12 % of
        // the records hitting either: r03_cmp16_mid), so they keep 8-byte loads.  (seq_off[0] == 0 is part of the ABI.)
        if constexpr (WIDE) {
            uint64_t rpair[2];
            __builtin_memcpy(rpair, P.rec_off + lo, 16);
            rstart = rpair[0], rend = rpair[1];
        } else {
            rstart = P.rec_off[lo], rend = P.rec_off[lo + 1];
        }
    }
    if (P.case_insensitive) {
        for (uint32_t i = 0; i < len; ++i)
            if (fold_ascii(tx[i]) != fold_ascii(pt[i])) return false;
    } else {
        // independent unaligned loads, no early-exit chain; the clamped / overlapping offsets are harmless
        uint64_t diff = 0;
        if (WIDE && len >= 16) {
            const uint32_t last = len - 16;
            uint64_t x[2], y[2], u[2], v[2];
            __builtin_memcpy(x, tx, 16);
            __builtin_memcpy(y, pt, 16);
            __builtin_memcpy(u, tx + last, 16);
            __builtin_memcpy(v, pt + last, 16);
            diff = (x[0] ^ y[0]) | (x[1] ^ y[1]) | (u[0] ^ v[0]) | (u[1] ^ v[1]);
            for (uint32_t o = 16; o < last; o += 16) {  // patterns longer than 32 bytes
                __builtin_memcpy(x, tx + o, 16);
                __builtin_memcpy(y, pt + o, 16);
                diff |= (x[0] ^ y[0]) | (x[1] ^ y[1]);
            }
        } else if (len >= 8) {
            const uint32_t last = len - 8;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t o = 8 * i < last ? 8 * i : last;
                uint64_t x, y;
                __builtin_memcpy(&x, tx + o, 8);
                __builtin_memcpy(&y, pt + o, 8);
                diff |= x ^ y;
            }
            for (uint32_t o = 32; o < len; o += 8) {  // longer patterns
                const uint32_t oo = o < last ? o : last;
                uint64_t x, y;
                __builtin_memcpy(&x, tx + oo, 8);
                __builtin_memcpy(&y, pt + oo, 8);
                diff |= x ^ y;
            }
        } else {
            for (uint32_t i = 0; i < len; ++i) diff |= (uint64_t)(tx[i] ^ pt[i]);
        }
        if (diff) return false;
    }
    // record containing p: largest r with rec_off[r] <= p.  Wrong guess (ragged records):
    // gallop from it to a bracket, then bisect.
    if (!(rstart <= p && p < rend)) {
        uint64_t hi;
        if (rstart <= p) {
            hi = lo + 1;
            uint64_t step = 1;
            while (hi < n && P.rec_off[hi] <= p) {
                lo = hi;
                step <<= 1;
                hi = (n - hi > step) ? hi + step : n;
            }
        } else {
            hi = lo;
            uint64_t step = 1;
            lo = lo - 1;  // lo > 0 here because rec_off[0] = 0 <= p
            while (P.rec_off[lo] > p) {
                hi = lo;
                step <<= 1;
                lo = lo > step ? lo - step : 0;
            }
        }
        while (hi - lo > 1) {  // invariant: rec_off[lo] <= p < rec_off[hi]  (rec_off[n] = n_bytes)
            const uint64_t mid = (lo + hi) >> 1;
            if (P.rec_off[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        rstart = P.rec_off[lo];
        rend = P.rec_off[lo + 1];
    }
    if (p + len > rend) return false;  // occurrence would cross a record boundary
    // ---- a true occurrence.  Its record is flagged by the caller (drain_hits); flagged records are
    // counted afterwards by mk_count_flags_kernel when counters are wanted.
    if constexpr ((MK_ABLATE & 32) != 0) {
        n_true++;
        return false;
    }
    n_true++;
    // (occurrences per pattern are counted from the emitted tuples by mk_hist_hits_kernel after the
    // scan: an atomic per occurrence in here cost 0.65 ms per 10 M occurrences, r02_hitrate_sweep)
    out.rec = lo;
    if (EMIT) {  // the caller stages the tuple (HitStage)
        out.pat = pat;
        out.pos = (uint32_t)(p - rstart);
        // mk_hit.pos is 32 bits: an occurrence 4 GiB or more into its record cannot be reported -- say so instead of
        // wrapping (the host returns MK_E_UNSUPPORTED at its next round trip: mk_matcher_check_device /
        // mk_order_hits_device; mk_scan_batch refuses such a record before it scans.  The word is cleared at the
        // start of every tuple scan: it describes the handle's last one)
        if (p - rstart > 0xFFFFFFFFull) atomicOr(P.error_word, 1u);
    }
    return true;
}

// ---- per-wave buffer of q-gram hits (LDS, 64 x 8 B) ---------------------------------------------
// True q-gram hits are rare; resolving them the moment they are found would run the dependent
// memory round trips of resolve_one with one or two active lanes.  They are collected instead
// and resolved up to 64 at a time (one per lane).  The buffer lives in LDS: a global-memory
// ring cost one store per hit whose acknowledgement the wave's next s_waitcnt vmcnt(0) had to
// wait for (loads and stores share the counter on gfx9) -- 0.1 ms per million hits on the
// slower boxes of the pool -- plus two L2 round trips per drain to read the ring back.
// An entry is {low 32 bits of the occurrence's text position, pattern}; the high bits are
// restored relative to the wave's newest queued position like a CandEntry's.
struct HitRing {
    uint2 *q;        // this wave's buffer (LDS)
    uint32_t count;  // wave-uniform, <= kHitSlots
    // EMIT: verified occurrences of this wave, staged in global memory and moved to the output
    // array kHitStage - 64 or more at a time.  The output cursor is ONE address: an atomic on it
    // costs ~10 ns whoever issues it, so reserving 64 slots per atomic caps the kernel at ~6 G
    // occurrences/s (every read hitting: 20.8 ms per 100 M reads instead of 10.2 without tuples).
    mk_hit *stage;
    uint32_t staged;  // wave-uniform
    // kernels for sparse hits: this wave's list of flagged records (global memory) and its fill
    uint32_t *flist;
    uint32_t nflag;  // wave-uniform
};

// move this wave's staged tuples to the output array: one cursor atomic for all of them.
// FLAGS: the records of the tuples get their flag bytes here (tuple kernels for sparse hits: once per
// ~1000 occurrences instead of once per drain, see drain_hits)
template <bool FLAGS>
__device__ __forceinline__ void flush_stage(const ScanParams &P, HitRing &hr, uint32_t lane) {
    const uint32_t n = hr.staged;
    if (n == 0) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(P.n_hits, (unsigned long long)n);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    // the entries were stored by other lanes of this wave (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (uint32_t i = lane; i < n; i += 64) {
        const uint4 v = reinterpret_cast<const uint4 *>(hr.stage)[i];
        if (base + i < P.hits_cap) reinterpret_cast<uint4 *>(P.hits)[base + i] = v;
        if constexpr (FLAGS && (MK_ABLATE & 256) == 0)
            reinterpret_cast<uint8_t *>(P.rec_flags32)[((uint64_t)v.y << 32) | v.x] = 1;  // mk_hit.rec
    }
    // the next tuples staged must not overtake these reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    hr.staged = 0;
}

template <bool EMIT, int FL>
__device__ __forceinline__ void drain_hits(const ScanParams &P, HitRing &hr, uint64_t newest_end, uint32_t lane, uint32_t &n_true) {
    constexpr bool WIDE = FL == 0;  // hit-dense text: flags stored directly
    const uint32_t n = hr.count;
    // entries were written by other lanes of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint2 e = hr.q[lane & (kHitSlots - 1)];
    __builtin_amdgcn_wave_barrier();
    bool hit = false;
    mk_hit out;
    if (lane < n) {
        uint64_t p = (newest_end & 0xFFFFFFFF00000000ull) | e.x;
        if (p >= newest_end) p -= 1ull << 32;
        hit = resolve_one<EMIT, FL>(P, e.y, p, n_true, out);
    }
    hr.count = 0;
    // The record's flag.  A byte stored into the 100 MB flag array is a partial write to a line that is
    // cached nowhere, and on gfx9 the wave's next s_waitcnt vmcnt -- the one its stream loads wait on --
    // also waits for that store to be acknowledged: on the boxes of the pool where that takes long, 1 % of
    // the reads hitting cost 0.17 ms per launch in flag stores alone (MK_ABLATE=256, r02_flag_target).
    // The kernels for sparse hits therefore append the record index to a per-wave list (one coalesced
    // store per drain) and a small kernel sets the bytes afterwards, where nothing waits for them; a wave
    // whose list is full stores directly.  With hit-dense text every list would overflow: those kernels
    // (WIDE) store directly.
    const uint64_t mm = __ballot(hit);
    if (mm == 0) return;
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
    const uint32_t n_hit = (uint32_t)__popcll(mm);
    if constexpr (EMIT) {
        // tuple kernels for sparse hits: the flags are set from the staged tuples when those are flushed
        if constexpr (WIDE && (MK_ABLATE & 256) == 0)
            if (hit) reinterpret_cast<uint8_t *>(P.rec_flags32)[out.rec] = 1;
        if (hit) hr.stage[hr.staged + below] = out;
        hr.staged += n_hit;
        if (hr.staged > kHitStage - 64) flush_stage<!WIDE>(P, hr, lane);  // no room for another full round
    } else if constexpr ((MK_ABLATE & 256) == 0) {
        if (!WIDE && hr.flist && hr.nflag + n_hit <= P.flag_cap) {
            if (hit) hr.flist[hr.nflag + below] = (uint32_t)out.rec;
            hr.nflag += n_hit;
        } else if (hit) {
            reinterpret_cast<uint8_t *>(P.rec_flags32)[out.rec] = 1;
        }
    }
}

// ---- level 2: up to 64 filter positives (one per lane) against the exact q-gram table -----
// One bucket (32 B, 4 entries) per lane and round.  probe_round() consumes a bucket that is
// already in registers: fingerprint matches are compacted (ballot/popcount) into the wave's
// hit ring; returns whether this lane must look at the next bucket (its bucket has overflowed).
// CS > 0: context kernels (filter.hpp: gf_has_ctx), CS = the sampling stride; an entry's fingerprint
// is then the q-gram hash mixed with the candidate's context bases under the mask of the entry's offset.
template <bool EMIT, int CS, int FL>
__device__ __forceinline__ bool probe_round(const ScanParams &P, bool active, uint32_t fp, uint32_t ctx, uint64_t t, uint4 v0, uint4 v1,
                                            uint32_t lane, HitRing &hr, uint64_t newest_end, uint32_t &n_true) {
    const uint32_t efp[4] = {v0.x, v0.z, v1.x, v1.z};
    const uint32_t epo[4] = {v0.y, v0.w, v1.y, v1.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t e_off = epo[k] & 15u, e_pat = (epo[k] & ~kBucketOverflow) >> 4;
        uint32_t want = fp;
        if constexpr (CS > 0) want = ctx_fp(fp, ctx & ctx_mask(e_off & (uint32_t)(CS - 1), (uint32_t)CS));
        const bool match = active && epo[k] != kEmptyPat && efp[k] == want && t >= e_off;
        const uint64_t mm = (MK_ABLATE & 64) ? 0ull : __ballot(match);
        if (mm) {  // uniform, rare
            const uint32_t cnt = (uint32_t)__popcll(mm);
            if (hr.count + cnt > kHitSlots) drain_hits<EMIT, FL>(P, hr, newest_end, lane, n_true);  // make room
            if (match) {
                const uint32_t below =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                hr.q[hr.count + below] = make_uint2((uint32_t)(t - e_off), e_pat);
            }
            hr.count += cnt;
        }
    }
    // the bucket overflowed when the table was built: the key may live in the next one
    return active && epo[0] != kEmptyPat && (epo[0] & kBucketOverflow) != 0;
}

__device__ __forceinline__ void load_bucket(const ScanParams &P, bool active, uint32_t b, uint4 &v0, uint4 &v1) {
    v0 = make_uint4(0, kEmptyPat, 0, kEmptyPat);
    v1 = v0;
    if (active) {
        const uint4 *bp = reinterpret_cast<const uint4 *>(P.table + (size_t)b * kBucketEntries);
        v0 = bp[0];
        v1 = bp[1];
    }
}

// synchronous probe: wave-uniform loop over the bucket chain (one iteration unless a home
// bucket is full), each iteration one memory round trip
template <bool EMIT, int CS, int FL>
__device__ __forceinline__ void probe_chain(const ScanParams &P, bool active, uint32_t b, uint32_t fp, uint32_t ctx, uint64_t t,
                                            uint32_t lane, HitRing &hr, uint64_t newest_end, uint32_t &n_true) {
    while (__ballot(active)) {
        uint4 v0, v1;
        load_bucket(P, active, b, v0, v1);
        active = probe_round<EMIT, CS, FL>(P, active, fp, ctx, t, v0, v1, lane, hr, newest_end, n_true);
        b = (b + 1) & P.table_mask;
    }
}

#ifndef MK_ISSUE_AT
#define MK_ISSUE_AT 56  // ring fill at which a group ends with an asynchronous level-2 probe
#endif
// hit-dense flavour (FL == 0): level 3 re-reads the occurrence's text, which the wave streamed a while ago -- by then
// ~80 MB have passed through the chip and the line comes back from the Infinity Cache (66 G random lines/s) instead of
// the XCD's L2.  Probing from 16 queued candidates and resolving from 16 queued q-gram hits (instead of 56 / a full
// buffer of 64) halves that distance: every read hitting 5.94 -> 5.43 ms, a third of them 4.0 -> 3.8
// (profiles/r03_dense_drain_ab.txt; 8 / 8 and flushing the parked candidates every group gain nothing more; in the
// sparse flavour, whose non-temporal stream is in no cache anyway, 16 / 16 costs 3-6 %).
#ifndef MK_ISSUE_AT_DENSE
#define MK_ISSUE_AT_DENSE 16
#endif
#ifndef MK_DRAIN_AT_DENSE
#define MK_DRAIN_AT_DENSE 16  // q-gram hits waiting at the end of a group from which they are resolved (65 = only when full)
#endif

// Geometry of one kernel variant.  QC > 0: q-gram length fixed at compile time (the k-mer
// sizes that matter get their own kernels: no runtime masks, no unused halo words);
// QC == 0: runtime q <= 16 (32-bit keys); QC == -1: runtime q in 17..32.
template <int S, int QC, bool CTX, int MC = 0>
struct Geo {
    static constexpr bool kFixed = QC > 0;
    static constexpr int kNS = 16 / S;                                     // samples per lane per chunk
    static constexpr int kSpan = kFixed ? (kNS - 1) * S + QC + (CTX ? 7 : 0) : 48;  // bases a lane looks at
    // (the short class of an MC kernel looks at up to 15 + 8 bases)
    static constexpr bool kNeedW1 = kSpan > 16 || MC != 0, kNeedW2 = kSpan > 32;      // halo words
};

// ---- second length class (MC kernels; matcher.cpp: plan_classes) ------------------------------------------------
// Patterns too short for the main class's q-grams form a class of their own: stride S2 in {1, 2, 4, 8}, q-grams of
// q2 <= 8 bases, level 1 = a plain table over the 4^q2 packed keys in LDS (exact on the 2-bit codes, no hash): one
// BYTE per key for q2 <= 6 (4 KiB; bit-field extract, ds_read_u8, shift-or: 3 vector operations per sample against the
// ~10 of a hashed Bloom probe) or one BIT per key for q2 = 7, 8 (8 KiB, ~6 operations).  Both classes are probed from
// the same packed registers in the same pass; their candidates share the ring, the exact table and level 3 (a class
// bit in the fingerprint keeps an occurrence from being found through the other class's samples a second time:
// filter.hpp, short_fp).
template <int S2, bool BYTES>
__device__ __forceinline__ uint32_t short_filter(const uint32_t *__restrict__ bm, uint32_t w0, uint32_t w1, uint32_t kmask) {
    uint32_t cand = 0;
#pragma unroll
    for (int j = 0; j < 16 / S2; ++j) {
        const int sh = 2 * j * S2;  // < 32
        // (keys of the byte table have at most 12 bits: those that end inside w0 need no alignbit)
        const uint32_t x = sh == 0 ? w0 : (BYTES && sh <= 20) ? (w0 >> sh) : __builtin_amdgcn_alignbit(w1, w0, sh);
        const uint32_t key = x & kmask;
        if constexpr (BYTES) {
            cand |= (uint32_t) reinterpret_cast<const uint8_t *>(bm)[key] << j;
        } else {
            const uint32_t word = bm[key >> 5];
            cand |= ((word >> (key & 31u)) & 1u) << j;
        }
    }
    return cand;
}

// bits [bit, bit+32) of the packed stream w0 | w1<<32 | w2<<64, bit a run-time value below 96
__device__ __forceinline__ uint32_t stream32_rt(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t bit) {
    const uint32_t k = bit >> 5;
    const uint32_t a = k == 0 ? w0 : k == 1 ? w1 : w2;
    const uint32_t b = k == 0 ? w1 : k == 1 ? w2 : 0u;
    return __builtin_amdgcn_alignbit(b, a, bit & 31u);  // shift 0 yields a
}

// bits [bit, bit+32) of the packed stream w0 | w1<<32 | w2<<64 (bit is a compile-time constant)
__device__ __forceinline__ uint32_t stream32(uint32_t w0, uint32_t w1, uint32_t w2, int bit) {
    const int k = bit >> 5, r = bit & 31;
    const uint32_t a = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : 0u;
    const uint32_t b = k == 0 ? w1 : k == 1 ? w2 : 0u;
    return r ? __builtin_amdgcn_alignbit(b, a, r) : a;
}

// Bloom hash of the q-gram that starts `sh` bits into the lane's packed stream; equals
// filter.hpp's bloom_hash(key) for the masked key.
template <int S, int QC>
__device__ __forceinline__ uint32_t sample_hash(uint32_t w0, uint32_t w1, uint32_t w2, int sh, uint32_t mask_lo,
                                                uint32_t mask_hi) {
    constexpr uint32_t C1 = 0x9E3779u, C2 = 0x85EBCBu;
    if constexpr (QC > 0) {
        const uint32_t lo = stream32(w0, w1, w2, sh);
        if constexpr (QC <= 12) {
            return __umul24(lo & ((1u << (2 * QC)) - 1u), C1);
        } else if constexpr (QC < 16) {
            const uint32_t l = lo & ((1u << (2 * QC)) - 1u);
            return __umul24(l, C1) + __umul24(l >> 24, C2);
        } else if constexpr (QC == 16) {
            return __umul24(lo, C1) + __umul24(lo >> 24, C2);
        } else if constexpr (QC < 24) {  // key bits 24 .. 2q-1 live in t's low bits
            const uint32_t t = stream32(w0, w1, w2, sh + 24) & ((1u << (2 * QC - 24)) - 1u);
            return __umul24(lo, C1) + __umul24(t, C2);
        } else if constexpr (QC == 24) {  // mul24 ignores t's top byte: no mask at all
            const uint32_t t = stream32(w0, w1, w2, sh + 24);
            return __umul24(lo, C1) + __umul24(t, C2);
        } else {  // 25..32: key bits 48..55 are added in place, bits >= 56 ignored
            const uint32_t t = stream32(w0, w1, w2, sh + 24);
            constexpr uint32_t top = QC >= 28 ? 0xFF000000u : (((1u << (2 * QC - 48)) - 1u) << 24);
            return __umul24(lo, C1) + __umul24(t, C2) + (t & top);
        }
    } else {
        const uint32_t lo = stream32(w0, w1, w2, sh) & mask_lo;
        const uint32_t hi = QC == 0 ? 0u : (stream32(w0, w1, w2, sh + 32) & mask_hi);
        const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 24);
        return __umul24(lo, C1) + __umul24(t, C2) + (t & 0xFF000000u);
    }
}

// ---- main kernel -----------------------------------------------------------------------
// MC: 0 = one length class; 1 = two classes, the short class's stride and table kind read at run time (a switch per
// chunk: ~0.17 ms per 15 GB, profiles/r04_mc_ablate.txt); 2 / 4 / 8 = two classes, byte table, THAT stride compiled in
// (the k-mer families only: a k-mer set plus a few short motifs is the case that matters)
template <int S, int QC, bool EMIT, bool GF, int FL, int MC = 0>
__global__ __launch_bounds__(kBlockThreads) void mk_scan_kernel(const ScanParams P) {
    static_assert(!(MC != 0 && GF && QC > 0), "two length classes with the main filter in global memory: the runtime-q kernels only "
                                              "(the context fingerprints of the fixed-q ones cover the main class's geometry)");
    constexpr bool NTL = FL != 0;  // non-temporal stream loads
    // context kernels: global filter with a compile-time q (filter.hpp: gf_has_ctx); kPipe: their
    // filter probes run one chunk ahead of their use (two samples per lane keeps that in registers)
    constexpr bool kCtx = GF && QC > 0;
    constexpr int CS = kCtx ? S : 0;
    using G = Geo<S, QC, kCtx, MC>;
    constexpr bool kPipe = kCtx && G::kNS <= 2 && (MK_ABLATE & 7) == 0;
    constexpr bool kWide = FL == 0;  // hit-dense flavour: flags stored directly (drain_hits)
    // MC kernels: the short class's table comes FIRST (its ds_read_u8 / ds_read_b32 then carry their base as the
    // instruction's 16-bit offset; behind the 128 KiB filter every probe paid a v_add for it), the filter behind it
    constexpr uint32_t kShortWords = MC != 0 ? kShortBitmapWords : 0;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kShortWords + kLdsBytes / 4];  // (short-class table +) filter + candidate rings
    uint32_t *const bloom = lds_all + kShortWords;
    uint32_t *lds_sums = bloom + kBloomWords;  // wave 0's candidate ring, once every wave is done with its own
    const uint32_t *short_bm = lds_all;  // MC
    if constexpr (!GF) {  // stage the filter image of the pattern set in LDS
        const uint4 *src = reinterpret_cast<const uint4 *>(P.bloom);
        uint4 *dst = reinterpret_cast<uint4 *>(bloom);
        for (uint32_t i = threadIdx.x; i < kBloomWords / 4; i += kBlockThreads) dst[i] = src[i];
    }
    if constexpr (MC != 0) {  // ... and the short class's table (also next to a main filter in global memory)
        const uint4 *src2 = reinterpret_cast<const uint4 *>(P.short_bitmap);
        uint4 *dst2 = reinterpret_cast<uint4 *>(lds_all);
        for (uint32_t i = threadIdx.x; i < kShortBitmapWords / 4; i += kBlockThreads) dst2[i] = src2[i];
    }
    if constexpr (!GF || MC != 0) __syncthreads();
    const uint2 *__restrict__ gbloom = reinterpret_cast<const uint2 *>(P.bloom);  // GF: filter blocks in global memory
    const uint32_t gmask = P.gbloom_blocks;  // number of 64-bit filter blocks in global memory

    if (P.counters && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&P.counters[P.n_pat + MK_SUM_RECORDS], (unsigned long long)P.n_rec);
        atomicAdd(&P.counters[P.n_pat + MK_SUM_BASES], (unsigned long long)P.n_bytes);
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // SGPR
    const uint64_t wave_id = (uint64_t)blockIdx.x * (kBlockThreads / 64) + wave_in_block;
    const uint64_t n_waves = (uint64_t)gridDim.x * (kBlockThreads / 64);
    constexpr int NS = G::kNS;
    constexpr uint64_t kTileBytes = (uint64_t)kTileChunks * kChunkBytes;
    const uint32_t mask_lo = P.key_mask_lo, mask_hi = P.key_mask_hi;
    const uint8_t *__restrict__ seq = P.seq;
    const uint64_t n_bytes = P.n_bytes;
    CandEntry *ring = reinterpret_cast<CandEntry *>(bloom + kBloomWords) + wave_in_block * kRingEntries;
    // context words of the queued candidates, parallel to `ring` (the filter image is not in LDS in
    // global-filter mode: its space is free)
    uint32_t *ctx_ring = bloom + wave_in_block * kRingEntries;
    uint32_t q_head = 0, q_count = 0, n_cand = 0;  // wave-uniform
    HitRing hr;  // this wave's q-gram-hit ring
    hr.q = reinterpret_cast<uint2 *>(bloom + kBloomWords + (kBlockThreads / 64) * kRingEntries * 2) + wave_in_block * kHitSlots;
    hr.count = 0;
    hr.stage = EMIT ? P.stage + wave_id * (uint64_t)kHitStage : nullptr;
    hr.staged = 0;
    hr.flist = (!kWide && !EMIT && P.flag_list) ? P.flag_list + wave_id * (uint64_t)P.flag_cap : nullptr;
    hr.nflag = 0;
    uint32_t n_true = 0;  // per lane: occurrences found
    uint32_t abl_acc = 0;              // ablation builds only

    // ---- level 1 for one 1 KiB chunk: pk_cur = this lane's 16 packed bases, pk_nxt = the
    // packed chunk that follows in the text (halo source for the last lanes).  Returns the
    // lane's candidate mask (bit j = sample j passed the filter).  Straight-line code.
    // Every reference-capturing lambda below is always_inline: past some size the inliner leaves
    // one of them as a real function, its captures (and the kernel arguments) then live in scratch
    // memory and the LDS ring is reached through flat instructions -- a 3x slower kernel.
    auto halo = [&](uint32_t pk_cur, uint32_t pk_nxt, uint32_t &w1, uint32_t &w2) __attribute__((always_inline)) {
        // lane i needs the packed dwords of lanes i+1 and i+2; DPP wave_shl:1 moves a whole
        // wave by one lane in one VALU op, lane 63 keeps `old` = the next chunk's lane
        w1 = 0;
        w2 = 0;
        if constexpr (G::kNeedW1) {
            const uint32_t n0 = __builtin_amdgcn_readlane(pk_nxt, 0);
            w1 = __builtin_amdgcn_update_dpp(n0, pk_cur, 0x130, 0xf, 0xf, false);
        }
        if constexpr (G::kNeedW2) {
            const uint32_t n1 = __builtin_amdgcn_readlane(pk_nxt, 1);
            w2 = __builtin_amdgcn_update_dpp(n1, w1, 0x130, 0xf, 0xf, false);
        }
    };
    auto filter_chunk = [&](uint32_t pk_cur, uint32_t pk_nxt, uint32_t &h0, uint32_t &h1) __attribute__((always_inline)) -> uint32_t {
        const uint32_t w0 = pk_cur;
        uint32_t w1, w2;
        halo(pk_cur, pk_nxt, w1, w2);
        if constexpr ((MK_ABLATE & 4) != 0) return ((w0 ^ w1 ^ w2) == 0x12345678u) ? 1u : 0u;  // loads + pack only
        uint32_t cand = 0;
        const char *bloom_bytes = reinterpret_cast<const char *>(bloom);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            uint32_t h = sample_hash<S, QC>(w0, w1, w2, 2 * j * S, mask_lo, mask_hi);
            if (j == 0) h0 = h;  // kept for the hand-off when a lane has at most two samples
            if (j == 1) h1 = h;
            uint2 blk;
            if constexpr ((MK_ABLATE & 2) != 0) {
                blk = make_uint2(h & 0x10101010u, h);
            } else if constexpr (GF) {
                blk = gbloom[gbloom_block(h, gmask)];  // random 8-byte read, L2 / Infinity Cache
                h = gbloom_bits(h);
            } else {
                blk = *reinterpret_cast<const uint2 *>(bloom_bytes + bloom_block_byte(h));  // ds_read_b64
            }
            // all three filter bits set?  (shift counts use the low 5 bits of their register)
            uint32_t m = (blk.x >> (h >> 27)) & (blk.y >> (h >> 22)) & (blk.y >> (h >> 17));
            if constexpr (GF) m &= blk.x >> (h >> 12);  // the global filter's fourth bit (d)
            cand |= (m & 1u) << j;
        }
        if constexpr (MC != 0) {  // the short class's samples: bits 16.. of the mask (wave-uniform stride)
            uint32_t c2;
#if defined(MK_MC_ABL) && MK_MC_ABL == 1  // profiling builds: the two-class kernel without its short-class samples
            c2 = 0;
#elif defined(MK_MC_ABL) && MK_MC_ABL == 2  // ... with one compile-time geometry instead of the switch
            c2 = short_filter<4, true>(short_bm, w0, w1, P.key2_mask);
#else
            if constexpr (MC > 1) {
                c2 = short_filter<MC, true>(short_bm, w0, w1, P.key2_mask);
            } else if (P.short_bytes) {
                switch (P.s2) {
                    case 1: c2 = short_filter<1, true>(short_bm, w0, w1, P.key2_mask); break;
                    case 2: c2 = short_filter<2, true>(short_bm, w0, w1, P.key2_mask); break;
                    case 4: c2 = short_filter<4, true>(short_bm, w0, w1, P.key2_mask); break;
                    default: c2 = short_filter<8, true>(short_bm, w0, w1, P.key2_mask); break;
                }
            } else {
                switch (P.s2) {
                    case 1: c2 = short_filter<1, false>(short_bm, w0, w1, P.key2_mask); break;
                    case 2: c2 = short_filter<2, false>(short_bm, w0, w1, P.key2_mask); break;
                    case 4: c2 = short_filter<4, false>(short_bm, w0, w1, P.key2_mask); break;
                    default: c2 = short_filter<8, false>(short_bm, w0, w1, P.key2_mask); break;
                }
            }
#endif
            cand |= c2 << 16;
        }
        if constexpr ((MK_ABLATE & 1) != 0) {  // keep the filter work alive, drop its result
            abl_acc += cand;
            cand = 0;
        }
        return cand;
    };

    // ---- level 2 drains.  Synchronous: take n (<= 64) candidates off the ring and walk their
    // bucket chains.  Asynchronous (the normal case): issue the bucket loads now, keep them in
    // registers, and consume them at the top of the next group -- by then the loads are older
    // than the stream loads the wave has waited for anyway, so the probe's memory round trip
    // overlaps a whole group of scanning instead of stalling the wave (and its prefetches).
    bool pend_on = false;  // wave-uniform
    bool pend_active = false;
    uint32_t pend_fp = 0, pend_b = 0, pend_ctx = 0;
    uint64_t pend_t = 0;
    uint4 pend_v0 = make_uint4(0, 0, 0, 0), pend_v1 = pend_v0;
    uint64_t newest_end = 0;  // wave-uniform: every queued position is < newest_end (and > newest_end - 4 GiB)
    auto take_from_ring = [&](uint32_t n, bool &active, uint32_t &b, uint32_t &fp, uint32_t &ctx, uint64_t &t) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const CandEntry e = ring[(q_head + lane) & (kRingEntries - 1)];
        ctx = 0;
        if constexpr (kCtx) ctx = ctx_ring[(q_head + lane) & (kRingEntries - 1)];
        __builtin_amdgcn_wave_barrier();
        active = lane < n;
        b = table_bucket(e.h, P.table_mask);
        fp = e.h;
        t = (newest_end & 0xFFFFFFFF00000000ull) | e.t_lo;
        if (t >= newest_end) t -= 1ull << 32;
        q_head = (q_head + n) & (kRingEntries - 1);
        q_count -= n;
    };
    auto drain_ring = [&](uint32_t n) __attribute__((always_inline)) {  // synchronous
        bool active;
        uint32_t b, fp;
        uint64_t t;
        uint32_t ctx;
        take_from_ring(n, active, b, fp, ctx, t);
        probe_chain<EMIT, CS, FL>(P, active, b, fp, ctx, t, lane, hr, newest_end, n_true);
    };
    auto issue_probe = [&](uint32_t n) __attribute__((always_inline)) {  // asynchronous: loads only
        take_from_ring(n, pend_active, pend_b, pend_fp, pend_ctx, pend_t);
        load_bucket(P, pend_active, pend_b, pend_v0, pend_v1);
        pend_on = true;
    };
    auto consume_probe = [&]() __attribute__((always_inline)) {
        const bool more = probe_round<EMIT, CS, FL>(P, pend_active, pend_fp, pend_ctx, pend_t, pend_v0, pend_v1, lane, hr, newest_end, n_true);
        if (__ballot(more))  // some home bucket had overflowed: finish those chains synchronously
            probe_chain<EMIT, CS, FL>(P, more, (pend_b + 1) & P.table_mask, pend_fp, pend_ctx, pend_t, lane, hr, newest_end, n_true);
        pend_on = false;
    };

    // ---- level 1 -> 2 hand-off: filter positives -> per-wave LDS ring (ballot/popcount
    // compaction); 64 at a time they are probed against the exact table, so the L2 round trip is
    // paid once per 64 candidates, not per chunk.  With ~2 positives per 1 KiB chunk almost every
    // chunk has one somewhere in the wave, and a compaction round per chunk costs as much as
    // half the filter itself; so a positive first parks in its lane's register slot, and the
    // slots are compacted into the ring only when some lane needs its slot a second time
    // (every ~5 chunks: a birthday collision among 64 lanes).
    uint32_t slot_h = 0, slot_t = 0, slot_c = 0;
    bool slot_full = false;
    auto flush_slots = [&]() __attribute__((always_inline)) {
        const uint64_t full = __ballot(slot_full);
        if (!full) return;
        if (q_count > 64) drain_ring(64);  // an append round adds <= 64 entries to the 128-entry ring
        if (slot_full) {
            const uint32_t below =
                __builtin_amdgcn_mbcnt_hi((uint32_t)(full >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)full, 0u));
            CandEntry e;
            e.h = slot_h;
            e.t_lo = slot_t;
            ring[(q_head + q_count + below) & (kRingEntries - 1)] = e;
            if constexpr (kCtx) ctx_ring[(q_head + q_count + below) & (kRingEntries - 1)] = slot_c;
        }
        q_count += (uint32_t)__popcll(full);
        n_cand += (uint32_t)__popcll(full);
        slot_full = false;
    };
    // wm1 (context kernels): the packed word of the 16 bases BEFORE this lane's (lane - 1's, or the
    // previous chunk's last lane's)
    auto queue_candidates = [&](uint32_t pk_cur, uint32_t pk_nxt, uint32_t wm1, uint32_t cand, uint64_t cpos, uint32_t h0, uint32_t h1) __attribute__((always_inline)) {
        const uint32_t w0 = pk_cur;
        uint32_t w1 = 0, w2 = 0;
        if constexpr (NS > 2 || kCtx || MC != 0) halo(pk_cur, pk_nxt, w1, w2);  // the hash is recomputed below / the context needs the halo
        newest_end = cpos + kChunkBytes;  // parked and queued positions are all below it
        const uint32_t t_base = (uint32_t)cpos + lane * 16;
        do {  // wave-uniform; one iteration unless a lane has several positives in this chunk
            if (__ballot(cand != 0 && slot_full)) flush_slots();
            if (cand != 0) {
                const uint32_t j = (uint32_t)__ffs(cand) - 1u;
                cand &= cand - 1;
                if constexpr (NS <= 2) {
                    slot_h = j ? h1 : h0;
                } else {  // same value as sample_hash / filter.hpp's bloom_hash of the masked key
                    const uint32_t sh = 2u * j * S;
                    const uint32_t klo = __builtin_amdgcn_alignbit(w1, w0, sh) & mask_lo;  // sh == 0 -> w0
                    const uint32_t khi = __builtin_amdgcn_alignbit(w2, w1, sh) & mask_hi;
                    const uint32_t t = __builtin_amdgcn_alignbit(khi, klo, 24);
                    slot_h = __umul24(klo, 0x9E3779u) + __umul24(t, 0x85EBCBu) + (t & 0xFF000000u);
                }
                if constexpr (kCtx) {  // bases t-7..t-1 and t+q..t+q+6 around the sampled q-gram at t = 16 lane + j S
                    const uint32_t pb = 2u * j * S;
                    const uint32_t prev7 = pb >= 14u ? (w0 >> (pb - 14u)) : __builtin_amdgcn_alignbit(w0, wm1, (pb + 18u) & 31u);
                    const uint32_t next7 = stream32_rt(w0, w1, w2, pb + 2u * (uint32_t)QC);
                    slot_c = (prev7 & 0x3FFFu) | ((next7 & 0x3FFFu) << 14);
                }
                slot_t = t_base + j * S;
                if constexpr (MC != 0) {
                    // bits 16.. of the mask are the short class's samples: fingerprint = its packed key (filter.hpp).
                    // Both forms are computed and one is selected: a branch here diverges (the lanes of a wave hold
                    // candidates of both classes) and would run both sides anyway.
                    const uint32_t j2 = (j - 16u) << P.s2_log2;  // base offset of the short sample inside the lane's 16
                    const uint32_t fp2 = short_fp(__builtin_amdgcn_alignbit(w1, w0, 2u * j2) & P.key2_mask);
                    const bool is2 = j >= 16u;
                    slot_h = is2 ? fp2 : main_fp(slot_h);
                    slot_t = is2 ? t_base + j2 : slot_t;
                }
                slot_full = true;
            }
        } while (__ballot(cand != 0));
    };

    // ---- main phase: tiles whose 32 chunk loads (31 scanned + halo) lie inside the text.
    // No bounds checks here; loads run one group (4 chunks = 4 KiB per wave, 64 KiB per CU)
    // ahead of their use.
    const uint64_t n_main_tiles = n_bytes >= kChunkBytes ? (n_bytes - kChunkBytes) / kTileBytes : 0;
    if (wave_id * (P.tile_run ? P.tile_run : 1u) < n_main_tiles) {
        // loader cursor (wave-uniform): pointer to the next chunk to fetch.  Past this wave's
        // last tile the pointer parks on the last main tile: the loads stay unconditional (a
        // branch around a load would force s_waitcnt vmcnt(0) at the join), their data unused.
        // Tile dealing: a wave takes `run` consecutive tiles, then jumps ahead by n_waves * run tiles
        // (run = 1: plain round-robin).  Longer runs keep a wave inside one 2 MiB page for several tiles.
        const uint32_t run = P.tile_run ? P.tile_run : 1u;
        const uint64_t jump = (n_waves - 1) * (uint64_t)run + 1;  // from the last tile of a run to the first of the next
        uint64_t ld_tile = wave_id * run;
        uint32_t ld_run = 0;
        uint32_t ld_g = 0;  // group index inside the tile (a tile is 8 groups of 4 chunk loads)
        const uint64_t last_tile = n_main_tiles - 1;
        const uint8_t *ld_ptr = seq + (ld_tile < last_tile ? ld_tile : last_tile) * kTileBytes + lane * 16;
        auto nt_load = [](const uint8_t *p) -> uint4 {
            // NTL (the normal case): non-temporal, the text is read once; keep L2 for the exact table and
            // the filter image (-15 % kernel time).  Plain loads are the variant for hit-dense text, where
            // level 3 re-reads every verified window: a non-temporal stream has left L2 by then (2.4x HBM
            // traffic when every read hits), a plain one is still in L2 / Infinity Cache.  The host picks
            // the variant per launch from the hit density it has seen (matcher.cpp).
            if constexpr (NTL) {
                const u32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
                return make_uint4(nv.x, nv.y, nv.z, nv.w);
            } else {
                return *reinterpret_cast<const uint4 *>(p);
            }
        };
        // four loads off one address register (immediate offsets); the tile wrap is checked
        // once per group so that the group stays one basic block
        auto load_group = [&](uint4 &a, uint4 &b, uint4 &c, uint4 &d) __attribute__((always_inline)) {
            a = nt_load(ld_ptr);
            b = nt_load(ld_ptr + kChunkBytes);
            c = nt_load(ld_ptr + 2 * kChunkBytes);
            d = nt_load(ld_ptr + 3 * kChunkBytes);
            ld_ptr += 4 * kChunkBytes;
            if (++ld_g == (uint32_t)(kTileChunks + 1) / 4) {  // next tile of this wave
                ld_g = 0;
                if (++ld_run == run) {
                    ld_run = 0;
                    ld_tile += jump;
                } else {
                    ld_tile += 1;
                }
                const uint64_t t = ld_tile < last_tile ? ld_tile : last_tile;
                ld_ptr = seq + t * kTileBytes + lane * 16;
            }
        };
        // Register pipeline, one group = 4 chunks deep.  A group (straight-line code, so the
        // compiler's s_waitcnt vmcnt(N) stay counted): pack the four chunks loaded one group
        // ago, re-issue the four loads, run level 1 on the four chunk pairs (pk_prev,p0) ..
        // (p2,p3).  Only if some lane has a filter positive does the wave enter the (looped,
        // single-copy) hand-off code.
        uint4 r0, r1, r2, r3;
        load_group(r0, r1, r2, r3);
        uint32_t pk_prev = 0;
        // context kernels: the packed chunk before pk_prev (its last lane feeds lane 0's context) and,
        // per tile, the packed 16 bases in front of the tile (uniform; the wave does not stream them)
        uint32_t pk_pp = 0, pk_tile_prev = 0;
        // kPipe: the chunk whose filter probes are in flight
        bool pd_on = false;  // wave-uniform
        uint32_t pd_h0 = 0, pd_h1 = 0, pd_cur = 0, pd_nxt = 0, pd_wm1 = 0;
        uint2 pd_b0 = make_uint2(0, 0), pd_b1 = pd_b0;
        uint64_t pd_cpos = 0;
        auto test_pending = [&]() __attribute__((always_inline)) {
            auto pass = [](uint32_t h, uint2 blk) -> uint32_t {
                const uint32_t hb = gbloom_bits(h);
                return (blk.x >> (hb >> 27)) & (blk.x >> (hb >> 12)) & (blk.y >> (hb >> 22)) & (blk.y >> (hb >> 17)) & 1u;
            };
            uint32_t cand = pass(pd_h0, pd_b0);
            if constexpr (NS > 1) cand |= pass(pd_h1, pd_b1) << 1;
            if (__ballot(cand != 0)) queue_candidates(pd_cur, pd_nxt, pd_wm1, cand, pd_cpos, pd_h0, pd_h1);
        };
        // Queued filter positives carry only the low 32 bits of their position, restored relative to
        // the wave's current position: none may stay queued while the wave advances 4 GiB.  A wave's
        // runs of tiles are n_waves * run * 31 KiB <= 1 GiB apart, so after every 1 GiB of advance whatever
        // is parked or queued is pushed on to level 2.  (Not 2 GiB: a push that finds a probe still pending
        // leaves the queued candidates for the NEXT push, and two intervals must stay below 4 GiB -- r02,
        // test_sparse_candidates_full_size caught exactly that.)  (Sparse candidates never reach the ring's
        // fill threshold by themselves: 1 pattern on 15 GB lost two hits in three before this).
        uint64_t last_push_base = 0;
        uint32_t sc_run = 0;
        for (uint64_t tile = wave_id * run; tile < n_main_tiles;) {
            const uint64_t base = tile * kTileBytes;
            if constexpr (kCtx) {  // 16 bases in front of the tile (one address for the whole wave)
                pk_tile_prev = 0;
                if (base >= 16) pk_tile_prev = pack16(*reinterpret_cast<const uint4 *>(seq + base - 16));
            }
            if (base - last_push_base >= (1ull << 30)) {
                last_push_base = base;
                flush_slots();
                if (hr.count) drain_hits<EMIT, FL>(P, hr, newest_end, lane, n_true);  // 32-bit positions too
                if (q_count && !pend_on) issue_probe(q_count < 64 ? q_count : 64);
            }
#pragma unroll 1
            for (int g = 0; g < (kTileChunks + 1) / 4; ++g) {
                const uint32_t p0 = pack16(r0), p1 = pack16(r1), p2 = pack16(r2), p3 = pack16(r3);
                // pack BEFORE re-issuing the loads into the same registers: without this pin the
                // compiler sinks the packs below the loads and keeps the raw data alive with 16
                // v_mov per group
                asm volatile("" ::"v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
                // (re-issuing the stream loads first is 2-3 % slower: tools/ab3.sh, r01)
                if (pend_on) consume_probe();  // its loads are older than the stream loads just waited for
                load_group(r0, r1, r2, r3);
                // one looped copy of filter + hand-off per chunk (rotating the packed registers); the
                // chunk scanned by pair k is 4g + k - 1: the first pair of a tile straddles tiles and is skipped
                {
                    uint32_t q0 = p0, q1 = p1, q2 = p2;
#pragma unroll 1
                    for (int k = 0; k < 4; ++k) {
                        const int ci = 4 * g + k - 1;
                        if constexpr (kPipe) {
                            // global filter: issue the probes of chunk ci, then test those of the chunk before
                            // -- its loads are the older ones, so the wait leaves the new probes in flight
                            uint32_t nh0 = 0, nh1 = 0, nwm1 = 0;
                            uint2 nb0 = make_uint2(0, 0), nb1 = nb0;
                            if (ci >= 0) {
                                uint32_t w1, w2;
                                halo(pk_prev, q0, w1, w2);
                                nh0 = sample_hash<S, QC>(pk_prev, w1, w2, 0, mask_lo, mask_hi);
                                nb0 = gbloom[gbloom_block(nh0, gmask)];
                                if constexpr (NS > 1) {
                                    nh1 = sample_hash<S, QC>(pk_prev, w1, w2, 2 * S, mask_lo, mask_hi);
                                    nb1 = gbloom[gbloom_block(nh1, gmask)];
                                }
                                nwm1 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(pk_pp, 63), pk_prev, 0x138, 0xf, 0xf, false);
                            }
                            if (pd_on) test_pending();
                            pd_on = ci >= 0;
                            pd_h0 = nh0;
                            pd_h1 = nh1;
                            pd_b0 = nb0;
                            pd_b1 = nb1;
                            pd_cur = pk_prev;
                            pd_nxt = q0;
                            pd_wm1 = nwm1;
                            pd_cpos = base + (uint64_t)(ci < 0 ? 0 : ci) * kChunkBytes;
                        } else if (ci >= 0) {
                            uint32_t h0 = 0, h1 = 0;
                            const uint32_t ck = filter_chunk(pk_prev, q0, h0, h1);
                            if constexpr ((MK_ABLATE & 4) != 0) n_cand += ck;
                            if (__ballot(ck != 0)) {
                                uint32_t wm1 = 0;
                                if constexpr (kCtx)
                                    wm1 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(pk_pp, 63), pk_prev, 0x138, 0xf, 0xf, false);
                                queue_candidates(pk_prev, q0, wm1, ck, base + (uint64_t)ci * kChunkBytes, h0, h1);
                            }
                        }
                        if constexpr (kCtx) pk_pp = ci < 0 ? pk_tile_prev : pk_prev;  // the chunk before the next pair's
                        pk_prev = q0;
                        q0 = q1;
                        q1 = q2;
                        q2 = p3;
                    }
                }
                if constexpr ((MK_ABLATE & 8) != 0) {  // hand-off cost only: forget the queued entries
                    if (q_count >= MK_ISSUE_AT) {
                        q_head = (q_head + q_count) & (kRingEntries - 1);
                        q_count = 0;
                    }
                } else if (!pend_on && q_count >= (FL == 0 ? MK_ISSUE_AT_DENSE : MK_ISSUE_AT)) {
                    issue_probe(q_count < 64 ? q_count : 64);
                }
                if constexpr (FL == 0 && MK_DRAIN_AT_DENSE <= 64) {
                    if (hr.count >= MK_DRAIN_AT_DENSE) drain_hits<EMIT, FL>(P, hr, newest_end, lane, n_true);
                }
                pk_prev = p3;
            }
            if (++sc_run == run) {  // the same sequence as the loader cursor's
                sc_run = 0;
                tile += jump;
            } else {
                tile += 1;
            }
        }
        if constexpr (kPipe) {
            if (pd_on) test_pending();
        }
    }
    if (pend_on) consume_probe();

    // ---- tail phase: the < 32 KiB behind the last main tile, with guarded loads.  One chunk per
    // wave, dealt to the waves that come after the last main tile in the round-robin (they have
    // one tile less than the others): a single wave walking all <= 32 chunks exposes one memory
    // round trip per chunk and finishes up to 60 us after everybody else (r02: 10 M x 150 bp
    // batches ran at 0.65 of the roofline mostly because of it).
    {
        const uint64_t tail_rank = (wave_id + n_waves - (n_main_tiles / (P.tile_run ? P.tile_run : 1u)) % n_waves) % n_waves;
        for (uint64_t cpos = n_main_tiles * kTileBytes + tail_rank * kChunkBytes; cpos < n_bytes; cpos += n_waves * kChunkBytes) {
            const uint32_t pk_cur = pack16(load16(seq, cpos + lane * 16, n_bytes));
            const uint32_t pk_nxt = pack16(load16(seq, cpos + kChunkBytes + lane * 16, n_bytes));
            uint32_t h0 = 0, h1 = 0;
            const uint32_t ck = filter_chunk(pk_cur, pk_nxt, h0, h1);
            if constexpr ((MK_ABLATE & 4) != 0) n_cand += ck;
            if (__ballot(ck != 0)) {
                uint32_t wm1 = 0;
                if constexpr (kCtx) {
                    const uint64_t mine = cpos + lane * 16;
                    if (mine >= 16) wm1 = pack16(load16(seq, mine - 16, n_bytes));
                }
                queue_candidates(pk_cur, pk_nxt, wm1, ck, cpos, h0, h1);
            }
        }
    }

    // drain what is left in this wave's slots and rings
    flush_slots();
    while (q_count) drain_ring(q_count < 64 ? q_count : 64);
    if (hr.count) drain_hits<EMIT, FL>(P, hr, newest_end, lane, n_true);
    if constexpr (EMIT) flush_stage<!kWide>(P, hr, lane);
    if (hr.flist && lane == 0) P.flag_counts[wave_id] = hr.nflag;  // every wave of the grid, empty lists too
    if ((MK_ABLATE & 1) != 0 && abl_acc == 0xFFFFFFFFu) n_cand++;
    if (P.counters) {
        // One global atomic per WORKGROUP and counter: a single address retires an atomic every ~11 ns,
        // so one per wave (4096 x 2 on one cache line) kept every launch alive for 80 us after its
        // last wave had finished -- a quarter of the kernel time of a 1.5 GB batch (r02_small_ablate).
        __syncthreads();  // every wave has drained its rings: wave 0's becomes the sums
        if (threadIdx.x < kLdsSumWords) lds_sums[threadIdx.x] = 0;
        __syncthreads();
        if (lane == 0 && n_cand) atomicAdd(&lds_sums[0], n_cand);
        if (n_true) atomicAdd(&lds_sums[1], n_true);  // per-wave totals stay far below 2^32
        __syncthreads();
        if (threadIdx.x == 0 && lds_sums[0]) atomicAdd(&P.counters[P.n_pat + MK_SUM_CANDIDATES], (unsigned long long)lds_sums[0]);
        if (threadIdx.x == 1 && lds_sums[1]) atomicAdd(&P.counters[P.n_pat + MK_SUM_HITS], (unsigned long long)lds_sums[1]);
    }
}

}  // namespace mk
