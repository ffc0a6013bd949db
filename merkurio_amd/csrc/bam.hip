// bam.hip -- a window of inflated BAM text on the device (r05): the record chain becomes a
// record table, the 4-bit sequences become the ASCII text the matcher scans, and the kept records leave with their
// `km` tag appended -- what the reference's reader thread, record loop and writer do one record at a time
// (src/cmd_tag.rs:503-615 `for record in reader`, :387-497 process_record, bam 0.1.4's record layout = SAM
// specification §4.2) -- so that between inflate and deflate the records never visit the host.
//
// BAM records are a CHAIN: record k + 1 starts 4 + block_size bytes behind record k, which only a serial walk can
// follow.  The chain is cut into pieces of kPiece bytes of text:
//   mk_bam_find_kernel   a wave per piece looks for the first position at or behind the piece's first byte where four
//                        records in a row have consistent fixed fields (sizes that add up, a printable NUL-terminated
//                        name, reference ids and positions >= -1) -- a GUESS;
//   mk_bam_walk_kernel   a lane per piece walks from its start to the first record start at or behind the next piece's
//                        first byte (counting, then -- with the pieces' record counts summed -- writing the table);
// and the guesses are PROVED by the host from two small arrays: piece 0 starts at a record start by contract, and if
// every walk lands exactly on the next piece's start, every start is one (induction) and the table is the serial
// walk's.  A start that is not met is replaced by the landing and the walks run again (a false guess costs a round; no
// guess survives the check).  The walk applies the checks of the CLI's serial parser (cli/io.cpp:
// parse_bam_records_serial); a record that fails them raises a status bit and the caller's host reader takes over, which
// also words the error.
#include <algorithm>

#include "scan_kernel.h"

namespace mk {

constexpr uint32_t kBamNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t bam_ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t bam_ld16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
__device__ __forceinline__ void bam_st32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void bam_st64(uint8_t *p, unsigned long long v) { __builtin_memcpy(p, &v, 8); }

// 4 + block_size if the bytes at p look like a record that lies inside text[0, n); 0: they do not; -1: they might, but the
// text ends first
__device__ __forceinline__ long long bam_plausible(const uint8_t *__restrict__ text, uint64_t n, uint64_t p) {
    if (n - p < 36) return -1;
    const uint8_t *r = text + p;
    const int32_t block = (int32_t)bam_ld32(r), ref = (int32_t)bam_ld32(r + 4), pos = (int32_t)bam_ld32(r + 8), l_seq = (int32_t)bam_ld32(r + 20),
                  nref = (int32_t)bam_ld32(r + 24), npos = (int32_t)bam_ld32(r + 28);
    if (block < 32 || block > (1 << 28) || ref < -1 || ref > (1 << 24) || pos < -1 || l_seq < 0 || nref < -1 || nref > (1 << 24) || npos < -1) return 0;
    const uint32_t l_name = r[12], n_cig = bam_ld16(r + 16);
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (l_name == 0 || fixed > (uint64_t)block) return 0;
    if (n - p - 4 < (uint64_t)block) return -1;
    if (r[36 + l_name - 1] != 0) return 0;
    for (uint32_t k = 0; k + 1 < l_name; ++k)
        if (r[36 + k] < 33 || r[36 + k] > 126) return 0;
    return 4 + (long long)block;
}

// start[p] (p >= 1) = the first position in [p * piece, min((p + 1) * piece, n)) from which four plausible records follow each
// other (or fewer, if the chain reaches the end of the text), kBamNone if there is none; start[0] = 0.  One wave per piece.
__global__ __launch_bounds__(64) void mk_bam_find_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t piece, uint32_t n_pieces,
                                                        uint32_t *__restrict__ start) {
    const uint32_t p = blockIdx.x;
    if (p >= n_pieces) return;
    if (p == 0) {
        if (threadIdx.x == 0) start[0] = 0;
        return;
    }
    const uint64_t b = (uint64_t)p * piece, e = min(b + piece, n);
    uint32_t found = kBamNone;
    for (uint64_t base = b; base < e; base += 64) {
        uint64_t q = base + threadIdx.x;
        bool ok = q < e;
        if (ok) {
            int k = 0;
            for (; k < 4; ++k) {
                const long long sz = bam_plausible(text, n, q);
                if (sz == 0) {
                    ok = false;
                    break;
                }
                if (sz < 0) {  // the text ends inside this record: nothing more can be checked
                    ok = k > 0;
                    break;
                }
                q += (uint64_t)sz;
                if (q == n) break;
            }
        }
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            found = (uint32_t)(base + (uint64_t)__ffsll((long long)mask) - 1);
            break;
        }
    }
    if (threadIdx.x == 0) start[p] = found;
}

// One lane per piece: from start[p] along the chain to the first record start at or behind the next piece's first byte (the
// end of the whole records for the last piece) -> land[p], records passed -> count[p].  EMIT: the records' table entries at
// base[p] + k; checks as parse_bam_records_serial (st[0] |= 1: not a record), st[1] / st[2] = smallest / largest l_seq.
template <bool EMIT>
__global__ __launch_bounds__(64) void mk_bam_walk_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t piece, uint32_t n_pieces,
                                                        const uint32_t *__restrict__ start, uint32_t *__restrict__ land, uint32_t *__restrict__ count,
                                                        const uint32_t *__restrict__ base, uint32_t *__restrict__ rec_off, uint32_t *__restrict__ rec_len,
                                                        uint32_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len, uint32_t *__restrict__ st) {
    const uint32_t p = blockIdx.x * 64 + threadIdx.x;
    uint32_t mn = 0xFFFFFFFFu, mx = 0, bad = 0;
    if (p < n_pieces) {
        uint64_t pos = start[p];
        const uint64_t stop = p + 1 < n_pieces ? (uint64_t)(p + 1) * piece : n;
        uint32_t cnt = 0, unfinished = 0;
        const uint32_t k0 = EMIT ? base[p] : 0;
        if (pos != kBamNone) {
            while (pos < stop) {
                if (n - pos < 4) {  // an unfinished record: the window's tail
                    unfinished = pos < n;
                    break;
                }
                const uint8_t *r = text + pos;
                const int32_t block = (int32_t)bam_ld32(r);
                if (block < 32) {
                    bad = 1;
                    break;
                }
                if (n - pos - 4 < (uint64_t)block) {
                    unfinished = 1;
                    break;
                }
                const uint32_t l_name = r[12], n_cig = bam_ld16(r + 16);
                const int32_t l_seq = (int32_t)bam_ld32(r + 20);
                const uint64_t fixed = 32ull + l_name + 4ull * n_cig;
                if (l_seq < 0 || fixed + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)block) {
                    bad = 1;
                    break;
                }
                if (EMIT) {
                    rec_off[k0 + cnt] = (uint32_t)pos;
                    rec_len[k0 + cnt] = (uint32_t)block + 4;
                    seq_start[k0 + cnt] = (uint32_t)(pos + 4 + fixed);
                    seq_len[k0 + cnt] = (uint32_t)l_seq;
                    mn = min(mn, (uint32_t)l_seq), mx = max(mx, (uint32_t)l_seq);
                }
                ++cnt;
                pos += 4 + (uint64_t)block;
            }
        }
        if (!EMIT) {
            land[p] = (uint32_t)pos;
            count[p] = cnt | unfinished << 31;  // (bit 31: the walk ended at a record the text does not hold to its end)
        }
    }
    if (EMIT) {
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_down(mn, o));
            mx = max(mx, (uint32_t)__shfl_down(mx, o));
        }
        if (__ballot(bad != 0) && threadIdx.x == 0) atomicOr(&st[0], 1u);
        if (threadIdx.x == 0) {
            atomicMin(&st[1], mn);
            atomicMax(&st[2], mx);
        }
    }
}

// record i's packed sequence (two bases per byte, high nibble first, "=ACMGRSVTWYHKDBN") -> the ASCII the matcher sees
// (record.sequence().to_vec(), src/cmd_tag.rs:395) at seq[fixed_len ? i * fixed_len : off[i]]; 16 lanes per record, four packed
// bytes = eight bases per lane and step
__global__ __launch_bounds__(256) void mk_bam_unpack_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ seq_start,
                                                           const uint32_t *__restrict__ seq_len, const unsigned long long *__restrict__ off, uint32_t fixed_len,
                                                           uint64_t n_rec, uint8_t *__restrict__ seq) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (i >= n_rec) return;
    const uint32_t len = seq_len[i];
    const uint8_t *__restrict__ src = text + seq_start[i];
    uint8_t *__restrict__ dst = seq + (fixed_len ? i * (uint64_t)fixed_len : off[i]);
    const unsigned long long tab_lo = 0x565352474D43413Dull;  // bytes 0..7  = '=' 'A' 'C' 'M' 'G' 'R' 'S' 'V'
    const unsigned long long tab_hi = 0x4E42444B48595754ull;  // bytes 8..15 = 'T' 'W' 'Y' 'H' 'K' 'D' 'B' 'N'
    for (uint32_t b0 = 8 * sub; b0 < len; b0 += 128) {  // bases [b0, b0 + 8) = packed bytes [b0 / 2, b0 / 2 + 4)
        const uint32_t w = bam_ld32(src + (b0 >> 1));   // (may read up to three bytes of the qualities behind the sequence)
        unsigned long long out = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t byte = (w >> (8 * (k >> 1))) & 0xFFu;
            const uint32_t nib = (k & 1) ? (byte & 15u) : (byte >> 4);
            const unsigned long long c = ((nib < 8 ? tab_lo : tab_hi) >> (8 * (nib & 7))) & 0xFFull;
            out |= c << (8 * k);
        }
        if (b0 + 8 <= len) {
            bam_st64(dst + b0, out);
        } else {
            for (uint32_t k = 0; b0 + k < len; ++k) dst[b0 + k] = (uint8_t)(out >> (8 * k));
        }
    }
}

// ---- the value of a record that already carries the tag (src/cmd_tag.rs:470-485): the found patterns and the ','-separated items of
// the existing Z value, sort_unstable + dedup (Rust's String order = bytewise), joined by ','.  Both kernels below enumerate that
// merge the same way: "the smallest item that is greater than the previous one", found by a scan over both lists -- quadratic in the
// number of items, which is a handful (values above kBamMergeBytes are left to the host path).
constexpr uint32_t kBamMergeBytes = 2048;

__device__ __forceinline__ int bam_cmp(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb) {
    const uint32_t m = min(na, nb);
    for (uint32_t k = 0; k < m; ++k)
        if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
    return na < nb ? -1 : na > nb ? 1 : 0;
}
// *best = the smallest item > prev (have_prev == false: the smallest of all); false: there is none
__device__ bool bam_merge_next(const uint8_t *__restrict__ ex, uint32_t nex, unsigned long long f0, unsigned long long f1, const uint32_t *__restrict__ found_pat,
                               const uint8_t *__restrict__ pat_bytes, const uint32_t *__restrict__ pat_off, bool have_prev, const uint8_t *prev, uint32_t nprev,
                               const uint8_t **best, uint32_t *nbest) {
    bool have = false;
    const uint8_t *bp = nullptr;
    uint32_t bn = 0;
    auto offer = [&](const uint8_t *p, uint32_t n) {
        if (have_prev && bam_cmp(p, n, prev, nprev) <= 0) return;
        if (!have || bam_cmp(p, n, bp, bn) < 0) have = true, bp = p, bn = n;
    };
    for (unsigned long long f = f0; f < f1; ++f) {
        const uint32_t pt = found_pat[f];
        offer(pat_bytes + pat_off[pt], pat_off[pt + 1] - pat_off[pt]);
    }
    for (uint32_t a = 0; a <= nex;) {  // (an empty value has been taken for "no tag" before: nex > 0; "a,,b" holds an empty item)
        uint32_t b = a;
        while (b < nex && ex[b] != ',') ++b;
        offer(ex + a, b - a);
        a = b + 1;
    }
    *best = bp, *nbest = bn;
    return have;
}

// Per record: keep or drop (src/cmd_tag.rs:457-467), and for a kept one the size of the record it leaves as --
// 4 + block_size + tag (2) + 'Z' + value + NUL, the value being its distinct matched patterns joined by ',' (:484-490; ascending
// pattern index = sort_unstable order of the sorted unique pattern list), merged with the record's existing Z value of that name if
// it has one (:470-485).  A kept record's optional fields are walked for that: fields that do not parse, a field of that name that is
// not a string (the reference refuses it), or a value that is not plain ASCII or very long set a status bit -- the caller's host
// path then does this window, where those rules live.
__global__ __launch_bounds__(256) void mk_bam_taglen_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ rec_off,
                                                           const uint32_t *__restrict__ rec_len, const uint32_t *__restrict__ seq_start,
                                                           const uint32_t *__restrict__ seq_len, const unsigned long long *__restrict__ found_off,
                                                           const uint32_t *__restrict__ found_pat, const uint32_t *__restrict__ pat_off, uint64_t n_rec,
                                                           const uint8_t *__restrict__ pat_bytes, uint32_t filter_matching, uint32_t invert, uint32_t tag0,
                                                           uint32_t tag1, uint8_t *__restrict__ keep, uint32_t *__restrict__ out_len,
                                                           uint32_t *__restrict__ ex_off, uint32_t *__restrict__ st) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0;
    if (i < n_rec) {
        const unsigned long long f0 = found_off[i], f1 = found_off[i + 1];
        const bool has = f1 > f0;
        const bool kept = filter_matching ? has : (invert ? !has : true);
        uint32_t len = 0, ex_at = 0, ex_n = 0;
        if (kept) {
            const uint32_t l = seq_len[i];
            uint64_t p = (uint64_t)seq_start[i] + (l + 1) / 2 + l;
            const uint64_t e = (uint64_t)rec_off[i] + rec_len[i];
            while (p < e) {
                if (e - p < 3) {
                    bad |= 2;
                    break;
                }
                const uint32_t t0 = text[p], t1 = text[p + 1], type = text[p + 2];
                const bool mine = t0 == tag0 && t1 == tag1 && ex_at == 0 && !(bad & 4);  // (record.tags().get(): the first field of that name)
                if (mine && type != 'Z') bad |= 4;  // the reference refuses it: the host path words that
                p += 3;
                if (type == 'A' || type == 'c' || type == 'C') {
                    p += 1;
                } else if (type == 's' || type == 'S') {
                    p += 2;
                } else if (type == 'i' || type == 'I' || type == 'f') {
                    p += 4;
                } else if (type == 'Z' || type == 'H') {
                    const uint64_t v0 = p;
                    uint32_t high = 0;
                    while (p < e && text[p]) high |= text[p], ++p;
                    if (mine && type == 'Z') {
                        // (not plain ASCII: the reference checks UTF-8 first -- the host path's business; so is a very long value)
                        if ((high & 0x80u) || p - v0 > kBamMergeBytes) bad |= 4;
                        else ex_at = (uint32_t)v0, ex_n = (uint32_t)(p - v0);
                    }
                    ++p;
                } else if (type == 'B') {
                    if (e - p < 5) {
                        bad |= 2;
                        break;
                    }
                    const uint32_t sub = text[p];
                    const int32_t cnt = (int32_t)bam_ld32(text + p + 1);
                    const uint64_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if (cnt < 0) {
                        bad |= 2;
                        break;
                    }
                    p += 5 + w * (uint64_t)cnt;
                } else {
                    bad |= 2;
                    break;
                }
            }
            if (p > e) bad |= 2;
            uint32_t vlen = 0;
            if (ex_n == 0) {  // no tag of that name, or an empty value ("do nothing if tag is empty", :472-473)
                ex_at = 0;
                for (unsigned long long k = f0; k < f1; ++k) {
                    const uint32_t pt = found_pat[k];
                    vlen += pat_off[pt + 1] - pat_off[pt];
                }
                if (has) vlen += (uint32_t)(f1 - f0) - 1;
            } else if (!bad) {
                const uint8_t *prev = nullptr, *it;
                uint32_t nprev = 0, nit, items = 0;
                while (bam_merge_next(text + ex_at, ex_n, f0, f1, found_pat, pat_bytes, pat_off, items != 0, prev, nprev, &it, &nit))
                    vlen += nit, prev = it, nprev = nit, ++items;
                vlen += items - 1;
            }
            len = rec_len[i] + 3 + vlen + 1;
        }
        keep[i] = kept ? 1 : 0;
        out_len[i] = len;
        ex_off[i] = ex_at;
    }
    if (__ballot(bad != 0)) {
        uint32_t all = bad;
        for (int o = 32; o > 0; o >>= 1) all |= (uint32_t)__shfl_down(all, o);
        if ((threadIdx.x & 63) == 0) atomicOr(&st[0], all);
    }
}

// the kept records with their tag appended, back to back at out + out_off[i] (record.tags_mut().push_string, src/cmd_tag.rs:488-490;
// the bytes BamWriter::append_tagged_raw of the CLI's host path produces); 16 lanes per record copy four bytes each and step
__global__ __launch_bounds__(256) void mk_bam_emit_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ rec_off,
                                                         const uint32_t *__restrict__ rec_len, const uint32_t *__restrict__ out_len,
                                                         const unsigned long long *__restrict__ out_off, const unsigned long long *__restrict__ found_off,
                                                         const uint32_t *__restrict__ found_pat, const uint8_t *__restrict__ pat_bytes,
                                                         const uint32_t *__restrict__ pat_off, const uint32_t *__restrict__ ex_off, uint64_t n_rec,
                                                         uint32_t tag0, uint32_t tag1, uint8_t *__restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (i >= n_rec) return;
    const uint32_t olen = out_len[i];
    if (!olen) return;
    const uint8_t *__restrict__ src = text + rec_off[i];
    uint8_t *__restrict__ dst = out + out_off[i];
    const uint32_t L = rec_len[i];
    uint32_t k = 4 + 4 * sub;
    for (; k + 4 <= L; k += 64) bam_st32(dst + k, bam_ld32(src + k));
    // (the last lane's step may leave up to three bytes: whoever's k is the first not to fit copies them)
    if (k < L && k + 4 > L)
        for (uint32_t j = k; j < L; ++j) dst[j] = src[j];
    if (sub == 0) {
        bam_st32(dst, olen - 4);
        uint8_t *t = dst + L;
        t[0] = (uint8_t)tag0, t[1] = (uint8_t)tag1, t[2] = 'Z';
        t += 3;
        const unsigned long long f0 = found_off[i], f1 = found_off[i + 1];
        const uint32_t ex_at = ex_off[i];
        if (ex_at == 0) {
            for (unsigned long long f = f0; f < f1; ++f) {
                if (f > f0) *t++ = ',';
                const uint32_t pt = found_pat[f];
                const uint32_t a = pat_off[pt], b = pat_off[pt + 1];
                for (uint32_t j = a; j < b; ++j) *t++ = pat_bytes[j];
            }
        } else {  // merged with the record's existing value (the old field stays where it is: push_string appends, :488-490)
            uint32_t ex_n = 0;
            while (text[ex_at + ex_n]) ++ex_n;
            const uint8_t *prev = nullptr, *it;
            uint32_t nprev = 0, nit, items = 0;
            while (bam_merge_next(text + ex_at, ex_n, f0, f1, found_pat, pat_bytes, pat_off, items != 0, prev, nprev, &it, &nit)) {
                if (items) *t++ = ',';
                for (uint32_t j = 0; j < nit; ++j) *t++ = it[j];
                prev = it, nprev = nit, ++items;
            }
        }
        *t = 0;
    }
}

// names (NUL included) of the records with a hit, for the log rows: name_len[i] = flags[i] ? l_read_name : 0 and where the name starts
__global__ __launch_bounds__(256) void mk_bam_names_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ rec_off,
                                                          const uint8_t *__restrict__ flags, uint64_t n_rec, uint32_t *__restrict__ name_start,
                                                          uint32_t *__restrict__ name_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    name_start[i] = rec_off[i] + 36;
    name_len[i] = flags[i] ? (uint32_t)text[rec_off[i] + 12] : 0u;
}

// ---- launchers ----------------------------------------------------------------------------------------------------------
void launch_bam_find(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, uint32_t *d_start, hipStream_t st) {
    hipLaunchKernelGGL(mk_bam_find_kernel, dim3(n_pieces), dim3(64), 0, st, d_text, n, piece, n_pieces, d_start);
}
void launch_bam_walk_count(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, const uint32_t *d_start, uint32_t *d_land,
                           uint32_t *d_count, hipStream_t st) {
    hipLaunchKernelGGL(mk_bam_walk_kernel<false>, dim3((n_pieces + 63) / 64), dim3(64), 0, st, d_text, n, piece, n_pieces, d_start, d_land, d_count,
                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
}
void launch_bam_walk_emit(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, const uint32_t *d_start, const uint32_t *d_base,
                          uint32_t *d_rec_off, uint32_t *d_rec_len, uint32_t *d_seq_start, uint32_t *d_seq_len, uint32_t *d_st, hipStream_t st) {
    hipLaunchKernelGGL(mk_bam_walk_kernel<true>, dim3((n_pieces + 63) / 64), dim3(64), 0, st, d_text, n, piece, n_pieces, d_start, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, d_base, d_rec_off, d_rec_len, d_seq_start, d_seq_len, d_st);
}
void launch_bam_unpack(const uint8_t *d_text, const uint32_t *d_seq_start, const uint32_t *d_seq_len, const unsigned long long *d_off, uint32_t fixed_len,
                       uint64_t n_rec, uint8_t *d_seq, hipStream_t st) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_bam_unpack_kernel, dim3((unsigned)((n_rec * 16 + 255) / 256)), dim3(256), 0, st, d_text, d_seq_start, d_seq_len, d_off, fixed_len,
                       n_rec, d_seq);
}
void launch_bam_taglen(const uint8_t *d_text, const uint32_t *d_rec_off, const uint32_t *d_rec_len, const uint32_t *d_seq_start, const uint32_t *d_seq_len,
                       const unsigned long long *d_found_off, const uint32_t *d_found_pat, const uint32_t *d_pat_off, const uint8_t *d_pat_bytes, uint64_t n_rec,
                       uint32_t filter_matching, uint32_t invert, uint32_t tag0, uint32_t tag1, uint8_t *d_keep, uint32_t *d_out_len, uint32_t *d_ex_off,
                       uint32_t *d_st, hipStream_t st) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_bam_taglen_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, d_text, d_rec_off, d_rec_len, d_seq_start, d_seq_len,
                       d_found_off, d_found_pat, d_pat_off, n_rec, d_pat_bytes, filter_matching, invert, tag0, tag1, d_keep, d_out_len, d_ex_off, d_st);
}
void launch_bam_emit(const uint8_t *d_text, const uint32_t *d_rec_off, const uint32_t *d_rec_len, const uint32_t *d_out_len, const unsigned long long *d_out_off,
                     const unsigned long long *d_found_off, const uint32_t *d_found_pat, const uint8_t *d_pat_bytes, const uint32_t *d_pat_off,
                     const uint32_t *d_ex_off, uint64_t n_rec, uint32_t tag0, uint32_t tag1, uint8_t *d_out, hipStream_t st) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_bam_emit_kernel, dim3((unsigned)((n_rec * 16 + 255) / 256)), dim3(256), 0, st, d_text, d_rec_off, d_rec_len, d_out_len, d_out_off,
                       d_found_off, d_found_pat, d_pat_bytes, d_pat_off, d_ex_off, n_rec, tag0, tag1, d_out);
}
void launch_bam_names(const uint8_t *d_text, const uint32_t *d_rec_off, const uint8_t *d_flags, uint64_t n_rec, uint32_t *d_name_start, uint32_t *d_name_len,
                      hipStream_t st) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_bam_names_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, d_text, d_rec_off, d_flags, n_rec, d_name_start,
                       d_name_len);
}

}  // namespace mk
