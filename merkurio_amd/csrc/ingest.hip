// ingest.hip -- FASTQ text -> record table + concatenated sequences, on the device (the step before
// the hot path; the reference gets its records from needletail's parse_fastx_file, src/cmd_extract.rs:281-282,321-328).
//
// The host used to index every record (four memchr per record on all threads), gather the sequence lines -- 47 % of a
// FASTQ's bytes -- into a batch buffer and upload that.  Here the RAW text of a window goes to the device as it is and
// five small kernels do the rest at HBM speed (a 1 GiB window: < 2 ms):
//   1. newlines per 16 KiB block            2. exclusive scan of the block counts
//   3. line starts (u32, one per line)      4. one lane per record: its four lines validated, sequence length
//   5. exclusive scan of the lengths (skipped when all reads have one length)
//   6. sequence lines copied into the scan buffer
// Only plain 4-line FASTQ is taken: '@' header, sequence, '+' line, quality of the sequence's length, '\n' or '\r\n'
// line ends, no blank lines.  Anything else (FASTA, wrapped or truncated records, blank lines) raises a status word and
// the caller parses that window with its own reader -- which also produces the reference's error messages -- so the
// device never has to decide what a malformed record means.  '@' as the first quality character is not a problem
// here: lines are counted, not guessed (the window starts at a record start).
#include <algorithm>

#include "scan_kernel.h"

namespace mk {

constexpr uint32_t kIngestThreads = 256;
constexpr uint32_t kIngestBytesPerThread = 64;
constexpr uint32_t kIngestBlockBytes = kIngestThreads * kIngestBytesPerThread;  // 16 KiB

// 0x80 in every byte of v that equals '\n' (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t nl_mask(uint32_t v) {
    const uint32_t x = v ^ 0x0A0A0A0Au;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// the 64 bytes of a thread as 16 dwords (bytes at or beyond n read as 0)
__device__ __forceinline__ void load64(const uint8_t *__restrict__ text, uint64_t pos, uint64_t n, uint32_t w[16]) {
    if (pos + 64 <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(text + pos);  // text is 16-byte aligned, pos a multiple of 64
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = p[k];
            w[4 * k] = v.x, w[4 * k + 1] = v.y, w[4 * k + 2] = v.z, w[4 * k + 3] = v.w;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    for (uint64_t i = pos; i < n; ++i) w[(i - pos) >> 2] |= (uint32_t)text[i] << (8 * ((i - pos) & 3));
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds) {  // 256 threads; every thread gets the total
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kIngestThreads) void mk_ingest_count_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ block_cnt) {
    __shared__ uint32_t lds[4];
    const uint64_t pos = (uint64_t)blockIdx.x * kIngestBlockBytes + threadIdx.x * kIngestBytesPerThread;
    uint32_t c = 0;
    if (pos < n) {
        uint32_t w[16];
        load64(text, pos, n, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) c += __popc(nl_mask(w[k]));
    }
    const uint32_t t = block_sum(c, lds);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = t;
}

// in place: cnt[b] -> newlines in front of block b; *total = all of them (one workgroup)
__global__ __launch_bounds__(1024) void mk_ingest_scan_blocks_kernel(uint32_t *__restrict__ cnt, uint32_t n_blocks, uint32_t *__restrict__ total) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? cnt[i] : 0;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += u;
        }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_sum[w];
        if (i < n_blocks) cnt[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// line_start[k] = first byte of line k (line 0 starts at 0; a text that ends in '\n' has an empty last "line" at n);
// one more entry behind the last: n + 1, the virtual newline that ends a text without a final '\n'
__global__ __launch_bounds__(kIngestThreads) void mk_ingest_lines_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ block_off,
                                                                         const uint32_t *__restrict__ total, uint32_t *__restrict__ line_start) {
    __shared__ uint32_t wave_sum[4];
    const uint64_t pos = (uint64_t)blockIdx.x * kIngestBlockBytes + threadIdx.x * kIngestBytesPerThread;
    uint32_t w[16];
    uint32_t c = 0;
    if (pos < n) {
        load64(text, pos, n, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) c += __popc(nl_mask(w[k]));
    }
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += u;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t k = block_off[blockIdx.x] + incl - c + 1;  // index of the line that starts behind this thread's first newline
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) k += wave_sum[wv];
    if (c) {
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            uint32_t mk_ = nl_mask(w[d]);
            while (mk_) {
                const uint32_t b = (uint32_t)__ffs(mk_) - 1u;  // bit 7 of the byte
                mk_ &= mk_ - 1;
                line_start[k++] = (uint32_t)(pos + 4 * d + (b >> 3) + 1);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        line_start[0] = 0;
        line_start[*total + 1] = (uint32_t)(n + 1);
    }
}

// st[0] status bits, st[1] smallest, st[2] largest sequence length
__global__ __launch_bounds__(256) void mk_ingest_records_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ line_start,
                                                               uint64_t n_rec, uint32_t *__restrict__ rec_start, uint32_t *__restrict__ seq_start,
                                                               uint32_t *__restrict__ seq_len, uint32_t *__restrict__ st) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad = 0, len = 0;
    // rec_start[n_rec]: the first byte behind the last whole record (= n when the text ends with that record)
    if (i == n_rec) rec_start[n_rec] = min(line_start[4 * n_rec], (uint32_t)n);
    bool live = i < n_rec;
    if (live) {
        const uint32_t l0 = line_start[4 * i], l1 = line_start[4 * i + 1], l2 = line_start[4 * i + 2], l3 = line_start[4 * i + 3],
                       l4 = line_start[4 * i + 4];
        // line k is [l_k, l_{k+1} - 1) without its '\n'; a '\r' in front of the '\n' belongs to the line end
        uint32_t e1 = l2 - 1, e3 = l4 - 1;
        if (e1 > l1 && text[e1 - 1] == '\r') --e1;
        if (e3 > l3 && e3 - 1 < n && text[e3 - 1] == '\r') --e3;
        len = e1 - l1;
        bad |= text[l0] != '@';
        bad |= l1 - l0 < 2;  // "@\n": no id at all is left to the host reader
        bad |= (l3 - l2 < 2) || text[l2] != '+';
        bad |= (e3 - l3) != len;
        rec_start[i] = l0;
        seq_start[i] = l1;
        seq_len[i] = len;
    }
    // per-wave reductions, one atomic each
    uint32_t mn = live ? len : 0xFFFFFFFFu, mx = live ? len : 0u;
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_down(mn, o));
        mx = max(mx, (uint32_t)__shfl_down(mx, o));
    }
    if (__ballot(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&st[0], 1u);
    // (47 000 waves of a 3 M-record window, two words: the atomics only where they would change something -- reads of one length
    // stop issuing them after the first waves; the plain loads may be stale, which only costs an atomic that changes nothing)
    if ((threadIdx.x & 63) == 0) {
        if (mn < __atomic_load_n(&st[1], __ATOMIC_RELAXED)) atomicMin(&st[1], mn);
        if (mx > __atomic_load_n(&st[2], __ATOMIC_RELAXED)) atomicMax(&st[2], mx);
    }
}

// ---- exclusive scan of u32 lengths into u64 offsets (tiles of 4096) -----------------------------------------------
constexpr uint32_t kScanTile = 4096;
__global__ __launch_bounds__(256) void mk_ingest_tile_sums_kernel(const uint32_t *__restrict__ len, uint64_t n, unsigned long long *__restrict__ tile_sum) {
    __shared__ unsigned long long part[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    unsigned long long s = 0;
    for (uint32_t k = threadIdx.x; k < kScanTile; k += 256)
        if (base + k < n) s += len[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void mk_ingest_scan_tiles_kernel(unsigned long long *__restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_tiles ? tile_sum[i] : 0;
        unsigned long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += u;
        }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_sum[w];
        if (i < n_tiles) tile_sum[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
}
// off[i] = sum of len[0, i) for i in [0, n]; one workgroup per tile, 16 elements per thread
__global__ __launch_bounds__(256) void mk_ingest_offsets_kernel(const uint32_t *__restrict__ len, uint64_t n, const unsigned long long *__restrict__ tile_base,
                                                               unsigned long long *__restrict__ off) {
    __shared__ unsigned long long wave_sum[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + threadIdx.x * 16;
    uint32_t v[16];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = base + k < n ? len[base + k] : 0;
        s += v[k];
    }
    unsigned long long incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += u;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long run = tile_base[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wave_sum[w];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (base + k <= n) off[base + k] = run;  // (index n: the total)
        run += v[k];
    }
}

// sequence line of record i -> seq[dst, dst + len): 16 lanes per record, dwords
__global__ __launch_bounds__(256) void mk_ingest_gather_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ seq_start,
                                                              const uint32_t *__restrict__ seq_len, const unsigned long long *__restrict__ off,
                                                              uint32_t fixed_len, uint64_t n_rec, uint8_t *__restrict__ seq, uint32_t skip_from) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (i >= n_rec) return;
    const uint32_t len = seq_len[i];
    if (len >= skip_from) return;  // (a chromosome-sized record is not a job for 16 lanes: the caller copies those one by one)
    const uint8_t *__restrict__ src = text + seq_start[i];
    uint8_t *__restrict__ dst = seq + (fixed_len ? i * (uint64_t)fixed_len : off[i]);
    // four bytes per lane and step (unaligned dword accesses); the lane whose step is the first not to fit copies the last one to three bytes
    uint32_t k = 4 * sub;
    for (; k + 4 <= len; k += 64) {
        uint32_t v;
        __builtin_memcpy(&v, src + k, 4);
        __builtin_memcpy(dst + k, &v, 4);
    }
    if (k < len)
        for (uint32_t j = k; j < len; ++j) dst[j] = src[j];
}

// length of record i's text if the record is kept (hit != invert), else 0: the lengths the offsets scan and the gather
// kernel then pack the kept records with
__global__ __launch_bounds__(256) void mk_ingest_select_kernel(const uint8_t *__restrict__ flags, uint32_t invert, const uint32_t *__restrict__ rec_start,
                                                              uint64_t n_rec, uint32_t n_text, uint32_t *__restrict__ sel_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint32_t end = i + 1 < n_rec ? rec_start[i + 1] : n_text;
    sel_len[i] = ((flags[i] != 0) != (invert != 0)) ? end - rec_start[i] : 0u;
}

// ---- FASTA (r05): '>' header lines, every other line is sequence; record.seq() is the sequence lines without their '\n' and '\r'
// bytes (needletail strips both; cli/io.cpp: FastxFile::append_seq).  On the device that is a byte compaction of the window's text:
// a byte is kept iff it lies in a sequence line and is neither '\n' nor '\r'.  Two passes over the text, a thread per 64 bytes:
// count (kept bytes, header lines) per thread -> per block -> exclusive scan; then the same walk writes the kept bytes to their
// place in the scan buffer and, at every header line, the record's text offset and sequence offset.  The state "my first byte lies
// in a header line" comes from the line table (mk_ingest_lines_kernel): the line that holds the thread's first byte starts at
// line_start[newlines in front of it].  Lines of any length (an unwrapped chromosome is ONE line) cost nothing extra: work is per
// byte, not per line.
__device__ __forceinline__ uint32_t eq_mask(uint32_t v, uint32_t c4) {  // 0x80 in every byte of v equal to the byte replicated in c4
    const uint32_t x = v ^ c4;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// counts of one thread's 64 bytes: low 32 bits kept bytes, high 32 bits header lines that START in them.  EMIT: the bytes are also
// written to stage[0 ...] (the block's LDS staging area, at this thread's place in it) and the headers' record entries stored
// (goff = where the thread's first kept byte lies in the scan buffer).
template <bool EMIT>
__device__ __forceinline__ unsigned long long fa_walk(const uint32_t w[16], uint64_t pos, uint32_t nbytes, bool hdr, bool at_line_start,
                                                      uint8_t *__restrict__ stage, unsigned long long goff, uint32_t rec, uint32_t *__restrict__ rec_start,
                                                      unsigned long long *__restrict__ off) {
    uint32_t kept = 0, heads = 0;
    bool ls = at_line_start;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t i = 4u * d + b;
            const uint32_t c = (w[d] >> (8 * b)) & 0xFFu;
            if (i < nbytes) {
                if (ls) {
                    hdr = c == '>';
                    if (hdr) {
                        if (EMIT) {
                            rec_start[rec + heads] = (uint32_t)(pos + i);
                            off[rec + heads] = goff + kept;
                        }
                        ++heads;
                    }
                }
                const bool keep = !hdr && c != '\n' && c != '\r';
                if (keep) {
                    if (EMIT) stage[kept] = (uint8_t)c;
                    ++kept;
                }
                ls = c == '\n';
            }
        }
    }
    return (unsigned long long)heads << 32 | kept;
}

// newlines in front of this thread's first byte, within its block (exclusive), from the per-thread counts
__device__ __forceinline__ uint32_t block_excl_u32(uint32_t c, uint32_t *wave_sum) {
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += u;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) before += wave_sum[wv];
    __syncthreads();
    return before;
}
__device__ __forceinline__ unsigned long long block_excl_u64(unsigned long long c, unsigned long long *wave_sum, unsigned long long *total) {
    unsigned long long incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += u;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = incl - c;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) before += wave_sum[wv];
    if (total) *total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    __syncthreads();
    return before;
}

// EMIT = false: block_sum[b] = (header lines << 32 | kept bytes) of block b.  EMIT = true: block_sum holds the exclusive scan of
// those; the kept bytes of the block are compacted in LDS (every thread writes its own at its place in the block's output) and
// leave for seq in coalesced dword stores -- a thread's kept bytes are a run of up to 64 bytes at an arbitrary address: written
// straight to global memory, byte by byte and 64 bytes apart from its neighbour's, they cost 2.3 ms per 508 MB of text (0.44 TB/s);
// every header line writes rec_start / off of its record; the thread that holds the text's last byte closes the tables:
// rec_start[records] = n, off[records] = kept bytes.
template <bool EMIT>
__global__ __launch_bounds__(kIngestThreads) void mk_ingest_fa_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ nl_block_off,
                                                                      const uint32_t *__restrict__ line_start, unsigned long long *__restrict__ block_sum,
                                                                      uint8_t *__restrict__ seq, uint32_t *__restrict__ rec_start,
                                                                      unsigned long long *__restrict__ off) {
    __shared__ uint32_t wave_sum[4];
    __shared__ unsigned long long wave_sum64[4];
    __shared__ uint8_t stage[EMIT ? kIngestBlockBytes : 4];
    const uint64_t pos = (uint64_t)blockIdx.x * kIngestBlockBytes + threadIdx.x * kIngestBytesPerThread;
    uint32_t w[16];
    uint32_t c = 0, nbytes = 0;
    if (pos < n) {
        load64(text, pos, n, w);
        nbytes = (uint32_t)min((uint64_t)64, n - pos);
#pragma unroll
        for (int k = 0; k < 16; ++k) c += __popc(nl_mask(w[k]));
        // (bytes at or beyond n read as 0: never '\n')
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = 0;
    }
    const uint32_t line = nl_block_off[blockIdx.x] + block_excl_u32(c, wave_sum);  // the line that holds byte `pos`
    bool hdr = false, at_ls = false;
    if (pos < n) {
        const uint32_t ls = line_start[line];
        at_ls = ls == (uint32_t)pos;
        hdr = text[ls] == '>';
    }
    const unsigned long long mine = fa_walk<false>(w, pos, nbytes, hdr, at_ls, nullptr, 0, 0, nullptr, nullptr);
    unsigned long long total = 0;
    const unsigned long long before = block_excl_u64(mine, wave_sum64, &total);
    if (!EMIT) {
        if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
        return;
    }
    const unsigned long long block_base = block_sum[blockIdx.x], base = block_base + before;
    if (nbytes) fa_walk<true>(w, pos, nbytes, hdr, at_ls, stage + (uint32_t)(before & 0xFFFFFFFFull), base & 0xFFFFFFFFull, (uint32_t)(base >> 32), rec_start, off);
    if (pos < n && pos + 64 >= n) {
        const unsigned long long end = base + mine;
        rec_start[end >> 32] = (uint32_t)n;
        off[end >> 32] = end & 0xFFFFFFFFull;
    }
    __syncthreads();
    // the block's kept bytes, LDS -> seq: four bytes per thread and step (an unaligned dword store), the last few one by one
    const uint32_t out_n = (uint32_t)(total & 0xFFFFFFFFull);
    uint8_t *__restrict__ dst = seq + (block_base & 0xFFFFFFFFull);
    for (uint32_t k = 4 * threadIdx.x; k < out_n; k += 4 * kIngestThreads) {
        if (k + 4 <= out_n) {
            uint32_t v;
            __builtin_memcpy(&v, stage + k, 4);
            __builtin_memcpy(dst + k, &v, 4);
        } else {
            for (uint32_t j = k; j < out_n; ++j) dst[j] = stage[j];
        }
    }
}

// FASTA index + gather of a window whose line table exists (launch_ingest_count + mk_ingest_lines_kernel have run): d_block64 gets the
// scanned per-block sums, d_block64[n_blocks] the totals (header lines << 32 | sequence bytes); the caller reads them, then
// launch_ingest_fasta_emit fills seq / rec_start / off (records + 1 entries each)
void launch_ingest_fasta_count(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl_block_off, const uint32_t *d_line_start,
                               unsigned long long *d_block64, hipStream_t st) {
    const uint32_t n_blocks = (uint32_t)((n + kIngestBlockBytes - 1) / kIngestBlockBytes);
    hipLaunchKernelGGL(mk_ingest_fa_kernel<false>, dim3(n_blocks), dim3(kIngestThreads), 0, st, d_text, n, d_nl_block_off, d_line_start, d_block64,
                       (uint8_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    // exclusive scan in place over n_blocks + 1 entries (the last one, cleared by the caller, receives the total)
    hipLaunchKernelGGL(mk_ingest_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, d_block64, n_blocks + 1);
}
void launch_ingest_fasta_emit(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl_block_off, const uint32_t *d_line_start,
                              unsigned long long *d_block64, uint8_t *d_seq, uint32_t *d_rec_start, unsigned long long *d_off, hipStream_t st) {
    const uint32_t n_blocks = (uint32_t)((n + kIngestBlockBytes - 1) / kIngestBlockBytes);
    hipLaunchKernelGGL(mk_ingest_fa_kernel<true>, dim3(n_blocks), dim3(kIngestThreads), 0, st, d_text, n, d_nl_block_off, d_line_start, d_block64, d_seq,
                       d_rec_start, d_off);
}
// the line table alone (FASTA: no record kernel follows it)
void launch_ingest_lines(const uint8_t *d_text, uint64_t n, const uint32_t *d_block_off, const uint32_t *d_total, uint32_t *d_line_start, hipStream_t st) {
    const uint32_t n_blocks = (uint32_t)((n + kIngestBlockBytes - 1) / kIngestBlockBytes);
    hipLaunchKernelGGL(mk_ingest_lines_kernel, dim3(n_blocks), dim3(kIngestThreads), 0, st, d_text, n, d_block_off, d_total, d_line_start);
}

// flags |= other (paired windows: a pair is kept if either mate hits, src/cmd_extract.rs:600-606)
__global__ __launch_bounds__(256) void mk_ingest_or_flags_kernel(uint8_t *__restrict__ flags, const uint8_t *__restrict__ other, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] |= other[i];
}
void launch_ingest_or_flags(uint8_t *d_flags, const uint8_t *d_other, uint64_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(mk_ingest_or_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_flags, d_other, n);
}

void launch_ingest_select(const uint8_t *d_flags, uint32_t invert, const uint32_t *d_rec_start, uint64_t n_rec, uint32_t n_text, uint32_t *d_sel_len,
                          hipStream_t st) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_ingest_select_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, d_flags, invert, d_rec_start, n_rec, n_text,
                       d_sel_len);
}

void launch_ingest_count(const uint8_t *d_text, uint64_t n, uint32_t *d_block_cnt, uint32_t *d_total, hipStream_t st) {
    const uint32_t n_blocks = (uint32_t)((n + kIngestBlockBytes - 1) / kIngestBlockBytes);
    hipLaunchKernelGGL(mk_ingest_count_kernel, dim3(n_blocks), dim3(kIngestThreads), 0, st, d_text, n, d_block_cnt);
    hipLaunchKernelGGL(mk_ingest_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, d_block_cnt, n_blocks, d_total);
}

void launch_ingest_records(const uint8_t *d_text, uint64_t n, const uint32_t *d_block_off, const uint32_t *d_total, uint32_t *d_line_start,
                           uint64_t n_rec, uint32_t *d_rec_start, uint32_t *d_seq_start, uint32_t *d_seq_len, uint32_t *d_status, hipStream_t st) {
    const uint32_t n_blocks = (uint32_t)((n + kIngestBlockBytes - 1) / kIngestBlockBytes);
    hipLaunchKernelGGL(mk_ingest_lines_kernel, dim3(n_blocks), dim3(kIngestThreads), 0, st, d_text, n, d_block_off, d_total, d_line_start);
    // (n_rec + 1 lanes: the last one writes the end of the records)
    hipLaunchKernelGGL(mk_ingest_records_kernel, dim3((unsigned)((n_rec + 1 + 255) / 256)), dim3(256), 0, st, d_text, n, d_line_start, n_rec, d_rec_start,
                       d_seq_start, d_seq_len, d_status);
}

void launch_ingest_offsets(const uint32_t *d_seq_len, uint64_t n_rec, unsigned long long *d_tile, unsigned long long *d_off, hipStream_t st) {
    const uint32_t n_tiles = (uint32_t)(n_rec / kScanTile + 1);  // (covers index n_rec as well)
    hipLaunchKernelGGL(mk_ingest_tile_sums_kernel, dim3(n_tiles), dim3(256), 0, st, d_seq_len, n_rec, d_tile);
    hipLaunchKernelGGL(mk_ingest_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, d_tile, n_tiles);
    hipLaunchKernelGGL(mk_ingest_offsets_kernel, dim3(n_tiles), dim3(256), 0, st, d_seq_len, n_rec, d_tile, d_off);
}

void launch_ingest_gather(const uint8_t *d_text, const uint32_t *d_seq_start, const uint32_t *d_seq_len, const unsigned long long *d_off,
                          uint32_t fixed_len, uint64_t n_rec, uint8_t *d_seq, hipStream_t st, uint32_t skip_from) {
    if (!n_rec) return;
    hipLaunchKernelGGL(mk_ingest_gather_kernel, dim3((unsigned)((n_rec * 16 + 255) / 256)), dim3(256), 0, st, d_text, d_seq_start, d_seq_len, d_off,
                       fixed_len, n_rec, d_seq, skip_from);
}

uint32_t ingest_block_bytes() { return kIngestBlockBytes; }
uint32_t ingest_scan_tile() { return kScanTile; }

}  // namespace mk
