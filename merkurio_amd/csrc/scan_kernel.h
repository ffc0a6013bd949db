// scan_kernel.h -- host-visible declarations of the gfx950 scan kernels (scan_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/merkurio_hip.h"
#include "filter.hpp"

namespace mk {

constexpr uint32_t kHitStage = 1024;  // tuples a wave stages before it reserves output slots
constexpr uint32_t kFlagListCap = 2048;  // flagged records a scan wave can list (sparse-hit kernels); beyond: direct stores

struct ScanParams {
    // text: concatenated records
    const uint8_t *seq;       // 16-byte aligned
    uint64_t n_bytes;         // == rec_off[n_rec]
    const uint64_t *rec_off;  // n_rec + 1
    uint64_t n_rec;
    // compiled pattern set
    const uint32_t *bloom;    // LDS mode: kBloomWords words; global mode: 2 * gbloom_blocks words
    uint32_t gbloom_blocks;   // global mode: number of 64-bit filter blocks (0 = LDS mode)
    const TableEntry *table;  // (table_mask + 1) buckets of kBucketEntries entries
    uint32_t table_mask;      // bucket index mask
    const uint8_t *pat_bytes;
    const uint32_t *pat_off;  // n_pat + 1
    uint32_t n_pat;
    uint32_t q;           // q-gram length (1..32)
    uint32_t key_mask_lo;  // low / high 32 bits of the 2q-bit key mask
    uint32_t key_mask_hi;
    // second length class (MC kernels; s2 == 0: none): stride, key mask of its q2 <= 8 base q-grams, table over the
    // packed keys (kShortBitmapWords words, device memory; staged in LDS by every workgroup)
    uint32_t s2;
    uint32_t s2_log2;
    uint32_t key2_mask;
    uint32_t short_bytes;  // 1: the table holds one byte per key (q2 <= 6), 0: one bit per key
    const uint32_t *short_bitmap;
    uint32_t case_insensitive;
    uint32_t uniform_len;  // > 0: every pattern has this length (pattern i starts at i * uniform_len)
    double rec_per_byte;  // n_rec / n_bytes: record-index estimate for the lookup in resolve_one
    uint32_t tile_run;    // consecutive tiles a wave takes before it jumps ahead (>= 1)
    // records of unequal length: rec_index[k] = index of the record that contains byte k * 64 Ki (one more
    // entry behind the last); null = equal lengths, the quotient p * rec_per_byte is the record
    const uint32_t *rec_index;
    // records of ONE length (rec_off[i] == i * rec_len, the caller's contract or mk_scan_batch's own check): the
    // record of a verified occurrence is computed, rec_off is never read (it may be null)
    uint32_t rec_len;
    double inv_rec_len;
    // per-scan-wave staging of verified occurrences (EMIT kernels; kHitStage tuples each)
    mk_hit *stage;
    // outputs
    uint32_t *rec_flags32;  // rec_flags viewed as 32-bit words (byte r = record r)
    // kernels for sparse hits: a wave appends the records it flags to its own list (flag_cap entries per
    // scan wave, its fill in flag_counts[wave]) instead of storing into rec_flags; launch_flag_scatter sets
    // the bytes afterwards.  null: flags are stored directly (also what a wave does once its list is full).
    uint32_t *flag_list;
    uint32_t *flag_counts;
    uint32_t flag_cap;
    mk_hit *hits;           // may be null when !EMIT
    uint64_t hits_cap;
    unsigned long long *n_hits;    // device counter (every occurrence, even beyond hits_cap)
    unsigned long long *counters;  // n_pat + MK_NUM_SUMMARY, may be null
    // sticky error word of the handle (never null): bit 0 = a tuple's position did not fit mk_hit.pos (an occurrence
    // 4 GiB or more into its record); set by the tuple kernels, read at the handle's next host round trip
    uint32_t *error_word;
};

// S = sampling stride (1,2,4,8,16); wide = q > 16 (64-bit keys); emit = write mk_hit tuples.
// Returns the kernel's name (static storage) or nullptr for an unsupported S.
// flavour: 1 sparse hits, 0 hit-dense text (scan_kernel_impl.hpp: FL; honoured by the kernels with the filter in LDS)
// p.s2 != 0 selects the two-class twin of the variant (filter in LDS only)
const char *launch_scan(const ScanParams &p, int S, bool wide, bool emit, bool global_filter, int flavour, int grid_blocks,
                        hipStream_t stream);

// host-side launcher of one kernel variant; defined (explicitly instantiated) in scan_variants.hip
template <int S, int QC, bool EMIT, bool GF, int FL, int MC = 0>
void launch_variant(const ScanParams &p, int grid_blocks, hipStream_t stream);

// static LDS bytes of one scan workgroup (filter + candidate rings + pattern counters)
uint32_t scan_lds_bytes();

// counters[pat] += occurrences of pat among the stored tuples (hits[0 .. min(*n_hits, hits_cap)))
void launch_hist_hits(const ScanParams &p, int grid_blocks, hipStream_t stream);

// rec_index[k] = largest r with rec_off[r] <= k * 64 Ki, for k = 0 .. ceil(n_bytes / 64 Ki) (one binary search
// per entry; records of unequal length only)
constexpr uint32_t kRecIndexShift = 16;
// flags32[0, n_words) = 0 and *n_hits = 0 (the start of every scan)
void launch_clear(uint32_t *flags32, uint64_t n_words, unsigned long long *n_hits, hipStream_t stream);
// rec_flags[list entry] = 1 for every entry the scan waves appended (n_waves lists of flag_cap entries)
void launch_flag_scatter(const uint32_t *flag_list, const uint32_t *flag_counts, uint32_t flag_cap, uint32_t n_waves, uint8_t *rec_flags,
                         hipStream_t stream);
void launch_rec_index(const uint64_t *rec_off, uint64_t n_rec, uint64_t n_bytes, uint32_t *rec_index, hipStream_t stream);
// ---- emission order (order_hits.hip: hand-written bin + LDS-sort kernels; order_hits_fallback.hip: library sort) ----
// How a tuple maps to its sort key (see order_hits.hip): fields (record, A, B); G = record << bits_a | A;
// bin = G >> shift; 8-byte key = (G mod 2^shift) << bits_b | B (or, binning on the whole triple: shift 0, the top
// b_hi bits of B in the bin index, the key = the low b_lo bits of B).
struct OrderKey {
    const uint32_t *pat_off;  // device: pattern i is pat_off[i+1] - pat_off[i] bytes long
    uint32_t uniform_len;     // != 0: every pattern has this length (no lookup)
    const uint32_t *rank;     // AC with mixed lengths: rank[pat] in (length descending, index ascending) order; null = identity
    const uint32_t *unrank;   // its inverse
    uint64_t rec_base;        // subtracted from every record index (the smallest one when re-binning; else 0)
    uint32_t ac;              // 1: Aho-Corasick order, 0: BNDMq order
    uint32_t bits_a;          // width of A inside G; 0 = histogram on the record alone (field widths not known yet)
    uint32_t bits_b;          // width of B (1..63)
    uint32_t b_hi;            // top bits of B that are part of the bin index (only with shift == 0; else 0)
    uint32_t b_lo;            // bits of B inside the key = bits_b - b_hi
    uint32_t shift;           // bin = G >> shift (<= 63)
    uint32_t n_bins;          // <= kOrderMaxBins
};
struct OrderScratch {
    unsigned long long *stats;  // [0..2] maxima of record, A, B; [3] largest bin; [4] ~(smallest record); [5] bins above kOrderLeafSmall
    uint32_t *g_cnt;            // n_bins
    uint32_t *bin_start;        // n_bins + 1
    uint32_t *cursor;           // n_bins
    uint32_t *big_list;         // n_bins: the bins above kOrderLeafSmall tuples
    uint64_t *keys;             // n
};
constexpr uint32_t kOrderMaxBins = 32768;  // LDS histogram: 128 KiB of u32 bins
constexpr uint32_t kOrderLeafMax = 16384;  // keys one workgroup sorts in LDS (136 KiB with padding)
constexpr uint32_t kOrderLeafSmall = 4096;  // bins up to this size share a launch of 256-lane workgroups; larger ones get their own
// zeroes nothing: the caller clears stats and g_cnt first.  Enqueues the histogram and the bin-start scan.
void launch_order_hist(const mk_hit *d_hits, uint64_t n, const OrderKey &L, const OrderScratch &S, int num_cus, hipStream_t st);
// scatter into bins + one LDS sort per bin; the sorted tuples replace d_hits[0, n)
void launch_order_scatter_leaf(mk_hit *d_hits, uint64_t n, const OrderKey &L, const OrderScratch &S, uint32_t max_bin, uint32_t n_big,
                               int num_cus, hipStream_t st);
// the kernels above use more than 64 KiB of dynamic LDS: raises their limit once per process
hipError_t order_kernels_prepare();
// library fallback (rocPRIM merge sort with the reference's comparator): tmp == nullptr only sets *tmp_bytes
hipError_t order_hits_library(mk_hit *d_hits, size_t n, bool ac, const uint32_t *d_pat_off, uint32_t uniform_len, void *tmp,
                              size_t *tmp_bytes, hipStream_t stream);

// ---- sets.hip: what the record loops derive from the ordered tuples ------------------------------------------
// tuples in (record, pattern, position) order -> the distinct patterns of every record (CSR) and their total;
// d_tile: scratch of max(ceil(n / 4096), ceil((n_rec + 1) / 4096)) * 8 bytes
void launch_pattern_sets(const mk_hit *d_hits, uint64_t n, uint64_t n_rec, uint32_t *d_found_pat, unsigned long long *d_found_off,
                         unsigned long long *d_total, void *d_tile, hipStream_t st);
void launch_count_u32(const uint32_t *d_list, uint64_t n, uint32_t *d_counts, uint32_t n_bins, hipStream_t st);
void launch_rows(const mk_hit *d_hits, uint64_t n, uint32_t file, mk_row *d_rows, hipStream_t st);
// paired extract: both mates' tuples in one list, the mate inside a key field (sets.hip)
void launch_pair_mark(mk_hit *d_hits, uint64_t n, uint32_t mate, bool ac, hipStream_t st);
void launch_rows_pair(const mk_hit *d_hits, uint64_t n, bool ac, mk_row *d_rows, hipStream_t st);
void launch_count_pair_heads(const mk_hit *d_hits, uint64_t n, uint32_t *d_counts, uint32_t n_bins, hipStream_t st);

// ---- build_tables.hip: the pattern set compiled into filter images + exact table on the device -----------------
struct BuildParams {
    const uint8_t *pat_bytes;  // device
    const uint32_t *pat_off;   // device, n_pat + 1
    uint32_t n_pat;
    uint32_t S, q;             // main class
    uint32_t split, S2, q2;    // length classes (split == 0: one class)
    uint32_t gbloom_blocks;    // != 0: the filter is a global-memory image of this many 64-bit blocks
    uint32_t gf_ctx;           // context fingerprints (filter.hpp: gf_has_ctx)
    uint32_t *bloom;           // zeroed: kBloomWords words (LDS image) or 2 * gbloom_blocks
    TableEntry *table;         // (bucket_mask + 1) * kBucketEntries slots; cleared by launch_build_tables
    uint32_t bucket_mask;
    uint32_t *short_table;     // zeroed kShortBitmapWords words (two classes only)
};
// clears the table and inserts every (pattern, offset) entry; the filter images must be zero
void launch_build_tables(const BuildParams &B, uint64_t n_slots, hipStream_t st);

// counts the flagged records of the scan into counters[n_pat + MK_SUM_RECORDS_HIT]
void launch_count_flags(const ScanParams &p, hipStream_t stream);

void launch_synth(uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len, uint32_t plant_every, const uint8_t *d_pat_bytes,
                  const uint32_t *d_pat_off, uint32_t n_pat, uint8_t *d_seq, uint64_t *d_seq_off, hipStream_t stream);

}  // namespace mk
