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
    uint32_t case_insensitive;
    uint32_t uniform_len;  // > 0: every pattern has this length (pattern i starts at i * uniform_len)
    double rec_per_byte;  // n_rec / n_bytes: record-index estimate for the lookup in resolve_one
    uint32_t tile_run;    // consecutive tiles a wave takes before it jumps ahead (>= 1)
    // records of unequal length: rec_index[k] = index of the record that contains byte k * 64 Ki (one more
    // entry behind the last); null = equal lengths, the quotient p * rec_per_byte is the record
    const uint32_t *rec_index;
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
};

// S = sampling stride (1,2,4,8,16); wide = q > 16 (64-bit keys); emit = write mk_hit tuples.
// Returns the kernel's name (static storage) or nullptr for an unsupported S.
// plain_loads: the stream is read with cacheable loads (hit-dense text); honoured by the k-mer-family
// kernels of the LDS filter, ignored by the others.
const char *launch_scan(const ScanParams &p, int S, bool wide, bool emit, bool global_filter, bool plain_loads, int grid_blocks,
                        hipStream_t stream);

// host-side launcher of one kernel variant; defined (explicitly instantiated) in scan_variants.hip
template <int S, int QC, bool EMIT, bool GF, bool NTL>
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
// order_hits.hip: device tuples sorted in place into the reference's emission order; tmp == nullptr only sets *tmp_bytes
hipError_t order_hits_device(mk_hit *d_hits, size_t n, bool ac, const uint32_t *d_pat_off, uint32_t uniform_len, void *tmp,
                             size_t *tmp_bytes, hipStream_t stream);

// counts the flagged records of the scan into counters[n_pat + MK_SUM_RECORDS_HIT]
void launch_count_flags(const ScanParams &p, hipStream_t stream);

void launch_synth(uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len, uint32_t plant_every, const uint8_t *d_pat_bytes,
                  const uint32_t *d_pat_off, uint32_t n_pat, uint8_t *d_seq, uint64_t *d_seq_off, hipStream_t stream);

}  // namespace mk
