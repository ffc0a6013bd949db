/*
 * merkurio_hip.h -- C ABI of the MI355X-native multi-pattern matcher that replaces
 * MerKurio's pattern_matching hot path (BNDMq / Aho-Corasick) for `merkurio extract|tag`.
 *
 * The reference has no trait / plugin / FFI for this path: its drivers hold
 * `(Option<AhoCorasick>, Vec<(String, BNDMq)>)` (src/cmd_extract.rs:259, src/cmd_tag.rs:234)
 * and call the matcher once per record.  A GPU wants batches, so this ABI exposes the same
 * observable contract per *batch of records*; every entry point below names the reference
 * interface it replaces (paths relative to the reference repo).  A Rust host binds these
 * symbols with `extern "C"`.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross the boundary;
 *  - every function returns 0 (MK_OK) or a negative MK_E_* code; mk_last_error() gives the
 *    text for the calling thread; nothing throws across the ABI (host allocation failures and
 *    hipErrorOutOfMemory come back as MK_E_NOMEM);
 *  - there is NO CPU fallback: without a usable HIP device creation fails with MK_E_HIP;
 *  - a matcher handle is bound to one HIP device and supports ONE scan in flight at a time
 *    (its tuple staging buffer, timing events and launch bookkeeping are per handle): enqueue
 *    the next scan of a handle on the same stream, or after the previous one has completed;
 *    distinct handles (one per GPU / per process / per stream) are independent;
 *  - nothing here reads environment variables: tuning goes through mk_matcher_options;
 *  - the caller owns every buffer it passes; the library copies patterns at create time.
 */
#ifndef MERKURIO_HIP_H
#define MERKURIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_ABI_VERSION 7

/* ---- error codes.  -1..-3 map 1:1 to PatternError (src/pattern_matching.rs:28-36) ---- */
#define MK_OK 0
#define MK_E_EMPTY_PATTERN (-1)    /* PatternError::EmptyPattern */
#define MK_E_INVALID_Q (-2)        /* PatternError::InvalidQGramLength(q) */
#define MK_E_PATTERN_TOO_LONG (-3) /* PatternError::PatternTooLong(len, 64) */
#define MK_E_NO_PATTERNS (-4)      /* "No k-mers found ..." (src/helpers.rs:128-130,158-160) */
#define MK_E_NOMEM (-5)
#define MK_E_PAIR_MISMATCH (-6) /* paired inputs differ in record count (src/cmd_extract.rs:465-468,608-612) */
#define MK_E_HIP (-7)           /* HIP runtime error / no device */
#define MK_E_CAPACITY (-8)      /* output buffer too small; required size is reported */
#define MK_E_INVALID_ARG (-9)
#define MK_E_UNSUPPORTED (-10)
#define MK_E_RCCL (-11) /* RCCL missing or a collective failed (counter reduction only) */
#define MK_E_CORRUPT (-12) /* a BGZF member that does not inflate to its ISIZE / CRC-32 (mk_bgzf_inflate) */

/* algorithm selector: what the reference's `-a` / `-q` / auto rule decides
 * (src/cmd_extract.rs:166-171, src/helpers.rs:203-211).  The device scan is the same for
 * both; the algorithm fixes the EMISSION ORDER and the pattern_hit_counts semantics. */
#define MK_ALGO_AUTO 0  /* AC iff case-insensitive, or n >= 14, or max_len > 64 */
#define MK_ALGO_AC 1    /* aho_corasick::AhoCorasick (DFA, overlapping search) */
#define MK_ALGO_BNDMQ 2 /* pattern_matching::BNDMq, one instance per pattern */

#define MK_FLAG_ASCII_CASE_INSENSITIVE 1u /* AhoCorasickBuilder::ascii_case_insensitive(true) */

/* scan modes = what the reference loops consume from the matcher */
#define MK_MODE_ANY 0  /* per-record "any hit" flag only  (find_match / first-hit break) */
#define MK_MODE_HITS 1 /* + every (record, pattern, start) in reference emission order */

typedef struct mk_matcher mk_matcher;

/* one occurrence: pattern `pat` (index into the sorted unique pattern list) starts at byte
 * `pos` (0-based) of record `rec`.  == (mat.pattern().as_usize(), mat.start()) of
 * src/cmd_extract.rs:341-342 and the `o` of BNDMq::find_iter (src/cmd_extract.rs:367).
 * Limits of this build: a single record shorter than 4 GiB (`pos` is 32 bits; a batch may hold
 * any number of bytes and records), at most 2^27 - 2 patterns and at most 2^26 exact-table entries
 * (patterns x sampling stride, the stride being 1..16 as mk_matcher_filter_info reports it):
 * larger sets fail at create time with MK_E_UNSUPPORTED. */
typedef struct {
    uint64_t rec;
    uint32_t pat;
    uint32_t pos;
} mk_hit;

/* one log row: mk_hit plus the file (mate) index; logger.log_fields arguments,
 * src/logger.rs:41 */
typedef struct {
    uint64_t rec;
    uint32_t pat;
    uint32_t pos;
    uint32_t file; /* 0 = file 1, 1 = file 2 */
    uint32_t _pad;
} mk_row;

/* the scalar counters of src/cmd_extract.rs:285-289 / src/cmd_tag.rs:360-363 */
typedef struct {
    uint64_t nb_records_tot;
    uint64_t nb_bases;
    uint64_t nb_hits_tot[2];
    uint64_t nb_records_hit[2];
    uint64_t nb_records_extracted;
} mk_counters;

/* ------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------- */
int mk_abi_version(void);
const char *mk_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int mk_device_count(void);

/* ------------------------------------------------------------------------------------
 * Pattern preparation (host side; defines the pattern index space)
 * ---------------------------------------------------------------------------------- */
/* helpers::read_kmers_from_file body, src/helpers.rs:152-156: content -> raw k-mer lines.
 * Outputs are malloc'd by the library; release with mk_free. */
int mk_read_kmers_from_text(const uint8_t *content, size_t len, uint8_t **out_bytes, uint32_t **out_off,
                            uint32_t *out_n);
/* helpers::parse_pattern_list, src/helpers.rs:76-133: case conversion, reverse-complement
 * extension (-r) or canonical form (-c), drop empty, sort_unstable, dedup. */
int mk_parse_pattern_list(const uint8_t *in_bytes, const uint32_t *in_off, uint32_t n_in, int reverse_complement,
                          int canonical, int lowercase, int uppercase, uint8_t **out_bytes, uint32_t **out_off,
                          uint32_t *out_n);
/* needletail Sequence::reverse_complement / sequence::canonical as used at
 * src/helpers.rs:103,117.  out must hold n bytes. */
void mk_reverse_complement(const uint8_t *in, size_t n, uint8_t *out);
void mk_canonical(const uint8_t *in, size_t n, uint8_t *out);
/* helpers::recommend_aho_corasick, src/helpers.rs:203-211 */
int mk_recommend_aho_corasick(size_t num_patterns, size_t max_len);
/* pattern_matching::tune_q_value, src/pattern_matching.rs:213-225 (0 for len >= 65) */
size_t mk_tune_q_value(size_t pattern_len);
/* pattern_preprocessing::generate_masks, src/pattern_preprocessing.rs:24-43 */
int mk_generate_masks(const uint8_t *pattern, size_t m, uint64_t masks[256], uint64_t *accept);
void mk_free(void *p);

/* ------------------------------------------------------------------------------------
 * Matcher construction
 * replaces: BNDMq::new per pattern (src/pattern_matching.rs:61-78, called at
 * src/cmd_extract.rs:267-276) and AhoCorasick::builder()...build() (src/cmd_extract.rs:260-265,
 * src/cmd_tag.rs:235-240).
 *
 * pat_bytes/pat_off: n_pat patterns, pattern i = pat_bytes[pat_off[i] .. pat_off[i+1]); the
 * list must already be the sorted unique list of parse_pattern_list (indices are reported
 * as given).  algo: MK_ALGO_*.  q: BNDMq q-gram length, 0 = tune_q_value per pattern;
 * ignored for AC.  q has no effect on results, but its validation errors are reproduced
 * (MK_E_INVALID_Q, MK_E_EMPTY_PATTERN, MK_E_PATTERN_TOO_LONG for BNDMq patterns > 64 B).
 * flags: MK_FLAG_* (case-insensitive forces AC like src/cmd_extract.rs:166-167).
 * device: HIP device ordinal.
 * ---------------------------------------------------------------------------------- */
int mk_matcher_create(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat, uint32_t algo, uint32_t q,
                      uint32_t flags, int32_t device, mk_matcher **out);
/* Same, with explicit tuning / test options (NULL = defaults = mk_matcher_create).  Options never
 * change results, only the filter geometry the scan kernel runs with. */
typedef struct {
    uint32_t struct_size;         /* = sizeof(mk_matcher_options); lets the struct grow compatibly */
    uint32_t force_stride;        /* 0 = geometry rule; else sampling stride 1, 2, 4, 8 or 16 */
    uint32_t force_global_filter; /* 1 = level-1 filter in global memory even for small sets */
    uint32_t gbloom_log2_blocks;  /* 0 = rule; else log2 of the number of 64-bit blocks of a global filter */
    uint32_t tile_run;            /* 0 = rule; else 1..8 consecutive 31 KiB tiles a scan wave takes before it jumps ahead */
    uint32_t gbloom_kib;          /* 0 = rule; else size of a global filter in KiB (any size, overrides gbloom_log2_blocks) */
    /* length classes (ABI 4).  A set whose shortest pattern is much shorter than the rest is split by length: the
     * short patterns get their own stride and q-gram table next to the main filter (one pass, one kernel). */
    uint32_t length_classes;      /* 0 = rule (split where the cost model says it pays); 1 = never split; 2 = split */
    uint32_t force_split_len;     /* 0 = rule; else patterns shorter than this form the short class */
    uint32_t force_stride2;       /* 0 = rule; else stride of the short class: 1, 2, 4 or 8 */
    uint32_t force_q2;            /* 0 = rule; else q-gram length of the short class, 1..8 (clamped to what its shortest pattern admits) */
} mk_matcher_options;
int mk_matcher_create_ex(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat, uint32_t algo, uint32_t q,
                         uint32_t flags, int32_t device, const mk_matcher_options *options, mk_matcher **out);
/* The filter geometry mk_matcher_create_ex would choose for patterns of these lengths, without creating anything (host
 * only, no device needed) -- this library's counterpart of what AhoCorasick::builder()...build() decides about its
 * automaton from the pattern list (src/cmd_extract.rs:260-265), which scans any list at one speed: main-class q-gram length and stride, whether the level-1 filter fits LDS, and the length
 * classes (see mk_matcher_class_info).  Any output pointer may be NULL. */
int mk_plan_geometry(const uint32_t *pat_len, uint32_t n_pat, const mk_matcher_options *options, uint32_t *q_gram, uint32_t *stride,
                     uint32_t *in_lds, uint32_t *split_len, uint32_t *n_short, uint32_t *q_gram2, uint32_t *stride2);
void mk_matcher_destroy(mk_matcher *m);
/* MK_ALGO_AC or MK_ALGO_BNDMQ after the auto rule was applied */
uint32_t mk_matcher_algo(const mk_matcher *m);
uint32_t mk_matcher_num_patterns(const mk_matcher *m);
/* filter geometry chosen at create time: q-gram length, sampling stride, table entries */
int mk_matcher_filter_info(const mk_matcher *m, uint32_t *q_gram, uint32_t *stride, uint64_t *entries,
                           uint64_t *table_bytes);
/* length classes chosen at create time: *split_len = 0 (one class: mk_matcher_filter_info describes it) or the
 * length below which a pattern belongs to the short class, the number of such patterns, and the short class's
 * q-gram length and stride (mk_matcher_filter_info then describes the main class; `entries` counts both) */
int mk_matcher_class_info(const mk_matcher *m, uint32_t *split_len, uint32_t *n_short, uint32_t *q_gram2, uint32_t *stride2);
/* where the level-1 filter lives: *in_lds = 1 (128 KiB image staged in LDS by every workgroup)
 * or 0 (large pattern sets: blocks in global memory, L2 / Infinity-Cache resident), and its size */
int mk_matcher_filter_mode(const mk_matcher *m, uint32_t *in_lds, uint64_t *filter_bytes);

/* ------------------------------------------------------------------------------------
 * Batched scan, host buffers
 * replaces, for a whole batch of records: BNDMq::find_match / find_iter / find_all
 * (src/pattern_matching.rs:128-153) and AhoCorasick::find_overlapping_iter
 * (src/cmd_extract.rs:332,480,507; src/cmd_tag.rs:393-396).
 *
 * record i = seq_bytes[seq_off[i] .. seq_off[i+1]) (newline-free sequence bytes exactly as
 * needletail's record.seq() / bam's record.sequence() hand them to the matcher).
 * rec_flags[i] = 1 iff any pattern occurs in record i.
 * MK_MODE_HITS: hits[0..*n_hits) = every occurrence, in the reference's emission order for
 * the matcher's algorithm: AC: record, end ascending, start ascending, pattern ascending;
 * BNDMq: record, pattern ascending, start ascending.
 * If more than hits_cap occurrences exist: returns MK_E_CAPACITY, *n_hits = required
 * count, rec_flags valid, hits content unspecified.  Never truncates silently.
 * ---------------------------------------------------------------------------------- */
int mk_scan_batch(mk_matcher *m, const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint32_t mode,
                  uint8_t *rec_flags, mk_hit *hits, uint64_t hits_cap, uint64_t *n_hits);

/* ------------------------------------------------------------------------------------
 * Batched scan, device-resident buffers (the kernel boundary; asynchronous)
 *
 * All pointers are device pointers on the matcher's device.  d_seq must be 16-byte aligned and
 * d_seq_off[0] must be 0 (offsets relative to d_seq);
 * d_rec_flags 4-byte aligned with its allocation padded to a multiple of 4 bytes.
 * The call enqueues on `stream` (a hipStream_t, NULL = default stream): clear of
 * d_rec_flags[0..n_rec) and of *d_n_hits, then the scan kernel (and, behind it, a small kernel that sets
 * the flag bytes of the records the scan has listed).  d_hits may be NULL in
 * MK_MODE_ANY.  Hits are written UNORDERED (order them with mk_order_hits_device, or with
 * mk_order_hits after copying back); in MK_MODE_HITS *d_n_hits counts every occurrence even beyond hits_cap (0 in MK_MODE_ANY).
 * d_counters (may be NULL): uint64[n_pat + MK_NUM_SUMMARY] accumulated (+=) by the scan:
 *   [0, n_pat)            occurrences per pattern (the AC meaning of pattern_hit_counts,
 *                         src/cmd_extract.rs:353).  MK_MODE_HITS only, counted from the stored tuples
 *                         (all of them unless *d_n_hits exceeds hits_cap); MK_MODE_ANY leaves these
 *                         entries untouched -- the reference's no-logging path counts nothing either
 *   [n_pat + MK_SUM_*]    see below
 * It is the vector a multi-GPU host sums across ranks (RCCL allReduce) at the end of a job.
 * ---------------------------------------------------------------------------------- */
#define MK_NUM_SUMMARY 8
#define MK_SUM_HITS 0        /* nb_hits_tot */
#define MK_SUM_RECORDS_HIT 1 /* nb_records_hit */
#define MK_SUM_RECORDS 2     /* nb_records_tot */
#define MK_SUM_BASES 3       /* nb_bases */
#define MK_SUM_CANDIDATES 4  /* filter positives sent to verification (diagnostic) */

int mk_scan_device(mk_matcher *m, const void *d_seq, uint64_t n_bytes, const void *d_seq_off, uint64_t n_rec,
                   uint32_t mode, void *d_rec_flags, void *d_hits, uint64_t hits_cap, void *d_n_hits,
                   void *d_counters, void *stream);

/* mk_scan_device only enqueues, so what a kernel finds wrong with its input comes back later: waits for `stream`,
 * then returns MK_E_UNSUPPORTED if the handle's LAST MK_MODE_HITS scan met an occurrence 4 GiB or more into its
 * record (mk_hit.pos cannot hold it; the flags of that scan are valid, its tuples are not), else MK_OK.
 * mk_order_hits_device makes the same check at its own round trip (from two tuples up: fewer need no ordering and are
 * not checked there); mk_scan_batch refuses such a record before it scans.  The condition is cleared by the call that
 * reports it and by the next MK_MODE_HITS scan on the handle. */
int mk_matcher_check_device(mk_matcher *m, void *stream);

/* Performance hint for mk_scan_device (never changes results): how many of 1000 records the caller
 * expects to contain a pattern.  Dense text (>= 145, or >= 95 when tuples are written) is streamed with cacheable loads, because the
 * exact verification re-reads every hit window; sparse text with non-temporal loads.  mk_scan_batch
 * and the driver-loop entry points maintain the value themselves from the batch they have just scanned. */
int mk_matcher_hint_hit_density(mk_matcher *m, uint32_t records_hit_per_1000);

/* Fixed-length batches (not a hint -- a statement about the data): with record_length > 0 every following
 * mk_scan_device on this handle takes record i to be d_seq[i * record_length, (i + 1) * record_length):
 * n_bytes must equal n_rec * record_length (else MK_E_INVALID_ARG) and d_seq_off is NOT read (it may be NULL --
 * 8 bytes per record the caller need not build).  The record of a verified occurrence is then computed instead
 * of looked up: one random memory transaction fewer per occurrence.  0 (the default) returns to offsets.
 * mk_scan_batch and the driver-loop entry points check their host offsets and do this by themselves. */
int mk_matcher_set_fixed_record_length(mk_matcher *m, uint32_t record_length);

/* Second hint for mk_scan_device: do the records of the batches differ in length (trimmed reads)?  With
 * equal_lengths == 0 every scan first builds a coarse record index (one entry per 64 KiB of text) for the
 * record lookup of verified occurrences; with 1 (the default) the index of a record is estimated from its
 * position, which is exact for equal lengths and merely slower otherwise.  mk_scan_batch decides by itself. */
int mk_matcher_hint_record_lengths(mk_matcher *m, int equal_lengths);

/* Sort hits (host memory) into the reference's emission order for this matcher:
 * Aho-Corasick find_overlapping_iter (src/cmd_extract.rs:332, src/cmd_tag.rs:393-396) per record end
 * ascending, longer pattern first, pattern id; BNDMq driver loop (src/cmd_extract.rs:365-384) per record
 * pattern-major, positions ascending. */
int mk_order_hits(const mk_matcher *m, mk_hit *hits, uint64_t n_hits);
/* The same order for tuples still on the device (d_hits as written by mk_scan_device, n_hits <= hits_cap of
 * them, 16-byte aligned), sorted in place on `stream` by hand-written kernels: tuples are counted into bins of
 * consecutive records, stored into their bin as 8-byte keys and each bin is sorted in LDS -- two passes over the
 * tuples and one over the keys (order_hits.hip).  The call waits ONCE for the stream (32 bytes of histogram
 * statistics fix the key layout; the caller has just read n_hits the same way) and returns with the rest enqueued.
 * Bins use the record count of the handle's last mk_scan_device; tuples of another batch are still ordered
 * correctly.  Batches that defeat the binning (a bin above 16384 tuples after re-binning on the top bits of the whole
 * record / end / pattern key,
 * more than 2^32 - 1 tuples, or record / end / pattern fields that do not fit 64 bits together) are ordered by a
 * library merge sort instead.  mk_scan_batch uses this by itself.  The scratch buffer (8 bytes per tuple) lives in
 * the handle (growing it synchronises the device; one call in flight per handle). */
int mk_order_hits_device(mk_matcher *m, void *d_hits, uint64_t n_hits, void *stream);
/* what the last mk_order_hits_device on this handle did: *path = 0 nothing (fewer than two tuples), 1 bins of
 * consecutive records, 2 bins on the top bits of the whole (record, end, pattern) key, 3 library merge sort; the number of
 * bins and the largest bin (paths 1 and 2) */
int mk_matcher_order_info(const mk_matcher *m, uint32_t *path, uint32_t *n_bins, uint32_t *max_bin);
/* (v6) how often each of those paths ran over the handle's life, driver-loop calls included: calls[p] for p = 0..3.  A host
 * program can tell from calls[3] whether its data ever sent the ordering to the library sort. */
int mk_matcher_order_stats(const mk_matcher *m, uint64_t calls[4]);

/* name of the scan kernel variant the last mk_scan_device on this handle launched, and its
 * launch geometry (for profiling / roofline bookkeeping) */
const char *mk_matcher_kernel_name(const mk_matcher *m);
int mk_matcher_launch_info(const mk_matcher *m, uint32_t *grid_blocks, uint32_t *block_threads, uint32_t *lds_bytes);
/* Per-launch kernel timing with hipEvents recorded on the launch stream immediately before
 * and after the scan kernel (not around the buffer clears).  enable_timing(slots) keeps the
 * last `slots` launches; kernel_times() waits for them, returns their durations in ms
 * (oldest first) and resets the window.  slots == 0 disables. */
int mk_matcher_enable_timing(mk_matcher *m, uint32_t slots);
int mk_matcher_kernel_times(mk_matcher *m, float *ms, uint32_t cap, uint32_t *n_out);

/* ------------------------------------------------------------------------------------
 * Driver-loop semantics on batches (host buffers)
 * These restate what the reference's record loops do with the matcher's answers, so that
 * counters, log rows and keep/drop decisions are bit-identical.
 * ---------------------------------------------------------------------------------- */
/* extract, single file: loop body src/cmd_extract.rs:321-406.
 * logging == 0: only keep[] and nb_records_extracted are produced (like the reference).
 * rows (may be NULL when logging == 0) receives the log rows in emission order; on overflow
 * returns MK_E_CAPACITY with *n_rows = required.  pattern_hit_counts[n_pat] is += updated. */
int mk_extract_single(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                      int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                      mk_counters *counters, uint32_t *pattern_hit_counts);
/* extract, paired: loop body src/cmd_extract.rs:463-612.  keep[i] applies to pair i. */
int mk_extract_paired(mk_matcher *m, const uint8_t *seq1, const uint64_t *off1, uint64_t n_rec1,
                      const uint8_t *seq2, const uint64_t *off2, uint64_t n_rec2, int logging, int invert,
                      uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *counters,
                      uint32_t *pattern_hit_counts);
/* tag: process_record matching + set logic, src/cmd_tag.rs:387-467.
 * found_off[n_rec+1] / found_pat: CSR of the distinct matched pattern indices per record,
 * ascending (= kmers_found after sort_unstable + dedup, src/cmd_tag.rs:484-485, before the
 * merge with a pre-existing tag value).  On found_cap overflow: MK_E_CAPACITY, found_off[n_rec]
 * = required. */
int mk_tag_records(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                   int filter_matching, int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                   mk_counters *counters, uint32_t *pattern_hit_counts, uint64_t *found_off, uint32_t *found_pat,
                   uint64_t found_cap);
/* extract, single FASTQ file, from the window's RAW TEXT (replaces needletail's record parsing for
 * this loop, src/cmd_extract.rs:281-282,321-328): text[0, n_text) starts at a record start and ends behind a whole
 * record (n_text < 4 GiB).  The bytes are uploaded as they are (fastest from memory of mk_host_alloc), the records are
 * indexed and their sequence lines gathered on the device, then the loop body of mk_extract_single runs.
 * Outputs: *n_rec records; rec_start[i] = offset of record i's '@' (rec_start[n_rec] = n_text; room for rec_cap + 1);
 * keep / rows / counters / pattern_hit_counts as mk_extract_single (row.rec indexes the window's records).
 * Only plain 4-line FASTQ is taken ('@' line, sequence, '+' line, quality of the same length; LF or CRLF; no blank
 * lines).  Anything else sets *status = 1 and produces nothing: the caller parses that window with its own reader
 * (which also words the reference's parse errors).  More than rec_cap records: MK_E_CAPACITY, *n_rec = required. */
int mk_extract_fastq_text(mk_matcher *m, const uint8_t *text, uint64_t n_text, int logging, int invert, uint64_t rec_cap,
                          uint64_t *n_rec, uint64_t *rec_start, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                          mk_counters *counters, uint32_t *pattern_hit_counts, uint32_t *status);
/* Optional overlap for a host that walks a file window by window: starts the upload of the NEXT window (page-locked
 * memory) on a stream of its own and returns; the mk_extract_fastq_text call that is then given the same pointer and
 * size finds its text on the device already.  May be called from another host thread while a call on the handle is in
 * progress, but must have returned before the call that consumes it starts; any other window simply uploads itself. */
int mk_upload_text_ahead(mk_matcher *m, const uint8_t *text, uint64_t n_text);
/* page-locked host memory for buffers that are uploaded (text windows, record batches): the DMA engines read it
 * directly, pageable memory is staged through a bounce buffer by one runtime thread.  Release with mk_host_free. */
int mk_host_alloc(size_t bytes, void **out);
void mk_host_free(void *p);
/* Where the last mk_extract_single / mk_tag_records call on this handle spent its time, in milliseconds:
 * ms[0] upload of the records (host -> device), ms[1] device work (scan, emission order, log rows, per-pattern
 * counts, per-record pattern sets -- all kernels), ms[2] download of the results, ms[3] host loops (flags -> keep,
 * adding counts).  Both calls run on the device from end to end; the host only copies and decides keep. */
int mk_matcher_batch_times(const mk_matcher *m, float ms[4]);

/* tag value, src/cmd_tag.rs:470-490: found patterns merged with an existing tag value
 * (split on ','), sort_unstable, dedup, join(",").  Writes a NUL-terminated string into
 * out (cap bytes); returns MK_E_CAPACITY with *out_len = required (excl. NUL) if too small. */
int mk_tag_value(const mk_matcher *m, const uint32_t *found_pat, uint64_t n_found, const char *existing, char *out,
                 size_t cap, size_t *out_len);

/* ------------------------------------------------------------------------------------
 * Synthetic workload generator (device; used by bench.py and the full-size parity tests)
 * Fills d_seq[0..n_rec*read_len) with uniform ACGT from a counter-based generator
 * (value at byte i depends only on (seed, i)), writes d_seq_off[i] = i*read_len, and plants
 * pattern (i * 2654435761 mod n_pat) of the matcher at a hashed offset in every record i
 * with hash(seed, i) % plant_every == 0 (plant_every == 0: none).
 * mk_synth_reads_host produces the identical bytes on the CPU for record range [rec0, rec0+n).
 * ---------------------------------------------------------------------------------- */
int mk_synth_reads_device(mk_matcher *m, uint64_t seed, uint64_t n_rec, uint32_t read_len, uint32_t plant_every,
                          void *d_seq, void *d_seq_off, void *stream);
int mk_synth_reads_host(const mk_matcher *m, uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len,
                        uint32_t plant_every, uint8_t *seq, uint64_t *seq_off);
/* records [rec0, rec0 + n_rec) of the same synthetic job (a rank's shard of a strong-scaled run):
 * d_seq_off is relative to the shard; rec0 * read_len must be a multiple of 32 */
int mk_synth_reads_device_range(mk_matcher *m, uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len,
                                uint32_t plant_every, void *d_seq, void *d_seq_off, void *stream);

/* ------------------------------------------------------------------------------------
 * Multi-GPU: the counter reduction (the path's only collective)
 * Records shard across GPUs with no data-path exchange; at the end of a job the per-GPU
 * vectors uint64[n_pat + MK_NUM_SUMMARY] (pattern_hit_counts | the scalars of
 * src/cmd_extract.rs:285-290, src/cmd_tag.rs:360-364) are summed with RCCL ncclAllReduce
 * over xGMI.  librccl is bound at run time; without it these calls return MK_E_RCCL.
 *
 * One process, one handle per GPU:  mk_reduce_counters(handles, n, vectors, len, host_sum)
 * sums d_counters[0..n) element-wise IN PLACE (every vector holds the sum on return) and
 * copies it to host_sum (may be NULL).  d_counters[i] is a device pointer on handles[i]'s
 * device.  Handles that share a device are added on that device first; the all-reduce runs
 * over the distinct devices.  Blocking: waits for all work enqueued on those devices.
 * Failure is total: if ncclCommInitAll fails nothing is cached; if the grouped all-reduce fails the communicators
 * of that device list are destroyed and the next call creates them again (MK_E_RCCL either way; the vectors of
 * handles that share a device may already hold their device-local sum).
 *
 * One process per GPU:  rank 0 calls mk_comm_unique_id and hands the id bytes to the other
 * ranks (file, MPI, torch.distributed, ...); every rank calls mk_comm_init (collective), then
 * mk_comm_reduce_counters enqueues the in-place all-reduce on `stream`.
 * ---------------------------------------------------------------------------------- */
#define MK_COMM_ID_BYTES 128
int mk_reduce_counters(mk_matcher *const *per_gpu, int n, void *const *d_counters, size_t len, uint64_t *host_sum);
/* (v6) creates the communicators mk_reduce_counters will use for these handles' devices and keeps them -- RCCL's set-up takes
 * seconds, the reduction itself microseconds: call it from a thread of its own when the job starts.  Nothing is synchronised,
 * nothing reduced; mk_reduce_counters works without it. */
int mk_reduce_prepare(mk_matcher *const *per_gpu, int n);
/* MK_OK if librccl could be bound in this process (ranks other than 0 can check before the collective init) */
int mk_comm_available(void);
int mk_comm_unique_id(uint8_t id[MK_COMM_ID_BYTES]);
int mk_comm_init(mk_matcher *m, const uint8_t id[MK_COMM_ID_BYTES], int rank, int n_ranks);
int mk_comm_reduce_counters(mk_matcher *m, void *d_counters, size_t len, void *stream);
/* ranks of the handle's communicator as RCCL reports them (ncclCommCount) */
int mk_comm_size(const mk_matcher *m, int *n_ranks);
int mk_comm_destroy(mk_matcher *m);

/* -------------------------------------------------------------------------------------
 * BGZF codec on the device (v5) -- the (de)compression either side of the hot path for BAM and bgzip'ed FASTA/FASTQ.
 * Replaces what the reference gets from `bam 0.1.4` / flate2 on `tag`'s reader and writer threads
 * (src/cmd_tag.rs:254-271 writer, :503-615 reader + record loop) and from needletail's gzip reader
 * (src/cmd_extract.rs:281): BGZF (SAM specification 4.1) is a series of independent gzip members of at most 64 KiB,
 * so members map to waves (deflate) and lanes (inflate).  A handle owns a stream and its device buffers, is bound to
 * one device and serialises its calls; independent of any mk_matcher.
 *
 * mk_bgzf_deflate: in[0, n) -> ceil(n / block_bytes) complete members back to back in out (block_bytes <= 65280;
 * 0 = 65280, what htslib writes).  The members inflate to `in` exactly, with correct CRC-32 / ISIZE / BSIZE; their
 * compressed bytes are this library's own parse (not zlib's).  No EOF marker is appended (mk_bgzf_eof gives its 28
 * bytes).  out_cap >= mk_bgzf_deflate_bound(n, block_bytes), else MK_E_CAPACITY with *out_len = that bound.
 *
 * mk_bgzf_inflate: the n_members members the caller found in in[0, n_in) by walking the BSIZE chain -> their text at
 * out + out_off.  Every member is checked: stream errors, ISIZE and CRC-32 (on the device).  A damaged member:
 * MK_E_CORRUPT, *bad_member = the first one, its status word in mk_last_error().  The call overwrites the span of out
 * its members cover (min out_off .. max out_off + isize: gaps between members included), nothing outside it.
 * --------------------------------------------------------------------------------------- */
typedef struct mk_codec mk_codec;
typedef struct mk_bgzf_member {
    uint64_t data_off; /* first byte of the raw DEFLATE stream (member start + 12 + XLEN) */
    uint64_t out_off;  /* where the member's text goes */
    uint32_t data_len; /* BSIZE + 1 - XLEN - 20 */
    uint32_t isize;    /* the trailer's ISIZE (<= 65536) */
    uint32_t crc;      /* the trailer's CRC-32 */
    uint32_t reserved;
} mk_bgzf_member;
int mk_codec_create(int device, mk_codec **out);
void mk_codec_destroy(mk_codec *c);
uint64_t mk_bgzf_deflate_bound(uint64_t n, uint32_t block_bytes);
int mk_bgzf_deflate(mk_codec *c, const uint8_t *in, uint64_t n, uint32_t block_bytes, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
/* the same for text that lies in several pieces (a writer's per-thread record buffers): the members are those of the
 * pieces' concatenation, which exists on the device only -- the host never copies the pieces together */
int mk_bgzf_deflate_pieces(mk_codec *c, const uint8_t *const *pieces, const uint64_t *sizes, uint64_t n_pieces, uint32_t block_bytes,
                           uint8_t *out, uint64_t out_cap, uint64_t *out_len);
int mk_bgzf_inflate(mk_codec *c, const uint8_t *in, uint64_t n_in, const mk_bgzf_member *members, uint64_t n_members, uint8_t *out,
                    uint64_t out_cap, uint64_t *bad_member);
/* mk_extract_fastq_text for a window of a bgzip'ed FASTQ: the members are uploaded as they are (a fifth of the text) and
 * inflated on the device STRAIGHT INTO the text buffer the ingest kernels read -- the text is never uploaded.  The window's
 * text = head[0, n_head) (the unfinished record the previous window ended with) followed by the members' text
 * (members[i].out_off = running sum of ISIZE from 0).  [0, io->n_used) of it are whole 4-line records and are what is indexed
 * and scanned, exactly as mk_extract_fastq_text does (same outputs, same *status = 1 for text that is not plain FASTQ: the
 * caller's own reader then takes the window); the rest, [io->n_used, io->n_text), is the next call's head.  What comes back
 * of the text is the caller's choice (mk_window_text):
 *   io->text != NULL: all of it, text[0, n_text) (text_cap >= n_head + sum of ISIZE, else MK_E_CAPACITY);
 *   io->text == NULL: only io->tail[0, n_tail) = the next head, and io->kept[0, n_kept_bytes) = the text of the KEPT records
 *     (keep[r] != 0), gathered on the device, back to back in record order -- record r's text is the next
 *     rec_start[r + 1] - rec_start[r] bytes of it.  (With logging AND invert the rows name records that are not kept: use
 *     the whole-text mode then.)  kept_cap too small: MK_E_CAPACITY with n_kept_bytes = the need; the last whole record must
 *     end within the window's last MiB, else *status = 1.
 * last != 0: no text follows, a final line without a line end counts, an unfinished record is a refusal (*status = 1).
 * A damaged member: MK_E_CORRUPT.  The codec handle lends its device buffers and must be on the matcher's device.
 * (The reference gets these records from needletail's gzip reader, src/cmd_extract.rs:281-282,321-328.) */
typedef struct mk_window_text {
    uint8_t *text;     /* in: whole-text mode buffer or NULL */
    uint64_t text_cap;
    uint8_t *tail;     /* in (text == NULL): the unfinished record at the window's end goes here */
    uint64_t tail_cap;
    uint8_t *kept;     /* in (text == NULL): the kept records' text goes here */
    uint64_t kept_cap;
    uint64_t n_text, n_used, n_tail, n_kept_bytes; /* out */
} mk_window_text;
int mk_extract_fastq_bgzf(mk_matcher *m, mk_codec *codec, const uint8_t *head, uint64_t n_head, const uint8_t *bgzf, uint64_t n_bgzf,
                          const mk_bgzf_member *members, uint64_t n_members, int last, mk_window_text *io, int logging, int invert,
                          uint64_t rec_cap, uint64_t *n_rec, uint64_t *rec_start, uint8_t *keep, mk_row *rows, uint64_t rows_cap,
                          uint64_t *n_rows, mk_counters *c, uint32_t *pattern_hit_counts, uint32_t *status);
/* -------------------------------------------------------------------------------------
 * Text windows of one or two input files (v6) -- the general form of mk_extract_fastq_text / mk_extract_fastq_bgzf: FASTQ or
 * FASTA, one input (the loop of src/cmd_extract.rs:321-406) or two (paired: :463-612), plain text or BGZF members, with heads.
 * Replaces needletail's parse_fastx_file + record.seq() for these loops (src/cmd_extract.rs:281-282, 412-418, 321-328, 463-468).
 *
 * A source's window text = head[0, n_head) ++ body, the body being text[0, n_text) (uploaded; fastest from mk_host_alloc memory;
 * mk_upload_text_ahead of the same pointer and size beforehand overlaps the copy with the previous window) or the text of
 * members[0, n_members) of bgzf[0, n_bgzf) (inflated on the device, never uploaded; needs `codec` on the matcher's device), or
 * device_text[0, n_device_text) (text that is on the device already: a gzip file inflated there by mk_gzip_inflate_device).  The
 * window starts at a record start.  ends_at_record != 0: its end is a record end (the end of the input, or a place the caller
 * chose); == 0: it may end anywhere -- the unfinished record stays behind as the tail.  The head is how a caller hands the
 * previous window's tail back in.
 * The records are indexed on the device: FASTQ -- '@' line, sequence, '+' line, quality of the same length, LF or CRLF, no blank
 * lines; FASTA -- '>' header lines, every other line sequence, which reaches the matcher without its '\n' / '\r' bytes
 * (record.seq(); a hit may span a line break: tests/fixtures/extract/fixed-width.log:8).  Anything else sets *status = 1 and
 * produces nothing: the caller's own reader takes that window (and words the reference's parse errors).
 * *n_rec = the records (two sources: PAIRS, record i of one with record i of the other) the call has processed = the smallest
 * number of whole records any source holds; per source: n_rec_seen = whole records it held, rec_start[0 .. *n_rec] = their offsets
 * in the window text (rec_start[*n_rec] = n_used, the bytes the processed records take), n_tail = n_window - n_used.  Outputs
 * keep / rows / counters / pattern_hit_counts: as mk_extract_single, resp. mk_extract_paired (row.rec = index in the window,
 * row.file = source).  What comes back of the text is the caller's choice, per source:
 *   tail (tail_cap):  the text behind the processed records, the next window's head     (MK_E_CAPACITY: n_tail = the need)
 *   kept (kept_cap):  the text of the KEPT records back to back, record r's being rec_start[r + 1] - rec_start[r] bytes -- for
 *                     callers that do not hold the text themselves (BGZF bodies)          (MK_E_CAPACITY: n_kept_bytes = the need)
 *   all  (all_cap):   the whole window text (logging with invert: rows name records that are not kept)
 * A damaged BGZF member: MK_E_CORRUPT.  More than rec_cap records: MK_E_CAPACITY with *n_rec = the need.
 * --------------------------------------------------------------------------------------- */
#define MK_TEXT_FASTQ 0
#define MK_TEXT_FASTA 1
typedef struct mk_window_source {
    /* in */
    const uint8_t *head;
    uint64_t n_head;
    const uint8_t *text;
    uint64_t n_text;
    const uint8_t *bgzf;
    uint64_t n_bgzf;
    const mk_bgzf_member *members; /* out_off = running sum of ISIZE from 0 */
    uint64_t n_members;
    const void *device_text; /* body that already lies on the matcher's device (a range of mk_gzip_text_device's text), or NULL */
    uint64_t n_device_text;
    uint32_t ends_at_record;
    uint32_t reserved;
    uint64_t *rec_start; /* room for rec_cap + 1, or NULL */
    uint8_t *tail;
    uint64_t tail_cap;
    uint8_t *kept;
    uint64_t kept_cap;
    uint8_t *all;
    uint64_t all_cap;
    /* out */
    uint64_t n_window, n_used, n_tail, n_kept_bytes, n_rec_seen;
} mk_window_source;
int mk_extract_window(mk_matcher *m, mk_codec *codec, uint32_t format, uint32_t n_sources, mk_window_source *sources, int logging, int invert,
                      uint64_t rec_cap, uint64_t *n_rec, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *counters,
                      uint32_t *pattern_hit_counts, uint32_t *status);

/* -------------------------------------------------------------------------------------
 * One gzip member inflated in parallel on the device (v6) -- a .fastq.gz / .fasta.gz as plain gzip (or pigz, or zlib) writes it: ONE
 * DEFLATE stream of thousands of blocks, which zlib can only walk from the front.  Replaces needletail's gzip reader under
 * `merkurio extract` (src/cmd_extract.rs:281-282) for such files.  The stream is cut where block starts can be FOUND (a dynamic
 * block's header is a pattern almost no bit position satisfies; a candidate is confirmed by decoding its block and meeting another
 * header behind it), the pieces are decoded side by side into 16-bit symbols -- bytes, or place-holders for the text in front of the
 * piece that a match reached into --, the place-holders are resolved piece by piece, and the text is checked against the member's
 * CRC-32 and ISIZE (merkurio_amd/csrc/codec/gzip_segments.hpp).
 * gz[0, n) = the whole member.  *taken = 1: the text lies on the codec's device, *text_bytes long, until the next call or
 * mk_gzip_text_release; read it with mk_gzip_text_read (host copy of a range) or hand windows of it to mk_extract_window
 * (mk_window_source::device_text).  *taken = 0: not a file for this path -- several members, a stream without findable block starts
 * where they are needed, a piece that does not meet its neighbour, an unusually compressible stream the buffers do not hold, a
 * CRC-32 / ISIZE that does not match --: nothing is reported as an error, the caller inflates the file with zlib (which then also
 * words what is wrong with a damaged one).
 * --------------------------------------------------------------------------------------- */
int mk_gzip_inflate_device(mk_codec *c, const uint8_t *gz, uint64_t n, uint64_t *text_bytes, uint32_t *taken);
int mk_gzip_text_read(mk_codec *c, uint64_t offset, uint8_t *out, uint64_t len);
const void *mk_gzip_text_device(const mk_codec *c, uint64_t *text_bytes);
int mk_gzip_text_release(mk_codec *c);
/* pieces of the last mk_gzip_inflate_device and its phases in milliseconds: upload, block search, pieces, resolution, CRC-32 */
int mk_gzip_info(const mk_codec *c, uint32_t *segments, float ms[5]);

/* -------------------------------------------------------------------------------------
 * `tag` on a window of a BAM file with the records RESIDENT ON THE DEVICE (v7) -- the reader loop, process_record and the writer
 * of src/cmd_tag.rs:503-615, :387-497, :254-271 for BAM -> BAM: only compressed members cross the host boundary, in both directions.
 *
 * The window's text = head[0, n_head) (the unfinished record the previous window ended with; the first window starts behind the
 * BAM header) ++ the text of members[0, n_members) of bgzf[0, n_bgzf) (out_off = running sum of ISIZE from 0), inflated on the
 * device and checked (CRC-32, ISIZE: MK_E_CORRUPT).  On the device: the records' block_size chain is indexed (pieces whose starts
 * are guessed and then proved by every piece's walk landing on the next piece's start: the table is the serial walk's), the 4-bit
 * sequences become the upper-case ASCII record.sequence() hands the matcher (:395), the scan, the emission order and the per-record
 * pattern sets run as in mk_tag_records, a record is kept by the rule of :457-467, and every kept record leaves as
 * block_size' | record | tag 'Z' value NUL -- the value being its distinct matched patterns, ascending, joined by ',' (:484-490),
 * merged with the items of the record's existing Z field of that name if it has one (:470-485; the old field stays)
 * -- packed back to back and deflated into BGZF members of block_bytes of text (0 = 65280; the last one shorter) in out[0, out_len).
 * The members inflate to exactly the bytes the CLI's host path writes for these records.
 * Outputs: n_window (bytes of text), n_rec whole records covering n_used bytes, tail[0, n_tail) = the text behind them (the next
 * window's head; tail_cap too small: MK_E_CAPACITY, n_tail = the need), n_kept, out_text_bytes (bytes of the kept records as
 * written), out_len (out_cap too small: MK_E_CAPACITY, out_len = the need).  out == NULL: nothing is written (`tag -s`).
 * logging != 0 (the reference's -l / -j): counters and pattern_hit_counts as mk_tag_records, rows[0, n_rows) in emission order
 * (row.rec = index in the window) and, per row, row_name[r] = offset of the record's NUL-terminated name in names[0, n_names_bytes)
 * (logger.log_fields' record.name(), :412); too small a rows / names buffer: MK_E_CAPACITY with n_rows / n_names_bytes = the need.
 * *status != 0: this window is not for the device and NOTHING was produced -- the caller's host reader takes it from the window's
 * first byte (and words the reference's errors): 1 = a record that fails the parser's checks or a chain that could not be proved,
 * 2 = optional fields that do not parse, 4 = a kept record whose field of the tag's name is not a string (the reference refuses it,
 * :482) or holds a value that is not plain ASCII or longer than 2 KiB, 8 = last != 0 and the text ends inside a record.
 * ms[]: milliseconds of upload, inflate, index, unpack + scan + sets, tag + emit, deflate, download; ms[7]: of these, growing device buffers.
 * --------------------------------------------------------------------------------------- */
typedef struct mk_bam_window {
    /* in */
    const uint8_t *head;
    uint64_t n_head;
    const uint8_t *bgzf;
    uint64_t n_bgzf;
    const mk_bgzf_member *members;
    uint64_t n_members;
    uint32_t last; /* no text follows this window */
    uint32_t filter_matching, invert;
    uint8_t tag[2];
    uint8_t reserved[2];
    uint32_t block_bytes;
    uint8_t *tail;
    uint64_t tail_cap;
    uint8_t *out;
    uint64_t out_cap;
    mk_row *rows;
    uint64_t rows_cap;
    uint64_t *row_name; /* room for rows_cap entries */
    uint8_t *names;
    uint64_t names_cap;
    /* optional: called from inside the call as soon as tail[0, n_tail) is known -- right after the record index, before the scan --
     * so that a caller with a second handle can start the next window (whose head this is) beside the rest of this one.  Not called
     * when the window is refused by the index (*status 1 or 8) or fails before that point. */
    void (*on_tail)(void *ctx, const uint8_t *tail, uint64_t n_tail);
    void *on_tail_ctx;
    /* out */
    uint64_t n_window, n_used, n_tail, n_rec, n_kept, out_text_bytes, out_len, n_rows, n_names_bytes;
    float ms[8];
} mk_bam_window;
int mk_tag_bam_window(mk_matcher *m, mk_codec *codec, mk_bam_window *w, int logging, mk_counters *counters, uint32_t *pattern_hit_counts,
                      uint32_t *status);
/* Tuning / test hook: bytes of text per piece of the record-chain index (0 = the default, 65536).  Results do not depend on it. */
int mk_matcher_set_bam_piece(mk_matcher *m, uint32_t piece_bytes);

/* walks the BSIZE chain of in[0, n): fills members[0, cap) (out_off = running sum of ISIZE), *n_members = how many there are,
 * *consumed = bytes of whole members, *text_bytes = sum of ISIZE.  MK_E_CORRUPT where a header is not BGZF; a trailing
 * partial member is not an error (*consumed < n).  Host code, no device. */
int mk_bgzf_members(const uint8_t *in, uint64_t n, mk_bgzf_member *members, uint64_t cap, uint64_t *n_members, uint64_t *consumed,
                    uint64_t *text_bytes);
/* the 28-byte empty member that ends a BGZF file */
const uint8_t *mk_bgzf_eof(void);
/* Tuning / test hook: a call is cut into device passes of at most `deflate_members` members (default 49 152 = 3.2 GB of
 * text) / `inflate_text_bytes` of text (default 3 GiB); 0 keeps the default.  Results do not depend on it. */
int mk_codec_set_pass_limits(mk_codec *c, uint64_t deflate_members, uint64_t inflate_text_bytes);
/* (v6) Tuning / test hook: which inflate kernel the handle's calls use -- 0 (default) by the number of members in the call,
 * 1 one lane per member (many members in flight, each slow), 2 one wave per member with the member's recent text in LDS (few in
 * flight, each fast), 3 / 4 / 5 / 6 the same with only the most recent 8 / 16 / 4 / 2 KiB of the member's text in LDS (12 / 7 / 19 / 25 waves per
 * CU instead of 4; matches that reach further read the text back from device memory).  mk_gzip_inflate_device gives a piece of
 * the stream to a wave, with 1 to a lane.  Results do not depend on it. */
int mk_codec_set_inflate_kernel(mk_codec *c, int which);
/* (v7) Tuning / test hook of mk_gzip_inflate_device: the nominal distance between two cuts of the stream, in compressed bytes -- a
 * power of two, 4 KiB ... 1 MiB; 0 (default): by the size of the stream (16 KiB up to ~200 MB of it, 64 KiB above).  A piece starts at the first DEFLATE block found behind a cut.  Results do not depend on it. */
int mk_codec_set_gzip_chunk(mk_codec *c, uint64_t chunk_bytes);
/* milliseconds of the handle's last call: [0] upload, [1] kernels, [2] download */
int mk_codec_times(const mk_codec *c, float ms[3]);

#ifdef __cplusplus
}
#endif
#endif
