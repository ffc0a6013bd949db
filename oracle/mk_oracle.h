/*
 * mk_oracle.h -- CPU restatement of MerKurio's pattern-matching hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call anything declared here; the product
 * path (merkurio_amd/, include/merkurio_hip.h) never does and fails loudly without a GPU.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against every
 * known-answer vector the reference holds for this path: the unit KATs of
 * src/pattern_matching.rs:353-488 and src/pattern_preprocessing.rs:54-84, the pattern-list
 * KATs of src/helpers.rs:319-431, the end-to-end fixtures tests/fixtures/{extract,tag}/ *,
 * the Aho-Corasick order vector tests/fixtures/extract/log.json, example-minimal and the
 * example-workflow goldens.  The reference itself (Rust, un-vendored crates) cannot be
 * built in this image, so no oracle/_ref exists; the aho-corasick 1.1.3 crate's published
 * algorithm (Standard match kind, DFA, overlapping search) is restated from its documented
 * contract and anchored on log.json.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference repository root).
 */
#ifndef MK_ORACLE_H
#define MK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes: map 1:1 to PatternError, src/pattern_matching.rs:28-36 */
#define MKO_OK 0
#define MKO_E_EMPTY_PATTERN (-1)
#define MKO_E_INVALID_Q (-2)
#define MKO_E_PATTERN_TOO_LONG (-3)
#define MKO_E_NO_PATTERNS (-4)
#define MKO_E_NOMEM (-5)
#define MKO_E_PAIR_MISMATCH (-6)

/* ---- src/pattern_preprocessing.rs:24-43 ---- */
int mko_generate_masks(const uint8_t *pattern, size_t m, uint64_t masks[256], uint64_t *accept);

/* ---- src/pattern_matching.rs:42-78 ---- */
typedef struct {
    size_t m;
    size_t q;
    uint64_t masks[256];
    uint64_t accept;
} mko_bndmq;

int mko_bndmq_new(const uint8_t *pattern, size_t m, size_t q, mko_bndmq *out);
/* src/pattern_matching.rs:82-130 (find_matches with |_| true) */
int mko_bndmq_find_match(const mko_bndmq *b, const uint8_t *text, size_t n);
/* src/pattern_matching.rs:133-209 (find_iter / Matches::next / find_all).
 * Writes up to cap starts into out; returns the total number of matches. */
size_t mko_bndmq_find_all(const mko_bndmq *b, const uint8_t *text, size_t n, size_t *out, size_t cap);
/* src/pattern_matching.rs:213-225; returns 0 for len >= 65 (the reference bails) */
size_t mko_tune_q_value(size_t pattern_len);
/* legacy BNDM, src/pattern_matching.rs:232-342 (not on the CLI path; kept for its KATs) */
size_t mko_bndm_find_all(const uint8_t *pattern, size_t m, const uint8_t *text, size_t n, size_t *out,
                         size_t cap);

/* ---- src/helpers.rs:203-211 ---- */
int mko_recommend_aho_corasick(size_t num_patterns, size_t max_len);
/* selection rule, src/cmd_extract.rs:166-171 / src/cmd_tag.rs:184-189.
 * q_given: user passed -q; returns 1 if Aho-Corasick is used. */
int mko_select_aho_corasick(int case_insensitive, int force_ac, int q_given, size_t num_patterns,
                            size_t max_len);

/* ---- pattern list, src/helpers.rs:76-133 (+ needletail reverse_complement / canonical) ---- */
/* in-place complement table semantics: needletail 0.6.3 `complement` */
uint8_t mko_complement(uint8_t c);
void mko_reverse_complement(const uint8_t *in, size_t n, uint8_t *out);
/* lexicographic min(seq, rc(seq)), src/helpers.rs:112-121 */
void mko_canonical(const uint8_t *in, size_t n, uint8_t *out);

typedef struct {
    uint32_t n;
    uint32_t *off;  /* n+1 */
    uint8_t *bytes; /* concatenated pattern text */
} mko_patterns;

/* Input: n_in raw patterns (already read from file or CLI). Applies case conversion,
 * reverse complement extension / canonicalisation, drops empty, sort_unstable + dedup.
 * Returns MKO_E_NO_PATTERNS if the result is empty. */
int mko_parse_pattern_list(const uint8_t *in_bytes, const uint32_t *in_off, uint32_t n_in,
                           int reverse_complement, int canonical, int lowercase, int uppercase,
                           mko_patterns *out);
/* src/helpers.rs:139-163: split file content into raw k-mer lines */
int mko_read_kmers_from_text(const uint8_t *content, size_t len, mko_patterns *out);
void mko_patterns_free(mko_patterns *p);

/* ---- Aho-Corasick DFA, overlapping search ([3P] aho-corasick 1.1.3;
 *      call sites src/cmd_extract.rs:260-265,332,480,507; src/cmd_tag.rs:235-240,393-396) ---- */
typedef struct mko_ac mko_ac;
int mko_ac_build(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat,
                 int ascii_case_insensitive, mko_ac **out);
void mko_ac_free(mko_ac *ac);
size_t mko_ac_num_states(const mko_ac *ac);
size_t mko_ac_table_bytes(const mko_ac *ac);
/* All (pattern, start) in emission order (end ascending; at one end: the state's own
 * patterns in pattern-id order, then the failure chain = shorter suffixes).
 * Writes up to cap entries; returns the total count. */
size_t mko_ac_find_overlapping(const mko_ac *ac, const uint8_t *text, size_t n, uint32_t *out_pat,
                               uint64_t *out_start, size_t cap);
/* first-hit break variant (logging off), src/cmd_extract.rs:332-335 */
int mko_ac_is_match(const mko_ac *ac, const uint8_t *text, size_t n);

/* ---- matcher bundle as the drivers hold it (src/cmd_extract.rs:259-277) ---- */
typedef struct {
    int use_ac;
    mko_ac *ac;
    uint32_t n_pat;
    mko_bndmq *bndmq; /* n_pat entries when !use_ac */
} mko_matcher;

/* q == 0 -> tune_q_value per pattern */
int mko_matcher_build(const mko_patterns *pl, int use_ac, size_t q, int case_insensitive,
                      mko_matcher *out);
void mko_matcher_free(mko_matcher *m);

/* ---- driver loops ---- */
typedef struct {
    uint8_t file;  /* 0 = file 1, 1 = file 2 */
    uint64_t rec;  /* record index inside its file */
    uint32_t pat;
    uint64_t pos;
} mko_row;

typedef struct {
    uint64_t nb_records_tot;
    uint64_t nb_bases;
    uint64_t nb_hits_tot[2];
    uint64_t nb_records_hit[2];
    uint64_t nb_records_extracted;
} mko_counters;

typedef struct {
    mko_row *rows;
    size_t n_rows, cap_rows;
} mko_rows;
void mko_rows_free(mko_rows *r);

/* extract, single file: src/cmd_extract.rs:321-406.
 * keep[n_rec] receives 1 where found != invert. pattern_hit_counts[n_pat] (u32 like the
 * reference) and counters are only updated when logging != 0, as in the reference. */
int mko_extract_single(const mko_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec,
                       int logging, int invert, uint8_t *keep, mko_rows *rows, mko_counters *c,
                       uint32_t *pattern_hit_counts);
/* extract, paired: src/cmd_extract.rs:463-612. n_rec2 != n_rec1 -> MKO_E_PAIR_MISMATCH. */
int mko_extract_paired(const mko_matcher *m, const uint8_t *seq1, const uint64_t *off1, uint64_t n_rec1,
                       const uint8_t *seq2, const uint64_t *off2, uint64_t n_rec2, int logging,
                       int invert, uint8_t *keep, mko_rows *rows, mko_counters *c,
                       uint32_t *pattern_hit_counts);
/* tag process_record matching + set logic: src/cmd_tag.rs:387-467.
 * found_off/found_pat: CSR of kmers_found per record *before* merge with an existing tag
 * (push order, duplicates included for AC). keep as per -m / -v. */
int mko_tag_records(const mko_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec,
                    int logging, int filter_matching, int invert, uint8_t *keep, mko_rows *rows,
                    mko_counters *c, uint32_t *pattern_hit_counts, uint64_t *found_off /* n_rec+1 */,
                    uint32_t **found_pat /* malloc'd */);

/* tag value, src/cmd_tag.rs:470-490: found patterns (+ existing tag value split on ',')
 * -> sort_unstable, dedup, join(","). Returns malloc'd NUL-terminated string. */
char *mko_tag_value(const mko_patterns *pl, const uint32_t *found_pat, size_t n_found,
                    const char *existing /* may be NULL */);

void mko_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
