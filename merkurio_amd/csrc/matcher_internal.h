// matcher_internal.h -- the opaque mk_matcher handle (shared by the host translation units)
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "filter.hpp"
#include "scan_kernel.h"
#include "host_common.h"

struct mk_matcher {
    int device = 0;
    int num_cus = 256;
    uint32_t algo = MK_ALGO_AC;
    uint32_t flags = 0;
    uint32_t n_pat = 0;
    std::vector<uint8_t> pat_bytes;
    std::vector<uint32_t> pat_off;
    // filter
    uint32_t uniform_len = 0;  // > 0: all patterns share this length
    uint32_t q = 0, S = 1;
    // two length classes (matcher.cpp: plan_classes): patterns shorter than split_len (0 = one class) are the short
    // class -- stride S2, q-grams of q2 <= 8 bases, level 1 = d_short_table (byte or bit per packed key)
    uint32_t split_len = 0, S2 = 0, q2 = 0, n_short = 0;
    uint32_t *d_short_table = nullptr;
    uint64_t entries = 0;
    uint32_t table_slots = 0;
    uint32_t *d_bloom = nullptr;
    uint32_t gbloom_blocks = 0;  // > 0: filter lives in global memory (large pattern sets)
    uint32_t tile_run = 0;       // 0 = rule (per scan, from the batch size); else forced by mk_matcher_options
    // records with a hit per 1000 records, as last observed by mk_scan_batch or told by
    // mk_matcher_hint_hit_density: >= kDensePerMille selects the plain-load kernel variant
    uint32_t hit_density_pm = 0;
    // records of unequal length (set by mk_scan_batch from the offsets it is given, or by
    // mk_matcher_hint_record_lengths): every scan then builds a coarse record index first
    bool ragged = false;
    // > 0: every record of a device scan has this length (mk_matcher_set_fixed_record_length: the caller's
    // contract) / of the mk_scan_batch call in progress (checked on its host offsets)
    uint32_t fixed_rec_len = 0, batch_rec_len = 0;
    uint32_t *d_rec_index = nullptr;
    size_t d_rec_index_cap = 0;
    mk::TableEntry *d_table = nullptr;
    uint8_t *d_pat_bytes = nullptr;
    uint32_t *d_pat_off = nullptr;
    // workspace of the host-buffer API
    hipStream_t stream = nullptr;
    uint8_t *d_seq = nullptr;
    size_t d_seq_cap = 0;
    uint64_t *d_off = nullptr;
    size_t d_off_cap = 0;
    uint8_t *d_flags = nullptr;
    size_t d_flags_cap = 0;
    mk_hit *d_hits = nullptr;
    size_t d_hits_cap = 0;
    unsigned long long *d_nhits = nullptr;
    uint32_t *d_error = nullptr;  // sticky device error word (scan_kernel.h: error_word)
    mk_hit *d_stage = nullptr;  // EMIT kernels: per-wave staging of hit tuples
    // kernels for sparse hits: per-scan-wave lists of flagged records (scan_kernel.h: flag_list)
    uint32_t *d_flag_list = nullptr;
    uint32_t *d_flag_counts = nullptr;
    void *d_sort_tmp = nullptr;  // scratch of mk_order_hits_device (order_hits.hip)
    size_t d_sort_tmp_cap = 0;
    // AC with patterns of unequal length: rank of a pattern in (length descending, index ascending) order and
    // its inverse -- the tie order of matches that end on the same byte (order_hits.hip)
    uint32_t *d_pat_rank = nullptr;
    uint32_t *d_pat_unrank = nullptr;
    uint64_t last_n_rec = 0;  // records of the last mk_scan_device: the binning bound of mk_order_hits_device
    // what the last mk_order_hits_device did: 0 nothing, 1 record bins, 2 (record, A) bins, 3 library sort
    uint32_t order_path = 0, order_bins = 0, order_max_bin = 0;
    uint64_t order_path_calls[4] = {0, 0, 0, 0};  // successful ordering calls by path, over the handle's life (mk_matcher_order_stats)
    bool order_prepared = false;  // the ordering kernels' dynamic-LDS limit has been raised on this device
    // scratch of the driver loops (host_loops.cpp): pattern sets, counters, rows
    void *d_aux = nullptr;
    size_t d_aux_cap = 0;
    void *d_pair = nullptr;  // paired extract: mate 1's tuples while mate 2 is scanned
    size_t d_pair_cap = 0;
    // mk_extract_fastq_text (ingest.hip): the window's raw text, block counts + status words, line / record tables
    // (r05: one slot per input file of a window -- mk_extract_window takes paired inputs)
    struct TextSlot {
        void *d_text = nullptr, *d_ing_a = nullptr, *d_ing_b = nullptr, *d_fa_seq = nullptr;  // d_fa_seq: FASTA sequences without line ends
        size_t d_text_cap = 0, d_ing_a_cap = 0, d_ing_b_cap = 0, d_fa_seq_cap = 0;
    } txt[2];
    uint32_t bam_piece = 0;  // mk_tag_bam_window: bytes of text per piece of the record-chain index (0 = 64 KiB; mk_matcher_set_bam_piece)
    uint8_t *d_flags2 = nullptr;  // paired windows: mate 1's flags while mate 2 is scanned; the keep flags the kept records are selected by
    size_t d_flags2_cap = 0;
    // mk_upload_text_ahead: text windows copied on a stream of their own while the current window is processed.  Four
    // slots (two windows of two inputs), so that an upload that arrives before the previous one was consumed cannot overwrite it; the slot
    // states are guarded by ahead_mu (the uploader may be another host thread than the one inside the extract call).
    struct AheadSlot {
        void *d = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        const uint8_t *text = nullptr;  // != nullptr: holds (or is receiving) text[0, n), not consumed yet
        uint64_t n = 0;
    } ahead[4];
    hipStream_t stream_ahead = nullptr;
    std::mutex ahead_mu;
    // where the last driver-loop call (mk_extract_single / mk_tag_records) spent its time, milliseconds:
    // [0] upload (H2D), [1] device work (scan, ordering, sets, counts), [2] download (D2H), [3] host loops
    float batch_ms[4] = {0, 0, 0, 0};
    const char *kernel_name = "";
    int last_grid = 0;
    // optional per-launch kernel timing (hipEvents recorded on the launch stream, tightly
    // around the scan kernel): bench.py's roofline figure
    std::vector<hipEvent_t> ev_start, ev_stop;
    uint64_t timed_launches = 0;
    // one-process-per-GPU counter reduction (reduce.cpp): this rank's RCCL communicator
    void *comm = nullptr;
    int comm_rank = 0, comm_size = 0;
};

namespace mk {
// matcher.cpp: host-buffer batches in steps (mk_scan_batch, host_loops.cpp)
int ensure_device(void **p, size_t *cap, size_t need);
extern thread_local double g_alloc_ms, g_free_ms;
int batch_check(const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint64_t *n_bytes);
int batch_upload(mk_matcher *m, const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint64_t n_bytes, uint32_t *batch_len);
int batch_scan(mk_matcher *m, uint64_t n_bytes, uint64_t n_rec, uint32_t mode, uint32_t batch_len, uint64_t cap, uint64_t limit,
               unsigned long long *found);
int batch_flags(mk_matcher *m, uint64_t n_rec, uint8_t *rec_flags, uint64_t *flagged_out);
// tuples on the device into emission order: ac_order ? Aho-Corasick's : (record, pattern, position)
int order_hits_on_device(mk_matcher *m, void *d_hits, uint64_t n_hits, bool ac_order, void *stream);
int hip_fail(hipError_t e, const char *what);
}  // namespace mk
