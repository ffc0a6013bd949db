// host_loops.cpp -- what the reference's record loops do with the matcher's answers, restated
// for batches so that keep/drop decisions, log rows and counters are bit-identical:
//   extract single   src/cmd_extract.rs:321-406
//   extract paired   src/cmd_extract.rs:463-612
//   tag              src/cmd_tag.rs:387-490
// The matching itself is the gfx950 scan kernel; nothing here searches text.  extract (single) and tag work on the
// device from end to end: scan -> tuples ordered on the device (order_hits.hip) -> log rows, per-pattern counts and
// the per-record pattern sets by the kernels of sets.hip -> results copied back.  The host threads only turn flags
// into keep decisions.  The paired loop orders both mates' tuples as one list, the mate inside a key field.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "matcher_internal.h"
#include "codec/codec_internal.h"
#include "codec/codec_kernels.h"

using namespace mk;

namespace {

// scan one batch, growing the hit buffer on MK_E_CAPACITY
int scan_all(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, uint32_t mode,
             std::vector<uint8_t> &flags, std::vector<mk_hit> &hits) {
    flags.assign(n_rec ? n_rec : 1, 0);
    uint64_t n = 0;
    if (mode == MK_MODE_ANY) return mk_scan_batch(m, seq, off, n_rec, mode, flags.data(), nullptr, 0, &n);
    hits.resize(std::max<uint64_t>(4096, n_rec / 8));
    int rc = mk_scan_batch(m, seq, off, n_rec, mode, flags.data(), hits.data(), hits.size(), &n);
    if (rc == MK_E_CAPACITY) {
        hits.resize(n);
        rc = mk_scan_batch(m, seq, off, n_rec, mode, flags.data(), hits.data(), hits.size(), &n);
    }
    if (rc) return rc;
    hits.resize(n);
    return MK_OK;
}

struct RowSink {
    mk_row *rows;
    uint64_t cap;
    uint64_t n = 0;
    void push(uint32_t file, const mk_hit &h) {
        if (rows && n < cap) {
            rows[n].rec = h.rec;
            rows[n].pat = h.pat;
            rows[n].pos = h.pos;
            rows[n].file = file;
            rows[n]._pad = 0;
        }
        ++n;
    }
};

// pattern_hit_counts for one file's ordered hits.
// AC: += 1 per hit (src/cmd_extract.rs:353).  BNDMq: += 1 per (record, pattern) that has at
// least one hit (src/cmd_extract.rs:380-383): hits are pattern-major inside a record, so a
// new (rec, pat) run starts whenever either changes.
void count_patterns(uint32_t algo, const std::vector<mk_hit> &hits, uint32_t *counts) {
    if (algo == MK_ALGO_AC) {
        for (auto &h : hits) counts[h.pat] += 1;
    } else {
        for (size_t i = 0; i < hits.size(); ++i)
            if (i == 0 || hits[i].rec != hits[i - 1].rec || hits[i].pat != hits[i - 1].pat) counts[hits[i].pat] += 1;
    }
}

uint64_t popcount_flags(const std::vector<uint8_t> &f, uint64_t n) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; ++i) c += f[i] != 0;
    return c;
}

// One driver-loop call on the device: upload, scan, then whatever the loop derives from the tuples, each step
// enqueued on the handle's stream; the time of the call is split into upload / device / download / host.
struct DeviceLoop {
    mk_matcher *m;
    hipStream_t st;
    unsigned long long found = 0;  // tuples of the scan (all of them are on the device)
    uint64_t n_bytes = 0;
    using clk = std::chrono::steady_clock;
    clk::time_point t_last;
    int phase = 0;
    explicit DeviceLoop(mk_matcher *m_) : m(m_), st(m_->stream), t_last(clk::now()) {
        for (float &x : m->batch_ms) x = 0;
    }
    void mark(int next) {  // the stream is idle at every call: time since the last mark goes to the current phase
        const auto t = clk::now();
        m->batch_ms[phase] += std::chrono::duration<float, std::milli>(t - t_last).count();
        t_last = t;
        phase = next;
    }
    void host_begin() { mark(3); }
    void finish() { mark(3); }

    int scan(const uint8_t *seq, const uint64_t *off, uint64_t n_rec, uint32_t mode, uint8_t *flags, uint64_t *flagged) {
        int rc = batch_check(seq, off, n_rec, &n_bytes);
        if (rc) return rc;
        uint32_t batch_len = 0;
        phase = 0;
        if ((rc = batch_upload(m, seq, off, n_rec, n_bytes, &batch_len))) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "upload failed");
        mark(1);
        // all tuples stay on the device: no caller-side limit.  The scan starts with the room the handle already has
        // (hit-dense batches -- tag / extract -l on already extracted reads -- overflowed n_rec / 8 on EVERY batch and
        // ran their scan twice; after the first such batch one scan suffices)
        const uint64_t cap0 = std::max<uint64_t>(std::max<uint64_t>(4096, n_rec / 8), m->d_hits_cap / sizeof(mk_hit));
        if ((rc = batch_scan(m, n_bytes, n_rec, mode, batch_len, cap0, ~0ull, &found))) return rc;
        mark(2);
        if ((rc = batch_flags(m, n_rec, flags, flagged))) return rc;
        mark(1);
        return MK_OK;
    }
    // the batch is already in m->d_seq / m->d_off (ingest.hip put it there): scan + flags
    int scan_resident(uint64_t n_seq_bytes, uint64_t n_rec, uint32_t mode, uint32_t batch_len, uint8_t *flags, uint64_t *flagged) {
        n_bytes = n_seq_bytes;
        const uint64_t cap0 = std::max<uint64_t>(std::max<uint64_t>(4096, n_rec / 8), m->d_hits_cap / sizeof(mk_hit));
        int rc = batch_scan(m, n_bytes, n_rec, mode, batch_len, cap0, ~0ull, &found);
        if (rc) return rc;
        mark(2);
        if ((rc = batch_flags(m, n_rec, flags, flagged))) return rc;
        mark(1);
        return MK_OK;
    }
    int order(bool ac_order) { return order_hits_on_device(m, m->d_hits, found, ac_order, st); }

    // mk_row per tuple (in their current order) -> rows[0, min(found, cap))
    int rows_to_host(uint32_t file, mk_row *rows, uint64_t cap) {
        const uint64_t n = std::min<uint64_t>(found, rows ? cap : 0);
        if (!n) return MK_OK;
        int rc = ensure_device(&m->d_aux, &m->d_aux_cap, n * sizeof(mk_row));
        if (rc) return rc;
        launch_rows(m->d_hits, n, file, (mk_row *)m->d_aux, st);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "row kernel failed");
        mark(2);
        if (hipMemcpy(rows, m->d_aux, n * sizeof(mk_row), hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the log rows failed");
        mark(1);
        return MK_OK;
    }

    // paired extract: the marked pair list (sets.hip) -> rows with their mate
    int pair_rows_to_host(bool ac, mk_row *rows, uint64_t cap) {
        const uint64_t n = std::min<uint64_t>(found, rows ? cap : 0);
        if (!n) return MK_OK;
        int rc = ensure_device(&m->d_aux, &m->d_aux_cap, n * sizeof(mk_row));
        if (rc) return rc;
        launch_rows_pair(m->d_hits, n, ac, (mk_row *)m->d_aux, st);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "row kernel failed");
        mark(2);
        if (hipMemcpy(rows, m->d_aux, n * sizeof(mk_row), hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the log rows failed");
        mark(1);
        return MK_OK;
    }
    // pattern_hit_counts of the ordered pair list: AC one per hit of either mate (:492,:520); BNDMq one per pair,
    // pattern and mate with a hit (:575-584)
    int pair_counts(bool ac, uint32_t *counts) {
        const uint32_t n_pat = m->n_pat;
        if (ac) {
            unsigned long long n = found;  // the histogram kernel reads the tuple count from the device
            if (hipMemcpyAsync(m->d_nhits, &n, sizeof(n), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return fail(MK_E_HIP, "copy failed");
            return pattern_counts(true, 0, counts);
        }
        int rc = ensure_device(&m->d_aux, &m->d_aux_cap, (size_t)n_pat * 4);
        if (rc) return rc;
        if (hipMemsetAsync(m->d_aux, 0, (size_t)n_pat * 4, st) != hipSuccess) return fail(MK_E_HIP, "memset failed");
        launch_count_pair_heads(m->d_hits, found, (uint32_t *)m->d_aux, n_pat, st);
        std::vector<uint32_t> v(n_pat);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "count kernel failed");
        mark(2);
        if (hipMemcpy(v.data(), m->d_aux, (size_t)n_pat * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the counts failed");
        mark(3);
        for (uint32_t i = 0; i < n_pat; ++i) counts[i] += v[i];
        mark(1);
        return MK_OK;
    }

    // pattern_hit_counts += this batch's: per hit (AC, src/cmd_extract.rs:353) or per (record, pattern) with a hit
    // (BNDMq, :380-383; the tuples must be in BNDMq order = set order)
    int pattern_counts(bool per_hit, uint64_t n_rec, uint32_t *counts) {
        if (!found) return MK_OK;
        const uint32_t n_pat = m->n_pat;
        if (per_hit) {
            int rc = ensure_device(&m->d_aux, &m->d_aux_cap, ((size_t)n_pat + MK_NUM_SUMMARY) * 8);
            if (rc) return rc;
            if (hipMemsetAsync(m->d_aux, 0, ((size_t)n_pat + MK_NUM_SUMMARY) * 8, st) != hipSuccess) return fail(MK_E_HIP, "memset failed");
            ScanParams p;
            memset(&p, 0, sizeof(p));
            p.hits = m->d_hits;
            p.n_hits = m->d_nhits;  // still holds `found`
            p.hits_cap = found;
            p.counters = (unsigned long long *)m->d_aux;
            p.n_pat = n_pat;
            launch_hist_hits(p, m->num_cus, st);
            std::vector<unsigned long long> v(n_pat);
            if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "histogram kernel failed");
            mark(2);
            if (hipMemcpy(v.data(), m->d_aux, (size_t)n_pat * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the counts failed");
            mark(3);
            for (uint32_t i = 0; i < n_pat; ++i) counts[i] += (uint32_t)v[i];
            mark(1);
            return MK_OK;
        }
        // BNDMq: the heads of the (record, pattern) runs = the entries of the pattern sets
        uint64_t n_found = 0;
        return pattern_sets(n_rec, nullptr, nullptr, 0, &n_found, counts);
    }

    // the distinct patterns of every record (tuples in set order) as a CSR in the handle's scratch (m->d_aux): *d_off_out[n_rec + 1],
    // *d_pat_out[*n_found]; want_counts: *d_cnt_out[n_pat] = entries per pattern (BNDMq's pattern_hit_counts).  The stream is idle on return.
    int pattern_sets_device(uint64_t n_rec, bool want_counts, unsigned long long **d_off_out, uint32_t **d_pat_out, uint32_t **d_cnt_out, uint64_t *n_found) {
        const uint32_t n_pat = m->n_pat;
        // (the set kernels rank run heads and sum tiles in 32 bits, sets.hip)
        if (found >= (1ull << 32)) return fail(MK_E_UNSUPPORTED, "%llu occurrences in one batch: the per-record pattern sets take at most 2^32 - 1", found);
        const size_t off_bytes = (n_rec + 1) * 8, pat_bytes = ((size_t)found * 4 + 15) & ~(size_t)15;
        const size_t tiles = std::max<size_t>((found + 4095) / 4096, (n_rec + 1 + 4095) / 4096) + 1;
        const size_t cnt_bytes = ((size_t)n_pat * 4 + 15) & ~(size_t)15;
        int rc = ensure_device(&m->d_aux, &m->d_aux_cap, off_bytes + pat_bytes + tiles * 8 + 16 + cnt_bytes);
        if (rc) return rc;
        char *base = (char *)m->d_aux;
        unsigned long long *d_off = (unsigned long long *)base;
        uint32_t *d_pat = (uint32_t *)(base + off_bytes);
        void *d_tile = base + off_bytes + pat_bytes;
        unsigned long long *d_total = (unsigned long long *)(base + off_bytes + pat_bytes + tiles * 8);
        uint32_t *d_cnt = (uint32_t *)(base + off_bytes + pat_bytes + tiles * 8 + 16);
        launch_pattern_sets(m->d_hits, found, n_rec, d_pat, d_off, d_total, d_tile, st);
        unsigned long long total = 0;
        if (hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return fail(MK_E_HIP, "copy failed");
        if (want_counts) {
            if (hipMemsetAsync(d_cnt, 0, cnt_bytes, st) != hipSuccess) return fail(MK_E_HIP, "memset failed");
        }
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "pattern-set kernels failed");
        *n_found = total;
        if (want_counts && total) {
            launch_count_u32(d_pat, total, d_cnt, n_pat, st);
            if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "count kernel failed");
        }
        *d_off_out = d_off, *d_pat_out = d_pat, *d_cnt_out = d_cnt;
        return MK_OK;
    }

    // the same, copied to the host.  found_off == nullptr: only the counts.
    int pattern_sets(uint64_t n_rec, uint64_t *found_off, uint32_t *found_pat, uint64_t found_cap, uint64_t *n_found, uint32_t *counts) {
        const uint32_t n_pat = m->n_pat;
        const size_t off_bytes = (n_rec + 1) * 8;
        unsigned long long *d_off = nullptr;
        uint32_t *d_pat = nullptr, *d_cnt = nullptr;
        int rc = pattern_sets_device(n_rec, counts != nullptr, &d_off, &d_pat, &d_cnt, n_found);
        if (rc) return rc;
        const uint64_t total = *n_found;
        std::vector<uint32_t> v;
        if (counts && total) v.resize(n_pat);
        mark(2);
        if (found_off && hipMemcpy(found_off, d_off, off_bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the set offsets failed");
        const uint64_t n_copy = std::min<uint64_t>(total, found_pat ? found_cap : 0);
        if (n_copy && hipMemcpy(found_pat, d_pat, n_copy * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the sets failed");
        if (!v.empty() && hipMemcpy(v.data(), d_cnt, (size_t)n_pat * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the counts failed");
        mark(3);
        for (size_t i = 0; i < v.size(); ++i) counts[i] += v[i];
        mark(1);
        return MK_OK;
    }
};

}  // namespace

extern "C" {

int mk_extract_single(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                      int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                      mk_counters *c, uint32_t *counts) {
    if (!m || !keep || !c || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_rows) *n_rows = 0;
    if (n_rec == 0) return MK_OK;
    MK_ABI_BEGIN
    DeviceLoop dl(m);
    std::vector<uint8_t> flags(n_rec);
    uint64_t flagged = 0;
    int rc = dl.scan(seq, off, n_rec, logging ? MK_MODE_HITS : MK_MODE_ANY, flags.data(), &flagged);
    if (rc) return rc;
    if (logging) {
        c->nb_records_tot += n_rec;              // :326
        c->nb_bases += off[n_rec] - off[0];      // :327
        c->nb_hits_tot[0] += dl.found;           // :354 / :378
        c->nb_records_hit[0] += flagged;         // :358-360 / :385-387
        // rows in emission order (:338-351 / :369-377) and pattern_hit_counts: AC one per hit (:353), BNDMq one per
        // record and pattern (:380-383) -- BNDMq's emission order is the set order, so its heads are counted in place
        if ((rc = dl.order(m->algo == MK_ALGO_AC))) return rc;
        if ((rc = dl.rows_to_host(0, rows, rows_cap))) return rc;
        if ((rc = dl.pattern_counts(m->algo == MK_ALGO_AC, n_rec, counts))) return rc;
    }
    dl.host_begin();
    for (uint64_t r = 0; r < n_rec; ++r) {  // :400-405
        keep[r] = (uint8_t)((flags[r] != 0) != (invert != 0));
        c->nb_records_extracted += keep[r];
    }
    dl.finish();
    if (n_rows) *n_rows = dl.found;
    if (logging && rows && dl.found > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)dl.found);
    return MK_OK;
    MK_ABI_END
}

// ---- text windows in, records out (r05: FASTA, paired inputs, heads): the raw bytes of a window of every
// input file go to the device as they are (plain text is uploaded, BGZF members are inflated there), are indexed and gathered on
// the device (ingest.hip), then the extract loop runs as in mk_extract_single / mk_extract_paired
}  // extern "C"

namespace mk {
void launch_ingest_count(const uint8_t *d_text, uint64_t n, uint32_t *d_block_cnt, uint32_t *d_total, hipStream_t st);
void launch_ingest_records(const uint8_t *d_text, uint64_t n, const uint32_t *d_block_off, const uint32_t *d_total, uint32_t *d_line_start,
                           uint64_t n_rec, uint32_t *d_rec_start, uint32_t *d_seq_start, uint32_t *d_seq_len, uint32_t *d_status, hipStream_t st);
void launch_ingest_lines(const uint8_t *d_text, uint64_t n, const uint32_t *d_block_off, const uint32_t *d_total, uint32_t *d_line_start, hipStream_t st);
void launch_ingest_offsets(const uint32_t *d_seq_len, uint64_t n_rec, unsigned long long *d_tile, unsigned long long *d_off, hipStream_t st);
void launch_ingest_gather(const uint8_t *d_text, const uint32_t *d_seq_start, const uint32_t *d_seq_len, const unsigned long long *d_off,
                          uint32_t fixed_len, uint64_t n_rec, uint8_t *d_seq, hipStream_t st, uint32_t skip_from = 0xFFFFFFFFu);
void launch_ingest_select(const uint8_t *d_flags, uint32_t invert, const uint32_t *d_rec_start, uint64_t n_rec, uint32_t n_text, uint32_t *d_sel_len,
                          hipStream_t st);
void launch_ingest_fasta_count(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl_block_off, const uint32_t *d_line_start,
                               unsigned long long *d_block64, hipStream_t st);
void launch_ingest_fasta_emit(const uint8_t *d_text, uint64_t n, const uint32_t *d_nl_block_off, const uint32_t *d_line_start,
                              unsigned long long *d_block64, uint8_t *d_seq, uint32_t *d_rec_start, unsigned long long *d_off, hipStream_t st);
void launch_ingest_or_flags(uint8_t *d_flags, const uint8_t *d_other, uint64_t n, hipStream_t st);
uint32_t ingest_block_bytes();
uint32_t ingest_scan_tile();
}  // namespace mk

namespace {

// One input of a window on the device: its text in slot T (m->txt[k]) and what the index kernels made of it
struct WindowSide {
    mk_matcher::TextSlot *T = nullptr;
    uint64_t n_window = 0;  // bytes of text (head + body)
    uint8_t last_byte = '\n';
    uint32_t total_nl = 0;
    uint64_t n_avail = 0;   // whole records found
    uint64_t n_used = 0;    // bytes of the first n records (n = what the call processes)
    uint32_t fixed = 0;     // > 0: every sequence has this length (FASTQ)
    uint64_t seq_total = 0; // FASTA: sequence bytes of all records
    uint32_t *d_block = nullptr, *d_total = nullptr, *d_st = nullptr, *d_line = nullptr, *d_rec_start = nullptr, *d_seq_start = nullptr,
             *d_seq_len = nullptr;
    unsigned long long *d_tile = nullptr, *d_block64 = nullptr, *d_fa_off = nullptr;
};

constexpr uint32_t kBigRecord = 1u << 20;  // records from here on are copied one by one when the kept records are packed

// text of a source -> T->d_text[0, n_window): head, then the body (uploaded, taken from an upload-ahead slot, or inflated from
// BGZF members).  Enqueued on st; the caller synchronises.  *corrupt_member: index of a damaged member (MK_E_CORRUPT).
int window_assemble(mk_matcher *m, mk_codec *codec, mk_window_source &S, WindowSide &W, DeviceLoop &dl) {
    hipStream_t st = dl.st;
    int rc;
    uint64_t body = S.n_text;
    for (uint64_t i = 0; i < S.n_members; ++i) {
        const mk_bgzf_member &b = S.members[i];
        if (b.data_off > S.n_bgzf || b.data_len > S.n_bgzf - b.data_off || b.isize > 65536 || b.out_off != body - S.n_text)
            return fail(MK_E_INVALID_ARG, "mk_extract_window: member %llu lies outside its buffer or its text is not in sequence", (unsigned long long)i);
        body += b.isize;
    }
    if ((S.n_text != 0) + (S.n_members != 0) + (S.n_device_text != 0) > 1)
        return fail(MK_E_INVALID_ARG, "mk_extract_window: a source's body is plain text, BGZF members or device text -- one of them");
    body += S.n_device_text;
    W.n_window = S.n_head + body;
    S.n_window = W.n_window;
    if (W.n_window == 0) return MK_OK;
    if (W.n_window >= 0xFFFFFFF0ull) return fail(MK_E_UNSUPPORTED, "a text window must be shorter than 4 GiB (%llu bytes)", (unsigned long long)W.n_window);
    mk_matcher::TextSlot &T = *W.T;
    // a body that mk_upload_text_ahead has already sent (same pointer, same size): without a head its buffer BECOMES the text
    // buffer, with one it is copied behind the head on the device
    bool ahead = false;
    if (S.n_text) {
        std::lock_guard<std::mutex> lk(m->ahead_mu);
        for (auto &a : m->ahead) {
            if (!a.text || a.text != S.text || ahead) continue;
            if (a.n != S.n_text) {  // (another window out of the same buffer: stale)
                if (hipEventSynchronize(a.ev) != hipSuccess) return fail(MK_E_HIP, "hipEventSynchronize failed");
                a.text = nullptr;
                continue;
            }
            if (hipStreamWaitEvent(st, a.ev, 0) != hipSuccess) return fail(MK_E_HIP, "hipStreamWaitEvent failed");
            if (S.n_head == 0) {
                std::swap(T.d_text, a.d);
                std::swap(T.d_text_cap, a.cap);
            } else {
                if ((rc = ensure_device(&T.d_text, &T.d_text_cap, W.n_window + 64))) return rc;
                if (hipMemcpyAsync((uint8_t *)T.d_text + S.n_head, a.d, S.n_text, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return fail(MK_E_HIP, "copy of the uploaded window failed");
                // (the slot may be refilled by the uploader as soon as it is marked free: let the copy out of it finish first)
                if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "copy of the uploaded window failed");
            }
            a.text = nullptr;
            ahead = true;
        }
    }
    if (!ahead && (rc = ensure_device(&T.d_text, &T.d_text_cap, W.n_window + 64))) return rc;
    if (S.n_head && hipMemcpyAsync(T.d_text, S.head, S.n_head, hipMemcpyHostToDevice, st) != hipSuccess) return fail(MK_E_HIP, "upload of the head failed");
    if (S.n_text && !ahead && hipMemcpyAsync((uint8_t *)T.d_text + S.n_head, S.text, S.n_text, hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(MK_E_HIP, "upload of the text failed");
    if (S.n_device_text && hipMemcpyAsync((uint8_t *)T.d_text + S.n_head, S.device_text, S.n_device_text, hipMemcpyDefault, st) != hipSuccess)
        return fail(MK_E_HIP, "copy of the device text failed");
    if (S.n_members) {
        if (!codec) return fail(MK_E_INVALID_ARG, "mk_extract_window: BGZF members need a codec handle");
        if (codec->device != m->device) return fail(MK_E_INVALID_ARG, "mk_extract_window: the codec and the matcher are on different devices");
        // the members' bytes and their table go up (a fifth of the text), the text is inflated behind the head in place
        std::lock_guard<std::mutex> lock(codec->mu);
        uint64_t in_lo = S.members[0].data_off, in_hi = in_lo;
        std::vector<mkz::Member> part(S.n_members);
        for (uint64_t i = 0; i < S.n_members; ++i) {
            in_lo = std::min<uint64_t>(in_lo, S.members[i].data_off);
            in_hi = std::max<uint64_t>(in_hi, S.members[i].data_off + S.members[i].data_len);
        }
        for (uint64_t i = 0; i < S.n_members; ++i)
            part[i] = mkz::Member{S.members[i].data_off - in_lo, S.n_head + S.members[i].out_off, S.members[i].data_len, S.members[i].isize, S.members[i].crc, 0};
        const uint64_t cn = in_hi - in_lo;
        if ((rc = ensure_device(&codec->d_in, &codec->in_cap, cn + mkz::kPad)) ||
            (rc = ensure_device(&codec->d_aux, &codec->aux_cap, (S.n_members + 1) * sizeof(mkz::Member))) ||
            (rc = ensure_device(&codec->d_len, &codec->len_cap, (S.n_members + 1) * 4ull)))
            return rc;
        if (hipMemcpyAsync(codec->d_in, S.bgzf + in_lo, cn, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemsetAsync((uint8_t *)codec->d_in + cn, 0, mkz::kPad, st) != hipSuccess ||
            hipMemcpyAsync(codec->d_aux, part.data(), S.n_members * sizeof(mkz::Member), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(MK_E_HIP, "upload of the members failed");
        if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "upload of the members failed");
        dl.mark(1);
        mkz::launch_inflate((const uint8_t *)codec->d_in, cn, (const mkz::Member *)codec->d_aux, (uint32_t)S.n_members, (uint8_t *)T.d_text,
                            (int32_t *)codec->d_len, codec->num_cus, st, codec->inflate_kernel);
        mkz::launch_crc_check((const uint8_t *)T.d_text, (const mkz::Member *)codec->d_aux, (uint32_t)S.n_members, (int32_t *)codec->d_len, st);
        std::vector<int32_t> st_words(S.n_members);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(st_words.data(), codec->d_len, S.n_members * 4ull, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "BGZF inflate failed");
        dl.mark(2);
        for (uint64_t i = 0; i < S.n_members; ++i)
            if (st_words[i])
                return fail(MK_E_CORRUPT, st_words[i] > 0 ? "BGZF member %llu: CRC-32 of the inflated text differs from the trailer's (status %d)"
                                                           : "BGZF member %llu does not inflate to its ISIZE (decoder status %d)",
                            (unsigned long long)i, st_words[i]);
    }
    return MK_OK;
}

// line table, whole records and their tables for the text in W.T (enqueued work waited for): W.n_avail, d_rec_start[n_avail + 1],
// FASTQ: d_seq_start / d_seq_len / W.fixed; FASTA: the sequences already lie in d_seq_out with offsets d_fa_off[n_avail + 1].
// *status = 1: this text is not what the device takes (the caller's reader decides what it is).
int window_index(mk_matcher *m, uint32_t format, bool ends_at_record, WindowSide &W, uint8_t *d_seq_out, hipStream_t st, uint32_t *status) {
    mk_matcher::TextSlot &T = *W.T;
    int rc;
    const uint64_t n_text = W.n_window;
    const uint8_t *d_text = (const uint8_t *)T.d_text;
    const uint32_t n_blocks = (uint32_t)((n_text + ingest_block_bytes() - 1) / ingest_block_bytes());
    // d_ing_a: newline count per block | total | status, min, max | (FASTA) u64 per block + 1
    const size_t a_words = (size_t)n_blocks + 8;
    if ((rc = ensure_device(&T.d_ing_a, &T.d_ing_a_cap, a_words * 4 + 16 + ((size_t)n_blocks + 2) * 8))) return rc;
    W.d_block = (uint32_t *)T.d_ing_a;
    W.d_total = W.d_block + n_blocks;
    W.d_st = W.d_total + 1;
    W.d_block64 = (unsigned long long *)(((uintptr_t)(W.d_block + a_words) + 15) & ~(uintptr_t)15);
    launch_ingest_count(d_text, n_text, W.d_block, W.d_total, st);
    uint8_t first_last[2] = {0, 0};
    if (hipMemcpyAsync(&W.total_nl, W.d_total, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&first_last[0], d_text, 1, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&first_last[1], d_text + n_text - 1, 1, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "newline count failed");
    W.last_byte = first_last[1];
    const uint32_t total_nl = W.total_nl;
    if (format == MK_TEXT_FASTA) {
        if (first_last[0] != '>') {  // blank lines or anything else in front of the first header: the host reader's business
            *status = 1;
            return MK_OK;
        }
        // line table | record starts | (u64) sequence offsets
        const size_t max_rec = (size_t)total_nl + 2;  // (a header per line at most)
        const size_t ws = ((size_t)total_nl + 4) * 4 + (max_rec + 2) * 4 + 16 + (max_rec + 2) * 8;
        if ((rc = ensure_device(&T.d_ing_b, &T.d_ing_b_cap, ws))) return rc;
        W.d_line = (uint32_t *)T.d_ing_b;
        W.d_rec_start = W.d_line + total_nl + 4;
        W.d_fa_off = (unsigned long long *)(((uintptr_t)(W.d_rec_start + max_rec + 2) + 15) & ~(uintptr_t)15);
        launch_ingest_lines(d_text, n_text, W.d_block, W.d_total, W.d_line, st);
        if (hipMemsetAsync(W.d_block64 + n_blocks, 0, 8, st) != hipSuccess) return fail(MK_E_HIP, "hipMemsetAsync failed");
        launch_ingest_fasta_count(d_text, n_text, W.d_block, W.d_line, W.d_block64, st);
        unsigned long long totals = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&totals, W.d_block64 + n_blocks, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "FASTA indexing failed");
        const uint64_t heads = totals >> 32;
        W.seq_total = totals & 0xFFFFFFFFull;
        launch_ingest_fasta_emit(d_text, n_text, W.d_block, W.d_line, W.d_block64, d_seq_out, W.d_rec_start, W.d_fa_off, st);
        if (hipGetLastError() != hipSuccess) return fail(MK_E_HIP, "FASTA gather failed to launch");
        // a record is whole once the next header (or the end of the input) has been seen
        W.n_avail = ends_at_record ? heads : (heads ? heads - 1 : 0);
        W.fixed = 0;
        return MK_OK;
    }
    uint64_t n_lines = total_nl;
    if (ends_at_record) {
        n_lines += W.last_byte != '\n' ? 1 : 0;
        if (n_lines % 4 != 0) {  // not whole 4-line records: the caller's reader decides what this text is
            *status = 1;
            return MK_OK;
        }
    }
    const uint64_t n_rec = n_lines / 4;
    W.n_avail = n_rec;
    // workspace: line starts | record starts (+1) | sequence starts | sequence lengths | tile sums
    const size_t n_tiles = n_rec / ingest_scan_tile() + 2;
    const size_t ws = ((size_t)total_nl + 4 + 3 * (n_rec + 2) + 8) * 4 + n_tiles * 8 + 64;
    if ((rc = ensure_device(&T.d_ing_b, &T.d_ing_b_cap, ws))) return rc;
    W.d_line = (uint32_t *)T.d_ing_b;
    W.d_rec_start = W.d_line + total_nl + 4;
    W.d_seq_start = W.d_rec_start + n_rec + 2;
    W.d_seq_len = W.d_seq_start + n_rec + 2;
    W.d_tile = (unsigned long long *)(((uintptr_t)(W.d_seq_len + n_rec + 2) + 15) & ~(uintptr_t)15);
    const uint32_t st_init[3] = {0u, 0xFFFFFFFFu, 0u};
    if (hipMemcpyAsync(W.d_st, st_init, sizeof(st_init), hipMemcpyHostToDevice, st) != hipSuccess) return fail(MK_E_HIP, "copy failed");
    launch_ingest_records(d_text, n_text, W.d_block, W.d_total, W.d_line, n_rec, W.d_rec_start, W.d_seq_start, W.d_seq_len, W.d_st, st);
    uint32_t st_host[3] = {0, 0, 0};
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(st_host, W.d_st, sizeof(st_host), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "record indexing failed");
    if (st_host[0]) {  // some record is not '@' / sequence / '+' / quality of equal length
        *status = 1;
        return MK_OK;
    }
    W.fixed = (n_rec && st_host[1] == st_host[2] && st_host[1] > 0) ? st_host[1] : 0;
    return MK_OK;
}

// the first n records of side W -> m->d_seq / m->d_off, scanned: flags (host + m->d_flags), tuples in m->d_hits (dl.found)
int window_scan(mk_matcher *m, uint32_t format, WindowSide &W, uint64_t n, DeviceLoop &dl, uint32_t mode, uint8_t *flags, uint64_t *flagged, uint64_t *n_seq_out) {
    hipStream_t st = dl.st;
    int rc;
    unsigned long long n_seq = 0;
    if ((rc = ensure_device((void **)&m->d_flags, &m->d_flags_cap, n + 8))) return rc;
    uint32_t fixed = 0;
    if (format == MK_TEXT_FASTA) {
        // (the sequences were compacted into d_seq by the index step; their offsets become the batch's)
        if ((rc = ensure_device((void **)&m->d_off, &m->d_off_cap, (n + 1) * sizeof(uint64_t)))) return rc;
        if (hipMemcpyAsync(m->d_off, W.d_fa_off, (n + 1) * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(&n_seq, W.d_fa_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "copy of the sequence offsets failed");
    } else {
        fixed = W.fixed;
        n_seq = (unsigned long long)n * fixed;
        if ((rc = ensure_device((void **)&m->d_off, &m->d_off_cap, (n + 1) * sizeof(uint64_t)))) return rc;
        if (!fixed) {
            launch_ingest_offsets(W.d_seq_len, n, W.d_tile, (unsigned long long *)m->d_off, st);
            if (hipMemcpyAsync(&n_seq, m->d_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return fail(MK_E_HIP, "offset scan failed");
        }
        launch_ingest_gather((const uint8_t *)W.T->d_text, W.d_seq_start, W.d_seq_len, (const unsigned long long *)m->d_off, fixed, n, m->d_seq, st);
        if (hipGetLastError() != hipSuccess) return fail(MK_E_HIP, "ingest kernels failed to launch");
    }
    m->ragged = !fixed;
    *n_seq_out = n_seq;
    *flagged = 0;
    // FASTA: one record of 4 GiB or more cannot be addressed by mk_hit.pos (as mk_scan_batch refuses it)
    if (n_seq == 0) {  // every sequence is empty: nothing can match
        std::fill(flags, flags + n, 0);
        dl.found = 0;
        if (n && hipMemsetAsync(m->d_flags, 0, n, st) != hipSuccess) return fail(MK_E_HIP, "hipMemsetAsync failed");
        return MK_OK;
    }
    if (format != MK_TEXT_FASTA) return dl.scan_resident(n_seq, n, mode, fixed, flags, flagged);
    // FASTA: the compacted sequences lie in the slot's own buffer -- it stands in for the scan buffer for this scan
    mk_matcher::TextSlot &T = *W.T;
    void *seq = m->d_seq;
    size_t cap = m->d_seq_cap;
    m->d_seq = (uint8_t *)T.d_fa_seq, m->d_seq_cap = T.d_fa_seq_cap;
    rc = dl.scan_resident(n_seq, n, mode, 0, flags, flagged);
    T.d_fa_seq = m->d_seq, T.d_fa_seq_cap = m->d_seq_cap;
    m->d_seq = (uint8_t *)seq, m->d_seq_cap = cap;
    return rc;
}

// text of the kept records of side W (keep flags in d_keep, already final: invert applied by the caller as 0), packed -> host
int window_kept(mk_matcher *m, uint32_t format, WindowSide &W, uint64_t n, const uint8_t *d_keep, const std::vector<uint32_t> &rs, const uint8_t *keep_host,
                mk_window_source &S, DeviceLoop &dl) {
    hipStream_t st = dl.st;
    S.n_kept_bytes = 0;
    if (!n) return MK_OK;
    int rc;
    // lengths by the flags, the offsets scan and the gather kernel of the sequences once more -- into the scan buffer, which has
    // done its work.  (FASTA has no per-record length table of its own: one is carved behind the record starts' u64 table.)
    uint32_t *d_len = W.d_seq_len;
    unsigned long long *d_tile = W.d_tile;
    if (format == MK_TEXT_FASTA) {
        const size_t n_tiles = n / ingest_scan_tile() + 2;
        if ((rc = ensure_device(&m->d_aux, &m->d_aux_cap, (n + 2) * 4 + n_tiles * 8 + 64))) return rc;
        d_len = (uint32_t *)m->d_aux;
        d_tile = (unsigned long long *)(((uintptr_t)(d_len + n + 2) + 15) & ~(uintptr_t)15);
    }
    unsigned long long total = 0;
    launch_ingest_select(d_keep, 0u, W.d_rec_start, n, (uint32_t)W.n_used, d_len, st);
    if ((rc = ensure_device((void **)&m->d_off, &m->d_off_cap, (n + 1) * sizeof(uint64_t)))) return rc;
    launch_ingest_offsets(d_len, n, d_tile, (unsigned long long *)m->d_off, st);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&total, m->d_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "selection of the kept records failed");
    S.n_kept_bytes = total;
    if (total > S.kept_cap) return fail(MK_E_CAPACITY, "the kept records take %llu bytes", total);
    if (!total) return MK_OK;
    if ((rc = ensure_device((void **)&m->d_seq, &m->d_seq_cap, total + 64))) return rc;
    launch_ingest_gather((const uint8_t *)W.T->d_text, W.d_rec_start, d_len, (const unsigned long long *)m->d_off, 0, n, m->d_seq, st, kBigRecord);
    // chromosome-sized records: one device copy each
    uint64_t at = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (!keep_host[r]) continue;
        const uint64_t len = (r + 1 < n ? rs[r + 1] : W.n_used) - rs[r];
        if (len >= kBigRecord && hipMemcpyAsync(m->d_seq + at, (const uint8_t *)W.T->d_text + rs[r], len, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(MK_E_HIP, "copy of a kept record failed");
        at += len;
    }
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(S.kept, m->d_seq, total, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "copy of the kept records failed");
    return MK_OK;
}

}  // namespace

extern "C" {

int mk_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(MK_E_INVALID_ARG, "null argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    return MK_OK;
}

void mk_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int mk_upload_text_ahead(mk_matcher *m, const uint8_t *text, uint64_t n_text) {
    if (!m || (n_text && !text)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_text == 0 || n_text >= 0xFFFFFFF0ull) return MK_OK;  // (nothing to do / the call itself will refuse it)
    MK_ABI_BEGIN
    if (hipSetDevice(m->device) != hipSuccess) return fail(MK_E_HIP, "hipSetDevice failed");
    std::lock_guard<std::mutex> lk(m->ahead_mu);
    if (!m->stream_ahead && hipStreamCreateWithFlags(&m->stream_ahead, hipStreamNonBlocking) != hipSuccess)
        return fail(MK_E_HIP, "hipStreamCreate failed");
    mk_matcher::AheadSlot *slot = nullptr;
    // a slot that still holds an earlier window from this very buffer is stale (the host did not come back for it and is now
    // refilling the buffer): its copy must have finished reading before the new bytes are trusted -- the caller has of course
    // already overwritten them, so all that is left to do is to wait and take the slot over
    for (auto &a : m->ahead)
        if (a.text == text) {
            if (hipEventSynchronize(a.ev) != hipSuccess) return fail(MK_E_HIP, "hipEventSynchronize failed");
            a.text = nullptr;
            slot = &a;
        }
    for (auto &a : m->ahead)
        if (!a.text && !slot) slot = &a;
    if (!slot) return MK_OK;  // every slot waits already: this window will upload itself
    if (!slot->ev && hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) return fail(MK_E_HIP, "hipEventCreate failed");
    int rc = ensure_device(&slot->d, &slot->cap, n_text + 64);
    if (rc) return rc;
    if (hipMemcpyAsync(slot->d, text, n_text, hipMemcpyHostToDevice, m->stream_ahead) != hipSuccess ||
        hipEventRecord(slot->ev, m->stream_ahead) != hipSuccess)
        return fail(MK_E_HIP, "upload of the next text window failed");
    slot->text = text;
    slot->n = n_text;
    return MK_OK;
    MK_ABI_END
}

int mk_extract_window(mk_matcher *m, mk_codec *codec, uint32_t format, uint32_t n_sources, mk_window_source *src, int logging, int invert,
                      uint64_t rec_cap, uint64_t *n_rec_out, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *c,
                      uint32_t *counts, uint32_t *status) {
    if (!m || !src || !n_rec_out || !status || !c || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_sources < 1 || n_sources > 2) return fail(MK_E_INVALID_ARG, "mk_extract_window: one source (single file) or two (paired files)");
    if (format > MK_TEXT_FASTA) return fail(MK_E_INVALID_ARG, "mk_extract_window: unknown text format %u", format);
    *n_rec_out = 0;
    *status = 0;
    if (n_rows) *n_rows = 0;
    for (uint32_t k = 0; k < n_sources; ++k) {
        mk_window_source &S = src[k];
        S.n_window = S.n_used = S.n_tail = S.n_kept_bytes = S.n_rec_seen = 0;
        if ((S.n_head && !S.head) || (S.n_text && !S.text) || (S.n_members && (!S.bgzf || !S.members)) || (S.n_device_text && !S.device_text) ||
            (S.kept_cap && !S.kept) ||
            (S.all_cap && !S.all) || (S.tail_cap && !S.tail))
            return fail(MK_E_INVALID_ARG, "mk_extract_window: a size without its buffer in source %u", k);
    }
    MK_ABI_BEGIN
    if (hipSetDevice(m->device) != hipSuccess) return fail(MK_E_HIP, "hipSetDevice failed");
    DeviceLoop dl(m);
    hipStream_t st = dl.st;
    int rc;
    const bool paired = n_sources == 2;
    const bool ac = m->algo == MK_ALGO_AC;
    WindowSide W[2];
    // ---- the text of every source on the device, then what the index kernels make of it
    uint64_t most = 0;
    for (uint32_t k = 0; k < n_sources; ++k) {
        W[k].T = &m->txt[k];
        if ((rc = window_assemble(m, codec, src[k], W[k], dl))) return rc;
        most = std::max(most, W[k].n_window);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "upload of the window failed");
    dl.mark(1);
    if ((rc = ensure_device((void **)&m->d_seq, &m->d_seq_cap, most + 64))) return rc;
    uint64_t n = ~0ull;
    for (uint32_t k = 0; k < n_sources; ++k) {
        if (W[k].n_window == 0) {  // nothing of this input in the window
            W[k].n_avail = 0;
        } else {
            uint8_t *fa_seq = nullptr;
            if (format == MK_TEXT_FASTA) {  // the index step compacts the sequences already: every input has a buffer of its own for them
                if ((rc = ensure_device(&W[k].T->d_fa_seq, &W[k].T->d_fa_seq_cap, W[k].n_window + 64))) return rc;
                fa_seq = (uint8_t *)W[k].T->d_fa_seq;
            }
            if ((rc = window_index(m, format, src[k].ends_at_record != 0, W[k], fa_seq, st, status))) return rc;
            if (*status) return MK_OK;
        }
        src[k].n_rec_seen = W[k].n_avail;
        n = std::min(n, W[k].n_avail);
    }
    *n_rec_out = n;
    if (n > rec_cap || !keep) return fail(MK_E_CAPACITY, "%llu records in the window, room for %llu", (unsigned long long)n, (unsigned long long)rec_cap);
    // ---- record tables -> host; bytes of the n records per source; the tails
    std::vector<uint32_t> rs[2];
    for (uint32_t k = 0; k < n_sources; ++k) {
        rs[k].assign(n + 1, 0);
        if (W[k].n_window && hipMemcpyAsync(rs[k].data(), W[k].d_rec_start, (n + 1) * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
            return fail(MK_E_HIP, "copy of the record table failed");
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "copy of the record table failed");
    for (uint32_t k = 0; k < n_sources; ++k) {
        mk_window_source &S = src[k];
        // all of a window that ends at a record end belongs to its records (a last line without '\n' included)
        W[k].n_used = (n == W[k].n_avail && S.ends_at_record) ? W[k].n_window : (W[k].n_window ? rs[k][n] : 0);
        rs[k][n] = (uint32_t)W[k].n_used;
        S.n_used = W[k].n_used;
        S.n_tail = W[k].n_window - W[k].n_used;
        if (S.rec_start)
            for (uint64_t r = 0; r <= n; ++r) S.rec_start[r] = rs[k][r];
        if (S.tail || S.tail_cap) {
            if (S.n_tail > S.tail_cap) return fail(MK_E_CAPACITY, "mk_extract_window: the text behind the window's records takes %llu bytes", (unsigned long long)S.n_tail);
            if (S.n_tail && hipMemcpyAsync(S.tail, (const uint8_t *)W[k].T->d_text + W[k].n_used, S.n_tail, hipMemcpyDeviceToHost, st) != hipSuccess)
                return fail(MK_E_HIP, "download of the tail failed");
        }
        if (S.all) {
            if (W[k].n_window > S.all_cap) return fail(MK_E_CAPACITY, "mk_extract_window: the window's text takes %llu bytes", (unsigned long long)W[k].n_window);
            if (W[k].n_window && hipMemcpyAsync(S.all, W[k].T->d_text, W[k].n_window, hipMemcpyDeviceToHost, st) != hipSuccess)
                return fail(MK_E_HIP, "download of the text failed");
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "download of the text failed");
    dl.mark(1);
    if (n == 0) return MK_OK;
    // ---- the loop bodies (src/cmd_extract.rs:321-406 single, :463-612 paired) on the first n records of every source
    const uint32_t mode = logging ? MK_MODE_HITS : MK_MODE_ANY;
    std::vector<uint8_t> f[2];
    uint64_t flagged[2] = {0, 0}, n_seq[2] = {0, 0};
    unsigned long long found[2] = {0, 0};
    for (uint32_t k = 0; k < n_sources; ++k) {
        f[k].resize(n);
        if ((rc = window_scan(m, format, W[k], n, dl, mode, f[k].data(), &flagged[k], &n_seq[k]))) return rc;
        found[k] = dl.found;
        if (paired && k == 0) {
            // mate 1's flags (the kept text is selected on the device) and tuples wait while mate 2 is scanned
            if ((rc = ensure_device((void **)&m->d_flags2, &m->d_flags2_cap, n + 8))) return rc;
            if (hipMemcpyAsync(m->d_flags2, m->d_flags, n, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(MK_E_HIP, "copy of the flags failed");
            if (logging && found[0]) {
                if ((rc = ensure_device(&m->d_pair, &m->d_pair_cap, found[0] * sizeof(mk_hit)))) return rc;
                if (hipMemcpyAsync(m->d_pair, m->d_hits, found[0] * sizeof(mk_hit), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return fail(MK_E_HIP, "copy of the first mate's tuples failed");
            }
            if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "copy of the first mate's results failed");
        }
    }
    uint64_t total_rows = 0;
    if (logging && !paired) {
        c->nb_records_tot += n;
        c->nb_bases += n_seq[0];
        c->nb_hits_tot[0] += found[0];
        c->nb_records_hit[0] += flagged[0];
        if ((rc = dl.order(ac))) return rc;
        if ((rc = dl.rows_to_host(0, rows, rows_cap))) return rc;
        if ((rc = dl.pattern_counts(ac, n, counts))) return rc;
        total_rows = found[0];
    } else if (logging) {
        c->nb_records_tot += 2 * n;  // :472
        c->nb_bases += n_seq[0] + n_seq[1];
        c->nb_hits_tot[0] += found[0];
        c->nb_hits_tot[1] += found[1];
        c->nb_records_hit[0] += flagged[0];
        c->nb_records_hit[1] += flagged[1];
        total_rows = found[0] + found[1];
        const unsigned long long n1 = found[0], n2 = found[1];
        if (!ac && std::max(n_seq[0], n_seq[1]) >= (1ull << 31))  // (BNDMq's pair order keeps the mate in bit 31 of the position)
            return fail(MK_E_UNSUPPORTED, "mk_extract_window: paired windows of 2 GiB of sequence or more under BNDMq");
        if (total_rows) {
            // one list: mate 2's tuples (already in the scan buffer), then mate 1's, the mate marked inside a key field
            if ((n1 + n2) * sizeof(mk_hit) > m->d_hits_cap) {  // grow the scan buffer, keeping mate 2's tuples
                void *bigger = nullptr;
                size_t cap = 0;
                if ((rc = ensure_device(&bigger, &cap, (n1 + n2) * sizeof(mk_hit)))) return rc;
                if (n2 && hipMemcpy(bigger, m->d_hits, n2 * sizeof(mk_hit), hipMemcpyDeviceToDevice) != hipSuccess) {
                    (void)hipFree(bigger);
                    return fail(MK_E_HIP, "copy of the second mate's tuples failed");
                }
                if (m->d_hits) (void)hipFree(m->d_hits);
                m->d_hits = (mk_hit *)bigger;
                m->d_hits_cap = cap;
            }
            if (n1 && hipMemcpyAsync(m->d_hits + n2, m->d_pair, n1 * sizeof(mk_hit), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return fail(MK_E_HIP, "copy of the first mate's tuples failed");
            launch_pair_mark(m->d_hits, n2, 1, ac, st);
            launch_pair_mark(m->d_hits + n2, n1, 0, ac, st);
            dl.found = total_rows;
            const uint64_t bound = m->last_n_rec;
            if (ac) m->last_n_rec = 2 * n;  // record' = 2 * record + mate: the bins of the ordering
            rc = dl.order(ac);
            m->last_n_rec = bound;
            if (rc) return rc;
            if ((rc = dl.pair_rows_to_host(ac, rows, rows_cap))) return rc;
            if ((rc = dl.pair_counts(ac, counts))) return rc;
        }
    }
    // ---- keep (single :400-405, paired :600-606), on the host for the caller and on the device for the kept records' text
    dl.host_begin();
    for (uint64_t r = 0; r < n; ++r) {
        const bool hit = f[0][r] || (paired && f[1][r]);
        keep[r] = (uint8_t)(hit != (invert != 0));
        c->nb_records_extracted += (paired ? 2u : 1u) * keep[r];
    }
    dl.mark(1);
    bool want_kept = false;
    for (uint32_t k = 0; k < n_sources; ++k) want_kept = want_kept || src[k].kept != nullptr || src[k].kept_cap != 0;
    if (want_kept) {
        if ((rc = ensure_device((void **)&m->d_flags2, &m->d_flags2_cap, n + 8))) return rc;
        if (hipMemcpyAsync(m->d_flags2, keep, n, hipMemcpyHostToDevice, st) != hipSuccess) return fail(MK_E_HIP, "upload of the keep flags failed");
        int cap_rc = MK_OK;
        for (uint32_t k = 0; k < n_sources; ++k) {
            if (!src[k].kept && !src[k].kept_cap) continue;
            rc = window_kept(m, format, W[k], n, m->d_flags2, rs[k], keep, src[k], dl);
            if (rc == MK_E_CAPACITY) cap_rc = rc;  // (every source reports its need before the call returns)
            else if (rc) return rc;
        }
        if (cap_rc) return cap_rc;
    }
    dl.finish();
    if (n_rows) *n_rows = total_rows;
    if (logging && rows && total_rows > rows_cap) return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)total_rows);
    return MK_OK;
    MK_ABI_END
}

// (v4 / v5 entry points, kept: one FASTQ source whose window ends at a record end / one bgzip'ed FASTQ source with a head)
int mk_extract_fastq_text(mk_matcher *m, const uint8_t *text, uint64_t n_text, int logging, int invert, uint64_t rec_cap, uint64_t *n_rec_out,
                          uint64_t *rec_start, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *c, uint32_t *counts,
                          uint32_t *status) {
    if (!m || !n_rec_out || !status || !c || (logging && !counts) || (n_text && !text)) return fail(MK_E_INVALID_ARG, "null argument");
    *n_rec_out = 0;
    *status = 0;
    if (n_rows) *n_rows = 0;
    if (n_text == 0) return MK_OK;
    mk_window_source S;
    memset(&S, 0, sizeof(S));
    S.text = text, S.n_text = n_text, S.ends_at_record = 1, S.rec_start = rec_start;
    if (!rec_start) rec_cap = 0;
    return mk_extract_window(m, nullptr, MK_TEXT_FASTQ, 1, &S, logging, invert, rec_cap, n_rec_out, keep, rows, rows_cap, n_rows, c, counts, status);
}

int mk_extract_fastq_bgzf(mk_matcher *m, mk_codec *codec, const uint8_t *head, uint64_t n_head, const uint8_t *bgzf, uint64_t n_bgzf,
                          const mk_bgzf_member *members, uint64_t n_members, int last, mk_window_text *io, int logging, int invert, uint64_t rec_cap, uint64_t *n_rec_out, uint64_t *rec_start, uint8_t *keep,
                          mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *c, uint32_t *counts, uint32_t *status) {
    if (!m || !codec || !io || !n_rec_out || !status || !c || (logging && !counts) || (n_head && !head) ||
        (n_members && (!bgzf || !members)))
        return fail(MK_E_INVALID_ARG, "null argument");
    io->n_text = io->n_used = io->n_tail = io->n_kept_bytes = 0;
    if (!io->text && (!io->tail || (io->kept_cap && !io->kept))) return fail(MK_E_INVALID_ARG, "mk_extract_fastq_bgzf: neither a text buffer nor tail / kept buffers");
    mk_window_source S;
    memset(&S, 0, sizeof(S));
    S.head = head, S.n_head = n_head, S.bgzf = bgzf, S.n_bgzf = n_bgzf, S.members = members, S.n_members = n_members;
    S.ends_at_record = last ? 1 : 0;
    S.rec_start = rec_start;
    if (!rec_start) rec_cap = 0;
    std::vector<uint8_t> dummy(1);
    if (io->text) {
        S.all = io->text, S.all_cap = io->text_cap;
    } else {
        S.tail = io->tail, S.tail_cap = io->tail_cap;
        S.kept = io->kept ? io->kept : dummy.data(), S.kept_cap = io->kept_cap;
    }
    const int rc = mk_extract_window(m, codec, MK_TEXT_FASTQ, 1, &S, logging, invert, rec_cap, n_rec_out, keep, rows, rows_cap, n_rows, c, counts, status);
    io->n_text = S.n_window, io->n_used = S.n_used, io->n_tail = io->text ? 0 : S.n_tail, io->n_kept_bytes = S.n_kept_bytes;
    // (v5 contract: a window without one whole record is the caller's reader's business -- unless nothing follows it)
    if (rc == MK_OK && !*status && S.n_window && S.n_used == 0) *status = last ? 0u : 1u;
    return rc;
}

// ---- `tag` on a window of BAM text that stays on the device (r05, ABI v7; kernels: bam.hip) ---------------------------------------
}  // extern "C"

namespace mk {
void launch_bam_find(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, uint32_t *d_start, hipStream_t st);
void launch_bam_walk_count(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, const uint32_t *d_start, uint32_t *d_land,
                           uint32_t *d_count, hipStream_t st);
void launch_bam_walk_emit(const uint8_t *d_text, uint64_t n, uint32_t piece, uint32_t n_pieces, const uint32_t *d_start, const uint32_t *d_base,
                          uint32_t *d_rec_off, uint32_t *d_rec_len, uint32_t *d_seq_start, uint32_t *d_seq_len, uint32_t *d_st, hipStream_t st);
void launch_bam_unpack(const uint8_t *d_text, const uint32_t *d_seq_start, const uint32_t *d_seq_len, const unsigned long long *d_off, uint32_t fixed_len,
                       uint64_t n_rec, uint8_t *d_seq, hipStream_t st);
void launch_bam_taglen(const uint8_t *d_text, const uint32_t *d_rec_off, const uint32_t *d_rec_len, const uint32_t *d_seq_start, const uint32_t *d_seq_len,
                       const unsigned long long *d_found_off, const uint32_t *d_found_pat, const uint32_t *d_pat_off, const uint8_t *d_pat_bytes, uint64_t n_rec,
                       uint32_t filter_matching, uint32_t invert, uint32_t tag0, uint32_t tag1, uint8_t *d_keep, uint32_t *d_out_len, uint32_t *d_ex_off,
                       uint32_t *d_st, hipStream_t st);
void launch_bam_emit(const uint8_t *d_text, const uint32_t *d_rec_off, const uint32_t *d_rec_len, const uint32_t *d_out_len, const unsigned long long *d_out_off,
                     const unsigned long long *d_found_off, const uint32_t *d_found_pat, const uint8_t *d_pat_bytes, const uint32_t *d_pat_off,
                     const uint32_t *d_ex_off, uint64_t n_rec, uint32_t tag0, uint32_t tag1, uint8_t *d_out, hipStream_t st);
void launch_bam_names(const uint8_t *d_text, const uint32_t *d_rec_off, const uint8_t *d_flags, uint64_t n_rec, uint32_t *d_name_start, uint32_t *d_name_len,
                      hipStream_t st);
}  // namespace mk

namespace {

constexpr int kBamProofRounds = 8;  // walks of the record chain before a window is left to the host reader

// the record chain of text[0, n) -> W's tables (d_rec_start = record offsets, d_seq_start, d_seq_len; rec_len behind them):
// *n_rec records covering *n_used bytes; *fixed > 0: every sequence has this length.  *status |= 1: not for the device.
int bam_index(mk_matcher *m, WindowSide &W, hipStream_t st, uint64_t *n_rec, uint64_t *n_used, uint32_t **d_rec_len, uint32_t *fixed, uint32_t *status) {
    mk_matcher::TextSlot &T = *W.T;
    const uint64_t n = W.n_window;
    const uint8_t *d_text = (const uint8_t *)T.d_text;
    const uint32_t piece = m->bam_piece ? m->bam_piece : 65536u;
    const uint32_t n_pieces = (uint32_t)((n + piece - 1) / piece);
    int rc;
    // d_ing_a: start | land | count | base (u32 per piece each) | st[4]
    if ((rc = ensure_device(&T.d_ing_a, &T.d_ing_a_cap, ((size_t)n_pieces * 4 + 8) * 4))) return rc;
    uint32_t *d_start = (uint32_t *)T.d_ing_a, *d_land = d_start + n_pieces, *d_count = d_land + n_pieces, *d_base = d_count + n_pieces,
             *d_st = d_base + n_pieces;
    launch_bam_find(d_text, n, piece, n_pieces, d_start, st);
    std::vector<uint32_t> start(n_pieces), land(n_pieces), count(n_pieces);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(start.data(), d_start, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
        return fail(MK_E_HIP, "BAM record search failed");
    bool proved = false;
    std::vector<uint8_t> valid(n_pieces);
    for (int round = 0; round < kBamProofRounds && !proved; ++round) {
        launch_bam_walk_count(d_text, n, piece, n_pieces, d_start, d_land, d_count, st);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(land.data(), d_land, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(count.data(), d_count, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "BAM record walk failed");
        // Piece 0 starts at a record start (the caller's contract); a walk that lands on the next piece's start makes that one too.
        // One pass from the left: a start that is not met is replaced by the landing -- which is right, because everything to its
        // left is -- and its own landing is then unknown (valid = 0) until the next walk, unless the new start lies behind the
        // piece altogether (a record longer than a piece: nothing starts in it, the walk is the identity).
        std::fill(valid.begin(), valid.end(), 1);
        proved = true;
        for (uint32_t p = 0; p < n_pieces; ++p) {
            if (!valid[p]) {
                const uint64_t stop = p + 1 < n_pieces ? (uint64_t)(p + 1) * piece : n;
                if (start[p] != 0xFFFFFFFFu && start[p] >= stop) {
                    land[p] = start[p], count[p] = 0, valid[p] = 1;
                } else {
                    proved = false;
                    continue;
                }
            }
            if (count[p] & 0x80000000u) {  // the text ends inside the record at land[p]: nothing whole follows it
                for (uint32_t q = p + 1; q < n_pieces; ++q) start[q] = land[q] = land[p], count[q] = 0;
                break;
            }
            if (p + 1 < n_pieces && land[p] != 0xFFFFFFFFu && start[p + 1] != land[p]) {
                start[p + 1] = land[p];
                valid[p + 1] = 0;
            }
        }
        if (hipMemcpyAsync(d_start, start.data(), (size_t)n_pieces * 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail(MK_E_HIP, "copy failed");
    }
    for (uint32_t p = 0; p < n_pieces; ++p) proved = proved && start[p] != 0xFFFFFFFFu;
    if (!proved) {
        *status |= 1;
        return MK_OK;
    }
    std::vector<uint32_t> base(n_pieces);
    uint64_t total = 0;
    for (uint32_t p = 0; p < n_pieces; ++p) {
        base[p] = (uint32_t)total;
        total += count[p] & 0x7FFFFFFFu;
    }
    *n_rec = total;
    *n_used = n_pieces ? land[n_pieces - 1] : 0;
    if (total >= 0xFFFFFFF0ull) return fail(MK_E_UNSUPPORTED, "%llu records in one BAM window", (unsigned long long)total);
    // tables: record offsets | sequence starts | sequence lengths | record lengths | out lengths | existing-tag offsets (u32, n + 2 each) | out offsets (u64) | tiles (u64)
    const size_t n_tiles = total / ingest_scan_tile() + 2;
    if ((rc = ensure_device(&T.d_ing_b, &T.d_ing_b_cap, 6 * (total + 2) * 4 + 16 + (total + 2) * 8 + n_tiles * 8 + 64))) return rc;
    W.d_rec_start = (uint32_t *)T.d_ing_b;
    W.d_seq_start = W.d_rec_start + total + 2;
    W.d_seq_len = W.d_seq_start + total + 2;
    *d_rec_len = W.d_seq_len + total + 2;
    const uint32_t st_init[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
    if (hipMemcpyAsync(d_base, base.data(), (size_t)n_pieces * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_st, st_init, sizeof(st_init), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(MK_E_HIP, "copy failed");
    launch_bam_walk_emit(d_text, n, piece, n_pieces, d_start, d_base, W.d_rec_start, *d_rec_len, W.d_seq_start, W.d_seq_len, d_st, st);
    uint32_t st_host[4] = {0, 0, 0, 0};
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(st_host, d_st, sizeof(st_host), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "BAM record indexing failed");
    if (st_host[0]) *status |= 1;  // a record the serial parser refuses ("truncated file"): the host reader words it
    *fixed = (total && st_host[1] == st_host[2] && st_host[1] > 0) ? st_host[1] : 0;
    W.d_st = d_st;
    return MK_OK;
}

double ms_since(std::chrono::steady_clock::time_point &t) {
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - t).count();
    t = now;
    return ms;
}

}  // namespace

extern "C" {

int mk_matcher_set_bam_piece(mk_matcher *m, uint32_t piece_bytes) {
    if (!m || (piece_bytes && piece_bytes < 64)) return fail(MK_E_INVALID_ARG, "mk_matcher_set_bam_piece: handle / a piece of at least 64 bytes");
    m->bam_piece = piece_bytes;
    return MK_OK;
}

int mk_tag_bam_window(mk_matcher *m, mk_codec *codec, mk_bam_window *w, int logging, mk_counters *c, uint32_t *counts, uint32_t *status) {
    if (!m || !codec || !w || !c || !status || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if ((w->n_head && !w->head) || (w->n_members && (!w->bgzf || !w->members)) || (w->tail_cap && !w->tail) || (w->out_cap && !w->out) ||
        (logging && w->rows_cap && (!w->rows || !w->row_name)) || (w->names_cap && !w->names))
        return fail(MK_E_INVALID_ARG, "mk_tag_bam_window: a size without its buffer");
    const uint32_t bb = w->block_bytes ? w->block_bytes : mkz::kMaxBlockBytes;
    if (bb > mkz::kMaxBlockBytes) return fail(MK_E_INVALID_ARG, "mk_tag_bam_window: block_bytes %u > %u", bb, mkz::kMaxBlockBytes);
    w->n_window = w->n_used = w->n_tail = w->n_rec = w->n_kept = w->out_text_bytes = w->out_len = w->n_rows = w->n_names_bytes = 0;
    for (float &x : w->ms) x = 0;
    *status = 0;
    MK_ABI_BEGIN
    if (hipSetDevice(m->device) != hipSuccess) return fail(MK_E_HIP, "hipSetDevice failed");
    DeviceLoop dl(m);
    hipStream_t st = dl.st;
    int rc;
    auto t = std::chrono::steady_clock::now();
    g_alloc_ms = 0, g_free_ms = 0;
    struct AllocMs {  // (device buffers grown inside the call: part of the phases above, reported on its own as ms[7])
        float *out;
        ~AllocMs() { *out = (float)g_alloc_ms; }
    } alloc_ms{&w->ms[7]};
    // ---- the text: head, then the members inflated behind it (window_assemble: upload, inflate, CRC-32 / ISIZE of every member)
    WindowSide W;
    W.T = &m->txt[0];
    mk_window_source S;
    memset(&S, 0, sizeof(S));
    S.head = w->head, S.n_head = w->n_head, S.bgzf = w->bgzf, S.n_bgzf = w->n_bgzf, S.members = w->members, S.n_members = w->n_members;
    if ((rc = window_assemble(m, codec, S, W, dl))) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "upload of the window failed");
    w->ms[1] = (float)ms_since(t);  // (upload + inflate: batch_ms splits them)
    w->ms[0] = m->batch_ms[0];
    w->ms[1] -= w->ms[0];
    w->n_window = W.n_window;
    const uint64_t n_text = W.n_window;
    if (n_text == 0) return MK_OK;
    // ---- the record chain
    uint64_t n = 0, n_used = 0;
    uint32_t *d_rec_len = nullptr, fixed = 0;
    if ((rc = bam_index(m, W, st, &n, &n_used, &d_rec_len, &fixed, status))) return rc;
    if (*status) return MK_OK;
    if (w->last && n_used != n_text) {  // the file ends inside a record
        *status = 8;
        return MK_OK;
    }
    w->n_rec = n, w->n_used = n_used, w->n_tail = n_text - n_used;
    if (w->n_tail > w->tail_cap) return fail(MK_E_CAPACITY, "mk_tag_bam_window: the text behind the window's records takes %llu bytes", (unsigned long long)w->n_tail);
    if (w->n_tail && hipMemcpyAsync(w->tail, (const uint8_t *)W.T->d_text + n_used, w->n_tail, hipMemcpyDeviceToHost, st) != hipSuccess)
        return fail(MK_E_HIP, "download of the tail failed");
    if (hipStreamSynchronize(st) != hipSuccess) return fail(MK_E_HIP, "download of the tail failed");
    if (w->on_tail) w->on_tail(w->on_tail_ctx, w->tail, w->n_tail);
    w->ms[2] = (float)ms_since(t);
    if (n == 0) return MK_OK;
    uint32_t *d_out_len = d_rec_len + n + 2, *d_ex_off = d_out_len + n + 2;
    unsigned long long *d_out_off = (unsigned long long *)(((uintptr_t)(d_ex_off + n + 2) + 15) & ~(uintptr_t)15);
    unsigned long long *d_tile = d_out_off + n + 2;
    // ---- sequences -> the scan buffer, scan, emission order, pattern sets
    unsigned long long n_seq = (unsigned long long)n * fixed;
    if ((rc = ensure_device((void **)&m->d_flags, &m->d_flags_cap, n + 8)) || (rc = ensure_device((void **)&m->d_off, &m->d_off_cap, (n + 1) * sizeof(uint64_t))))
        return rc;
    if (!fixed) {
        launch_ingest_offsets(W.d_seq_len, n, d_tile, (unsigned long long *)m->d_off, st);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&n_seq, m->d_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "offset scan failed");
    }
    if ((rc = ensure_device((void **)&m->d_seq, &m->d_seq_cap, n_seq + 64))) return rc;
    launch_bam_unpack((const uint8_t *)W.T->d_text, W.d_seq_start, W.d_seq_len, (const unsigned long long *)m->d_off, fixed, n, m->d_seq, st);
    if (hipGetLastError() != hipSuccess) return fail(MK_E_HIP, "sequence unpacking failed to launch");
    m->ragged = !fixed;
    std::vector<uint8_t> flags(n);
    uint64_t flagged = 0;
    const bool ac = m->algo == MK_ALGO_AC;
    if (n_seq == 0) {
        dl.found = 0;
        if (hipMemsetAsync(m->d_flags, 0, n, st) != hipSuccess) return fail(MK_E_HIP, "hipMemsetAsync failed");
    } else if ((rc = dl.scan_resident(n_seq, n, MK_MODE_HITS, fixed, flags.data(), &flagged))) {
        return rc;
    }
    bool set_order = false;
    uint64_t n_rows = 0;
    // (counters of this window: added to the caller's only when the window is done -- a refused or repeated window counts nothing)
    mk_counters lc;
    memset(&lc, 0, sizeof(lc));
    std::vector<uint32_t> lcounts(logging ? m->n_pat : 0, 0);
    if (logging) {  // src/cmd_tag.rs:400-416, :443-451
        lc.nb_hits_tot[0] = dl.found;
        lc.nb_records_tot = n;
        lc.nb_bases = n_seq;
        lc.nb_records_hit[0] = flagged;
        n_rows = dl.found;
        if ((rc = dl.order(ac))) return rc;
        if ((rc = dl.rows_to_host(0, w->rows, w->rows_cap))) return rc;
        set_order = !ac;
        if (ac && (rc = dl.pattern_counts(true, n, lcounts.data()))) return rc;
        // the names of the records with a hit, NUL-terminated, in record order; a row finds its record's by a walk along both
        if (flagged) {
            uint32_t *d_name_start = d_out_len, *d_name_len = (uint32_t *)d_out_off;  // (free until the tag step)
            unsigned long long *d_name_off = (unsigned long long *)m->d_off;          // (the scan is done with the sequence offsets)
            unsigned long long total = 0;
            launch_bam_names((const uint8_t *)W.T->d_text, W.d_rec_start, m->d_flags, n, d_name_start, d_name_len, st);
            launch_ingest_offsets(d_name_len, n, d_tile, d_name_off, st);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&total, d_name_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                return fail(MK_E_HIP, "selection of the names failed");
            w->n_names_bytes = total;
            if (total <= w->names_cap && n_rows <= w->rows_cap) {
                // (the unpacked sequences have been scanned: their buffer holds the names now)
                if ((rc = ensure_device((void **)&m->d_seq, &m->d_seq_cap, total + 64))) return rc;
                launch_ingest_gather((const uint8_t *)W.T->d_text, d_name_start, d_name_len, d_name_off, 0, n, m->d_seq, st);
                if (hipGetLastError() != hipSuccess || hipMemcpyAsync(w->names, m->d_seq, total, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess)
                    return fail(MK_E_HIP, "download of the names failed");
                // rows are in record order (both emission orders are record-major): r = the flagged record whose name starts at `at`
                uint64_t r = 0, at = 0;
                while (r < n && !flags[r]) ++r;
                for (uint64_t k = 0; k < n_rows; ++k) {
                    const uint64_t rec = w->rows[k].rec;
                    while (r < rec && at < total) {
                        at += strlen((const char *)w->names + at) + 1;
                        ++r;
                        while (r < n && !flags[r]) ++r;
                    }
                    w->row_name[k] = at;
                }
            }
        }
    }
    w->n_rows = n_rows;
    if (!set_order && (rc = dl.order(false))) return rc;
    unsigned long long *d_found_off = nullptr;
    uint32_t *d_found_pat = nullptr, *d_cnt = nullptr;
    uint64_t n_found = 0;
    if ((rc = dl.pattern_sets_device(n, logging && !ac, &d_found_off, &d_found_pat, &d_cnt, &n_found))) return rc;
    if (logging && !ac && n_found) {  // BNDMq: one count per record and pattern (:431-433)
        if (hipMemcpy(lcounts.data(), d_cnt, (size_t)m->n_pat * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(MK_E_HIP, "copy of the counts failed");
    }
    w->ms[3] = (float)ms_since(t);
    // ---- keep, tag, pack
    if ((rc = ensure_device((void **)&m->d_flags2, &m->d_flags2_cap, n + 8))) return rc;
    if (hipMemsetAsync(W.d_st, 0, 4, st) != hipSuccess) return fail(MK_E_HIP, "hipMemsetAsync failed");
    launch_bam_taglen((const uint8_t *)W.T->d_text, W.d_rec_start, d_rec_len, W.d_seq_start, W.d_seq_len, d_found_off, d_found_pat, m->d_pat_off, m->d_pat_bytes, n,
                      w->filter_matching != 0, w->invert != 0, w->tag[0], w->tag[1], m->d_flags2, d_out_len, d_ex_off, W.d_st, st);
    launch_ingest_offsets(d_out_len, n, d_tile, d_out_off, st);
    unsigned long long out_text = 0;
    uint32_t st_tag = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&out_text, d_out_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&st_tag, W.d_st, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(flags.data(), m->d_flags2, n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "tag kernels failed");
    if (st_tag) {  // optional fields the device does not decide about: this window is the host path's
        *status = st_tag & 6u;
        return MK_OK;
    }
    uint64_t kept = 0;
    for (uint64_t r = 0; r < n; ++r) kept += flags[r];
    w->n_kept = kept;
    lc.nb_records_extracted = kept;
    w->out_text_bytes = out_text;
    if (logging && (n_rows > w->rows_cap || w->n_names_bytes > w->names_cap))
        return fail(MK_E_CAPACITY, "mk_tag_bam_window: %llu rows and %llu bytes of names", (unsigned long long)n_rows, (unsigned long long)w->n_names_bytes);
    auto commit = [&] {
        c->nb_records_tot += lc.nb_records_tot, c->nb_bases += lc.nb_bases, c->nb_hits_tot[0] += lc.nb_hits_tot[0];
        c->nb_records_hit[0] += lc.nb_records_hit[0], c->nb_records_extracted += lc.nb_records_extracted;
        for (size_t i = 0; i < lcounts.size(); ++i) counts[i] += lcounts[i];
        dl.finish();
    };
    if ((!w->out && !w->out_cap) || out_text == 0) {  // tag -s (the checks have run, nothing is written) / nothing is kept
        commit();
        return MK_OK;
    }
    mk_matcher::TextSlot &O = m->txt[1];
    if ((rc = ensure_device(&O.d_text, &O.d_text_cap, out_text + mkz::kPad + 64))) return rc;
    launch_bam_emit((const uint8_t *)W.T->d_text, W.d_rec_start, d_rec_len, d_out_len, d_out_off, d_found_off, d_found_pat, m->d_pat_bytes, m->d_pat_off, d_ex_off, n,
                    w->tag[0], w->tag[1], (uint8_t *)O.d_text, st);
    if (hipGetLastError() != hipSuccess || hipMemsetAsync((uint8_t *)O.d_text + out_text, 0, mkz::kPad, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(MK_E_HIP, "record output kernel failed");
    w->ms[4] = (float)ms_since(t);
    // ---- BGZF members of the output text (the codec's buffers; its kernels on this stream)
    {
        std::lock_guard<std::mutex> lock(codec->mu);
        const uint64_t blocks64 = (out_text + bb - 1) / bb;
        if (blocks64 >= 0xFFFFFFFFull) return fail(MK_E_UNSUPPORTED, "mk_tag_bam_window: %llu output members", (unsigned long long)blocks64);
        const uint32_t blocks = (uint32_t)blocks64;
        const uint32_t grid = mkz::deflate_grid(blocks, codec->num_cus);
        if ((rc = ensure_device(&codec->d_crc, &codec->crc_cap, blocks * 4ull)) ||
            (rc = ensure_device(&codec->d_tokens, &codec->tokens_cap, (uint64_t)grid * mkz::kTokensPerWave * 4)) ||
            (rc = ensure_device(&codec->d_slots, &codec->slots_cap, (uint64_t)blocks * mkz::kSlotBytes)) ||
            (rc = ensure_device(&codec->d_len, &codec->len_cap, (blocks + 1) * 4ull)) || (rc = ensure_device(&codec->d_off, &codec->off_cap, (blocks + 2) * 8ull)))
            return rc;
        // the packed members go where the window's text was: it has been read for the last time by the record output kernel (a buffer
        // of the output's size less to grow -- growing device buffers is what a job's first windows spend most of their time on)
        void *d_packed = W.T->d_text;
        if (W.T->d_text_cap < mk_bgzf_deflate_bound(out_text, bb)) {
            if ((rc = ensure_device(&codec->d_out, &codec->out_cap, mk_bgzf_deflate_bound(out_text, bb)))) return rc;
            d_packed = codec->d_out;
        }
        uint64_t *d_total = (uint64_t *)codec->d_off + blocks;
        mkz::launch_crc((const uint8_t *)O.d_text, out_text, bb, blocks, (uint32_t *)codec->d_crc, st);
        mkz::launch_deflate((const uint8_t *)O.d_text, out_text, bb, blocks, (const uint32_t *)codec->d_crc, (uint32_t *)codec->d_tokens, (uint8_t *)codec->d_slots,
                            (uint32_t *)codec->d_len, (uint32_t *)(d_total + 1), grid, st);
        mkz::launch_pack((const uint8_t *)codec->d_slots, (const uint32_t *)codec->d_len, (uint64_t *)codec->d_off, d_total, blocks, (uint8_t *)d_packed, st);
        uint64_t total = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "BGZF deflate of the tagged records failed");
        w->ms[5] = (float)ms_since(t);
        w->out_len = total;
        if (total > w->out_cap) return fail(MK_E_CAPACITY, "mk_tag_bam_window: the members take %llu bytes", (unsigned long long)total);
        if (hipMemcpyAsync(w->out, d_packed, total, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(MK_E_HIP, "download of the members failed");
        w->ms[6] = (float)ms_since(t);
    }
    commit();
    return MK_OK;
    MK_ABI_END
}

// the pair loop with the rows merged on the host (mates of 2 GiB or more under BNDMq: the device pair order keeps
// the mate in bit 31 of the position)
static int extract_paired_host_merge(mk_matcher *m, const uint8_t *seq1, const uint64_t *off1, const uint8_t *seq2,
                                     const uint64_t *off2, uint64_t n_rec, int invert, uint8_t *keep, mk_row *rows,
                                     uint64_t rows_cap, uint64_t *n_rows, mk_counters *c, uint32_t *counts) {
    std::vector<uint8_t> f1, f2;
    std::vector<mk_hit> h1, h2;
    int rc = scan_all(m, seq1, off1, n_rec, MK_MODE_HITS, f1, h1);
    if (rc) return rc;
    rc = scan_all(m, seq2, off2, n_rec, MK_MODE_HITS, f2, h2);
    if (rc) return rc;
    RowSink sink{rows, rows_cap};
    c->nb_records_tot += 2 * n_rec;  // :472
    c->nb_bases += (off1[n_rec] - off1[0]) + (off2[n_rec] - off2[0]);
    c->nb_hits_tot[0] += h1.size();
    c->nb_hits_tot[1] += h2.size();
    c->nb_records_hit[0] += popcount_flags(f1, n_rec);
    c->nb_records_hit[1] += popcount_flags(f2, n_rec);
    count_patterns(m->algo, h1, counts);  // BNDMq: once per mate that hit (:575-584)
    count_patterns(m->algo, h2, counts);
    size_t i1 = 0, i2 = 0;  // merge the two ordered hit lists pair by pair
    while (i1 < h1.size() || i2 < h2.size()) {
        const uint64_t r1 = i1 < h1.size() ? h1[i1].rec : ~0ull, r2 = i2 < h2.size() ? h2[i2].rec : ~0ull;
        const uint64_t r = std::min(r1, r2);
        size_t e1 = i1, e2 = i2;
        while (e1 < h1.size() && h1[e1].rec == r) ++e1;
        while (e2 < h2.size() && h2[e2].rec == r) ++e2;
        if (m->algo == MK_ALGO_AC) {  // all of mate 1, then all of mate 2 (:480-533)
            for (size_t k = i1; k < e1; ++k) sink.push(0, h1[k]);
            for (size_t k = i2; k < e2; ++k) sink.push(1, h2[k]);
        } else {  // per pattern: mate-1 hits then mate-2 hits (:543-585)
            size_t a = i1, b = i2;
            while (a < e1 || b < e2) {
                const uint32_t pa = a < e1 ? h1[a].pat : 0xFFFFFFFFu, pb = b < e2 ? h2[b].pat : 0xFFFFFFFFu;
                const uint32_t p = std::min(pa, pb);
                while (a < e1 && h1[a].pat == p) sink.push(0, h1[a++]);
                while (b < e2 && h2[b].pat == p) sink.push(1, h2[b++]);
            }
        }
        i1 = e1;
        i2 = e2;
    }
    for (uint64_t r = 0; r < n_rec; ++r) {  // :600-606
        const bool found = f1[r] || f2[r];
        keep[r] = (uint8_t)(found != (invert != 0));
        c->nb_records_extracted += 2 * keep[r];
    }
    if (n_rows) *n_rows = sink.n;
    if (rows && sink.n > rows_cap) return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)sink.n);
    return MK_OK;
}

int mk_extract_paired(mk_matcher *m, const uint8_t *seq1, const uint64_t *off1, uint64_t n_rec1,
                      const uint8_t *seq2, const uint64_t *off2, uint64_t n_rec2, int logging, int invert,
                      uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows, mk_counters *c,
                      uint32_t *counts) {
    if (!m || !keep || !c || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_rows) *n_rows = 0;
    if (n_rec1 != n_rec2)  // src/cmd_extract.rs:465-468, :608-612
        return fail(MK_E_PAIR_MISMATCH,
                    "The two input files have a different number of records. Please provide valid paired-end read files.");
    const uint64_t n_rec = n_rec1;
    if (n_rec == 0) return MK_OK;
    if (!off1 || !off2) return fail(MK_E_INVALID_ARG, "null buffer");
    MK_ABI_BEGIN
    const bool ac = m->algo == MK_ALGO_AC;
    if (logging && !ac) {  // the device pair order of BNDMq keeps the mate in bit 31 of the position
        bool small = true;
        if ((off1[n_rec] - off1[0]) >= (1ull << 31) || (off2[n_rec] - off2[0]) >= (1ull << 31))
            for (uint64_t i = 0; i < n_rec && small; ++i)
                small = off1[i + 1] - off1[i] < (1ull << 31) && off2[i + 1] - off2[i] < (1ull << 31);
        if (!small) return extract_paired_host_merge(m, seq1, off1, seq2, off2, n_rec, invert, keep, rows, rows_cap, n_rows, c, counts);
    }
    DeviceLoop dl(m);
    std::vector<uint8_t> f1(n_rec), f2(n_rec);
    uint64_t flagged1 = 0, flagged2 = 0;
    const uint32_t mode = logging ? MK_MODE_HITS : MK_MODE_ANY;
    int rc = dl.scan(seq1, off1, n_rec, mode, f1.data(), &flagged1);
    if (rc) return rc;
    const unsigned long long n1 = dl.found;
    if (logging && n1) {  // mate 1's tuples wait in their own buffer while mate 2 is scanned
        if ((rc = ensure_device(&m->d_pair, &m->d_pair_cap, n1 * sizeof(mk_hit)))) return rc;
        if (hipMemcpyAsync(m->d_pair, m->d_hits, n1 * sizeof(mk_hit), hipMemcpyDeviceToDevice, dl.st) != hipSuccess ||
            hipStreamSynchronize(dl.st) != hipSuccess)
            return fail(MK_E_HIP, "copy of the first mate's tuples failed");
    }
    if ((rc = dl.scan(seq2, off2, n_rec, mode, f2.data(), &flagged2))) return rc;
    const unsigned long long n2 = dl.found;
    uint64_t total_rows = 0;
    if (logging) {
        c->nb_records_tot += 2 * n_rec;  // :472
        c->nb_bases += (off1[n_rec] - off1[0]) + (off2[n_rec] - off2[0]);
        c->nb_hits_tot[0] += n1;
        c->nb_hits_tot[1] += n2;
        c->nb_records_hit[0] += flagged1;
        c->nb_records_hit[1] += flagged2;
        total_rows = n1 + n2;
        if (total_rows) {
            // one list: mate 2's tuples (already in the scan buffer), then mate 1's, the mate marked inside a key field
            if ((n1 + n2) * sizeof(mk_hit) > m->d_hits_cap) {  // grow the scan buffer, keeping mate 2's tuples
                void *bigger = nullptr;
                size_t cap = 0;
                if ((rc = ensure_device(&bigger, &cap, (n1 + n2) * sizeof(mk_hit)))) return rc;
                if (n2 && hipMemcpy(bigger, m->d_hits, n2 * sizeof(mk_hit), hipMemcpyDeviceToDevice) != hipSuccess) {
                    (void)hipFree(bigger);
                    return fail(MK_E_HIP, "copy of the second mate's tuples failed");
                }
                if (m->d_hits) (void)hipFree(m->d_hits);
                m->d_hits = (mk_hit *)bigger;
                m->d_hits_cap = cap;
            }
            if (n1 && hipMemcpyAsync(m->d_hits + n2, m->d_pair, n1 * sizeof(mk_hit), hipMemcpyDeviceToDevice, dl.st) != hipSuccess)
                return fail(MK_E_HIP, "copy of the first mate's tuples failed");
            launch_pair_mark(m->d_hits, n2, 1, ac, dl.st);
            launch_pair_mark(m->d_hits + n2, n1, 0, ac, dl.st);
            dl.found = total_rows;
            const uint64_t bound = m->last_n_rec;
            if (ac) m->last_n_rec = 2 * n_rec;  // record' = 2 * record + mate: the bins of the ordering
            rc = dl.order(ac);
            m->last_n_rec = bound;
            if (rc) return rc;
            if ((rc = dl.pair_rows_to_host(ac, rows, rows_cap))) return rc;
            if ((rc = dl.pair_counts(ac, counts))) return rc;
        }
    }
    dl.host_begin();
    for (uint64_t r = 0; r < n_rec; ++r) {  // :600-606
        const bool found = f1[r] || f2[r];
        keep[r] = (uint8_t)(found != (invert != 0));
        c->nb_records_extracted += 2 * keep[r];
    }
    dl.finish();
    if (n_rows) *n_rows = total_rows;
    if (logging && rows && total_rows > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)total_rows);
    return MK_OK;
    MK_ABI_END
}

int mk_tag_records(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                   int filter_matching, int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                   mk_counters *c, uint32_t *counts, uint64_t *found_off, uint32_t *found_pat, uint64_t found_cap) {
    if (!m || !keep || !c || !found_off || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_rows) *n_rows = 0;
    found_off[0] = 0;
    if (n_rec == 0) return MK_OK;
    MK_ABI_BEGIN
    DeviceLoop dl(m);
    std::vector<uint8_t> flags(n_rec);
    uint64_t flagged = 0;
    // the tag loop always needs the matched-pattern SET (src/cmd_tag.rs:392-442)
    int rc = dl.scan(seq, off, n_rec, MK_MODE_HITS, flags.data(), &flagged);
    if (rc) return rc;
    const bool ac = m->algo == MK_ALGO_AC;
    bool set_order = false;  // are the tuples in (record, pattern, position) order?
    if (logging) {
        c->nb_hits_tot[0] += dl.found;
        c->nb_records_tot += n_rec;  // :446-450 (counted before filtering)
        c->nb_bases += off[n_rec] - off[0];
        c->nb_records_hit[0] += flagged;
        if ((rc = dl.order(ac))) return rc;
        if ((rc = dl.rows_to_host(0, rows, rows_cap))) return rc;
        set_order = !ac;
        if (ac && (rc = dl.pattern_counts(true, n_rec, counts))) return rc;  // one per hit (:412); BNDMq: below, from the sets
    }
    // distinct matched patterns per record, ascending: kmers_found after sort_unstable + dedup (:484-485)
    if (!set_order && (rc = dl.order(false))) return rc;
    uint64_t n_found = 0;
    if ((rc = dl.pattern_sets(n_rec, found_off, found_pat, found_cap, &n_found, (logging && !ac) ? counts : nullptr))) return rc;
    dl.host_begin();
    for (uint64_t r = 0; r < n_rec; ++r) {  // :457-467
        const bool has = flags[r] != 0;
        keep[r] = (uint8_t)(filter_matching ? has : (invert ? !has : true));
        c->nb_records_extracted += keep[r];
    }
    dl.finish();
    if (n_rows) *n_rows = logging ? dl.found : 0;
    if (n_found > found_cap) return fail(MK_E_CAPACITY, "found_pat too small: need %llu", (unsigned long long)n_found);
    if (logging && rows && dl.found > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)dl.found);
    return MK_OK;
    MK_ABI_END
}

int mk_matcher_batch_times(const mk_matcher *m, float ms[4]) {
    if (!m || !ms) return fail(MK_E_INVALID_ARG, "null argument");
    for (int i = 0; i < 4; ++i) ms[i] = m->batch_ms[i];
    return MK_OK;
}

int mk_tag_value(const mk_matcher *m, const uint32_t *found_pat, uint64_t n_found, const char *existing, char *out,
                 size_t cap, size_t *out_len) {
    if (!m || (!found_pat && n_found)) return fail(MK_E_INVALID_ARG, "null argument");
    MK_ABI_BEGIN
    std::vector<std::string> items;
    for (uint64_t i = 0; i < n_found; ++i) {
        const uint32_t p = found_pat[i];
        if (p >= m->n_pat) return fail(MK_E_INVALID_ARG, "pattern index %u out of range", p);
        items.emplace_back((const char *)m->pat_bytes.data() + m->pat_off[p], m->pat_off[p + 1] - m->pat_off[p]);
    }
    if (existing && existing[0]) {  // src/cmd_tag.rs:470-481: non-empty Z value split on ','
        const char *s = existing;
        for (;;) {
            const char *e = strchr(s, ',');
            items.emplace_back(s, e ? (size_t)(e - s) : strlen(s));
            if (!e) break;
            s = e + 1;
        }
    }
    std::sort(items.begin(), items.end());  // :484-485
    items.erase(std::unique(items.begin(), items.end()), items.end());
    std::string joined;
    for (size_t i = 0; i < items.size(); ++i) {
        if (i) joined += ',';
        joined += items[i];
    }
    if (out_len) *out_len = joined.size();
    if (!out || cap < joined.size() + 1) return fail(MK_E_CAPACITY, "tag buffer too small: need %zu", joined.size() + 1);
    memcpy(out, joined.c_str(), joined.size() + 1);
    return MK_OK;
    MK_ABI_END
}

}  // extern "C"
