// host_loops.cpp -- what the reference's record loops do with the matcher's answers, restated
// for batches so that keep/drop decisions, log rows and counters are bit-identical:
//   extract single   src/cmd_extract.rs:321-406
//   extract paired   src/cmd_extract.rs:463-612
//   tag              src/cmd_tag.rs:387-490
// The matching itself is mk_scan_batch (gfx950 kernel); nothing here searches text.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "matcher_internal.h"

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

}  // namespace

extern "C" {

int mk_extract_single(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                      int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                      mk_counters *c, uint32_t *counts) {
    if (!m || !keep || !c || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_rows) *n_rows = 0;
    MK_ABI_BEGIN
    std::vector<uint8_t> flags;
    std::vector<mk_hit> hits;
    int rc = scan_all(m, seq, off, n_rec, logging ? MK_MODE_HITS : MK_MODE_ANY, flags, hits);
    if (rc) return rc;
    RowSink sink{rows, rows_cap};
    if (logging) {
        c->nb_records_tot += n_rec;                                  // :326
        c->nb_bases += n_rec ? off[n_rec] - off[0] : 0;              // :327
        for (auto &h : hits) sink.push(0, h);                        // :338-351 / :369-377
        c->nb_hits_tot[0] += hits.size();                            // :354 / :378
        c->nb_records_hit[0] += popcount_flags(flags, n_rec);        // :358-360 / :385-387
        count_patterns(m->algo, hits, counts);
    }
    for (uint64_t r = 0; r < n_rec; ++r) {  // :400-405
        keep[r] = (uint8_t)((flags[r] != 0) != (invert != 0));
        c->nb_records_extracted += keep[r];
    }
    if (n_rows) *n_rows = sink.n;
    if (logging && rows && sink.n > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)sink.n);
    return MK_OK;
    MK_ABI_END
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
    MK_ABI_BEGIN
    std::vector<uint8_t> f1, f2;
    std::vector<mk_hit> h1, h2;
    const uint32_t mode = logging ? MK_MODE_HITS : MK_MODE_ANY;
    int rc = scan_all(m, seq1, off1, n_rec, mode, f1, h1);
    if (rc) return rc;
    rc = scan_all(m, seq2, off2, n_rec, mode, f2, h2);
    if (rc) return rc;
    RowSink sink{rows, rows_cap};
    if (logging) {
        c->nb_records_tot += 2 * n_rec;  // :472
        c->nb_bases += n_rec ? (off1[n_rec] - off1[0]) + (off2[n_rec] - off2[0]) : 0;
        c->nb_hits_tot[0] += h1.size();
        c->nb_hits_tot[1] += h2.size();
        c->nb_records_hit[0] += popcount_flags(f1, n_rec);
        c->nb_records_hit[1] += popcount_flags(f2, n_rec);
        count_patterns(m->algo, h1, counts);  // BNDMq: once per mate that hit (:575-584)
        count_patterns(m->algo, h2, counts);
        // merge the two ordered hit lists pair by pair
        size_t i1 = 0, i2 = 0;
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
    }
    for (uint64_t r = 0; r < n_rec; ++r) {  // :600-606
        const bool found = f1[r] || f2[r];
        keep[r] = (uint8_t)(found != (invert != 0));
        c->nb_records_extracted += 2 * keep[r];
    }
    if (n_rows) *n_rows = sink.n;
    if (logging && rows && sink.n > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)sink.n);
    return MK_OK;
    MK_ABI_END
}

int mk_tag_records(mk_matcher *m, const uint8_t *seq, const uint64_t *off, uint64_t n_rec, int logging,
                   int filter_matching, int invert, uint8_t *keep, mk_row *rows, uint64_t rows_cap, uint64_t *n_rows,
                   mk_counters *c, uint32_t *counts, uint64_t *found_off, uint32_t *found_pat, uint64_t found_cap) {
    if (!m || !keep || !c || !found_off || (logging && !counts)) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_rows) *n_rows = 0;
    MK_ABI_BEGIN
    std::vector<uint8_t> flags;
    std::vector<mk_hit> hits;
    // the tag loop always needs the matched-pattern SET (src/cmd_tag.rs:392-442)
    int rc = scan_all(m, seq, off, n_rec, MK_MODE_HITS, flags, hits);
    if (rc) return rc;
    RowSink sink{rows, rows_cap};
    if (logging) {
        for (auto &h : hits) sink.push(0, h);
        c->nb_hits_tot[0] += hits.size();
        count_patterns(m->algo, hits, counts);
        c->nb_records_tot += n_rec;  // :446-450 (counted before filtering)
        c->nb_bases += n_rec ? off[n_rec] - off[0] : 0;
        c->nb_records_hit[0] += popcount_flags(flags, n_rec);
    }
    // distinct matched patterns per record, ascending
    uint64_t w = 0;
    size_t i = 0;
    std::vector<uint32_t> tmp;
    for (uint64_t r = 0; r < n_rec; ++r) {
        found_off[r] = w;
        tmp.clear();
        while (i < hits.size() && hits[i].rec == r) tmp.push_back(hits[i++].pat);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (uint32_t p : tmp) {
            if (found_pat && w < found_cap) found_pat[w] = p;
            ++w;
        }
        const bool has = flags[r] != 0;  // :457-467
        keep[r] = (uint8_t)(filter_matching ? has : (invert ? !has : true));
        c->nb_records_extracted += keep[r];
    }
    found_off[n_rec] = w;
    if (n_rows) *n_rows = sink.n;
    if (w > found_cap) return fail(MK_E_CAPACITY, "found_pat too small: need %llu", (unsigned long long)w);
    if (logging && rows && sink.n > rows_cap)
        return fail(MK_E_CAPACITY, "rows buffer too small: need %llu", (unsigned long long)sink.n);
    return MK_OK;
    MK_ABI_END
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
