// tag_windows.cpp -- `merkurio tag` BAM -> BAM (or -S) with the records resident on the device (r05; SURVEY.md §8 rows a11 / f-3;
// the reader loop, process_record and the writer of src/cmd_tag.rs:503-615, :387-497, :254-271).
//
// The r04 path inflated a window of the BAM into host memory (device codec), indexed its record chain and un-nibbled the
// sequences on the host threads, sent those to the scan, appended the tags on the host threads and handed the records back to
// the device to be deflated: the text crossed the host boundary three times.  Here a window is the compressed members AS THEY ARE
// STORED: mk_tag_bam_window inflates them, indexes the records, unpacks, scans, tags and deflates on the device; what comes back
// is the window's last unfinished record (the next window's head), the log rows with their record names, and finished BGZF
// members that go to the file as they are.
//
// Two windows per device are in flight, each on a handle (and stream) of its own (--gpus N: consecutive windows on different devices): window k + 1 needs only the TAIL of window k -- known right
// after k's record index (mk_bam_window::on_tail) -- so its upload and inflate run beside k's scan, tag, deflate and download, and
// its members are copied into page-locked memory (a mapped file is not a DMA source) beside all of that.  Results are emitted in
// window order; the writer thread writes window k - 1 meanwhile.
//
// A window the device refuses -- a record that fails the parser's checks, optional fields that do not parse, a kept record whose
// field of the tag's name is not a plain string, a damaged member -- hands the input back to the host reader AT THAT WINDOW'S FIRST BYTE
// (SamFile::seek_bam): the r04 path takes the rest of the file and words the reference's errors.  (The window behind it may have
// been started already: its results are dropped.)
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "../../../include/merkurio_hip.h"
#include "commands.hpp"
#include "extract_common.hpp"
#include "io.hpp"

namespace cli {

namespace {

struct PinnedBuffer {
    uint8_t *p = nullptr;
    uint64_t cap = 0;
    ~PinnedBuffer() {
        if (p && !g_process_is_ending) mk_host_free(p);
    }
    void need(uint64_t n) {
        if (n <= cap) return;
        if (p) mk_host_free(p);
        p = nullptr;
        cap = 0;
        void *q = nullptr;
        const uint64_t want = n + n / 8 + (1u << 20);
        if (mk_host_alloc(want, &q) != MK_OK) bail(std::string("Error during BAM record parsing: ") + mk_last_error());
        p = (uint8_t *)q;
        cap = want;
    }
};

// members [m0, m1) of the file: their table re-based to the first one's DEFLATE stream, and that byte range of the file
struct WindowMembers {
    size_t m0 = 0, m1 = 0;
    uint64_t file_lo = 0, file_hi = 0, text = 0;
    std::vector<mk_bgzf_member> mem;
};

// file[lo, hi) -> dst on a few host threads (first touch of the mapping's pages included)
void copy_in(const uint8_t *file, uint64_t lo, uint64_t hi, uint8_t *dst) {
    const uint64_t n = hi - lo;
    const size_t T = std::max<size_t>(1, std::min<size_t>(std::max(1u, io_threads() / 2), (size_t)(n >> 22) + 1));
    run_threads(T, [&](size_t t) { memcpy(dst + n * t / T, file + lo + n * t / T, (size_t)(n * (t + 1) / T - n * t / T)); });
}

// what the two workers share: whose head is known, whose turn it is to emit, and how the job ends early
struct Pipe {
    std::mutex mu;
    std::condition_variable cv;
    size_t heads_ready = 0;  // the head of window `heads_ready` is in `head` (windows before it have theirs already)
    std::vector<uint8_t> head;
    size_t emit_turn = 0;               // windows before this one have been emitted
    bool stop = false;                  // a window was refused or failed: nothing further is emitted
    size_t refused = ~(size_t)0;        // the window the host reader takes over at ...
    std::vector<uint8_t> refused_head;  // ... and the head it was given
    std::string error;
};

struct TailCtx {
    Pipe *pipe;
    size_t k;
};

void on_tail(void *ctx, const uint8_t *tail, uint64_t n_tail) {
    TailCtx *t = (TailCtx *)ctx;
    std::lock_guard<std::mutex> lk(t->pipe->mu);
    if (t->pipe->heads_ready != t->k) return;  // (a call repeated with larger buffers reports its tail again)
    t->pipe->head.assign(tail, tail + n_tail);
    t->pipe->heads_ready = t->k + 1;
    t->pipe->cv.notify_all();
}

}  // namespace

bool tag_bam_windows_on_device(const TagArgs &a, SamFile &sam, const std::vector<TagHandle> &handles, Loggers &lg, const Patterns &pats,
                               const std::string &in_name, BamWriter *bw, uint64_t window_bytes) {
    const WindowSource &src = sam.source();
    const size_t n_mem = src.n_bgzf_members();
    const uint8_t *file = src.file_bytes();
    const bool timing = getenv("MERKURIO_TIMING") != nullptr;
    uint64_t n_pending = 0;
    const char *pend = sam.bam_pending(&n_pending);
    Pipe pipe;
    pipe.head.assign(pend, pend + n_pending);
    // the windows: runs of members of ~window_bytes of text (the same whatever the heads turn out to be)
    std::vector<WindowMembers> wins;
    for (size_t m0 = src.next_member(); m0 < n_mem;) {
        WindowMembers W;
        W.m0 = m0;
        size_t m1 = m0;
        while (m1 < n_mem && (W.text < window_bytes || m1 == m0)) {
            uint64_t off;
            uint32_t len, isize, crc;
            src.bgzf_member_at(m1, &off, &len, &isize, &crc);
            if (m1 == m0) W.file_lo = off;
            W.file_hi = off + len;
            W.mem.push_back(mk_bgzf_member{off - W.file_lo, W.text, len, isize, crc, 0});
            W.text += isize;
            ++m1;
        }
        W.m1 = m1;
        m0 = m1;
        wins.push_back(std::move(W));
    }
    if (wins.empty()) {
        if (pipe.head.empty()) return true;
        WindowMembers W;  // (records behind the header that open() has inflated already, and no member behind them)
        W.m0 = W.m1 = n_mem;
        wins.push_back(std::move(W));
    }
    const size_t n_win = wins.size();
    // (window k runs on handle k mod n_workers: with the handles of several devices in a row, consecutive windows go to different devices)
    const size_t n_workers = std::max<size_t>(1, std::min<size_t>(handles.size(), n_win));
    double t_dev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // (BAM-shaped text deflates to a third; a window that does not fit is done again with the size it asked for)
    auto out_guess = [](uint64_t text) { return text / 3 + (4u << 20); };

    auto worker = [&](size_t id) {
        mk_matcher *m = handles[id].m;
        mk_counters &c = *handles[id].counters;
        std::vector<uint32_t> &counts = *handles[id].pattern_counts;
        mk_codec *codec = nullptr;
        if (mk_codec_create(handles[id].device, &codec) != MK_OK) bail(std::string("Error during BAM record parsing: ") + mk_last_error());
        struct CodecGuard {
            mk_codec *c;
            ~CodecGuard() {
                if (!g_process_is_ending) mk_codec_destroy(c);
            }
        } guard{codec};
        PinnedBuffer stage;
        std::vector<uint8_t> tail(1u << 20), names(1u << 16);
        BamWriter::RawBuffer out;
        struct OutGuard {
            BamWriter::RawBuffer &b;
            ~OutGuard() { BamWriter::free_raw_buffer(b); }
        } out_guard{out};
        std::vector<mk_row> rows(4096);
        std::vector<uint64_t> row_name(4096);
        for (size_t k = id; k < n_win; k += n_workers) {
            const WindowMembers &X = wins[k];
            {
                std::lock_guard<std::mutex> lk(pipe.mu);
                if (pipe.stop) return;
            }
            if (!X.mem.empty()) {
                stage.need(X.file_hi - X.file_lo);
                copy_in(file, X.file_lo, X.file_hi, stage.p);
            }
            if (bw && out.cap < out_guess(X.text + (1u << 20))) {
                BamWriter::free_raw_buffer(out);
                out = bw->take_raw_buffer(out_guess(X.text + (1u << 20)));
            }
            std::vector<uint8_t> head;
            {
                std::unique_lock<std::mutex> lk(pipe.mu);
                pipe.cv.wait(lk, [&] { return pipe.stop || pipe.heads_ready >= k; });
                if (pipe.stop) return;
                head = pipe.head;
            }
            TailCtx tctx{&pipe, k};
            mk_bam_window w;
            memset(&w, 0, sizeof(w));
            w.head = head.data(), w.n_head = head.size();
            w.bgzf = stage.p, w.n_bgzf = X.file_hi - X.file_lo;
            w.members = X.mem.data(), w.n_members = X.mem.size();
            w.last = k + 1 == n_win, w.filter_matching = a.filter_matching, w.invert = a.invert_match;
            w.tag[0] = (uint8_t)a.tag[0], w.tag[1] = (uint8_t)a.tag[1];
            w.on_tail = on_tail, w.on_tail_ctx = &tctx;
            mk_counters wc;
            std::vector<uint32_t> wcounts(lg.active ? pats.list.size() : 0, 0);
            uint32_t status = 0;
            int rc;
            for (;;) {
                memset(&wc, 0, sizeof(wc));
                std::fill(wcounts.begin(), wcounts.end(), 0);
                w.tail = tail.data(), w.tail_cap = tail.size();
                w.out = bw ? out.p : nullptr, w.out_cap = bw ? out.cap : 0;
                w.rows = rows.data(), w.rows_cap = rows.size(), w.row_name = row_name.data(), w.names = names.data(), w.names_cap = names.size();
                rc = mk_tag_bam_window(m, codec, &w, lg.active, &wc, wcounts.data(), &status);
                if (rc != MK_E_CAPACITY) break;
                bool grew = false;
                if (w.n_tail > tail.size()) tail.resize(w.n_tail + (1u << 20)), grew = true;
                if (bw && w.out_len > out.cap) {
                    BamWriter::free_raw_buffer(out);
                    out = bw->take_raw_buffer(w.out_len);
                    grew = true;
                }
                if (w.n_rows > rows.size()) rows.resize(w.n_rows), row_name.resize(w.n_rows), grew = true;
                if (w.n_names_bytes > names.size()) names.resize(w.n_names_bytes), grew = true;
                if (!grew) break;
            }
            const bool refused = rc == MK_E_CORRUPT || (rc == MK_OK && status != 0);
            if (rc == MK_OK && !status) on_tail(&tctx, tail.data(), w.n_tail);  // (an empty window returns before the library reports it)
            std::string err;
            if (!refused && rc != MK_OK) err = std::string("Error during matching: ") + mk_last_error();
            // results leave in window order
            std::unique_lock<std::mutex> lk(pipe.mu);
            pipe.cv.wait(lk, [&] { return pipe.stop || pipe.emit_turn == k; });
            if (pipe.stop) return;  // (an earlier window ended the job: this one's results are dropped)
            if (refused || !err.empty()) {
                pipe.stop = true;
                if (refused) {
                    pipe.refused = k;
                    pipe.refused_head = head;
                    if (timing)
                        fprintf(stderr, "[timing] window %llu left to the host reader (%s)\n", (unsigned long long)k,
                                rc == MK_E_CORRUPT ? "a damaged member"
                                : status & 1       ? "record chain"
                                : status & 2       ? "optional fields"
                                : status & 4       ? "existing tag"
                                                   : "unfinished record");
                } else {
                    pipe.error = err;
                }
                pipe.cv.notify_all();
                return;
            }
            lk.unlock();
            // (only the worker whose turn it is gets here: the counters of its device, the loggers and the writer are its alone)
            c.nb_records_tot += wc.nb_records_tot, c.nb_bases += wc.nb_bases, c.nb_hits_tot[0] += wc.nb_hits_tot[0];
            c.nb_records_hit[0] += wc.nb_records_hit[0], c.nb_records_extracted += wc.nb_records_extracted;
            for (size_t i = 0; i < wcounts.size(); ++i) counts[i] += wcounts[i];
            for (int i = 0; i < 8; ++i) t_dev[i] += w.ms[i];
            std::string emit_err;
            try {
                if (lg.active)
                    emit_log_rows(
                        lg, pats, rows.data(), w.n_rows,
                        [&](const mk_row &r) {
                            const char *nm = (const char *)names.data() + row_name[&r - rows.data()];
                            return std::pair<const char *, size_t>(nm, strlen(nm));
                        },
                        [&](const mk_row &) -> const std::string & { return in_name; });
                if (bw && w.out_len) {
                    bw->put_members(out, w.out_len);
                    out = BamWriter::RawBuffer();
                }
            } catch (const Error &e) {
                emit_err = e.what()[0] ? e.what() : "error";
            }
            lk.lock();
            if (!emit_err.empty()) pipe.stop = true, pipe.error = emit_err;
            pipe.emit_turn = k + 1;
            pipe.cv.notify_all();
            if (pipe.stop) return;
        }
    };
    // (a worker that throws must not leave the other one waiting)
    run_threads(n_workers, [&](size_t id) {
        try {
            worker(id);
        } catch (const Error &e) {
            std::lock_guard<std::mutex> lk(pipe.mu);
            if (pipe.error.empty()) pipe.error = e.what()[0] ? e.what() : "error";
            pipe.stop = true;
            pipe.cv.notify_all();
        }
    });
    if (!pipe.error.empty()) bail(pipe.error);
    if (timing)
        fprintf(stderr,
                "[timing] %llu of %llu windows on the device (%llu in flight): upload %.3f, inflate %.3f, record index %.3f, unpack + scan + sets %.3f, "
                "tag + pack %.3f, deflate %.3f, download %.3f s (of these, growing device buffers: %.3f s)\n",
                (unsigned long long)pipe.emit_turn, (unsigned long long)n_win, (unsigned long long)n_workers, t_dev[0] / 1e3, t_dev[1] / 1e3, t_dev[2] / 1e3,
                t_dev[3] / 1e3, t_dev[4] / 1e3, t_dev[5] / 1e3, t_dev[6] / 1e3, t_dev[7] / 1e3);
    if (pipe.refused != ~(size_t)0) {
        sam.seek_bam(wins[pipe.refused].m0, (const char *)pipe.refused_head.data(), pipe.refused_head.size());
        return false;
    }
    return true;
}

}  // namespace cli
