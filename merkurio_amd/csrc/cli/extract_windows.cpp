// extract_windows.cpp -- `merkurio extract` over text windows that are indexed on the device.
//
// Everything the reference's loops read through needletail (src/cmd_extract.rs:281-282 single, :412-418 paired; records consumed
// at :321-328 and :463-468) goes to the GPU as raw text: one FASTQ or FASTA file or a pair of them, plain, gzip (inflated here,
// a stream has no independent pieces) or bgzip'ed (the members go up as they are and are inflated there), on one device or dealt
// window by window to several.  mk_extract_window does the rest: record index, sequences, scan, emission order, rows, counts, keep.
//
//   reader thread   bodies of window k + 1 (a record-aligned slice of every plain / inflated input, staged into page-locked
//                   memory and sent ahead; a group of BGZF members), then -- where windows depend on each other -- the tails of
//                   window k as its heads
//   device threads  window k on device k mod N: one mk_extract_window call; a window the device refuses (not plain 4-line FASTQ /
//                   FASTA, or too large for it) is parsed by the host reader's own code instead, which also words the reference's
//                   parse errors
//   main thread     results in window order: log rows (reference emission order), kept records
//
// Windows depend on each other when a window can end inside a record (BGZF members: the host never sees the text) or when the two
// files of a pair hold different numbers of records per window (the leftovers of one are carried): then window k + 1 cannot start
// before window k has returned its tails, whatever device it runs on; a single plain or gzip input has no such chain and its
// windows run on their devices independently.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "extract_common.hpp"

namespace cli {

namespace {

constexpr uint64_t kFirstWindow = 8ull << 20;  // (page-locked staging needs the HIP runtime, which is still starting: a small first window)

struct Pinned {
    void *p = nullptr;
    uint64_t cap = 0;
    void need(uint64_t n) {
        if (n <= cap) return;
        mk_host_free(p);
        p = nullptr;
        cap = 0;
        mk_check(mk_host_alloc((size_t)(n + n / 8 + 4096), &p), "Error allocating page-locked memory");
        cap = n + n / 8 + 4096;
    }
};

// one input file as a producer of window bodies
struct Input {
    std::string path;
    FastxStream s;
    bool bgzf_dev = false;  // its members go to the device as they are
    // a plain gzip file whose ONE member the device has inflated in parallel pieces (mk_gzip_inflate_device): the text lies there,
    // windows are ranges of it
    bool gz_dev = false;
    mk_codec *gz_codec = nullptr;
    const uint8_t *gz_text = nullptr;
    uint64_t gz_bytes = 0, gz_next = 0;
    bool dev_text() const { return bgzf_dev || gz_dev; }  // the host never holds this input's text: tails / kept records come back
    int fd = -1;            // plain file: windows are staged with pread()
    size_t next_member = 0;
    bool exhausted = false;
    const char *first_text = nullptr;  // the first raw window (prepare())
    uint64_t first_n = 0, first_resume = 0;
    bool have_first = false;
    std::vector<Pinned> pins;
    std::deque<int> free_pins;
};

struct Side {  // one input's part of a window
    std::vector<char> head;
    const char *body = nullptr;  // plain text body ...
    uint64_t n_body = 0;
    int pin = -1;                 // ... in this page-locked buffer of its input (-1: where the reader left it)
    std::vector<char> own;        // (first window of a compressed input: a copy of its own)
    std::vector<mk_bgzf_member> grp;  // ... or BGZF members
    const uint8_t *dev_body = nullptr;  // ... or a range of text that lies on the device
    uint64_t n_dev_body = 0, dev_off = 0;
    bool ends = true;             // the body ends at a record end
    // results
    std::vector<uint64_t> rec_start;
    uint64_t n_window = 0, n_used = 0;
    std::vector<char> tail;       // the text behind the processed records: the next window's head
    enum Layout { HEAD_BODY, PACKED_KEPT, WHOLE } layout = HEAD_BODY;
    std::vector<char> text;       // what the writer reads from unless HEAD_BODY: the kept records, the whole window, the host parser's text
    FastxFile parsed;             // host-parsed window: the records (parsed.data = text)
};

struct Win {
    uint64_t k = 0;
    int dev = 0;
    Side side[2];
    std::string error;  // a window that only carries a message: raised when the writer reaches it (after everything in front of it)
    uint64_t n_rec = 0, n_rows = 0;
    std::vector<uint8_t> keep;
    std::vector<mk_row> rows;
    mk_counters cb;
    std::vector<uint32_t> cnt;
    bool by_host = false;
};

}  // namespace

struct WindowExtract::Impl {
    ExtractArgs a;
    std::vector<int> devs;
    int n_in = 1;
    Input in[2];
    bool fastq = true;
    bool chained = false;
    uint64_t plain_target = 128ull << 20, bgzf_target = 1ull << 30;
    const bool timing = getenv("MERKURIO_TIMING") != nullptr;
    bool whole_text = false;  // -v with a log: the rows name records that are not kept -- BGZF windows hand their whole text back

    // ---- set by run()
    const Patterns *pats = nullptr;
    Loggers *lg = nullptr;
    std::vector<mk_matcher *> ms;
    std::vector<mk_codec *> codecs;
    uint64_t batch_bytes = 128ull << 20;

    // ---- pipeline state
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<std::unique_ptr<Win>>> queue;  // per device
    std::map<uint64_t, std::unique_ptr<Win>> done;        // finished windows by number
    std::map<uint64_t, std::vector<std::vector<char>>> tails;  // tails of finished windows (chained inputs)
    uint64_t written = 0;                                   // windows the writer is through with
    bool reader_done = false;
    uint64_t n_windows = 0;  // (valid once reader_done)
    std::string failure;
    bool failed = false;
    uint64_t last_head[2] = {0, 0};
    std::atomic<uint64_t> rec_cap_seen{0};

    void fail_all(const std::string &msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) failure = msg;
        failed = true;
        cv.notify_all();
    }

    // ---- reader side ---------------------------------------------------------------------------------------------------------
    // the window's bytes into pinned memory, on all host threads (page-cache pages cannot be DMA sources).  A plain file is read with
    // pread() -- the kernel copies from the page cache without a page fault per 4 KiB of a mapping --, inflated text is copied.
    void stage(Input &I, Pinned &dst, const char *src, uint64_t n, uint64_t file_off) {
        dst.need(n);
        const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), (size_t)(n >> 22) + 1));
        const int fd = I.fd;
        run_threads(T, [&](size_t t) {
            const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
            uint64_t done_ = lo;
            while (fd >= 0 && done_ < hi) {
                const ssize_t got = pread(fd, (char *)dst.p + done_, (size_t)(hi - done_), (off_t)(file_off + done_));
                if (got <= 0) break;
                done_ += (uint64_t)got;
            }
            if (done_ < hi) memcpy((char *)dst.p + done_, src + done_, (size_t)(hi - done_));  // (no descriptor, or a short read)
        });
    }

    // the next body of input i into S; first: the window prepare() has read.  staged: page-locked staging + upload ahead are possible
    void next_body(int i, Side &S, uint64_t target, bool staged, mk_matcher *m_ahead) {
        Input &I = in[i];
        S.body = nullptr, S.n_body = 0, S.pin = -1, S.grp.clear(), S.ends = true, S.dev_body = nullptr, S.n_dev_body = 0;
        if (I.exhausted) return;
        if (I.gz_dev) {
            const uint64_t n = std::min<uint64_t>(target, I.gz_bytes - I.gz_next);
            S.dev_body = I.gz_text + I.gz_next, S.n_dev_body = n, S.dev_off = I.gz_next;
            I.gz_next += n;
            I.exhausted = I.gz_next >= I.gz_bytes;
            S.ends = I.exhausted;
            return;
        }
        if (I.bgzf_dev) {
            const WindowSource &ws = I.s.source();
            const size_t nm = ws.n_bgzf_members();
            uint64_t text = 0;
            size_t g = I.next_member;
            while (g < nm && (text < target || g == I.next_member)) {
                mk_bgzf_member e{};
                ws.bgzf_member_at(g, &e.data_off, &e.data_len, &e.isize, &e.crc);
                e.out_off = text;
                text += e.isize;
                S.grp.push_back(e);
                ++g;
            }
            I.next_member = g;
            I.exhausted = g >= nm;
            S.ends = I.exhausted;
            return;
        }
        const char *text = nullptr;
        uint64_t n = 0, resume = 0;
        bool more;
        if (I.have_first) {
            more = true, text = I.first_text, n = I.first_n, resume = I.first_resume;
            I.have_first = false;
        } else {
            more = I.s.raw_fill(target, &text, &n, &resume);
        }
        if (!more) {
            // nothing, or text that does not start like a record (garbage behind the last record): the rest goes out as one last body,
            // the device refuses it and the host parser words the reference's error
            I.exhausted = true;
            if (I.s.raw_rest(&text, &n) && n) {
                S.own.assign(text, text + n);
                S.body = S.own.data(), S.n_body = n;
            }
            return;
        }
        I.s.raw_consume();
        if (!staged) {
            if (I.s.raw_is_plain() || I.s.source().mapped()) {
                S.body = text;  // (a mapping, or a whole inflated bzip2 / xz / zstd text: stays where it is)
            } else {
                S.own.assign(text, text + n);  // (the reader's buffers are recycled two windows on)
                S.body = S.own.data();
            }
            S.n_body = n;
            return;
        }
        int pin;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || !I.free_pins.empty(); });
            if (failed) return;
            pin = I.free_pins.front();
            I.free_pins.pop_front();
        }
        const double t_stage = PhaseTimer::now();
        stage(I, I.pins[pin], text, n, resume - n);
        S.pin = pin, S.body = (const char *)I.pins[pin].p, S.n_body = n;
        if (m_ahead) mk_check(mk_upload_text_ahead(m_ahead, (const uint8_t *)S.body, n), "Error uploading the next window");
        if (timing) fprintf(stderr, "[timing]   reader: input %d, %.0f MB staged + sent ahead in %.1f ms\n", i, n / 1e6, (PhaseTimer::now() - t_stage) * 1e3);
    }

    void reader_loop() {
        const size_t N = devs.size();
        for (uint64_t k = 0;; ++k) {
            std::unique_ptr<Win> W(new Win);
            W->k = k, W->dev = (int)(k % N);
            bool any_body = false;
            for (int i = 0; i < n_in; ++i) {
                const uint64_t full = in[i].dev_text() ? bgzf_target : (k == 0 ? std::min(plain_target, kFirstWindow) : plain_target);
                // (the previous window's leftovers count against this one: the two files of a pair then advance at the same rate)
                const uint64_t target = std::max<uint64_t>(1u << 16, full > last_head[i] ? full - last_head[i] : 0);
                next_body(i, W->side[i], target, k > 0, ms.empty() ? nullptr : ms[W->dev]);
                any_body = any_body || W->side[i].n_body || !W->side[i].grp.empty() || W->side[i].n_dev_body;
            }
            bool any_head = false;
            if (chained && k > 0) {  // the tails of window k - 1 are this window's heads
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || tails.count(k - 1); });
                if (failed) return;
                std::vector<std::vector<char>> t = std::move(tails[k - 1]);
                tails.erase(k - 1);
                lk.unlock();
                for (int i = 0; i < n_in; ++i) {
                    W->side[i].head = std::move(t[i]);
                    last_head[i] = W->side[i].head.size();
                    any_head = any_head || !W->side[i].head.empty();
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (failed) return;
            }
            if (!any_body && !any_head) {
                std::lock_guard<std::mutex> lk(mu);
                reader_done = true;
                n_windows = k;
                cv.notify_all();
                return;
            }
            if (n_in == 2) {
                // one file has nothing left while the other still holds records: the reference notices at this point, after the common
                // part (src/cmd_extract.rs:465-468 file 2 ends first; :608-612 file 2 still has records)
                bool none[2];
                for (int i = 0; i < 2; ++i) none[i] = W->side[i].head.empty() && !W->side[i].n_body && W->side[i].grp.empty() && !W->side[i].n_dev_body;
                if (none[0] != none[1]) {
                    W->error = none[1] ? "Error during FASTQ record parsing of second file. Do the two input files contain the same number of records?"
                                       : "The two input files have a different number of records. Please provide valid paired-end read files.";
                    // (what the other file still holds may be nothing but blank lines at its end: the host parser says)
                    Side &S = W->side[none[0] ? 1 : 0];
                    bool only_blank = S.grp.empty() && !S.n_dev_body;
                    for (char ch : S.head) only_blank = only_blank && (ch == '\n' || ch == '\r');
                    for (uint64_t q = 0; q < S.n_body && only_blank; ++q) only_blank = S.body[q] == '\n' || S.body[q] == '\r';
                    if (only_blank) W->error.clear(), W->by_host = true;  // an empty window: nothing to do, the job goes on to its end
                }
            }
            const bool stop = !W->error.empty();
            {
                std::lock_guard<std::mutex> lk(mu);
                queue[W->dev].push_back(std::move(W));
                cv.notify_all();
            }
            if (stop) {
                std::lock_guard<std::mutex> lk(mu);
                reader_done = true;
                n_windows = k + 1;
                cv.notify_all();
                return;
            }
        }
    }

    // ---- device side -----------------------------------------------------------------------------------------------------------
    // contiguous text of a side on the host: head ++ body (BGZF members inflated by zlib)
    void side_text(int i, Side &S, std::vector<char> &out) {
        uint64_t body = S.n_body + S.n_dev_body;
        for (auto &e : S.grp) body += e.isize;
        out.resize(S.head.size() + body);
        if (!S.head.empty()) memcpy(out.data(), S.head.data(), S.head.size());
        if (S.n_body) memcpy(out.data() + S.head.size(), S.body, S.n_body);
        if (S.n_dev_body)
            mk_check(mk_gzip_text_read(in[i].gz_codec, S.dev_off, (uint8_t *)out.data() + S.head.size(), S.n_dev_body), "Error reading the inflated text back");
        if (!S.grp.empty())
            inflate_bgzf_members_host(in[i].s.source().file_bytes(), S.grp.data(), S.grp.size(), out.data() + S.head.size(), in[i].path);
    }

    // a window the device did not take: parsed by the host reader's code (which words the reference's errors), scanned in batches
    void host_window(Win &W, mk_matcher *m) {
        W.by_host = true;
        uint64_t n = ~0ull;
        uint64_t stop[2] = {0, 0};
        for (int i = 0; i < n_in; ++i) {
            Side &S = W.side[i];
            side_text(i, S, S.text);
            S.parsed.fastq = fastq;
            S.parsed.data = S.text.data();
            S.parsed.data_n = S.text.size();
            S.parsed.recs.clear();
            stop[i] = S.parsed.parse_span_partial(0, S.text.size(), !S.ends);
            n = std::min<uint64_t>(n, S.parsed.recs.size());
        }
        for (int i = 0; i < n_in; ++i) {
            Side &S = W.side[i];
            // the text behind record n - 1 is the tail (the marker of record n sits one byte in front of its id)
            const uint64_t cut = n < S.parsed.recs.size() ? S.parsed.recs[n].id_b - 1 : stop[i];
            S.tail.assign(S.text.begin() + (ptrdiff_t)cut, S.text.end());
            // (what is left behind the last record of an input that has ended can only be blank lines)
            if (S.ends && n == S.parsed.recs.size()) S.tail.clear();
            S.parsed.recs.resize(n);
            S.n_used = cut;
        }
        W.n_rec = n;
        W.keep.assign(n, 0);
        W.rows.clear();
        W.n_rows = 0;
        memset(&W.cb, 0, sizeof(W.cb));
        W.cnt.assign(pats->list.size(), 0);
        std::vector<uint8_t> s1, s2;
        std::vector<uint64_t> o1, o2;
        std::vector<mk_row> rows(4096);
        for (size_t b0 = 0; b0 < n;) {
            size_t b1 = b0;
            uint64_t bytes = 0;
            while (b1 < n && (bytes < batch_bytes || b1 == b0)) {
                bytes += W.side[0].parsed.raw_len(b1) + (n_in == 2 ? W.side[1].parsed.raw_len(b1) : 0);
                ++b1;
            }
            W.side[0].parsed.gather(b0, b1, s1, o1);
            if (n_in == 2) W.side[1].parsed.gather(b0, b1, s2, o2);
            const uint64_t nb = b1 - b0;
            uint64_t n_rows = 0;
            for (;;) {
                mk_counters cb;
                memset(&cb, 0, sizeof(cb));
                std::vector<uint32_t> cnt_b(W.cnt.size(), 0);
                const int rc = n_in == 2 ? mk_extract_paired(m, s1.data(), o1.data(), nb, s2.data(), o2.data(), nb, lg->active, a.invert_match,
                                                             W.keep.data() + b0, rows.data(), rows.size(), &n_rows, &cb, cnt_b.data())
                                         : mk_extract_single(m, s1.data(), o1.data(), nb, lg->active, a.invert_match, W.keep.data() + b0, rows.data(),
                                                             rows.size(), &n_rows, &cb, cnt_b.data());
                if (rc == MK_E_CAPACITY && n_rows > rows.size()) {
                    rows.resize(n_rows);
                    continue;
                }
                mk_check(rc, "Error during matching");
                W.cb.nb_records_tot += cb.nb_records_tot, W.cb.nb_bases += cb.nb_bases;
                W.cb.nb_hits_tot[0] += cb.nb_hits_tot[0], W.cb.nb_hits_tot[1] += cb.nb_hits_tot[1];
                W.cb.nb_records_hit[0] += cb.nb_records_hit[0], W.cb.nb_records_hit[1] += cb.nb_records_hit[1];
                W.cb.nb_records_extracted += cb.nb_records_extracted;
                for (size_t q = 0; q < W.cnt.size(); ++q) W.cnt[q] += cnt_b[q];
                break;
            }
            if (lg->active)
                for (uint64_t r = 0; r < n_rows; ++r) {
                    mk_row row = rows[r];
                    row.rec += b0;
                    W.rows.push_back(row);
                }
            b0 = b1;
        }
        W.n_rows = W.rows.size();
    }

    void device_window(Win &W, mk_matcher *m, mk_codec *codec) {
        mk_window_source src[2];
        memset(src, 0, sizeof(src));
        uint64_t cap_text = 0;
        bool any_bgzf = false;
        for (int i = 0; i < n_in; ++i) {
            Side &S = W.side[i];
            mk_window_source &Q = src[i];
            Q.head = (const uint8_t *)S.head.data(), Q.n_head = S.head.size();
            uint64_t body = S.n_body;
            if (!S.grp.empty()) {
                Q.bgzf = in[i].s.source().file_bytes(), Q.n_bgzf = in[i].s.source().file_size();
                Q.members = S.grp.data(), Q.n_members = S.grp.size();
                for (auto &e : S.grp) body += e.isize;
                any_bgzf = true;
            } else if (S.n_dev_body) {
                Q.device_text = S.dev_body, Q.n_device_text = S.n_dev_body;
                body += S.n_dev_body;
            } else {
                Q.text = (const uint8_t *)S.body, Q.n_text = S.n_body;
            }
            Q.ends_at_record = S.ends ? 1 : 0;
            cap_text = std::max(cap_text, S.head.size() + body);
            if (in[i].dev_text()) {  // the host never sees this text: what it needs of it comes back
                if (whole_text) {
                    S.text.resize(S.head.size() + body + 16);
                    Q.all = (uint8_t *)S.text.data(), Q.all_cap = S.text.size();
                    S.layout = Side::WHOLE;
                } else {
                    S.text.resize(std::max<uint64_t>(1u << 20, std::min<uint64_t>(S.head.size() + body, std::max<uint64_t>(64u << 20, (S.head.size() + body) / 8))));
                    Q.kept = (uint8_t *)S.text.data(), Q.kept_cap = S.text.size();
                    S.layout = Side::PACKED_KEPT;
                }
                S.tail.resize(std::max<size_t>(S.tail.size(), 1u << 20));
                Q.tail = (uint8_t *)S.tail.data(), Q.tail_cap = S.tail.size();
            }
        }
        if (cap_text >= 0xFFFFFFF0ull) return host_window(W, m);  // (a FASTA record of 4 GiB or more: the host path's own limits apply)
        // (a guess the call corrects: MK_E_CAPACITY comes back before anything is scanned)
        uint64_t rec_cap = std::max<uint64_t>(rec_cap_seen.load(), cap_text / (fastq ? 192 : 1024) + 16);
        W.rows.resize(std::max<size_t>(W.rows.size(), 4096));
        W.cnt.assign(pats->list.size(), 0);
        uint32_t status = 0;
        for (int attempt = 0;; ++attempt) {
            for (int i = 0; i < n_in; ++i) {
                W.side[i].rec_start.resize(rec_cap + 1);
                src[i].rec_start = W.side[i].rec_start.data();
            }
            if (W.keep.size() < rec_cap) W.keep.resize(rec_cap);
            memset(&W.cb, 0, sizeof(W.cb));
            std::fill(W.cnt.begin(), W.cnt.end(), 0);
            const int rc = mk_extract_window(m, any_bgzf ? codec : nullptr, fastq ? MK_TEXT_FASTQ : MK_TEXT_FASTA, (uint32_t)n_in, src, lg->active,
                                             a.invert_match, rec_cap, &W.n_rec, W.keep.data(), W.rows.data(), W.rows.size(), &W.n_rows, &W.cb,
                                             W.cnt.data(), &status);
            if (rc == MK_E_CAPACITY && attempt < 8) {  // the call states every need: grow what was too small, once more
                bool grown = false;
                if (W.n_rec > rec_cap) rec_cap = W.n_rec + W.n_rec / 16, grown = true, rec_cap_seen = std::max<uint64_t>(rec_cap_seen.load(), rec_cap);
                if (W.n_rows > W.rows.size()) W.rows.resize(W.n_rows), grown = true;
                for (int i = 0; i < n_in; ++i) {
                    Side &S = W.side[i];
                    if (src[i].tail && src[i].n_tail > src[i].tail_cap) {
                        S.tail.resize(src[i].n_tail + (1u << 16));
                        src[i].tail = (uint8_t *)S.tail.data(), src[i].tail_cap = S.tail.size();
                        grown = true;
                    }
                    if (src[i].kept && src[i].n_kept_bytes > src[i].kept_cap) {
                        S.text.resize(src[i].n_kept_bytes + (src[i].n_kept_bytes >> 3));
                        src[i].kept = (uint8_t *)S.text.data(), src[i].kept_cap = S.text.size();
                        grown = true;
                    }
                }
                if (grown) continue;
            }
            if (rc == MK_E_CORRUPT) bail("Error while decompressing " + in[src[1].n_members && !src[0].n_members ? 1 : 0].path);
            if (rc == MK_E_UNSUPPORTED || rc == MK_E_NOMEM) {
                // too much for the device in one piece (a pair of windows above 2 GiB under BNDMq; no memory for the window): the
                // host path takes this window in batches, and BGZF windows shrink from here on
                if (rc == MK_E_NOMEM) {
                    std::lock_guard<std::mutex> lk(mu);
                    bgzf_target = std::max<uint64_t>(64ull << 20, bgzf_target / 2);
                }
                return host_window(W, m);
            }
            mk_check(rc, "Error during matching");
            break;
        }
        if (status != 0) return host_window(W, m);
        for (int i = 0; i < n_in; ++i) {
            Side &S = W.side[i];
            S.n_window = src[i].n_window, S.n_used = src[i].n_used;
            S.rec_start.resize(W.n_rec + 1);
            if (in[i].dev_text()) {
                S.tail.resize(src[i].n_tail);
                if (S.layout == Side::PACKED_KEPT) S.text.resize(src[i].n_kept_bytes);
                if (S.layout == Side::WHOLE) S.text.resize(src[i].n_window);
            } else {  // the host holds this text: the tail is the rest of head ++ body
                S.tail.clear();
                const uint64_t nh = S.head.size();
                if (S.n_used < nh) S.tail.insert(S.tail.end(), S.head.begin() + (ptrdiff_t)S.n_used, S.head.end());
                const uint64_t from = S.n_used > nh ? S.n_used - nh : 0;
                if (from < S.n_body) S.tail.insert(S.tail.end(), S.body + from, S.body + S.n_body);
                // ... and what the writer will need of it -- the kept records (all of it when the rows name records that are not
                // kept) -- is copied out here, so that the page-locked buffer goes back to the reader now, not after the write-out
                pack_host_text(W, S);
            }
        }
    }

    // S.text = the kept records of a host-held window back to back (layout PACKED_KEPT), or its whole text (WHOLE)
    void pack_host_text(const Win &W, Side &S) {
        const uint64_t nh = S.head.size();
        auto piece = [&](uint64_t b, uint64_t e, char *dst) {  // text[b, e) of head ++ body
            if (b < nh) {
                const uint64_t k = std::min(e, nh) - b;
                memcpy(dst, S.head.data() + b, (size_t)k);
                dst += k, b += k;
            }
            if (b < e) memcpy(dst, S.body + (b - nh), (size_t)(e - b));
        };
        if (whole_text) {
            S.text.resize(S.n_used);
            const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), (size_t)(S.n_used >> 22) + 1));
            run_threads(T, [&](size_t t) { piece(S.n_used * t / T, S.n_used * (t + 1) / T, S.text.data() + S.n_used * t / T); });
            S.layout = Side::WHOLE;
            return;
        }
        std::vector<uint64_t> kept, at;
        uint64_t total = 0;
        for (uint64_t r = 0; r < W.n_rec; ++r)
            if (W.keep[r]) {
                kept.push_back(r);
                at.push_back(total);
                total += S.rec_start[r + 1] - S.rec_start[r];
            }
        S.text.resize(total);
        const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), (size_t)(total >> 22) + 1));
        run_threads(T, [&](size_t t) {
            for (size_t q = kept.size() * t / T; q < kept.size() * (t + 1) / T; ++q) piece(S.rec_start[kept[q]], S.rec_start[kept[q] + 1], S.text.data() + at[q]);
        });
        S.layout = Side::PACKED_KEPT;
    }

    void device_loop(size_t d) {
        mk_matcher *m = ms[d];
        for (;;) {
            std::unique_ptr<Win> W;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || !queue[d].empty() || reader_done; });
                if (failed) return;
                if (queue[d].empty()) return;  // (reader_done)
                W = std::move(queue[d].front());
                queue[d].pop_front();
            }
            const double t_call = PhaseTimer::now();
            if (W->error.empty() && !W->by_host) {
                // an error of this window (a malformed record, a damaged member) is raised when the writer reaches the window: the
                // reference has written everything in front of it by then
                try {
                    device_window(*W, m, codecs.empty() ? nullptr : codecs[d]);
                } catch (const Error &e) {
                    W->error = e.what()[0] ? e.what() : "error";
                    for (int i = 0; i < n_in; ++i) W->side[i].tail.clear();
                }
            }
            if (timing) {
                float ms4[4] = {0, 0, 0, 0};
                (void)mk_matcher_batch_times(m, ms4);
                fprintf(stderr, "[timing]   device %zu, window %llu: call %.1f ms (upload %.1f, device %.1f, download %.1f, host %.1f)%s\n", d,
                        (unsigned long long)W->k, (PhaseTimer::now() - t_call) * 1e3, ms4[0], ms4[1], ms4[2], ms4[3], W->by_host ? " [host parser]" : "");
            }
            std::lock_guard<std::mutex> lk(mu);
            for (int i = 0; i < n_in; ++i) {  // (the text the writer needs has been copied out of the staging buffers: device and host windows alike)
                Side &S = W->side[i];
                if (S.pin >= 0) in[i].free_pins.push_back(S.pin);
                S.pin = -1, S.body = nullptr, S.n_body = 0;
                std::vector<char>().swap(S.own);
                if (!chained) std::vector<char>().swap(S.head);
            }
            if (chained) {
                std::vector<std::vector<char>> t(2);
                for (int i = 0; i < n_in; ++i) t[i] = W->side[i].tail;  // (a copy: the window itself goes to the writer)
                tails[W->k] = std::move(t);
            }
            const uint64_t k = W->k;
            done[k] = std::move(W);
            cv.notify_all();
        }
    }

    // ---- writer side -----------------------------------------------------------------------------------------------------------
    // where record r of a device-indexed side lies: (text piece, its size, begin, end)
    struct Span {
        const char *p;
        uint64_t n, b, e;
    };
    static Span span_of(const Side &S, uint64_t b, uint64_t e) {
        if (S.layout == Side::HEAD_BODY) bail("Error during matching: a window reached the writer without its text");
        return Span{S.text.data(), S.text.size(), b, e};
    }

    void write_window(Win &W, Sink &w1, Sink &w2, const std::string &name1, const std::string &name2, PhaseTimer &tm) {
        const uint64_t n = W.n_rec;
        Sink *w[2] = {&w1, &w2};
        const std::string *names[2] = {&name1, &name2};
        if (W.by_host) {
            emit_log_rows(
                *lg, *pats, W.rows.data(), lg->active ? W.n_rows : 0,
                [&](const mk_row &r) {
                    const FastxFile &ff = W.side[r.file].parsed;
                    const auto &rec = ff.recs[r.rec];
                    return std::pair<const char *, size_t>(ff.data + rec.id_b, rec.id_e - rec.id_b);
                },
                [&](const mk_row &r) -> const std::string & { return *names[r.file]; });
            if (!a.suppress_output)
                for (uint64_t r = 0; r < n; ++r)
                    if (W.keep[r])
                        for (int i = 0; i < n_in; ++i) W.side[i].parsed.write(r, *w[i]);
            return;
        }
        // record r of side i in the text the writer holds: its own table for packed kept records
        std::vector<uint64_t> kept_of;               // window index of every kept record (PACKED_KEPT)
        std::vector<uint64_t> packed_start[2];
        bool packed = false;
        for (int i = 0; i < n_in; ++i) packed = packed || W.side[i].layout == Side::PACKED_KEPT;
        if (packed) {
            for (uint64_t r = 0; r < n; ++r)
                if (W.keep[r]) kept_of.push_back(r);
            for (int i = 0; i < n_in; ++i) {
                const Side &S = W.side[i];
                if (S.layout != Side::PACKED_KEPT) continue;
                uint64_t at = 0;
                for (uint64_t r : kept_of) {
                    packed_start[i].push_back(at);
                    at += S.rec_start[r + 1] - S.rec_start[r];
                }
                packed_start[i].push_back(at);
                if (at != S.text.size()) bail("Error during matching: the kept records' text does not have the size of its record table");
            }
        }
        auto record_span = [&](int i, uint64_t r) -> Span {
            const Side &S = W.side[i];
            if (S.layout == Side::PACKED_KEPT) {
                const size_t q = (size_t)(std::lower_bound(kept_of.begin(), kept_of.end(), r) - kept_of.begin());
                if (q >= kept_of.size() || kept_of[q] != r) bail("Error during matching: a log row names a record whose text did not come back");
                return Span{S.text.data(), S.text.size(), packed_start[i][q], packed_start[i][q + 1]};
            }
            return span_of(S, S.rec_start[r], S.rec_start[r + 1]);
        };
        // id of a record: its header line without the marker and the line end
        auto id_of = [&](const mk_row &row) {
            const Span sp = record_span((int)row.file, row.rec);
            const uint64_t b = sp.b + 1;
            const char *nl = (const char *)memchr(sp.p + b, '\n', (size_t)(sp.e - b));
            uint64_t e = nl ? (uint64_t)(nl - sp.p) : sp.e;
            if (e > b && sp.p[e - 1] == '\r') --e;
            return std::pair<const char *, size_t>(sp.p + b, (size_t)(e - b));
        };
        emit_log_rows(*lg, *pats, W.rows.data(), lg->active ? W.n_rows : 0, id_of, [&](const mk_row &r) -> const std::string & { return *names[r.file]; });
        tm.mark("window: log rows");
        if (!a.suppress_output) {
            // kept records are written by the host reader's own code from the record's lines -- formatted by the host threads into
            // one buffer per thread and input, written in order (a window of 2 x 128 MB holds ~8 000 kept records at the usual 1 %,
            // a window of extracted reads 800 000)
            std::vector<uint64_t> kept;
            for (uint64_t r = 0; r < n; ++r)
                if (W.keep[r]) kept.push_back(r);
            const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), kept.size() / 1024));
            std::vector<std::string> out[2];
            out[0].resize(T), out[1].resize(T);
            run_threads(T, [&](size_t t) {
                FastxFile one;
                one.fastq = fastq;
                std::string line;
                for (size_t q = kept.size() * t / T; q < kept.size() * (t + 1) / T; ++q) {
                    const uint64_t r = kept[q];
                    for (int i = 0; i < n_in; ++i) {
                        const Span sp = record_span(i, r);
                        one.data = sp.p;
                        one.data_n = sp.n;
                        one.recs.clear();
                        one.parse_span(sp.b, sp.e);
                        if (one.recs.size() != 1) bail("Error during matching: the device's record table and the host parser disagree");
                        const FastxFile::Rec &rec = one.recs[0];
                        // record.write(_, None) re-emits the lines (FASTQ: with a bare '+'): a record that is stored that way already
                        // (and ends in its line end) is written as one piece
                        const bool crlf = rec.id_e < sp.n && sp.p[rec.id_e] == '\r';
                        const uint64_t nl = crlf ? 2 : 1;
                        const bool verbatim = sp.p[sp.e - 1] == '\n' && (fastq ? (rec.qual_b == rec.raw_e + 2 * nl + 1 && rec.qual_e + nl == sp.e)
                                                                               : (rec.raw_e + nl == sp.e && rec.raw_b == rec.id_e + nl));
                        std::string &o = out[i][t];
                        if (verbatim) {
                            o.append(sp.p + sp.b, (size_t)(sp.e - sp.b));
                        } else {  // (FastxFile::write, into the buffer)
                            const char *eol = crlf ? "\r\n" : "\n";
                            o.append(fastq ? "@" : ">", 1).append(sp.p + rec.id_b, (size_t)(rec.id_e - rec.id_b)).append(eol, nl);
                            o.append(sp.p + rec.raw_b, (size_t)(rec.raw_e - rec.raw_b)).append(eol, nl);
                            if (fastq) o.append("+", 1).append(eol, nl).append(sp.p + rec.qual_b, (size_t)(rec.qual_e - rec.qual_b)).append(eol, nl);
                        }
                    }
                }
            });
            for (size_t t = 0; t < T; ++t)
                for (int i = 0; i < n_in; ++i) w[i]->write(out[i][t]);
        }
        tm.mark("window: records out");
    }
};

WindowExtract::~WindowExtract() {
    if (!impl) return;
    for (int i = 0; i < 2; ++i)
        if (impl->in[i].fd >= 0) close(impl->in[i].fd);
    // (page-locked buffers and codec handles are not released at the end of the run: unpinning costs more than the process has left to live)
    if (!g_process_is_ending) {
        for (mk_codec *c : impl->codecs) mk_codec_destroy(c);
        for (int i = 0; i < 2; ++i)
            if (impl->in[i].gz_codec) mk_codec_destroy(impl->in[i].gz_codec);
        for (int i = 0; i < 2; ++i)
            for (Pinned &p : impl->in[i].pins) mk_host_free(p.p);
    }
    delete impl;
}

// the first byte of a bgzip'ed text that is not a line end: zlib on its first members, here
static int first_text_byte_of_bgzf(const WindowSource &ws, const std::string &path) {
    for (size_t g = 0; g < ws.n_bgzf_members() && g < 64; ++g) {
        mk_bgzf_member e{};
        ws.bgzf_member_at(g, &e.data_off, &e.data_len, &e.isize, &e.crc);
        e.out_off = 0;
        if (!e.isize) continue;
        std::vector<char> out(e.isize);
        inflate_bgzf_members_host(ws.file_bytes(), &e, 1, out.data(), path);
        for (char ch : out)
            if (ch != '\n' && ch != '\r') return (unsigned char)ch;
    }
    return -1;
}

bool WindowExtract::prepare(const ExtractArgs &a, const std::vector<int> &devs) {
    impl = new Impl;
    Impl &J = *impl;
    J.a = a;
    J.devs = devs;
    J.n_in = a.in_fastq_2 ? 2 : 1;
    J.in[0].path = a.in_fastx;
    if (a.in_fastq_2) J.in[1].path = *a.in_fastq_2;
    const bool logs = a.out_log || a.json_log;
    const uint64_t window_bytes = (uint64_t)a.window_mb << 20;
    // (windows of 128 MB: the copy into pinned memory of window k + 1 overlaps upload + scan of window k; page-locking a buffer costs
    // ~0.1 ms per MB, so the staging buffers stay small.  With logs every window also pays the ordering / row / count round trips:
    // fewer, larger windows)
    J.plain_target = std::min<uint64_t>(window_bytes, logs ? 256ull << 20 : 128ull << 20);
    J.whole_text = a.invert_match && logs;
    // (a launch of the inflate kernel lasts as long as its slowest member whatever it holds: few, large windows -- and when the text
    // stays on the device, --window-mb's reason, the host's memory, does not apply: 3 GiB unless the flag was given)
    J.bgzf_target = a.window_mb_given ? std::max<uint64_t>(1u << 16, std::min<uint64_t>(window_bytes, 3ull << 30))
                                      : (J.whole_text ? 1ull << 30 : 3ull << 30);
    if (J.n_in == 2) J.bgzf_target = std::max<uint64_t>(1u << 16, J.bgzf_target / 2);
    int kind[2] = {-1, -1};
    for (int i = 0; i < J.n_in; ++i) {
        Input &I = J.in[i];
        I.s.open(I.path);
        I.bgzf_dev = !a.host_codec && I.s.raw_is_bgzf();
        if (I.bgzf_dev) {
            kind[i] = first_text_byte_of_bgzf(I.s.source(), I.path);
        } else {
            I.have_first = I.s.raw_fill(std::min(J.plain_target, kFirstWindow), &I.first_text, &I.first_n, &I.first_resume);
            if (I.have_first) kind[i] = I.s.raw_fastq() ? '@' : '>';
        }
        if (I.s.raw_is_plain()) I.fd = open(I.path.c_str(), O_RDONLY);
    }
    for (int i = 0; i < J.n_in; ++i)
        if (kind[i] != '@' && kind[i] != '>') return false;  // empty, or not FASTA / FASTQ: the host reader words what it is
    if (J.n_in == 2 && kind[0] != kind[1]) return false;
    J.fastq = kind[0] == '@';
    // (a plain gzip input may become device text once the HIP runtime is up, run(): its windows then end anywhere too)
    J.chained = J.n_in == 2 || J.in[0].bgzf_dev || (!a.host_codec && J.in[0].s.raw_is_gzip());
    return true;
}

void WindowExtract::run(const ExtractArgs &a, const Patterns &pats, Loggers &lg, const std::vector<mk_matcher *> &ms, const std::vector<int> &devs,
                        Sink &w1, Sink &w2, const std::string &name1, const std::string &name2, std::vector<mk_counters> &dev_c,
                        std::vector<std::vector<uint32_t>> &dev_counts, PhaseTimer &tm) {
    Impl &J = *impl;
    J.pats = &pats, J.lg = &lg, J.ms = ms;
    J.batch_bytes = (uint64_t)a.batch_mb << 20;
    const size_t N = devs.size();
    J.queue.resize(N);
    bool any_bgzf = false;
    for (int i = 0; i < J.n_in; ++i) any_bgzf = any_bgzf || J.in[i].bgzf_dev;
    if (any_bgzf) {
        for (size_t d = 0; d < N; ++d) {
            mk_codec *c = nullptr;
            mk_check(mk_codec_create(devs[d], &c), "Error setting up the BGZF codec");
            J.codecs.push_back(c);
        }
        // a window's text, its sequences, its members and its tables all live on the device: keep it to a sixth of what is free there
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(devs[0]) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
            J.bgzf_target = std::min<uint64_t>(J.bgzf_target, std::max<uint64_t>(64ull << 20, (uint64_t)free_b / (6 * (uint64_t)J.n_in)));
    }
    // a plain gzip input: ONE DEFLATE stream, which zlib walks from the front at 0.5 GB/s -- the device inflates it in parallel
    // pieces (mk_gzip_inflate_device) and keeps the text; what it does not take (several members, no findable block starts, a
    // stream the buffers do not hold, a damaged file) stays with the zlib reader, which also words the errors
    run_threads((size_t)J.n_in, [&](size_t ii) {  // (the two files of a pair side by side: a handle and a stream each)
        const int i = (int)ii;
        Input &I = J.in[i];
        if (a.host_codec || !I.s.raw_is_gzip()) return;
        mk_check(mk_codec_create(devs[0], &I.gz_codec), "Error setting up the gzip codec");
        uint64_t text_bytes = 0;
        uint32_t taken = 0;
        const double t_gz = PhaseTimer::now();
        mk_check(mk_gzip_inflate_device(I.gz_codec, I.s.source().file_bytes(), I.s.source().file_size(), &text_bytes, &taken), "Error inflating the input");
        if (J.timing) {
            uint32_t seg = 0;
            float ms5[5] = {0, 0, 0, 0, 0};
            (void)mk_gzip_info(I.gz_codec, &seg, ms5);
            fprintf(stderr, "[timing]   gzip input %d on the device: %s, %.3f s (%u pieces; upload %.1f, block search %.1f, pieces %.1f, resolution %.1f, CRC %.1f ms)\n", i,
                    taken ? "taken" : "NOT taken (zlib reads it)", PhaseTimer::now() - t_gz, seg, ms5[0], ms5[1], ms5[2], ms5[3], ms5[4]);
        }
        if (!taken) {
            mk_codec_destroy(I.gz_codec);
            I.gz_codec = nullptr;
            return;
        }
        I.gz_dev = true;
        I.have_first = false;  // (the window zlib had read for prepare() is not needed)
        I.gz_text = (const uint8_t *)mk_gzip_text_device(I.gz_codec, &I.gz_bytes);
        I.gz_next = 0;
        I.exhausted = I.gz_bytes == 0;
    });
    bool any_dev_text = false;
    for (int i = 0; i < J.n_in; ++i) any_dev_text = any_dev_text || J.in[i].dev_text();
    if (any_dev_text && !any_bgzf) {  // (the same bound as for BGZF windows: the text, its sequences and tables live on the device)
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(devs[0]) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
            J.bgzf_target = std::min<uint64_t>(J.bgzf_target, std::max<uint64_t>(64ull << 20, (uint64_t)free_b / (6 * (uint64_t)J.n_in)));
    }
    // page-locked staging buffers per plain / inflated input: one being filled, one per device in flight, one being written out
    for (int i = 0; i < J.n_in; ++i) {
        Input &I = J.in[i];
        if (I.dev_text()) continue;
        I.pins.resize(N + 1);
    }
    // (page-locking costs ~0.2 ms per MB: the buffers are made on threads of their own, one per input, beside the first window --
    // which goes up from where it lies --, and handed to the reader as they appear)
    std::vector<std::thread> pin_threads;
    for (int i = 0; i < J.n_in; ++i) {
        if (J.in[i].dev_text()) continue;
        pin_threads.emplace_back([&J, i] {
            Input &I = J.in[i];
            for (size_t q = 0; q < I.pins.size(); ++q) {
                try {
                    I.pins[q].need(J.plain_target + (1u << 20));
                } catch (const Error &e) {
                    J.fail_all(e.what());
                    return;
                }
                std::lock_guard<std::mutex> lk(J.mu);
                I.free_pins.push_back((int)q);
                J.cv.notify_all();
            }
        });
    }
    std::vector<std::thread> th;
    auto guarded = [&](auto fn) {
        return [&J, fn] {
            try {
                fn();
            } catch (const Error &e) {
                J.fail_all(e.what()[0] ? e.what() : "error");
            } catch (const std::exception &e) {
                J.fail_all(std::string("Error: ") + e.what());
            }
        };
    };
    th.emplace_back(guarded([&J] { J.reader_loop(); }));
    for (size_t d = 0; d < N; ++d) th.emplace_back(guarded([&J, d] { J.device_loop(d); }));
    std::string writer_error;
    try {
        for (uint64_t k = 0;; ++k) {
            std::unique_ptr<Win> W;
            {
                std::unique_lock<std::mutex> lk(J.mu);
                J.cv.wait(lk, [&] { return J.failed || J.done.count(k) || (J.reader_done && k >= J.n_windows); });
                if (J.failed) break;
                if (!J.done.count(k)) break;  // every window is written
                W = std::move(J.done[k]);
                J.done.erase(k);
            }
            if (!W->error.empty()) bail(W->error);
            tm.mark(W->by_host ? "window: host parser + scan" : "window: H2D + index + scan + D2H");
            mk_counters &c = dev_c[W->dev];
            c.nb_records_tot += W->cb.nb_records_tot, c.nb_bases += W->cb.nb_bases;
            c.nb_hits_tot[0] += W->cb.nb_hits_tot[0], c.nb_hits_tot[1] += W->cb.nb_hits_tot[1];
            c.nb_records_hit[0] += W->cb.nb_records_hit[0], c.nb_records_hit[1] += W->cb.nb_records_hit[1];
            c.nb_records_extracted += W->cb.nb_records_extracted;
            for (size_t q = 0; q < W->cnt.size(); ++q) dev_counts[W->dev][q] += W->cnt[q];
            J.write_window(*W, w1, w2, name1, name2, tm);
            {
                std::lock_guard<std::mutex> lk(J.mu);
                J.written = k + 1;
                J.cv.notify_all();
            }
        }
    } catch (const Error &e) {
        writer_error = e.what()[0] ? e.what() : "error";
        J.fail_all(writer_error);
    }
    for (auto &t : th) t.join();
    for (auto &t : pin_threads) t.join();
    if (!writer_error.empty()) bail(writer_error);
    if (J.failed) bail(J.failure);
}

}  // namespace cli
