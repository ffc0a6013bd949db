// tag_windows.cpp -- `merkurio tag` BAM -> BAM (or -S) with the records resident on the device (r05; SURVEY.md §8 rows a11 / f-3;
// the reader loop, process_record and the writer of src/cmd_tag.rs:503-615, :387-497, :254-271).
//
// The r04 path inflated a window of the BAM into host memory (device codec), indexed its record chain and un-nibbled the
// sequences on the host threads, sent those to the scan, appended the tags on the host threads and handed the records back to
// the device to be deflated: the text crossed the host boundary three times.  Here a window is the compressed members AS THEY ARE
// STORED: mk_tag_bam_window inflates them, indexes the records, unpacks, scans, tags and deflates on the device; what comes back
// is the window's last unfinished record (the next window's head), the log rows with their record names, and finished BGZF
// members that go to the file as they are.  While window k is on the device, a host thread copies the members of window k + 1
// into page-locked memory (a mapped file is not a DMA source) and the writer thread writes window k - 1.
//
// A window the device refuses -- a record that fails the parser's checks, optional fields that do not parse, a kept record that
// already carries the tag, a damaged member -- hands the input back to the host reader AT THAT WINDOW'S FIRST BYTE
// (SamFile::seek_bam): the r04 path takes the rest of the file and words the reference's errors.
#include <algorithm>
#include <cstring>
#include <future>

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
        if (p) mk_host_free(p);
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

void plan_window(const WindowSource &src, size_t m0, uint64_t head_bytes, uint64_t window_bytes, WindowMembers &W) {
    const size_t n_mem = src.n_bgzf_members();
    W.m0 = m0;
    W.mem.clear();
    W.text = 0;
    W.file_lo = W.file_hi = 0;
    size_t m1 = m0;
    uint64_t text = head_bytes;
    while (m1 < n_mem && (text < window_bytes || m1 == m0)) {
        uint64_t off;
        uint32_t len, isize, crc;
        src.bgzf_member_at(m1, &off, &len, &isize, &crc);
        if (m1 == m0) W.file_lo = off;
        W.file_hi = off + len;
        W.mem.push_back(mk_bgzf_member{off - W.file_lo, W.text, len, isize, crc, 0});
        W.text += isize;
        text += isize;
        ++m1;
    }
    W.m1 = m1;
}

// file[lo, hi) -> dst on all host threads (first touch of the mapping's pages included)
void copy_in(const uint8_t *file, uint64_t lo, uint64_t hi, uint8_t *dst) {
    const uint64_t n = hi - lo;
    const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), (size_t)(n >> 22) + 1));
    run_threads(T, [&](size_t t) { memcpy(dst + n * t / T, file + lo + n * t / T, (size_t)(n * (t + 1) / T - n * t / T)); });
}

}  // namespace

bool tag_bam_windows_on_device(const TagArgs &a, SamFile &sam, mk_matcher *m, int device, Loggers &lg, const Patterns &pats, const std::string &in_name,
                               BamWriter *bw, mk_counters &c, std::vector<uint32_t> &counts, uint64_t window_bytes) {
    const WindowSource &src = sam.source();
    const size_t n_mem = src.n_bgzf_members();
    const uint8_t *file = src.file_bytes();
    const bool timing = getenv("MERKURIO_TIMING") != nullptr;
    size_t m0 = src.next_member();
    uint64_t n_pending = 0;
    const char *pend = sam.bam_pending(&n_pending);
    std::vector<uint8_t> head(pend, pend + n_pending);
    mk_codec *codec = nullptr;
    if (mk_codec_create(device, &codec) != MK_OK) bail(std::string("Error during BAM record parsing: ") + mk_last_error());
    struct CodecGuard {
        mk_codec *c;
        ~CodecGuard() {
            if (!g_process_is_ending) mk_codec_destroy(c);
        }
    } guard{codec};
    std::vector<uint8_t> tail(1u << 20), names(1u << 16);
    std::vector<mk_row> rows(4096);
    std::vector<uint64_t> row_name(4096);
    PinnedBuffer stage[2];
    WindowMembers W[2];
    std::vector<uint8_t> outs[2];  // the output members of a window (recycled by the writer thread: BamWriter::take_raw_buffer)
    // (BAM-shaped text deflates to a third; a window that does not fit is done again with the size it asked for)
    auto out_guess = [](uint64_t text) { return text / 2 + (4u << 20); };
    int cur = 0;
    // the first window is small (nothing runs beside its copy), the later ones are copied beside their predecessors
    plan_window(src, m0, head.size(), std::min<uint64_t>(window_bytes, 64ull << 20), W[cur]);
    if (!W[cur].mem.empty()) {
        stage[cur].need(W[cur].file_hi - W[cur].file_lo);
        copy_in(file, W[cur].file_lo, W[cur].file_hi, stage[cur].p);
    }
    if (bw) outs[cur] = bw->take_raw_buffer(out_guess(head.size() + W[cur].text));
    double t_dev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n_windows = 0;
    bool first = true;
    while (W[cur].m1 > W[cur].m0 || (first && !head.empty())) {
        first = false;
        WindowMembers &X = W[cur];
        const bool last = X.m1 >= n_mem;
        // the next window's members travel into the other staging buffer beside this window's device work
        std::future<void> next;
        if (!last) {
            plan_window(src, X.m1, 0, window_bytes, W[cur ^ 1]);
            next = std::async(std::launch::async, [&, nx = cur ^ 1] {
                stage[nx].need(W[nx].file_hi - W[nx].file_lo);
                copy_in(file, W[nx].file_lo, W[nx].file_hi, stage[nx].p);
                if (bw) outs[nx] = bw->take_raw_buffer(out_guess(W[nx].text + (1u << 20)));
            });
        } else {
            W[cur ^ 1].m0 = W[cur ^ 1].m1 = X.m1;
            W[cur ^ 1].mem.clear();
        }
        struct Join {
            std::future<void> &f;
            ~Join() {
                if (f.valid()) f.wait();
            }
        } join{next};
        mk_bam_window w;
        memset(&w, 0, sizeof(w));
        w.head = head.data(), w.n_head = head.size();
        w.bgzf = stage[cur].p, w.n_bgzf = X.file_hi - X.file_lo;
        w.members = X.mem.data(), w.n_members = X.mem.size();
        w.last = last, w.filter_matching = a.filter_matching, w.invert = a.invert_match;
        w.tag[0] = (uint8_t)a.tag[0], w.tag[1] = (uint8_t)a.tag[1];
        std::vector<uint8_t> &out = outs[cur];
        uint32_t status = 0;
        int rc;
        for (;;) {
            w.tail = tail.data(), w.tail_cap = tail.size();
            w.out = bw ? out.data() : nullptr, w.out_cap = bw ? out.size() : 0;
            w.rows = rows.data(), w.rows_cap = rows.size(), w.row_name = row_name.data(), w.names = names.data(), w.names_cap = names.size();
            rc = mk_tag_bam_window(m, codec, &w, lg.active, &c, counts.data(), &status);
            if (rc != MK_E_CAPACITY) break;
            bool grew = false;
            if (w.n_tail > tail.size()) tail.resize(w.n_tail + (1u << 20)), grew = true;
            if (bw && w.out_len > out.size()) out.resize(w.out_len), grew = true;
            if (w.n_rows > rows.size()) rows.resize(w.n_rows), row_name.resize(w.n_rows), grew = true;
            if (w.n_names_bytes > names.size()) names.resize(w.n_names_bytes), grew = true;
            if (!grew) break;
        }
        if (rc == MK_E_CORRUPT || (rc == MK_OK && status != 0)) {
            // not for the device: the host reader takes the input from this window's first byte (and words what is wrong with it)
            if (timing)
                fprintf(stderr, "[timing] window %llu left to the host reader (%s)\n", (unsigned long long)n_windows,
                        rc == MK_E_CORRUPT ? "a damaged member" : status & 1 ? "record chain" : status & 2 ? "optional fields" : status & 4 ? "existing tag" : "unfinished record");
            if (next.valid()) next.wait();  // (it reads the member table of `src`)
            sam.seek_bam(X.m0, (const char *)head.data(), head.size());
            return false;
        }
        mk_check(rc, "Error during matching");
        for (int k = 0; k < 8; ++k) t_dev[k] += w.ms[k];
        ++n_windows;
        if (lg.active)
            emit_log_rows(
                lg, pats, rows.data(), w.n_rows,
                [&](const mk_row &r) {
                    const char *nm = (const char *)names.data() + row_name[&r - rows.data()];
                    return std::pair<const char *, size_t>(nm, strlen(nm));
                },
                [&](const mk_row &) -> const std::string & { return in_name; });
        if (bw && w.out_len) bw->put_members(std::move(out), w.out_len);
        head.assign(tail.begin(), tail.begin() + w.n_tail);
        if (next.valid()) next.get();
        cur ^= 1;
    }
    if (timing)
        fprintf(stderr,
                "[timing] %llu windows on the device: upload %.3f, inflate %.3f, record index %.3f, unpack + scan + sets %.3f, tag + pack %.3f, deflate %.3f, "
                "download %.3f s\n",
                (unsigned long long)n_windows, t_dev[0] / 1e3, t_dev[1] / 1e3, t_dev[2] / 1e3, t_dev[3] / 1e3, t_dev[4] / 1e3, t_dev[5] / 1e3, t_dev[6] / 1e3);
    return true;
}

}  // namespace cli
