// extract_common.hpp -- what the two extract drivers share (commands.cpp: the host-parsed path; extract_windows.cpp: text windows
// indexed on the device): the pattern list, the loggers, the formatting of a batch's log rows, timing marks.
#pragma once
#include <ctime>
#include <cstdlib>
#include <future>
#include <string>
#include <algorithm>
#include <vector>

#include "../../../include/merkurio_hip.h"
#include "commands.hpp"
#include "io.hpp"

namespace cli {

// MERKURIO_TIMING=1: phase wall times on stderr (where does an end-to-end run spend its time)
struct PhaseTimer {
    bool on = getenv("MERKURIO_TIMING") != nullptr;
    double t0 = now();
    static double now() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + ts.tv_nsec * 1e-9;
    }
    void mark(const char *what) {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[timing] %-28s %8.3f s\n", what, t - t0);
        t0 = t;
    }
};


inline void mk_check(int rc, const char *what) {
    if (rc != MK_OK) bail(std::string(what) + ": " + mk_last_error());
}

struct Patterns {
    std::vector<std::string> list;
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off;
};

struct Loggers {
    TextLogger text;
    JsonLogger json;
    bool active = false;
    bool has_json = false;
};

// The log rows of a batch (reference emission order), formatted by the host threads -- a batch in which every read
// hits carries millions of rows, and one thread building them took four times the rest of the run -- and written
// in order.  id_of(row) -> the record id's bytes; file_of(row) -> the file name to log.
template <class IdOf, class FileOf>
inline void emit_log_rows(Loggers &lg, const Patterns &pats, const mk_row *rows, uint64_t n_rows, IdOf id_of, FileOf file_of) {
    if (!lg.active || n_rows == 0) return;
    const bool text = lg.text.out != nullptr, json = lg.has_json;
    const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), n_rows / 2048));
    std::vector<std::string> tb(T), jb(T);
    const bool first_row_of_log = json && lg.json.first;
    run_threads(T, [&](size_t t) {
        const uint64_t lo = n_rows * t / T, hi = n_rows * (t + 1) / T;
        if (text) tb[t].reserve((hi - lo) * 96);
        if (json) jb[t].reserve((hi - lo) * 176);
        for (uint64_t k = lo; k < hi; ++k) {
            const mk_row &r = rows[k];
            const std::pair<const char *, size_t> id = id_of(r);
            const std::string &file = file_of(r);
            if (text) TextLogger::format(tb[t], file, id.first, id.second, pats.list[r.pat], r.pos);
            if (json) JsonLogger::format(jb[t], !(first_row_of_log && k == 0), file, id.first, id.second, pats.list[r.pat], r.pos);
        }
    });
    for (size_t t = 0; t < T; ++t) {
        if (text) lg.text.out->write(tb[t]);
        if (json) lg.json.out->write(jb[t]);
    }
    if (json) lg.json.first = false;
}


// ---- extract over text windows indexed on the device (extract_windows.cpp) ------------------------------------------------------
// Everything `extract` reads -- one FASTQ / FASTA file or a pair, plain, gzip or bgzip'ed, on one GPU or several -- goes to the
// device as windows of raw text (mk_extract_window); the host parser takes single windows the device refuses.
struct WindowExtract {
    struct Impl;
    Impl *impl = nullptr;
    ~WindowExtract();
    // Opens the inputs and reads their first windows (beside the HIP start-up).  false: these inputs are not for this path
    // (neither FASTQ nor FASTA, or a pair of different kinds) -- nothing was consumed, the caller's host reader takes the job.
    bool prepare(const ExtractArgs &a, const std::vector<int> &devs);
    // The job: windows -> matchers (window k on device k mod N) -> log rows and kept records in record order.
    // c / counts: the job's counters (per device: dev_c / dev_counts, reduced by the caller when there are several).
    void run(const ExtractArgs &a, const Patterns &pats, Loggers &lg, const std::vector<mk_matcher *> &ms, const std::vector<int> &devs, Sink &w1,
             Sink &w2, const std::string &name1, const std::string &name2, std::vector<mk_counters> &dev_c,
             std::vector<std::vector<uint32_t>> &dev_counts, PhaseTimer &tm);
};

// tag_windows.cpp: BAM input whose records stay on the device (mk_tag_bam_window); bw == nullptr: no output (-S).
// true: the whole input has been processed; false: `sam` has been positioned where the host reader has to carry on.
// handles: matchers with the device they live on and where their windows' counters are added (two handles per device keep two
// windows per device in flight; window k runs on handle k mod handles.size()).
struct SamFile;
struct BamWriter;
struct TagHandle {
    mk_matcher *m;
    int device;
    mk_counters *counters;
    std::vector<uint32_t> *pattern_counts;
};
bool tag_bam_windows_on_device(const TagArgs &a, SamFile &sam, const std::vector<TagHandle> &handles, Loggers &lg, const Patterns &pats,
                               const std::string &in_name, BamWriter *bw, uint64_t window_bytes);

}  // namespace cli
