// commands.cpp -- the `extract` and `tag` drivers of the C++ host program: everything around
// the hot path (argument semantics, record I/O, log / JSON emission, summaries), restated from
// src/cmd_extract.rs:143-717 and src/cmd_tag.rs:155-689.  All matching goes through the C ABI
// (mk_extract_single / mk_extract_paired / mk_tag_records): no text is searched on the host.
#include "commands.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <future>
#include <memory>
#include <mutex>

#include <algorithm>
#include <cstring>
#include <ctime>

#include <hip/hip_runtime_api.h>

#include "../../../include/merkurio_hip.h"
#include "io.hpp"
#include "extract_common.hpp"

namespace cli {

static const char *kProgram = "merkurio";
static const char *kVersion = "1.0.0";  // crate version of the reference tree (Cargo.toml:3)

// The handles of a finished command.  The program that is about to end (main.cpp: every output is flushed and closed when
// run_extract / run_tag return, then the process leaves through _exit) does not free device memory, streams and
// communicators one by one first: with the HIP runtime's own exit handlers that teardown was 0.15 s of a 0.6 s run.
bool g_process_is_ending = false;
static void release_matchers(const std::vector<mk_matcher *> &ms) {
    if (g_process_is_ending) return;
    for (mk_matcher *x : ms) mk_matcher_destroy(x);
}

// How the emission order of the job's hit tuples was restored (order_hits.hip): calls per path over all handles.  The
// reference's JSON `meta_information` has a fixed key set (the fixture comparators check it), so this goes to stderr: in
// timing mode always, and as a note whenever the library merge sort (path 3, the slow fallback for batches that defeat both
// binnings) ran at all -- real data should tell whether it ever does.
static void report_order_paths(const std::vector<mk_matcher *> &ms) {
    uint64_t tot[4] = {0, 0, 0, 0};
    for (mk_matcher *m : ms) {
        uint64_t c4[4] = {0, 0, 0, 0};
        if (m && mk_matcher_order_stats(m, c4) == MK_OK)
            for (int k = 0; k < 4; ++k) tot[k] += c4[k];
    }
    if (getenv("MERKURIO_TIMING"))
        fprintf(stderr, "[timing] emission-order calls: %llu without work, %llu on record bins, %llu on whole-key bins, %llu library sort\n",
                (unsigned long long)tot[0], (unsigned long long)tot[1], (unsigned long long)tot[2], (unsigned long long)tot[3]);
    if (tot[3])
        fprintf(stderr, "Note: %llu of %llu emission-order calls fell back to the library merge sort (hit tuples that defeat both binnings)\n",
                (unsigned long long)tot[3], (unsigned long long)(tot[1] + tot[2] + tot[3]));
}

// helpers::parse_pattern_list (src/helpers.rs:76-133) through the library
static Patterns load_patterns(const CommonArgs &a) {
    std::vector<uint8_t> raw;
    std::vector<uint32_t> raw_off{0};
    if (a.kmer_file) {  // the file has priority (src/helpers.rs:85-89)
        if (is_directory(*a.kmer_file))
            bail("Problem parsing pattern list.: Problem reading k-mers from file: \"" + *a.kmer_file +
                 "\": K-mer file path '" + *a.kmer_file + "' is a directory, not a file.");
        FILE *f = fopen(a.kmer_file->c_str(), "rb");
        if (!f) bail("Problem parsing pattern list.: Problem reading k-mers from file: \"" + *a.kmer_file + "\": File not found.");
        std::string content;
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) content.append(buf, n);
        fclose(f);
        uint8_t *b = nullptr;
        uint32_t *o = nullptr, cnt = 0;
        if (mk_read_kmers_from_text((const uint8_t *)content.data(), content.size(), &b, &o, &cnt) != MK_OK)
            bail(std::string("Problem parsing pattern list.: Problem reading k-mers from file: \"") + *a.kmer_file + "\": " +
                 mk_last_error());
        raw.assign(b, b + o[cnt]);
        raw_off.assign(o, o + cnt + 1);
        mk_free(b);
        mk_free(o);
    } else {
        for (auto &s : a.kmer_seq) {
            raw.insert(raw.end(), s.begin(), s.end());
            raw_off.push_back((uint32_t)raw.size());
        }
    }
    uint8_t *b = nullptr;
    uint32_t *o = nullptr, cnt = 0;
    raw.push_back(0);
    if (mk_parse_pattern_list(raw.data(), raw_off.data(), (uint32_t)raw_off.size() - 1, a.reverse_complement, a.canonical,
                              a.lowercase, a.uppercase, &b, &o, &cnt) != MK_OK)
        bail(std::string("Problem parsing pattern list.: ") + mk_last_error());
    Patterns p;
    p.bytes.assign(b, b + o[cnt]);
    p.off.assign(o, o + cnt + 1);
    for (uint32_t i = 0; i < cnt; ++i) p.list.emplace_back((const char *)b + o[i], o[i + 1] - o[i]);
    mk_free(b);
    mk_free(o);
    return p;
}

static void open_loggers(const CommonArgs &a, Loggers &lg) {
    if (a.out_log) {
        lg.text.out.reset(new Sink());
        lg.text.out->open(*a.out_log);
        if (!lg.text.out->f) bail("Problem creating log file: " + *a.out_log);
    }
    if (a.json_log) {
        lg.json.out.reset(new Sink());
        lg.json.out->open(*a.json_log);
        if (!lg.json.out->f) bail("Error creating JSON log file: " + *a.json_log);
        lg.json.begin();
        lg.has_json = true;
    }
    lg.active = a.out_log || a.json_log;
}

// ---- --gpus N: one matcher handle + one host thread per device, contiguous record ranges ---------
// [lo, hi) of shard d of n units over `parts` shards, sizes differing by at most one (the same
// rule as merkurio_amd/sharding.py): concatenating shard outputs in device order reproduces the
// single-device output order (pairs are units, never split: src/cmd_extract.rs:463-468)
static std::pair<size_t, size_t> shard_range(size_t n, size_t parts, size_t d) {
    const size_t base = n / parts, rem = n % parts;
    const size_t lo = d * base + std::min(d, rem);
    return {lo, lo + base + (d < rem ? 1 : 0)};
}

// the scalars of src/cmd_extract.rs:285-290 / src/cmd_tag.rs:360-364 and pattern_hit_counts of every
// device, summed by the library's RCCL all-reduce (mk_reduce_counters): the job's only collective.
// The per-device vectors are host values by the time the job ends (the driver loops keep them per batch), so
// when RCCL cannot be bound or the collective fails the same sum is taken on the host, with a warning -- a
// job whose records and log rows are already written must not end without its summary.
static void reduce_device_counters(const std::vector<mk_matcher *> &ms, const std::vector<int> &devs,
                                   const std::vector<mk_counters> &cs, const std::vector<std::vector<uint32_t>> &counts,
                                   mk_counters &c, std::vector<uint32_t> &total_counts) {
    const size_t n_pat = total_counts.size(), len = n_pat + 8;
    std::vector<std::vector<uint64_t>> vec(ms.size(), std::vector<uint64_t>(len, 0));
    for (size_t d = 0; d < ms.size(); ++d) {
        std::vector<uint64_t> &v = vec[d];
        for (size_t k = 0; k < n_pat; ++k) v[k] = counts[d][k];
        v[n_pat + 0] = cs[d].nb_records_tot; v[n_pat + 1] = cs[d].nb_bases;
        v[n_pat + 2] = cs[d].nb_hits_tot[0]; v[n_pat + 3] = cs[d].nb_hits_tot[1];
        v[n_pat + 4] = cs[d].nb_records_hit[0]; v[n_pat + 5] = cs[d].nb_records_hit[1];
        v[n_pat + 6] = cs[d].nb_records_extracted;
    }
    std::vector<uint64_t> sum(len, 0);
    std::string why;
    bool reduced = false;
    if (mk_comm_available() != MK_OK) {
        why = mk_last_error();
    } else {
        std::vector<void *> dptr(ms.size(), nullptr);
        bool ok = true;
        for (size_t d = 0; d < ms.size() && ok; ++d) {
            ok = hipSetDevice(devs[d]) == hipSuccess && hipMalloc(&dptr[d], len * sizeof(uint64_t)) == hipSuccess &&
                 hipMemcpy(dptr[d], vec[d].data(), len * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) why = "device buffer for the counter vector: " + std::string(hipGetErrorString(hipGetLastError()));
        }
        if (ok) {
            reduced = mk_reduce_counters(ms.data(), (int)ms.size(), dptr.data(), len, sum.data()) == MK_OK;
            if (!reduced) why = mk_last_error();
        }
        for (size_t d = 0; d < ms.size(); ++d)
            if (dptr[d]) {
                (void)hipSetDevice(devs[d]);
                (void)hipFree(dptr[d]);
            }
    }
    if (!reduced) {
        fprintf(stderr, "Warning: per-GPU counters summed on the host (RCCL reduction unavailable: %s)\n", why.c_str());
        std::fill(sum.begin(), sum.end(), 0);
        for (auto &v : vec)
            for (size_t k = 0; k < len; ++k) sum[k] += v[k];
    }
    for (size_t k = 0; k < n_pat; ++k) total_counts[k] = (uint32_t)sum[k];
    c.nb_records_tot = sum[n_pat + 0]; c.nb_bases = sum[n_pat + 1];
    c.nb_hits_tot[0] = sum[n_pat + 2]; c.nb_hits_tot[1] = sum[n_pat + 3];
    c.nb_records_hit[0] = sum[n_pat + 4]; c.nb_records_hit[1] = sum[n_pat + 5];
    c.nb_records_extracted = sum[n_pat + 6];
}

static mk_matcher *make_matcher(const CommonArgs &a, const Patterns &p, bool *use_ac, int device = -1) {
    // src/cmd_extract.rs:166-171: -I forces AC; otherwise auto unless -q / -a were given
    bool ac = a.aho_corasick;
    if (a.case_insensitive)
        ac = true;
    else if (!a.q_size && !a.aho_corasick)
        ac = mk_recommend_aho_corasick(p.list.size(), std::max_element(p.list.begin(), p.list.end(), [](auto &x, auto &y) {
                                                          return x.size() < y.size();
                                                      })->size()) != 0;
    *use_ac = ac;
    mk_matcher *m = nullptr;
    mk_check(mk_matcher_create(p.bytes.data(), p.off.data(), (uint32_t)p.list.size(), ac ? MK_ALGO_AC : MK_ALGO_BNDMQ,
                               a.q_size ? (uint32_t)*a.q_size : 0, a.case_insensitive ? MK_FLAG_ASCII_CASE_INSENSITIVE : 0,
                               device < 0 ? a.device : device, &m),
             "Error");
    return m;
}

// --gpus N: the N handles live on devices (--device + d) mod the number of visible GPUs, so that the
// multi-device path can be rehearsed on a box with fewer GPUs (several handles then share a device)
static std::vector<int> device_list(const CommonArgs &a) {
    // (one GPU: no device query here -- the first HIP call starts the runtime, 0.2 s, and belongs on the matcher thread)
    const int avail = a.gpus > 1 ? std::max(1, mk_device_count()) : 1;
    std::vector<int> devs;
    for (int d = 0; d < std::max(1, a.gpus); ++d) devs.push_back(a.gpus > 1 ? (a.device + d) % avail : a.device);
    return devs;
}

static std::vector<mk_matcher *> make_matchers(const CommonArgs &a, const Patterns &p, const std::vector<int> &devs, bool *use_ac) {
    std::vector<mk_matcher *> ms(devs.size(), nullptr);
    try {
        run_threads(devs.size(), [&](size_t d) {
            bool ac = false;
            ms[d] = make_matcher(a, p, &ac, devs[d]);
            if (d == 0) *use_ac = ac;
        });
    } catch (...) {
        for (mk_matcher *m : ms) mk_matcher_destroy(m);
        throw;
    }
    return ms;
}

static void write_summary(TextLogger &t, const Patterns &p, const std::vector<uint32_t> &counts, const mk_counters &c,
                          bool paired) {
    // src/cmd_extract.rs:616-670 / src/cmd_tag.rs:618-647
    size_t found = 0;
    for (uint32_t x : counts) found += x > 0;
    char buf[128];
    snprintf(buf, sizeof(buf), "#\n#Number of patterns found: %zu/%zu (%.2f %%)\n", found, counts.size(),
             (double)found / (double)counts.size() * 100.0);
    t.header(buf);
    t.header("#Pattern\tCount\n");
    for (size_t i = 0; i < counts.size(); ++i) t.header("#" + p.list[i] + "\t" + std::to_string(counts[i]) + "\n");
    t.header("#\n#Total number of records searched: " + std::to_string(c.nb_records_tot) + "\n");
    t.header("#Total number of characters searched: " + std::to_string(c.nb_bases) + "\n");
    t.header("#Total number of hits: " + std::to_string(c.nb_hits_tot[0] + c.nb_hits_tot[1]) + "\n");
    t.header("#Number of distinct records with a hit: " + std::to_string(c.nb_records_hit[0] + c.nb_records_hit[1]) + "\n");
    if (paired) {
        t.header("#\n#Total number of hits in file 1: " + std::to_string(c.nb_hits_tot[0]) + "\n");
        t.header("#Total number of hits in file 2: " + std::to_string(c.nb_hits_tot[1]) + "\n");
        t.header("#Number of distinct records with a hit in file 1: " + std::to_string(c.nb_records_hit[0]) + "\n");
        t.header("#Number of distinct records with a hit in file 2: " + std::to_string(c.nb_records_hit[1]) + "\n");
        t.header("#Total number of extracted records: " + std::to_string(c.nb_records_extracted) + "\n");
    }
    t.flush();
}

static Json command_line_json(const std::vector<std::string> &argv) {
    Json a = Json::array();
    for (auto &s : argv) a.push(Json::string(s));
    return a;
}
static std::string join(const std::vector<std::string> &v) {
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) s += (i ? " " : "") + v[i];
    return s;
}

static void write_log_header(TextLogger &t, const char *title, const std::vector<std::string> &argv, const std::string *tag,
                             size_t n_pat, bool invert) {
    t.header(std::string("#SeqKatcher ") + title + " log\n");  // literal of src/cmd_extract.rs:232
    t.header("#" + timestamp_now() + "\n");
    t.header(std::string("#Running ") + kProgram + " version " + kVersion + "\n");
    t.header("#Command line: " + join(argv) + "\n");
    if (tag) t.header("#Tag used for labeling records: " + *tag + "\n");
    t.header("#Searching for " + std::to_string(n_pat) + " pattern" + (n_pat > 1 ? "s" : "") + " " +
             (invert ? "(inverted matching)" : "") + "\n");
    t.header("#\n#File\tRecord\tPattern\tPosition (zero-based)\n");
    t.flush();
}

// ---- extract ------------------------------------------------------------------------------------
int run_extract(const ExtractArgs &a, const std::vector<std::string> &argv) {
    const std::string conflict = check_log_flag_conflict(a.out_log ? &*a.out_log : nullptr, a.json_log ? &*a.json_log : nullptr,
                                                         a.out_fastx ? &*a.out_fastx : nullptr, a.suppress_output);
    if (!conflict.empty()) bail(conflict);
    Patterns pats = load_patterns(a);
    if (is_directory(a.in_fastx)) bail("Record file path '" + a.in_fastx + "' is a directory, not a file.");
    if (a.in_fastq_2 && is_directory(*a.in_fastq_2)) bail("Second read file path '" + *a.in_fastq_2 + "' is a directory, not a file.");
    const std::string name1 = file_name(a.in_fastx), name2 = a.in_fastq_2 ? file_name(*a.in_fastq_2) : "";
    Loggers lg;
    open_loggers(a, lg);
    if (lg.active) write_log_header(lg.text, "extract", argv, nullptr, pats.list.size(), a.invert_match);
    PhaseTimer tm;
    bool use_ac = false;
    // HIP initialisation + pattern-set compilation (0.1-0.3 s) runs beside the input parsing
    const std::vector<int> devs = device_list(a);
    std::future<std::vector<mk_matcher *>> fm = std::async(std::launch::async, [&] { return make_matchers(a, pats, devs, &use_ac); });

    // Default: everything `extract` reads goes to the device as windows of raw text (extract_windows.cpp: FASTQ or FASTA, one file or
    // a pair, plain / gzip / bgzip'ed, one GPU or several).  --host-ingest, and inputs that path does not take, are parsed here.
    WindowExtract wx;
    bool by_windows = false;
    // The host reader: the inputs are read a window at a time (--window-mb of text, decompressed if need be): the host holds
    // one window and its record index, like the reference, which streams records.
    FastxStream s1, s2;
    FastxFile &f1 = s1.view, &f2 = s2.view;
    const bool paired = (bool)a.in_fastq_2;
    const uint64_t window_bytes = (uint64_t)a.window_mb << 20;
    bool more1 = false, more2 = false;
    try {
        by_windows = !a.host_ingest && wx.prepare(a, devs);
        if (!by_windows) {
            s1.open(a.in_fastx);
            if (paired) s2.open(*a.in_fastq_2);
            more1 = s1.fill(window_bytes);
            if (paired) more2 = s2.fill(window_bytes);
        }
    } catch (...) {
        fm.get();  // a matcher error comes first, as in the serial order of the reference
        throw;
    }
    tm.mark("open + first window");
    const std::vector<mk_matcher *> ms = fm.get();
    mk_matcher *m = ms[0];
    // --gpus N: RCCL's communicators for the final counter reduction are set up beside the job (seconds, against a job of one)
    std::future<void> comm_ready;
    if (ms.size() > 1) comm_ready = std::async(std::launch::async, [&ms] { (void)mk_reduce_prepare(ms.data(), (int)ms.size()); });
    // from here on (the HIP runtime is up) the members of a bgzip'ed input are inflated by the device codec (mk_bgzf_inflate)
    if (!a.host_codec) set_bgzf_device(devs[0], a.device_codec_always);
    tm.mark("matcher create (HIP init), remainder");
    // writers: src/cmd_extract.rs:297-318, :420-460
    Sink w1, w2;
    if (a.out_fastx) {
        std::string p = with_extension(*a.out_fastx, identify_uncompressed_type(a.in_fastx));
        if (paired) {
            w1.open(add_suffix_to_file_prefix(p, "_1"));
            w2.open(add_suffix_to_file_prefix(p, "_2"));
            if (!w1.f || !w2.f) bail("Error writing to paired-end file; no such directory: \"" + p + "\"");
        } else {
            w1.open(p);
            if (!w1.f) bail("Error writing to output file; no such directory: \"" + p + "\"");
        }
    } else {
        w1.open("STDOUT");
        if (paired) w2.open("STDOUT");
    }

    mk_counters c;
    memset(&c, 0, sizeof(c));
    std::vector<uint32_t> counts(pats.list.size(), 0);
    const uint64_t batch_bytes = (uint64_t)a.batch_mb << 20;
    // what a batch's results turn into: log rows (reference emission order) and the kept records
    auto emit_rows = [&](size_t b0, const mk_row *rows, uint64_t n_rows) {
        emit_log_rows(
            lg, pats, rows, n_rows,
            [&](const mk_row &r) {
                const FastxFile &ff = r.file ? f2 : f1;
                const auto &rec = ff.recs[b0 + r.rec];
                return std::pair<const char *, size_t>(ff.data + rec.id_b, rec.id_e - rec.id_b);
            },
            [&](const mk_row &r) -> const std::string & { return r.file ? name2 : name1; });
    };
    auto emit_records = [&](size_t b0, const uint8_t *keep, uint64_t nb) {
        if (a.suppress_output) return;
        for (uint64_t k = 0; k < nb; ++k)
            if (keep[k]) {
                f1.write(b0 + k, w1);
                if (paired) f2.write(b0 + k, w2);
            }
    };
    // Scans records [r0, r1) on matcher `mm` in double-buffered batches: while batch k is on the GPU
    // and its results are consumed, a second thread gathers the sequences of batch k + 1.
    // on_batch(first record, #records, keep, rows, #rows) is called in record order.
    // batch buffers of one device thread; they live as long as the job (a fresh 128 MB buffer costs its page
    // faults again: 30 ms per batch instead of 4)
    struct Batch {
        size_t b0 = 0, b1 = 0;
        std::vector<uint8_t> s1, s2;
        std::vector<uint64_t> o1, o2;
    };
    struct DevBuffers {
        Batch bufs[2];
        std::vector<uint8_t> keep;
        std::vector<mk_row> rows = std::vector<mk_row>(4096);
    };
    std::vector<DevBuffers> dev_bufs(ms.size());
    double call_ms[4] = {0, 0, 0, 0};  // MERKURIO_TIMING: phases inside mk_extract_single / _paired, summed (one device)
    auto scan_range = [&](mk_matcher *mm, DevBuffers &DB, size_t r0, size_t r1, mk_counters &cc, std::vector<uint32_t> &cnts, auto on_batch) {
        Batch(&bufs)[2] = DB.bufs;
        std::vector<uint8_t> &keep = DB.keep;
        std::vector<mk_row> &rows = DB.rows;
        auto fill = [&](Batch &b, size_t from) {
            size_t i = from;
            uint64_t bytes = 0;
            while (i < r1 && (bytes < batch_bytes || i == from)) {
                bytes += f1.raw_len(i) + (paired ? f2.raw_len(i) : 0);
                ++i;
            }
            b.b0 = from;
            b.b1 = i;
            f1.gather(from, i, b.s1, b.o1);
            if (paired) f2.gather(from, i, b.s2, b.o2);
        };
        int cur = 0;
        bufs[0].b0 = r1;
        if (r0 < r1) fill(bufs[0], r0);
        while (bufs[cur].b0 < r1) {
            Batch &B = bufs[cur];
            const size_t b0 = B.b0, i = B.b1;
            std::future<void> next;
            Batch &N = bufs[cur ^ 1];
            N.b0 = r1;  // "no further batch" unless filled below
            if (i < r1) next = std::async(std::launch::async, [&, i] { fill(N, i); });
            const uint64_t nb = i - b0;
            keep.assign(nb, 0);
            uint64_t n_rows = 0;
            try {
                for (;;) {
                    mk_counters cb;
                    memset(&cb, 0, sizeof(cb));
                    std::vector<uint32_t> cnt_b(cnts.size(), 0);
                    int rc = paired ? mk_extract_paired(mm, B.s1.data(), B.o1.data(), nb, B.s2.data(), B.o2.data(), nb, lg.active,
                                                        a.invert_match, keep.data(), rows.data(), rows.size(), &n_rows, &cb, cnt_b.data())
                                    : mk_extract_single(mm, B.s1.data(), B.o1.data(), nb, lg.active, a.invert_match, keep.data(),
                                                        rows.data(), rows.size(), &n_rows, &cb, cnt_b.data());
                    if (rc == MK_E_CAPACITY && n_rows > rows.size()) {
                        rows.resize(n_rows);
                        continue;
                    }
                    mk_check(rc, "Error during matching");
                    if (tm.on && ms.size() == 1) {  // where the call itself spent its time: upload / device / download / host loop
                        float ms4[4] = {0, 0, 0, 0};
                        if (mk_matcher_batch_times(mm, ms4) == MK_OK)
                            for (int k = 0; k < 4; ++k) call_ms[k] += ms4[k];
                    }
                    cc.nb_records_tot += cb.nb_records_tot; cc.nb_bases += cb.nb_bases;
                    cc.nb_hits_tot[0] += cb.nb_hits_tot[0]; cc.nb_hits_tot[1] += cb.nb_hits_tot[1];
                    cc.nb_records_hit[0] += cb.nb_records_hit[0]; cc.nb_records_hit[1] += cb.nb_records_hit[1];
                    cc.nb_records_extracted += cb.nb_records_extracted;
                    for (size_t k = 0; k < cnts.size(); ++k) cnts[k] += cnt_b[k];
                    break;
                }
                on_batch(b0, nb, keep.data(), rows.data(), lg.active ? n_rows : 0);
            } catch (...) {
                if (next.valid()) next.wait();  // the gather thread still writes into this frame
                throw;
            }
            if (next.valid()) next.get();
            cur ^= 1;
        }
    };
    // per-device counters of a --gpus N job (summed once, at the end, by RCCL)
    std::vector<mk_counters> dev_c(ms.size());
    std::vector<std::vector<uint32_t>> dev_counts(ms.size(), std::vector<uint32_t>(counts.size(), 0));
    for (auto &x : dev_c) memset(&x, 0, sizeof(x));
    if (by_windows) {
        wx.run(a, pats, lg, ms, devs, w1, w2, name1, name2, dev_c, dev_counts, tm);
        if (ms.size() == 1) {
            c = dev_c[0];
            counts = dev_counts[0];
        }
    }
    while (more1 || more2) {
        if (paired && !more2)  // src/cmd_extract.rs:465-468: file 2 ends first
            bail("Error during FASTQ record parsing of second file. Do the two input files contain the same number of records?");
        if (paired && !more1)  // src/cmd_extract.rs:608-612: file 2 still has records
            bail("The two input files have a different number of records. Please provide valid paired-end read files.");
        // pairs are matched by ordinal: both windows advance by the same number of records
        const size_t n = paired ? std::min(f1.recs.size(), f2.recs.size()) : f1.recs.size();
        // the next window is inflated and indexed while this one is scanned
        s1.consume(n);
        if (paired) s2.consume(n);
        std::future<void> next_window = std::async(std::launch::async, [&] {
            s1.prefetch(window_bytes);
            if (paired) s2.prefetch(window_bytes);
        });
        try {
        if (ms.size() == 1) {
            scan_range(m, dev_bufs[0], 0, n, c, counts, [&](size_t b0, uint64_t nb, const uint8_t *keep, const mk_row *rows, uint64_t n_rows) {
                tm.mark("batch: gather + H2D + scan + D2H");
                emit_rows(b0, rows, n_rows);
                emit_records(b0, keep, nb);
                tm.mark("batch: rows + records out");
            });
        } else {
            // --gpus N: device d scans the contiguous record (pair) range shard_range(n, N, d) of the window on
            // its own host thread; results are buffered per device and emitted in device order = record order
            struct Shard {
                size_t r0 = 0, r1 = 0;
                std::vector<uint8_t> keep;
                std::vector<mk_row> rows;  // rec = index inside the shard
            };
            std::vector<Shard> shards(ms.size());
            for (size_t d = 0; d < ms.size(); ++d) {
                auto [lo, hi] = shard_range(n, ms.size(), d);
                shards[d].r0 = lo;
                shards[d].r1 = hi;
                shards[d].keep.assign(hi - lo, 0);
            }
            run_threads(ms.size(), [&](size_t d) {
                Shard &S = shards[d];
                scan_range(ms[d], dev_bufs[d], S.r0, S.r1, dev_c[d], dev_counts[d],
                           [&](size_t b0, uint64_t nb, const uint8_t *keep, const mk_row *rows, uint64_t n_rows) {
                               memcpy(S.keep.data() + (b0 - S.r0), keep, nb);
                               for (uint64_t k = 0; k < n_rows; ++k) {
                                   mk_row r = rows[k];
                                   r.rec += b0 - S.r0;
                                   S.rows.push_back(r);
                               }
                           });
            });
            tm.mark("window: scan on all devices");
            for (auto &S : shards) {
                emit_rows(S.r0, S.rows.data(), S.rows.size());
                emit_records(S.r0, S.keep.data(), S.r1 - S.r0);
            }
        }
        } catch (...) {
            next_window.wait();  // it works on s1 / s2
            throw;
        }
        next_window.get();  // a malformed record in the next window is reported now, after this one was written
        more1 = s1.fill(window_bytes);
        if (paired) more2 = s2.fill(window_bytes);
        tm.mark("next window");
    }
    if (ms.size() > 1) {
        tm.mark("records done");
        comm_ready.get();
        reduce_device_counters(ms, devs, dev_c, dev_counts, c, counts);
        tm.mark("counter reduction (RCCL set-up began with the job)");
    }
    w1.flush();
    w2.flush();
    tm.mark("log rows + write records");
    if (tm.on && (call_ms[0] + call_ms[1] + call_ms[2] + call_ms[3]) > 0)
        fprintf(stderr, "[timing] inside the batch calls: upload %.3f s, device %.3f s, download %.3f s, host loop %.3f s\n", call_ms[0] * 1e-3,
                call_ms[1] * 1e-3, call_ms[2] * 1e-3, call_ms[3] * 1e-3);
    report_order_paths(ms);
    if (lg.active) {
        lg.text.flush();
        write_summary(lg.text, pats, counts, c, paired);
    }
    if (lg.has_json) {  // src/cmd_extract.rs:673-714
        Json files = Json::object();
        files.set("kmer_file", a.kmer_file ? Json::string(*a.kmer_file) : Json::null());
        files.set("record_file_1", Json::string(name1));
        files.set("record_file_2", paired ? Json::string(name2) : Json::null());
        Json meta = Json::object();
        meta.set("program", Json::string(kProgram)).set("version", Json::string(kVersion));
        meta.set("timestamp", Json::string(timestamp_now())).set("subcommand", Json::string("extract"));
        meta.set("command_line", command_line_json(argv));
        meta.set("search_algorithm", Json::string(use_ac ? "Aho-Corasick" : "BNDMq"));
        meta.set("inverted_matching", Json::boolean(a.invert_match)).set("case_insensitive", Json::boolean(a.case_insensitive));
        meta.set("input_files", files);
        Json cj = Json::object();
        size_t found = 0;
        for (size_t k = 0; k < counts.size(); ++k) {
            cj.set(pats.list[k], Json::integer(counts[k]));
            found += counts[k] > 0;
        }
        Json sum = Json::object();
        sum.set("number_of_patterns_searched", Json::integer((long long)pats.list.size()));
        sum.set("number_of_patterns_found", Json::integer((long long)found));
        sum.set("number_of_records_searched", Json::integer((long long)c.nb_records_tot));
        sum.set("number_of_characters_searched", Json::integer((long long)c.nb_bases));
        sum.set("number_of_matches", Json::integer((long long)(c.nb_hits_tot[0] + c.nb_hits_tot[1])));
        sum.set("number_of_distinct_records_with_a_hit", Json::integer((long long)(c.nb_records_hit[0] + c.nb_records_hit[1])));
        Json pe = Json::object();
        pe.set("searching_paired_end_reads", Json::boolean(paired));
        pe.set("number_of_hits_in_file_1", Json::integer((long long)c.nb_hits_tot[0]));
        pe.set("number_of_hits_in_file_2", paired ? Json::integer((long long)c.nb_hits_tot[1]) : Json::null());
        pe.set("number_of_distinct_records_with_a_hit_in_file_1", Json::integer((long long)c.nb_records_hit[0]));
        pe.set("number_of_distinct_records_with_a_hit_in_file_2", paired ? Json::integer((long long)c.nb_records_hit[1]) : Json::null());
        pe.set("number_of_extracted_records", Json::integer((long long)c.nb_records_extracted));
        lg.json.finalize(meta, cj, sum, &pe);
    }
    release_matchers(ms);
    return 0;
}

// ---- tag ----------------------------------------------------------------------------------------
int run_tag(const TagArgs &a, const std::vector<std::string> &argv) {
    const std::string conflict = check_log_flag_conflict(a.out_log ? &*a.out_log : nullptr, a.json_log ? &*a.json_log : nullptr,
                                                         a.out_file ? &*a.out_file : nullptr, a.suppress_output);
    if (!conflict.empty()) bail(conflict);
    if (is_directory(a.in_file)) bail("Record file path '" + a.in_file + "' is a directory, not a file.");
    const std::string in_name = file_name(a.in_file);
    Patterns pats = load_patterns(a);
    if (a.threads < 1) bail("Number of threads must be at least 1.");
    if (a.threads_given) set_io_threads_cap((unsigned)a.threads);
    if (a.tag.size() != 2) bail("Tag must be exactly two characters long.");
    Loggers lg;
    open_loggers(a, lg);
    // extension logic: src/cmd_tag.rs:293-308
    const std::string in_ext = extension(a.in_file);
    if (in_ext.empty()) bail("Could not detect the file extension: \"" + a.in_file + "\"");
    std::string out_ext = "STDOUT";
    if (a.out_file) {
        out_ext = extension(*a.out_file);
        if (out_ext.empty()) out_ext = in_ext;
    }
    if (lg.active) write_log_header(lg.text, "tag", argv, &a.tag, pats.list.size(), a.invert_match);
    bool use_ac = false;
    PhaseTimer tm;
    const std::vector<int> devs = device_list(a);
    std::future<std::vector<mk_matcher *>> fm = std::async(std::launch::async, [&] { return make_matchers(a, pats, devs, &use_ac); });

    SamFile sam;
    try {
        if (a.device_codec_always && !a.host_codec) {  // (the header's members too: the HIP runtime has to be up first)
            fm.wait();
            set_bgzf_device(devs[0], true);
        }
        sam.open(a.in_file);  // "Input file must be a BAM or SAM file." for other extensions; header only
    } catch (...) {
        fm.get();  // a matcher error comes first, as in the serial order of the reference
        throw;
    }
    tm.mark("open + header");
    const std::vector<mk_matcher *> ms = fm.get();
    mk_matcher *m = ms[0];
    std::future<void> comm_ready;  // (--gpus N: RCCL's set-up beside the job, as in extract)
    if (ms.size() > 1) comm_ready = std::async(std::launch::async, [&ms] { (void)mk_reduce_prepare(ms.data(), (int)ms.size()); });
    // from here on (the HIP runtime is up) the BGZF members of a BAM window are inflated by the device codec (mk_bgzf_inflate)
    if (!a.host_codec) set_bgzf_device(devs[0], a.device_codec_always);
    tm.mark("matcher (HIP init), remainder");
    if (out_ext != "sam" && out_ext != "bam" && out_ext != "STDOUT") bail("Output file must be a BAM or SAM file.");
    Sink w;
    BamWriter bw;
    const bool to_bam = out_ext == "bam" && !a.suppress_output;
    // header + @PG line (src/cmd_tag.rs:509-514)
    const std::string out_header =
        sam.header + "@PG\tID:" + kProgram + "\tPN:" + kProgram + "\tCL:" + join(argv) + "\tVN:" + kVersion + "\n";
    if (to_bam && !a.host_codec) bw.use_device(devs[0]);  // members deflated by the device codec (mk_bgzf_deflate)
    if (to_bam) {  // BAM -> BAM passes raw records through: they keep the input's reference ids
        if (sam.is_bam)
            bw.open(with_extension(*a.out_file, out_ext), out_header, &sam.ref_names, &sam.ref_lens);
        else
            bw.open(with_extension(*a.out_file, out_ext), out_header);
    } else if (!a.suppress_output) {
        w.open(out_ext == "STDOUT" ? "STDOUT" : with_extension(*a.out_file, out_ext));
        if (!w.f) bail("Error writing SAM file: " + with_extension(*a.out_file, out_ext));
        w.write(out_header);
    }

    mk_counters c;
    memset(&c, 0, sizeof(c));
    std::vector<uint32_t> counts(pats.list.size(), 0);
    const uint64_t batch_bytes = (uint64_t)a.batch_mb << 20;
    const uint64_t window_bytes = (uint64_t)a.window_mb << 20;
    // The input is read a window at a time (--window-mb of SAM text / inflated BAM); inside a window,
    // one batch = a slab of records whose sequences fill --batch-mb: gather (upper-case / un-nibble)
    // -> mk_tag_records -> log rows -> tag + encode the kept records.  Only the input buffer and its
    // record index are whole-file; device buffers, hit rows and matched-pattern sets are per batch.
    // The results of a batch: its log rows and the encoded output of its kept records, in record order.
    struct BatchOut {
        std::vector<mk_row> rows;  // rec = global record index
        std::vector<std::vector<uint8_t>> bin;  // BAM output: per-thread encoded records, in order
        std::vector<std::string> txt;           // SAM output
    };
    std::mutex txt_pool_mu;
    std::vector<std::string> txt_pool;
    auto emit = [&](BatchOut &o) {  // (takes the batch's encoded records with it)
        if (lg.active)
            emit_log_rows(
                lg, pats, o.rows.data(), o.rows.size(),
                [&](const mk_row &r) {
                    const auto &rec = sam.recs[r.rec];
                    return std::pair<const char *, size_t>(sam.data + rec.off + (sam.is_bam ? 36 : 0), rec.name_len);
                },
                [&](const mk_row &) -> const std::string & { return in_name; });
        for (auto &b : o.bin) bw.put_encoded(std::move(b));  // moved, not copied: the pieces are joined on the device
        for (auto &t : o.txt) w.write(t);
        {  // the text buffers go back to the encoder threads (pages mapped: a fresh one costs a fault per 4 KiB)
            std::lock_guard<std::mutex> lk(txt_pool_mu);
            for (auto &t : o.txt)
                if (t.capacity() >= (1u << 20) && txt_pool.size() < 64) {
                    t.clear();
                    txt_pool.push_back(std::move(t));
                }
        }
    };
    // scan buffers of one device thread, reused by every batch of every window
    struct TagBuffers {
        std::vector<uint8_t> seq, keep;
        std::vector<uint64_t> off, foff;
        std::vector<uint32_t> fpat = std::vector<uint32_t>(1024);
        std::vector<mk_row> rows = std::vector<mk_row>(4096);
    };
    std::vector<TagBuffers> dev_bufs(ms.size());
    auto scan_range = [&](mk_matcher *mm, TagBuffers &TB, size_t r0, size_t r1, mk_counters &cc, std::vector<uint32_t> &cnts,
                          size_t enc_threads, auto on_batch) {
        std::vector<uint8_t> &seq = TB.seq, &keep = TB.keep;
        std::vector<uint64_t> &off = TB.off, &foff = TB.foff;
        std::vector<uint32_t> &fpat = TB.fpat;
        std::vector<mk_row> &rows = TB.rows;
        for (size_t b0 = r0; b0 < r1;) {
            size_t b1 = b0;
            uint64_t bytes = 0;
            while (b1 < r1 && (bytes < batch_bytes || b1 == b0)) bytes += sam.recs[b1++].l_seq;
            const size_t nb = b1 - b0;
            sam.gather(b0, b1, seq, off);
            if (ms.size() == 1) tm.mark("  batch: gather");
            keep.assign(nb, 0);
            foff.assign(nb + 1, 0);
            uint64_t n_rows = 0;
            for (;;) {
                mk_counters cb;
                memset(&cb, 0, sizeof(cb));
                std::vector<uint32_t> cnt_b(cnts.size(), 0);
                int rc = mk_tag_records(mm, seq.data(), off.data(), nb, lg.active, a.filter_matching, a.invert_match, keep.data(),
                                        rows.data(), rows.size(), &n_rows, &cb, cnt_b.data(), foff.data(), fpat.data(), fpat.size());
                if (rc == MK_E_CAPACITY && (n_rows > rows.size() || foff[nb] > fpat.size())) {
                    rows.resize(std::max<uint64_t>(rows.size(), n_rows));
                    fpat.resize(std::max<uint64_t>(fpat.size(), foff[nb]));
                    continue;
                }
                mk_check(rc, "Error during matching");
                cc.nb_records_tot += cb.nb_records_tot; cc.nb_bases += cb.nb_bases;
                cc.nb_hits_tot[0] += cb.nb_hits_tot[0]; cc.nb_records_hit[0] += cb.nb_records_hit[0];
                cc.nb_records_extracted += cb.nb_records_extracted;
                for (size_t k = 0; k < cnts.size(); ++k) cnts[k] += cnt_b[k];
                break;
            }
            if (ms.size() == 1) tm.mark("  batch: mk_tag_records");
            BatchOut out;
            if (lg.active) {
                out.rows.assign(rows.begin(), rows.begin() + n_rows);
                for (auto &r : out.rows) r.rec += b0;
            }
            // tag + encode the kept records (src/cmd_tag.rs:457-497) on the host threads, in record order
            std::vector<size_t> kept;
            for (size_t k = 0; k < nb; ++k)
                if (keep[k]) kept.push_back(k);
            if (a.suppress_output) {  // the reference still validates existing tags of kept records
                for (size_t k : kept) {
                    std::string existing;
                    if (sam.find_tag(b0 + k, a.tag, &existing) == 2) bail("Invalid tag value format. Expected string value.");
                }
            } else {
                const size_t T = std::max<size_t>(1, std::min<size_t>(enc_threads, kept.size() / 4096 + 1));
                out.bin.resize(to_bam ? T : 0);
                out.txt.resize(to_bam ? 0 : T);
                run_threads(T, [&](size_t t) {
                    std::vector<char> val(4096);
                    std::string line;
                    {  // one allocation for the slice's output instead of a doubling series of copies
                        size_t est = 0;
                        for (size_t i = kept.size() * t / T; i < kept.size() * (t + 1) / T; ++i) est += sam.recs[b0 + kept[i]].len + 24 + a.tag.size();
                        if (to_bam) {
                            out.bin[t] = bw.take_buffer();  // (one the writer thread has written out, if there is one)
                            out.bin[t].reserve(est);
                        } else {
                            {
                                std::lock_guard<std::mutex> lk(txt_pool_mu);
                                if (!txt_pool.empty()) {
                                    out.txt[t] = std::move(txt_pool.back());
                                    txt_pool.pop_back();
                                }
                            }
                            out.txt[t].reserve(est);
                        }
                    }
                    for (size_t i = kept.size() * t / T; i < kept.size() * (t + 1) / T; ++i) {
                        const size_t k = kept[i], g = b0 + k;
                        std::string existing;
                        const int has = sam.find_tag(g, a.tag, &existing);
                        if (has == 2) bail("Invalid tag value format. Expected string value.");
                        size_t need = 0;
                        for (;;) {
                            int rc = mk_tag_value(mm, fpat.data() + foff[k], foff[k + 1] - foff[k], has == 1 ? existing.c_str() : nullptr,
                                                  val.data(), val.size(), &need);
                            if (rc == MK_E_CAPACITY) {
                                val.resize(need + 1);
                                continue;
                            }
                            mk_check(rc, "Error building tag value");
                            break;
                        }
                        if (to_bam && sam.is_bam) {
                            BamWriter::append_tagged_raw(sam.raw(g), sam.raw_len(g), a.tag, val.data(), need, out.bin[t]);
                        } else if (to_bam) {
                            line.clear();
                            sam.append_line(g, line);
                            line += '\t';
                            line += a.tag;
                            line += ":Z:";
                            line.append(val.data(), need);
                            bw.encode_record(line, out.bin[t]);
                        } else {
                            std::string &o = out.txt[t];
                            sam.append_line(g, o);
                            o += '\t';
                            o += a.tag;
                            o += ":Z:";
                            o.append(val.data(), need);
                            o += '\n';
                        }
                    }
                });
            }
            if (ms.size() == 1) tm.mark("  batch: tag values + encode");
            on_batch(std::move(out));
            b0 = b1;
        }
    };
    // per-device counters of a --gpus N job (summed once, at the end, by RCCL)
    std::vector<mk_counters> dev_c(ms.size());
    std::vector<std::vector<uint32_t>> dev_counts(ms.size(), std::vector<uint32_t>(counts.size(), 0));
    for (auto &x : dev_c) memset(&x, 0, sizeof(x));
    bool device_done = false;
    // BAM -> BAM (or no output at all) on one device: the records stay on the device between inflate and deflate
    // (tag_windows.cpp; two windows in flight, each on a handle of its own); false: a window was not for the device and the loop
    // below takes the input from there
    std::vector<mk_matcher *> seconds;
    if (sam.bam_on_bgzf() && (to_bam || a.suppress_output) && !a.host_codec && !a.host_ingest) {
        if (to_bam) bw.use_device(devs[0]);
        // (240 MiB of text: the tagged records of a window then fill one round of the deflate kernel's 4 096 resident waves, not one and a bit)
        // (an explicit --window-mb is honoured up to 2 GiB: a window's text, head included, has to stay below 4 GiB on the device)
        const uint64_t dev_window = a.window_mb_given ? std::min<uint64_t>(window_bytes, 2048ull << 20) : (240ull << 20);
        seconds.assign(ms.size(), nullptr);
        run_threads(ms.size(), [&](size_t d) {
            bool ac2 = false;
            seconds[d] = make_matcher(a, pats, &ac2, devs[d]);
        });
        // window k runs on handle k mod 2N: devices in turn, and two windows per device in flight.  One device: the job's counters;
        // several: the per-device vectors that RCCL sums at the end
        std::vector<TagHandle> handles;
        for (int rep = 0; rep < 2; ++rep)
            for (size_t d = 0; d < ms.size(); ++d)
                handles.push_back(TagHandle{rep ? seconds[d] : ms[d], devs[d], ms.size() == 1 ? &c : &dev_c[d], ms.size() == 1 ? &counts : &dev_counts[d]});
        device_done = tag_bam_windows_on_device(a, sam, handles, lg, pats, in_name, to_bam ? &bw : nullptr, dev_window);
        tm.mark(device_done ? "windows on the device" : "windows on the device (the rest: host reader)");
    }
    // (the first window is small: nothing can run beside its read; the later, large ones are read beside their predecessors)
    bool more_windows = !device_done && sam.fill(std::min<uint64_t>(window_bytes, 128ull << 20));
    while (more_windows) {
        const size_t n = sam.recs.size();
        tm.mark("window: read (inflate) + index");
        // the next window of a compressed input is inflated (device codec: the host threads are free for the batches
        // below) and indexed beside this one; an error in it is reported after this window was written
        std::future<void> next_window = std::async(std::launch::async, [&] { sam.prefetch(window_bytes); });
        try {
        if (ms.size() == 1) {
            scan_range(m, dev_bufs[0], 0, n, c, counts, io_threads(), [&](BatchOut &&o) {
                emit(o);
                tm.mark("  batch: rows + write");
            });
        } else {
            // --gpus N: contiguous record ranges of the window per device, one host thread each; batch results
            // are kept per device and emitted in device order = record order
            std::vector<std::vector<BatchOut>> outs(ms.size());
            run_threads(ms.size(), [&](size_t d) {
                auto [lo, hi] = shard_range(n, ms.size(), d);
                scan_range(ms[d], dev_bufs[d], lo, hi, dev_c[d], dev_counts[d], std::max<size_t>(1, io_threads() / ms.size()),
                           [&](BatchOut &&o) { outs[d].push_back(std::move(o)); });
            });
            tm.mark("window: scan + tag on all devices");
            for (auto &v : outs)
                for (auto &o : v) emit(o);
        }
        } catch (...) {
            next_window.wait();  // it works on sam
            throw;
        }
        tm.mark("window done");
        next_window.get();
        more_windows = sam.fill(window_bytes);
    }
    if (ms.size() > 1) {
        comm_ready.get();
        reduce_device_counters(ms, devs, dev_c, dev_counts, c, counts);
        tm.mark("counter reduction");
    }
    w.flush();
    bw.close();
    tm.mark("write");
    if (to_bam && getenv("MERKURIO_TIMING"))
        fprintf(stderr, "[timing] BGZF deflate on its own threads (%s): %.3f s\n", a.host_codec ? "zlib, host" : "device codec", bw.deflate_seconds);
    if (getenv("MERKURIO_TIMING") && bgzf_device_seconds() > 0)
        fprintf(stderr, "[timing] BGZF inflate calls of the device codec (inside the window reads): %.3f s\n", bgzf_device_seconds());
    report_order_paths(ms);
    if (lg.active) {
        lg.text.flush();
        write_summary(lg.text, pats, counts, c, false);
    }
    if (lg.has_json) {  // src/cmd_tag.rs:650-686
        Json files = Json::object();
        files.set("kmer_file", a.kmer_file ? Json::string(*a.kmer_file) : Json::null());
        files.set("record_file_1", Json::string(in_name));
        Json meta = Json::object();
        meta.set("program", Json::string(kProgram)).set("version", Json::string(kVersion));
        meta.set("timestamp", Json::string(timestamp_now())).set("subcommand", Json::string("tag"));
        meta.set("command_line", command_line_json(argv));
        meta.set("search_algorithm", Json::string(use_ac ? "Aho-Corasick" : "BNDMq"));
        meta.set("inverted_matching", Json::boolean(a.invert_match)).set("case_insensitive", Json::boolean(a.case_insensitive));
        meta.set("input_files", files).set("tag", Json::string(a.tag));
        Json cj = Json::object();
        size_t found = 0;
        for (size_t k = 0; k < counts.size(); ++k) {
            cj.set(pats.list[k], Json::integer(counts[k]));
            found += counts[k] > 0;
        }
        Json sum = Json::object();
        sum.set("number_of_patterns_searched", Json::integer((long long)pats.list.size()));
        sum.set("number_of_patterns_found", Json::integer((long long)found));
        sum.set("number_of_records_searched", Json::integer((long long)c.nb_records_tot));
        sum.set("number_of_characters_searched", Json::integer((long long)c.nb_bases));
        sum.set("number_of_matches", Json::integer((long long)c.nb_hits_tot[0]));
        sum.set("number_of_distinct_records_with_a_hit", Json::integer((long long)c.nb_records_hit[0]));
        lg.json.finalize(meta, cj, sum, nullptr);
    }
    release_matchers(ms);
    release_matchers(seconds);
    return 0;
}

}  // namespace cli
