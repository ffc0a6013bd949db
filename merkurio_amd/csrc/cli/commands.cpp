// commands.cpp -- the `extract` and `tag` drivers of the C++ host program: everything around
// the hot path (argument semantics, record I/O, log / JSON emission, summaries), restated from
// src/cmd_extract.rs:143-717 and src/cmd_tag.rs:155-689.  All matching goes through the C ABI
// (mk_extract_single / mk_extract_paired / mk_tag_records): no text is searched on the host.
#include "commands.hpp"

#include <future>

#include <algorithm>
#include <cstring>
#include <ctime>

#include "../../../include/merkurio_hip.h"
#include "io.hpp"

namespace cli {

static const char *kProgram = "merkurio";
static const char *kVersion = "1.0.0";  // crate version of the reference tree (Cargo.toml:3)

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

static void mk_check(int rc, const char *what) {
    if (rc != MK_OK) bail(std::string(what) + ": " + mk_last_error());
}

struct Patterns {
    std::vector<std::string> list;
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off;
};

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

struct Loggers {
    TextLogger text;
    JsonLogger json;
    bool active = false;
    bool has_json = false;
};

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

static mk_matcher *make_matcher(const CommonArgs &a, const Patterns &p, bool *use_ac) {
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
                               a.device, &m),
             "Error");
    return m;
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
    std::future<mk_matcher *> fm = std::async(std::launch::async, [&] { return make_matcher(a, pats, &use_ac); });

    FastxFile f1, f2;
    const bool paired = (bool)a.in_fastq_2;
    try {
        f1.parse(a.in_fastx);
        if (paired) {
            f2.parse(*a.in_fastq_2);
            if (f2.recs.size() < f1.recs.size())
                bail("Error during FASTQ record parsing of second file. Do the two input files contain the same number of records?");
            if (f2.recs.size() > f1.recs.size())
                bail("The two input files have a different number of records. Please provide valid paired-end read files.");
        }
    } catch (...) {
        fm.get();  // a matcher error comes first, as in the serial order of the reference
        throw;
    }
    tm.mark("read + parse input");
    mk_matcher *m = fm.get();
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
    const size_t n = f1.recs.size();
    const uint64_t batch_bytes = (uint64_t)a.batch_mb << 20;
    // double-buffered batches: while batch k is on the GPU and its records are written out, a
    // second thread gathers the sequences of batch k + 1
    struct Batch {
        size_t b0 = 0, b1 = 0;
        std::vector<uint8_t> s1, s2;
        std::vector<uint64_t> o1, o2;
    } bufs[2];
    std::vector<uint8_t> keep;
    std::vector<mk_row> rows(4096);
    auto fill = [&](Batch &b, size_t from) {
        size_t i = from;
        uint64_t bytes = 0;
        while (i < n && (bytes < batch_bytes || i == from)) {
            bytes += f1.raw_len(i) + (paired ? f2.raw_len(i) : 0);
            ++i;
        }
        b.b0 = from;
        b.b1 = i;
        f1.gather(from, i, b.s1, b.o1);
        if (paired) f2.gather(from, i, b.s2, b.o2);
    };
    int cur = 0;
    if (n) fill(bufs[0], 0);
    tm.mark("batch: gather sequences (first)");
    while (n && bufs[cur].b0 < n) {
        Batch &B = bufs[cur];
        const size_t b0 = B.b0, i = B.b1;
        std::vector<uint8_t> &s1 = B.s1, &s2 = B.s2;
        std::vector<uint64_t> &o1 = B.o1, &o2 = B.o2;
        std::future<void> next;
        Batch &N = bufs[cur ^ 1];
        N.b0 = n;  // "no further batch" unless filled below
        if (i < n) next = std::async(std::launch::async, [&, i] { fill(N, i); });
        const uint64_t nb = i - b0;
        keep.assign(nb, 0);
        uint64_t n_rows = 0;
        for (;;) {
            mk_counters cb;
            memset(&cb, 0, sizeof(cb));
            std::vector<uint32_t> cnt_b(counts.size(), 0);
            int rc = paired ? mk_extract_paired(m, s1.data(), o1.data(), nb, s2.data(), o2.data(), nb, lg.active, a.invert_match,
                                                keep.data(), rows.data(), rows.size(), &n_rows, &cb, cnt_b.data())
                            : mk_extract_single(m, s1.data(), o1.data(), nb, lg.active, a.invert_match, keep.data(), rows.data(),
                                                rows.size(), &n_rows, &cb, cnt_b.data());
            if (rc == MK_E_CAPACITY && n_rows > rows.size()) {
                rows.resize(n_rows);
                continue;
            }
            mk_check(rc, "Error during matching");
            c.nb_records_tot += cb.nb_records_tot; c.nb_bases += cb.nb_bases;
            c.nb_hits_tot[0] += cb.nb_hits_tot[0]; c.nb_hits_tot[1] += cb.nb_hits_tot[1];
            c.nb_records_hit[0] += cb.nb_records_hit[0]; c.nb_records_hit[1] += cb.nb_records_hit[1];
            c.nb_records_extracted += cb.nb_records_extracted;
            for (size_t k = 0; k < counts.size(); ++k) counts[k] += cnt_b[k];
            break;
        }
        tm.mark("batch: H2D + scan + D2H");
        if (lg.active)
            for (uint64_t k = 0; k < n_rows; ++k) {
                const mk_row &r = rows[k];
                const FastxFile &ff = r.file ? f2 : f1;
                const std::string id = ff.id(b0 + r.rec);
                lg.text.row(r.file ? name2 : name1, id, pats.list[r.pat], r.pos);
                if (lg.has_json) lg.json.row(r.file ? name2 : name1, id, pats.list[r.pat], r.pos);
            }
        if (!a.suppress_output)
            for (uint64_t k = 0; k < nb; ++k)
                if (keep[k]) {
                    f1.write(b0 + k, w1);
                    if (paired) f2.write(b0 + k, w2);
                }
        if (next.valid()) next.get();
        tm.mark("batch: rows + records out, wait for next gather");
        cur ^= 1;
    }
    w1.flush();
    w2.flush();
    tm.mark("log rows + write records");
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
    mk_matcher_destroy(m);
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
    std::future<mk_matcher *> fm = std::async(std::launch::async, [&] { return make_matcher(a, pats, &use_ac); });

    SamFile sam;
    try {
        sam.parse(a.in_file);  // "Input file must be a BAM or SAM file." for other extensions
    } catch (...) {
        fm.get();  // a matcher error comes first, as in the serial order of the reference
        throw;
    }
    tm.mark("parse");
    mk_matcher *m = fm.get();
    tm.mark("matcher (HIP init), remainder");
    if (out_ext != "sam" && out_ext != "bam" && out_ext != "STDOUT") bail("Output file must be a BAM or SAM file.");
    Sink w;
    BamWriter bw;
    const bool to_bam = out_ext == "bam" && !a.suppress_output;
    // header + @PG line (src/cmd_tag.rs:509-514)
    const std::string out_header =
        sam.header + "@PG\tID:" + kProgram + "\tPN:" + kProgram + "\tCL:" + join(argv) + "\tVN:" + kVersion + "\n";
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

    const size_t n = sam.recs.size();
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off;
    sam.gather(seq, off);
    tm.mark("gather sequences");
    mk_counters c;
    memset(&c, 0, sizeof(c));
    std::vector<uint32_t> counts(pats.list.size(), 0);
    std::vector<uint8_t> keep(std::max<size_t>(n, 1), 0);
    std::vector<mk_row> rows(4096);
    std::vector<uint64_t> foff(n + 1, 0);
    std::vector<uint32_t> fpat(1024);
    uint64_t n_rows = 0;
    for (;;) {
        memset(&c, 0, sizeof(c));
        std::fill(counts.begin(), counts.end(), 0);
        int rc = mk_tag_records(m, seq.data(), off.data(), n, lg.active, a.filter_matching, a.invert_match, keep.data(), rows.data(),
                                rows.size(), &n_rows, &c, counts.data(), foff.data(), fpat.data(), fpat.size());
        if (rc == MK_E_CAPACITY && (n_rows > rows.size() || foff[n] > fpat.size())) {
            rows.resize(std::max<uint64_t>(rows.size(), n_rows));
            fpat.resize(std::max<uint64_t>(fpat.size(), foff[n]));
            continue;
        }
        mk_check(rc, "Error during matching");
        break;
    }
    tm.mark("scan");
    if (lg.active)
        for (uint64_t k = 0; k < n_rows; ++k) {
            const mk_row &r = rows[k];
            const std::string name = sam.name(r.rec);
            lg.text.row(in_name, name, pats.list[r.pat], r.pos);
            if (lg.has_json) lg.json.row(in_name, name, pats.list[r.pat], r.pos);
        }
    // tag + write kept records (src/cmd_tag.rs:457-497): tag values and output encoding are built
    // on every host thread, a slab of records at a time, and written in record order
    std::vector<size_t> kept;
    kept.reserve(n);
    for (size_t k = 0; k < n; ++k)
        if (keep[k]) kept.push_back(k);
    const size_t kSlab = 1 << 17;
    for (size_t c0 = 0; c0 < kept.size() && !a.suppress_output; c0 += kSlab) {
        const size_t c1 = std::min(kept.size(), c0 + kSlab);
        const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), (c1 - c0) / 4096 + 1));
        std::vector<std::vector<uint8_t>> bin(to_bam ? T : 0);
        std::vector<std::string> txt(to_bam ? 0 : T);
        run_threads(T, [&](size_t t) {
            std::vector<char> val(4096);
            std::string line;
            for (size_t i = c0 + (c1 - c0) * t / T; i < c0 + (c1 - c0) * (t + 1) / T; ++i) {
                const size_t k = kept[i];
                std::string existing;
                const int has = sam.find_tag(k, a.tag, &existing);
                if (has == 2) bail("Invalid tag value format. Expected string value.");
                size_t need = 0;
                for (;;) {
                    int rc = mk_tag_value(m, fpat.data() + foff[k], foff[k + 1] - foff[k], has == 1 ? existing.c_str() : nullptr,
                                          val.data(), val.size(), &need);
                    if (rc == MK_E_CAPACITY) {
                        val.resize(need + 1);
                        continue;
                    }
                    mk_check(rc, "Error building tag value");
                    break;
                }
                if (to_bam && sam.is_bam) {
                    BamWriter::append_tagged_raw(sam.raw(k), sam.raw_len(k), a.tag, val.data(), need, bin[t]);
                } else if (to_bam) {
                    line.clear();
                    sam.append_line(k, line);
                    line += '\t';
                    line += a.tag;
                    line += ":Z:";
                    line.append(val.data(), need);
                    bw.encode_record(line, bin[t]);
                } else {
                    std::string &o = txt[t];
                    sam.append_line(k, o);
                    o += '\t';
                    o += a.tag;
                    o += ":Z:";
                    o.append(val.data(), need);
                    o += '\n';
                }
            }
        });
        for (size_t t = 0; t < T; ++t) {
            if (to_bam)
                bw.put_encoded(bin[t]);
            else
                w.write(txt[t]);
        }
    }
    if (a.suppress_output)  // the reference still validates existing tags of kept records
        for (size_t k : kept) {
            std::string existing;
            if (sam.find_tag(k, a.tag, &existing) == 2) bail("Invalid tag value format. Expected string value.");
        }
    w.flush();
    bw.close();
    tm.mark("write");
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
    mk_matcher_destroy(m);
    return 0;
}

}  // namespace cli
