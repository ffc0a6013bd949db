// main.cpp -- `merkurio extract|tag`: the reference's command line (src/main.rs:14-54 and the
// clap structs of cmd_extract.rs / cmd_tag.rs) in front of the MI355X matcher library.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "commands.hpp"

#include <unistd.h>

using namespace cli;

namespace {

struct Spec {
    char short_name;
    const char *long_name;
    int kind;  // 0 flag, 1 one value, 2 one-or-more values, 3 optional value (default STDOUT)
};

const Spec kCommon[] = {{'s', "kmer-seq", 2},      {'f', "kmer-file", 1},    {'r', "reverse-complement", 0}, {'c', "canonical", 0},
                        {'l', "out-log", 3},       {'j', "json-log", 3},     {'S', "suppress-output", 0},    {'v', "invert-match", 0},
                        {'I', "case-insensitive", 0}, {'L', "lowercase", 0}, {'U', "uppercase", 0},          {'q', "q-size", 1},
                        {'a', "aho-corasick", 0},  {0, "device", 1},         {0, "batch-mb", 1},           {0, "gpus", 1},                 {0, "window-mb", 1},
                        {0, "host-ingest", 0},     {0, "host-codec", 0},     {0, "device-codec-always", 0}};
const Spec kExtract[] = {{'i', "in-fastx", 1}, {'1', "in-fastx", 1}, {'2', "in-fastq-2", 1}, {'o', "out-fastx", 1}};
const Spec kTag[] = {{'i', "in-file", 1}, {'o', "out-file", 1}, {'t', "tag", 1}, {'p', "threads", 1}, {'m', "filter-matching", 0}};

[[noreturn]] void usage_error(const std::string &msg) {
    fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", msg.c_str());
    exit(2);
}

void print_help(const char *sub) {
    if (!sub) {
        puts("SeqKatcher has two subcommands, 'extract' and 'tag'.\n\nUsage: merkurio <COMMAND>\n\nCommands:\n"
             "  extract  Search for query sequences in FASTA/Q files and extract records containing the patterns\n"
             "  tag      Tag records in a BAM/SAM file with the presence of query sequences\n\n"
             "Options:\n  -h, --help     Print help\n  -V, --version  Print version");
    } else if (!strcmp(sub, "extract")) {
        puts("Usage: merkurio extract [OPTIONS] --in-fastx <IN_FASTX> <--kmer-seq <KMER_SEQ>...|--kmer-file <KMER_FILE>>\n\n"
             "  -i, -1, --in-fastx <PATH>    (compressed: gzip) FASTQ/A input\n  -2, --in-fastq-2 <PATH>      second FASTQ file (paired-end)\n"
             "  -s, --kmer-seq <SEQ>...      query sequences\n  -f, --kmer-file <PATH>       file with one k-mer per line\n"
             "  -o, --out-fastx <PATH>       output path (extension derived from the input)\n  -r, --reverse-complement     also search reverse complements\n"
             "  -c, --canonical              search canonical forms only\n  -l, --out-log [<PATH>]       text log (stdout without a value)\n"
             "  -j, --json-log [<PATH>]      JSON log\n  -S, --suppress-output        write no records (requires -l/-j)\n"
             "  -v, --invert-match           select non-matching records\n  -I, --case-insensitive       (always Aho-Corasick)\n"
             "  -L, --lowercase | -U, --uppercase   convert the patterns\n  -q, --q-size <Q>             force BNDMq with this q\n"
             "  -a, --aho-corasick           force Aho-Corasick\n      --device <N>             HIP device ordinal [0]\n"
             "      --gpus <N>               shard the records over N GPUs (device, device+1, ...) [1]\n"
             "      --batch-mb <MB>          sequence bytes per GPU batch [128]\n"
             "      --window-mb <MB>         input text read and held per window [1024]\n"
             "      --host-ingest            parse FASTQ records on the host threads (default: a single FASTQ input is indexed on the GPU)\n"
             "      --host-codec             bgzip'ed input inflated by zlib on the host threads (default: on the GPU)\n"
             "      --device-codec-always    BGZF members go through the GPU codec however few they are (default: from 8192 members per call)");
    } else {
        puts("Usage: merkurio tag [OPTIONS] --in-file <IN_FILE> <--kmer-seq <KMER_SEQ>...|--kmer-file <KMER_FILE>>\n\n"
             "  -i, --in-file <PATH>         SAM/BAM input\n  -o, --out-file <PATH>        SAM output (stdout if absent)\n"
             "  -s, --kmer-seq <SEQ>... | -f, --kmer-file <PATH>\n  -t, --tag <TAG>              two-character tag [km]\n"
             "  -m, --filter-matching        keep only records with a hit\n  -v, --invert-match           keep only records without a hit\n"
             "  -p, --threads <N>            at most N host threads for the BAM/SAM codec work [all cores]\n  -r -c -l -j -S -I -L -U -q -a --device --gpus --batch-mb  as for extract\n"
             "      --host-codec             BGZF members deflated by zlib on the host threads (default: on the GPU)\n"
             "      --host-ingest            BAM records indexed and tagged on the host threads (default: BAM -> BAM keeps them on the GPU)\n"
             "      --device-codec-always    BGZF input inflated on the GPU however few members a window holds");
    }
}

struct Parsed {
    std::vector<std::pair<std::string, std::vector<std::string>>> opts;  // long name -> values
    const std::vector<std::string> *get(const char *name) const {
        for (auto &o : opts)
            if (o.first == name) return &o.second;
        return nullptr;
    }
};

Parsed parse(const std::vector<std::string> &args, const Spec *extra, size_t n_extra, const char *sub) {
    std::vector<Spec> specs(kCommon, kCommon + sizeof(kCommon) / sizeof(Spec));
    specs.insert(specs.end(), extra, extra + n_extra);
    Parsed p;
    size_t i = 0;  // next argument to look at
    auto is_opt = [](const std::string &s) { return s.size() >= 2 && s[0] == '-' && !(s[1] >= '0' && s[1] <= '9' && s != "-1" && s != "-2"); };
    // records one occurrence of option `sp`; value-taking options consume `inline_val` (-q5, --q-size=5)
    // or the following argument(s)
    auto commit = [&](const Spec *sp, bool has_inline, const std::string &inline_val) {
        std::vector<std::string> vals;
        if (has_inline) {
            vals.push_back(inline_val);
        } else if (sp->kind == 1) {
            if (i >= args.size()) usage_error(std::string("a value is required for '--") + sp->long_name + "' but none was supplied");
            vals.push_back(args[i++]);
        } else if (sp->kind == 2) {
            while (i < args.size() && !is_opt(args[i])) vals.push_back(args[i++]);
            if (vals.empty()) usage_error(std::string("a value is required for '--") + sp->long_name + "' but none was supplied");
        } else if (sp->kind == 3) {
            if (i < args.size() && !is_opt(args[i]))
                vals.push_back(args[i++]);
            else
                vals.push_back("STDOUT");
        }
        for (auto &o : p.opts)
            if (o.first == sp->long_name) {
                if (sp->kind != 2) usage_error(std::string("the argument '--") + sp->long_name + "' cannot be used multiple times");
                o.second.insert(o.second.end(), vals.begin(), vals.end());
                return;
            }
        p.opts.emplace_back(sp->long_name, vals);
    };
    while (i < args.size()) {
        const std::string a = args[i];
        if (a == "-h" || a == "--help") {
            print_help(sub);
            exit(0);
        }
        ++i;
        if (a.rfind("--", 0) == 0) {
            std::string name = a.substr(2), inline_val;
            bool has_inline = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) {
                inline_val = name.substr(eq + 1);
                name = name.substr(0, eq);
                has_inline = true;
            }
            const Spec *sp = nullptr;
            for (auto &s : specs)
                if (name == s.long_name) sp = &s;
            if (!sp) usage_error("unexpected argument '" + a + "' found");
            if (has_inline && sp->kind == 0) usage_error("unexpected value '" + inline_val + "' for '--" + name + "' found; no more were expected");
            commit(sp, has_inline, inline_val);
        } else if (a.size() >= 2 && a[0] == '-') {
            // clap-style cluster of short options: -rl, -vI, -rq5, -Sl log.txt; the first option
            // that takes a value ends the cluster (the rest of the word, if any, is its value)
            size_t k = 1;
            while (k < a.size()) {
                const Spec *sp = nullptr;
                for (auto &s : specs)
                    if (s.short_name && a[k] == s.short_name) sp = &s;
                if (!sp) usage_error(k == 1 ? "unexpected argument '" + a + "' found" : "unexpected argument '-" + std::string(1, a[k]) + "' found");
                ++k;
                if (sp->kind == 0) {
                    commit(sp, false, "");
                    continue;
                }
                if (k < a.size())
                    commit(sp, true, a.substr(a[k] == '=' ? k + 1 : k));
                else
                    commit(sp, false, "");
                break;
            }
        } else {
            usage_error("unexpected argument '" + a + "' found");
        }
    }
    return p;
}

size_t to_num(const std::string &s, const char *what) {
    char *e = nullptr;
    long long v = strtoll(s.c_str(), &e, 10);
    if (s.empty() || *e || v < 0) usage_error(std::string("invalid value '") + s + "' for '" + what + "'");
    return (size_t)v;
}

void fill_common(const Parsed &p, CommonArgs &c, bool has_out) {
    auto flag = [&](const char *n) { return p.get(n) != nullptr; };
    if (auto v = p.get("kmer-seq")) c.kmer_seq = *v;
    if (auto v = p.get("kmer-file")) c.kmer_file = (*v)[0];
    c.reverse_complement = flag("reverse-complement");
    c.canonical = flag("canonical");
    if (auto v = p.get("out-log")) c.out_log = (*v)[0];
    if (auto v = p.get("json-log")) c.json_log = (*v)[0];
    c.suppress_output = flag("suppress-output");
    c.invert_match = flag("invert-match");
    c.case_insensitive = flag("case-insensitive");
    c.lowercase = flag("lowercase");
    c.uppercase = flag("uppercase");
    if (auto v = p.get("q-size")) c.q_size = to_num((*v)[0], "--q-size <Q_SIZE>");
    c.aho_corasick = flag("aho-corasick");
    if (auto v = p.get("device")) c.device = (int)to_num((*v)[0], "--device");
    if (auto v = p.get("gpus")) c.gpus = (int)std::max<size_t>(1, to_num((*v)[0], "--gpus"));
    if (auto v = p.get("window-mb")) c.window_mb = (int)std::max<size_t>(1, to_num((*v)[0], "--window-mb"));
    if (auto v = p.get("batch-mb")) c.batch_mb = (int)std::max<size_t>(1, to_num((*v)[0], "--batch-mb"));
    if (p.get("host-ingest")) c.host_ingest = true;
    if (p.get("host-codec")) c.host_codec = true;
    if (p.get("device-codec-always")) c.device_codec_always = true;
    c.window_mb_given = p.get("window-mb") != nullptr;
    // clap ArgGroups (src/cmd_extract.rs:33-62, src/cmd_tag.rs:29-66)
    if (c.kmer_seq.empty() == !c.kmer_file) {
        if (c.kmer_file)
            usage_error("the argument '--kmer-seq <KMER_SEQ>...' cannot be used with '--kmer-file <KMER_FILE>'");
        usage_error("the following required arguments were not provided:\n  <--kmer-seq <KMER_SEQ>...|--kmer-file <KMER_FILE>>");
    }
    if (c.q_size && c.aho_corasick) usage_error("the argument '--q-size <Q_SIZE>' cannot be used with '--aho-corasick'");
    if ((int)c.case_insensitive + (int)c.lowercase + (int)c.uppercase > 1)
        usage_error("the arguments '--case-insensitive', '--lowercase' and '--uppercase' cannot be used together");
    if (c.canonical && c.reverse_complement) usage_error("the argument '--canonical' cannot be used with '--reverse-complement'");
    if (c.suppress_output && has_out) usage_error("the argument '--suppress-output' cannot be used with the output file argument");
    if (c.suppress_output && !c.out_log && !c.json_log)
        usage_error("the following required arguments were not provided:\n  <--out-log [<OUT_LOG>]|--json-log [<JSON_LOG>]>");
}

}  // namespace

// A finished command has flushed and closed every output (its Sinks and writers are locals of run_extract / run_tag):
// the process leaves without running the HIP runtime's exit handlers and the destructors of a job that is over --
// 0.15 s of a 0.6 s run on 20 M reads (profiles/r04_e2e_extract.txt).  Errors take the ordinary way out.
static int leave(int rc) {
    // (the exit handlers are skipped: this flush is the last chance to notice records or log text on stdout that hit
    // ENOSPC / EPIPE -- such a run must not end with status 0)
    if (fflush(stdout) != 0 || ferror(stdout)) {
        fprintf(stderr, "Error: writing to standard output failed\n");
        if (rc == 0) rc = 1;
    }
    fflush(stderr);
    // (MERKURIO_SLOW_EXIT: the ordinary way out after all -- a profiler's exit handler is what writes its trace)
    if (rc == 0 && !getenv("MERKURIO_SLOW_EXIT")) _exit(0);
    return rc;
}

int main(int argc, char **argv) {
    std::vector<std::string> all(argv, argv + argc);
    if (argc < 2 || all[1] == "-h" || all[1] == "--help" || all[1] == "help") {
        print_help(nullptr);
        return argc < 2 ? 2 : 0;
    }
    if (all[1] == "-V" || all[1] == "--version") {
        puts("merkurio 1.0.0 (MI355X-native matcher)");
        return 0;
    }
    const std::string sub = all[1];
    std::vector<std::string> rest(all.begin() + 2, all.end());
    try {
        if (sub == "extract") {
            if (rest.empty()) {
                print_help("extract");
                return 2;
            }
            Parsed p = parse(rest, kExtract, sizeof(kExtract) / sizeof(Spec), "extract");
            ExtractArgs a;
            auto in = p.get("in-fastx");
            if (!in) usage_error("the following required arguments were not provided:\n  --in-fastx <IN_FASTX>");
            a.in_fastx = (*in)[0];
            if (auto v = p.get("in-fastq-2")) a.in_fastq_2 = (*v)[0];
            if (auto v = p.get("out-fastx")) a.out_fastx = (*v)[0];
            fill_common(p, a, (bool)a.out_fastx);
            g_process_is_ending = true;
            return leave(run_extract(a, all));
        }
        if (sub == "tag") {
            if (rest.empty()) {
                print_help("tag");
                return 2;
            }
            Parsed p = parse(rest, kTag, sizeof(kTag) / sizeof(Spec), "tag");
            TagArgs a;
            auto in = p.get("in-file");
            if (!in) usage_error("the following required arguments were not provided:\n  --in-file <IN_FILE>");
            a.in_file = (*in)[0];
            if (auto v = p.get("out-file")) a.out_file = (*v)[0];
            if (auto v = p.get("tag")) a.tag = (*v)[0];
            if (auto v = p.get("threads")) {
                a.threads = (int)to_num((*v)[0], "--threads <THREADS>");
                a.threads_given = true;
            }
            a.filter_matching = p.get("filter-matching") != nullptr;
            fill_common(p, a, (bool)a.out_file);
            if (a.filter_matching && a.invert_match)
                usage_error("the argument '--filter-matching' cannot be used with '--invert-match'");
            g_process_is_ending = true;
            return leave(run_tag(a, all));
        }
        usage_error("unrecognized subcommand '" + sub + "'");
    } catch (const Error &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
