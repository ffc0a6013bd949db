// test harness: reads a FASTA/FASTQ file through the CLI's windowed reader (FastxStream,
// merkurio_amd/csrc/cli/io.cpp) and prints one line per record -- id, sequence (line breaks squeezed
// out), quality -- plus the number of windows; tests/test_cli_cpu.py compares with a Python parse.
// usage: harness <file> <window bytes> [consume at most N records per window (0 = all)] [digest | raw]
// digest: one line per record with id, sequence length and FNV-1a hash instead of the text
// raw: through raw_fill() / raw_rest() (the windows extract hands to the GPU unparsed)
#include <cstdio>
#include <cstdlib>

#include "io.hpp"
using namespace cli;
int main(int argc, char **argv) {
    try {
        FastxStream s;
        s.open(argv[1]);
        const uint64_t w = strtoull(argv[2], nullptr, 10);
        const size_t cap = (argc > 3 && strtoull(argv[3], nullptr, 10)) ? strtoull(argv[3], nullptr, 10) : (size_t)-1;
        const bool digest = argc > 4 && argv[4][0] == 'd';
        size_t windows = 0, total = 0;
        auto print = [&](const FastxFile &f, size_t i) {
            std::vector<uint8_t> seq;
            f.append_seq(i, seq);
            const auto &r = f.recs[i];
            printf("%s\t%.*s\t%.*s\n", f.id(i).c_str(), (int)seq.size(), (const char *)seq.data(), (int)(r.qual_e - r.qual_b), f.data + r.qual_b);
        };
        if (argc > 4 && argv[4][0] == 'r') {
            // raw windows (extract's text windows: FASTQ or FASTA, handed to the GPU unparsed): every window starts at a record start
            // and holds whole records; its records are printed by the host parser, the checker of the device's index.  What
            // raw_fill() does not take (text that does not start like a record) comes out of raw_rest() as one last piece.
            const char *text = nullptr;
            uint64_t n = 0, resume = 0;
            bool more = s.raw_fill(w, &text, &n, &resume);
            auto parse_and_print = [&](const std::vector<char> &copy) {
                FastxFile f;
                f.fastq = s.raw_fastq();
                f.data = copy.data();
                f.data_n = copy.size();
                if (!copy.empty() && copy[0] != (f.fastq ? '@' : '>')) {
                    printf("#error raw window does not start at a record\n");
                    exit(0);
                }
                f.parse_span(0, copy.size());
                for (size_t i = 0; i < f.recs.size(); ++i) print(f, i);
                total += f.recs.size();
            };
            while (more) {
                std::vector<char> copy(text, text + n);  // (the CLI works on its pinned copy while the stream moves on)
                s.raw_consume();
                // the next window is fetched while this one is in use, as the CLI's reader thread does
                more = s.raw_fill(w, &text, &n, &resume);
                ++windows;
                parse_and_print(copy);
            }
            if (s.raw_rest(&text, &n) && n) {
                std::vector<char> copy(text, text + n);
                size_t p = 0;
                while (p < copy.size() && (copy[p] == '\n' || copy[p] == '\r')) ++p;
                copy.erase(copy.begin(), copy.begin() + (ptrdiff_t)p);
                if (!copy.empty()) ++windows, parse_and_print(copy);
            }
            printf("#windows %zu records %zu\n", windows, total);
            return 0;
        }
        while (s.fill(w)) {
            ++windows;
            const size_t n = std::min(cap, s.view.recs.size());
            for (size_t i = 0; i < n; ++i) {
                std::vector<uint8_t> seq;
                s.view.append_seq(i, seq);
                const auto &r = s.view.recs[i];
                if (digest) {
                    uint64_t h = 1469598103934665603ull;
                    for (uint8_t c : seq) h = (h ^ c) * 1099511628211ull;
                    printf("%s\t%zu\t%llu\n", s.view.id(i).c_str(), seq.size(), (unsigned long long)h);
                    continue;
                }
                printf("%s\t%.*s\t%.*s\n", s.view.id(i).c_str(), (int)seq.size(), (const char *)seq.data(), (int)(r.qual_e - r.qual_b),
                       s.view.data + r.qual_b);
            }
            total += n;
            s.consume(n);
        }
        printf("#windows %zu records %zu\n", windows, total);
    } catch (const Error &e) {
        printf("#error %s\n", e.what());
    }
    return 0;
}
