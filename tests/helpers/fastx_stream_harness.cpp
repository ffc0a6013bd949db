// test harness: reads a FASTA/FASTQ file through the CLI's windowed reader (FastxStream,
// merkurio_amd/csrc/cli/io.cpp) and prints one line per record -- id, sequence (line breaks squeezed
// out), quality -- plus the number of windows; tests/test_cli_cpu.py compares with a Python parse.
// usage: harness <file> <window bytes> [consume at most N records per window (0 = all)] [digest]
// with a fourth argument: one line per record with id, sequence length and FNV-1a hash instead of the text
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
        const bool digest = argc > 4;
        size_t windows = 0, total = 0;
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
