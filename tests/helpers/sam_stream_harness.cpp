// test harness: reads a SAM / BAM file through the CLI's windowed reader (SamFile::open / fill,
// merkurio_amd/csrc/cli/io.cpp) and prints the header, then one line per record: name, the sequence as
// the matcher sees it, the value of an existing km tag, the record as SAM text.
// usage: harness <file> <window bytes> [prefetch]   (prefetch: the next window is read by SamFile::prefetch on a second thread
//        while the current one is printed, as `merkurio tag` does)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>

#include "io.hpp"
using namespace cli;
int main(int argc, char **argv) {
    try {
        SamFile f;
        f.open(argv[1]);
        const uint64_t w = strtoull(argv[2], nullptr, 10);
        fputs(f.header.c_str(), stdout);
        size_t windows = 0, total = 0;
        const bool ahead = argc > 3 && !strcmp(argv[3], "prefetch");
        bool more = f.fill(w);
        while (more) {
            ++windows;
            std::future<void> next;
            if (ahead) next = std::async(std::launch::async, [&] { f.prefetch(w); });
            std::vector<uint8_t> seq;
            std::vector<uint64_t> off;
            f.gather(0, f.recs.size(), seq, off);
            for (size_t i = 0; i < f.recs.size(); ++i) {
                std::string km, line;
                const int has = f.find_tag(i, "km", &km);
                f.append_line(i, line);
                printf("%s\t%.*s\t%d:%s\t|%s\n", f.name(i).c_str(), (int)(off[i + 1] - off[i]), (const char *)seq.data() + off[i], has, km.c_str(),
                       line.c_str());
            }
            total += f.recs.size();
            if (ahead) next.get();
            more = f.fill(w);
        }
        printf("#windows %zu records %zu\n", windows, total);
    } catch (const Error &e) {
        printf("#error %s\n", e.what());
    }
    return 0;
}
