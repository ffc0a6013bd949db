// test harness: runs the CLI's bzip2 / xz / zstd decoders (merkurio_amd/csrc/cli/decompress.cpp) on files and
// prints size + FNV-1a hash of what came out; used by tests/test_cli_cpu.py (no GPU needed)
#include "io.hpp"  // -I merkurio_amd/csrc/cli
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
using namespace cli;
int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        int fd = open(argv[i], O_RDONLY); struct stat st; fstat(fd, &st);
        void *m = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        std::vector<char> out;
        try {
            bool ok = inflate_by_magic(argv[i], (const unsigned char *)m, st.st_size, out);
            unsigned long h = 1469598103934665603ul; for (char c : out) h = (h ^ (unsigned char)c) * 1099511628211ul;
            printf("%s: %d %zu bytes hash %lx\n", argv[i], ok, out.size(), h);
        } catch (const Error &e) { printf("%s: error %s\n", argv[i], e.what()); }
    }
}
// the harness links decompress.cpp alone: the memory check lives in io.cpp
namespace cli { void require_host_memory(uint64_t, const std::string &) {} }
