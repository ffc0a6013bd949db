// test harness: formats log rows with the CLI's direct formatters (TextLogger::format / JsonLogger::format,
// merkurio_amd/csrc/cli/util.cpp) -- tests/test_cli_cpu.py compares with Python's json module.
// input file: records separated by 0x1e, fields (file, id, pattern, position) by 0x1f; output: <text rows> 0x1d <json rows>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "util.hpp"
using namespace cli;
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    std::string in;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) in.append(buf, n);
    fclose(f);
    std::string text, json;
    bool first = true;
    size_t p = 0;
    while (p < in.size()) {
        size_t e = in.find('\x1e', p);
        if (e == std::string::npos) e = in.size();
        std::vector<std::string> fld;
        size_t q = p;
        while (q <= e) {
            size_t g = in.find('\x1f', q);
            if (g == std::string::npos || g > e) g = e;
            fld.emplace_back(in, q, g - q);
            q = g + 1;
        }
        TextLogger::format(text, fld[0], fld[1].data(), fld[1].size(), fld[2], strtoull(fld[3].c_str(), nullptr, 10));
        JsonLogger::format(json, !first, fld[0], fld[1].data(), fld[1].size(), fld[2], strtoull(fld[3].c_str(), nullptr, 10));
        first = false;
        p = e + 1;
    }
    fwrite(text.data(), 1, text.size(), stdout);
    fputc('\x1d', stdout);
    fwrite(json.data(), 1, json.size(), stdout);
    return 0;
}
