// host_patterns.cpp -- pattern-list preparation and the small pure functions of the matcher
// surface.  Product code (C++), independent of oracle/.
//   parse_pattern_list / read_kmers_from_file   src/helpers.rs:76-163
//   recommend_aho_corasick                      src/helpers.rs:203-211
//   tune_q_value                                src/pattern_matching.rs:213-225
//   generate_masks                              src/pattern_preprocessing.rs:24-43
//   reverse_complement / canonical              needletail 0.6.3 (used at src/helpers.rs:103,117)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_common.h"

namespace {

struct ComplementTable {
    uint8_t t[256];
    ComplementTable() {
        for (int i = 0; i < 256; ++i) t[i] = (uint8_t)i;  // everything else passes through
        const char *from = "ACGTRYKMBVDH", *to = "TGCAYRMKVBHD";
        for (int i = 0; from[i]; ++i) {
            t[(uint8_t)from[i]] = (uint8_t)to[i];
            t[(uint8_t)(from[i] | 0x20)] = (uint8_t)(to[i] | 0x20);  // case preserved
        }
    }
};
const ComplementTable kComp;

int export_list(const std::vector<std::string> &v, uint8_t **out_bytes, uint32_t **out_off, uint32_t *out_n) {
    size_t total = 0;
    for (auto &s : v) total += s.size();
    uint8_t *b = (uint8_t *)malloc(total ? total : 1);
    uint32_t *o = (uint32_t *)malloc((v.size() + 1) * sizeof(uint32_t));
    if (!b || !o) {
        free(b);
        free(o);
        return mk::fail(MK_E_NOMEM, "out of memory");
    }
    uint32_t w = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        o[i] = w;
        memcpy(b + w, v[i].data(), v[i].size());
        w += (uint32_t)v[i].size();
    }
    o[v.size()] = w;
    *out_bytes = b;
    *out_off = o;
    *out_n = (uint32_t)v.size();
    return MK_OK;
}

bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

}  // namespace

extern "C" {

void mk_reverse_complement(const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = kComp.t[in[n - 1 - i]];
}

void mk_canonical(const uint8_t *in, size_t n, uint8_t *out) {
    // lexicographic minimum of the sequence and its reverse complement
    for (size_t i = 0; i < n; ++i) {
        uint8_t r = kComp.t[in[n - 1 - i]];
        if (r != in[i]) {
            if (r < in[i])
                mk_reverse_complement(in, n, out);
            else
                memmove(out, in, n);
            return;
        }
    }
    memmove(out, in, n);
}

int mk_recommend_aho_corasick(size_t num_patterns, size_t max_len) { return num_patterns >= 14 || max_len > 64; }

size_t mk_tune_q_value(size_t n) {
    static const struct {
        size_t hi, q;
    } steps[] = {{1, 1}, {3, 2}, {8, 3}, {30, 4}, {55, 5}, {64, 6}};
    for (auto &s : steps)
        if (n <= s.hi) return s.q;
    return 0;  // the reference bails: "Pattern length is too long for BNDMq."
}

int mk_generate_masks(const uint8_t *pattern, size_t m, uint64_t masks[256], uint64_t *accept) {
    if (!masks || !accept || (!pattern && m)) return mk::fail(MK_E_INVALID_ARG, "null argument");
    std::fill(masks, masks + 256, 0);
    *accept = 0;
    if (m > 64)
        return mk::fail(MK_E_PATTERN_TOO_LONG,
                        "Pattern length %zu is too large for this architecture when using BNDM (max 64).", m);
    for (size_t j = 0; j < m; ++j) masks[pattern[j]] |= 1ull << (m - 1 - j);
    if (m) *accept = 1ull << (m - 1);
    return MK_OK;
}

int mk_read_kmers_from_text(const uint8_t *content, size_t len, uint8_t **out_bytes, uint32_t **out_off,
                            uint32_t *out_n) {
    if (!out_bytes || !out_off || !out_n) return mk::fail(MK_E_INVALID_ARG, "null output");
    MK_ABI_BEGIN
    std::vector<std::string> v;
    size_t i = 0;
    while (i < len) {  // str::lines(): split on '\n', strip one trailing '\r'
        const uint8_t *nl = (const uint8_t *)memchr(content + i, '\n', len - i);
        size_t e = nl ? (size_t)(nl - content) : len;
        size_t le = e;
        if (le > i && content[le - 1] == '\r') --le;
        // filter BEFORE trimming: drop empty lines and lines starting with '#' or '>'
        if (le > i && content[i] != '#' && content[i] != '>') {
            size_t a = i, b = le;
            while (a < b && is_space(content[a])) ++a;
            while (b > a && is_space(content[b - 1])) --b;
            v.emplace_back((const char *)content + a, b - a);
        }
        i = e + 1;
    }
    if (v.empty()) return mk::fail(MK_E_NO_PATTERNS, "No k-mers found in the file.");
    return export_list(v, out_bytes, out_off, out_n);
    MK_ABI_END
}

int mk_parse_pattern_list(const uint8_t *in_bytes, const uint32_t *in_off, uint32_t n_in, int reverse_complement,
                          int canonical, int lowercase, int uppercase, uint8_t **out_bytes, uint32_t **out_off,
                          uint32_t *out_n) {
    if (!out_bytes || !out_off || !out_n || (n_in && (!in_off || !in_bytes)))
        return mk::fail(MK_E_INVALID_ARG, "null argument");
    MK_ABI_BEGIN
    std::vector<std::string> v;
    v.reserve((size_t)n_in * (reverse_complement ? 2 : 1));
    for (uint32_t i = 0; i < n_in; ++i) v.emplace_back((const char *)in_bytes + in_off[i], in_off[i + 1] - in_off[i]);
    if (lowercase) {
        for (auto &s : v)
            for (auto &c : s)
                if (c >= 'A' && c <= 'Z') c = (char)(c | 0x20);
    } else if (uppercase) {
        for (auto &s : v)
            for (auto &c : s)
                if (c >= 'a' && c <= 'z') c = (char)(c & ~0x20);
    }
    if (reverse_complement) {
        for (uint32_t i = 0; i < n_in; ++i) {
            std::string rc(v[i].size(), '\0');
            mk_reverse_complement((const uint8_t *)v[i].data(), v[i].size(), (uint8_t *)&rc[0]);
            v.push_back(std::move(rc));
        }
    }
    if (canonical) {
        for (auto &s : v) {
            std::string c(s.size(), '\0');
            mk_canonical((const uint8_t *)s.data(), s.size(), (uint8_t *)&c[0]);
            s.swap(c);
        }
    }
    v.erase(std::remove_if(v.begin(), v.end(), [](const std::string &s) { return s.empty(); }), v.end());
    std::sort(v.begin(), v.end());  // std::string compares bytes as unsigned char: same as Rust's str Ord
    v.erase(std::unique(v.begin(), v.end()), v.end());
    if (v.empty()) return mk::fail(MK_E_NO_PATTERNS, "No k-mers found in file or provided sequence.");
    return export_list(v, out_bytes, out_off, out_n);
    MK_ABI_END
}

}  // extern "C"
