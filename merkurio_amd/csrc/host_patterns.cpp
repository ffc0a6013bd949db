// host_patterns.cpp -- pattern-list preparation and the small pure functions of the matcher
// surface.  Product code (C++), independent of oracle/.
//   parse_pattern_list / read_kmers_from_file   src/helpers.rs:76-163
//   recommend_aho_corasick                      src/helpers.rs:203-211
//   tune_q_value                                src/pattern_matching.rs:213-225
//   generate_masks                              src/pattern_preprocessing.rs:24-43
//   reverse_complement / canonical              needletail 0.6.3 (used at src/helpers.rs:103,117)
#include <sched.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
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

bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// ---- flat pattern lists: one arena + a 16-byte reference per pattern ------------------------------------------
// key = the first 8 bytes, big-endian, zero padded: unequal keys order two patterns like their bytes do (a pattern that
// ends inside the 8 bytes sorts before its extensions); equal keys fall through to memcmp + length.
struct Ref {
    uint64_t key;
    uint32_t off, len;
};
inline Ref make_ref(const uint8_t *arena, uint32_t off, uint32_t len) {
    uint64_t k = 0;
    const uint8_t *p = arena + off;
    for (uint32_t i = 0; i < 8 && i < len; ++i) k |= (uint64_t)p[i] << (56 - 8 * i);
    return Ref{k, off, len};
}
struct RefLess {
    const uint8_t *A;
    bool operator()(const Ref &a, const Ref &b) const {
        if (a.key != b.key) return a.key < b.key;
        const uint32_t n = a.len < b.len ? a.len : b.len;
        const int c = n ? memcmp(A + a.off, A + b.off, n) : 0;
        return c ? c < 0 : a.len < b.len;
    }
};
inline bool ref_equal(const uint8_t *A, const Ref &a, const Ref &b) {
    return a.key == b.key && a.len == b.len && memcmp(A + a.off, A + b.off, a.len) == 0;
}

unsigned thread_count(uint64_t n_items) {
    if (n_items < 50000) return 1;
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;  // the cores this process may really use (a container's share)
    if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = std::min<unsigned>(hw ? hw : 1, (unsigned)CPU_COUNT(&set));
    return std::max(1u, std::min(hw, 16u));
}

// f(lo, hi) over [0, n) in T contiguous slices, one thread each (the caller's thread takes the first)
template <class F>
void parallel_for(unsigned T, uint64_t n, F f) {
    if (T <= 1 || n < T) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex mu;
    auto run = [&](uint64_t lo, uint64_t hi) {
        try {
            f(lo, hi);
        } catch (...) {
            std::lock_guard<std::mutex> g(mu);
            if (!err) err = std::current_exception();
        }
    };
    unsigned started = 1;
    try {
        for (; started < T; ++started) th.emplace_back(run, n * started / T, n * (started + 1) / T);
    } catch (...) {  // no more threads to be had: the caller's thread takes the slices that got none
    }
    run(0, n / T);
    for (unsigned t = started; t < T; ++t) run(n * t / T, n * (t + 1) / T);
    for (auto &x : th) x.join();
    if (err) std::rethrow_exception(err);
}

// sort + dedup.  T > 1: sample sort -- T - 1 splitters from a sorted sample, every thread classifies its slice and
// scatters it, then sorts and dedups one range; equal patterns always land in the same range.
void sort_unique(const uint8_t *A, std::vector<Ref> &refs, unsigned T) {
    const RefLess less{A};
    const uint64_t n = refs.size();
    if (T <= 1 || n < 4096) {
        std::sort(refs.begin(), refs.end(), less);
        refs.erase(std::unique(refs.begin(), refs.end(), [A](const Ref &a, const Ref &b) { return ref_equal(A, a, b); }), refs.end());
        return;
    }
    std::vector<Ref> sample;
    const uint64_t step = std::max<uint64_t>(1, n / (T * 64ull));
    for (uint64_t i = step / 2; i < n; i += step) sample.push_back(refs[i]);
    std::sort(sample.begin(), sample.end(), less);
    std::vector<Ref> split;
    for (unsigned t = 1; t < T; ++t) split.push_back(sample[sample.size() * t / T]);
    std::vector<uint64_t> count((size_t)T * T, 0);  // [slice][range]
    std::vector<uint8_t> cls(n);
    parallel_for(T, T, [&](uint64_t t0, uint64_t t1) {
        for (uint64_t t = t0; t < t1; ++t)
            for (uint64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
                const unsigned b = (unsigned)(std::upper_bound(split.begin(), split.end(), refs[i], less) - split.begin());
                cls[i] = (uint8_t)b;
                count[t * T + b]++;
            }
    });
    std::vector<uint64_t> start((size_t)T * T), range_lo(T + 1, 0);
    uint64_t acc = 0;
    for (unsigned b = 0; b < T; ++b) {
        range_lo[b] = acc;
        for (unsigned t = 0; t < T; ++t) {
            start[(size_t)t * T + b] = acc;
            acc += count[(size_t)t * T + b];
        }
    }
    range_lo[T] = acc;
    std::vector<Ref> out(n);
    parallel_for(T, T, [&](uint64_t t0, uint64_t t1) {
        for (uint64_t t = t0; t < t1; ++t) {
            uint64_t *cur = &start[t * T];
            for (uint64_t i = n * t / T; i < n * (t + 1) / T; ++i) out[cur[cls[i]]++] = refs[i];
        }
    });
    std::vector<uint64_t> kept(T, 0);
    parallel_for(T, T, [&](uint64_t b0, uint64_t b1) {
        for (uint64_t b = b0; b < b1; ++b) {
            Ref *lo = out.data() + range_lo[b], *hi = out.data() + range_lo[b + 1];
            std::sort(lo, hi, less);
            kept[b] = (uint64_t)(std::unique(lo, hi, [A](const Ref &x, const Ref &y) { return ref_equal(A, x, y); }) - lo);
        }
    });
    uint64_t w = 0;
    for (unsigned b = 0; b < T; ++b) {  // close the gaps the duplicates left
        if (w != range_lo[b]) std::move(out.begin() + range_lo[b], out.begin() + range_lo[b] + kept[b], out.begin() + w);
        w += kept[b];
    }
    out.resize(w);
    refs.swap(out);
}

int export_refs(const uint8_t *A, const std::vector<Ref> &v, unsigned T, uint8_t **out_bytes, uint32_t **out_off, uint32_t *out_n) {
    uint64_t total = 0;
    for (auto &r : v) total += r.len;
    if (total > 0xFFFFFFFFull) return mk::fail(MK_E_UNSUPPORTED, "pattern list too large (%llu bytes)", (unsigned long long)total);
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
        w += v[i].len;
    }
    o[v.size()] = w;
    parallel_for(T, v.size(), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) memcpy(b + o[i], A + v[i].off, v[i].len);
    });
    *out_bytes = b;
    *out_off = o;
    *out_n = (uint32_t)v.size();
    return MK_OK;
}

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
    if (len > 0xFFFFFFFFull) return mk::fail(MK_E_UNSUPPORTED, "k-mer file too large (%zu bytes)", len);
    std::vector<Ref> v;  // (offset, length) into `content`; the key field is not used here
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
            v.push_back(Ref{0, (uint32_t)a, (uint32_t)(b - a)});
        }
        i = e + 1;
    }
    if (v.empty()) return mk::fail(MK_E_NO_PATTERNS, "No k-mers found in the file.");
    return export_refs(content, v, thread_count(v.size()), out_bytes, out_off, out_n);
    MK_ABI_END
}

int mk_parse_pattern_list(const uint8_t *in_bytes, const uint32_t *in_off, uint32_t n_in, int reverse_complement,
                          int canonical, int lowercase, int uppercase, uint8_t **out_bytes, uint32_t **out_off,
                          uint32_t *out_n) {
    if (!out_bytes || !out_off || !out_n || (n_in && (!in_off || !in_bytes)))
        return mk::fail(MK_E_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < n_in; ++i)
        if (in_off[i + 1] < in_off[i]) return mk::fail(MK_E_INVALID_ARG, "pattern offsets not monotone");
    MK_ABI_BEGIN
    // One flat arena (the transformed inputs, then their reverse complements) and one 16-byte reference per pattern:
    // no allocation per pattern.  Every stage runs on all host threads for lists of 50 k patterns and more
    // (500 k 21-mers: 0.25 s on one thread, r02; see profiles/r04_compile_time.txt).
    const uint64_t total_in = n_in ? (uint64_t)in_off[n_in] - in_off[0] : 0;
    const uint64_t n_all = (uint64_t)n_in * (reverse_complement ? 2 : 1);
    if (n_all > 0xFFFFFFFFull || total_in * (reverse_complement ? 2 : 1) > 0xFFFFFFFFull)
        return mk::fail(MK_E_UNSUPPORTED, "pattern list too large (%llu patterns)", (unsigned long long)n_all);
    std::vector<uint8_t> arena(total_in * (reverse_complement ? 2 : 1) + 8);
    std::vector<Ref> refs(n_all);
    const uint32_t base = n_in ? in_off[0] : 0;
    const unsigned T = thread_count(n_all);
    uint8_t *const A = arena.data();
    parallel_for(T, n_in, [&](uint64_t lo, uint64_t hi) {
        std::vector<uint8_t> tmp;
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t o = in_off[i] - base, len = in_off[i + 1] - in_off[i];
            uint8_t *d = A + o;
            memcpy(d, in_bytes + in_off[i], len);
            if (lowercase) {
                for (uint32_t k = 0; k < len; ++k)
                    if (d[k] >= 'A' && d[k] <= 'Z') d[k] |= 0x20;
            } else if (uppercase) {
                for (uint32_t k = 0; k < len; ++k)
                    if (d[k] >= 'a' && d[k] <= 'z') d[k] &= (uint8_t)~0x20;
            }
            if (reverse_complement) {  // src/helpers.rs:100-107: the list is extended by the reverse complements
                uint8_t *r = A + total_in + o;
                mk_reverse_complement(d, len, r);
                if (canonical) {  // (-r and -c together: both copies are canonicalised, like the reference's two loops)
                    tmp.resize(len);
                    mk_canonical(r, len, tmp.data());
                    memcpy(r, tmp.data(), len);
                }
                refs[n_in + i] = make_ref(A, (uint32_t)(total_in + o), len);
            }
            if (canonical) {
                tmp.resize(len);
                mk_canonical(d, len, tmp.data());
                memcpy(d, tmp.data(), len);
            }
            refs[i] = make_ref(A, o, len);
        }
    });
    // drop empty, sort (bytes as unsigned char: Rust's str Ord), dedup -- src/helpers.rs:124-126
    refs.erase(std::remove_if(refs.begin(), refs.end(), [](const Ref &r) { return r.len == 0; }), refs.end());
    sort_unique(A, refs, T);
    if (refs.empty()) return mk::fail(MK_E_NO_PATTERNS, "No k-mers found in file or provided sequence.");
    return export_refs(A, refs, T, out_bytes, out_off, out_n);
    MK_ABI_END
}

}  // extern "C"
