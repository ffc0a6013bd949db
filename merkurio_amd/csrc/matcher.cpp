// matcher.cpp -- host side of the C ABI: matcher handle, pattern-set compiler (q-gram filter +
// exact table), device/host batched scans, emission-order restoration.
// Reference interfaces replaced: BNDMq::new (src/pattern_matching.rs:61-78), the AhoCorasick
// builder (src/cmd_extract.rs:260-265), find_match / find_iter / find_overlapping_iter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_common.h"
#include "matcher_internal.h"
#include "scan_kernel.h"

namespace mk {

// message of the calling thread's last failure: a fixed buffer, so that reporting an
// out-of-memory condition never allocates
thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    (void)hipGetLastError();  // clear the sticky error
    return fail(e == hipErrorOutOfMemory ? MK_E_NOMEM : MK_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
}

#define MK_HIP(call)                                       \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)

// ---- filter geometry -----------------------------------------------------------------------
// Pick (q, S): sampling stride S in {16,8,4,2,1}, q-gram length q = min(32, Lmin - S + 1).
// Larger S = fewer probes per base (the scan is VALU-bound: ~17 vector ops per probe) but S x
// more filter entries, i.e. more false positives to verify.  Measured on MI355X with 10 k
// 31-mers (profiles/r01_stride_sweep.txt): S=8 (80 k entries, 0.2 % of bases become
// candidates) beats S=4 by 14 % and S=16 by 60 %.  Rule (r04): among the strides with <= 96 k
// entries, the one the cost model likes best -- samples against candidates, the latter being the
// filter's false positives and the TRUE q-gram matches of random text (n / 4^q per base, which no
// filter removes).  (Until r04: the largest S with q >= 14, for every set: 10 000 patterns of
// 15..31 bases sampled at S=2, 4.15 ms per 15 GB, where S=4 / q=12 costs one sample in two and
// 9 M more candidates; profiles/r04_mixed_sets.txt.)
constexpr uint64_t kMaxLdsEntries = 98304;
constexpr double kMCandPerLaunch = 15000.0;  // million bases of the cost model's 15 GB launch
static double bloom_fp(double entries) {  // blocked filter of filter.hpp: bit a in the low word, b and c in the high word
    const double lo = 1.0 - exp(-entries / (kBloomBlocks * 32.0)), hi = 1.0 - exp(-2.0 * entries / (kBloomBlocks * 32.0));
    return lo * hi * hi;
}
// ms per 15 GB launch of a hashed class of n patterns at stride S, q-grams of q bases (fit: profiles/r04_mixed_sets.txt):
// 0.42 per sample per 16 bases + the candidates -- true q-gram matches of random text (n / 4^q per base, at most one
// per sample) and the filter's false positives -- at per_mcand ms per million
static double hashed_cost(uint64_t n, uint32_t S, uint32_t q, double per_mcand) {
    const double per_base = std::min(1.0 / S, (double)n / pow(4.0, (double)std::min(q, 30u))) + bloom_fp((double)n * S) / S;
    return 0.42 * (16.0 / S) + kMCandPerLaunch * per_mcand * per_base;
}
static void choose_geometry(uint32_t lmin, uint64_t n_pat, const mk_matcher_options &opt, uint32_t *q, uint32_t *S,
                            uint32_t *gblocks) {
    const int forced = (int)opt.force_stride;  // tuning / test hooks: mk_matcher_create_ex only
    const bool force_global = opt.force_global_filter != 0;
    *gblocks = 0;
    if (n_pat <= kMaxLdsEntries && !force_global) {  // LDS filter: the stride the cost model likes best
        *q = std::min<uint32_t>(32, lmin);
        *S = 1;
        double best = 1e300;
        for (uint32_t s : {16u, 8u, 4u, 2u, 1u}) {
            if (s > lmin) continue;
            const uint32_t qq = std::min<uint32_t>(32, lmin - s + 1);
            if (forced) {
                if ((int)s != forced) continue;
                *q = qq;
                *S = s;
                return;
            }
            if (s > 1 && n_pat * s > kMaxLdsEntries) continue;
            // (the model knows nothing of bucket chains: more than ~2 table entries per distinct q-gram -- thousands of
            // patterns on a q-gram of a base or two -- and every candidate walks its key's whole chain)
            if (s > 1 && qq < 16 && (double)n_pat * s > 2.0 * pow(4.0, (double)qq)) continue;
            const double c = hashed_cost(n_pat, s, qq, 0.0152);
            if (c < best) {
                best = c;
                *q = qq;
                *S = s;
            }
        }
        return;
    }
    // Large set: the filter moves to global memory, where every sample costs a random 8-byte
    // read.  The L2 serves one request per channel and clock: ~270 G probes/s while the filter
    // stays within the 4 MiB of an XCD's L2, 120 G/s at 8 MiB, 56 G/s from HBM
    // (profiles/r01_gather_rate.txt) -- and the scan runs at that limit (S=4: 781 M probes in
    // 2.85 ms).  So: the largest stride whose filter (8 entries per 64-bit block) still fits in
    // L2 and whose q-grams keep 14 bases (S=8, q=14 on 500 k 21-mers: 3.35 ms against 3.85 ms
    // at S=4, tools/c5_stride8.sh); failing that, the largest stride with q >= 16.
    *q = std::min<uint32_t>(32, lmin);
    *S = 1;
    bool chosen = false;
    for (uint32_t s : {16u, 8u, 4u, 2u}) {
        if (s > lmin) continue;
        const uint32_t qq = std::min<uint32_t>(32, lmin - s + 1);
        if (forced ? (int)s != forced : (qq < 14 || n_pat * s > (8ull << 19))) continue;
        *q = qq;
        *S = s;
        chosen = true;
        break;
    }
    if (!chosen && !forced)
        for (uint32_t s : {16u, 8u, 4u, 2u}) {
            if (s > lmin) continue;
            const uint32_t qq = std::min<uint32_t>(32, lmin - s + 1);
            if (qq < 16 || n_pat * s * 4 > (1ull << 25)) continue;
            *q = qq;
            *S = s;
            break;
        }
    // Filter size: measured with 500 k 21-mers (2 M entries, profiles/r01_gbloom_sweep.txt) a
    // 2 MiB image (8 entries per 64-bit block, 1.2 % of samples pass) is fastest because it
    // stays resident in the 4 MiB XCD L2; 8 MiB (0.09 % pass) is 1.8x slower.  ~8 entries/block.
    // r02 (profiles/r02_c5_filter_size.txt, tools/probes/stream_policy.hip): next to the text stream the
    // L2 serves ~208 G random reads/s from a table of up to 3 MiB, 176 G/s at 4 MiB, 105 G/s at 8 MiB;
    // 8 bits per entry, at least 1 MiB, at most 3 MiB.
    uint64_t blocks = std::min<uint64_t>(std::max<uint64_t>(n_pat * *S / 8, 1ull << 17), 3ull << 17);
    if (opt.gbloom_log2_blocks) blocks = 1ull << opt.gbloom_log2_blocks;  // tuning hooks
    if (opt.gbloom_kib) blocks = (uint64_t)opt.gbloom_kib * 128;
    *gblocks = (uint32_t)blocks;
}

// ---- length classes ---------------------------------------------------------------------------
// One geometry for the whole set is dictated by its SHORTEST pattern: 10 000 31-mers next to one
// 8-mer scan at S=1, q=8 -- 71.7 ms per 15 GB instead of 2.4 (profiles/r04_mixed_sets.txt), where
// the reference's DFA (src/cmd_extract.rs:260-265) scans any list at one speed.  The set is
// therefore split by length where that pays: patterns shorter than `split` form a SHORT class with
// its own stride S2 and q-grams of q2 <= 8 bases looked up in a plain table in LDS (filter.hpp,
// scan_kernel_impl.hpp: MC kernels), the others keep the hashed filter with the geometry THEIR
// shortest pattern admits.  The split is chosen with a cost model fitted to the stride sweep of
// the headline set and to the two-class kernel at several short-class geometries (ms per 15 GB
// launch on an MI355X, profiles/r04_mixed_sets.txt): 1.13 + 0.42 per hashed sample per 16 bases
// + 0.075 / 0.15 per byte- / bit-table sample; a million candidates cost 0.0152 in a one-class
// kernel and 0.025 in a two-class kernel (its hand-off serves both classes).
struct ClassPlan {
    uint32_t split = 0;  // 0 = one class; else patterns shorter than this are the short class
    uint32_t S2 = 0, q2 = 0;
    uint32_t lmin_main = 0;
    uint64_t n_main = 0, n_short = 0;
    double cost = 0;
};
static double main_cost(uint32_t lmin, uint64_t n, const mk_matcher_options &opt, bool two_class) {
    uint32_t q, S, gb;
    choose_geometry(lmin, n, opt, &q, &S, &gb);
    return hashed_cost(n, S, q, two_class ? 0.025 : 0.0152);
}
static double short_cost(uint32_t S2, uint32_t q2, uint64_t n) {
    const double per_sample = std::min(1.0, (double)n * S2 / pow(4.0, (double)q2));
    return (q2 <= kShortByteMaxQ ? 0.075 : 0.15) * (16.0 / S2) + kMCandPerLaunch * 0.025 * per_sample / S2;
}
// lens: pattern lengths (any order).  Only sets whose main class fits the LDS filter are split.
static ClassPlan plan_classes(std::vector<uint32_t> lens, const mk_matcher_options &opt) {
    ClassPlan best;
    const uint64_t n = lens.size();
    const auto mm = std::minmax_element(lens.begin(), lens.end());
    best.n_main = n;
    best.lmin_main = *mm.first;
    // (k-mer lists -- one length -- leave here without sorting half a million lengths)
    if (opt.length_classes == 1 || *mm.first == *mm.second) return best;
    if (opt.force_global_filter || n > kMaxLdsEntries) {
        // Main filter in global memory (hundreds of thousands of patterns): its kernels need 14-base q-grams at a stride
        // of 2 or more, i.e. patterns of 15 bases; ONE shorter pattern sends the whole set to S = 1 with its own length
        // as q.  Patterns below 15 bases therefore form the short class whenever the rest can keep a real stride.
        constexpr uint32_t kGfMinLen = 15;
        if (*mm.first >= kGfMinLen || *mm.second < kGfMinLen || (opt.force_stride && opt.length_classes != 2 && !opt.force_split_len)) return best;
        uint32_t split = 0xFFFFFFFFu;
        uint64_t n_short = 0;
        for (uint32_t l : lens) {
            if (l < kGfMinLen)
                ++n_short;
            else
                split = std::min(split, l);
        }
        if (opt.force_split_len) {
            split = opt.force_split_len;
            n_short = 0;
            for (uint32_t l : lens) n_short += l < split;
            if (n_short == 0 || n_short == n) return best;
        }
        uint32_t q, S, gb;
        choose_geometry(split, n - n_short, opt, &q, &S, &gb);
        if (S < 2) return best;
        for (uint32_t s2 : {8u, 4u, 2u, 1u}) {
            if (s2 > *mm.first || (opt.force_stride2 && s2 != opt.force_stride2)) continue;
            const uint32_t qmax = std::min<uint32_t>(kShortMaxQ, *mm.first - s2 + 1);
            for (uint32_t q2 : {qmax, std::min(qmax, kShortByteMaxQ)}) {
                if (opt.force_q2 && q2 != std::min<uint32_t>(opt.force_q2, qmax)) continue;
                const double c = short_cost(s2, q2, n_short);
                if (best.split == 0 || c < best.cost) {
                    best.split = split;
                    best.S2 = s2;
                    best.q2 = q2;
                    best.lmin_main = split;
                    best.n_main = n - n_short;
                    best.n_short = n_short;
                    best.cost = c;
                }
            }
        }
        return best;
    }
    std::sort(lens.begin(), lens.end());
    if (opt.force_stride && opt.length_classes != 2 && !opt.force_split_len) return best;  // a forced stride means the whole set
    best.cost = main_cost(lens[0], n, opt, false);
    const double single = best.cost;
    for (uint64_t i = 1; i < n; ++i) {  // short class = lens[0 .. i), split at every distinct length
        if (lens[i] == lens[i - 1]) continue;
        const uint32_t split = lens[i];
        if (opt.force_split_len && split != opt.force_split_len) continue;
        if (lens[0] > 64 && !opt.force_split_len) break;  // (such a set is sampled at S=16 as one class anyway)
        {  // a main class that itself needs stride 1 is not split (no such kernel: 16 + 16 samples per lane spill)
            uint32_t q, S, gb;
            choose_geometry(split, n - i, opt, &q, &S, &gb);
            if (S < 2) continue;
        }
        const double mc = main_cost(split, n - i, opt, true);
        for (uint32_t s2 : {8u, 4u, 2u, 1u}) {
            if (s2 > lens[0] || (opt.force_stride2 && s2 != opt.force_stride2)) continue;
            const uint32_t qmax = std::min<uint32_t>(kShortMaxQ, lens[0] - s2 + 1);
            for (uint32_t q2 : {qmax, std::min(qmax, kShortByteMaxQ)}) {
                if (opt.force_q2 && q2 != std::min<uint32_t>(opt.force_q2, qmax)) continue;
                if (!opt.force_stride2 && !opt.force_q2 && s2 > 1 && (double)i * s2 > 2.0 * pow(4.0, (double)q2)) continue;  // bucket chains
                const double c = mc + short_cost(s2, q2, i);
                const bool forced = opt.length_classes == 2 || opt.force_split_len;
                // a split must pay for the second code path: 10 % below the single class
                if ((best.split == 0 && (forced || c < 0.9 * single)) || (best.split != 0 && c < best.cost)) {
                    best.split = split;
                    best.S2 = s2;
                    best.q2 = q2;
                    best.lmin_main = split;
                    best.n_main = n - i;
                    best.n_short = i;
                    best.cost = c;
                }
            }
        }
    }
    return best;
}

}  // namespace mk

using namespace mk;

thread_local double mk::g_free_ms = 0;  // ... of it, in hipFree
thread_local double mk::g_alloc_ms = 0;  // time the calling thread has spent growing device buffers (diagnostic: mk_bam_window::ms[7])

int mk::ensure_device(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return MK_OK;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double free_ms = 0;
    if (*p) {
        (void)hipFree(*p);  // (waits for the device: every stream's kernels, another handle's included)
        free_ms = ms_since(t0);
        g_free_ms += free_ms;
    }
    *p = nullptr;
    *cap = 0;
    // (room to grow without another allocation: a quarter for small buffers, a sixteenth from 64 MiB on)
    size_t want = need + (need < (64u << 20) ? need / 4 : need / 16) + 4096;
    hipError_t e = hipMalloc(p, want);
    g_alloc_ms += ms_since(t0);
#ifdef MK_ALLOC_LOG  // (tagged diagnostic build, tools/tag_alloc_log.sh: which buffer growths cost what)
    fprintf(stderr, "[alloc] %8.1f MiB: hipFree %7.2f ms, hipMalloc %7.2f ms\n", want / 1048576.0, free_ms, ms_since(t0) - free_ms);
#endif
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    *cap = want;
    return MK_OK;
}

static int ensure(void **p, size_t *cap, size_t need) { return ensure_device(p, cap, need); }

// options of mk_matcher_create_ex / mk_plan_geometry -> a complete, validated struct (NULL = defaults)
static int read_options(const mk_matcher_options *options, mk_matcher_options *out) {
    mk_matcher_options opt;
    memset(&opt, 0, sizeof(opt));
    if (options) {  // a caller built against an older, shorter struct passes its own size
        if (options->struct_size < sizeof(uint32_t) || options->struct_size > 4096)
            return fail(MK_E_INVALID_ARG, "mk_matcher_options.struct_size %u is not plausible", options->struct_size);
        memcpy(&opt, options, std::min<size_t>(options->struct_size, sizeof(opt)));
        const uint32_t fs = opt.force_stride;
        if (fs != 0 && fs != 1 && fs != 2 && fs != 4 && fs != 8 && fs != 16)
            return fail(MK_E_INVALID_ARG, "force_stride %u: must be 0, 1, 2, 4, 8 or 16", fs);
        if (opt.gbloom_log2_blocks != 0 && (opt.gbloom_log2_blocks < 10 || opt.gbloom_log2_blocks > 25))
            return fail(MK_E_INVALID_ARG, "gbloom_log2_blocks %u out of range (10..25)", opt.gbloom_log2_blocks);
        if (opt.tile_run > 8) return fail(MK_E_INVALID_ARG, "tile_run %u out of range (0..8)", opt.tile_run);
        if (opt.gbloom_kib > (1u << 18)) return fail(MK_E_INVALID_ARG, "gbloom_kib %u out of range (<= 256 MiB)", opt.gbloom_kib);
        if (opt.length_classes > 2) return fail(MK_E_INVALID_ARG, "length_classes %u: must be 0 (rule), 1 or 2", opt.length_classes);
        const uint32_t f2 = opt.force_stride2;
        if (f2 != 0 && f2 != 1 && f2 != 2 && f2 != 4 && f2 != 8)
            return fail(MK_E_INVALID_ARG, "force_stride2 %u: must be 0, 1, 2, 4 or 8", f2);
        if (opt.force_q2 > kShortMaxQ) return fail(MK_E_INVALID_ARG, "force_q2 %u: must be 0..%u", opt.force_q2, kShortMaxQ);
    }
    *out = opt;
    return MK_OK;
}

extern "C" {

int mk_abi_version(void) { return MK_ABI_VERSION; }
const char *mk_last_error(void) { return g_last_error; }

int mk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mk_free(void *p) { free(p); }

int mk_plan_geometry(const uint32_t *pat_len, uint32_t n_pat, const mk_matcher_options *options, uint32_t *q_gram, uint32_t *stride,
                     uint32_t *in_lds, uint32_t *split_len, uint32_t *n_short, uint32_t *q_gram2, uint32_t *stride2) {
    if (!pat_len || n_pat == 0) return fail(MK_E_NO_PATTERNS, "No k-mers found in file or provided sequence.");
    mk_matcher_options opt;
    int rc = read_options(options, &opt);
    if (rc) return rc;
    MK_ABI_BEGIN
    std::vector<uint32_t> lens(pat_len, pat_len + n_pat);
    if (*std::min_element(lens.begin(), lens.end()) == 0) return fail(MK_E_EMPTY_PATTERN, "Pattern is empty.");
    const ClassPlan plan = plan_classes(std::move(lens), opt);
    uint32_t q = 0, S = 1, gb = 0;
    choose_geometry(plan.lmin_main, plan.n_main, opt, &q, &S, &gb);
    if (q_gram) *q_gram = q;
    if (stride) *stride = S;
    if (in_lds) *in_lds = gb ? 0 : 1;
    if (split_len) *split_len = plan.split;
    if (n_short) *n_short = (uint32_t)plan.n_short;
    if (q_gram2) *q_gram2 = plan.q2;
    if (stride2) *stride2 = plan.S2;
    return MK_OK;
    MK_ABI_END
}

int mk_matcher_create(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat, uint32_t algo, uint32_t q,
                      uint32_t flags, int32_t device, mk_matcher **out) {
    return mk_matcher_create_ex(pat_bytes, pat_off, n_pat, algo, q, flags, device, nullptr, out);
}

int mk_matcher_create_ex(const uint8_t *pat_bytes, const uint32_t *pat_off, uint32_t n_pat, uint32_t algo, uint32_t q,
                         uint32_t flags, int32_t device, const mk_matcher_options *options, mk_matcher **out) {
    if (!out) return fail(MK_E_INVALID_ARG, "out is null");
    *out = nullptr;
    mk_matcher_options opt;
    int rc_opt = read_options(options, &opt);
    if (rc_opt) return rc_opt;
    MK_ABI_BEGIN
    if (n_pat == 0 || !pat_off || !pat_bytes) return fail(MK_E_NO_PATTERNS, "No k-mers found in file or provided sequence.");
    if (algo > MK_ALGO_BNDMQ) return fail(MK_E_INVALID_ARG, "unknown algo %u", algo);
    if (n_pat > kMaxPatterns) return fail(MK_E_UNSUPPORTED, "too many patterns (%u)", n_pat);
    const bool ci = flags & MK_FLAG_ASCII_CASE_INSENSITIVE;
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0;
    for (uint32_t i = 0; i < n_pat; ++i) {
        if (pat_off[i + 1] < pat_off[i]) return fail(MK_E_INVALID_ARG, "pattern offsets not monotone");
        uint32_t len = pat_off[i + 1] - pat_off[i];
        lmin = std::min(lmin, len);
        lmax = std::max(lmax, len);
    }
    // algorithm selection: src/cmd_extract.rs:166-171, src/helpers.rs:203-211
    uint32_t use = algo;
    if (ci)
        use = MK_ALGO_AC;
    else if (algo == MK_ALGO_AUTO)
        use = (q == 0 && mk_recommend_aho_corasick(n_pat, lmax)) ? MK_ALGO_AC : MK_ALGO_BNDMQ;
    // validation errors of BNDMq::new, in pattern order (src/cmd_extract.rs:267-276)
    if (use == MK_ALGO_BNDMQ) {
        for (uint32_t i = 0; i < n_pat; ++i) {
            size_t len = pat_off[i + 1] - pat_off[i];
            size_t qq = q ? q : mk_tune_q_value(len);
            if (!q && len >= 65) return fail(MK_E_PATTERN_TOO_LONG, "Pattern length is too long for BNDMq.");
            if (len == 0) return fail(MK_E_EMPTY_PATTERN, "Pattern is empty.");
            if (qq == 0 || qq > len)
                return fail(MK_E_INVALID_Q, "Invalid q-gram length: %zu. Must be between 1 and pattern length.", qq);
            if (len > 64)
                return fail(MK_E_PATTERN_TOO_LONG,
                            "Pattern length %zu is too large for this architecture when using BNDM (max 64).", len);
        }
    } else if (lmin == 0) {
        return fail(MK_E_EMPTY_PATTERN, "Pattern is empty.");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MK_E_HIP, "no HIP device available: the MI355X scan path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MK_E_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    MK_HIP(hipSetDevice(device));

    struct Deleter {
        void operator()(mk_matcher *p) const { mk_matcher_destroy(p); }
    };
    std::unique_ptr<mk_matcher, Deleter> owner(new mk_matcher());
    mk_matcher *m = owner.get();
    m->device = device;
    m->algo = use;
    m->flags = flags;
    m->n_pat = n_pat;
    m->pat_bytes.assign(pat_bytes, pat_bytes + pat_off[n_pat]);
    m->pat_off.assign(pat_off, pat_off + n_pat + 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        m->num_cus = prop.multiProcessorCount;

    m->uniform_len = (lmin == lmax) ? lmin : 0;
    m->tile_run = opt.tile_run;
    // ---- compile the pattern set: length classes, Bloom filter + exact table
    ClassPlan plan;
    {
        std::vector<uint32_t> lens(n_pat);
        for (uint32_t i = 0; i < n_pat; ++i) lens[i] = pat_off[i + 1] - pat_off[i];
        plan = plan_classes(std::move(lens), opt);
    }
    if ((opt.length_classes == 2 || opt.force_split_len) && !plan.split)
        return fail(MK_E_INVALID_ARG, "the pattern set cannot be split into two length classes as the options ask "
                                      "(split length %u, stride %u, shortest pattern %u, longest %u)",
                    opt.force_split_len, opt.force_stride2, lmin, lmax);
    m->split_len = plan.split;
    m->S2 = plan.S2;
    m->q2 = plan.q2;
    m->n_short = (uint32_t)plan.n_short;
    choose_geometry(plan.lmin_main, plan.n_main, opt, &m->q, &m->S, &m->gbloom_blocks);
    if (opt.force_stride && m->S != opt.force_stride)
        return fail(MK_E_INVALID_ARG, "force_stride %u is longer than the shortest pattern (%u)", opt.force_stride, plan.lmin_main);
    const uint32_t q_f = m->q, S = m->S;
    m->entries = plan.n_main * S + plan.n_short * plan.S2;
    // load <= 0.5 while the table shares L2 with the text stream; a global-filter set's table is
    // HBM-resident anyway, and at load <= 0.25 a lookup all but never has to walk to a second bucket
    uint64_t slots = 64;
    while (slots < (m->gbloom_blocks ? 4 : 2) * m->entries) slots <<= 1;
    if (slots > (1ull << 27) && m->gbloom_blocks) {  // the largest sets keep load <= 0.5 rather than being refused
        slots = 64;
        while (slots < 2 * m->entries) slots <<= 1;
    }
    if (slots > (1ull << 27))  // bucket index has 26 bits: at most 2^26 table entries (patterns x stride)
        return fail(MK_E_UNSUPPORTED, "pattern set too large (%llu table entries)", (unsigned long long)m->entries);
    m->table_slots = (uint32_t)slots;
    const bool two = plan.split != 0;
    // (two classes next to a global filter run the runtime-q kernels, which carry no context fingerprints)
    const bool gf_ctx = m->gbloom_blocks != 0 && gf_has_ctx(S, q_f) && !two;
    const size_t bloom_words = m->gbloom_blocks ? (size_t)m->gbloom_blocks * 2 : (size_t)kBloomWords;
    // (a failing MK_HIP returns; `owner` then releases whatever was allocated so far)
    MK_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    MK_HIP(hipMalloc((void **)&m->d_bloom, bloom_words * sizeof(uint32_t)));
    MK_HIP(hipMalloc((void **)&m->d_table, slots * sizeof(TableEntry)));
    MK_HIP(hipMalloc((void **)&m->d_pat_bytes, m->pat_bytes.size() + 16));
    MK_HIP(hipMalloc((void **)&m->d_pat_off, (n_pat + 1) * sizeof(uint32_t)));
    MK_HIP(hipMalloc((void **)&m->d_nhits, sizeof(unsigned long long)));
    MK_HIP(hipMalloc((void **)&m->d_error, 16));
    MK_HIP(hipMemset(m->d_error, 0, 16));
    // staging of verified occurrences for the hit-tuple kernels, kHitStage tuples per scan wave.
    // Allocated here, not by the first MK_MODE_HITS scan: mk_scan_device only enqueues.
    MK_HIP(hipMalloc((void **)&m->d_stage, (size_t)m->num_cus * (kBlockThreads / 64) * kHitStage * sizeof(mk_hit)));
    MK_HIP(hipMalloc((void **)&m->d_flag_list, (size_t)m->num_cus * (kBlockThreads / 64) * kFlagListCap * sizeof(uint32_t)));
    MK_HIP(hipMalloc((void **)&m->d_flag_counts, (size_t)m->num_cus * (kBlockThreads / 64) * sizeof(uint32_t)));
    MK_HIP(hipMemcpy(m->d_pat_bytes, m->pat_bytes.data(), m->pat_bytes.size(), hipMemcpyHostToDevice));
    MK_HIP(hipMemsetAsync(m->d_pat_bytes + m->pat_bytes.size(), 0, 16, m->stream));  // (level 3 may read up to 15 bytes past a pattern)
    MK_HIP(hipMemcpy(m->d_pat_off, m->pat_off.data(), (n_pat + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    // filter images + exact table: built on the device from the patterns just uploaded (build_tables.hip; r04 --
    // one host thread took 0.43 s for the 4 M entries of 500 k 21-mers, profiles/r02_compile_time.txt)
    MK_HIP(hipMemsetAsync(m->d_bloom, 0, bloom_words * sizeof(uint32_t), m->stream));
    if (two) {
        MK_HIP(hipMalloc((void **)&m->d_short_table, kShortBitmapWords * sizeof(uint32_t)));
        MK_HIP(hipMemsetAsync(m->d_short_table, 0, kShortBitmapWords * sizeof(uint32_t), m->stream));
    }
    {
        BuildParams B;
        memset(&B, 0, sizeof(B));
        B.pat_bytes = m->d_pat_bytes;
        B.pat_off = m->d_pat_off;
        B.n_pat = n_pat;
        B.S = S;
        B.q = q_f;
        B.split = plan.split;
        B.S2 = plan.S2;
        B.q2 = plan.q2;
        B.gbloom_blocks = m->gbloom_blocks;
        B.gf_ctx = gf_ctx ? 1 : 0;
        B.bloom = m->d_bloom;
        B.table = m->d_table;
        B.bucket_mask = m->table_slots / kBucketEntries - 1;
        B.short_table = m->d_short_table;
        launch_build_tables(B, slots, m->stream);
        MK_HIP(hipGetLastError());
        MK_HIP(hipStreamSynchronize(m->stream));
    }
    if (use == MK_ALGO_AC && !m->uniform_len) {
        // matches that end on one byte are emitted longest first, then by pattern id (aho-corasick's overlapping
        // DFA walk, src/cmd_extract.rs:332-351): that tie order as a rank the device sort can use as a key field
        std::vector<uint32_t> unrank(n_pat), rank(n_pat);
        if (lmax <= (1u << 20)) {  // stable counting sort on the length, longest first
            std::vector<uint32_t> start((size_t)lmax + 2, 0);
            for (uint32_t i = 0; i < n_pat; ++i) start[lmax - (pat_off[i + 1] - pat_off[i]) + 1]++;
            for (uint32_t l = 0; l <= lmax; ++l) start[l + 1] += start[l];
            for (uint32_t i = 0; i < n_pat; ++i) unrank[start[lmax - (pat_off[i + 1] - pat_off[i])]++] = i;
        } else {
            for (uint32_t i = 0; i < n_pat; ++i) unrank[i] = i;
            std::stable_sort(unrank.begin(), unrank.end(), [pat_off](uint32_t a, uint32_t b) {
                return pat_off[a + 1] - pat_off[a] > pat_off[b + 1] - pat_off[b];
            });
        }
        for (uint32_t r = 0; r < n_pat; ++r) rank[unrank[r]] = r;
        MK_HIP(hipMalloc((void **)&m->d_pat_rank, n_pat * sizeof(uint32_t)));
        MK_HIP(hipMalloc((void **)&m->d_pat_unrank, n_pat * sizeof(uint32_t)));
        MK_HIP(hipMemcpy(m->d_pat_rank, rank.data(), n_pat * sizeof(uint32_t), hipMemcpyHostToDevice));
        MK_HIP(hipMemcpy(m->d_pat_unrank, unrank.data(), n_pat * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    *out = owner.release();
    return MK_OK;
    MK_ABI_END
}

void mk_matcher_destroy(mk_matcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->comm) (void)mk_comm_destroy(m);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->stream_ahead) {
        (void)hipStreamSynchronize(m->stream_ahead);
        (void)hipStreamDestroy(m->stream_ahead);
        for (auto &a : m->ahead)
            if (a.ev) (void)hipEventDestroy(a.ev);
    }
    for (auto e : m->ev_start) (void)hipEventDestroy(e);
    for (auto e : m->ev_stop) (void)hipEventDestroy(e);
    for (void *p : {(void *)m->d_bloom, (void *)m->d_table, (void *)m->d_pat_bytes, (void *)m->d_pat_off,
                    (void *)m->d_seq, (void *)m->d_off, (void *)m->d_flags, (void *)m->d_hits, (void *)m->d_nhits,
                    (void *)m->d_stage, (void *)m->d_rec_index, (void *)m->d_flag_list, (void *)m->d_flag_counts, m->d_sort_tmp,
                    (void *)m->d_pat_rank, (void *)m->d_pat_unrank, (void *)m->d_error, m->d_aux, m->d_pair, (void *)m->d_short_table, m->txt[0].d_text, m->txt[0].d_ing_a, m->txt[0].d_ing_b, m->txt[1].d_text, m->txt[1].d_ing_a, m->txt[1].d_ing_b, m->txt[0].d_fa_seq, m->txt[1].d_fa_seq, (void *)m->d_flags2, m->ahead[0].d, m->ahead[1].d, m->ahead[2].d, m->ahead[3].d})
        if (p) (void)hipFree(p);
    delete m;
}

uint32_t mk_matcher_algo(const mk_matcher *m) { return m ? m->algo : 0; }
uint32_t mk_matcher_num_patterns(const mk_matcher *m) { return m ? m->n_pat : 0; }
const char *mk_matcher_kernel_name(const mk_matcher *m) { return m ? m->kernel_name : ""; }

int mk_matcher_filter_info(const mk_matcher *m, uint32_t *q_gram, uint32_t *stride, uint64_t *entries,
                           uint64_t *table_bytes) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (q_gram) *q_gram = m->q;
    if (stride) *stride = m->S;
    if (entries) *entries = m->entries;
    if (table_bytes) *table_bytes = (uint64_t)m->table_slots * sizeof(TableEntry);
    return MK_OK;
}

int mk_matcher_class_info(const mk_matcher *m, uint32_t *split_len, uint32_t *n_short, uint32_t *q_gram2, uint32_t *stride2) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (split_len) *split_len = m->split_len;
    if (n_short) *n_short = m->n_short;
    if (q_gram2) *q_gram2 = m->q2;
    if (stride2) *stride2 = m->S2;
    return MK_OK;
}

int mk_matcher_filter_mode(const mk_matcher *m, uint32_t *in_lds, uint64_t *filter_bytes) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (in_lds) *in_lds = m->gbloom_blocks ? 0 : 1;
    if (filter_bytes) *filter_bytes = m->gbloom_blocks ? (uint64_t)m->gbloom_blocks * 8 : (uint64_t)kBloomBytes;
    return MK_OK;
}

int mk_scan_device(mk_matcher *m, const void *d_seq, uint64_t n_bytes, const void *d_seq_off, uint64_t n_rec,
                   uint32_t mode, void *d_rec_flags, void *d_hits, uint64_t hits_cap, void *d_n_hits,
                   void *d_counters, void *stream) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (mode > MK_MODE_HITS) return fail(MK_E_INVALID_ARG, "unknown mode %u", mode);
    if (((uintptr_t)d_seq & 15) != 0) return fail(MK_E_INVALID_ARG, "d_seq must be 16-byte aligned");
    if (((uintptr_t)d_rec_flags & 3) != 0) return fail(MK_E_INVALID_ARG, "d_rec_flags must be 4-byte aligned");
    // records of one length (mk_matcher_set_fixed_record_length, or mk_scan_batch's own check of its offsets):
    // the offsets array is never read
    const uint32_t rec_len = m->batch_rec_len ? m->batch_rec_len : m->fixed_rec_len;
    if (rec_len && n_bytes != n_rec * (uint64_t)rec_len)
        return fail(MK_E_INVALID_ARG, "fixed record length %u: %llu records are not %llu bytes", rec_len, (unsigned long long)n_rec,
                    (unsigned long long)n_bytes);
    if (!d_n_hits || !d_rec_flags || (!d_seq_off && n_rec && !rec_len)) return fail(MK_E_INVALID_ARG, "null device buffer");
    if (mode == MK_MODE_HITS && !d_hits && hits_cap) return fail(MK_E_INVALID_ARG, "d_hits is null");
    hipStream_t st = (hipStream_t)stream;
    MK_HIP(hipSetDevice(m->device));
    launch_clear((uint32_t *)d_rec_flags, (n_rec + 3) / 4, (unsigned long long *)d_n_hits, st);  // also clears *d_n_hits
    MK_HIP(hipGetLastError());
    // the sticky error word belongs to the handle's LAST tuple scan: an earlier batch's condition must not make
    // mk_order_hits_device refuse the tuples of this one (it is reported by whichever check comes first)
    if (mode == MK_MODE_HITS) MK_HIP(hipMemsetAsync(m->d_error, 0, sizeof(uint32_t), st));
    m->last_n_rec = n_rec;
    if (n_rec == 0 || n_bytes == 0) return MK_OK;
    ScanParams p;
    memset(&p, 0, sizeof(p));
    p.seq = (const uint8_t *)d_seq;
    p.n_bytes = n_bytes;
    p.rec_off = (const uint64_t *)d_seq_off;
    p.n_rec = n_rec;
    const uint64_t tile_bytes = (uint64_t)kTileChunks * kChunkBytes;
    const uint64_t n_tiles = (n_bytes + tile_bytes - 1) / tile_bytes;
    p.bloom = m->d_bloom;
    p.gbloom_blocks = m->gbloom_blocks;
    p.table = m->d_table;
    p.table_mask = m->table_slots / kBucketEntries - 1;  // bucket mask
    p.pat_bytes = m->d_pat_bytes;
    p.pat_off = m->d_pat_off;
    p.n_pat = m->n_pat;
    p.q = m->q;
    const uint64_t kmask = m->q >= 32 ? ~0ull : ((1ull << (2 * m->q)) - 1);
    p.key_mask_lo = (uint32_t)kmask;
    p.key_mask_hi = (uint32_t)(kmask >> 32);
    p.case_insensitive = (m->flags & MK_FLAG_ASCII_CASE_INSENSITIVE) ? 1 : 0;
    p.s2 = m->split_len ? m->S2 : 0;
    p.s2_log2 = m->S2 >= 8 ? 3 : m->S2 >= 4 ? 2 : m->S2 >= 2 ? 1 : 0;
    p.key2_mask = (1u << (2 * m->q2)) - 1u;
    p.short_bytes = m->q2 <= kShortByteMaxQ;
    p.short_bitmap = m->d_short_table;
    p.uniform_len = m->uniform_len;
    p.rec_flags32 = (uint32_t *)d_rec_flags;
    p.hits = (mk_hit *)d_hits;
    p.hits_cap = (mode == MK_MODE_HITS) ? hits_cap : 0;
    p.n_hits = (unsigned long long *)d_n_hits;
    p.counters = (unsigned long long *)d_counters;
    p.error_word = m->d_error;
    const uint64_t waves_per_block = kBlockThreads / 64;
    uint64_t blocks = (n_tiles + waves_per_block - 1) / waves_per_block;
    if (blocks > (uint64_t)m->num_cus) blocks = m->num_cus;
    p.stage = m->d_stage;
    p.rec_per_byte = (double)n_rec / (double)n_bytes;
    // tiles a wave scans back to back before it jumps ahead by n_waves * run tiles: runs of 4 keep a wave
    // inside one 2 MiB page for 124 KiB (-4 % on 15 GB batches, profiles/r02_tile_run.txt); small batches
    // keep runs short so that every wave still gets tiles
    const uint64_t tiles_per_wave = n_tiles / (blocks * waves_per_block);
    p.tile_run = m->tile_run ? m->tile_run : tiles_per_wave >= 16 ? 4 : tiles_per_wave >= 8 ? 2 : 1;
    p.rec_index = nullptr;
    p.rec_len = rec_len;
    p.inv_rec_len = rec_len ? 1.0 / (double)rec_len : 0.0;
    if (m->ragged && !rec_len && n_rec < (1ull << 32)) {  // coarse record index: one entry per 64 KiB of text, built per scan
        int rc_i = ensure((void **)&m->d_rec_index, &m->d_rec_index_cap, ((n_bytes >> kRecIndexShift) + 2) * sizeof(uint32_t));
        if (rc_i) return rc_i;
        launch_rec_index(p.rec_off, n_rec, n_bytes, m->d_rec_index, st);
        p.rec_index = m->d_rec_index;
    }
    const size_t slots = m->ev_start.size();
    const size_t slot = slots ? (size_t)(m->timed_launches % slots) : 0;
    if (slots) MK_HIP(hipEventRecord(m->ev_start[slot], st));
    // Two kernel flavours (scan_kernel_impl.hpp), picked from the hit density the matcher has last seen.
    // Hit-dense text (tag on already extracted reads: every other record hits): level 3 re-reads each
    // verified window, so the stream is read with cacheable loads and the re-read finds it in L2 / MALL,
    // level 3 itself uses 16-byte loads, and flags are stored directly.  Every read hitting: 9.1 -> 6.0 ms
    // per 15 GB.  Everything else: non-temporal stream (16 % faster without hits), 8-byte compare loads,
    // flagged records listed per wave and their bytes set by a small kernel afterwards.  The flavours cross
    // at 10 % (tuples) / 14 % (flags only) of the records (profiles/r03_crossover.txt; r02: 12 %).
    // r03 (dense flavour probing / resolving early, profiles/r03_crossover.txt): flags only 1 in 8 records sparse 3.24 vs
    // dense 3.32 ms, 1 in 6 3.68 vs 3.44; with tuples 1 in 12 3.29 vs 3.39, 1 in 10 3.49 vs 3.43
    const uint32_t kDensePerMille = mode == MK_MODE_HITS ? 95 : 145;
    const bool plain_loads = m->hit_density_pm >= kDensePerMille && !m->split_len;  // (two-class kernels: one flavour)
    // (r03: a third flavour -- the sparse kernel with 16-byte compare loads as its own instantiation, for 2-12 % of
    // the records hitting -- gained nothing at any density: profiles/r03_cmp16_mid.txt)
    const int flavour = plain_loads ? 0 : 1;
    // the flag-only kernels for sparse hits list the records they flag (one list per scan wave) and a small
    // kernel sets the flag bytes afterwards (scan_kernel_impl.hpp: drain_hits); record indices in the lists are
    // 32 bits.  (Their tuple-emitting twins set the flags from the tuples they stage.)
    const bool listed = !(plain_loads && m->gbloom_blocks == 0) && mode != MK_MODE_HITS && n_rec < (1ull << 32);
    p.flag_list = listed ? m->d_flag_list : nullptr;
    p.flag_counts = m->d_flag_counts;
    p.flag_cap = kFlagListCap;
    const char *name = launch_scan(p, (int)m->S, m->q > 16, mode == MK_MODE_HITS, m->gbloom_blocks != 0, flavour, (int)blocks, st);
    if (!name) return fail(MK_E_UNSUPPORTED, "no kernel for stride %u", m->S);
    if (slots) {
        MK_HIP(hipEventRecord(m->ev_stop[slot], st));
        m->timed_launches++;
    }
    if (listed)
        launch_flag_scatter(m->d_flag_list, m->d_flag_counts, kFlagListCap, (uint32_t)(blocks * waves_per_block), (uint8_t *)d_rec_flags, st);
    if (d_counters) {
        launch_count_flags(p, st);
        if (mode == MK_MODE_HITS && p.hits && p.hits_cap) launch_hist_hits(p, m->num_cus, st);
    }
    m->kernel_name = name;
    m->last_grid = (int)blocks;
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// reads and clears the handle's sticky device error word (after the work on `st` has completed)
static int take_device_errors(mk_matcher *m, hipStream_t st) {
    uint32_t w = 0;
    MK_HIP(hipMemcpyAsync(&w, m->d_error, sizeof(w), hipMemcpyDeviceToHost, st));
    MK_HIP(hipStreamSynchronize(st));
    if (!w) return MK_OK;
    MK_HIP(hipMemsetAsync(m->d_error, 0, sizeof(w), st));
    return fail(MK_E_UNSUPPORTED, "an occurrence lies 4 GiB or more into its record: a single record must be shorter than 4 GiB "
                                  "(mk_hit.pos is 32 bits); the tuples of that scan are not usable");
}

int mk_matcher_check_device(mk_matcher *m, void *stream) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    MK_HIP(hipSetDevice(m->device));
    return take_device_errors(m, (hipStream_t)stream);
}

int mk_matcher_hint_hit_density(mk_matcher *m, uint32_t records_hit_per_1000) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    m->hit_density_pm = records_hit_per_1000 > 1000 ? 1000 : records_hit_per_1000;
    return MK_OK;
}

int mk_matcher_set_fixed_record_length(mk_matcher *m, uint32_t record_length) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    m->fixed_rec_len = record_length;
    return MK_OK;
}

int mk_matcher_hint_record_lengths(mk_matcher *m, int equal_lengths) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    m->ragged = !equal_lengths;
    return MK_OK;
}

int mk_matcher_enable_timing(mk_matcher *m, uint32_t slots) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    MK_ABI_BEGIN
    MK_HIP(hipSetDevice(m->device));
    for (auto e : m->ev_start) (void)hipEventDestroy(e);
    for (auto e : m->ev_stop) (void)hipEventDestroy(e);
    m->ev_start.assign(slots, nullptr);
    m->ev_stop.assign(slots, nullptr);
    m->timed_launches = 0;
    for (uint32_t i = 0; i < slots; ++i) {
        MK_HIP(hipEventCreate(&m->ev_start[i]));
        MK_HIP(hipEventCreate(&m->ev_stop[i]));
    }
    return MK_OK;
    MK_ABI_END
}

int mk_matcher_kernel_times(mk_matcher *m, float *ms, uint32_t cap, uint32_t *n_out) {
    if (!m || !n_out) return fail(MK_E_INVALID_ARG, "null argument");
    const uint64_t slots = m->ev_start.size();
    const uint64_t n = std::min<uint64_t>(m->timed_launches, slots);
    *n_out = (uint32_t)n;
    if (n > cap) return fail(MK_E_CAPACITY, "need room for %llu timings", (unsigned long long)n);
    MK_HIP(hipSetDevice(m->device));
    // oldest retained launch first
    const uint64_t first = m->timed_launches - n;
    for (uint64_t i = 0; i < n; ++i) {
        const size_t slot = (size_t)((first + i) % slots);
        MK_HIP(hipEventSynchronize(m->ev_stop[slot]));
        MK_HIP(hipEventElapsedTime(&ms[i], m->ev_start[slot], m->ev_stop[slot]));
    }
    m->timed_launches = 0;
    return MK_OK;
}

int mk_matcher_launch_info(const mk_matcher *m, uint32_t *grid_blocks, uint32_t *block_threads, uint32_t *lds_bytes) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (grid_blocks) *grid_blocks = (uint32_t)m->last_grid;
    if (block_threads) *block_threads = kBlockThreads;
    if (lds_bytes) *lds_bytes = scan_lds_bytes();
    return MK_OK;
}

// Tuples still on the device, sorted in place into the reference's emission order (order_hits.hip): histogram of
// the tuples over bins of consecutive records, one 32-byte read-back that fixes the key layout, scatter into the
// bins as 8-byte keys, one LDS sort per bin.  The scratch buffer lives in the handle and growing it synchronises
// the device, like every other workspace of the handle.
static uint32_t bits_of(uint64_t v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }

static int order_library(mk_matcher *m, mk_hit *d_hits, uint64_t n, bool ac, hipStream_t st) {
    size_t need = 0;
    MK_HIP(order_hits_library(d_hits, n, ac, m->d_pat_off, m->uniform_len, nullptr, &need, st));
    int rc = ensure(&m->d_sort_tmp, &m->d_sort_tmp_cap, need ? need : 16);
    if (rc) return rc;
    MK_HIP(order_hits_library(d_hits, n, ac, m->d_pat_off, m->uniform_len, m->d_sort_tmp, &need, st));
    m->order_path = 3;
    return MK_OK;
}

int mk_order_hits_device(mk_matcher *m, void *d_hits, uint64_t n_hits, void *stream) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    return mk::order_hits_on_device(m, d_hits, n_hits, m->algo == MK_ALGO_AC, stream);
}

}  // extern "C"

// ac: Aho-Corasick emission order; !ac: (record, pattern, position) -- BNDMq's emission order and, for any matcher,
// the order in which a record's distinct patterns are adjacent and ascending (sets.hip)
static int order_hits_on_device_impl(mk_matcher *m, void *d_hits, uint64_t n_hits, bool ac_order, void *stream);
int mk::order_hits_on_device(mk_matcher *m, void *d_hits, uint64_t n_hits, bool ac_order, void *stream) {
    const int rc = order_hits_on_device_impl(m, d_hits, n_hits, ac_order, stream);
    if (rc == MK_OK) ++m->order_path_calls[m->order_path & 3];  // (mk_matcher_order_stats: does the library sort ever fire on real data?)
    return rc;
}
static int order_hits_on_device_impl(mk_matcher *m, void *d_hits, uint64_t n_hits, bool ac_order, void *stream) {
    using namespace mk;
    m->order_path = 0;
    if (n_hits < 2) return MK_OK;
    if (!d_hits) return fail(MK_E_INVALID_ARG, "null buffer");
    if (((uintptr_t)d_hits & 15) != 0) return fail(MK_E_INVALID_ARG, "d_hits must be 16-byte aligned");
    MK_ABI_BEGIN
    MK_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    mk_hit *hits = (mk_hit *)d_hits;
    const uint64_t n = n_hits;
    if (n >= (1ull << 32)) return order_library(m, hits, n, ac_order, st);  // bin cursors are 32 bits
    if (!m->order_prepared) {  // once per handle (function attributes are per device)
        MK_HIP(order_kernels_prepare());
        m->order_prepared = true;
    }
    // bins: ~2048 tuples each, at most kOrderMaxBins, a power of two
    uint32_t log_bins = 0;
    while (log_bins < 15 && ((uint64_t)2048 << log_bins) < n) ++log_bins;
    // scratch: stats | counts | starts | cursors | keys
    const size_t head = 64 + (size_t)(4 * kOrderMaxBins + 16) * sizeof(uint32_t);
    int rc = ensure(&m->d_sort_tmp, &m->d_sort_tmp_cap, head + n * sizeof(uint64_t));
    if (rc) return rc;
    OrderScratch S;
    S.stats = (unsigned long long *)m->d_sort_tmp;
    S.g_cnt = (uint32_t *)((char *)m->d_sort_tmp + 64);
    S.bin_start = S.g_cnt + kOrderMaxBins;
    S.cursor = S.bin_start + kOrderMaxBins + 8;
    S.big_list = S.cursor + kOrderMaxBins;
    S.keys = (uint64_t *)((char *)m->d_sort_tmp + head);
    OrderKey L;
    memset(&L, 0, sizeof(L));
    L.pat_off = m->d_pat_off;
    L.uniform_len = m->uniform_len;
    L.rank = ac_order ? m->d_pat_rank : nullptr;
    L.unrank = ac_order ? m->d_pat_unrank : nullptr;
    L.ac = ac_order ? 1 : 0;
    // first attempt: bins of 2^s consecutive records, s from the record count of the handle's last scan
    uint64_t rec_bound = m->last_n_rec ? m->last_n_rec : (1ull << 32);
    unsigned long long stats[6] = {0, 0, 0, 0, 0, 0};
    uint32_t s = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        s = bits_of(rec_bound - 1) > log_bins ? bits_of(rec_bound - 1) - log_bins : 0;
        L.bits_a = 0;
        L.bits_b = 1;
        L.b_hi = 0;
        L.b_lo = 63;  // (histogram on the record alone: B >> b_lo must vanish whatever B is)
        L.shift = s;
        L.n_bins = 1u << log_bins;
        MK_HIP(hipMemsetAsync(m->d_sort_tmp, 0, 64 + (size_t)L.n_bins * sizeof(uint32_t), st));
        launch_order_hist(hits, n, L, S, m->num_cus, st);
        MK_HIP(hipGetLastError());
        MK_HIP(hipMemcpyAsync(stats, S.stats, sizeof(stats), hipMemcpyDeviceToHost, st));
        uint32_t err_word = 0;  // the scan that wrote these tuples may have met a record it cannot address: same round trip
        MK_HIP(hipMemcpyAsync(&err_word, m->d_error, sizeof(err_word), hipMemcpyDeviceToHost, st));
        MK_HIP(hipStreamSynchronize(st));
        if (err_word) return take_device_errors(m, st);
        // done unless a record lies beyond the bound (tuples of another batch than the handle's last scan), or a
        // bin overflows while the bound is at least twice the largest record seen (the bins are coarser than they
        // need be): once more with the exact bound
        const bool beyond = (stats[0] >> s) >= L.n_bins;
        const bool loose = stats[3] > kOrderLeafMax && bits_of(stats[0]) < bits_of(rec_bound - 1);
        if (!beyond && !loose) break;
        rec_bound = stats[0] + 1;
    }
    uint32_t bits_rec = bits_of(stats[0]);
    const uint32_t bits_a = std::max(1u, bits_of(stats[1])), bits_b = std::max(1u, bits_of(stats[2]));
    if (bits_rec + bits_a + bits_b > 64) return order_library(m, hits, n, ac_order, st);  // the triple does not fit one 64-bit key
    if (s + bits_a > 63) return order_library(m, hits, n, ac_order, st);  // (cannot happen below 2^31 records)
    L.bits_a = bits_a;
    L.bits_b = bits_b;
    L.b_hi = 0;
    L.b_lo = bits_b;
    m->order_path = 1;
    if (stats[3] > kOrderLeafMax || (stats[0] >> s) >= L.n_bins) {
        // a bin overflows its LDS sort (few huge records, or hits clustered in one stretch of the batch): bin on the
        // top bits of (record, A) instead, with as many bins as the histogram kernel can hold
        L.rec_base = ~stats[4];  // the smallest record: the bins span the records that occur, not [0, largest]
        bits_rec = bits_of(stats[0] - L.rec_base);
        // the bins are the top bits of the whole (record, A, B) triple: a bin whose keys share all but 14 bits cannot
        // overflow, whatever the distribution
        const uint32_t total = bits_rec + bits_a + bits_b;
        // ~256 tuples per bin on average (room for skew), and where the triple is short enough, so many bins that only
        // 14 bits stay in the key: such a bin cannot hold more than 2^14 distinct tuples
        const uint32_t lb = std::min(15u, std::max(log_bins + 3, total > 14 ? total - 14 : 0u));
        const uint32_t drop = total > lb ? total - lb : 0;  // low bits of the triple that stay in the key
        if (drop >= bits_b) {
            L.b_hi = 0;
            L.b_lo = bits_b;
            L.shift = drop - bits_b;
        } else {
            L.b_hi = bits_b - drop;
            L.b_lo = drop;
            L.shift = 0;
        }
        L.n_bins = 1u << std::min(lb, total);
        MK_HIP(hipMemsetAsync(m->d_sort_tmp, 0, 64 + (size_t)L.n_bins * sizeof(uint32_t), st));
        launch_order_hist(hits, n, L, S, m->num_cus, st);
        MK_HIP(hipGetLastError());
        MK_HIP(hipMemcpyAsync(stats, S.stats, sizeof(stats), hipMemcpyDeviceToHost, st));
        MK_HIP(hipStreamSynchronize(st));
        if (stats[3] > kOrderLeafMax) return order_library(m, hits, n, ac_order, st);
        m->order_path = 2;
    } else {
        L.shift = s + bits_a;  // ((rec << bits_a) | A) >> (s + bits_a) == rec >> s: the bins just counted
    }
    m->order_bins = L.n_bins;
    m->order_max_bin = (uint32_t)stats[3];
    launch_order_scatter_leaf(hits, n, L, S, (uint32_t)stats[3], (uint32_t)stats[5], m->num_cus, st);
    MK_HIP(hipGetLastError());
    return MK_OK;
    MK_ABI_END
}

extern "C" {

int mk_matcher_order_stats(const mk_matcher *m, uint64_t calls[4]) {
    if (!m || !calls) return fail(MK_E_INVALID_ARG, "null argument");
    for (int i = 0; i < 4; ++i) calls[i] = m->order_path_calls[i];
    return MK_OK;
}

int mk_matcher_order_info(const mk_matcher *m, uint32_t *path, uint32_t *n_bins, uint32_t *max_bin) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (path) *path = m->order_path;
    if (n_bins) *n_bins = m->order_bins;
    if (max_bin) *max_bin = m->order_max_bin;
    return MK_OK;
}

int mk_order_hits(const mk_matcher *m, mk_hit *hits, uint64_t n_hits) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (m->algo == MK_ALGO_AC) {
        const uint32_t *off = m->pat_off.data();
        // aho-corasick overlapping DFA search: end ascending; at one end the state's own
        // patterns (longest, i.e. smallest start) first, pattern id ascending within a state
        std::sort(hits, hits + n_hits, [off](const mk_hit &a, const mk_hit &b) {
            if (a.rec != b.rec) return a.rec < b.rec;
            uint64_t ea = (uint64_t)a.pos + (off[a.pat + 1] - off[a.pat]);
            uint64_t eb = (uint64_t)b.pos + (off[b.pat + 1] - off[b.pat]);
            if (ea != eb) return ea < eb;
            if (a.pos != b.pos) return a.pos < b.pos;
            return a.pat < b.pat;
        });
    } else {
        // BNDMq driver loop: pattern-major, positions ascending (src/cmd_extract.rs:365-384)
        std::sort(hits, hits + n_hits, [](const mk_hit &a, const mk_hit &b) {
            if (a.rec != b.rec) return a.rec < b.rec;
            if (a.pat != b.pat) return a.pat < b.pat;
            return a.pos < b.pos;
        });
    }
    return MK_OK;
}

}  // extern "C"

namespace mk {

// ---- host-buffer batches, in steps (mk_scan_batch and the driver loops of host_loops.cpp) ----------------------
// argument checks of a host batch; *n_bytes = bytes of the batch
int batch_check(const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint64_t *n_bytes) {
    if (!seq_off) return fail(MK_E_INVALID_ARG, "null buffer");
    *n_bytes = seq_off[n_rec] - seq_off[0];
    if (*n_bytes && !seq_bytes) return fail(MK_E_INVALID_ARG, "null sequence buffer");
    if (*n_bytes >= (1ull << 32))  // mk_hit.pos is 32 bits: refuse a record it cannot address instead of wrapping
        for (uint64_t i = 0; i < n_rec; ++i)
            if (seq_off[i + 1] - seq_off[i] >= (1ull << 32))
                return fail(MK_E_UNSUPPORTED, "record %llu is %llu bytes long: a single record must be shorter than 4 GiB",
                            (unsigned long long)i, (unsigned long long)(seq_off[i + 1] - seq_off[i]));
    return MK_OK;
}

// records and offsets -> the handle's device buffers (enqueued on its stream); *batch_len = the length all records
// share (0 = they differ): such a batch needs no record lookup at all, a ragged one gets a coarse record index
int batch_upload(mk_matcher *m, const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint64_t n_bytes, uint32_t *batch_len) {
    MK_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = ensure((void **)&m->d_seq, &m->d_seq_cap, n_bytes + 64))) return rc;
    if ((rc = ensure((void **)&m->d_off, &m->d_off_cap, (n_rec + 1) * sizeof(uint64_t)))) return rc;
    if ((rc = ensure((void **)&m->d_flags, &m->d_flags_cap, n_rec + 8))) return rc;
    const uint64_t base = seq_off[0];
    const uint64_t len0 = seq_off[1] - seq_off[0];
    bool equal = true;
    for (uint64_t i = 1; i < n_rec && equal; ++i) equal = seq_off[i + 1] - seq_off[i] == len0;
    m->ragged = !equal;
    *batch_len = (equal && len0 > 0 && len0 < (1ull << 32)) ? (uint32_t)len0 : 0;
    std::vector<uint64_t> rel;
    const uint64_t *off_src = seq_off;
    if (base != 0) {  // device offsets are relative to the first byte uploaded
        rel.resize(n_rec + 1);
        for (uint64_t i = 0; i <= n_rec; ++i) rel[i] = seq_off[i] - base;
        off_src = rel.data();
    }
    if (n_bytes) MK_HIP(hipMemcpyAsync(m->d_seq, seq_bytes + base, n_bytes, hipMemcpyHostToDevice, m->stream));
    MK_HIP(hipMemcpyAsync(m->d_off, off_src, (n_rec + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
    if (!rel.empty()) MK_HIP(hipStreamSynchronize(m->stream));  // `rel` must outlive the copy
    return MK_OK;
}

// scans the uploaded batch; tuples (MK_MODE_HITS) stay UNORDERED in m->d_hits.  The device buffer starts at `cap`
// tuples and the scan is repeated once with the exact size when more turn up and limit allows it (limit: the most
// the caller can take; beyond it *found is reported and the tuples are incomplete).  Waits for the stream.
int batch_scan(mk_matcher *m, uint64_t n_bytes, uint64_t n_rec, uint32_t mode, uint32_t batch_len, uint64_t cap, uint64_t limit,
               unsigned long long *found) {
    struct BatchLen {  // holds for the scans of THIS call only; a length set for device scans does not apply to them
        mk_matcher *m;
        uint32_t saved;
        ~BatchLen() {
            m->batch_rec_len = 0;
            m->fixed_rec_len = saved;
        }
    } guard{m, m->fixed_rec_len};
    m->fixed_rec_len = 0;
    m->batch_rec_len = batch_len;
    if (mode != MK_MODE_HITS) cap = 0;
    *found = 0;
    int rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (cap && (rc = ensure((void **)&m->d_hits, &m->d_hits_cap, cap * sizeof(mk_hit)))) return rc;
        rc = mk_scan_device(m, m->d_seq, n_bytes, m->d_off, n_rec, mode, m->d_flags, m->d_hits, cap, m->d_nhits, nullptr, m->stream);
        if (rc) return rc;
        MK_HIP(hipMemcpyAsync(found, m->d_nhits, sizeof(*found), hipMemcpyDeviceToHost, m->stream));
        MK_HIP(hipStreamSynchronize(m->stream));
        if (mode != MK_MODE_HITS || *found <= cap || *found > limit) break;
        cap = *found;  // the device buffer was the limit: once more, with room for all of them
    }
    return MK_OK;
}

// flags of the scanned batch -> host; what the batch looked like steers the load flavour of the next one
int batch_flags(mk_matcher *m, uint64_t n_rec, uint8_t *rec_flags, uint64_t *flagged_out) {
    MK_HIP(hipMemcpy(rec_flags, m->d_flags, n_rec, hipMemcpyDeviceToHost));
    uint64_t flagged = 0;
    for (uint64_t i = 0; i < n_rec; ++i) flagged += rec_flags[i] != 0;
    m->hit_density_pm = n_rec ? (uint32_t)(flagged * 1000 / n_rec) : 0;
    if (flagged_out) *flagged_out = flagged;
    return MK_OK;
}

}  // namespace mk

extern "C" {

int mk_scan_batch(mk_matcher *m, const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_rec, uint32_t mode,
                  uint8_t *rec_flags, mk_hit *hits, uint64_t hits_cap, uint64_t *n_hits) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (mode > MK_MODE_HITS) return fail(MK_E_INVALID_ARG, "unknown mode %u", mode);
    if (n_hits) *n_hits = 0;
    if (n_rec == 0) return MK_OK;
    if (!seq_off || !rec_flags) return fail(MK_E_INVALID_ARG, "null buffer");
    uint64_t n_bytes = 0;
    int rc = batch_check(seq_bytes, seq_off, n_rec, &n_bytes);
    if (rc) return rc;
    MK_ABI_BEGIN
    uint32_t batch_len = 0;
    if ((rc = batch_upload(m, seq_bytes, seq_off, n_rec, n_bytes, &batch_len))) return rc;
    unsigned long long found = 0;
    if ((rc = batch_scan(m, n_bytes, n_rec, mode, batch_len, std::max<uint64_t>(hits_cap, 1024), hits_cap, &found))) return rc;
    if ((rc = batch_flags(m, n_rec, rec_flags, nullptr))) return rc;
    if (mode == MK_MODE_HITS) {
        if (n_hits) *n_hits = found;
        if (found > hits_cap)
            return fail(MK_E_CAPACITY, "hits buffer too small: %llu occurrences, capacity %llu", found,
                        (unsigned long long)hits_cap);
        if (found) {
            // emission order: on the device for anything but a handful of tuples (one host thread sorts
            // ~8 M tuples/s; a batch where every read hits would spend 1000x its scan time there)
            constexpr uint64_t kSortOnDevice = 4096;
            if (found >= kSortOnDevice && (rc = mk_order_hits_device(m, m->d_hits, found, m->stream))) return rc;
            MK_HIP(hipMemcpyAsync(hits, m->d_hits, found * sizeof(mk_hit), hipMemcpyDeviceToHost, m->stream));
            MK_HIP(hipStreamSynchronize(m->stream));
            if (found < kSortOnDevice) mk_order_hits(m, hits, found);
        }
    }
    return MK_OK;
    MK_ABI_END
}

int mk_synth_reads_device(mk_matcher *m, uint64_t seed, uint64_t n_rec, uint32_t read_len, uint32_t plant_every,
                          void *d_seq, void *d_seq_off, void *stream) {
    return mk_synth_reads_device_range(m, seed, 0, n_rec, read_len, plant_every, d_seq, d_seq_off, stream);
}

int mk_synth_reads_device_range(mk_matcher *m, uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len,
                                uint32_t plant_every, void *d_seq, void *d_seq_off, void *stream) {
    if (!m || !d_seq || !d_seq_off) return fail(MK_E_INVALID_ARG, "null argument");
    if (((uint64_t)rec0 * read_len) % 32 != 0)
        return fail(MK_E_INVALID_ARG, "rec0 * read_len must be a multiple of 32 (generator block)");
    MK_HIP(hipSetDevice(m->device));
    launch_synth(seed, rec0, n_rec, read_len, plant_every, m->d_pat_bytes, m->d_pat_off, m->n_pat, (uint8_t *)d_seq,
                 (uint64_t *)d_seq_off, (hipStream_t)stream);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int mk_synth_reads_host(const mk_matcher *m, uint64_t seed, uint64_t rec0, uint64_t n_rec, uint32_t read_len,
                        uint32_t plant_every, uint8_t *seq, uint64_t *seq_off) {
    if (!m || !seq || !seq_off) return fail(MK_E_INVALID_ARG, "null argument");
    for (uint64_t r = 0; r <= n_rec; ++r) seq_off[r] = r * read_len;
    const uint64_t g0 = rec0 * read_len, nb = n_rec * read_len;
    for (uint64_t i = 0; i < nb; ++i) {
        const uint64_t g = g0 + i;
        seq[i] = synth_base(synth_block(seed, g >> 5), (uint32_t)(g & 31));
    }
    if (plant_every) {
        for (uint64_t k = 0; k < n_rec; ++k) {
            const uint64_t r = rec0 + k;
            const uint64_t h = synth_rec_hash(seed, r);
            if (h % plant_every != 0) continue;
            const uint32_t pat = (uint32_t)((r * 2654435761ull) % m->n_pat);
            const uint32_t a = m->pat_off[pat], len = m->pat_off[pat + 1] - a;
            if (len > read_len) continue;
            const uint32_t o = (uint32_t)((h >> 32) % (read_len - len + 1));
            memcpy(seq + k * read_len + o, m->pat_bytes.data() + a, len);
        }
    }
    return MK_OK;
}

}  // extern "C"
