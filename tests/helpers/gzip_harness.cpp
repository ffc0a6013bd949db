// test harness (CPU): the serial pieces of the parallel gzip inflater (merkurio_amd/csrc/codec/gzip_segments.hpp) run the way the
// kernels of gzip_inflate.hip run them -- block starts searched from nominal cuts, segments decoded into 16-bit symbols with
// place-holders for the unknown 32 KiB in front, contexts resolved segment by segment, symbols translated -- and the text written to
// stdout; tests/test_codec_cpu.py compares it with zlib's.
// usage: gzip_harness <file.gz> <nominal chunk bytes>     (stderr: "segments N")
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gzip_segments.hpp"

int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> gz;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) gz.insert(gz.end(), buf, buf + n);
    fclose(f);
    const uint64_t chunk = strtoull(argv[2], nullptr, 10);
    // RFC 1952 header
    if (gz.size() < 18 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8) return 3;
    size_t p = 10;
    const uint8_t flg = gz[3];
    if (flg & 4) p += 2 + (gz[p] | gz[p + 1] << 8);
    if (flg & 8) p += strlen((const char *)&gz[p]) + 1;
    if (flg & 16) p += strlen((const char *)&gz[p]) + 1;
    if (flg & 2) p += 2;
    const uint64_t n_in = gz.size() - 8 - p;
    std::vector<uint8_t> in(gz.begin() + p, gz.end() - 8);
    in.resize(n_in + mkz::kStreamPad + 16, 0);
    std::vector<uint16_t> t(mkz::kLaneTableU16);
    // 1. block starts: the first confirmed start at or behind every nominal cut
    std::vector<uint64_t> starts{0};
    for (uint64_t cut = chunk; cut < n_in; cut += chunk) {
        uint64_t found = ~0ull;
        for (uint64_t bit = cut * 8; bit < (cut + 4 * chunk) * 8 && bit + 64 < n_in * 8; ++bit) {
            // the levels the search kernel runs on their own must be necessary conditions of the whole test, position by position
            uint32_t v = 0;
            for (int k = 0; k < 13; ++k) v |= (uint32_t)((in[(bit + k) >> 3] >> ((bit + k) & 7)) & 1u) << k;
            const bool l1 = mkz::seg_header_bits_plausible(v), l2 = mkz::seg_header_cl_plausible(in.data(), n_in, bit);
            const bool l3 = mkz::seg_header_plausible(in.data(), n_in, bit);
            if ((l2 && !l1) || (l3 && !l2)) {
                fprintf(stderr, "levels of the header test disagree at bit %llu: %d %d %d\n", (unsigned long long)bit, (int)l1, (int)l2, (int)l3);
                return 3;
            }
            if (l3 && mkz::seg_confirm_block_start(in.data(), n_in, bit, t.data())) {
                found = bit;
                break;
            }
        }
        if (found != ~0ull && found > starts.back()) starts.push_back(found);
    }
    fprintf(stderr, "segments %zu\n", starts.size());
    // 2. segments
    std::vector<std::vector<uint16_t>> seg(starts.size());
    std::vector<uint64_t> n_out(starts.size());
    for (size_t j = 0; j < starts.size(); ++j) {
        const uint64_t end = j + 1 < starts.size() ? starts[j + 1] : ~0ull;
        uint64_t cap = 1 << 16;
        for (;;) {
            seg[j].assign(mkz::kSegPrefix + cap + mkz::kSegSlack, 0);
            for (uint32_t k = 0; k < mkz::kSegPrefix; ++k) seg[j][k] = (uint16_t)(mkz::kSegUnknown | k);
            uint64_t stop = 0;
            bool fin = false;
            const int rc = mkz::inflate_segment(in.data(), n_in, starts[j], end, 0, seg[j].data() + mkz::kSegPrefix, cap, t.data(), &n_out[j], &stop, &fin);
            if (rc == mkz::kSegOverflow) {
                cap *= 4;
                continue;
            }
            if (rc != 0 || (j + 1 < starts.size() ? (stop != end || fin) : !fin)) {
                fprintf(stderr, "segment %zu: rc %d stop %llu end %llu final %d\n", j, rc, (unsigned long long)stop, (unsigned long long)end, (int)fin);
                return 4;
            }
            break;
        }
    }
    // 3. contexts and translation
    std::vector<uint8_t> ctx(mkz::kSegPrefix, 0), text;
    for (size_t j = 0; j < starts.size(); ++j) {
        const uint16_t *o = seg[j].data() + mkz::kSegPrefix;
        const size_t at = text.size();
        text.resize(at + n_out[j]);
        for (uint64_t i = 0; i < n_out[j]; ++i) {
            const uint16_t v = o[i];
            if (v >= 256 && !(v & mkz::kSegUnknown)) return 5;
            text[at + i] = v & mkz::kSegUnknown ? ctx[v & 0x7fff] : (uint8_t)v;
        }
        // the next context: the last 32 KiB of everything so far
        if (text.size() >= mkz::kSegPrefix) memcpy(ctx.data(), text.data() + text.size() - mkz::kSegPrefix, mkz::kSegPrefix);
        else if (!text.empty()) memcpy(ctx.data() + mkz::kSegPrefix - text.size(), text.data(), text.size());
    }
    if (!text.empty()) fwrite(text.data(), 1, text.size(), stdout);
    return 0;
}
