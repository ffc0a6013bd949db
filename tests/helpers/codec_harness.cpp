// codec_harness.cpp -- host build (g++) of the serial pieces of the device BGZF codec
// (merkurio_amd/csrc/codec/*.hpp), driven by tests/test_codec_cpu.py against zlib:
//   inflate <container> <out>   container = records {u32 n_in, u32 n_out, n_in bytes of raw DEFLATE}; writes the
//                               inflated bytes of every record back to back, prints one status per record
//   deflate <in> <block> <out>  cuts <in> into blocks, encodes each as ONE dynamic DEFLATE block with the shared code
//                               builder / header writer over a plain greedy parse; container {u32 n, bytes} out
//   crc <in> <pieces>           CRC-32 folded from <pieces> pieces, as the kernels fold the lanes' pieces
// Test infrastructure only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "codec/inflate_serial.hpp"

static std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<uint8_t> v;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

static int do_inflate(const char *in_path, const char *out_path) {
    std::vector<uint8_t> c = slurp(in_path);
    FILE *o = fopen(out_path, "wb");
    size_t at = 0;
    std::vector<uint32_t> lane(mkz::kLaneTableU16 / 2);  // the lane's decoder block (LDS on the device)
    while (at + 8 <= c.size()) {
        uint32_t n_in, n_out;
        memcpy(&n_in, &c[at], 4), memcpy(&n_out, &c[at + 4], 4);
        at += 8;
        std::vector<uint8_t> in(c.begin() + at, c.begin() + at + n_in);
        in.resize(n_in + mkz::kStreamPad, 0);
        at += n_in;
        std::vector<uint8_t> out(n_out + 8, 0xee);  // 8 readable bytes of slack, all of them guard bytes
        const int rc = mkz::inflate_stream(in.data(), n_in, out.data(), n_out, reinterpret_cast<uint16_t *>(lane.data()));
        bool intact = true;
        for (int g = 0; g < 8; ++g) intact &= out[n_out + g] == 0xee;
        printf("%d\n", intact ? rc : -99);
        fwrite(out.data(), 1, n_out, o);
    }
    fclose(o);
    return 0;
}

struct Token { uint16_t len, dist; uint8_t lit; };

static void parse_greedy(const uint8_t *b, size_t n, std::vector<Token> &tok) {
    std::vector<int32_t> head(1 << 15, -1);
    size_t i = 0;
    while (i < n) {
        uint32_t best = 0, bdist = 0;
        if (i + 4 <= n) {
            uint32_t w;
            memcpy(&w, b + i, 4);
            const uint32_t h = (w * 0x9e3779b1u) >> 17;
            const int32_t c = head[h];
            head[h] = (int32_t)i;
            if (c >= 0 && i - (size_t)c <= 32768) {
                uint32_t l = 0;
                while (l < 258 && i + l < n && b[c + l] == b[i + l]) ++l;
                if (l >= 4) best = l, bdist = (uint32_t)(i - (size_t)c);
            }
        }
        if (best) tok.push_back({(uint16_t)best, (uint16_t)(bdist - 1), 0}), i += best;
        else tok.push_back({0, 0, b[i]}), i += 1;
    }
}

static void encode_block(const uint8_t *b, size_t n, std::vector<uint8_t> &out) {
    std::vector<Token> tok;
    parse_greedy(b, n, tok);
    uint32_t fll[288] = {0}, fd[32] = {0};
    for (auto &t : tok) {
        if (t.len) {
            uint32_t idx, nb, xb, ds;
            mkz::length_symbol(t.len, idx, nb, xb);
            fll[257 + idx]++;
            mkz::distance_symbol(t.dist + 1u, ds, nb, xb);
            fd[ds]++;
        } else fll[t.lit]++;
    }
    fll[256] = 1;
    uint32_t kll[288], kd[32];
    const int mll = mkz::symbol_keys(fll, mkz::kLitLen, kll), md = mkz::symbol_keys(fd, mkz::kDist, kd);
    std::sort(kll, kll + mll), std::sort(kd, kd + md);
    mkz::BlockCodes c;
    static mkz::HeaderScratch h;
    mkz::block_codes_from_sorted(kll, mll, kd, md, c, h.huff);
    const uint32_t hdr_bits = mkz::plan_dynamic_header(c.ll_len, c.d_len, h);
    std::vector<uint32_t> words(n / 2 + 4096, 0);
    mkz::BitSink bs{words.data(), 0};
    mkz::write_dynamic_header(bs, h, true);
    if (bs.bitpos != hdr_bits) { fprintf(stderr, "header size %u != planned %u\n", bs.bitpos, hdr_bits); exit(3); }
    for (auto &t : tok) {
        if (t.len) {
            uint32_t idx, nb, xb;
            mkz::length_symbol(t.len, idx, nb, xb);
            mkz::put_bits(bs, c.ll_code[257 + idx], c.ll_len[257 + idx]);
            mkz::put_bits(bs, xb, nb);
            uint32_t ds;
            mkz::distance_symbol(t.dist + 1u, ds, nb, xb);
            mkz::put_bits(bs, c.d_code[ds], c.d_len[ds]);
            mkz::put_bits(bs, xb, nb);
        } else mkz::put_bits(bs, c.ll_code[t.lit], c.ll_len[t.lit]);
    }
    mkz::put_bits(bs, c.ll_code[256], c.ll_len[256]);
    const uint32_t nbytes = (bs.bitpos + 7) / 8;
    out.resize(nbytes);
    memcpy(out.data(), words.data(), nbytes);
}

static int do_deflate(const char *in_path, size_t block, const char *out_path) {
    std::vector<uint8_t> in = slurp(in_path);
    FILE *o = fopen(out_path, "wb");
    for (size_t at = 0; at < in.size() || at == 0; at += block) {
        const size_t n = std::min(block, in.size() - at);
        std::vector<uint8_t> z;
        encode_block(in.data() + at, n, z);
        const uint32_t zn = (uint32_t)z.size();
        fwrite(&zn, 4, 1, o);
        fwrite(z.data(), 1, zn, o);
        if (in.empty()) break;
    }
    fclose(o);
    return 0;
}

static int do_crc(const char *in_path, size_t pieces) {
    std::vector<uint8_t> in = slurp(in_path);
    uint32_t table[256];
    for (uint32_t i = 0; i < 256; ++i) table[i] = mkz::crc_table_entry(i);
    const size_t n = in.size(), piece = (n + pieces - 1) / std::max<size_t>(pieces, 1);
    uint32_t reg = 0;  // the register after the pieces folded so far
    for (size_t k = 0; k < pieces; ++k) {
        const size_t b = std::min(n, k * piece), e = std::min(n, b + piece);
        uint32_t r = k == 0 ? 0xffffffffu : 0u;
        for (size_t i = b; i < e; ++i) r = table[(r ^ in[i]) & 255u] ^ (r >> 8);
        reg = k == 0 ? r : (mkz::crc_mulmod(reg, mkz::crc_x_pow_bytes(e - b)) ^ r);
    }
    printf("%08x\n", reg ^ 0xffffffffu);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && !strcmp(argv[1], "inflate")) return do_inflate(argv[2], argv[3]);
    if (argc >= 5 && !strcmp(argv[1], "deflate")) return do_deflate(argv[2], (size_t)atol(argv[3]), argv[4]);
    if (argc >= 4 && !strcmp(argv[1], "crc")) return do_crc(argv[2], (size_t)atol(argv[3]));
    fprintf(stderr, "usage: codec_harness inflate|deflate|crc ...\n");
    return 2;
}
