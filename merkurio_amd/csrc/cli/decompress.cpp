// decompress.cpp -- bzip2 / xz / zstd input for `merkurio extract` (the reference reads .gz, .bz2 and
// .xz through needletail's `compression` feature: Cargo.toml:26, src/helpers.rs:48-68, README.md:39).
// This image ships the runtime libraries but not their headers, so the three streaming decoders
// are bound with dlopen and locally declared prototypes (stable public ABIs: libbz2 1.0,
// liblzma 5, libzstd 1).  A format whose library is genuinely absent fails with a clear message.
#include <dlfcn.h>

#include <cstring>

#include "io.hpp"

namespace cli {

namespace {

void *open_any(std::initializer_list<const char *> names) {
    for (const char *n : names)
        if (void *h = dlopen(n, RTLD_NOW)) return h;
    return nullptr;
}

// the output buffer of an inflater: starts small (at most 64 MiB, whatever the compressed size) and doubles; every
// doubling beyond 256 MiB first checks that the host has the memory (a clear error instead of the OOM killer)
constexpr size_t kFirstBuffer = 64u << 20;
size_t first_buffer(size_t compressed) { return std::min<size_t>(std::max<size_t>(compressed * 4, 1u << 20), kFirstBuffer); }
void grow(std::vector<char> &out, size_t used, const std::string &path) {
    if (out.size() - used < (1u << 16)) {
        const size_t want = std::max<size_t>(out.size() * 2, 1u << 22);
        if (want > (256u << 20)) require_host_memory(out.size() + want, path);
        out.resize(want);
    }
}

// ---- bzip2 (bzlib.h) ------------------------------------------------------------------------------
struct bz_stream {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
constexpr int BZ_OK = 0, BZ_STREAM_END = 4;

void inflate_bz2(const std::string &path, const unsigned char *d, size_t n, std::vector<char> &out) {
    void *h = open_any({"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"});
    if (!h) bail("bzip2 input needs libbz2.so.1.0, which is not installed: " + path);
    auto init = (int (*)(bz_stream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
    auto run = (int (*)(bz_stream *))dlsym(h, "BZ2_bzDecompress");
    auto end = (int (*)(bz_stream *))dlsym(h, "BZ2_bzDecompressEnd");
    if (!init || !run || !end) bail("libbz2 lacks the BZ2_bzDecompress API: " + path);
    size_t used = 0, pos = 0;
    out.resize(first_buffer(n));
    while (pos < n) {  // concatenated streams are legal (pbzip2 writes them)
        if (n - pos < 4 || memcmp(d + pos, "BZh", 3) != 0) {
            if (pos == 0) bail("Error while decompressing " + path);
            break;  // trailing garbage after a complete stream: ignored, as bzip2 itself does
        }
        bz_stream s;
        memset(&s, 0, sizeof(s));
        if (init(&s, 0, 0) != BZ_OK) bail("Error while decompressing " + path);
        s.next_in = (char *)const_cast<unsigned char *>(d + pos);
        size_t fed = 0;  // input handed over in <= 1 GiB pieces (32-bit counters)
        int r = BZ_OK;
        for (;;) {
            if (s.avail_in == 0 && pos + fed < n) {
                const size_t piece = std::min<size_t>(n - pos - fed, 1u << 30);
                s.avail_in = (unsigned)piece;
                fed += piece;
            }
            grow(out, used, path);
            const size_t room = std::min<size_t>(out.size() - used, 1u << 30);
            s.next_out = out.data() + used;
            s.avail_out = (unsigned)room;
            r = run(&s);
            used += room - s.avail_out;
            if (r == BZ_STREAM_END) break;
            if (r != BZ_OK || (s.avail_in == 0 && pos + fed >= n && s.avail_out != 0)) {
                end(&s);
                bail("Error while decompressing " + path);
            }
        }
        pos += fed - s.avail_in;
        end(&s);
    }
    out.resize(used);
}

// ---- xz (lzma/base.h, lzma/container.h) -------------------------------------------------------------
struct lzma_stream {
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    const void *allocator;
    void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
constexpr int LZMA_OK = 0, LZMA_STREAM_END = 1, LZMA_RUN = 0, LZMA_FINISH = 3;
constexpr uint32_t LZMA_CONCATENATED = 0x08;

void inflate_xz(const std::string &path, const unsigned char *d, size_t n, std::vector<char> &out) {
    void *h = open_any({"liblzma.so.5", "liblzma.so"});
    if (!h) bail("xz input needs liblzma.so.5, which is not installed: " + path);
    auto dec = (int (*)(lzma_stream *, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
    auto code = (int (*)(lzma_stream *, int))dlsym(h, "lzma_code");
    auto end = (void (*)(lzma_stream *))dlsym(h, "lzma_end");
    if (!dec || !code || !end) bail("liblzma lacks the lzma_stream_decoder API: " + path);
    lzma_stream s;
    memset(&s, 0, sizeof(s));  // LZMA_STREAM_INIT
    if (dec(&s, UINT64_MAX, LZMA_CONCATENATED) != LZMA_OK) bail("Error while decompressing " + path);
    s.next_in = d;
    s.avail_in = n;
    size_t used = 0;
    out.resize(first_buffer(n));
    for (;;) {
        grow(out, used, path);
        s.next_out = (uint8_t *)out.data() + used;
        s.avail_out = out.size() - used;
        const size_t room = s.avail_out;
        const int r = code(&s, s.avail_in ? LZMA_RUN : LZMA_FINISH);
        used += room - s.avail_out;
        if (r == LZMA_STREAM_END) break;
        if (r != LZMA_OK) {
            end(&s);
            bail("Error while decompressing " + path);
        }
    }
    end(&s);
    out.resize(used);
}

// ---- zstd (zstd.h) ------------------------------------------------------------------------------
struct ZSTD_inBuffer {
    const void *src;
    size_t size, pos;
};
struct ZSTD_outBuffer {
    void *dst;
    size_t size, pos;
};

void inflate_zstd(const std::string &path, const unsigned char *d, size_t n, std::vector<char> &out) {
    void *h = open_any({"libzstd.so.1", "libzstd.so"});
    if (!h) bail("zstd input needs libzstd.so.1, which is not installed: " + path);
    auto create = (void *(*)())dlsym(h, "ZSTD_createDStream");
    auto free_ = (size_t (*)(void *))dlsym(h, "ZSTD_freeDStream");
    auto run = (size_t (*)(void *, ZSTD_outBuffer *, ZSTD_inBuffer *))dlsym(h, "ZSTD_decompressStream");
    auto is_err = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    if (!create || !free_ || !run || !is_err) bail("libzstd lacks the ZSTD_decompressStream API: " + path);
    void *z = create();
    if (!z) bail("Error while decompressing " + path);
    ZSTD_inBuffer in{d, n, 0};
    size_t used = 0, last = 0;
    out.resize(first_buffer(n));
    while (in.pos < in.size || last != 0) {  // frames may be concatenated; last != 0: a frame is still open
        grow(out, used, path);
        ZSTD_outBuffer ob{out.data() + used, out.size() - used, 0};
        const size_t before = in.pos;
        last = run(z, &ob, &in);
        used += ob.pos;
        if (is_err(last) || (in.pos == before && ob.pos == 0 && in.pos >= in.size)) {  // error, or truncated input
            free_(z);
            bail("Error while decompressing " + path);
        }
    }
    free_(z);
    out.resize(used);
}

}  // namespace

bool inflate_by_magic(const std::string &path, const unsigned char *d, size_t n, std::vector<char> &out) {
    if (n >= 3 && !memcmp(d, "BZh", 3)) {
        inflate_bz2(path, d, n, out);
        return true;
    }
    if (n >= 6 && d[0] == 0xFD && !memcmp(d + 1, "7zXZ", 4) && d[5] == 0) {
        inflate_xz(path, d, n, out);
        return true;
    }
    if (n >= 4 && d[0] == 0x28 && d[1] == 0xB5 && d[2] == 0x2F && d[3] == 0xFD) {
        inflate_zstd(path, d, n, out);
        return true;
    }
    return false;
}

}  // namespace cli
