#include "io.hpp"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../../../include/merkurio_hip.h"

// The device codec is reached through weak references: the reader / writer harnesses of the CPU tests compile this file
// without libmerkurio_hip.so (they never select a device: set_bgzf_device / BamWriter::use_device), the CLI links it.
extern "C" {
__attribute__((weak)) int mk_codec_create(int device, mk_codec **out);
__attribute__((weak)) void mk_codec_destroy(mk_codec *c);
__attribute__((weak)) uint64_t mk_bgzf_deflate_bound(uint64_t n, uint32_t block_bytes);
__attribute__((weak)) int mk_bgzf_deflate_pieces(mk_codec *c, const uint8_t *const *pieces, const uint64_t *sizes, uint64_t n_pieces, uint32_t block_bytes,
                                                 uint8_t *out, uint64_t out_cap, uint64_t *out_len);
__attribute__((weak)) int mk_bgzf_inflate(mk_codec *c, const uint8_t *in, uint64_t n_in, const mk_bgzf_member *members, uint64_t n_members, uint8_t *out,
                                          uint64_t out_cap, uint64_t *bad_member);
__attribute__((weak)) const char *mk_last_error(void);
__attribute__((weak)) int mk_host_alloc(size_t bytes, void **out);
__attribute__((weak)) void mk_host_free(void *p);
}

#include <algorithm>
#include <chrono>
#include <cstring>
#include <ctime>
#include <mutex>
#include <thread>

namespace cli {


static unsigned g_io_threads_cap = 0;  // 0 = no cap
void set_io_threads_cap(unsigned n) { g_io_threads_cap = n; }

unsigned io_threads() {
    static const unsigned cached = [] {
        if (const char *e = getenv("MERKURIO_IO_THREADS")) return (unsigned)std::max(1, atoi(e));
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        // containers: the cgroup CPU quota, not the machine's core count, is what this process gets
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
                hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, atoll(q) / period));
            fclose(f);
        }
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = std::min<unsigned>(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
        return std::min(hw, 32u);
    }();
    return g_io_threads_cap ? std::min(cached, g_io_threads_cap) : cached;
}

// Host memory this process may still take: MemAvailable, capped by the cgroup limit.  0 = unknown.
static uint64_t host_memory_available() {
    uint64_t avail = 0;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), f)) {
            unsigned long long kb = 0;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) avail = (uint64_t)kb << 10;
        }
        fclose(f);
    }
    unsigned long long lim = 0, cur = 0;
    FILE *fl = fopen("/sys/fs/cgroup/memory.max", "r"), *fc = fopen("/sys/fs/cgroup/memory.current", "r");
    if (fl && fc && fscanf(fl, "%llu", &lim) == 1 && fscanf(fc, "%llu", &cur) == 1 && lim > cur) {
        const uint64_t room = lim - cur;
        if (avail == 0 || room < avail) avail = room;
    }
    if (fl) fclose(fl);
    if (fc) fclose(fc);
    return avail;
}

// This build holds a decompressed input (and its record index) in host memory; the reference streams
// it.  Fail early and say so instead of being killed by the kernel half way through.
void require_host_memory(uint64_t need, const std::string &path) {
    const uint64_t avail = host_memory_available();
    if (avail == 0 || need <= avail) return;
    char buf[256];
    snprintf(buf, sizeof(buf), " needs about %.1f GiB of host memory once decompressed, %.1f GiB are available. ",
             need / 1073741824.0, avail / 1073741824.0);
    bail(path + buf +
         "This build keeps a compressed input in memory while it is processed (plain-text inputs are memory-mapped and do not "
         "count): decompress it first, split it, or run on a host with more memory.");
}

// BGZF = a series of gzip members, each with the extra subfield 'B','C' holding the member's
// size (SAM spec 4.1).  Walk the member headers, size the output from the ISIZE trailers, then
// inflate the members independently.  Returns false (out untouched) if the file is not BGZF all
// the way through -- the caller falls back to the serial gzip reader.
// Member table of a BGZF file held in d[0,n): false if it is not BGZF all the way through.
struct BgzfMember {
    size_t data_off, data_len;  // raw deflate stream
    size_t out_off;
    uint32_t isize, crc;
};
static bool bgzf_members(const uint8_t *d, size_t n, std::vector<BgzfMember> &mem, size_t *total_out) {
    size_t p = 0, total = 0;
    while (p < n) {
        if (n - p < 18 || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return false;
        const size_t xlen = d[p + 10] | (size_t)d[p + 11] << 8;
        if (n - p < 12 + xlen + 8) return false;
        size_t bsize = 0;
        for (size_t x = p + 12; x + 4 <= p + 12 + xlen;) {  // extra subfields
            const size_t slen = d[x + 2] | (size_t)d[x + 3] << 8;
            if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2 && x + 6 <= p + 12 + xlen) bsize = (d[x + 4] | (size_t)d[x + 5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || (d[p + 3] & ~4) || n - p < bsize) return false;
        BgzfMember b;
        b.data_off = p + 12 + xlen;
        b.data_len = bsize - (12 + xlen) - 8;
        memcpy(&b.crc, d + p + bsize - 8, 4);
        memcpy(&b.isize, d + p + bsize - 4, 4);
        b.out_off = total;
        total += b.isize;
        mem.push_back(b);
        p += bsize;
    }
    if (total_out) *total_out = total;
    return !mem.empty();
}
// The device codec for BGZF input (set_bgzf_device; merkurio_amd/csrc/codec/): one handle per process, created by the
// first window that is large enough to be worth a launch.  < 0: zlib on the host threads (--host-codec).
static int g_bgzf_device = -1;
static std::mutex g_bgzf_mu;
static mk_codec *g_bgzf_codec = nullptr;
static double g_bgzf_device_seconds = 0;
static size_t g_bgzf_device_min_members = 0;  // set below
void set_bgzf_device(int device, bool always);
double bgzf_device_seconds() { return g_bgzf_device_seconds; }
// A launch of the inflate kernel takes 35-60 ms whether it holds 64 members or 49 152 (one lane per member, bound by
// the latency of a lane's serial decode: profiles/r04_codec_kernels.txt); zlib on 16 host threads inflates ~4 GB/s.
// Below ~0.5 GB of text per call the host threads are done first (extract's 128 MB raw windows: 0.63-0.70 s on the
// host, 0.89-0.91 s through the device: profiles/r04_e2e_extract_bgzf.txt).
constexpr size_t kDeviceInflateMinMembers = 8192;
void set_bgzf_device(int device, bool always) {
    g_bgzf_device = mk_bgzf_inflate ? device : -1;
    g_bgzf_device_min_members = always ? 1 : kDeviceInflateMinMembers;
}

// inflates members [m0, m1) into out + (member.out_off - mem[m0].out_off): on the device (one lane per member,
// mk_bgzf_inflate: stream errors, ISIZE and CRC-32 checked there) or on the host threads
static void bgzf_inflate_range(const uint8_t *d, const std::vector<BgzfMember> &mem, size_t m0, size_t m1, char *out,
                               const std::string &path) {
    const size_t cnt = m1 - m0;
    // (a device-side failure that is not about the data -- no memory for a third set of codec buffers next to the matcher's
    // and the BAM writer's, a lost device -- is no reason to give up on an input zlib can read: said once, then the host path)
    auto device_gave_up = [&](const char *what) {
        fprintf(stderr, "Warning: BGZF input is inflated on the host threads from here on (%s: %s)\n", what, mk_last_error());
        g_bgzf_device = -1;
    };
    std::unique_lock<std::mutex> lock(g_bgzf_mu);
    if (g_bgzf_device >= 0 && cnt >= g_bgzf_device_min_members && !g_bgzf_codec && mk_codec_create(g_bgzf_device, &g_bgzf_codec) != MK_OK)
        device_gave_up("mk_codec_create");
    if (g_bgzf_device >= 0 && cnt >= g_bgzf_device_min_members) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t base = mem[m0].out_off, in_lo = mem[m0].data_off;
        std::vector<mk_bgzf_member> tab(cnt);
        uint64_t text = 0, in_hi = in_lo;
        for (size_t i = 0; i < cnt; ++i) {
            const BgzfMember &b = mem[m0 + i];
            tab[i] = mk_bgzf_member{b.data_off - in_lo, b.out_off - base, (uint32_t)b.data_len, b.isize, b.crc, 0};
            text = std::max<uint64_t>(text, b.out_off - base + b.isize);
            in_hi = std::max<uint64_t>(in_hi, b.data_off + b.data_len);
        }
        uint64_t bad = 0;
        const int rc = mk_bgzf_inflate(g_bgzf_codec, d + in_lo, in_hi - in_lo, tab.data(), cnt, (uint8_t *)out, text, &bad);
        if (rc == MK_E_CORRUPT) bail("Error while decompressing " + path);
        if (rc == MK_OK) {
            g_bgzf_device_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        device_gave_up("mk_bgzf_inflate");
    }
    lock.unlock();
    const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), cnt / 16 + 1));
    const size_t base = mem[m0].out_off;
    run_threads(T, [&](size_t t) {
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) bail("Error while decompressing " + path);
        for (size_t i = m0 + cnt * t / T; i < m0 + cnt * (t + 1) / T; ++i) {
            const BgzfMember &b = mem[i];
            inflateReset(&z);
            z.next_in = const_cast<Bytef *>(d + b.data_off);
            z.avail_in = (uInt)b.data_len;
            Bytef none = 0;  // (an empty member -- the end-of-file marker -- into an empty buffer: zlib refuses a NULL next_out)
            z.next_out = b.isize ? (Bytef *)out + (b.out_off - base) : &none;
            z.avail_out = b.isize;
            const int r = inflate(&z, Z_FINISH);
            if ((r != Z_STREAM_END && !(b.isize == 0 && r == Z_BUF_ERROR)) || z.avail_out != 0 ||
                (uint32_t)crc32(crc32(0, nullptr, 0), b.isize ? (const Bytef *)out + (b.out_off - base) : &none, b.isize) != b.crc) {
                inflateEnd(&z);
                bail("Error while decompressing " + path);
            }
        }
        inflateEnd(&z);
    });
}

static bool inflate_bgzf_parallel(const std::string &path, std::vector<char> &out) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) {
        close(fd);
        return false;
    }
    const size_t n = (size_t)st.st_size;
    void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    const uint8_t *d = (const uint8_t *)m;
    std::vector<BgzfMember> mem;
    size_t total = 0;
    if (!bgzf_members(d, n, mem, &total)) {
        munmap(m, n);
        return false;
    }
    try {
        require_host_memory(total + total / 8, path);  // + record index
        out.resize(total);
        bgzf_inflate_range(d, mem, 0, mem.size(), out.data(), path);
    } catch (...) {
        munmap(m, n);
        throw;
    }
    munmap(m, n);
    return true;
}

std::vector<char> read_file_maybe_gz(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) bail("No such file or directory: " + path);
    unsigned char magic[6] = {0};
    size_t got = fread(magic, 1, sizeof(magic), f);
    fclose(f);
    if ((got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h') ||
        (got >= 6 && magic[0] == 0xFD && !memcmp(magic + 1, "7zXZ", 4)) ||
        (got >= 4 && magic[0] == 0x28 && magic[1] == 0xB5 && magic[2] == 0x2F && magic[3] == 0xFD)) {
        FileBytes fb;  // bzip2 / xz / zstd: FileBytes::load inflates by magic
        fb.load(path);
        return std::vector<char>(fb.p, fb.p + fb.n);
    }
    {  // BGZF (BAM, bgzip'ed FASTQ): independent <= 64 KiB members, inflated on every host thread
        std::vector<char> out;
        if (inflate_bgzf_parallel(path, out)) return out;
    }
    // gzopen reads plain files transparently and walks concatenated gzip members
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) bail("Cannot open " + path);
    gzbuffer(g, 1 << 20);
    std::vector<char> out;
    size_t cap = 1 << 22;
    out.resize(cap);
    size_t n = 0;
    for (;;) {
        if (n == cap) {
            require_host_memory(cap * 2, path);  // the buffer doubles: both copies exist for a moment
            cap *= 2;
            out.resize(cap);
        }
        int r = gzread(g, out.data() + n, (unsigned)std::min<size_t>(cap - n, 1u << 30));
        if (r < 0) {
            gzclose(g);
            bail("Error while decompressing " + path);
        }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(g);
    out.resize(n);
    return out;
}

FileBytes::~FileBytes() {
    if (map) munmap(map, map_len);
}

void FileBytes::load(const std::string &path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) bail("No such file or directory: " + path);
    unsigned char magic[2] = {0, 0};
    ssize_t got = pread(fd, magic, 2, 0);
    struct stat st;
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        // no MAP_POPULATE: the inputs are read a window at a time; the reader asks for each window ahead of
        // its parse (FastxStream::prefetch, SamFile::fill), the kernel reads ahead sequentially
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            map = m;
            map_len = (uint64_t)st.st_size;
            p = (const char *)m;
            n = map_len;
            // bzip2 / xz / zstd (needletail's `compression` feature, Cargo.toml:26): inflated by magic
            std::vector<char> out;
            if (inflate_by_magic(path, (const unsigned char *)p, (size_t)n, out)) {
                munmap(map, map_len);
                map = nullptr;
                map_len = 0;
                owned = std::move(out);
                p = owned.data();
                n = owned.size();
            }
            return;
        }
    }
    close(fd);
    owned = read_file_maybe_gz(path);
    p = owned.data();
    n = owned.size();
}

// ---- FASTA / FASTQ ------------------------------------------------------------------------------
static uint64_t line_end(const char *d, uint64_t n, uint64_t b) {  // index of '\n' or n
    const void *p = memchr(d + b, '\n', n - b);
    return p ? (uint64_t)((const char *)p - d) : n;
}
static uint64_t strip_cr(const char *d, uint64_t b, uint64_t e) { return (e > b && d[e - 1] == '\r') ? e - 1 : e; }

// records of d[p, stop): p is at a record start (or blank lines before one); a record that starts
// before `stop` is parsed completely even if it runs past it
// partial_ok (windowed reading: more text follows d[0,n)): a record that cannot be shown to end inside
// d[0,n) is not an error, parsing stops in front of it.  Returns the offset where parsing stopped.
static uint64_t parse_fastx_range(const char *d, uint64_t n, uint64_t p, uint64_t stop, bool fastq,
                                  std::vector<FastxFile::Rec> &recs, bool partial_ok = false) {
    while (p < stop) {
        if (d[p] == '\n' || d[p] == '\r') {  // blank line between records
            ++p;
            continue;
        }
        FastxFile::Rec r{};
        if (!fastq) {
            if (d[p] != '>') bail("Error during FASTQ/A record parsing.");
            uint64_t e = line_end(d, n, p);
            r.id_b = p + 1;
            r.id_e = strip_cr(d, p + 1, e);
            uint64_t s = std::min(e + 1, n);
            r.raw_b = s;
            uint64_t q = s;  // the sequence runs to the line end before the next line that starts with '>'
            for (;;) {
                if (q >= n || d[q] == '>') break;
                q = std::min(line_end(d, n, q) + 1, n);
            }
            if (partial_ok && q >= n) return p;  // no following header in sight: the record may go on
            uint64_t re = q;
            while (re > s && (d[re - 1] == '\n' || d[re - 1] == '\r')) --re;  // drop the final line end
            r.raw_e = re;
            r.qual_b = r.qual_e = 0;
            p = q;
        } else {
            if (d[p] != '@') bail("Error during FASTQ/A record parsing.");
            uint64_t e1 = line_end(d, n, p);
            if (partial_ok) {  // all four lines, line ends included, must be inside the window
                uint64_t e = e1;
                for (int k = 0; k < 3 && e < n; ++k) e = line_end(d, n, e + 1);
                if (e >= n) return p;
            }
            if (e1 >= n) bail("Error during FASTQ/A record parsing.");
            r.id_b = p + 1;
            r.id_e = strip_cr(d, p + 1, e1);
            uint64_t e2 = line_end(d, n, e1 + 1);
            r.raw_b = e1 + 1;
            r.raw_e = strip_cr(d, e1 + 1, e2);
            if (e2 + 1 >= n || d[e2 + 1] != '+') bail("Error during FASTQ/A record parsing.");
            uint64_t e3 = line_end(d, n, e2 + 1);
            if (e3 >= n) bail("Error during FASTQ/A record parsing.");
            uint64_t e4 = line_end(d, n, e3 + 1);
            r.qual_b = e3 + 1;
            r.qual_e = strip_cr(d, e3 + 1, e4);
            if (r.qual_e - r.qual_b != r.raw_e - r.raw_b) bail("Error during FASTQ/A record parsing.");
            p = std::min(e4 + 1, n);
        }
        recs.push_back(r);
    }
    return p;
}

// first record start at or after `from` (line-aligned).  FASTQ: a line starting with '@' whose
// second-next line starts with '+' (a quality line may start with '@', but then the line two
// below it is a sequence line, which never starts with '+').
static uint64_t next_record_start(const char *d, uint64_t n, uint64_t from, bool fastq) {
    uint64_t p = from == 0 ? 0 : std::min(line_end(d, n, from - 1) + 1, n);
    while (p < n) {
        if (!fastq) {
            if (d[p] == '>') return p;
        } else if (d[p] == '@') {
            uint64_t e1 = line_end(d, n, p);
            uint64_t e2 = e1 < n ? line_end(d, n, e1 + 1) : n;
            if (e2 + 1 < n && d[e2 + 1] == '+') return p;
        }
        p = std::min(line_end(d, n, p) + 1, n);
    }
    return n;
}

uint64_t FastxFile::append_seq(size_t i, std::vector<uint8_t> &out) const {
    const Rec &r = recs[i];
    const size_t before = out.size();
    if (fastq) {
        out.insert(out.end(), data + r.raw_b, data + r.raw_e);
    } else {
        for (uint64_t k = r.raw_b; k < r.raw_e; ++k)
            if (data[k] != '\n' && data[k] != '\r') out.push_back((uint8_t)data[k]);
    }
    return out.size() - before;
}

void FastxFile::gather(size_t b0, size_t b1, std::vector<uint8_t> &seq, std::vector<uint64_t> &off) const {
    const size_t nb = b1 - b0;
    off.resize(nb + 1);
    off[0] = 0;
    if (!fastq) {  // FASTA: line breaks have to be squeezed out; few, long records
        seq.clear();
        for (size_t i = b0; i < b1; ++i) {
            append_seq(i, seq);
            off[i - b0 + 1] = seq.size();
        }
        seq.push_back(0);
        return;
    }
    for (size_t i = 0; i < nb; ++i) off[i + 1] = off[i] + raw_len(b0 + i);
    seq.resize(off[nb] + 1);
    const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)io_threads(), nb / 65536 + 1));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (size_t i = nb * t / T; i < nb * (t + 1) / T; ++i)
                memcpy(seq.data() + off[i], data + recs[b0 + i].raw_b, off[i + 1] - off[i]);
        });
    for (auto &x : th) x.join();
    seq[off[nb]] = 0;
}

void FastxFile::write(size_t i, Sink &w) const {
    // needletail's SequenceRecord::write(writer, None) ends every line with the record's own line
    // ending, detected on its header line: CRLF input stays CRLF (the wrapped lines inside a FASTA
    // raw_seq keep theirs anyway)
    const Rec &r = recs[i];
    const bool crlf = r.id_e < data_n && data[r.id_e] == '\r';
    const char *nl = crlf ? "\r\n" : "\n";
    const size_t nl_len = crlf ? 2 : 1;
    w.write(fastq ? "@" : ">", 1);
    w.write(data + r.id_b, r.id_e - r.id_b);
    w.write(nl, nl_len);
    w.write(data + r.raw_b, r.raw_e - r.raw_b);
    w.write(nl, nl_len);
    if (fastq) {
        w.write("+", 1);
        w.write(nl, nl_len);
        w.write(data + r.qual_b, r.qual_e - r.qual_b);
        w.write(nl, nl_len);
    }
}

// ---- WindowSource: the bytes of an input, decompressed a window at a time -----------------------------
WindowSource::~WindowSource() {
    if (zs) {
        inflateEnd((z_stream *)zs);
        delete (z_stream *)zs;
    }
}

void WindowSource::open(const std::string &p) {
    path = p;
    int fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) bail("No such file or directory: " + p);
    unsigned char magic[2] = {0, 0};
    const ssize_t got = pread(fd, magic, 2, 0);
    struct stat st;
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (gz && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) bail("Cannot open " + p);
        src.map = m;
        src.map_len = (uint64_t)st.st_size;
        src.p = (const char *)m;
        src.n = src.map_len;
        std::vector<BgzfMember> mem;
        if (bgzf_members((const uint8_t *)src.p, src.n, mem, nullptr)) {
            kind = BGZF;
            for (auto &b : mem) members.push_back(Member{b.data_off, b.data_len, b.isize, b.crc});
        } else {
            kind = GZIP;
            z_stream *z = new z_stream;
            memset(z, 0, sizeof(*z));
            if (inflateInit2(z, 15 + 32) != Z_OK) {  // gzip or zlib header, detected
                delete z;
                bail("Error while decompressing " + p);
            }
            zs = z;
        }
        return;
    }
    close(fd);
    kind = PLAIN;
    src.load(p);  // mmap; bzip2 / xz / zstd are inflated whole by magic
}

bool WindowSource::more_into(std::vector<char> &buf, uint64_t &buf_len, uint64_t want) {
    if (kind == PLAIN || src_eof) return false;
    if (want < (1u << 16)) want = 1u << 16;
    require_host_memory(buf_len + want + (buf_len + want) / 8, path);
    if (kind == BGZF) {
        size_t m1 = (size_t)src_pos;
        uint64_t add = 0;
        while (m1 < members.size() && (add < want || m1 == (size_t)src_pos)) add += members[m1++].isize;
        if (buf.size() < buf_len + add) buf.resize(buf_len + add);
        std::vector<BgzfMember> part;
        uint64_t o = 0;
        for (size_t i = (size_t)src_pos; i < m1; ++i) {
            part.push_back(BgzfMember{(size_t)members[i].data_off, (size_t)members[i].data_len, (size_t)o, members[i].isize, members[i].crc});
            o += members[i].isize;
        }
        if (!part.empty()) bgzf_inflate_range((const uint8_t *)src.p, part, 0, part.size(), buf.data() + buf_len, path);
        buf_len += add;
        src_pos = m1;
        if (src_pos >= members.size()) src_eof = true;
        return true;
    }
    // GZIP: zlib streaming over the mapped file, concatenated members included
    z_stream *z = (z_stream *)zs;
    if (buf.size() < buf_len + want) buf.resize(buf_len + want);
    uint64_t room = buf.size() - buf_len;
    while (room > 0 && !src_eof) {
        z->next_in = (Bytef *)const_cast<char *>(src.p + src_pos);
        const uInt in_piece = (uInt)std::min<uint64_t>(src.n - src_pos, 1u << 30);
        z->avail_in = in_piece;
        z->next_out = (Bytef *)buf.data() + buf_len;
        const uInt out_piece = (uInt)std::min<uint64_t>(room, 1u << 30);
        z->avail_out = out_piece;
        const int r = inflate(z, Z_NO_FLUSH);
        src_pos += in_piece - z->avail_in;
        buf_len += out_piece - z->avail_out;
        room -= out_piece - z->avail_out;
        if (r == Z_STREAM_END) {
            if (src_pos >= src.n) {
                src_eof = true;
            } else if (inflateReset(z) != Z_OK) {  // next member
                bail("Error while decompressing " + path);
            }
        } else if (r != Z_OK && r != Z_BUF_ERROR) {
            bail("Error while decompressing " + path);
        } else if (src_pos >= src.n && z->avail_out != 0) {
            bail("Error while decompressing " + path);  // the input ends inside a member
        }
    }
    return true;
}

// ---- FastxStream: FASTA / FASTQ records, one window at a time -----------------------------------------
// parses d[from, stop) into spare_recs (spare_data / spare_n / spare_end)
void FastxStream::parse_window(const char *d, uint64_t n, uint64_t from, uint64_t stop, bool partial_ok) {
    spare_data = d;
    spare_n = n;
    spare_recs.clear();
    uint64_t p = from;
    while (p < n && (d[p] == '\n' || d[p] == '\r')) ++p;
    spare_end = p;
    if (p >= n) return;
    if (!started) {
        if (d[p] != '>' && d[p] != '@') bail("Error during FASTQ/A record parsing.");
        fastq = d[p] == '@';
        started = true;
    }
    if (stop <= p) stop = std::min(n, p + 1);
    // split at record starts and parse the pieces on host threads
    uint64_t T = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)io_threads(), (stop - p) / (16u << 20) + 1));
    std::vector<uint64_t> cut(T + 1);
    cut[0] = p;
    cut[T] = stop;
    for (uint64_t t = 1; t < T; ++t) cut[t] = std::min(stop, std::max(cut[t - 1], next_record_start(d, n, p + (stop - p) * t / T, fastq)));
    // A record longer than a piece (a chromosome in a FASTA) leaves no record start between the last split
    // targets and `stop`: those cuts were clamped to `stop`, i.e. to the middle of that record.  Drop the empty
    // trailing pieces, so that the piece that owns the record is the LAST one: it reports where the window really
    // ends, and it is the one that may find its record cut off by the end of a compressed buffer.
    while (T > 1 && cut[T - 1] >= stop) --T;  // cut[T] == stop still holds
    std::vector<std::vector<FastxFile::Rec>> parts(T);
    std::vector<uint64_t> ends(T, 0);
    run_threads((size_t)T, [&](size_t t) {
        parts[t].reserve((cut[t + 1] - cut[t]) / 200 + 16);
        // only the last piece can run into the end of the window
        ends[t] = parse_fastx_range(d, n, cut[t], cut[t + 1], fastq, parts[t], partial_ok && t + 1 == T);
    });
    size_t total = 0;
    for (auto &v : parts) total += v.size();
    spare_recs.reserve(total);
    for (auto &v : parts) spare_recs.insert(spare_recs.end(), v.begin(), v.end());
    spare_end = ends[T - 1];
}

void FastxStream::prefetch(uint64_t window_bytes) {
    if (have_spare) return;
    if (window_bytes < (1u << 16)) window_bytes = 1u << 16;
    if (src.mapped()) {
        const uint64_t n = src.text_size();
        // a record that starts inside the window is parsed completely (the text behind it is mapped too)
        if (cursor < n) {
            const uint64_t a0 = cursor & ~(uint64_t)4095;  // page-aligned hint: this window's pages, now
            (void)madvise(const_cast<char *>(src.text()) + a0, (size_t)(std::min(n, cursor + window_bytes + (1u << 20)) - a0), MADV_WILLNEED);
            parse_window(src.text(), n, cursor, std::min(n, cursor + window_bytes), false);
        }
        else
            parse_window(src.text(), n, n, n, false);
        have_spare = true;
        return;
    }
    // compressed: the unconsumed tail of the current buffer opens the other one, then inflate until a
    // complete record is in it
    std::vector<char> &nb = bufs[cur ^ 1];
    uint64_t &nl = lens[cur ^ 1];
    const uint64_t tail = lens[cur] > cursor ? lens[cur] - cursor : 0;
    if (nb.size() < tail) nb.resize(tail);
    if (tail) memcpy(nb.data(), bufs[cur].data() + cursor, tail);
    nl = tail;
    uint64_t want = window_bytes > nl ? window_bytes - nl : 0;
    for (;;) {
        if (want) src.more_into(nb, nl, want);
        parse_window(nb.data(), nl, 0, nl, !src.exhausted());
        if (!spare_recs.empty() || src.exhausted()) break;
        want = std::max<uint64_t>(window_bytes, nl);  // one record larger than the window: take more
    }
    have_spare = true;
}

bool FastxStream::fill(uint64_t window_bytes) {
    prefetch(window_bytes);
    have_spare = false;
    view.recs.swap(spare_recs);
    spare_recs.clear();
    view.data = spare_data;
    view.data_n = spare_n;
    view.fastq = fastq;
    cur_end = spare_end;
    if (!src.mapped()) cur ^= 1;
    cursor = cur_end;  // until consume() says otherwise: the whole window
    return !view.recs.empty();
}

// ---- raw windows (device ingest) ------------------------------------------------------------------------------
// the last record start of d[0, n) at or behind `from` (n if there is none)
static uint64_t last_record_start(const char *d, uint64_t n, uint64_t from, bool fastq) {
    uint64_t last = n;
    for (uint64_t r = next_record_start(d, n, from, fastq); r < n; r = next_record_start(d, n, r + 1, fastq)) last = r;
    if (last == n && !fastq && from > 0) {
        // FASTA records can be far longer than the stretch that was searched (a chromosome): walk back to the last '>' at a line start
        for (uint64_t p = from; p > 0;) {
            const void *g = memrchr(d, '>', (size_t)p);
            if (!g) break;
            p = (uint64_t)((const char *)g - d);
            if (p == 0 || d[p - 1] == '\n') return p;
        }
    }
    return last;
}

void FastxFile::parse_span(uint64_t b, uint64_t e) { parse_fastx_range(data, data_n, b, e, fastq, recs); }
uint64_t FastxFile::parse_span_partial(uint64_t b, uint64_t e, bool partial_ok) { return parse_fastx_range(data, data_n, b, e, fastq, recs, partial_ok); }

void inflate_bgzf_members_host(const uint8_t *file, const mk_bgzf_member *members, size_t n, char *out, const std::string &path) {
    std::vector<BgzfMember> mem(n);
    for (size_t i = 0; i < n; ++i) mem[i] = BgzfMember{(size_t)members[i].data_off, (size_t)members[i].data_len, (size_t)members[i].out_off, members[i].isize, members[i].crc};
    if (n == 0) return;
    const int saved = g_bgzf_device;  // (this is the host's own checker path: never the device codec)
    g_bgzf_device = -1;
    try {
        bgzf_inflate_range(file, mem, 0, n, out + mem[0].out_off, path);
    } catch (...) {
        g_bgzf_device = saved;
        throw;
    }
    g_bgzf_device = saved;
}

bool FastxStream::raw_fill(uint64_t window_bytes, const char **text, uint64_t *n_out, uint64_t *resume) {
    *resume = 0;
    if (window_bytes < (1u << 16)) window_bytes = 1u << 16;
    if (src.mapped()) {
        const char *d = src.text();
        const uint64_t n = src.text_size();
        uint64_t p = cursor;
        while (p < n && (d[p] == '\n' || d[p] == '\r')) ++p;
        if (p >= n) return false;
        if (!started) {
            if (d[p] != '@' && d[p] != '>') return false;
            fastq = d[p] == '@';
            started = true;
        }
        if (d[p] != (fastq ? '@' : '>')) return false;
        uint64_t end = std::min(n, p + window_bytes);
        if (end < n) {
            const uint64_t a0 = p & ~(uint64_t)4095;
            (void)madvise(const_cast<char *>(d) + a0, (size_t)(std::min(n, end + (1u << 20)) - a0), MADV_WILLNEED);
            end = next_record_start(d, n, end, fastq);  // the first record start at or behind the target
        }
        cursor = p;
        raw_next = end;
        *resume = end;
        *text = d + p;
        *n_out = end - p;
        return end > p;
    }
    // compressed: the unconsumed tail of the current buffer opens the other one (as prefetch() does), then inflate
    std::vector<char> &nb = bufs[cur ^ 1];
    uint64_t &nl = lens[cur ^ 1];
    const uint64_t tail = lens[cur] > cursor ? lens[cur] - cursor : 0;
    if (nb.size() < tail) nb.resize(tail);
    if (tail) memcpy(nb.data(), bufs[cur].data() + cursor, tail);
    nl = tail;
    uint64_t want = window_bytes > nl ? window_bytes - nl : 0;
    uint64_t cut = 0;
    for (;;) {
        if (want) src.more_into(nb, nl, want);
        uint64_t p = 0;
        while (p < nl && (nb[p] == '\n' || nb[p] == '\r')) ++p;
        if (p >= nl && src.exhausted()) {
            cur ^= 1;
            cursor = nl;
            return false;
        }
        if (p < nl && !started && (nb[p] == '@' || nb[p] == '>')) {
            fastq = nb[p] == '@';
            started = true;
        }
        if (p < nl && (!started || nb[p] != (fastq ? '@' : '>'))) {  // neither FASTQ nor FASTA: leave everything to fill()
            cur ^= 1;
            cursor = 0;
            return false;
        }
        if (src.exhausted()) {
            cut = nl;
            break;
        }
        // more text follows: hand out whole records only -- everything in front of the last record start
        cut = last_record_start(nb.data(), nl, nl > (1u << 20) ? nl - (1u << 20) : 0, fastq);
        if (cut < nl && cut > p) break;
        want = std::max<uint64_t>(window_bytes, nl);  // one record larger than the window: take more
    }
    cur ^= 1;
    cursor = 0;  // (until raw_consume: a refused window is parsed again by fill())
    raw_next = cut;
    *text = nb.data();
    *n_out = cut;
    return cut > 0;
}

void FastxStream::raw_consume() { cursor = raw_next; }

bool FastxStream::raw_rest(const char **text, uint64_t *n_out) {
    if (src.mapped()) {
        const uint64_t n = src.text_size();
        if (cursor >= n) return false;
        *text = src.text() + cursor;
        *n_out = n - cursor;
        cursor = n;
        return true;
    }
    std::vector<char> &nb = bufs[cur ^ 1];
    uint64_t &nl = lens[cur ^ 1];
    const uint64_t tail = lens[cur] > cursor ? lens[cur] - cursor : 0;
    if (nb.size() < tail) nb.resize(tail);
    if (tail) memcpy(nb.data(), bufs[cur].data() + cursor, tail);
    nl = tail;
    while (!src.exhausted()) src.more_into(nb, nl, 64u << 20);
    cur ^= 1;
    cursor = nl;
    *text = nb.data();
    *n_out = nl;
    return nl > 0;
}

void FastxStream::consume(size_t n) { cursor = n < view.recs.size() ? view.recs[n].id_b - 1 : cur_end; }

// ---- SAM / BAM ------------------------------------------------------------------------------------
// SAM text: header lines ('@') may appear anywhere a line starts; records keep their file order.
// The text is cut at line starts into one piece per host thread; a record is five offsets.
// lines of d[from, stop) (both at line starts); keep_header: '@' lines are appended to out.header
// (whole-file parse), else they are skipped (windowed reading took the header at open())
static void parse_sam_text(const char *d, uint64_t n, uint64_t from, uint64_t stop, bool keep_header, SamFile &out) {
    const uint64_t span = stop - from;
    const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), span / (8u << 20) + 1));
    std::vector<uint64_t> cut(T + 1, stop);
    cut[0] = from;
    for (size_t t = 1; t < T; ++t) {
        uint64_t p = std::max<uint64_t>(cut[t - 1], from + span * t / T);
        if (p > 0 && p < stop) p = line_end(d, n, p - 1) + 1;  // first line start at or after p
        cut[t] = std::min(p, stop);
    }
    std::vector<std::vector<SamFile::Rec>> parts(T);
    std::vector<std::string> headers(T);
    run_threads(T, [&](size_t t) {
        uint64_t p = cut[t];
        const uint64_t stop = cut[t + 1];
        parts[t].reserve((stop - p) / 300 + 16);
        while (p < stop) {
            const uint64_t e = line_end(d, n, p);
            const uint64_t le = strip_cr(d, p, e);
            if (le > p) {
                if (d[p] == '@') {
                    if (keep_header) {
                        headers[t].append(d + p, le - p);
                        headers[t] += '\n';
                    }
                } else {
                    SamFile::Rec r;
                    r.off = p;
                    r.len = (uint32_t)(le - p);
                    const char *q = (const char *)memchr(d + p, '\t', le - p);
                    if (!q) bail("Error during SAM record parsing: too few fields");
                    r.name_len = (uint32_t)(q - (d + p));
                    for (int f = 1; f < 9; ++f) {  // q: the tab in front of field f + 1
                        q = (const char *)memchr(q + 1, '\t', d + le - (q + 1));
                        if (!q) bail("Error during SAM record parsing: too few fields");
                    }
                    const char *sb = q + 1;
                    const char *se = (const char *)memchr(sb, '\t', d + le - sb);
                    if (!se) se = d + le;
                    r.seq_off = (uint64_t)(sb - d);
                    r.l_seq = (se - sb == 1 && *sb == '*') ? 0 : (uint32_t)(se - sb);
                    parts[t].push_back(r);
                }
            }
            p = e + 1;
        }
    });
    size_t total = 0;
    for (auto &v : parts) total += v.size();
    out.recs.reserve(total);
    for (size_t t = 0; t < T; ++t) {
        out.header += headers[t];
        out.recs.insert(out.recs.end(), parts[t].begin(), parts[t].end());
    }
}

namespace {
struct Cur {
    const uint8_t *p, *e;
    template <class T>
    T get() {
        if ((size_t)(e - p) < sizeof(T)) bail("Error during BAM record parsing: truncated file");
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t *take(size_t n) {
        if ((size_t)(e - p) < n) bail("Error during BAM record parsing: truncated file");
        const uint8_t *r = p;
        p += n;
        return r;
    }
};
}  // namespace

static void aux_to_text(Cur &c, std::string &out) {
    while (c.p < c.e) {
        const uint8_t *tag = c.take(2);
        char type = (char)c.get<uint8_t>();
        out += '\t';
        out.append((const char *)tag, 2);
        char buf[64];
        switch (type) {
        case 'A': out += ":A:"; out += (char)c.get<uint8_t>(); break;
        case 'c': out += ":i:" + std::to_string((int)c.get<int8_t>()); break;
        case 'C': out += ":i:" + std::to_string((unsigned)c.get<uint8_t>()); break;
        case 's': out += ":i:" + std::to_string((int)c.get<int16_t>()); break;
        case 'S': out += ":i:" + std::to_string((unsigned)c.get<uint16_t>()); break;
        case 'i': out += ":i:" + std::to_string(c.get<int32_t>()); break;
        case 'I': out += ":i:" + std::to_string(c.get<uint32_t>()); break;
        case 'f': snprintf(buf, sizeof(buf), "%g", (double)c.get<float>()); out += ":f:"; out += buf; break;
        case 'Z':
        case 'H': {
            out += type == 'Z' ? ":Z:" : ":H:";
            const uint8_t *s = c.p;
            while (c.p < c.e && *c.p) ++c.p;
            out.append((const char *)s, c.p - s);
            if (c.p < c.e) ++c.p;
            break;
        }
        case 'B': {
            char sub = (char)c.get<uint8_t>();
            int32_t cnt = c.get<int32_t>();
            out += ":B:";
            out += sub;
            for (int32_t k = 0; k < cnt; ++k) {
                out += ',';
                switch (sub) {
                case 'c': out += std::to_string((int)c.get<int8_t>()); break;
                case 'C': out += std::to_string((unsigned)c.get<uint8_t>()); break;
                case 's': out += std::to_string((int)c.get<int16_t>()); break;
                case 'S': out += std::to_string((unsigned)c.get<uint16_t>()); break;
                case 'i': out += std::to_string(c.get<int32_t>()); break;
                case 'I': out += std::to_string(c.get<uint32_t>()); break;
                case 'f': snprintf(buf, sizeof(buf), "%g", (double)c.get<float>()); out += buf; break;
                default: bail("Error during BAM record parsing: bad B array subtype");
                }
            }
            break;
        }
        default: bail("Error during BAM record parsing: unknown tag type");
        }
    }
}

static const char kBamSeq[] = "=ACMGRSVTWYHKDBN";

// one BAM alignment record (at its block_size field) -> SAM text line appended to s
static void bam_record_to_sam(const uint8_t *at, const std::vector<std::string> &refs, std::string &s) {
    static const char kCig[] = "MIDNSHP=X";
    const int32_t n_ref = (int32_t)refs.size();
    int32_t block;
    memcpy(&block, at, 4);
    Cur r{at + 4, at + 4 + block};
    int32_t ref_id = r.get<int32_t>(), pos = r.get<int32_t>();
    uint8_t l_name = r.get<uint8_t>(), mapq = r.get<uint8_t>();
    (void)r.get<uint16_t>();
    uint16_t n_cig = r.get<uint16_t>(), flag = r.get<uint16_t>();
    int32_t l_seq = r.get<int32_t>(), next_ref = r.get<int32_t>(), next_pos = r.get<int32_t>(), tlen = r.get<int32_t>();
    const uint8_t *nm = r.take(l_name);
    s.reserve(s.size() + (size_t)block + (size_t)l_seq + 64);
    s.append((const char *)nm, l_name ? l_name - 1u : 0u);
    s += '\t' + std::to_string(flag) + '\t';
    s += ref_id >= 0 && ref_id < n_ref ? refs[ref_id] : "*";
    s += '\t' + std::to_string(pos + 1) + '\t' + std::to_string(mapq) + '\t';
    if (n_cig == 0) s += '*';
    for (uint16_t k = 0; k < n_cig; ++k) {
        uint32_t v = r.get<uint32_t>();
        s += std::to_string(v >> 4);
        s += (v & 15) < 9 ? kCig[v & 15] : '?';
    }
    s += '\t';
    s += next_ref < 0 ? "*" : (next_ref == ref_id ? "=" : (next_ref < n_ref ? refs[next_ref] : "*"));
    s += '\t' + std::to_string(next_pos + 1) + '\t' + std::to_string(tlen) + '\t';
    if (l_seq < 0) bail("Error during BAM record parsing: truncated file");
    const uint8_t *sq = r.take(((size_t)l_seq + 1) / 2);
    if (l_seq == 0) {
        s += '*';
    } else {
        const size_t at0 = s.size();
        s.resize(at0 + (size_t)l_seq);
        for (int32_t k = 0; k < l_seq; ++k) s[at0 + k] = kBamSeq[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
    }
    s += '\t';
    const uint8_t *ql = r.take((size_t)l_seq);
    if (l_seq == 0 || ql[0] == 0xFF) {
        s += '*';
    } else {
        const size_t at0 = s.size();
        s.resize(at0 + (size_t)l_seq);
        for (int32_t k = 0; k < l_seq; ++k) s[at0 + k] = (char)(ql[k] + 33);
    }
    aux_to_text(r, s);
}

// BAM header + reference dictionary at d[0,n).  false: the bytes end inside it (take more input).
static bool parse_bam_header(const char *d, uint64_t n, SamFile &out, uint64_t *end) {
    const uint8_t *p = (const uint8_t *)d, *e = p + n;
    auto need = [&](size_t k) { return (size_t)(e - p) >= k; };
    if (!need(8)) return false;
    if (memcmp(p, "BAM\1", 4) != 0) bail("Error reading BAM file: bad magic");
    int32_t l_text;
    memcpy(&l_text, p + 4, 4);
    if (l_text < 0) bail("Error during BAM record parsing: truncated file");
    p += 8;
    if (!need((size_t)l_text + 4)) return false;
    size_t tl = (size_t)l_text;
    while (tl && p[tl - 1] == 0) --tl;
    out.header.assign((const char *)p, tl);
    if (!out.header.empty() && out.header.back() != '\n') out.header += '\n';
    p += l_text;
    int32_t n_ref;
    memcpy(&n_ref, p, 4);
    p += 4;
    out.ref_names.clear();
    out.ref_lens.clear();
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!need(4)) return false;
        int32_t l_name;
        memcpy(&l_name, p, 4);
        if (l_name < 0) bail("Error during BAM record parsing: truncated file");
        if (!need(4 + (size_t)l_name + 4)) return false;
        out.ref_names.emplace_back((const char *)p + 4, l_name > 0 ? (size_t)l_name - 1 : 0);
        uint32_t l_ref;
        memcpy(&l_ref, p + 4 + l_name, 4);
        out.ref_lens.push_back(l_ref);
        p += 4 + (size_t)l_name + 4;
    }
    *end = (uint64_t)((const char *)p - d);
    return true;
}

// Records from d + from on: a chain of block_size fields; the fixed part of each record is validated here
// so that later (parallel) accesses stay inside the record.  Stops after the record that crosses `soft_stop`
// (window size) or in front of a record that does not end inside d[0,n) -- an error unless partial_ok.
// Returns the offset behind the last record taken.
// ---- the BAM record chain on all host threads -------------------------------------------------------------------
// A record's place is only known from the one before it (block_size hops): 8 M hops are 0.1 s per GiB on one thread.
// The chain is walked in pieces instead: every thread but the first FINDS a record start behind its piece's first byte --
// a position where four records in a row have consistent fixed fields (sizes that add up, a printable NUL-terminated name,
// reference ids and positions >= -1) -- and walks from there; the pieces are accepted only if every walk ends EXACTLY on
// the next piece's start, which no false start survives.  Anything else falls back to the serial walk below.
static int64_t bam_record_size_if_plausible(const uint8_t *d, uint64_t n, uint64_t p) {  // 4 + block_size, 0 = not a record, -1 = runs out of d
    if (n - p < 36) return -1;
    int32_t block, ref, pos, l_seq, nref, npos;
    memcpy(&block, d + p, 4), memcpy(&ref, d + p + 4, 4), memcpy(&pos, d + p + 8, 4), memcpy(&l_seq, d + p + 20, 4);
    memcpy(&nref, d + p + 24, 4), memcpy(&npos, d + p + 28, 4);
    if (block < 32 || block > (1 << 28) || ref < -1 || ref > (1 << 24) || pos < -1 || l_seq < 0 || nref < -1 || nref > (1 << 24) || npos < -1) return 0;
    const uint8_t l_name = d[p + 12];
    uint16_t n_cig;
    memcpy(&n_cig, d + p + 16, 2);
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (l_name == 0 || fixed > (uint64_t)block) return 0;
    if (n - p - 4 < (uint64_t)block) return -1;
    const uint8_t *name = d + p + 36;
    if (name[l_name - 1] != 0) return 0;
    for (uint32_t k = 0; k + 1 < l_name; ++k)
        if (name[k] < 33 || name[k] > 126) return 0;
    return 4 + (int64_t)block;
}
// the first position at or behind `from` (and before `stop`) where four plausible records follow each other; ~0 = none
static uint64_t bam_find_record_start(const uint8_t *d, uint64_t n, uint64_t from, uint64_t stop) {
    for (uint64_t p = from; p < stop; ++p) {
        uint64_t q = p;
        int k = 0;
        for (; k < 4; ++k) {
            const int64_t sz = bam_record_size_if_plausible(d, n, q);
            if (sz <= 0) break;
            q += (uint64_t)sz;
        }
        if (k == 4) return p;
    }
    return ~0ull;
}

static uint64_t parse_bam_records_serial(const char *d, uint64_t n, uint64_t from, uint64_t soft_stop, bool partial_ok, SamFile &out);

static uint64_t parse_bam_records(const char *d, uint64_t n, uint64_t from, uint64_t soft_stop, bool partial_ok, SamFile &out) {
    const uint64_t span = soft_stop > from ? soft_stop - from : 0;
    const size_t T = std::min<size_t>(io_threads(), (size_t)(span >> 24));  // pieces of >= 16 MB
    if (T < 2) return parse_bam_records_serial(d, n, from, soft_stop, partial_ok, out);
    std::vector<uint64_t> start(T + 1, ~0ull);
    start[0] = from;
    start[T] = soft_stop;
    run_threads(T - 1, [&](size_t t) {
        const uint64_t c = from + span * (t + 1) / T;
        start[t + 1] = bam_find_record_start((const uint8_t *)d, n, c, std::min<uint64_t>(soft_stop, c + (1u << 20)));
    });
    for (size_t t = 1; t < T; ++t)
        if (start[t] == ~0ull || start[t] <= start[t - 1]) return parse_bam_records_serial(d, n, from, soft_stop, partial_ok, out);
    std::vector<SamFile> part(T);
    std::vector<uint64_t> end(T, 0);
    std::vector<std::string> err(T);
    run_threads(T, [&](size_t t) {
        try {
            // a piece ends where the next one starts; only the last one has the window's own end (and its partial_ok)
            end[t] = parse_bam_records_serial(d, t + 1 < T ? start[t + 1] : n, start[t], start[t + 1], t + 1 < T ? false : partial_ok, part[t]);
        } catch (const Error &e) {
            err[t] = e.what();
        }
    });
    bool ok = true;
    for (size_t t = 0; t < T; ++t) ok = ok && err[t].empty() && (t + 1 == T || end[t] == start[t + 1]);
    if (!ok) return parse_bam_records_serial(d, n, from, soft_stop, partial_ok, out);  // (also words the errors in file order)
    size_t total = out.recs.size();
    for (auto &x : part) total += x.recs.size();
    out.recs.reserve(total);
    for (auto &x : part) out.recs.insert(out.recs.end(), x.recs.begin(), x.recs.end());
    return end[T - 1];
}

static uint64_t parse_bam_records_serial(const char *d, uint64_t n, uint64_t from, uint64_t soft_stop, bool partial_ok, SamFile &out) {
    uint64_t p = from;
    while (p < n && p < soft_stop) {
        if (n - p < 4) {
            if (partial_ok) break;
            bail("Error during BAM record parsing: truncated file");
        }
        int32_t block;
        memcpy(&block, d + p, 4);
        if (block < 32) bail("Error during BAM record parsing: truncated file");
        if (n - p - 4 < (uint64_t)block) {
            if (partial_ok) break;
            bail("Error during BAM record parsing: truncated file");
        }
        const uint8_t *r = (const uint8_t *)d + p + 4;
        const uint8_t l_name = r[8];
        uint16_t n_cig;
        int32_t l_seq;
        memcpy(&n_cig, r + 12, 2);
        memcpy(&l_seq, r + 16, 4);
        const uint64_t fixed = 32ull + l_name + 4ull * n_cig;
        if (l_seq < 0 || fixed + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)block)
            bail("Error during BAM record parsing: truncated file");
        SamFile::Rec rec;
        rec.off = p;
        rec.len = (uint32_t)block + 4;
        rec.name_len = l_name ? l_name - 1u : 0u;
        rec.seq_off = rec.off + 4 + fixed;
        rec.l_seq = (uint32_t)l_seq;
        out.recs.push_back(rec);
        p += 4 + (uint64_t)block;
    }
    return p;
}

void SamFile::open(const std::string &path) {
    const std::string ext = extension(path);
    if (ext.empty()) bail("Could not detect the file extension: \"" + path + "\"");
    if (ext != "sam" && ext != "bam") bail("Input file must be a BAM or SAM file.");
    header.clear();
    recs.clear();
    ref_names.clear();
    ref_lens.clear();
    is_bam = ext == "bam";
    src.open(path);
    cursor = 0;
    if (is_bam) {
        uint64_t end = 0;
        while (!parse_bam_header(bytes(), n_bytes(), *this, &end)) {
            if (src.exhausted()) bail("Error during BAM record parsing: truncated file");
            src.more_into(buf, buf_len, 1u << 16);
        }
        cursor = end;
    } else {
        // header = the '@' lines in front of the first record (SAM spec 1.3); a compressed text needs them
        // complete in the buffer first
        for (;;) {
            const char *d = bytes();
            const uint64_t n = n_bytes();
            uint64_t p = 0;
            bool complete = false;
            header.clear();
            while (p < n) {
                const uint64_t e = line_end(d, n, p);
                if (e >= n && !src.exhausted()) break;  // the line may go on
                const uint64_t le = strip_cr(d, p, e);
                if (le > p && d[p] != '@') {
                    complete = true;
                    break;
                }
                if (le > p) {
                    header.append(d + p, le - p);
                    header += '\n';
                }
                p = std::min(e + 1, n);
            }
            if (complete || p >= n) {
                if (complete || src.exhausted()) {
                    cursor = p;
                    break;
                }
            }
            if (!src.more_into(buf, buf_len, 1u << 16)) {
                cursor = p;
                break;
            }
        }
    }
    data = bytes();
}

void SamFile::seek_bam(size_t member, const char *head, uint64_t n_head) {
    std::vector<char> b(head, head + n_head);  // (head may point into buf)
    buf.swap(b);
    buf_len = n_head;
    cursor = 0;
    recs.clear();
    have_next = false;
    src.seek_member(member);
    data = buf.data();
}

void SamFile::drop_front(uint64_t k) {
    if (src.mapped() || k == 0) return;
    if (k > buf_len) k = buf_len;
    memmove(buf.data(), buf.data() + k, buf_len - k);
    buf_len -= k;
}

// The next records of a compressed input: b[0, bl) holds the text not yet turned into records (it starts at a record /
// line start), more is inflated behind it until ~window_bytes are there, the records are parsed into rs; cur = the
// bytes of b they cover.  false: no record is left.
bool SamFile::next_window(std::vector<char> &b, uint64_t &bl, uint64_t &cur, std::vector<Rec> &rs, uint64_t window_bytes) {
    rs.clear();
    cur = 0;
    for (;;) {
        if (bl < window_bytes) src.more_into(b, bl, window_bytes - bl);
        const char *d = b.data();
        const uint64_t n = bl;
        if (cur >= n && src.exhausted()) return false;
        SamFile part;  // (the parsers append to a SamFile's records)
        part.is_bam = is_bam;
        if (is_bam) {
            cur = parse_bam_records(d, n, cur, std::min<uint64_t>(n, cur + window_bytes), !src.exhausted(), part);
        } else {
            uint64_t stop = std::min(n, cur + window_bytes);
            if (stop < n) stop = std::min(line_end(d, n, stop > 0 ? stop - 1 : 0) + 1, n);  // first line start at or after
            if (stop >= n && !src.exhausted()) {  // the last line of the buffer may go on: complete lines only
                stop = n;
                while (stop > cur && d[stop - 1] != '\n') --stop;
            }
            if (stop > cur) parse_sam_text(d, n, cur, stop, false, part);
            cur = stop;
        }
        if (!part.recs.empty()) {
            rs.swap(part.recs);
            return true;
        }
        if (src.exhausted() && cur >= n) return false;
        if (src.exhausted()) {  // bytes are left that are no record
            if (is_bam) bail("Error during BAM record parsing: truncated file");
            return false;
        }
        src.more_into(b, bl, std::max<uint64_t>(window_bytes, bl));  // one record larger than the window
    }
}

// (compressed input) the window after the current one, prepared beside it: the unconsumed tail of the current window
// opens a second buffer, the inflate (device codec or host threads) and the record index run on the calling thread while
// the current window's `data` / `recs` stay as they are; the next fill() swaps the result in
void SamFile::prefetch(uint64_t window_bytes) {
    if (src.mapped() || have_next) return;
    if (window_bytes < (1u << 16)) window_bytes = 1u << 16;
    const uint64_t tail = buf_len > cursor ? buf_len - cursor : 0;
    if (nbuf.size() < tail) nbuf.resize(tail);
    if (tail) memcpy(nbuf.data(), buf.data() + cursor, tail);
    nbuf_len = tail;
    next_more = next_window(nbuf, nbuf_len, ncursor, nrecs, window_bytes);
    have_next = true;
}

bool SamFile::fill(uint64_t window_bytes) {
    if (window_bytes < (1u << 16)) window_bytes = 1u << 16;
    if (!src.mapped()) {  // compressed: drop what became records, take more
        if (have_next) {  // prefetch() has done that already
            have_next = false;
            buf.swap(nbuf);
            buf_len = nbuf_len;
            cursor = ncursor;
            recs.swap(nrecs);
            nrecs.clear();
            data = buf.data();
            return next_more;
        }
        drop_front(cursor);
        const bool more = next_window(buf, buf_len, cursor, recs, window_bytes);
        data = buf.data();
        return more;
    }
    recs.clear();
    for (;;) {
        const char *d = src.text();
        const uint64_t n = src.text_size();
        data = d;
        if (cursor >= n) return false;
        if (is_bam) {
            cursor = parse_bam_records(d, n, cursor, std::min<uint64_t>(n, cursor + window_bytes), false, *this);
        } else {
            uint64_t stop = std::min(n, cursor + window_bytes);
            if (stop < n) stop = std::min(line_end(d, n, stop > 0 ? stop - 1 : 0) + 1, n);  // first line start at or after
            if (stop > cursor) parse_sam_text(d, n, cursor, stop, false, *this);
            cursor = stop;
        }
        if (!recs.empty()) return true;
        // a window without records (blank lines): next one
    }
}

void SamFile::gather(size_t b0, size_t b1, std::vector<uint8_t> &seq, std::vector<uint64_t> &off) const {
    const size_t n = b1 - b0;
    off.resize(n + 1);
    off[0] = 0;
    for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + recs[b0 + i].l_seq;
    seq.resize(off[n] + 1);
    const size_t T = std::max<size_t>(1, std::min<size_t>(io_threads(), n / 65536 + 1));
    run_threads(T, [&](size_t t) {
        for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) {
            uint8_t *o = seq.data() + off[i];
            const uint32_t l = recs[b0 + i].l_seq;
            const uint8_t *src = (const uint8_t *)data + recs[b0 + i].seq_off;
            if (is_bam) {
                for (uint32_t k = 0; k < l; ++k) o[k] = (uint8_t)kBamSeq[(src[k >> 1] >> ((~k & 1) << 2)) & 15];
            } else {  // the matcher sees upper-case text (bam crate: sequences are stored packed, case is lost)
                for (uint32_t k = 0; k < l; ++k) o[k] = (src[k] >= 'a' && src[k] <= 'z') ? (uint8_t)(src[k] & ~0x20) : src[k];
            }
        }
    });
    seq[off[n]] = 0;
}

void SamFile::append_line(size_t i, std::string &out) const {
    if (is_bam)
        bam_record_to_sam((const uint8_t *)data + recs[i].off, ref_names, out);
    else
        out.append(data + recs[i].off, recs[i].len);
}

int SamFile::find_tag(size_t i, const std::string &tag, std::string *val) const {
    if (!is_bam) {
        const char *b = data + recs[i].off, *e = b + recs[i].len;
        for (int f = 0; f < 11; ++f) {
            b = (const char *)memchr(b, '\t', e - b);
            if (!b) return 0;
            ++b;
        }
        while (b < e) {
            const char *t = (const char *)memchr(b, '\t', e - b);
            if (!t) t = e;
            if (t - b >= 5 && b[0] == tag[0] && b[1] == tag[1] && b[2] == ':') {
                if (b[3] != 'Z' || b[4] != ':') return 2;
                val->assign(b + 5, t - (b + 5));
                return 1;
            }
            b = t + 1;
        }
        return 0;
    }
    // BAM: walk the optional fields behind name, CIGAR, SEQ and QUAL
    const uint8_t *r = raw(i);
    Cur c{r + (recs[i].seq_off - recs[i].off - 4) + ((size_t)recs[i].l_seq + 1) / 2 + recs[i].l_seq, r + raw_len(i)};
    while (c.p < c.e) {
        const uint8_t *tg = c.take(2);
        const char type = (char)c.get<uint8_t>();
        const bool mine = tg[0] == (uint8_t)tag[0] && tg[1] == (uint8_t)tag[1];
        if (mine && type != 'Z') return 2;
        switch (type) {
        case 'A': case 'c': case 'C': c.take(1); break;
        case 's': case 'S': c.take(2); break;
        case 'i': case 'I': case 'f': c.take(4); break;
        case 'Z': case 'H': {
            const uint8_t *s0 = c.p;
            while (c.p < c.e && *c.p) ++c.p;
            if (mine) {
                val->assign((const char *)s0, c.p - s0);
                return 1;
            }
            if (c.p < c.e) ++c.p;
            break;
        }
        case 'B': {
            const char sub = (char)c.get<uint8_t>();
            const int32_t cnt = c.get<int32_t>();
            const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (cnt < 0) bail("Error during BAM record parsing: truncated file");
            c.take(w * (size_t)cnt);
            break;
        }
        default: bail("Error during BAM record parsing: unknown tag type");
        }
    }
    return 0;
}

// ---- BAM output ---------------------------------------------------------------------------------
BamWriter::~BamWriter() {
    try {
        close();  // (the command closes the writer itself: an error surfaces there, not here)
    } catch (...) {
    }
}

void BamWriter::use_device(int device) { device_ = mk_bgzf_deflate_pieces ? device : -1; }

constexpr size_t kBgzfBlock = 0xff00;

void BamWriter::put(const void *p, size_t n) {  // small writes (header, single records) share an open piece
    if (pieces.empty() || pieces.back().size() + n > pieces.back().capacity()) {
        pieces.emplace_back();
        pieces.back().reserve(std::max<size_t>(n, 1u << 20));
    }
    pieces.back().insert(pieces.back().end(), (const uint8_t *)p, (const uint8_t *)p + n);
    pieces_bytes += n;
    if (pieces_bytes >= (run_members + 1) * kBgzfBlock) flush(false);
}

std::vector<uint8_t> BamWriter::take_buffer() {
    std::lock_guard<std::mutex> lk(mu_);
    if (free_.empty()) return {};
    std::vector<uint8_t> b = std::move(free_.back());
    free_.pop_back();
    return b;
}

void BamWriter::put_encoded(std::vector<uint8_t> &&bytes) {
    if (bytes.empty()) return;
    pieces_bytes += bytes.size();
    pieces.push_back(std::move(bytes));
    if (pieces_bytes >= (run_members + 1) * kBgzfBlock) flush(false);
}

constexpr size_t kMaxQueuedRuns = 6;

void BamWriter::free_raw_buffer(RawBuffer &b) {
    if (b.p) {
        if (b.pinned) mk_host_free(b.p);
        else free(b.p);
    }
    b = RawBuffer();
}

BamWriter::RawBuffer BamWriter::take_raw_buffer(size_t min_size) {
    RawBuffer b;
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t k = 0; k < free_raw_.size(); ++k)
            if (free_raw_[k].cap >= min_size || k + 1 == free_raw_.size()) {
                b = free_raw_[k];
                free_raw_.erase(free_raw_.begin() + k);
                break;
            }
    }
    if (b.cap >= min_size) return b;
    free_raw_buffer(b);
    const size_t want = min_size + min_size / 8;
    void *q = nullptr;
    if (mk_host_alloc && mk_host_alloc(want, &q) == MK_OK) {  // (weak: the CPU harnesses link without the library)
        b.p = (uint8_t *)q, b.cap = want, b.pinned = true;
    } else {
        b.p = (uint8_t *)malloc(want), b.cap = want, b.pinned = false;
        if (!b.p) bail("Error writing BAM file: out of memory");
    }
    return b;
}

void BamWriter::put_members(RawBuffer buffer, size_t used) {
    flush(true);
    Run run;
    run.raw = true;
    run.raw_buf = buffer;
    run.raw_used = used;
    std::unique_lock<std::mutex> lk(mu_);
    if (failed_) {
        free_raw_buffer(buffer);
        std::rethrow_exception(failed_);
    }
    if (!writer_.joinable()) writer_ = std::thread([this] { writer_loop(); });
    cv_.wait(lk, [&] { return queue_.size() < kMaxQueuedRuns; });
    queue_.push_back(std::move(run));
    lk.unlock();
    cv_.notify_all();
}

// one BGZF member: gzip header with the BC extra field, raw deflate, crc32, isize
static void bgzf_compress(const uint8_t *in, size_t n, std::vector<uint8_t> &out) {
    out.resize(n + n / 8 + 1024);
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) bail("Error writing BAM file: zlib init failed");
    z.next_in = const_cast<Bytef *>(in);
    z.avail_in = (uInt)n;
    z.next_out = out.data() + 18;
    z.avail_out = (uInt)(out.size() - 26);
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) bail("Error writing BAM file: deflate failed");
    const size_t clen = z.total_out;
    deflateEnd(&z);
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data(), hdr, 16);
    const uint16_t bsize = (uint16_t)(clen + 25);  // total block size - 1
    memcpy(out.data() + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), in, (uInt)n);
    const uint32_t isize = (uint32_t)n;
    memcpy(out.data() + 18 + clen, &crc, 4);
    memcpy(out.data() + 22 + clen, &isize, 4);
    out.resize(clen + 26);
}


void BamWriter::writer_loop() {
    for (;;) {
        Run run;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !queue_.empty() || closing_; });
            if (queue_.empty()) return;
            run = std::move(queue_.front());
            queue_.pop_front();
            busy_ = true;
        }
        cv_.notify_all();
        try {
            if (run.raw) {
                const bool ok = failed_ || fwrite(run.raw_buf.p, 1, run.raw_used, f) == run.raw_used;
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (free_raw_.size() < 4) {
                        free_raw_.push_back(run.raw_buf);
                        run.raw_buf = RawBuffer();
                    }
                }
                free_raw_buffer(run.raw_buf);
                if (!ok) bail("Error writing BAM file");
            } else if (!failed_) {
                compress_and_write(run);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!failed_) failed_ = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            busy_ = false;
        }
        cv_.notify_all();
    }
}

// whole members of the stream so far (all of it at close) go to the writer thread; the bytes behind the last whole
// member -- less than one member -- stay, as the first piece of what follows
void BamWriter::flush(bool all) {
    const size_t take = all ? pieces_bytes : pieces_bytes / kBgzfBlock * kBgzfBlock;
    if (!take) return;
    Run run;
    std::vector<uint8_t> rest;
    size_t at = 0;
    for (auto &p : pieces) {
        if (at + p.size() <= take) {
            at += p.size();
            run.push_back(std::move(p));
        } else {
            const size_t keep = at < take ? take - at : 0;  // bytes of this piece that belong to the run
            rest.insert(rest.end(), p.begin() + keep, p.end());
            at += p.size();
            if (keep) {
                p.resize(keep);
                run.push_back(std::move(p));
            }
        }
    }
    pieces.clear();
    pieces_bytes = rest.size();
    if (!rest.empty()) pieces.push_back(std::move(rest));
    std::unique_lock<std::mutex> lk(mu_);
    if (failed_) std::rethrow_exception(failed_);
    if (!writer_.joinable()) writer_ = std::thread([this] { writer_loop(); });
    cv_.wait(lk, [&] { return queue_.size() < kMaxQueuedRuns; });
    queue_.push_back(std::move(run));
    lk.unlock();
    cv_.notify_all();
}

void BamWriter::compress_and_write(Run &run_) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t total = 0;
    for (auto &p : run_) total += p.size();
    if (device_ >= 0) {  // the device codec: every member of the run in one call, the pieces joined on the device
        if (!codec_) {
            mk_codec *c = nullptr;
            if (mk_codec_create(device_, &c) != MK_OK) bail(std::string("Error writing BAM file: ") + mk_last_error());
            codec_ = c;
        }
        std::vector<const uint8_t *> ptr(run_.size());
        std::vector<uint64_t> len(run_.size());
        for (size_t k = 0; k < run_.size(); ++k) ptr[k] = run_[k].data(), len[k] = run_[k].size();
        z_.resize(mk_bgzf_deflate_bound(total, (uint32_t)kBgzfBlock));
        uint64_t zn = 0;
        if (mk_bgzf_deflate_pieces((mk_codec *)codec_, ptr.data(), len.data(), ptr.size(), (uint32_t)kBgzfBlock, z_.data(), z_.size(), &zn) != MK_OK)
            bail(std::string("Error writing BAM file: ") + mk_last_error());
        deflate_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (fwrite(z_.data(), 1, zn, f) != zn) bail("Error writing BAM file");
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto &p : run_)
                if (p.capacity() >= (1u << 20) && free_.size() < 64) {
                    p.clear();
                    free_.push_back(std::move(p));
                }
        }
        run_.clear();
        return;
    }
    // zlib on every host thread (deflate is the slow side of BAM output: ~50 MB/s per core); members written in order
    flat_.clear();
    flat_.reserve(total);
    for (auto &p : run_) flat_.insert(flat_.end(), p.begin(), p.end());
    run_.clear();
    const std::vector<uint8_t> &run = flat_;
    const size_t blocks = (run.size() + kBgzfBlock - 1) / kBgzfBlock;
    std::vector<std::vector<uint8_t>> outs(blocks);
    const size_t T = std::min<size_t>(io_threads(), blocks);
    std::vector<std::string> errs(T);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            try {
                for (size_t i = t; i < blocks; i += T)
                    bgzf_compress(run.data() + i * kBgzfBlock, std::min(kBgzfBlock, run.size() - i * kBgzfBlock), outs[i]);
            } catch (const Error &e) {
                errs[t] = e.what();
            }
        });
    for (auto &x : th) x.join();
    for (auto &e : errs)
        if (!e.empty()) bail(e);
    deflate_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto &o : outs)
        if (fwrite(o.data(), 1, o.size(), f) != o.size()) bail("Error writing BAM file");
}

void BamWriter::open(const std::string &path, const std::string &header_text, const std::vector<std::string> *names,
                     const std::vector<uint32_t> *lens) {
    f = fopen(path.c_str(), "wb");
    if (!f) bail("Error writing BAM file: " + path);
    std::vector<uint32_t> ref_len;
    if (names && lens) {
        ref_names = *names;
        ref_len = *lens;
    }
    size_t b = names && lens ? header_text.size() : 0;
    while (b < header_text.size()) {  // @SQ SN:name LN:len
        size_t e = header_text.find('\n', b);
        if (e == std::string::npos) e = header_text.size();
        const std::string ln = header_text.substr(b, e - b);
        if (ln.rfind("@SQ", 0) == 0) {
            std::string sn;
            uint32_t lnv = 0;
            size_t q = 0;
            while (q < ln.size()) {
                size_t t = ln.find('\t', q);
                if (t == std::string::npos) t = ln.size();
                if (ln.compare(q, 3, "SN:") == 0) sn = ln.substr(q + 3, t - q - 3);
                if (ln.compare(q, 3, "LN:") == 0) lnv = (uint32_t)strtoul(ln.c_str() + q + 3, nullptr, 10);
                q = t + 1;
            }
            ref_names.push_back(sn);
            ref_len.push_back(lnv);
        }
        b = e + 1;
    }
    put("BAM\1", 4);
    const int32_t l_text = (int32_t)header_text.size();
    put(&l_text, 4);
    put(header_text.data(), header_text.size());
    const int32_t n_ref = (int32_t)ref_names.size();
    put(&n_ref, 4);
    for (int32_t i = 0; i < n_ref; ++i) {
        const int32_t l_name = (int32_t)ref_names[i].size() + 1;
        put(&l_name, 4);
        put(ref_names[i].c_str(), (size_t)l_name);
        put(&ref_len[i], 4);
    }
}

static int reg2bin(int64_t beg, int64_t end) {  // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

void BamWriter::write_record(const std::string &line) {
    std::vector<uint8_t> r;
    encode_record(line, r);
    put(r.data(), r.size());
}

// (r04: no allocation per record or field -- fields are (pointer, length) views of the line, the record is built in
// place behind dst, bases go through a 256-entry table; 8 M records: 0.66 -> 0.2 s of SAM -> BAM on 16 threads)
namespace {
struct Field {
    const char *p;
    size_t n;
    bool is(const char *lit) const { return n == strlen(lit) && memcmp(p, lit, n) == 0; }
};
// strtol / strtoul / strtoll on a field that is not NUL-terminated: optional blanks and sign, digits up to the first other byte
long long field_int(const char *p, size_t n) {
    size_t i = 0;
    while (i < n && (p[i] == ' ' || p[i] == '\t')) ++i;
    bool neg = false;
    if (i < n && (p[i] == '+' || p[i] == '-')) neg = p[i++] == '-';
    unsigned long long v = 0;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) v = v * 10 + (unsigned)(p[i] - '0');
    return neg ? -(long long)v : (long long)v;
}
float field_float(const char *p, size_t n) {
    char tmp[64];
    const size_t k = n < sizeof(tmp) - 1 ? n : sizeof(tmp) - 1;
    memcpy(tmp, p, k);
    tmp[k] = 0;
    return strtof(tmp, nullptr);
}
struct NibbleTable {
    uint8_t code[256];
    NibbleTable() {
        memset(code, 15, sizeof(code));
        static const char kSeq[] = "=ACMGRSVTWYHKDBN";
        for (int k = 0; k < 16; ++k) {
            code[(uint8_t)kSeq[k]] = (uint8_t)k;
            if (kSeq[k] >= 'A' && kSeq[k] <= 'Z') code[(uint8_t)(kSeq[k] + 32)] = (uint8_t)k;
        }
    }
};
const NibbleTable kNibble;
template <class T>
void put_le(std::vector<uint8_t> &r, T v) {
    r.insert(r.end(), (const uint8_t *)&v, (const uint8_t *)&v + sizeof(T));
}
}  // namespace

void BamWriter::encode_record(const std::string &line, std::vector<uint8_t> &dst) const {
    Field fld[11];
    // optional fields: the first 32 in place (every tagged record has at least one: a local vector allocated per record),
    // a vector only for a record with more
    Field opt[32];
    std::vector<Field> opt_more;
    size_t n_opt = 0;
    size_t nf = 0;
    {
        const char *d = line.data();
        const size_t n = line.size();
        size_t b = 0;
        for (;;) {
            const void *t = b <= n ? memchr(d + b, '\t', n - b) : nullptr;
            const size_t e = t ? (size_t)((const char *)t - d) : n;
            if (nf < 11) fld[nf] = Field{d + b, e - b};
            else if (n_opt < 32) opt[n_opt++] = Field{d + b, e - b};
            else opt_more.push_back(Field{d + b, e - b}), ++n_opt;
            ++nf;
            if (!t) break;
            b = e + 1;
        }
    }
    if (nf < 11) bail("Error writing record to output file: too few SAM fields");
    if (fld[0].n > 254) bail("Error writing record to output file: read name longer than 254 bytes");  // l_read_name is one byte
    auto ref_id = [&](const Field &f) -> int32_t {
        if (f.is("*")) return -1;
        for (size_t i = 0; i < ref_names.size(); ++i)
            if (ref_names[i].size() == f.n && memcmp(ref_names[i].data(), f.p, f.n) == 0) return (int32_t)i;
        return -1;
    };
    const size_t start = dst.size();
    std::vector<uint8_t> &r = dst;  // built in place: block_size is patched in at the end
    put_le<int32_t>(r, 0);
    const int32_t rid = ref_id(fld[2]), pos = (int32_t)field_int(fld[3].p, fld[3].n) - 1;
    const uint8_t mapq = (uint8_t)field_int(fld[4].p, fld[4].n);
    const uint16_t flag = (uint16_t)field_int(fld[1].p, fld[1].n);
    uint32_t cig[64];
    std::vector<uint32_t> cig_more;
    size_t n_cig_ops = 0;
    int64_t ref_span = 0;
    if (!fld[5].is("*")) {
        const char *c = fld[5].p, *const ce = c + fld[5].n;
        while (c < ce) {
            uint32_t len = 0;
            while (c < ce && *c >= '0' && *c <= '9') len = len * 10 + (uint32_t)(*c++ - '0');
            static const char ops[] = "MIDNSHP=X";
            const char *o = c < ce ? (const char *)memchr(ops, *c, 9) : nullptr;
            if (!o) bail("Error writing record to output file: bad CIGAR");
            const uint32_t op = (uint32_t)(o - ops);
            if (n_cig_ops < 64) cig[n_cig_ops] = len << 4 | op;
            else cig_more.push_back(len << 4 | op);
            ++n_cig_ops;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_span += len;
            ++c;
        }
    }
    const Field &seq = fld[9];
    const int32_t l_seq = seq.is("*") ? 0 : (int32_t)seq.n;
    const int32_t next_rid = fld[6].is("=") ? rid : ref_id(fld[6]);
    const int32_t next_pos = (int32_t)field_int(fld[7].p, fld[7].n) - 1, tlen = (int32_t)field_int(fld[8].p, fld[8].n);
    const uint8_t l_name = (uint8_t)(fld[0].n + 1);
    const uint16_t bin = (uint16_t)reg2bin(pos, pos + (ref_span ? ref_span : 1)), n_cig = (uint16_t)n_cig_ops;
    put_le(r, rid), put_le(r, pos), put_le(r, l_name), put_le(r, mapq), put_le(r, bin), put_le(r, n_cig), put_le(r, flag);
    put_le(r, l_seq), put_le(r, next_rid), put_le(r, next_pos), put_le(r, tlen);
    r.insert(r.end(), (const uint8_t *)fld[0].p, (const uint8_t *)fld[0].p + (l_name - 1));
    r.push_back(0);
    for (size_t k = 0; k < n_cig_ops; ++k) put_le<uint32_t>(r, k < 64 ? cig[k] : cig_more[k - 64]);
    {
        const size_t at = r.size();
        r.resize(at + (size_t)(l_seq + 1) / 2 + (size_t)l_seq);
        uint8_t *o = r.data() + at;
        const uint8_t *sq = (const uint8_t *)seq.p;
        int32_t k = 0;
        for (; k + 1 < l_seq; k += 2) *o++ = (uint8_t)(kNibble.code[sq[k]] << 4 | kNibble.code[sq[k + 1]]);
        if (k < l_seq) *o++ = (uint8_t)(kNibble.code[sq[k]] << 4);
        if (fld[10].is("*")) {
            memset(o, 0xFF, (size_t)l_seq);
        } else {
            if ((int32_t)fld[10].n != l_seq) bail("Error writing record to output file: SEQ and QUAL lengths differ");
            const uint8_t *q = (const uint8_t *)fld[10].p;
            for (int32_t i = 0; i < l_seq; ++i) o[i] = (uint8_t)(q[i] - 33);
        }
    }
    for (size_t ti = 0; ti < n_opt; ++ti) {  // TAG:TYPE:VALUE
        const Field &t = ti < 32 ? opt[ti] : opt_more[ti - 32];
        if (t.n < 5 || t.p[2] != ':' || t.p[4] != ':') bail("Error writing record to output file: bad optional field");
        r.insert(r.end(), (const uint8_t *)t.p, (const uint8_t *)t.p + 2);
        const char type = t.p[3];
        const char *v = t.p + 5;
        const size_t vn = t.n - 5;
        if (type == 'A') {
            r.push_back('A');
            r.push_back(vn == 0 ? 0 : (uint8_t)v[0]);
        } else if (type == 'i') {  // smallest integer type that holds the value, like samtools
            const long long x = field_int(v, vn);
            if (x >= 0) {
                if (x <= 0xff) r.push_back('C'), put_le(r, (uint8_t)x);
                else if (x <= 0xffff) r.push_back('S'), put_le(r, (uint16_t)x);
                else r.push_back('I'), put_le(r, (uint32_t)x);
            } else {
                if (x >= -128) r.push_back('c'), put_le(r, (int8_t)x);
                else if (x >= -32768) r.push_back('s'), put_le(r, (int16_t)x);
                else r.push_back('i'), put_le(r, (int32_t)x);
            }
        } else if (type == 'f') {
            r.push_back('f');
            put_le(r, field_float(v, vn));
        } else if (type == 'Z' || type == 'H') {
            r.push_back((uint8_t)type);
            r.insert(r.end(), (const uint8_t *)v, (const uint8_t *)v + vn);
            r.push_back(0);
        } else if (type == 'B') {
            r.push_back('B');
            const char sub = vn == 0 ? 'c' : v[0];
            r.push_back((uint8_t)sub);
            const size_t cnt_at = r.size();
            put_le<int32_t>(r, 0);
            int32_t cnt = 0;
            size_t q = 1;
            while (q < vn) {  // ",a,b,c"
                const void *c = memchr(v + q + 1, ',', vn - q - 1);
                const size_t e = c ? (size_t)((const char *)c - v) : vn;
                const char *it = v + q + 1;
                const size_t in = e - q - 1;
                switch (sub) {
                case 'c': put_le(r, (int8_t)field_int(it, in)); break;
                case 'C': put_le(r, (uint8_t)field_int(it, in)); break;
                case 's': put_le(r, (int16_t)field_int(it, in)); break;
                case 'S': put_le(r, (uint16_t)field_int(it, in)); break;
                case 'i': put_le(r, (int32_t)field_int(it, in)); break;
                case 'I': put_le(r, (uint32_t)field_int(it, in)); break;
                case 'f': put_le(r, field_float(it, in)); break;
                default: bail("Error writing record to output file: bad B array subtype");
                }
                ++cnt;
                if (!c) break;
                q = e;
            }
            memcpy(r.data() + cnt_at, &cnt, 4);
        } else {
            bail("Error writing record to output file: unknown optional field type");
        }
    }
    const int32_t block_size = (int32_t)(r.size() - start - 4);
    memcpy(r.data() + start, &block_size, 4);
}

void BamWriter::close() {
    if (!f) return;
    std::exception_ptr err;
    try {
        flush(true);
    } catch (...) {
        err = std::current_exception();
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        closing_ = true;
    }
    cv_.notify_all();
    if (writer_.joinable()) writer_.join();
    if (!err && failed_) err = failed_;
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!err) fwrite(eof, 1, sizeof(eof), f);
    fclose(f);
    f = nullptr;
    if (codec_) mk_codec_destroy((mk_codec *)codec_), codec_ = nullptr;
    for (auto &b : free_raw_) free_raw_buffer(b);
    free_raw_.clear();
    closing_ = false;
    failed_ = nullptr;
    if (err) std::rethrow_exception(err);
}

void BamWriter::append_tagged_raw(const uint8_t *rec, uint32_t len, const std::string &tag, const char *val, size_t val_len,
                                  std::vector<uint8_t> &dst) {
    const int32_t block_size = (int32_t)(len + 3 + val_len + 1);
    dst.insert(dst.end(), (const uint8_t *)&block_size, (const uint8_t *)&block_size + 4);
    dst.insert(dst.end(), rec, rec + len);
    dst.push_back((uint8_t)tag[0]);
    dst.push_back((uint8_t)tag[1]);
    dst.push_back('Z');
    dst.insert(dst.end(), (const uint8_t *)val, (const uint8_t *)val + val_len);
    dst.push_back(0);
}

}  // namespace cli
