// io.hpp -- record readers/writers of the `merkurio` CLI (host side, either side of the hot
// path).  Observable behaviour follows what the reference gets from its crates
// (SURVEY.md §5): needletail 0.6.3 for FASTA/FASTQ (src/cmd_extract.rs:281,327-340,403) and
// bam 0.1.4 for SAM/BAM (src/cmd_tag.rs:470-497,503-615).
#pragma once
#include <cstdint>
#include <condition_variable>
#include <deque>
#include <exception>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "util.hpp"

namespace cli {

// host threads for ingest / codec work: hardware concurrency, capped at 32, MERKURIO_IO_THREADS overrides
unsigned io_threads();
// `tag -p N` given explicitly: at most N host threads for the codec work (the reference's meaning of -p:
// BAM (de)compression threads, src/cmd_tag.rs:102-104,506); without it every core the process may use
void set_io_threads_cap(unsigned n);
// BGZF input (BAM, bgzip'ed FASTA/FASTQ) is inflated by the device codec on this HIP device from now on (windows of at
// least a few hundred members; < 0: zlib on the host threads); seconds spent in those calls so far
void set_bgzf_device(int device, bool always = false);
double bgzf_device_seconds();

// runs fn(t) for t in [0, T) on T host threads; the first cli::Error is re-raised on the caller
template <class F>
void run_threads(size_t T, F fn) {
    if (T <= 1) {
        fn((size_t)0);
        return;
    }
    std::vector<std::string> errs(T);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            try {
                fn(t);
            } catch (const Error &e) {
                errs[t] = e.what()[0] ? e.what() : "error";
            }
        });
    for (auto &x : th) x.join();
    for (auto &e : errs)
        if (!e.empty()) bail(e);
}

// whole file into memory; transparently inflates gzip (magic 1f 8b; BGZF members in parallel) and,
// like needletail's `compression` feature, bzip2 / xz / zstd (decompress.cpp)
std::vector<char> read_file_maybe_gz(const std::string &path);
// raises cli::Error if `need` bytes of host memory are not available (decompressed inputs are held in memory)
void require_host_memory(uint64_t need, const std::string &path);
// bzip2 / xz / zstd by magic bytes through the system's runtime libraries (bound with dlopen):
// false = none of the three; raises cli::Error on corrupt input or a missing library
bool inflate_by_magic(const std::string &path, const unsigned char *data, size_t n, std::vector<char> &out);

// ---- FASTA / FASTQ ------------------------------------------------------------------------------
// a whole input file in memory: mmap for plain files, an inflated copy for gzip
struct FileBytes {
    const char *p = nullptr;
    uint64_t n = 0;
    std::vector<char> owned;  // gzip path
    void *map = nullptr;      // mmap path
    uint64_t map_len = 0;
    ~FileBytes();
    void load(const std::string &path);
    FileBytes() = default;
    FileBytes(const FileBytes &) = delete;
    FileBytes &operator=(const FileBytes &) = delete;
};

struct FastxFile {
    FileBytes file;
    const char *data = nullptr;  // = file.p (FastxStream: the current window)
    uint64_t data_n = 0;         // bytes behind `data`
    bool fastq = false;
    struct Rec {
        uint64_t id_b, id_e;    // header line without the marker ('>' / '@') and line end
        uint64_t raw_b, raw_e;  // sequence as stored, inner line breaks included (FASTA) -- what write() emits
        uint64_t qual_b, qual_e;
    };
    std::vector<Rec> recs;
    std::string id(size_t i) const { return std::string(data + recs[i].id_b, recs[i].id_e - recs[i].id_b); }
    // record.seq(): newline-free sequence appended to `out`; returns num_bases()
    uint64_t append_seq(size_t i, std::vector<uint8_t> &out) const;
    // sequences of records [b0, b1) concatenated into seq (+1 pad byte) with offsets off[0..b1-b0];
    // FASTQ records are copied by several host threads
    void gather(size_t b0, size_t b1, std::vector<uint8_t> &seq, std::vector<uint64_t> &off) const;
    // number of sequence bytes of record i as stored (upper bound of num_bases)
    uint64_t raw_len(size_t i) const { return recs[i].raw_e - recs[i].raw_b; }
    // record.write(writer, None): FASTA keeps the original wrapping, FASTQ is 4 lines
    void write(size_t i, Sink &w) const;
    // appends the records of data[b, e) (b at a record start) to recs
    void parse_span(uint64_t b, uint64_t e);
    // the same for text that may end inside a record (partial_ok: more text follows data[0, data_n)): parsing stops in front of a
    // record that cannot be shown to end in it; returns where it stopped
    uint64_t parse_span_partial(uint64_t b, uint64_t e, bool partial_ok);
};

// An input file as a sequence of (decompressed) bytes that is looked at through a sliding window, so that
// the host holds one window at a time, like the reference, which streams records.  Plain files are
// memory-mapped: the "window" is the whole mapping and costs nothing.  gzip is inflated incrementally
// (zlib streaming, concatenated members included), BGZF a group of members at a time on all host
// threads; bzip2 / xz / zstd inputs are inflated whole (their libraries are bound through dlopen without
// any block index) and then behave like plain text.
struct WindowSource {
    void open(const std::string &path);
    bool mapped() const { return kind == PLAIN; }  // text()/text_size() is the complete text
    bool is_file_mapping() const { return kind == PLAIN && src.map != nullptr; }  // (not an inflated bzip2 / xz / zstd copy)
    const char *text() const { return src.p; }
    uint64_t text_size() const { return src.n; }
    // compressed kinds: appends up to ~want more inflated bytes to dst[0,len) (some progress unless the
    // source is exhausted); false if it already was
    bool more_into(std::vector<char> &dst, uint64_t &len, uint64_t want);
    bool exhausted() const { return kind == PLAIN || src_eof; }
    const std::string &name() const { return path; }
    // BGZF: the file as stored and its member table, for a caller that inflates on the device (extract's
    // mk_extract_fastq_bgzf path); seek_member() makes member i the next one more_into() inflates
    bool is_bgzf() const { return kind == BGZF; }
    bool is_gzip() const { return kind == GZIP; }  // one or more plain gzip members: file_bytes() is the file as stored
    const uint8_t *file_bytes() const { return (const uint8_t *)src.p; }
    uint64_t file_size() const { return src.n; }
    size_t n_bgzf_members() const { return members.size(); }
    void bgzf_member_at(size_t i, uint64_t *data_off, uint32_t *data_len, uint32_t *isize, uint32_t *crc) const {
        *data_off = members[i].data_off, *data_len = (uint32_t)members[i].data_len, *isize = members[i].isize, *crc = members[i].crc;
    }
    void seek_member(size_t i) {
        src_pos = i;
        src_eof = i >= members.size();
    }
    size_t next_member() const { return (size_t)src_pos; }  // BGZF: the first member more_into() has not inflated yet
    ~WindowSource();
    WindowSource() = default;
    WindowSource(const WindowSource &) = delete;
    WindowSource &operator=(const WindowSource &) = delete;

   private:
    enum Kind { PLAIN, GZIP, BGZF } kind = PLAIN;
    std::string path;
    FileBytes src;  // the file as stored (PLAIN: the text itself, or inflated bzip2 / xz / zstd)
    bool src_eof = false;
    void *zs = nullptr;    // GZIP: z_stream
    uint64_t src_pos = 0;  // GZIP: next compressed byte; BGZF: next member
    struct Member { uint64_t data_off, data_len; uint32_t isize, crc; };
    std::vector<Member> members;
};

// A FASTA/FASTQ input read window by window (needletail's parse_fastx_file streams records:
// src/cmd_extract.rs:281,321), with the next window parsed (and inflated) while the current one is in use:
//     bool more = s.fill(W);
//     while (more) { n = s.view.recs.size(); s.consume(n); s.prefetch(W) on another thread;
//                    ...records [0, n) of s.view...; join; more = s.fill(W); }
struct FastxStream {
    FastxFile view;  // the current window: view.data / view.recs / view.fastq; all FastxFile accessors work on it
    void open(const std::string &path) { src.open(path); }
    // Makes the next window current: every complete record among the unconsumed bytes plus up to
    // `window_bytes` new ones (parsed here unless prefetch() already did).  false: no record is left.
    bool fill(uint64_t window_bytes);
    // The first n records of the current window are all this window hands out; the next window starts
    // with record n.  The records stay valid until the next fill().
    void consume(size_t n);
    // Parses the window behind the consumed records into a spare index (and, for compressed input, a second
    // buffer); may run on another thread while `view` is being used.  Call after consume().
    void prefetch(uint64_t window_bytes);
    // RAW windows (extract_windows.cpp: a window's text goes to the GPU as it is and is indexed there, mk_extract_window): the
    // next ~window_bytes of text, from a record start to a record start, NOT parsed; FASTQ ('@' records) or FASTA ('>' records),
    // whichever the first record is.  false: no text is left, or it does not start like a record (nothing consumed).  The text
    // stays valid until the raw_fill() after next; raw_consume() accepts the window, without it the next fill() / raw_fill()
    // starts at the same place.  Use before any fill().
    bool raw_fill(uint64_t window_bytes, const char **text, uint64_t *n, uint64_t *resume);
    void raw_consume();
    // everything that is left of the input, as it is (after raw_fill() returned false: text that does not start like a record);
    // consumed.  false: nothing is left
    bool raw_rest(const char **text, uint64_t *n);
    // the raw windows are slices of a plain (memory-mapped) file: window = file[resume - n, resume)
    bool raw_is_plain() const { return src.mapped() && src.is_file_mapping(); }
    // after the first raw_fill(): the input is FASTQ ('@' records) or FASTA ('>' records)
    bool raw_fastq() const { return fastq; }
    // bgzip'ed input: the caller walks source()'s members itself and sends them to the device as they are
    bool raw_is_bgzf() const { return src.is_bgzf(); }
    bool raw_is_gzip() const { return src.is_gzip(); }
    const WindowSource &source() const { return src; }

   private:
    WindowSource src;
    uint64_t cursor = 0;  // mapped text: first byte of the next window; compressed: offset of the unconsumed tail in bufs[cur]
    bool started = false, fastq = false;
    std::vector<char> bufs[2];  // compressed input: the current window lives in bufs[cur]
    uint64_t lens[2] = {0, 0};
    int cur = 0;
    bool have_spare = false;
    std::vector<FastxFile::Rec> spare_recs;
    const char *spare_data = nullptr;
    uint64_t spare_n = 0, spare_end = 0, cur_end = 0;  // *_end: where the records of a window end
    uint64_t raw_next = 0;  // raw windows: where the window handed out last ends
    void parse_window(const char *d, uint64_t n, uint64_t from, uint64_t stop, bool partial_ok);
};

// BGZF members inflated by zlib on the host threads, whatever set_bgzf_device() says: out + member.out_off receives each
// member's text (the host checker of windows the device refused or could not hold); CRC-32 and ISIZE checked
}  // namespace cli
struct mk_bgzf_member;
namespace cli {
void inflate_bgzf_members_host(const uint8_t *file, const mk_bgzf_member *members, size_t n, char *out, const std::string &path);

// ---- SAM / BAM ------------------------------------------------------------------------------------
// A whole SAM or BAM input held as ONE buffer (mmap of the text / inflated BAM); records are
// index entries into it, nothing is copied or converted until it is needed: sequences are
// gathered (upper-cased / un-nibbled) straight into the concatenated scan buffer, SAM text of a
// BAM record is produced only for records that are written as text, and BAM -> BAM output passes
// the raw record through with the new tag appended (as the reference's bam crate does).
struct SamFile {
    std::string header;  // header text, every line '\n'-terminated
    const char *data = nullptr;  // the current window (whole text for a memory-mapped SAM)
    bool is_bam = false;
    std::vector<std::string> ref_names;  // BAM: reference dictionary of the binary header
    std::vector<uint32_t> ref_lens;
    struct Rec {
        uint64_t off;      // SAM: start of the text line; BAM: the record's block_size field
        uint32_t len;      // SAM: line length without line end; BAM: 4 + block_size
        uint32_t name_len; // QNAME length (SAM: at off; BAM: at off + 36)
        uint64_t seq_off;  // SAM: SEQ field text; BAM: packed 4-bit sequence
        uint32_t l_seq;    // bases ('*' -> 0)
    };
    std::vector<Rec> recs;
    // A window of records at a time (the bam crate's readers stream records, src/cmd_tag.rs:515,567):
    // open() (by extension: "sam" / "bam", src/cmd_tag.rs:503-615) reads the header and a BAM's reference
    // dictionary, every fill() replaces `recs` by the next records worth ~window_bytes of input; `data`
    // then points at that window.
    void open(const std::string &path);
    bool fill(uint64_t window_bytes);
    // (compressed input; a no-op otherwise) reads, inflates and indexes the window after the current one on the calling
    // thread -- meant for a second thread, beside the work on the current window, whose `data` / `recs` stay valid; the
    // next fill() hands the prepared window out.  An error of that window is thrown here.
    void prefetch(uint64_t window_bytes);
    std::string name(size_t i) const { return std::string(data + recs[i].off + (is_bam ? 36 : 0), recs[i].name_len); }
    // SEQ of records [b0, b1) as the matcher sees it (upper-case ASCII), concatenated (+1 pad byte)
    void gather(size_t b0, size_t b1, std::vector<uint8_t> &seq, std::vector<uint64_t> &off) const;
    // record i as one SAM text line (no line end) appended to out
    void append_line(size_t i, std::string &out) const;
    // BAM input: raw record bytes after the block_size field
    const uint8_t *raw(size_t i) const { return (const uint8_t *)data + recs[i].off + 4; }
    uint32_t raw_len(size_t i) const { return recs[i].len - 4; }
    // value of an existing `tag` field: 0 = absent, 1 = Z value in *val, 2 = present with a
    // non-string type (the reference bails: "Invalid tag value format...")
    int find_tag(size_t i, const std::string &tag, std::string *val) const;
    // BAM input for a caller that keeps the records on the device (run_tag: mk_tag_bam_window takes windows of members as they are
    // stored).  After open(): the file and its member table (source()), the bytes open() inflated behind the header but has not
    // turned into records (the first window's head), and the first member it has not touched.  seek_bam() hands the input back to
    // fill(): the records continue with head[0, n_head) followed by the text of member `member` onwards.
    bool bam_on_bgzf() const { return is_bam && src.is_bgzf(); }
    const WindowSource &source() const { return src; }
    const char *bam_pending(uint64_t *n) const {
        *n = buf_len > cursor ? buf_len - cursor : 0;
        return buf.data() + cursor;
    }
    void seek_bam(size_t member, const char *head, uint64_t n_head);

   private:
    WindowSource src;
    std::vector<char> buf;  // compressed input: the window
    uint64_t buf_len = 0;
    uint64_t cursor = 0;    // mapped input: next unread byte; compressed: bytes of the window already turned into records
    const char *bytes() const { return src.mapped() ? src.text() : buf.data(); }
    uint64_t n_bytes() const { return src.mapped() ? src.text_size() : buf_len; }
    void drop_front(uint64_t k);
    bool next_window(std::vector<char> &b, uint64_t &bl, uint64_t &cur, std::vector<Rec> &rs, uint64_t window_bytes);
    std::vector<char> nbuf;  // the prefetched window
    uint64_t nbuf_len = 0, ncursor = 0;
    std::vector<Rec> nrecs;
    bool have_next = false, next_more = false;
};
// BAM writer (BGZF): encodes SAM text lines against the header's @SQ dictionary.  Used for
// `tag -o out.bam` (src/cmd_tag.rs:254-271); output is checked by reading it back.
// The uncompressed stream is kept as the pieces it arrives in (the per-thread record buffers of a batch are moved in,
// not copied) and cut into 65 280-byte BGZF members when enough has gathered: on the device (use_device:
// mk_bgzf_deflate_pieces uploads the pieces back to back, the codec of merkurio_amd/csrc/codec/ does the rest) or, with
// --host-codec, by zlib on every host thread.  A flush runs on a thread of its own while the caller encodes the next
// batch; members reach the file in order.
struct BamWriter {
    FILE *f = nullptr;
    std::vector<std::vector<uint8_t>> pieces;  // bytes not handed to a flush yet, in stream order
    size_t pieces_bytes = 0;
    std::vector<std::string> ref_names;
    ~BamWriter();
    // reference dictionary from the @SQ lines of header_text, or (BAM -> BAM pass-through, where
    // records keep their reference ids) the input's own binary dictionary
    void open(const std::string &path, const std::string &header_text, const std::vector<std::string> *names = nullptr,
              const std::vector<uint32_t> *lens = nullptr);
    // members are deflated on this HIP device from now on (the handle is created by the first flush)
    void use_device(int device);
    // pass-through of a raw BAM record with one more Z tag appended (thread-safe, see encode_record)
    static void append_tagged_raw(const uint8_t *rec, uint32_t len, const std::string &tag, const char *val, size_t val_len,
                                  std::vector<uint8_t> &dst);
    void write_record(const std::string &sam_line);
    // block_size + record bytes of one SAM text line appended to dst (thread-safe: encoding is the
    // slow part of SAM -> BAM and runs on every host thread), then put_encoded() in record order
    void encode_record(const std::string &sam_line, std::vector<uint8_t> &dst) const;
    void put_encoded(std::vector<uint8_t> &&bytes);
    // complete BGZF members made elsewhere (mk_tag_bam_window: the tagged records deflated on the device): what has been put so
    // far is closed with a member of its own, then these bytes follow it in the file as they are
    // (the buffer: page-locked memory the device writes into directly -- a fresh pageable buffer costs a page fault per 4 KiB inside
    // the copy, more than the copy itself --, handed back and forth between the caller and the writer thread)
    struct RawBuffer {
        uint8_t *p = nullptr;
        size_t cap = 0;
        bool pinned = false;
    };
    void put_members(RawBuffer buffer, size_t used);
    // a buffer of at least min_size bytes for such members: one the writer thread has written out, or a new one
    RawBuffer take_raw_buffer(size_t min_size);
    static void free_raw_buffer(RawBuffer &b);
    // an empty buffer for the next slice of encoded records: one the writer thread is done with (its pages are mapped
    // already: a fresh 16 MB vector costs 4 000 page faults), or a new one
    std::vector<uint8_t> take_buffer();
    void close();
    double deflate_seconds = 0;  // inside the flush threads (timing mode)
    size_t run_members = 1536;   // members per run handed to the writer thread (100 MB of text: two rounds of the deflate kernel's waves)

   private:
    struct Run : std::vector<std::vector<uint8_t>> {  // a run of whole members, as the pieces it arrived in
        bool raw = false;                               // the run IS members already (put_members) ...
        RawBuffer raw_buf;                              // ... in the first raw_used bytes of this buffer
        size_t raw_used = 0;
    };
    void put(const void *p, size_t n);
    void flush(bool all);
    void writer_loop();
    void compress_and_write(Run &run);
    int device_ = -1;  // < 0: zlib on the host threads
    void *codec_ = nullptr;
    std::vector<uint8_t> z_, flat_;  // the run being written: its members; its text in one piece (host codec only)
    // runs wait in a queue for the writer thread (a batch hands over several at once: with one flush in flight the
    // caller stood behind its own previous run); at most kMaxQueuedRuns of them, then the caller waits
    std::deque<Run> queue_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread writer_;
    bool closing_ = false, busy_ = false;
    std::exception_ptr failed_;
    std::vector<std::vector<uint8_t>> free_;  // buffers of written runs (guarded by mu_)
    std::vector<RawBuffer> free_raw_;  // buffers of written raw runs (guarded by mu_)
};

}  // namespace cli
