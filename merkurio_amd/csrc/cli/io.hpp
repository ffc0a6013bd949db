// io.hpp -- record readers/writers of the `merkurio` CLI (host side, either side of the hot
// path).  Observable behaviour follows what the reference gets from its crates
// (SURVEY.md §5): needletail 0.6.3 for FASTA/FASTQ (src/cmd_extract.rs:281,327-340,403) and
// bam 0.1.4 for SAM/BAM (src/cmd_tag.rs:470-497,503-615).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "util.hpp"

namespace cli {

// host threads for ingest / codec work: hardware concurrency, capped at 32, MERKURIO_IO_THREADS overrides
unsigned io_threads();

// whole file into memory; transparently inflates gzip (magic 1f 8b); BGZF members in parallel.  bz2 / xz / zstd
// inputs are recognised and rejected with a clear message (no such libraries in this build).
std::vector<char> read_file_maybe_gz(const std::string &path);

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
    const char *data = nullptr;  // = file.p
    bool fastq = false;
    struct Rec {
        uint64_t id_b, id_e;    // header line without the marker ('>' / '@') and line end
        uint64_t raw_b, raw_e;  // sequence as stored, inner line breaks included (FASTA) -- what write() emits
        uint64_t qual_b, qual_e;
    };
    std::vector<Rec> recs;
    void parse(const std::string &path);
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
};

// ---- SAM / BAM ------------------------------------------------------------------------------------
struct SamFile {
    std::string header;  // header text, every line '\n'-terminated
    struct Rec {
        std::string line;  // the record as one SAM text line, no line end (BAM input: converted)
        std::string name;  // QNAME
        std::string seq;   // SEQ as the matcher sees it: upper-case, "" for '*'
    };
    std::vector<Rec> recs;
    void parse(const std::string &path);  // by extension: "sam" / "bam" (src/cmd_tag.rs:503-615)
};
// BAM writer (BGZF): encodes SAM text lines against the header's @SQ dictionary.  Used for
// `tag -o out.bam` (src/cmd_tag.rs:254-271); output is checked by reading it back.
struct BamWriter {
    FILE *f = nullptr;
    std::vector<uint8_t> block;  // uncompressed bytes of the block being filled (< 64 KiB per BGZF block)
    std::vector<std::vector<uint8_t>> pending;  // full blocks waiting for the next parallel deflate
    std::vector<std::string> ref_names;
    ~BamWriter();
    void open(const std::string &path, const std::string &header_text);
    void write_record(const std::string &sam_line);
    void close();

   private:
    void put(const void *p, size_t n);
    void flush_block();
    void flush_pending();
};

// value of an existing `tag:Z:` field of a SAM line: returns 0 = absent, 1 = Z value in *val,
// 2 = present with a non-string type (the reference bails: "Invalid tag value format...")
int sam_find_tag(const std::string &line, const std::string &tag, std::string *val);

}  // namespace cli
