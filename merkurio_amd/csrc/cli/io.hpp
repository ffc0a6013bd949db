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

// whole file into memory; transparently inflates gzip / BGZF (magic 1f 8b).  bz2 / xz / zstd
// inputs are recognised and rejected with a clear message (no such libraries in this build).
std::vector<char> read_file_maybe_gz(const std::string &path);

// ---- FASTA / FASTQ ------------------------------------------------------------------------------
struct FastxFile {
    std::vector<char> data;  // the (decompressed) file
    bool fastq = false;
    struct Rec {
        uint64_t id_b, id_e;    // header line without the marker ('>' / '@') and line end
        uint64_t raw_b, raw_e;  // sequence as stored, inner line breaks included (FASTA) -- what write() emits
        uint64_t qual_b, qual_e;
    };
    std::vector<Rec> recs;
    void parse(const std::string &path);
    std::string id(size_t i) const { return std::string(data.data() + recs[i].id_b, recs[i].id_e - recs[i].id_b); }
    // record.seq(): newline-free sequence appended to `out`; returns num_bases()
    uint64_t append_seq(size_t i, std::vector<uint8_t> &out) const;
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
// value of an existing `tag:Z:` field of a SAM line: returns 0 = absent, 1 = Z value in *val,
// 2 = present with a non-string type (the reference bails: "Invalid tag value format...")
int sam_find_tag(const std::string &line, const std::string &tag, std::string *val);

}  // namespace cli
