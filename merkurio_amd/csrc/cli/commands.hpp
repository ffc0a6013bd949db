// commands.hpp -- argument structs of the two subcommands (clap structs of
// src/cmd_extract.rs:32-141 and src/cmd_tag.rs:29-150) and their drivers.
#pragma once
#include <optional>
#include <string>
#include <vector>

#include "util.hpp"

namespace cli {

struct CommonArgs {
    std::vector<std::string> kmer_seq;     // -s
    std::optional<std::string> kmer_file;  // -f
    bool reverse_complement = false;       // -r
    bool canonical = false;                // -c
    std::optional<std::string> out_log;    // -l [path]   ("STDOUT" when given without a value)
    std::optional<std::string> json_log;   // -j [path]
    bool suppress_output = false;          // -S
    bool invert_match = false;             // -v
    bool case_insensitive = false;         // -I
    bool lowercase = false;                // -L
    bool uppercase = false;                // -U
    std::optional<size_t> q_size;          // -q
    bool aho_corasick = false;             // -a
    // opt-in extras of this build (defaults reproduce the reference behaviour)
    int device = 0;       // --device
    int gpus = 1;         // --gpus N: records sharded over N devices, outputs in device order, counters reduced with RCCL
    int window_mb = 1024;  // --window-mb: text (decompressed) read, indexed and held per window of an extract input
    int batch_mb = 128;  // --batch-mb: sequence bytes per GPU batch (tools/batch_mb.sh: 128-256 MB is fastest end to end)
    bool host_codec = false;   // --host-codec: BGZF members are deflated / inflated by zlib on the host threads instead of the device codec
    bool device_codec_always = false;  // --device-codec-always: BGZF input goes through the device codec however few members a call holds (tests: the
                                       // reference's own small BAM / bgzip'ed fixtures through mk_bgzf_inflate)
    bool window_mb_given = false;      // --window-mb was on the command line (its default does not bind the paths that keep the text on the device)
    bool host_ingest = false;  // --host-ingest: extract parses its records on the host threads even where the device could index them; tag keeps the r04 path
                               // (BAM records indexed and tagged on the host threads) where BAM -> BAM would keep them on the device
};

// set by main(): the process ends right after the command (handles are not destroyed one by one, commands.cpp)
extern bool g_process_is_ending;

struct ExtractArgs : CommonArgs {
    std::string in_fastx;                   // -i / -1
    std::optional<std::string> in_fastq_2;  // -2
    std::optional<std::string> out_fastx;   // -o
};

struct TagArgs : CommonArgs {
    std::string in_file;                  // -i
    std::optional<std::string> out_file;  // -o
    std::string tag = "km";               // -t
    int threads = 1;                      // -p
    bool threads_given = false;           // -p on the command line: caps the codec threads (default: all cores)
    bool filter_matching = false;         // -m
};

int run_extract(const ExtractArgs &a, const std::vector<std::string> &argv);
int run_tag(const TagArgs &a, const std::vector<std::string> &argv);

}  // namespace cli
