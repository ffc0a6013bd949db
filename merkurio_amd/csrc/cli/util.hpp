// util.hpp -- small host utilities of the `merkurio` CLI: errors, path helpers
// (src/helpers.rs:16-68), an ordered JSON value with serde_json-compatible pretty printing
// (src/logger.rs:108-190), and the two loggers (src/logger.rs:11-191).
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cli {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] inline void bail(const std::string &msg) { throw Error(msg); }

// ---- paths (std::path semantics used by the reference) -------------------------------------
std::string file_name(const std::string &path);                   // Path::file_name
std::string extension(const std::string &path);                   // Path::extension ("" if none)
std::string with_extension(const std::string &path, const std::string &ext);  // Path::with_extension
bool is_directory(const std::string &path);
// helpers::add_suffix_to_file_prefix, src/helpers.rs:29-43: "a.b.gz" + "_1" -> "a_1.b.gz"
std::string add_suffix_to_file_prefix(const std::string &path, const std::string &suffix);
// helpers::identify_uncompressed_type, src/helpers.rs:48-68
std::string identify_uncompressed_type(const std::string &path);
// helpers::check_log_flag_conflict, src/helpers.rs:172-200; returns "" or the error text
std::string check_log_flag_conflict(const std::string *out_log, const std::string *json_log, const std::string *out_file,
                                    bool suppress_output);
// jiff's `Zoned::now().round(Unit::Second)` Display form: 2025-07-22T08:44:34+02:00[Europe/Vienna]
std::string timestamp_now();

// ---- JSON value with sorted object keys (serde_json's default Map is a BTreeMap) -------------
struct Json {
    enum Kind { Null, Bool, Int, Str, Arr, Obj } kind = Null;
    bool b = false;
    long long i = 0;
    std::string s;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;
    static Json null() { return Json(); }
    static Json boolean(bool v) { Json j; j.kind = Bool; j.b = v; return j; }
    static Json integer(long long v) { Json j; j.kind = Int; j.i = v; return j; }
    static Json string(const std::string &v) { Json j; j.kind = Str; j.s = v; return j; }
    static Json array() { Json j; j.kind = Arr; return j; }
    static Json object() { Json j; j.kind = Obj; return j; }
    Json &set(const std::string &k, Json v) { obj[k] = std::move(v); return *this; }
    Json &push(Json v) { arr.push_back(std::move(v)); return *this; }
};
// serde_json::to_string_pretty (2-space indent)
std::string json_pretty(const Json &v, int indent = 0);

// ---- output sink: file or stdout --------------------------------------------------------------
struct Sink {
    FILE *f = nullptr;
    bool owned = false;
    std::string buf;
    ~Sink();
    void open(const std::string &path);  // "STDOUT" or a path
    void write(const std::string &s) { write(s.data(), s.size()); }
    void write(const char *p, size_t n);
    void flush();
};

// logger::BufferedLogger (text log).  The reference keeps every row in memory as well
// (`records`, src/logger.rs:45-46); that is a memory quirk, not behaviour, and is not copied.
struct TextLogger {
    std::unique_ptr<Sink> out;  // null: no text log
    void header(const std::string &s) { if (out) { out->write(s); } }
    void row(const std::string &file, const std::string &id, const std::string &pattern, uint64_t pos);
    // the bytes row() writes, appended to b (rows of a batch are formatted by several host threads, then written in order)
    static void format(std::string &b, const std::string &file, const char *id, size_t id_len, const std::string &pattern, uint64_t pos);
    void flush() { if (out) out->flush(); }
};

// logger::JsonLogger (streaming pretty JSON)
struct JsonLogger {
    std::unique_ptr<Sink> out;
    bool first = true;
    void begin();  // writes `{\n  "matching_records": [\n`
    void row(const std::string &file, const std::string &id, const std::string &pattern, uint64_t pos);
    // the bytes row() writes (with the separator line in front unless it is the log's first row), appended to b
    static void format(std::string &b, bool separator, const std::string &file, const char *id, size_t id_len, const std::string &pattern,
                       uint64_t pos);
    void finalize(const Json &meta, const Json &pattern_hit_counts, const Json &summary, const Json *paired);
};

}  // namespace cli
