#include "util.hpp"

#include <sys/stat.h>

#include <cstring>
#include <ctime>
#include <fstream>

namespace cli {

std::string file_name(const std::string &path) {
    size_t e = path.size();
    while (e > 0 && path[e - 1] == '/') --e;
    size_t b = path.rfind('/', e ? e - 1 : 0);
    b = (b == std::string::npos || b >= e) ? 0 : b + 1;
    return path.substr(b, e - b);
}

static size_t ext_dot(const std::string &name) {  // index of the extension dot inside a file name
    size_t d = name.rfind('.');
    if (d == std::string::npos || d == 0) return std::string::npos;
    return d;
}

std::string extension(const std::string &path) {
    const std::string n = file_name(path);
    size_t d = ext_dot(n);
    return d == std::string::npos ? "" : n.substr(d + 1);
}

std::string with_extension(const std::string &path, const std::string &ext) {
    const std::string n = file_name(path);
    if (n.empty()) return path;
    const size_t dir_len = path.rfind(n);
    size_t d = ext_dot(n);
    std::string stem = d == std::string::npos ? n : n.substr(0, d);
    std::string out = path.substr(0, dir_len) + stem;
    if (!ext.empty()) out += "." + ext;
    return out;
}

bool is_directory(const std::string &path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

std::string add_suffix_to_file_prefix(const std::string &path, const std::string &suffix) {
    const std::string n = file_name(path);
    if (n.empty()) bail("Invalid file name");
    const size_t dir_len = path.rfind(n);
    size_t d = n.find('.');
    std::string out = d == std::string::npos ? n + suffix : n.substr(0, d) + suffix + n.substr(d);
    return path.substr(0, dir_len) + out;
}

std::string identify_uncompressed_type(const std::string &path) {
    if (is_directory(path)) bail("The path points to a directory.");
    const std::string ext = extension(path);
    if (ext.empty()) bail("Path has no extension");
    if (ext == "gz" || ext == "bz" || ext == "bz2" || ext == "xz") {
        const std::string inner = extension(with_extension(path, ""));
        if (inner.empty()) bail("Could not determine uncompressed file type");
        return inner;
    }
    return ext;
}

std::string check_log_flag_conflict(const std::string *out_log, const std::string *json_log, const std::string *out_file,
                                    bool suppress_output) {
    const bool l_stdout = out_log && *out_log == "STDOUT", j_stdout = json_log && *json_log == "STDOUT";
    if (l_stdout && j_stdout)
        return "Cannot use both -l/--out-log and -j/--json-log with no arguments (both to stdout). Please specify a "
               "file for at least one.";
    if ((l_stdout || j_stdout) && !out_file && !suppress_output)
        return "Cannot write log to stdout when normal output is also stdout. Specify an output file with -o or "
               "suppress output with -S.";
    return "";
}

std::string timestamp_now() {
    time_t t = time(nullptr);
    struct tm lt;
    localtime_r(&t, &lt);
    char buf[64], off[16];
    strftime(buf, sizeof(buf), "%Y-%m-%dT%H:%M:%S", &lt);
    long o = lt.tm_gmtoff;
    snprintf(off, sizeof(off), "%c%02ld:%02ld", o < 0 ? '-' : '+', labs(o) / 3600, labs(o) % 3600 / 60);
    std::string zone;
    if (const char *tz = getenv("TZ"); tz && *tz) zone = tz[0] == ':' ? tz + 1 : tz;
    if (zone.empty()) {
        std::ifstream f("/etc/timezone");
        std::getline(f, zone);
    }
    if (zone.empty()) zone = "UTC";
    return std::string(buf) + off + "[" + zone + "]";
}

// ---- JSON ---------------------------------------------------------------------------------------
static void json_escape(const std::string &s, std::string &out) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
            if (c < 0x20) {
                char b[8];
                snprintf(b, sizeof(b), "\\u%04x", c);
                out += b;
            } else {
                out += (char)c;
            }
        }
    }
    out += '"';
}

static void json_write(const Json &v, int depth, std::string &out) {
    auto pad = [&](int d) { out.append((size_t)d * 2, ' '); };
    switch (v.kind) {
    case Json::Null: out += "null"; break;
    case Json::Bool: out += v.b ? "true" : "false"; break;
    case Json::Int: out += std::to_string(v.i); break;
    case Json::Str: json_escape(v.s, out); break;
    case Json::Arr:
        if (v.arr.empty()) {
            out += "[]";
            break;
        }
        out += "[\n";
        for (size_t k = 0; k < v.arr.size(); ++k) {
            pad(depth + 1);
            json_write(v.arr[k], depth + 1, out);
            out += k + 1 < v.arr.size() ? ",\n" : "\n";
        }
        pad(depth);
        out += "]";
        break;
    case Json::Obj:
        if (v.obj.empty()) {
            out += "{}";
            break;
        }
        out += "{\n";
        {
            size_t k = 0;
            for (auto &kv : v.obj) {
                pad(depth + 1);
                json_escape(kv.first, out);
                out += ": ";
                json_write(kv.second, depth + 1, out);
                out += ++k < v.obj.size() ? ",\n" : "\n";
            }
        }
        pad(depth);
        out += "}";
        break;
    }
}

std::string json_pretty(const Json &v, int) {
    std::string out;
    json_write(v, 0, out);
    return out;
}

// ---- sinks / loggers ----------------------------------------------------------------------------
Sink::~Sink() {
    flush();
    if (f && owned) fclose(f);
}
void Sink::open(const std::string &path) {
    if (path == "STDOUT") {
        f = stdout;
        owned = false;
    } else {
        f = fopen(path.c_str(), "wb");
        owned = true;
    }
}
void Sink::write(const char *p, size_t n) {
    buf.append(p, n);
    if (buf.size() >= (1u << 20)) flush();
}
void Sink::flush() {
    if (f && !buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
    buf.clear();
    if (f) fflush(f);
}

static void append_u64(std::string &b, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) b += tmp[--n];
}

void TextLogger::format(std::string &b, const std::string &file, const char *id, size_t id_len, const std::string &pattern, uint64_t pos) {
    b += file;  // src/logger.rs:48-55
    b += '\t';
    b.append(id, id_len);
    b += '\t';
    b += pattern;
    b += '\t';
    append_u64(b, pos);
    b += '\n';
}

void TextLogger::row(const std::string &file, const std::string &id, const std::string &pattern, uint64_t pos) {
    if (!out) return;
    format(out->buf, file, id.data(), id.size(), pattern, pos);
    if (out->buf.size() >= (1u << 20)) out->flush();
}

void JsonLogger::begin() { out->write("{\n  \"matching_records\": [\n"); }  // src/logger.rs:97

static void json_escape_raw(const char *p, size_t n, std::string &out) { json_escape(std::string(p, n), out); }

// One hit as the reference writes it: serde_json's pretty object (keys in alphabetical order, position a STRING:
// src/logger.rs:116-121) re-indented by four spaces (src/logger.rs:123-128), rows separated by a line holding a comma
// (src/logger.rs:111-113).  Written out directly -- the generic Json value + pretty printer took a microsecond per row.
void JsonLogger::format(std::string &b, bool separator, const std::string &file, const char *id, size_t id_len, const std::string &pattern,
                        uint64_t pos) {
    if (separator) b += ",\n";
    b += "    {\n      \"file\": ";
    json_escape(file, b);
    b += ",\n      \"pattern\": ";
    json_escape(pattern, b);
    b += ",\n      \"position\": \"";
    append_u64(b, pos);
    b += "\",\n      \"record_id\": ";
    json_escape_raw(id, id_len, b);
    b += "\n    }\n";
}

void JsonLogger::row(const std::string &file, const std::string &id, const std::string &pattern, uint64_t pos) {
    std::string b;
    format(b, !first, file, id.data(), id.size(), pattern, pos);
    first = false;
    out->write(b);
}

static void write_indented(Sink &s, const Json &v, int indent) {  // src/logger.rs:145-155 (+ pop of the last '\n')
    const std::string pretty = json_pretty(v);
    size_t b = 0;
    bool first = true;
    while (b < pretty.size()) {
        size_t e = pretty.find('\n', b);
        if (e == std::string::npos) e = pretty.size();
        if (!first) {
            s.write("\n");
            s.write(std::string((size_t)indent, ' '));
        }
        s.write(pretty.data() + b, e - b);
        first = false;
        b = e + 1;
    }
}

void JsonLogger::finalize(const Json &meta, const Json &counts, const Json &summary, const Json *paired) {
    out->write("  ],\n  \"meta_information\": ");
    write_indented(*out, meta, 2);
    if (paired) {
        out->write(",\n  \"paired_end_reads_statistics\": ");
        write_indented(*out, *paired, 2);
    }
    out->write(",\n  \"pattern_hit_counts\": ");
    write_indented(*out, counts, 2);
    out->write(",\n  \"summary_statistics\": ");
    write_indented(*out, summary, 2);
    out->write("\n}\n");
    out->flush();
}

}  // namespace cli
