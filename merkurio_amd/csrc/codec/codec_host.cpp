// codec_host.cpp -- host side of the C ABI's BGZF codec (include/merkurio_hip.h, v5): the mk_codec handle, its device
// buffers, chunking of a call into device-sized pieces, the BSIZE walk.  Kernels: bgzf_deflate.hip, bgzf_inflate.hip.
// Reference interface replaced: the BGZF reader / writer of `bam 0.1.4` as `merkurio tag` drives it
// (src/cmd_tag.rs:254-271 `BamWriter::build().write_header(..).from_path`, :503-506 `BamReader::from_path(.., threads)`)
// and needletail's gzip reader on bgzip'ed inputs (src/cmd_extract.rs:281).
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../host_common.h"
#include "codec_internal.h"
#include "codec_kernels.h"
#include "gzip_segments.hpp"

namespace mk {
int hip_fail(hipError_t e, const char *what);
int ensure_device(void **p, size_t *cap, size_t need);
}  // namespace mk

namespace {

// Members per device pass (mk_codec_set_pass_limits overrides).  Deflate deals its members from a counter to a resident
// grid of 16 waves per CU, so a pass only has to be large: 49 152 members = 3.2 GB of text (its 64 KiB slot per member is
// what bounds it).  Inflate is bound by the latency of its slowest lane, 25-60 ms per launch whatever it holds
// (profiles/r04_codec_kernels.txt): a pass takes as many members as the part holds streams, 3 GiB of text.
constexpr uint64_t kDeflateChunkBlocks = 49152;
constexpr uint64_t kInflateChunkText = 3ull << 30;

#define MKC_HIP(call, what)                                   \
    do {                                                      \
        const hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return mk::hip_fail(e_, what);  \
    } while (0)

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
}

// ---- host buffers <-> device through the handle's two page-locked staging buffers (r05) -------------------------------------
// A caller's plain (pageable) buffer cannot be a DMA source or target: the runtime stages it through a bounce buffer on ONE thread,
// and a freshly allocated output buffer additionally takes a page fault per 4 KiB inside that copy -- mk_bgzf_inflate of 1 GB spent
// 335 of its 376 ms there (kernels 41 ms; tools/codec_real.py before this change).  Here the host threads copy piece k + 1 into /
// out of one staging buffer (first touch of the caller's pages included, on all of them) while the DMA engine moves piece k through
// the other.  Buffers that are page-locked already (mk_host_alloc) and small ones go the direct way.
constexpr uint64_t kStageBytes = 32ull << 20, kStageFrom = 4ull << 20;

unsigned copy_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc ? hc : 4u));
}
void parallel_copy(uint8_t *dst, const uint8_t *src, uint64_t n) {
    const unsigned T = (unsigned)std::min<uint64_t>(copy_threads(), n / (1u << 20) + 1);
    if (T <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([=] { memcpy(dst + n * t / T, src + n * t / T, (size_t)(n * (t + 1) / T - n * t / T)); });
    memcpy(dst, src, (size_t)(n / T));
    for (auto &x : th) x.join();
}
bool is_page_locked(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (an ordinary malloc'ed pointer is "invalid value" to this query: not an error of ours)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
int stage_buffers(mk_codec *c) {
    for (int k = 0; k < 2; ++k) {
        if (!c->h_stage[k]) MKC_HIP(hipHostMalloc(&c->h_stage[k], kStageBytes, hipHostMallocDefault), "hipHostMalloc of a staging buffer");
        if (!c->ev_stage[k]) MKC_HIP(hipEventCreateWithFlags(&c->ev_stage[k], hipEventDisableTiming), "hipEventCreate");
    }
    return MK_OK;
}
// src[0, n) -> d_dst, enqueued on the handle's stream; the host side of it is done when this returns
int upload(mk_codec *c, void *d_dst, const uint8_t *src, uint64_t n) {
    if (n < kStageFrom || is_page_locked(src)) {
        MKC_HIP(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, c->stream), "upload");
        return MK_OK;
    }
    int rc = stage_buffers(c);
    if (rc) return rc;
    int k = 0;
    for (uint64_t at = 0; at < n; at += kStageBytes, k ^= 1) {
        const uint64_t m = std::min<uint64_t>(kStageBytes, n - at);
        if (at >= 2 * kStageBytes) MKC_HIP(hipEventSynchronize(c->ev_stage[k]), "hipEventSynchronize");  // its previous piece has left
        parallel_copy((uint8_t *)c->h_stage[k], src + at, m);
        MKC_HIP(hipMemcpyAsync((uint8_t *)d_dst + at, c->h_stage[k], m, hipMemcpyHostToDevice, c->stream), "upload");
        MKC_HIP(hipEventRecord(c->ev_stage[k], c->stream), "hipEventRecord");
    }
    // the staging buffers are reused by the next transfer of the call: wait until both have been read
    MKC_HIP(hipEventSynchronize(c->ev_stage[0]), "hipEventSynchronize");
    MKC_HIP(hipEventSynchronize(c->ev_stage[1]), "hipEventSynchronize");
    return MK_OK;
}
// d_src[0, n) -> dst; waits for the stream (everything enqueued before it included)
int download(mk_codec *c, uint8_t *dst, const void *d_src, uint64_t n) {
    if (n < kStageFrom || is_page_locked(dst)) {
        if (n) MKC_HIP(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, c->stream), "download");
        MKC_HIP(hipStreamSynchronize(c->stream), "download");
        return MK_OK;
    }
    int rc = stage_buffers(c);
    if (rc) return rc;
    // piece j travels into buffer j & 1 while the host threads copy piece j - 1 out of the other
    const uint64_t pieces = (n + kStageBytes - 1) / kStageBytes;
    for (uint64_t j = 0; j <= pieces; ++j) {
        if (j < pieces) {
            const uint64_t at = j * kStageBytes, m = std::min<uint64_t>(kStageBytes, n - at);
            MKC_HIP(hipMemcpyAsync(c->h_stage[j & 1], (const uint8_t *)d_src + at, m, hipMemcpyDeviceToHost, c->stream), "download");
            MKC_HIP(hipEventRecord(c->ev_stage[j & 1], c->stream), "hipEventRecord");
        }
        if (j > 0) {
            const uint64_t at = (j - 1) * kStageBytes, m = std::min<uint64_t>(kStageBytes, n - at);
            MKC_HIP(hipEventSynchronize(c->ev_stage[(j - 1) & 1]), "hipEventSynchronize");
            parallel_copy(dst + at, (const uint8_t *)c->h_stage[(j - 1) & 1], m);
        }
    }
    return MK_OK;
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

const uint8_t kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

}  // namespace

extern "C" {

int mk_codec_create(int device, mk_codec **out) {
    MK_ABI_BEGIN
    if (!out) return mk::fail(MK_E_INVALID_ARG, "mk_codec_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return mk::fail(MK_E_HIP, "no HIP device (the BGZF codec has no CPU path)");
    if (device < 0 || device >= n) return mk::fail(MK_E_INVALID_ARG, "mk_codec_create: device %d of %d", device, n);
    MKC_HIP(hipSetDevice(device), "hipSetDevice");
    mk_codec *c = new mk_codec;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&c->ev[k]);
    if (e != hipSuccess) {
        mk_codec_destroy(c);
        return mk::hip_fail(e, "mk_codec_create");
    }
    *out = c;
    return MK_OK;
    MK_ABI_END
}

void mk_codec_destroy(mk_codec *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : {c->d_in, c->d_crc, c->d_tokens, c->d_slots, c->d_len, c->d_off, c->d_out, c->d_aux, c->d_gz_in, c->d_gz_sym, c->d_gz_tab, c->d_gz_ctx, c->d_gz_text})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
        if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
        if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

uint64_t mk_bgzf_deflate_bound(uint64_t n, uint32_t block_bytes) {
    const uint64_t bb = block_bytes ? block_bytes : mkz::kMaxBlockBytes;
    const uint64_t blocks = (n + bb - 1) / bb;
    return n + blocks * 31;  // a stored member: 18 + 5 + text + 8
}

const uint8_t *mk_bgzf_eof(void) { return kEof; }

int mk_codec_set_pass_limits(mk_codec *c, uint64_t deflate_members, uint64_t inflate_text_bytes) {
    if (!c) return mk::fail(MK_E_INVALID_ARG, "mk_codec_set_pass_limits: NULL handle");
    std::lock_guard<std::mutex> lock(c->mu);
    c->deflate_pass_blocks = deflate_members;
    c->inflate_pass_text = inflate_text_bytes;
    return MK_OK;
}

int mk_codec_set_gzip_chunk(mk_codec *c, uint64_t chunk_bytes) {
    if (!c || (chunk_bytes && (chunk_bytes < 4096 || chunk_bytes > (1u << 20) || (chunk_bytes & (chunk_bytes - 1)))))
        return mk::fail(MK_E_INVALID_ARG, "mk_codec_set_gzip_chunk: handle / a power of two of 4 KiB ... 1 MiB");
    std::lock_guard<std::mutex> lock(c->mu);
    c->gzip_chunk = chunk_bytes;
    return MK_OK;
}
int mk_codec_set_inflate_kernel(mk_codec *c, int which) {
    if (!c || which < 0 || which > 6) return mk::fail(MK_E_INVALID_ARG, "mk_codec_set_inflate_kernel: handle / selector");
    std::lock_guard<std::mutex> lock(c->mu);
    c->inflate_kernel = which;
    return MK_OK;
}

int mk_codec_times(const mk_codec *c, float ms[3]) {
    if (!c || !ms) return mk::fail(MK_E_INVALID_ARG, "mk_codec_times: NULL argument");
    for (int k = 0; k < 3; ++k) ms[k] = c->ms[k];
    return MK_OK;
}

int mk_bgzf_deflate_pieces(mk_codec *c, const uint8_t *const *pieces, const uint64_t *sizes, uint64_t n_pieces, uint32_t block_bytes,
                           uint8_t *out, uint64_t out_cap, uint64_t *out_len) {
    MK_ABI_BEGIN
    if (!c || !out_len || (n_pieces && (!pieces || !sizes))) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_deflate: NULL argument");
    uint64_t n = 0;
    for (uint64_t k = 0; k < n_pieces; ++k) {
        if (sizes[k] && !pieces[k]) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_deflate: piece %llu is NULL", (unsigned long long)k);
        n += sizes[k];
    }
    if (n && !out) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_deflate: NULL argument");
    const uint32_t bb = block_bytes ? block_bytes : mkz::kMaxBlockBytes;
    if (bb > mkz::kMaxBlockBytes) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_deflate: block_bytes %u > %u", bb, mkz::kMaxBlockBytes);
    const uint64_t bound = mk_bgzf_deflate_bound(n, bb);
    *out_len = 0;
    if (out_cap < bound) {
        *out_len = bound;
        return mk::fail(MK_E_CAPACITY, "mk_bgzf_deflate: out_cap %llu < bound %llu", (unsigned long long)out_cap, (unsigned long long)bound);
    }
    std::lock_guard<std::mutex> lock(c->mu);
    MKC_HIP(hipSetDevice(c->device), "hipSetDevice");
    c->ms[0] = c->ms[1] = c->ms[2] = 0;
    const uint64_t chunk_bytes = (c->deflate_pass_blocks ? c->deflate_pass_blocks : kDeflateChunkBlocks) * bb;
    uint64_t written = 0;
    uint64_t piece = 0, piece_at = 0;  // where the text of the next device pass starts
    for (uint64_t at = 0; at < n; at += chunk_bytes) {
        const uint64_t cn = std::min<uint64_t>(chunk_bytes, n - at);
        const uint32_t blocks = (uint32_t)((cn + bb - 1) / bb);
        const uint32_t grid = mkz::deflate_grid(blocks, c->num_cus);
        int rc;
        if ((rc = mk::ensure_device(&c->d_in, &c->in_cap, cn + mkz::kPad)) || (rc = mk::ensure_device(&c->d_crc, &c->crc_cap, blocks * 4ull)) ||
            (rc = mk::ensure_device(&c->d_tokens, &c->tokens_cap, (uint64_t)grid * mkz::kTokensPerWave * 4)) ||
            (rc = mk::ensure_device(&c->d_slots, &c->slots_cap, (uint64_t)blocks * mkz::kSlotBytes)) ||
            (rc = mk::ensure_device(&c->d_len, &c->len_cap, blocks * 4ull)) || (rc = mk::ensure_device(&c->d_off, &c->off_cap, (blocks + 2) * 8ull)) ||
            (rc = mk::ensure_device(&c->d_out, &c->out_cap, mk_bgzf_deflate_bound(cn, bb))))
            return rc;
        uint64_t *d_total = (uint64_t *)c->d_off + blocks;
        const double t_up = now_ms();
        for (uint64_t done = 0; done < cn;) {  // the pieces land back to back: the concatenation exists on the device only
            while (piece_at == sizes[piece]) ++piece, piece_at = 0;
            const uint64_t k = std::min<uint64_t>(cn - done, sizes[piece] - piece_at);
            if ((rc = upload(c, (uint8_t *)c->d_in + done, pieces[piece] + piece_at, k))) return rc;
            done += k, piece_at += k;
        }
        MKC_HIP(hipMemsetAsync((uint8_t *)c->d_in + cn, 0, mkz::kPad, c->stream), "hipMemsetAsync");
        MKC_HIP(hipStreamSynchronize(c->stream), "upload of the text");
        c->ms[0] += (float)(now_ms() - t_up);
        MKC_HIP(hipEventRecord(c->ev[1], c->stream), "hipEventRecord");
        mkz::launch_crc((const uint8_t *)c->d_in, cn, bb, blocks, (uint32_t *)c->d_crc, c->stream);
        mkz::launch_deflate((const uint8_t *)c->d_in, cn, bb, blocks, (const uint32_t *)c->d_crc, (uint32_t *)c->d_tokens, (uint8_t *)c->d_slots,
                            (uint32_t *)c->d_len, (uint32_t *)(d_total + 1), grid, c->stream);
        mkz::launch_pack((const uint8_t *)c->d_slots, (const uint32_t *)c->d_len, (uint64_t *)c->d_off, d_total, blocks, (uint8_t *)c->d_out, c->stream);
        MKC_HIP(hipGetLastError(), "BGZF deflate kernels");
        MKC_HIP(hipEventRecord(c->ev[2], c->stream), "hipEventRecord");
        uint64_t total = 0;
        MKC_HIP(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream), "download of the size");
        MKC_HIP(hipStreamSynchronize(c->stream), "BGZF deflate");
        if (total > out_cap - written) return mk::fail(MK_E_CAPACITY, "mk_bgzf_deflate: members exceed the bound (internal error)");
        const double t_down = now_ms();
        if ((rc = download(c, out + written, c->d_out, total))) return rc;
        c->ms[1] += elapsed(c->ev[1], c->ev[2]), c->ms[2] += (float)(now_ms() - t_down);
        written += total;
    }
    *out_len = written;
    return MK_OK;
    MK_ABI_END
}

int mk_bgzf_deflate(mk_codec *c, const uint8_t *in, uint64_t n, uint32_t block_bytes, uint8_t *out, uint64_t out_cap, uint64_t *out_len) {
    if (n && !in) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_deflate: NULL argument");
    return mk_bgzf_deflate_pieces(c, &in, &n, n ? 1 : 0, block_bytes, out, out_cap, out_len);
}

int mk_bgzf_members(const uint8_t *in, uint64_t n, mk_bgzf_member *members, uint64_t cap, uint64_t *n_members, uint64_t *consumed,
                    uint64_t *text_bytes) {
    if (!n_members || (n && !in)) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_members: NULL argument");
    uint64_t at = 0, k = 0, text = 0;
    while (n - at >= 18) {
        const uint8_t *h = in + at;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return mk::fail(MK_E_CORRUPT, "not a BGZF member at byte %llu", (unsigned long long)at);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (n - at < 12 + (uint64_t)xlen) break;
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {  // the extra field's subfields: SI1 SI2 SLEN data
            const uint8_t *f = h + 12 + x;
            const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = f[4] | (uint32_t)f[5] << 8, found = true;
            x += 4 + slen;
        }
        if (!found) return mk::fail(MK_E_CORRUPT, "gzip member without a BC subfield at byte %llu", (unsigned long long)at);
        const uint64_t size = (uint64_t)bsize + 1;
        if (size < 12 + (uint64_t)xlen + 8) return mk::fail(MK_E_CORRUPT, "BGZF member with BSIZE %u at byte %llu", bsize, (unsigned long long)at);
        if (n - at < size) break;
        uint32_t crc, isize;
        memcpy(&crc, h + size - 8, 4), memcpy(&isize, h + size - 4, 4);
        if (isize > 65536) return mk::fail(MK_E_CORRUPT, "BGZF member with ISIZE %u at byte %llu", isize, (unsigned long long)at);
        // RFC 1952's optional fields between the extra field and the DEFLATE stream (bgzip sets none of them, a gzip member
        // that carries a BC subfield may): FNAME / FCOMMENT are zero-terminated, FHCRC is two bytes; reserved bits refuse
        uint64_t skip = 12 + (uint64_t)xlen;
        if (h[3] & 0xE0) return mk::fail(MK_E_UNSUPPORTED, "gzip member with reserved FLG bits (0x%02x) at byte %llu", h[3], (unsigned long long)at);
        for (int bit = 3; bit <= 4; ++bit)
            if (h[3] & (1 << bit)) {
                const void *z = skip < size - 8 ? memchr(h + skip, 0, (size_t)(size - 8 - skip)) : nullptr;
                if (!z) return mk::fail(MK_E_CORRUPT, "gzip member with an unterminated %s at byte %llu", bit == 3 ? "FNAME" : "FCOMMENT", (unsigned long long)at);
                skip = (uint64_t)((const uint8_t *)z - h) + 1;
            }
        if (h[3] & 2) skip += 2;
        if (skip + 8 > size) return mk::fail(MK_E_CORRUPT, "BGZF member whose header fields overrun BSIZE at byte %llu", (unsigned long long)at);
        if (members && k < cap) members[k] = mk_bgzf_member{at + skip, text, (uint32_t)(size - skip - 8), isize, crc, 0};
        ++k, text += isize, at += size;
    }
    *n_members = k;
    if (consumed) *consumed = at;
    if (text_bytes) *text_bytes = text;
    return MK_OK;
}

// ---- one gzip member in parallel pieces (gzip_segments.hpp says how; kernels: gzip_inflate.hip) ---------------------------------
int mk_gzip_inflate_device(mk_codec *c, const uint8_t *gz, uint64_t n, uint64_t *text_bytes, uint32_t *taken) {
    MK_ABI_BEGIN
    if (!c || !text_bytes || !taken || (n && !gz)) return mk::fail(MK_E_INVALID_ARG, "mk_gzip_inflate_device: NULL argument");
    *text_bytes = 0, *taken = 0;
    std::lock_guard<std::mutex> lock(c->mu);
    c->gz_text_bytes = 0, c->gz_segments = 0;
    for (float &x : c->gz_ms) x = 0;
    // RFC 1952: the member's header, then the DEFLATE stream up to the 8 trailer bytes.  One member is what is taken; whether the
    // stream really ends where the file does shows when the last segment meets its final block there.
    if (n < 18 + 2 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8 || (gz[3] & 0xE0)) return MK_OK;  // (not gzip / reserved flags: not taken)
    uint64_t p = 10;
    if (gz[3] & 4) {
        if (p + 2 > n) return MK_OK;
        p += 2 + (gz[p] | (uint64_t)gz[p + 1] << 8);
    }
    for (int bit = 3; bit <= 4; ++bit)
        if (gz[3] & (1 << bit)) {
            const void *z = p < n ? memchr(gz + p, 0, (size_t)(n - p)) : nullptr;
            if (!z) return MK_OK;
            p = (uint64_t)((const uint8_t *)z - gz) + 1;
        }
    if (gz[3] & 2) p += 2;
    if (p + 8 + 2 > n) return MK_OK;
    const uint64_t n_in = n - 8 - p;
    uint32_t want_crc, want_isize;
    memcpy(&want_crc, gz + n - 8, 4), memcpy(&want_isize, gz + n - 4, 4);
    MKC_HIP(hipSetDevice(c->device), "hipSetDevice");
    int rc;
    // nominal chunks of compressed bytes (a block of zlib's is 15-40 KiB of them).  A piece is decoded by one wave at that wave's pace
    // (~13 MB/s of text), so what counts is the largest piece and how many rounds of the device's decoder slots (17 per CU) the
    // pieces make.  Cuts every 16 KiB give every block its own piece (no piece can be smaller); up to a round and a half of them
    // that is the fastest (61 MB of stream: 27 ms against 37); above, cuts every 64 KiB (pieces of ~2 blocks, a third of the
    // search waves) are (245 MB: 63 ms against 67).  At most 32 768 chunks (the prefix kernel's grid; a stream of more than 2 GiB
    // gets larger chunks).
    uint64_t chunk = c->gzip_chunk ? c->gzip_chunk : (n_in / (32u << 10) <= (uint64_t)c->num_cus * 17 * 3 / 2 ? 16u << 10 : 64u << 10);
    while (n_in / chunk > 32768) chunk *= 2;
    const uint32_t n_chunks = (uint32_t)std::max<uint64_t>(1, n_in / chunk);
    if ((rc = mk::ensure_device(&c->d_gz_in, &c->gz_in_cap, n_in + mkz::kPad + 16))) return rc;
    // tables, all 64-bit: starts[n_chunks] | seg_bits[J + 1] | seg_off[J] | seg_cap[J] | n_out[J] | text_off[J] | status (i32) [J] | bad
    const size_t tab_words = (size_t)n_chunks * 7 + 16;
    if ((rc = mk::ensure_device(&c->d_gz_tab, &c->gz_tab_cap, tab_words * 8))) return rc;
    unsigned long long *d_starts = (unsigned long long *)c->d_gz_tab;
    double t0 = now_ms();
    if ((rc = upload(c, c->d_gz_in, gz + p, n_in))) return rc;
    MKC_HIP(hipMemsetAsync((uint8_t *)c->d_gz_in + n_in, 0, mkz::kPad + 16, c->stream), "hipMemsetAsync");
    MKC_HIP(hipStreamSynchronize(c->stream), "upload of the stream");
    c->gz_ms[0] = (float)(now_ms() - t0);
    // ---- block starts
    t0 = now_ms();
    std::vector<unsigned long long> starts(n_chunks, ~0ull);
    mkz::launch_gzip_find((const uint8_t *)c->d_gz_in, n_in, chunk, n_chunks, chunk, d_starts, c->stream);  // (a start behind the next cut is the next chunk's to find)
    MKC_HIP(hipGetLastError(), "gzip block search");
    if (n_chunks > 1) MKC_HIP(hipMemcpyAsync(starts.data() + 1, d_starts + 1, (n_chunks - 1) * 8ull, hipMemcpyDeviceToHost, c->stream), "download of the block starts");
    MKC_HIP(hipStreamSynchronize(c->stream), "gzip block search");
    c->gz_ms[1] = (float)(now_ms() - t0);
    std::vector<unsigned long long> seg_bits{0ull};
    for (uint32_t k = 1; k < n_chunks; ++k)
        if (starts[k] != ~0ull && starts[k] > seg_bits.back()) seg_bits.push_back(starts[k]);
    uint32_t J = (uint32_t)seg_bits.size();
    seg_bits.push_back(~0ull);
    // ---- segments -> symbols.  Room per segment: 12 x its compressed bytes (FASTQ / FASTA / text: 3-6 x), then 48 x once more.
    // A start the search took for one and that is none (as good as never: gzip_segments.hpp) shows here: the piece in front of it
    // runs over it (kSegDesync, and says at which block boundary it stands).  Such starts are dropped and the pieces decoded
    // again, a few times at most.
    std::vector<unsigned long long> seg_off, seg_cap, n_out, text_off;
    std::vector<int32_t> status;
    unsigned long long *d_bits = nullptr, *d_off = nullptr, *d_cap = nullptr, *d_nout = nullptr, *d_toff = nullptr;
    int32_t *d_status = nullptr;
    uint32_t *d_bad = nullptr;
    uint64_t total = 0;
    bool done = false;
    t0 = now_ms();
    // (ISIZE says how much text there is in all -- below 4 GiB of it: the first guess per piece is half as much again as the stream's
    // own ratio, which spares a 1.27 GB FASTQ two thirds of a 6 GB allocation; a stream of 4 GiB of text or more starts at 12)
    const uint64_t whole_ratio = n_in ? (uint64_t)want_isize / n_in + 1 : 1;
    uint64_t ratio = (n_in < (1ull << 29) && want_isize > n_in) ? std::min<uint64_t>(12, whole_ratio + whole_ratio / 2 + 1) : 12;
    for (int attempt = 0, dropped_rounds = 0; attempt < 6 && !done; ++attempt) {
        seg_off.assign(J, 0), seg_cap.assign(J, 0), n_out.assign(J, 0), status.assign(J, 0);
        d_bits = d_starts + n_chunks, d_off = d_bits + (J + 1), d_cap = d_off + J, d_nout = d_cap + J, d_toff = d_nout + J;
        d_status = (int32_t *)(d_toff + J);
        d_bad = (uint32_t *)(d_status + J + (J & 1));
        uint64_t elems = 0;
        for (uint32_t j = 0; j < J; ++j) {
            const uint64_t b0 = seg_bits[j] >> 3, b1 = j + 1 < J ? seg_bits[j + 1] >> 3 : n_in;
            seg_cap[j] = 65536 + ratio * (b1 - b0 + 1);
            seg_off[j] = elems;
            elems += mkz::kSegPrefix + seg_cap[j] + mkz::kSegSlack;
            elems = (elems + 7) & ~7ull;  // (16-byte aligned buffers)
        }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && elems * 2 > c->gz_sym_cap && elems * 2 - c->gz_sym_cap > free_b / 2) return MK_OK;  // not taken: too large for this device
        if ((rc = mk::ensure_device(&c->d_gz_sym, &c->gz_sym_cap, elems * 2 + 64))) return rc;
        MKC_HIP(hipMemcpyAsync(d_bits, seg_bits.data(), (J + 1) * 8ull, hipMemcpyHostToDevice, c->stream), "upload of the segment table");
        MKC_HIP(hipMemcpyAsync(d_off, seg_off.data(), J * 8ull, hipMemcpyHostToDevice, c->stream), "upload of the segment table");
        MKC_HIP(hipMemcpyAsync(d_cap, seg_cap.data(), J * 8ull, hipMemcpyHostToDevice, c->stream), "upload of the segment table");
        mkz::launch_gzip_segments((const uint8_t *)c->d_gz_in, n_in, d_bits, d_off, d_cap, J, (uint16_t *)c->d_gz_sym, d_nout, d_status, c->num_cus, c->stream,
                                  c->inflate_kernel == 1);
        MKC_HIP(hipGetLastError(), "gzip segment decode");
        MKC_HIP(hipMemcpyAsync(n_out.data(), d_nout, J * 8ull, hipMemcpyDeviceToHost, c->stream), "download");
        MKC_HIP(hipMemcpyAsync(status.data(), d_status, J * 4ull, hipMemcpyDeviceToHost, c->stream), "download");
        MKC_HIP(hipStreamSynchronize(c->stream), "gzip segment decode");
        // the chain from the stream's first bit: a piece that ends on the next start proves that start; the first piece that runs
        // over its end disproves every start in front of the block boundary it stands at (what lies behind is looked at next time)
        bool overflow = false, failed = false;
        uint32_t bad_from = J;
        for (uint32_t j = 0; j < J && bad_from == J; ++j) {
            if (status[j] == mkz::kSegDesync && j + 1 < J) bad_from = j;
            else if (status[j] == mkz::kSegOverflow) overflow = true;
            else if (status[j] != 0) failed = true;
        }
        if (failed) return MK_OK;  // (a segment that does not decode, or a stream without its final block: zlib will say what this file is)
        if (bad_from < J) {
            if (++dropped_rounds > 3) return MK_OK;
            const unsigned long long stands_at = n_out[bad_from];
            std::vector<unsigned long long> kept(seg_bits.begin(), seg_bits.begin() + bad_from + 1);
            for (uint32_t j = bad_from + 1; j < J; ++j)
                if (seg_bits[j] >= stands_at) kept.push_back(seg_bits[j]);
            if (kept.size() == J) return MK_OK;  // (nothing to drop: not what this is for)
            J = (uint32_t)kept.size();
            kept.push_back(~0ull);
            seg_bits.swap(kept);
            continue;
        }
        for (uint32_t j = 0; j < J; ++j) overflow = overflow || status[j] == mkz::kSegOverflow;
        if (!overflow) done = true;
        else if (ratio == 48) break;
        else ratio = 48;
    }
    c->gz_segments = J;
    c->gz_ms[2] = (float)(now_ms() - t0);
    if (!done) return MK_OK;
    text_off.assign(J, 0);
    for (uint32_t j = 0; j < J; ++j) text_off[j] = total, total += n_out[j];
    if ((uint32_t)total != want_isize) return MK_OK;
    // ---- contexts, text, CRC-32
    t0 = now_ms();
    // (the contexts, and behind them two sets of J - 1 maps of 32 768 16-bit elements: launch_gzip_resolve)
    if ((rc = mk::ensure_device(&c->d_gz_ctx, &c->gz_ctx_cap, (size_t)J * mkz::kSegPrefix * 5 + 16))) return rc;
    if ((rc = mk::ensure_device(&c->d_gz_text, &c->gz_text_cap, total + 64))) return rc;
    MKC_HIP(hipMemcpyAsync(d_toff, text_off.data(), J * 8ull, hipMemcpyHostToDevice, c->stream), "upload of the text offsets");
    MKC_HIP(hipMemsetAsync(d_bad, 0, 4, c->stream), "hipMemsetAsync");
    mkz::launch_gzip_resolve((const uint16_t *)c->d_gz_sym, d_off, d_nout, d_toff, J, (uint8_t *)c->d_gz_ctx, (uint8_t *)c->d_gz_text, d_bad, c->stream);
    MKC_HIP(hipGetLastError(), "gzip resolution");
    uint32_t bad = 0;
    MKC_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream), "download");
    MKC_HIP(hipStreamSynchronize(c->stream), "gzip resolution");
    c->gz_ms[3] = (float)(now_ms() - t0);
    if (bad) return MK_OK;
    t0 = now_ms();
    {
        // CRC-32 of the text: per 65 280-byte block on the device (the BGZF writer's kernel), folded here:
        // crc(A ++ B) = crc(A) * x^(8 |B|) mod P  ^  crc(B)
        const uint32_t bb = mkz::kMaxBlockBytes;
        const uint32_t blocks = (uint32_t)((total + bb - 1) / bb);
        if ((rc = mk::ensure_device(&c->d_crc, &c->crc_cap, blocks * 4ull + 16))) return rc;
        mkz::launch_crc((const uint8_t *)c->d_gz_text, total, bb, blocks, (uint32_t *)c->d_crc, c->stream);
        std::vector<uint32_t> crcs(blocks);
        MKC_HIP(hipGetLastError(), "CRC kernel");
        if (blocks) MKC_HIP(hipMemcpyAsync(crcs.data(), c->d_crc, blocks * 4ull, hipMemcpyDeviceToHost, c->stream), "download of the CRCs");
        MKC_HIP(hipStreamSynchronize(c->stream), "CRC kernel");
        const uint32_t shift_full = mkz::crc_x_pow_bytes(bb);
        uint32_t crc = 0;
        for (uint32_t b = 0; b < blocks; ++b) {
            const uint64_t len = b + 1 < blocks ? bb : total - (uint64_t)b * bb;
            crc = mkz::crc_mulmod(crc, len == bb ? shift_full : mkz::crc_x_pow_bytes(len)) ^ crcs[b];
        }
        if (crc != want_crc) return MK_OK;
    }
    c->gz_ms[4] = (float)(now_ms() - t0);
    c->gz_text_bytes = total;
    *text_bytes = total, *taken = 1;
    // what is left to keep is the text: the symbols (24 x the compressed bytes) and the contexts go back to the device now
    for (void **q : {&c->d_gz_sym, &c->d_gz_ctx, &c->d_gz_in}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    c->gz_sym_cap = c->gz_ctx_cap = c->gz_in_cap = 0;
    return MK_OK;
    MK_ABI_END
}

int mk_gzip_text_read(mk_codec *c, uint64_t offset, uint8_t *out, uint64_t len) {
    MK_ABI_BEGIN
    if (!c || (len && !out)) return mk::fail(MK_E_INVALID_ARG, "mk_gzip_text_read: NULL argument");
    std::lock_guard<std::mutex> lock(c->mu);
    if (offset > c->gz_text_bytes || len > c->gz_text_bytes - offset) return mk::fail(MK_E_INVALID_ARG, "mk_gzip_text_read: [%llu, +%llu) is outside the text", (unsigned long long)offset, (unsigned long long)len);
    MKC_HIP(hipSetDevice(c->device), "hipSetDevice");
    return download(c, out, (const uint8_t *)c->d_gz_text + offset, len);
    MK_ABI_END
}

const void *mk_gzip_text_device(const mk_codec *c, uint64_t *text_bytes) {
    if (text_bytes) *text_bytes = c ? c->gz_text_bytes : 0;
    return c && c->gz_text_bytes ? c->d_gz_text : nullptr;
}

int mk_gzip_text_release(mk_codec *c) {
    if (!c) return mk::fail(MK_E_INVALID_ARG, "mk_gzip_text_release: NULL handle");
    std::lock_guard<std::mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    for (void **p : {&c->d_gz_in, &c->d_gz_sym, &c->d_gz_tab, &c->d_gz_ctx, &c->d_gz_text}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->gz_in_cap = c->gz_sym_cap = c->gz_tab_cap = c->gz_ctx_cap = c->gz_text_cap = 0;
    c->gz_text_bytes = 0;
    return MK_OK;
}

int mk_gzip_info(const mk_codec *c, uint32_t *segments, float ms[5]) {
    if (!c) return mk::fail(MK_E_INVALID_ARG, "mk_gzip_info: NULL handle");
    if (segments) *segments = c->gz_segments;
    if (ms)
        for (int k = 0; k < 5; ++k) ms[k] = c->gz_ms[k];
    return MK_OK;
}

int mk_bgzf_inflate(mk_codec *c, const uint8_t *in, uint64_t n_in, const mk_bgzf_member *members, uint64_t n_members, uint8_t *out,
                    uint64_t out_cap, uint64_t *bad_member) {
    MK_ABI_BEGIN
    if (!c || (n_members && (!in || !members)) || (out_cap && !out)) return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_inflate: NULL argument");
    static_assert(sizeof(mk_bgzf_member) == sizeof(mkz::Member), "member layouts");
    if (bad_member) *bad_member = 0;
    for (uint64_t i = 0; i < n_members; ++i) {
        const mk_bgzf_member &m = members[i];
        if (m.data_off > n_in || m.data_len > n_in - m.data_off || m.isize > 65536 || m.out_off > out_cap || m.isize > out_cap - m.out_off) {
            if (bad_member) *bad_member = i;
            return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_inflate: member %llu lies outside its buffers", (unsigned long long)i);
        }
    }
    std::lock_guard<std::mutex> lock(c->mu);
    MKC_HIP(hipSetDevice(c->device), "hipSetDevice");
    c->ms[0] = c->ms[1] = c->ms[2] = 0;
    std::vector<mkz::Member> part;
    std::vector<int32_t> status;
    for (uint64_t m0 = 0; m0 < n_members;) {
        // a run of members whose compressed bytes and text are both contiguous enough to move in one piece each
        uint64_t m1 = m0, text = 0, in_lo = members[m0].data_off, in_hi = in_lo, out_lo = members[m0].out_off, out_hi = out_lo;
        const uint64_t pass_text = c->inflate_pass_text ? c->inflate_pass_text : kInflateChunkText;
        while (m1 < n_members && (m1 == m0 || text + members[m1].isize <= pass_text)) {
            const mk_bgzf_member &m = members[m1];
            in_lo = std::min(in_lo, m.data_off), in_hi = std::max(in_hi, m.data_off + m.data_len);
            out_lo = std::min(out_lo, m.out_off), out_hi = std::max(out_hi, m.out_off + m.isize);
            text += m.isize, ++m1;
        }
        if (out_hi - out_lo > 2 * kInflateChunkText || in_hi - in_lo > 2 * kInflateChunkText)
            return mk::fail(MK_E_INVALID_ARG, "mk_bgzf_inflate: members %llu.. are scattered over more than a device pass holds",
                            (unsigned long long)m0);
        const uint32_t cnt = (uint32_t)(m1 - m0);
        part.resize(cnt);
        for (uint32_t k = 0; k < cnt; ++k) {
            const mk_bgzf_member &m = members[m0 + k];
            part[k] = mkz::Member{m.data_off - in_lo, m.out_off - out_lo, m.data_len, m.isize, m.crc, 0};
        }
        const uint64_t cn = in_hi - in_lo, tn = out_hi - out_lo;
        int rc;
        if ((rc = mk::ensure_device(&c->d_in, &c->in_cap, cn + mkz::kPad)) || (rc = mk::ensure_device(&c->d_aux, &c->aux_cap, cnt * sizeof(mkz::Member))) ||
            (rc = mk::ensure_device(&c->d_out, &c->out_cap, tn + 16)) || (rc = mk::ensure_device(&c->d_len, &c->len_cap, cnt * 4ull)))
            return rc;
        const double t_up = now_ms();
        if ((rc = upload(c, c->d_in, in + in_lo, cn))) return rc;
        MKC_HIP(hipMemsetAsync((uint8_t *)c->d_in + cn, 0, mkz::kPad, c->stream), "hipMemsetAsync");
        MKC_HIP(hipMemcpyAsync(c->d_aux, part.data(), cnt * sizeof(mkz::Member), hipMemcpyHostToDevice, c->stream), "upload of the member table");
        MKC_HIP(hipStreamSynchronize(c->stream), "upload of the members");
        c->ms[0] += (float)(now_ms() - t_up);
        MKC_HIP(hipEventRecord(c->ev[1], c->stream), "hipEventRecord");
        mkz::launch_inflate((const uint8_t *)c->d_in, cn, (const mkz::Member *)c->d_aux, cnt, (uint8_t *)c->d_out, (int32_t *)c->d_len, c->num_cus, c->stream,
                            c->inflate_kernel);
        mkz::launch_crc_check((const uint8_t *)c->d_out, (const mkz::Member *)c->d_aux, cnt, (int32_t *)c->d_len, c->stream);
        MKC_HIP(hipGetLastError(), "BGZF inflate kernels");
        MKC_HIP(hipEventRecord(c->ev[2], c->stream), "hipEventRecord");
        status.resize(cnt);
        MKC_HIP(hipMemcpyAsync(status.data(), c->d_len, cnt * 4ull, hipMemcpyDeviceToHost, c->stream), "download of the status words");
        MKC_HIP(hipStreamSynchronize(c->stream), "BGZF inflate");
        // (a damaged member's text is not handed out: the status words first)
        bool all_ok = true;
        for (uint32_t k = 0; k < cnt && all_ok; ++k) all_ok = status[k] == 0;
        const double t_down = now_ms();
        if (all_ok && tn && (rc = download(c, out + out_lo, c->d_out, tn))) return rc;
        c->ms[1] += elapsed(c->ev[1], c->ev[2]), c->ms[2] += (float)(now_ms() - t_down);
        for (uint32_t k = 0; k < cnt; ++k)
            if (status[k]) {
                if (bad_member) *bad_member = m0 + k;
                if (status[k] > 0)  // the decoder was content (0), the checksum kernel raised its bit
                    return mk::fail(MK_E_CORRUPT, "BGZF member %llu: CRC-32 of the inflated text differs from the trailer's", (unsigned long long)(m0 + k));
                return mk::fail(MK_E_CORRUPT, "BGZF member %llu does not inflate to its ISIZE (decoder status %d)", (unsigned long long)(m0 + k), status[k]);
            }
        m0 = m1;
    }
    return MK_OK;
    MK_ABI_END
}

}  // extern "C"
