// codec_internal.h -- the opaque mk_codec handle (codec_host.cpp; host_loops.cpp borrows its device buffers for
// mk_extract_fastq_bgzf, which inflates a window of members straight into the matcher's text buffer)
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "../host_common.h"

struct mk_codec {
    int device = 0, num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex mu;
    // deflate: input chunk, per-block CRCs / sizes / offsets, token scratch, member slots, packed members
    // inflate: compressed chunk (d_in), member table (d_aux), text (d_out), status words (d_len)
    void *d_in = nullptr, *d_crc = nullptr, *d_tokens = nullptr, *d_slots = nullptr, *d_len = nullptr, *d_off = nullptr, *d_out = nullptr,
         *d_aux = nullptr;
    size_t in_cap = 0, crc_cap = 0, tokens_cap = 0, slots_cap = 0, len_cap = 0, off_cap = 0, out_cap = 0, aux_cap = 0;
    // host buffers travel through two page-locked staging buffers, filled / emptied by the host threads beside the DMA (codec_host.cpp)
    void *h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    // mk_gzip_inflate_device: the compressed stream, the segments' symbols, segment tables, contexts, and the text, which stays
    // here until the next such call or mk_gzip_text_release
    void *d_gz_in = nullptr, *d_gz_sym = nullptr, *d_gz_tab = nullptr, *d_gz_ctx = nullptr, *d_gz_text = nullptr;
    size_t gz_in_cap = 0, gz_sym_cap = 0, gz_tab_cap = 0, gz_ctx_cap = 0, gz_text_cap = 0;
    uint64_t gz_text_bytes = 0;
    uint32_t gz_segments = 0;  // of the last call (diagnostic)
    float gz_ms[5] = {0, 0, 0, 0, 0};  // upload, block search, segments, resolve + translate, CRC
    int inflate_kernel = 0;  // mk_codec_set_inflate_kernel: 0 = chosen per call, 1 = a lane per member, 2 = a wave per member
    float ms[3] = {0, 0, 0};
    uint64_t gzip_chunk = 0;  // mk_codec_set_gzip_chunk; 0 = by the stream's size
    uint64_t deflate_pass_blocks = 0, inflate_pass_text = 0;  // mk_codec_set_pass_limits; 0 = the defaults below
};

