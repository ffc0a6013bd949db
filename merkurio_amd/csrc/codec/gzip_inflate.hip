// gzip_inflate.hip -- one gzip member (ONE DEFLATE stream) inflated in parallel on the device (r05): the kernels around the serial
// pieces of gzip_segments.hpp, which says what the scheme is.  Replaces needletail's gzip reader under `merkurio extract`
// (src/cmd_extract.rs:281-282) for files plain gzip wrote; mk_gzip_inflate_device (codec_host.cpp) drives it.
//
//   mk_gzip_find_kernel      a wave per nominal chunk of the stream: 64 bit positions per step through the three levels of the header
//                            test, each level on full waves (survivors queued in LDS); a survivor of all three is the chunk's start if the
//                            first 2 048 symbols of its block decode (by the wave, dry: gzip_segments_wave.hpp)
//   mk_gzip_prefix_kernel    the place-holders in front of every segment's symbol buffer
//   mk_gzip_segments_wave_kernel (gzip_segments_wave.hip) a wave per segment: the default
//   mk_gzip_segments_kernel  a lane per segment (1 .. 64 per wave, as few as the part's residency allows: the lanes of a wave move in
//                            lockstep): inflate_segment into 16-bit symbols; tables in LDS, 836 B per lane
//   mk_gzip_maps_*_kernel    the 32 KiB context in front of every segment: "context j + 1 in terms of context j" composed over
//                            doubling distances, log2(segments) rounds over all segments at once
//   mk_gzip_translate_kernel a workgroup per segment: symbols -> bytes through the segment's context, coalesced
// Bound: latency of the serial decode per segment, hidden by the number of segments in flight (thousands) -- not HBM, not MFMA.
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "gzip_segments.hpp"
#include "gzip_segments_wave.hpp"

namespace mkz {

// The test of a position has three levels of very different cost and pass rate (gzip_segments.hpp); run position by position, one
// lane of 64 is in the expensive one at any time and the others wait for it (53-58 ms of a 156 ms call).  Here the levels are run
// on FULL waves: level 1 on all 64 positions of a step from three wave-uniform dwords (no per-lane stream reader), its survivors
// queued in LDS; level 2 (the code-length code) on 64 queued positions at a time, its survivors queued again; level 3 (all code
// lengths) on what gathers there, at the latest every kDeepEvery steps.  Any start in the range will do:
// the pieces are cut wherever starts were found.
constexpr uint32_t kFindQueue = 192, kDeepEvery = 128;

__global__ __launch_bounds__(64) void mk_gzip_find_kernel(const uint8_t *__restrict__ in, uint64_t n_in, uint64_t chunk_bytes, uint32_t n_chunks,
                                                          uint64_t search_bytes, unsigned long long *__restrict__ starts) {
    __shared__ SegWaveTables S;                          // (the decode tables of the look at a candidate's block)
    __shared__ uint32_t q1[kFindQueue], q2[kFindQueue];  // bit positions relative to bit0
    const uint32_t c = blockIdx.x + 1;  // (chunk 0 starts where the stream does)
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t bit0 = (uint64_t)c * chunk_bytes * 8;
    const uint64_t bit1 = c + 1 == n_chunks ? n_in * 8 : min(bit0 + search_bytes * 8, n_in * 8);  // (the last one: to the stream's end)
    unsigned long long found = ~0ull;
    uint32_t n1 = 0, n2 = 0, steps = 0;  // (wave-uniform)
    auto push = [&](uint32_t *q, uint32_t &n, bool mine, uint32_t value) {
        const uint64_t m = __ballot(mine);
        if (mine) q[n + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = value;
        n += (uint32_t)__popcll(m);
    };
    // One loop, one place per level (each of the two expensive tests is instantiated once): what to do next is decided from the
    // queues' fill -- the deepest level that has a full wave of work (or, once the range is exhausted, any work) goes first.
    uint64_t base = bit0, buf_bit = bit0;  // (bit0 is a multiple of 32: cuts are multiples of 4 KiB)
    uint32_t look;
    {
        const uint64_t byte = (bit0 >> 3) + 4 * lane;
        look = byte <= n_in + 8 ? load_le32(in + byte) : 0u;
    }
    bool draining = false, deep_due = false;
    while (found == ~0ull) {
        if (n2 >= 64 || (n2 && (draining ? n1 == 0 : deep_due))) {
            // level 3 on the positions in q2 (at most 127 are there: the first 64 now)
            const uint32_t take = min(n2, 64u);
            const bool have = lane < take;
            const uint32_t rel = have ? q2[lane] : 0u;
            const uint64_t bit = bit0 + rel;
            const bool pass = have && seg_header_plausible(in, n_in, bit);
            uint64_t m = __ballot(pass);
            while (m && found == ~0ull) {  // the survivors, lowest queue slot first: the wave decodes the first symbols of the block
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint64_t at = bit0 + (uint32_t)__builtin_amdgcn_readlane((int)rel, (int)l);
                if (wave_confirm_block_start(in, n_in, at, S)) found = at;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t rest = n2 - take;
            uint32_t a = 0;
            if (lane < rest) a = q2[take + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) q2[lane] = a;
            n2 = rest;
            deep_due = false;
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (n1 >= 64 || (n1 && draining)) {
            // level 2 on the first min(n1, 64) positions of q1
            const uint32_t take = min(n1, 64u);
            const bool have = lane < take;
            const uint32_t rel = have ? q1[lane] : 0u;
            const bool pass = have && seg_header_cl_plausible(in, n_in, bit0 + rel);
            __builtin_amdgcn_wave_barrier();
            const uint32_t rest = n1 - take;  // (moves to the queue's front: at most 63 entries)
            uint32_t a = 0;
            if (lane < rest) a = q1[take + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) q1[lane] = a;
            n1 = rest;
            push(q2, n2, pass, rel);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (base < bit1) {
            // level 1: the 13 bits at (base + lane) out of the 96 bits that hold all 64 of them -- three dwords out of the 64 the wave
            // holds in registers (a dword per lane, loaded in a row every 31 steps: a step does not wait for memory)
            if (base - buf_bit > 61 * 32) {
                buf_bit = base;
                const uint64_t byte = (base >> 3) + 4 * lane;
                look = byte <= n_in + 8 ? load_le32(in + byte) : 0u;
            }
            const uint32_t k = uni((uint32_t)((base - buf_bit) >> 5));
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)look, (int)k), w1 = (uint32_t)__builtin_amdgcn_readlane((int)look, (int)k + 1),
                           w2 = (uint32_t)__builtin_amdgcn_readlane((int)look, (int)k + 2);
            const uint32_t sh = (uint32_t)(base & 7) + lane;  // 0 .. 70
            const uint64_t lo = (uint64_t)w0 | (uint64_t)w1 << 32, hi = (uint64_t)w1 | (uint64_t)w2 << 32;
            const uint32_t v = (uint32_t)((sh < 32 ? lo >> sh : hi >> (sh - 32)) & 0x1fffu);
            const uint64_t bit = base + lane;
            const bool pass = bit < bit1 && bit + 64 <= n_in * 8 && seg_header_bits_plausible(v);
            push(q1, n1, pass, (uint32_t)(bit - bit0));
            __builtin_amdgcn_wave_barrier();
            base += 64;
            if ((++steps % kDeepEvery) == 0) deep_due = true;
            continue;
        }
        if (draining) break;  // the range is exhausted and both queues are empty
        draining = true;
    }
    if (lane == 0) starts[c] = found;
}
__global__ __launch_bounds__(256) void mk_gzip_prefix_kernel(uint16_t *__restrict__ sym, const unsigned long long *__restrict__ seg_off, uint32_t n_seg) {
    const uint32_t j = blockIdx.y;
    if (j >= n_seg) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;  // (gridDim.x = kSegPrefix / 256)
    sym[seg_off[j] + k] = (uint16_t)(kSegUnknown | k);
}

// seg_bits[j] = first bit of segment j; seg_bits[n_seg] = ~0 (the last one runs to the final block).  seg_off[j] = element offset of
// its buffer in sym (prefix first), seg_cap[j] = its capacity.  -> n_out[j], status[j] (0, or a negative code)
__global__ __launch_bounds__(64, 4) void mk_gzip_segments_kernel(const uint8_t *__restrict__ in, uint64_t n_in, const unsigned long long *__restrict__ seg_bits,
                                                                 const unsigned long long *__restrict__ seg_off, const unsigned long long *__restrict__ seg_cap,
                                                                 uint32_t n_seg, uint16_t *__restrict__ sym, unsigned long long *__restrict__ n_out,
                                                                 int32_t *__restrict__ status) {
    extern __shared__ uint32_t lanes[];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    uint16_t *const t = reinterpret_cast<uint16_t *>(lanes + threadIdx.x * (kLaneTableU16 / 2));
    const uint64_t end = seg_bits[j + 1];
    uint64_t produced = 0, stop = 0;
    bool fin = false;
    int rc = inflate_segment(in, n_in, seg_bits[j], end, 0, sym + seg_off[j] + kSegPrefix, seg_cap[j], t, &produced, &stop, &fin);
    // both ends must be what the search said they are: the next start reached exactly (and not behind the final block), or the stream's end
    if (rc == 0 && (end != ~0ull ? (stop != end || fin) : !fin)) rc = kSegDesync;
    n_out[j] = rc == kSegDesync ? stop : produced;  // (a piece that ran over its end: where it stands, a block boundary)
    status[j] = rc;
}

// ---- contexts: ctx[j] = the 32 KiB of text in front of segment j (ctx[0]: nothing, zeros).
// ctx[j + 1] is the last 32 KiB of (ctx[j] ++ segment j's text) -- a chain through every segment, and in FASTQ a real one: a read
// name's prefix is a match to the previous read's, back to the first read of the file.  Walked in order (one workgroup, the running
// context in LDS: the first version) it costs ~5 us per segment, 19-22 ms of a 108 ms call.  Here the chain is cut by composing MAPS:
// map j (32 768 16-bit elements) says what ctx[j + 1] is in terms of an EARLIER context -- element k is a byte, or the place-holder
// of an element of that context.  Map j starts out in terms of ctx[j] (segment j's last symbols; where the segment is shorter than
// 32 KiB, place-holders of the context's own tail); composing it with map j - r (which gives ctx[j] when r = 1) puts it in terms of
// a context r segments further back.  With r = 1, 2, 4, ... every map is in terms of ctx[0] after log2(segments) rounds: 12 rounds
// of ~0.75 GB of traffic for 3 700 segments.
__global__ __launch_bounds__(256) void mk_gzip_maps_init_kernel(const uint16_t *__restrict__ sym, const unsigned long long *__restrict__ seg_off,
                                                                const unsigned long long *__restrict__ n_out, uint16_t *__restrict__ maps) {
    const uint32_t j = blockIdx.y;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;  // (gridDim.x = kSegPrefix / 256)
    const uint16_t *o = sym + seg_off[j] + kSegPrefix;
    const uint64_t n = n_out[j];
    const uint64_t keep = n >= kSegPrefix ? 0 : kSegPrefix - n;  // elements of the old context that survive
    maps[(uint64_t)j * kSegPrefix + k] = k >= keep ? o[n - (kSegPrefix - k)] : (uint16_t)(kSegUnknown | (uint32_t)(k + n));
}
// dst[j] = src[j] composed with src[j - r] (j >= r), else src[j].  8 elements per thread.
__global__ __launch_bounds__(256) void mk_gzip_maps_step_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, uint32_t r) {
    const uint32_t j = blockIdx.y;
    const uint32_t k = (blockIdx.x * 256 + threadIdx.x) * 8;  // (gridDim.x = kSegPrefix / 2048)
    const uint4 v = *reinterpret_cast<const uint4 *>(src + (uint64_t)j * kSegPrefix + k);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if (j >= r) {
        const uint16_t *back = src + (uint64_t)(j - r) * kSegPrefix;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t lo = w[q] & 0xffffu, hi = w[q] >> 16;
            if (lo & kSegUnknown) lo = back[lo & 0x7fffu];
            if (hi & kSegUnknown) hi = back[hi & 0x7fffu];
            w[q] = lo | hi << 16;
        }
    }
    *reinterpret_cast<uint4 *>(dst + (uint64_t)j * kSegPrefix + k) = make_uint4(w[0], w[1], w[2], w[3]);
}
// ctx[j + 1] from map j, now in terms of ctx[0] = zeros (blockIdx.y = 0: ctx[0] itself).  8 elements per thread.
__global__ __launch_bounds__(256) void mk_gzip_maps_final_kernel(const uint16_t *__restrict__ maps, uint8_t *__restrict__ ctx) {
    const uint32_t k = (blockIdx.x * 256 + threadIdx.x) * 8;
    uint2 bytes = make_uint2(0, 0);
    if (blockIdx.y) {
        const uint4 v = *reinterpret_cast<const uint4 *>(maps + (uint64_t)(blockIdx.y - 1) * kSegPrefix + k);
        auto two = [](uint32_t w) -> uint32_t {  // two elements -> two bytes (a place-holder of ctx[0]: 0)
            const uint32_t lo = w & 0xffffu, hi = w >> 16;
            return ((lo & kSegUnknown) ? 0u : lo & 0xffu) | ((hi & kSegUnknown) ? 0u : hi & 0xffu) << 8;
        };
        bytes = make_uint2(two(v.x) | two(v.y) << 16, two(v.z) | two(v.w) << 16);
    }
    *reinterpret_cast<uint2 *>(ctx + (uint64_t)blockIdx.y * kSegPrefix + k) = bytes;
}

// text[text_off[j] + i] = the byte symbol i of segment j stands for
__global__ __launch_bounds__(1024) void mk_gzip_translate_kernel(const uint16_t *__restrict__ sym, const unsigned long long *__restrict__ seg_off,
                                                                 const unsigned long long *__restrict__ n_out, const unsigned long long *__restrict__ text_off,
                                                                 const uint8_t *__restrict__ ctx, uint32_t n_seg, uint8_t *__restrict__ text, uint32_t *__restrict__ bad) {
    __shared__ uint8_t c[kSegPrefix];
    const uint32_t j = blockIdx.x;
    if (j >= n_seg) return;
    const uint8_t *cj = ctx + (uint64_t)j * kSegPrefix;
    for (uint32_t k = threadIdx.x * 16; k < kSegPrefix; k += 1024 * 16) *reinterpret_cast<uint4 *>(&c[k]) = *reinterpret_cast<const uint4 *>(&cj[k]);
    __syncthreads();
    const uint16_t *o = sym + seg_off[j] + kSegPrefix;
    uint8_t *dst = text + text_off[j];
    const uint64_t n = n_out[j];
    uint32_t wrong = 0;
    for (uint64_t i = threadIdx.x; i < n; i += 1024) {
        const uint16_t v = o[i];
        wrong |= (v >= 256 && !(v & kSegUnknown));
        dst[i] = (v & kSegUnknown) ? c[v & 0x7fffu] : (uint8_t)v;
    }
    if (wrong) atomicOr(bad, 1u);
}

void launch_gzip_find(const uint8_t *in, uint64_t n_in, uint64_t chunk_bytes, uint32_t n_chunks, uint64_t search_bytes, unsigned long long *starts,
                      hipStream_t s) {
    if (n_chunks > 1)
        hipLaunchKernelGGL(mk_gzip_find_kernel, dim3(n_chunks - 1), dim3(64), 0, s, in, n_in, chunk_bytes, n_chunks, search_bytes, starts);
}

void launch_gzip_segments(const uint8_t *in, uint64_t n_in, const unsigned long long *seg_bits, const unsigned long long *seg_off,
                          const unsigned long long *seg_cap, uint32_t n_seg, uint16_t *sym, unsigned long long *n_out, int32_t *status, int num_cus,
                          hipStream_t s, bool lane_per_segment) {
    if (!n_seg) return;
    hipLaunchKernelGGL(mk_gzip_prefix_kernel, dim3(kSegPrefix / 256, n_seg), dim3(256), 0, s, sym, seg_off, n_seg);
    if (!lane_per_segment) {
        launch_gzip_segments_wave(in, n_in, seg_bits, seg_off, seg_cap, n_seg, sym, n_out, status, s);
        return;
    }
    const uint32_t lanes = inflate_lanes(n_seg, num_cus);
    hipLaunchKernelGGL(mk_gzip_segments_kernel, dim3((n_seg + lanes - 1) / lanes), dim3(lanes), lanes * (kLaneTableU16 / 2) * 4, s, in, n_in, seg_bits, seg_off,
                       seg_cap, n_seg, sym, n_out, status);
}

void launch_gzip_resolve(const uint16_t *sym, const unsigned long long *seg_off, const unsigned long long *n_out, const unsigned long long *text_off,
                         uint32_t n_seg, uint8_t *ctx, uint8_t *text, uint32_t *bad, hipStream_t s) {
    if (!n_seg) return;
    // ctx: n_seg x 32 KiB of contexts, then two sets of n_seg - 1 maps (gzip_resolve_bytes)
    uint16_t *maps[2] = {reinterpret_cast<uint16_t *>(ctx + (uint64_t)n_seg * kSegPrefix), nullptr};
    maps[1] = maps[0] + (uint64_t)(n_seg - 1) * kSegPrefix;
    int cur = 0;
    if (n_seg > 1) {
        hipLaunchKernelGGL(mk_gzip_maps_init_kernel, dim3(kSegPrefix / 256, n_seg - 1), dim3(256), 0, s, sym, seg_off, n_out, maps[0]);
        for (uint32_t r = 1; r < n_seg - 1; r *= 2, cur ^= 1)
            hipLaunchKernelGGL(mk_gzip_maps_step_kernel, dim3(kSegPrefix / 2048, n_seg - 1), dim3(256), 0, s, maps[cur], maps[cur ^ 1], r);
    }
    hipLaunchKernelGGL(mk_gzip_maps_final_kernel, dim3(kSegPrefix / 2048, n_seg), dim3(256), 0, s, maps[cur], ctx);
    hipLaunchKernelGGL(mk_gzip_translate_kernel, dim3(n_seg), dim3(1024), 0, s, sym, seg_off, n_out, text_off, ctx, n_seg, text, bad);
}

}  // namespace mkz
