// gzip_inflate.hip -- one gzip member (ONE DEFLATE stream) inflated in parallel on the device (r05): the kernels around the serial
// pieces of gzip_segments.hpp, which says what the scheme is.  Replaces needletail's gzip reader under `merkurio extract`
// (src/cmd_extract.rs:281-282) for files plain gzip wrote; mk_gzip_inflate_device (codec_host.cpp) drives it.
//
//   mk_gzip_find_kernel      a wave per nominal chunk of the stream: 64 bit positions per step through seg_header_plausible (registers
//                            only; almost every position fails within its first dwords), the rare survivor confirmed by ONE lane that
//                            decodes its block dry and looks for the next header (LDS: one decoder's tables per wave)
//   mk_gzip_prefix_kernel    the place-holders in front of every segment's symbol buffer
//   mk_gzip_segments_kernel  a lane per segment (1 .. 64 per wave, as few as the part's residency allows: the lanes of a wave move in
//                            lockstep): inflate_segment into 16-bit symbols; tables in LDS, 836 B per lane
//   mk_gzip_context_kernel   ONE workgroup walks the segments in order: the 32 KiB context in front of segment j + 1 from the context in
//                            front of segment j (in LDS) and segment j's last symbols -- the only sequential step, ~2 us per segment
//   mk_gzip_translate_kernel a workgroup per segment: symbols -> bytes through the segment's context, coalesced
// Bound: latency of the serial decode per segment, hidden by the number of segments in flight (thousands) -- not HBM, not MFMA.
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "gzip_segments.hpp"

namespace mkz {

__global__ __launch_bounds__(64) void mk_gzip_find_kernel(const uint8_t *__restrict__ in, uint64_t n_in, uint64_t chunk_bytes, uint32_t n_chunks,
                                                          uint64_t search_bytes, unsigned long long *__restrict__ starts) {
    __shared__ uint16_t t[kLaneTableU16];
    const uint32_t c = blockIdx.x + 1;  // (chunk 0 starts where the stream does)
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t bit0 = (uint64_t)c * chunk_bytes * 8;
    const uint64_t bit1 = min(bit0 + search_bytes * 8, n_in * 8);
    unsigned long long found = ~0ull;
    for (uint64_t base = bit0; base < bit1 && found == ~0ull; base += 64) {
        const uint64_t bit = base + lane;
        const bool pass = bit < bit1 && seg_header_plausible(in, n_in, bit);
        uint64_t m = __ballot(pass);
        while (m && found == ~0ull) {  // the survivors of this step, lowest position first: confirmed by their own lane, one at a time
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            int ok = 0;
            if (lane == l) ok = seg_confirm_block_start(in, n_in, bit, t) ? 1 : 0;
            ok = __shfl(ok, (int)l);
            if (ok) found = base + l;
        }
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
    n_out[j] = produced;
    status[j] = rc;
}

// ctx[j] = the 32 KiB of text in front of segment j (ctx[0]: nothing, zeros).  One workgroup of 1024; the running context lives in LDS.
__global__ __launch_bounds__(1024) void mk_gzip_context_kernel(const uint16_t *__restrict__ sym, const unsigned long long *__restrict__ seg_off,
                                                               const unsigned long long *__restrict__ n_out, uint32_t n_seg, uint8_t *__restrict__ ctx) {
    __shared__ uint8_t cur[2][kSegPrefix];
    for (uint32_t k = threadIdx.x; k < kSegPrefix; k += 1024) cur[0][k] = 0, ctx[k] = 0;
    __syncthreads();
    int b = 0;
    for (uint32_t j = 0; j + 1 < n_seg; ++j, b ^= 1) {
        const uint16_t *o = sym + seg_off[j] + kSegPrefix;
        const uint64_t n = n_out[j];
        uint8_t *next = ctx + (uint64_t)(j + 1) * kSegPrefix;
        // the last 32 KiB of (context ++ segment's text): what the segment does not cover comes from the old context, shifted
        const uint64_t keep = n >= kSegPrefix ? 0 : kSegPrefix - n;  // bytes of the old context that survive
        uint16_t v[kSegPrefix / 1024];
#pragma unroll
        for (uint32_t q = 0; q < kSegPrefix / 1024; ++q) {
            const uint32_t k = q * 1024 + threadIdx.x;
            v[q] = k >= keep ? o[n - (kSegPrefix - k)] : (uint16_t)0;
        }
#pragma unroll
        for (uint32_t q = 0; q < kSegPrefix / 1024; ++q) {
            const uint32_t k = q * 1024 + threadIdx.x;
            const uint8_t byte = k < keep ? cur[b][k + n] : ((v[q] & kSegUnknown) ? cur[b][v[q] & 0x7fffu] : (uint8_t)v[q]);
            cur[b ^ 1][k] = byte;
            next[k] = byte;
        }
        __syncthreads();
    }
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
                          hipStream_t s) {
    if (!n_seg) return;
    hipLaunchKernelGGL(mk_gzip_prefix_kernel, dim3(kSegPrefix / 256, n_seg), dim3(256), 0, s, sym, seg_off, n_seg);
    const uint32_t lanes = inflate_lanes(n_seg, num_cus);
    hipLaunchKernelGGL(mk_gzip_segments_kernel, dim3((n_seg + lanes - 1) / lanes), dim3(lanes), lanes * (kLaneTableU16 / 2) * 4, s, in, n_in, seg_bits, seg_off,
                       seg_cap, n_seg, sym, n_out, status);
}

void launch_gzip_resolve(const uint16_t *sym, const unsigned long long *seg_off, const unsigned long long *n_out, const unsigned long long *text_off,
                         uint32_t n_seg, uint8_t *ctx, uint8_t *text, uint32_t *bad, hipStream_t s) {
    if (!n_seg) return;
    hipLaunchKernelGGL(mk_gzip_context_kernel, dim3(1), dim3(1024), 0, s, sym, seg_off, n_out, n_seg, ctx);
    hipLaunchKernelGGL(mk_gzip_translate_kernel, dim3(n_seg), dim3(1024), 0, s, sym, seg_off, n_out, text_off, ctx, n_seg, text, bad);
}

}  // namespace mkz
