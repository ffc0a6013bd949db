// gzip_segments_wave.hip -- the pieces of ONE gzip stream decoded a WAVE per piece (r05): the kernel around wave_inflate_segment()
// of gzip_segments_wave.hpp.  mk_gzip_segments_kernel (gzip_inflate.hip) gives a piece to a lane, which runs the serial decoder of
// gzip_segments.hpp: ~2 us per token, 76-82 ms for the 3 700 pieces of a 1.27 GB FASTQ -- half of the call; here 43-46 ms.
// Same results symbol for symbol (the host harness checks inflate_segment() against zlib; the GPU tests check the text both lead to
// against zlib, and that both take the same streams).
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "gzip_segments_wave.hpp"

namespace mkz {

__global__ __launch_bounds__(64) void mk_gzip_segments_wave_kernel(const uint8_t *__restrict__ in, uint64_t n_in, const unsigned long long *__restrict__ seg_bits,
                                                                   const unsigned long long *__restrict__ seg_off, const unsigned long long *__restrict__ seg_cap,
                                                                   uint32_t n_seg, uint16_t *__restrict__ sym, unsigned long long *__restrict__ n_out,
                                                                   int32_t *__restrict__ status_out) {
    __shared__ __attribute__((aligned(16))) uint16_t ring[kSegRing];
    __shared__ SegWaveTables S;
    const uint32_t j = blockIdx.x;
    if (j >= n_seg) return;
    const uint64_t bit_end = seg_bits[j + 1];
    uint64_t produced = 0, stop = 0;
    bool fin = false;
    int status = wave_inflate_segment<false>(in, n_in, seg_bits[j], bit_end, 0, sym + seg_off[j] + kSegPrefix, seg_cap[j], ring, S, 0, &produced, &stop, &fin);
    // both ends must be what the search said they are: the next start reached exactly (and not behind the final block), or the stream's end
    if (status == 0 && (bit_end != ~0ull ? (stop != bit_end || fin) : !fin)) status = kSegDesync;
    if (lane_id() == 0) {
        n_out[j] = status == kSegDesync ? stop : produced;  // (a piece that ran over its end: where it stands, a block boundary)
        status_out[j] = status;
    }
}

void launch_gzip_segments_wave(const uint8_t *in, uint64_t n_in, const unsigned long long *seg_bits, const unsigned long long *seg_off,
                               const unsigned long long *seg_cap, uint32_t n_seg, uint16_t *sym, unsigned long long *n_out, int32_t *status, hipStream_t s) {
    if (!n_seg) return;
    hipLaunchKernelGGL(mk_gzip_segments_wave_kernel, dim3(n_seg), dim3(64), 0, s, in, n_in, seg_bits, seg_off, seg_cap, n_seg, sym, n_out, status);
}

}  // namespace mkz
