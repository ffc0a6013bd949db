// bgzf_inflate.hip -- BGZF members inflated on the device, ONE LANE per member.
//
// Replaces, behind mk_bgzf_inflate (include/merkurio_hip.h), the inflate the reference gets from flate2 inside
// `bam 0.1.4`'s reader (src/cmd_tag.rs:503-506) and needletail's gzip reader for bgzip'ed FASTA/FASTQ
// (src/cmd_extract.rs:281).  DEFLATE decoding is serial inside a stream -- every codeword's position depends on the
// one before it -- but BGZF cuts a file into independent members of <= 64 KiB, thousands per window: the lanes of a
// wave each decode their own member with the serial decoder of inflate_serial.hpp (the code the host harness checks
// against zlib; that header says what shapes it), each lane's decoder block -- canonical code descriptions, symbol
// orders, a 64-byte window of its stream: 836 B -- side by side in dynamic LDS (52 KiB for a full wave of 64 members;
// a call that does not fill the part is spread over narrower waves, see inflate_lanes).  Bound: latency of dependent
// LDS / L2 accesses under divergence, hidden only by the number of members in flight -- not HBM, not MFMA.
// The CRC-32 of every member's text is checked by mk_bgzf_crc_check_kernel (bgzf_deflate.hip) afterwards.
#include <hip/hip_runtime.h>

#include "codec_kernels.h"
#include "inflate_serial.hpp"

namespace mkz {

static_assert(kPad >= kStreamPad, "input buffers carry the decoder's padding");

// blockDim.x = the lanes of a wave that hold a member (1, 2, 4 ... 64): LDS is sized for them at launch
__global__ __launch_bounds__(64, 4) void mk_bgzf_inflate_kernel(const uint8_t *__restrict__ in, uint64_t n_in, const Member *__restrict__ members,
                                                             uint32_t n_members, uint8_t *__restrict__ out, int32_t *__restrict__ status) {
    extern __shared__ uint32_t lanes[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_members) return;
    const Member m = members[i];
    uint16_t *const t = reinterpret_cast<uint16_t *>(lanes + threadIdx.x * (kLaneTableU16 / 2));
    status[i] = inflate_stream(in + m.data_off, m.data_len, out + m.out_off, m.isize, t);
}

// The lanes of a wave turn the decoder's loop together, so a wave is as slow as its slowest lane on every turn, and a launch
// lasts as long as its slowest wave whatever it holds.  A call is therefore spread over as many waves as the part holds at
// once (111 VGPRs: 4 waves per SIMD, 16 per CU, 4 096 on an MI355X) and each wave gets as FEW members as that allows -- down
// to one: 4 012 members (256 MB of text) take 26 ms with one lane per wave, 48 ms with eight, and 64 lanes in lockstep cost
// 2.3 x the single-lane latency of a member (profiles/r04_codec_inflate_steps.txt).  Full waves only when there are members
// for all of them.
uint32_t inflate_lanes(uint32_t n_members, int num_cus) {
    uint32_t lanes = 1;
    while (lanes < 64 && (n_members + lanes - 1) / lanes > (uint32_t)num_cus * 16) lanes *= 2;
    return lanes;
}

// Which decoder a call gets (r05).  A wave per member (bgzf_inflate_wave.hip) finishes a member in 2-5 ms; with only the most recent
// 4 KiB of the member's text in LDS seventeen of them fit a CU (4 352 on the part), with 2 KiB twenty-five: its time grows with the
// members per slot -- and with them per CU: the kernel is bound by its scalar instructions (~770 000 per member on one scalar unit
// per CU).  A lane per member takes 15-25 ms for its slowest lane whatever the call holds and stays there up to tens of thousands
// of members.  zlib level-6 members of BAM records (FASTQ), kernels: 64 MB 5.3 (3.4) ms with the 4 KiB ring, 5.4 (3.5) with 2 KiB,
// 21 (14) with a lane per member; 256 MB 10.2 (6.7) / 9.0 (5.6) / 23 (16); 1 GiB 30.4 (19.6) / 26.9 (17.3) / 40.5 (25.0); at 2 GiB
// the lane kernel's 41 (25) ms win (profiles/r05_codec_real_rings2.txt; once a flush stopped writing the L2 back -- see the
// fence in bgzf_inflate_wave.hip -- the 2 KiB ring, which flushes twice as often, became the better one for all but small calls).
constexpr uint32_t kWaveMembersPerCuSmall = 8, kWaveMembersPerCuMax = 96;  // (<= 2 048 members: the 4 KiB ring; <= 24 576: 2 KiB)

void launch_inflate(const uint8_t *in, uint64_t n_in, const Member *members, uint32_t n_members, uint8_t *out, int32_t *status, int num_cus,
                    hipStream_t s, int which) {
    if (!n_members) return;
    if (which >= 3 && which <= 6) {
        launch_inflate_wave(in, n_in, members, n_members, out, status, s, which == 3 ? 8192u : which == 4 ? 16384u : which == 5 ? 4096u : 2048u);
        return;
    }
    if (which == 2) {
        launch_inflate_wave(in, n_in, members, n_members, out, status, s);
        return;
    }
    if (which == 0 && n_members <= (uint32_t)num_cus * kWaveMembersPerCuMax) {
        launch_inflate_wave(in, n_in, members, n_members, out, status, s, n_members <= (uint32_t)num_cus * kWaveMembersPerCuSmall ? 4096u : 2048u);
        return;
    }
    const uint32_t lanes = inflate_lanes(n_members, num_cus);
    hipLaunchKernelGGL(mk_bgzf_inflate_kernel, dim3((n_members + lanes - 1) / lanes), dim3(lanes), lanes * (kLaneTableU16 / 2) * 4, s, in, n_in, members,
                       n_members, out, status);
}

}  // namespace mkz
