// codec_kernels.h -- launchers of the device BGZF codec (bgzf_deflate.hip, bgzf_inflate.hip), called by the host side
// of the C ABI's mk_codec entry points (codec_host.cpp).
//
// BGZF (SAM specification §4.1) = gzip members of at most 64 KiB, each a raw DEFLATE stream (RFC 1951) between an
// 18-byte header (with the member's size in a 'BC' extra field) and CRC-32 + ISIZE.  The reference reads and writes
// them through `bam 0.1.4` / flate2 on `tag`'s reader and writer threads (src/cmd_tag.rs:254-271,503-615) and reads
// bgzip'ed FASTA/FASTQ through needletail (src/cmd_extract.rs:281).  Members are independent: one wave (deflate) or
// one lane (inflate) per member.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mkz {

constexpr uint32_t kSlotBytes = 65536;       // a member never exceeds 64 KiB (BSIZE is 16 bits)
constexpr uint32_t kMaxBlockBytes = 0xff00;  // input bytes per member (what htslib / the host writer use)
constexpr uint32_t kTokensPerWave = 65536;   // token scratch of one resident deflate wave (u32 each)
constexpr uint32_t kPad = 128;               // readable bytes every input buffer needs behind its last byte (inflate_serial.hpp: kStreamPad)

// one member of an inflate call (the host walks the BSIZE chain and fills these)
struct Member {
    uint64_t data_off;  // raw DEFLATE stream inside the compressed buffer
    uint64_t out_off;   // where its text goes
    uint32_t data_len, isize, crc;
    uint32_t pad_;
};

// CRC-32 of every block_bytes-sized block of in[0, n) (the last one may be shorter) -> crc[b]
void launch_crc(const uint8_t *in, uint64_t n, uint32_t block_bytes, uint32_t n_blocks, uint32_t *crc, hipStream_t s);
// CRC-32 of the members' text out[out_off, out_off + isize) compared with Member::crc; status[i] |= 0x100 on a mismatch
void launch_crc_check(const uint8_t *out, const Member *members, uint32_t n_members, int32_t *status, hipStream_t s);

// Deflate: block b of in -> one complete BGZF member in slots[b * kSlotBytes ...], its size in slot_len[b].
// tokens: kTokensPerWave u32 per launched wave (deflate_grid(n_blocks) waves); `in` readable up to in + n + kPad.
uint32_t deflate_grid(uint32_t n_blocks, int num_cus);
// next_block: one u32 of device memory (the kernel's work counter; zeroed by the launcher)
void launch_deflate(const uint8_t *in, uint64_t n, uint32_t block_bytes, uint32_t n_blocks, const uint32_t *crc, uint32_t *tokens,
                    uint8_t *slots, uint32_t *slot_len, uint32_t *next_block, uint32_t grid, hipStream_t s);
// slot_len[0, n_blocks) -> slot_off (exclusive sums, u64) and *total; then the members back to back
void launch_pack(const uint8_t *slots, const uint32_t *slot_len, uint64_t *slot_off, uint64_t *total, uint32_t n_blocks, uint8_t *packed,
                 hipStream_t s);

// Inflate: member i of `in` -> out[out_off, out_off + isize); status[i] = 0 or an error code of inflate_serial.hpp.
// `in` readable up to in + n_in + kPad.
// (1 to 64 members per wave, by how many there are: inflate_lanes)
uint32_t inflate_lanes(uint32_t n_members, int num_cus);
// which: 3 / 4 = a wave per member with a ring of 8 / 16 KiB (measurement); 0 = by the number of members (up to kWaveMembersMax per CU-quarter... see bgzf_inflate.hip), 1 = one lane per member
// (bgzf_inflate.hip), 2 = one wave per member with the text in LDS (bgzf_inflate_wave.hip)
void launch_inflate(const uint8_t *in, uint64_t n_in, const Member *members, uint32_t n_members, uint8_t *out, int32_t *status, int num_cus,
                    hipStream_t s, int which = 0);
// ring_bytes: 32768 (all of a member's reach in LDS, 4 waves per CU), 16384 (7) or 8192 (12; far matches read flushed text back)
void launch_inflate_wave(const uint8_t *in, uint64_t n_in, const Member *members, uint32_t n_members, uint8_t *out, int32_t *status, hipStream_t s,
                         uint32_t ring_bytes = 32768);

// ---- one gzip member inflated in parallel pieces (gzip_inflate.hip, gzip_segments.hpp) ----
// starts[c] (c = 1 .. n_chunks - 1) = bit position of the first confirmed block start at or behind byte c * chunk_bytes, searched
// over search_bytes; ~0 = none
void launch_gzip_find(const uint8_t *in, uint64_t n_in, uint64_t chunk_bytes, uint32_t n_chunks, uint64_t search_bytes, unsigned long long *starts,
                      hipStream_t s);
// segment j = bits [seg_bits[j], seg_bits[j + 1]) (the last entry ~0: to the final block) -> 16-bit symbols in sym[seg_off[j] + 32768 ...],
// at most seg_cap[j] of them; n_out[j], status[j]
void launch_gzip_segments(const uint8_t *in, uint64_t n_in, const unsigned long long *seg_bits, const unsigned long long *seg_off,
                          const unsigned long long *seg_cap, uint32_t n_seg, uint16_t *sym, unsigned long long *n_out, int32_t *status, int num_cus,
                          hipStream_t s, bool lane_per_segment = false);
// (the same with a wave per segment: gzip_segments_wave.hip; the place-holders must lie in front of every segment already)
void launch_gzip_segments_wave(const uint8_t *in, uint64_t n_in, const unsigned long long *seg_bits, const unsigned long long *seg_off,
                               const unsigned long long *seg_cap, uint32_t n_seg, uint16_t *sym, unsigned long long *n_out, int32_t *status, hipStream_t s);
// contexts (n_seg x 32 KiB) and the text: text[text_off[j] ...] = segment j's bytes; *bad != 0: a symbol that is neither
void launch_gzip_resolve(const uint16_t *sym, const unsigned long long *seg_off, const unsigned long long *n_out, const unsigned long long *text_off,
                         uint32_t n_seg, uint8_t *ctx, uint8_t *text, uint32_t *bad, hipStream_t s);

}  // namespace mkz
