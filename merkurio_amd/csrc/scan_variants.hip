// scan_variants.hip -- instantiates the scan kernel (scan_kernel_impl.hpp) for one GROUP of variants.
// Compiled once per group with -DMK_TU=<n> (merkurio_amd/build.py runs the groups in parallel: the
// ~80 variants take minutes in one translation unit).  S = sampling stride, QC = q-gram length
// (> 0 fixed at compile time, 0 runtime q <= 16, -1 runtime q in 17..32), GF = filter in global memory.
#include "scan_kernel_impl.hpp"

#ifndef MK_TU
#error "compile with -DMK_TU=0..16"
#endif

namespace mk {

template <int S, int QC, bool EMIT, bool GF, int FL, int MC>
void launch_variant(const ScanParams &p, int grid_blocks, hipStream_t stream) {
    hipLaunchKernelGGL((mk_scan_kernel<S, QC, EMIT, GF, FL, MC>), dim3(grid_blocks), dim3(kBlockThreads), 0, stream, p);
}

#define MK_INST(S_, QC_, GF_)                                                                    \
    template void launch_variant<S_, QC_, false, GF_, 1, 0>(const ScanParams &, int, hipStream_t); \
    template void launch_variant<S_, QC_, true, GF_, 1, 0>(const ScanParams &, int, hipStream_t)
// two length classes (filter.hpp): the variant with the short class's table probed next to the main filter;
// MC_ = 1: stride / table kind of the short class at run time, 2 / 4 / 8: byte table with that stride compiled in
#define MK_INST_MC(S_, QC_, MC_)                                                                       \
    template void launch_variant<S_, QC_, false, false, 1, MC_>(const ScanParams &, int, hipStream_t); \
    template void launch_variant<S_, QC_, true, false, 1, MC_>(const ScanParams &, int, hipStream_t)
// two length classes next to a main filter in global memory (runtime-q kernels)
#define MK_INST_MC_GF(S_, QC_)                                                                       \
    template void launch_variant<S_, QC_, false, true, 1, 1>(const ScanParams &, int, hipStream_t); \
    template void launch_variant<S_, QC_, true, true, 1, 1>(const ScanParams &, int, hipStream_t)
// the same variant with plain (cacheable) stream loads, for hit-dense text
#define MK_INST_PLAIN(S_, QC_, GF_)                                                               \
    template void launch_variant<S_, QC_, false, GF_, 0, 0>(const ScanParams &, int, hipStream_t); \
    template void launch_variant<S_, QC_, true, GF_, 0, 0>(const ScanParams &, int, hipStream_t)

#if MK_TU == 0  // LDS filter, q fixed at compile time: the 31-mer and 21-mer families
uint32_t scan_lds_bytes() { return kLdsBytes; }
MK_INST(16, 16, false);
MK_INST(8, 24, false);
MK_INST(4, 28, false);
MK_INST(4, 18, false);
#elif MK_TU == 6  // the k-mer families again, plain stream loads
MK_INST_PLAIN(16, 16, false);
MK_INST_PLAIN(8, 24, false);
MK_INST_PLAIN(4, 28, false);
MK_INST_PLAIN(4, 18, false);
#elif MK_TU == 7  // plain stream loads, runtime q <= 16
MK_INST_PLAIN(1, 0, false);
MK_INST_PLAIN(2, 0, false);
MK_INST_PLAIN(4, 0, false);
MK_INST_PLAIN(8, 0, false);
MK_INST_PLAIN(16, 0, false);
#elif MK_TU == 8  // plain stream loads, runtime q in 17..32
MK_INST_PLAIN(1, -1, false);
MK_INST_PLAIN(2, -1, false);
MK_INST_PLAIN(4, -1, false);
MK_INST_PLAIN(8, -1, false);
MK_INST_PLAIN(16, -1, false);
#elif MK_TU == 1  // LDS filter, runtime q <= 16
MK_INST(1, 0, false);
MK_INST(2, 0, false);
MK_INST(4, 0, false);
MK_INST(8, 0, false);
MK_INST(16, 0, false);
#elif MK_TU == 2  // LDS filter, runtime q in 17..32
MK_INST(1, -1, false);
MK_INST(2, -1, false);
MK_INST(4, -1, false);
MK_INST(8, -1, false);
MK_INST(16, -1, false);
#elif MK_TU == 3  // global filter, q fixed
MK_INST(8, 14, true);
MK_INST(4, 18, true);
MK_INST(8, 24, true);
#elif MK_TU == 4  // global filter, runtime q <= 16
MK_INST(1, 0, true);
MK_INST(2, 0, true);
MK_INST(4, 0, true);
MK_INST(8, 0, true);
MK_INST(16, 0, true);
#elif MK_TU == 5  // global filter, runtime q in 17..32
MK_INST(1, -1, true);
MK_INST(2, -1, true);
MK_INST(4, -1, true);
MK_INST(8, -1, true);
MK_INST(16, -1, true);
#elif MK_TU == 9  // two length classes, main class with q fixed at compile time, short class at run time
MK_INST_MC(16, 16, 1);
MK_INST_MC(8, 24, 1);
MK_INST_MC(4, 28, 1);
MK_INST_MC(4, 18, 1);
#elif MK_TU == 10  // two length classes, main class runtime q <= 16 (a main class at stride 1 is never split: matcher.cpp)
MK_INST_MC(2, 0, 1);
MK_INST_MC(4, 0, 1);
MK_INST_MC(8, 0, 1);
MK_INST_MC(16, 0, 1);
#elif MK_TU == 11  // two length classes, main class runtime q in 17..32
MK_INST_MC(2, -1, 1);
MK_INST_MC(4, -1, 1);
MK_INST_MC(8, -1, 1);
MK_INST_MC(16, -1, 1);
#elif MK_TU == 12  // the k-mer families with the short class's stride compiled in (byte table): stride 2
MK_INST_MC(16, 16, 2);
MK_INST_MC(8, 24, 2);
MK_INST_MC(4, 28, 2);
MK_INST_MC(4, 18, 2);
#elif MK_TU == 13  // ... stride 4
MK_INST_MC(16, 16, 4);
MK_INST_MC(8, 24, 4);
MK_INST_MC(4, 28, 4);
MK_INST_MC(4, 18, 4);
#elif MK_TU == 14  // ... stride 8
MK_INST_MC(16, 16, 8);
MK_INST_MC(8, 24, 8);
MK_INST_MC(4, 28, 8);
MK_INST_MC(4, 18, 8);
#elif MK_TU == 15  // two length classes, main filter in global memory, runtime q <= 16
MK_INST_MC_GF(2, 0);
MK_INST_MC_GF(4, 0);
MK_INST_MC_GF(8, 0);
MK_INST_MC_GF(16, 0);
#elif MK_TU == 16  // ... runtime q in 17..32
MK_INST_MC_GF(2, -1);
MK_INST_MC_GF(4, -1);
MK_INST_MC_GF(8, -1);
MK_INST_MC_GF(16, -1);
#else
#error "unknown MK_TU"
#endif

}  // namespace mk
