// order_hits_fallback.hip -- library fallback of the emission-order step (rocPRIM device merge sort over the
// 16-byte tuples with the reference's comparator).  order_hits.hip restores the order with hand-written
// kernels; this path only runs when a batch defeats their binning (one bin above 16384 tuples after both
// binning attempts) or its (record, end, pattern) triple does not fit a 64-bit key.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_merge_sort.hpp>

#include "../../include/merkurio_hip.h"

namespace mk {

struct EmissionOrder {
    const uint32_t *pat_off;  // device: pattern i is pat_off[i+1] - pat_off[i] bytes long
    uint32_t uniform_len;     // != 0: every pattern has this length (no lookup)
    bool ac;
    __device__ __forceinline__ uint32_t len(uint32_t p) const { return uniform_len ? uniform_len : pat_off[p + 1] - pat_off[p]; }
    __device__ __forceinline__ bool operator()(const mk_hit &a, const mk_hit &b) const {
        if (a.rec != b.rec) return a.rec < b.rec;
        if (ac) {  // src/cmd_extract.rs:332-351: end ascending, longer pattern (smaller start) first, pattern id
            const uint64_t ea = (uint64_t)a.pos + len(a.pat), eb = (uint64_t)b.pos + len(b.pat);
            if (ea != eb) return ea < eb;
            if (a.pos != b.pos) return a.pos < b.pos;
            return a.pat < b.pat;
        }
        if (a.pat != b.pat) return a.pat < b.pat;  // src/cmd_extract.rs:365-384: pattern-major
        return a.pos < b.pos;
    }
};

hipError_t order_hits_library(mk_hit *d_hits, size_t n, bool ac, const uint32_t *d_pat_off, uint32_t uniform_len, void *tmp,
                              size_t *tmp_bytes, hipStream_t stream) {
    EmissionOrder cmp{d_pat_off, uniform_len, ac};
    return rocprim::merge_sort(tmp, *tmp_bytes, d_hits, d_hits, n, cmp, stream, false);
}

}  // namespace mk
