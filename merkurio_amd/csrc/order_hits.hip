// order_hits.hip -- the reference's emission order, restored on the device.
//
// The scan kernel writes (record, pattern, position) tuples as its waves find them.  The reference
// emits them in the order of its search loops:
//   Aho-Corasick  find_overlapping_iter per record (src/cmd_extract.rs:332, src/cmd_tag.rs:393-396):
//                 end ascending; at one end the longer pattern (smaller start) first, then pattern id
//   BNDMq         one pattern after the other per record (src/cmd_extract.rs:365-384): pattern-major,
//                 positions ascending
// A batch where every read hits carries 10^7..10^8 tuples: one host thread sorts 3.4 M tuples in 0.4 s
// (more than the scan of the whole batch takes by three orders of magnitude), the device in a few
// milliseconds.  The sort itself is rocPRIM's device merge sort (a plain library sort over 16-byte
// tuples with the comparator below); nothing else of the path uses a library kernel.
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
        if (ac) {
            const uint64_t ea = (uint64_t)a.pos + len(a.pat), eb = (uint64_t)b.pos + len(b.pat);
            if (ea != eb) return ea < eb;
            if (a.pos != b.pos) return a.pos < b.pos;
            return a.pat < b.pat;
        }
        if (a.pat != b.pat) return a.pat < b.pat;
        return a.pos < b.pos;
    }
};

// tmp == nullptr: only *tmp_bytes is set (the scratch the sort of n tuples needs)
hipError_t order_hits_device(mk_hit *d_hits, size_t n, bool ac, const uint32_t *d_pat_off, uint32_t uniform_len, void *tmp,
                             size_t *tmp_bytes, hipStream_t stream) {
    EmissionOrder cmp{d_pat_off, uniform_len, ac};
    return rocprim::merge_sort(tmp, *tmp_bytes, d_hits, d_hits, n, cmp, stream, false);
}

}  // namespace mk
