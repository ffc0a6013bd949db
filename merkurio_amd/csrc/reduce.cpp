// reduce.cpp -- the one collective of the path: the sum of the per-GPU counter vectors
// (pattern_hit_counts | summary scalars; SURVEY.md §8e, src/cmd_extract.rs:285-290,
// src/cmd_tag.rs:360-364) over the GPUs of a job, done by RCCL (ncclAllReduce over xGMI).
//
// Two shapes of host program are served:
//   * one process driving several GPUs (the C++ `merkurio --gpus N`): mk_reduce_counters
//     over one handle per device (ncclCommInitAll);
//   * one process per GPU (bench.py, a Rust host under a launcher): mk_comm_unique_id on one
//     rank, the 128 id bytes carried to the others by whatever the host has (a file, MPI,
//     torch.distributed), mk_comm_init on every rank, then mk_comm_reduce_counters.
//
// librccl is bound at run time (dlopen by soname, so that inside a PyTorch process the copy
// PyTorch already mapped is the one used): a host that never reduces needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

// The handful of RCCL declarations this file needs, written out here: the library builds on hosts that have
// the ROCm runtime but not the RCCL development headers, and binds librccl at run time or not at all.  The
// names and values are RCCL's public, stable C ABI (rccl.h: ncclResult_t, ncclDataType_t, ncclRedOp_t,
// NCCL_UNIQUE_ID_BYTES = 128); where the header is installed the static_asserts below compare them.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define MK_HAVE_RCCL_H 1
#endif
#endif
#ifndef MK_HAVE_RCCL_H
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif
static_assert(sizeof(ncclUniqueId) == 128 && (int)ncclUint64 == 5 && (int)ncclSum == 0 && (int)ncclSuccess == 0,
              "RCCL ABI constants differ from the ones declared in reduce.cpp");

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "host_common.h"
#include "matcher_internal.h"

namespace mk {
int hip_fail(hipError_t e, const char *what);
void launch_add_u64(unsigned long long *dst, const unsigned long long *src, size_t len, hipStream_t stream);
}  // namespace mk
using namespace mk;

#define MK_HIP_R(call)                                     \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;  // optional: only mk_comm_size asks for it
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = "";
};

std::mutex g_mu;
Rccl g_rccl;
bool g_rccl_tried = false;

// binds librccl once; returns nullptr (with g_rccl.why set) if it is not usable
Rccl *rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl_tried) return g_rccl.lib ? &g_rccl : nullptr;
    g_rccl_tried = true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 not found: %s", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(h, n);
        if (!p) {
            ok = false;
            snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks %s", n);
        }
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) return nullptr;
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.lib = h;
    return &g_rccl;
}

int rccl_fail(Rccl *R, ncclResult_t r, const char *what) {
    return fail(MK_E_RCCL, "%s failed: %s", what, R && R->GetErrorString ? R->GetErrorString(r) : "?");
}

#define MK_NCCL(R, call)                                        \
    do {                                                        \
        ncclResult_t r_ = (call);                               \
        if (r_ != ncclSuccess) return rccl_fail(R, r_, #call);  \
    } while (0)

// single-process communicators, one set per distinct device list, kept for the life of the process
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comm_sets;

}  // namespace

static_assert(MK_COMM_ID_BYTES == sizeof(ncclUniqueId), "id size");

extern "C" {

// Setting up RCCL's communicators takes seconds (ncclCommInitAll: 5.7 s for a 2-handle job on the round's box) -- as long as a whole
// 6 GB extract job: a host that knows it will reduce at the end starts this on a thread of its own at the beginning, beside the job.
int mk_reduce_prepare(mk_matcher *const *per_gpu, int n) {
    if (!per_gpu || n <= 0) return fail(MK_E_INVALID_ARG, "null argument");
    MK_ABI_BEGIN
    Rccl *R = rccl();
    if (!R) return fail(MK_E_RCCL, "%s", g_rccl.why);
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) {
        if (!per_gpu[i]) return fail(MK_E_INVALID_ARG, "null handle (%d)", i);
        if (std::find(devs.begin(), devs.end(), per_gpu[i]->device) == devs.end()) devs.push_back(per_gpu[i]->device);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm_sets.count(devs)) return MK_OK;
    std::vector<ncclComm_t> c(devs.size(), nullptr);
    MK_NCCL(R, R->CommInitAll(c.data(), (int)devs.size(), devs.data()));
    g_comm_sets.emplace(devs, std::move(c));
    return MK_OK;
    MK_ABI_END
}

int mk_reduce_counters(mk_matcher *const *per_gpu, int n, void *const *d_counters, size_t len, uint64_t *host_sum) {
    if (!per_gpu || !d_counters || n <= 0) return fail(MK_E_INVALID_ARG, "null argument");
    for (int i = 0; i < n; ++i)
        if (!per_gpu[i] || !d_counters[i]) return fail(MK_E_INVALID_ARG, "null handle or counter vector (%d)", i);
    if (len == 0) return MK_OK;
    MK_ABI_BEGIN
    Rccl *R = rccl();
    if (!R) return fail(MK_E_RCCL, "%s", g_rccl.why);
    // everything enqueued on the handles' devices so far must be visible to the reduction
    std::vector<int> devs;  // distinct devices, in order of first appearance
    std::vector<int> leader(n);  // index (into per_gpu) of the first handle on the same device
    for (int i = 0; i < n; ++i) {
        const int d = per_gpu[i]->device;
        auto it = std::find(devs.begin(), devs.end(), d);
        if (it == devs.end()) {
            devs.push_back(d);
            leader[i] = i;
            MK_HIP_R(hipSetDevice(d));
            MK_HIP_R(hipDeviceSynchronize());
        } else {
            for (int j = 0; j < i; ++j)
                if (per_gpu[j]->device == d) {
                    leader[i] = j;
                    break;
                }
        }
    }
    // handles that share a device are summed on that device first (RCCL wants one rank per GPU)
    for (int i = 0; i < n; ++i) {
        if (leader[i] == i) continue;
        mk_matcher *L = per_gpu[leader[i]];
        MK_HIP_R(hipSetDevice(L->device));
        launch_add_u64((unsigned long long *)d_counters[leader[i]], (const unsigned long long *)d_counters[i], len, L->stream);
        MK_HIP_R(hipGetLastError());
    }
    // The communicator set of this device list is created once and kept -- but only while it works: a set whose
    // collective failed is destroyed and forgotten, so that the failure is total (nothing half-initialised stays
    // cached) and the next call starts from ncclCommInitAll again.  A failing ncclCommInitAll caches nothing either:
    // RCCL creates all ranks of the list or none.  One reduction at a time per process (g_mu).
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_comm_sets.find(devs);
    if (it == g_comm_sets.end()) {
        std::vector<ncclComm_t> c(devs.size(), nullptr);
        MK_NCCL(R, R->CommInitAll(c.data(), (int)devs.size(), devs.data()));
        it = g_comm_sets.emplace(devs, std::move(c)).first;
    }
    std::vector<ncclComm_t> *comms = &it->second;
    auto drop_set = [&]() {
        for (ncclComm_t c : *comms)
            if (c) (void)R->CommDestroy(c);
        g_comm_sets.erase(it);
    };
    ncclResult_t r = R->GroupStart();
    if (r != ncclSuccess) {
        drop_set();
        return rccl_fail(R, r, "ncclGroupStart");
    }
    for (size_t k = 0; k < devs.size(); ++k) {
        int li = 0;
        for (int i = 0; i < n; ++i)
            if (leader[i] == i && per_gpu[i]->device == devs[k]) li = i;
        hipError_t e = hipSetDevice(devs[k]);
        if (e == hipSuccess)
            r = R->AllReduce(d_counters[li], d_counters[li], len, ncclUint64, ncclSum, (*comms)[k], per_gpu[li]->stream);
        if (e != hipSuccess || r != ncclSuccess) {
            (void)R->GroupEnd();  // closes the group; whatever it launched belongs to communicators that are dropped now
            drop_set();
            return e != hipSuccess ? hip_fail(e, "hipSetDevice") : rccl_fail(R, r, "ncclAllReduce");
        }
    }
    r = R->GroupEnd();
    if (r != ncclSuccess) {
        drop_set();
        return rccl_fail(R, r, "ncclGroupEnd");
    }
    // every vector holds the sum on return
    for (int i = 0; i < n; ++i) {
        if (leader[i] == i) continue;
        mk_matcher *L = per_gpu[leader[i]];
        MK_HIP_R(hipSetDevice(L->device));
        MK_HIP_R(hipMemcpyAsync(d_counters[i], d_counters[leader[i]], len * sizeof(uint64_t), hipMemcpyDeviceToDevice, L->stream));
    }
    for (int i = 0; i < n; ++i) {
        if (leader[i] != i) continue;
        MK_HIP_R(hipSetDevice(per_gpu[i]->device));
        MK_HIP_R(hipStreamSynchronize(per_gpu[i]->stream));
    }
    if (host_sum) {
        MK_HIP_R(hipSetDevice(per_gpu[0]->device));
        MK_HIP_R(hipMemcpy(host_sum, d_counters[0], len * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return MK_OK;
    MK_ABI_END
}

int mk_comm_available(void) {
    if (!rccl()) return fail(MK_E_RCCL, "%s", g_rccl.why);
    return MK_OK;
}

int mk_comm_unique_id(uint8_t id[MK_COMM_ID_BYTES]) {
    if (!id) return fail(MK_E_INVALID_ARG, "null id");
    Rccl *R = rccl();
    if (!R) return fail(MK_E_RCCL, "%s", g_rccl.why);
    ncclUniqueId u;
    MK_NCCL(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, MK_COMM_ID_BYTES);
    return MK_OK;
}

int mk_comm_init(mk_matcher *m, const uint8_t id[MK_COMM_ID_BYTES], int rank, int n_ranks) {
    if (!m || !id) return fail(MK_E_INVALID_ARG, "null argument");
    if (n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(MK_E_INVALID_ARG, "rank %d of %d", rank, n_ranks);
    if (m->comm) return fail(MK_E_INVALID_ARG, "matcher already has a communicator");
    Rccl *R = rccl();
    if (!R) return fail(MK_E_RCCL, "%s", g_rccl.why);
    ncclUniqueId u;
    memcpy(u.internal, id, MK_COMM_ID_BYTES);
    MK_HIP_R(hipSetDevice(m->device));
    ncclComm_t c = nullptr;
    MK_NCCL(R, R->CommInitRank(&c, n_ranks, u, rank));
    m->comm = c;
    m->comm_rank = rank;
    m->comm_size = n_ranks;
    return MK_OK;
}

int mk_comm_reduce_counters(mk_matcher *m, void *d_counters, size_t len, void *stream) {
    if (!m || (!d_counters && len)) return fail(MK_E_INVALID_ARG, "null argument");
    if (!m->comm) return fail(MK_E_INVALID_ARG, "mk_comm_init has not been called on this matcher");
    if (len == 0) return MK_OK;
    Rccl *R = rccl();
    if (!R) return fail(MK_E_RCCL, "%s", g_rccl.why);
    MK_HIP_R(hipSetDevice(m->device));
    MK_NCCL(R, R->AllReduce(d_counters, d_counters, len, ncclUint64, ncclSum, (ncclComm_t)m->comm, (hipStream_t)stream));
    return MK_OK;
}

int mk_comm_size(const mk_matcher *m, int *n_ranks) {
    if (!m || !n_ranks) return fail(MK_E_INVALID_ARG, "null argument");
    if (!m->comm) return fail(MK_E_INVALID_ARG, "mk_comm_init has not been called on this matcher");
    *n_ranks = m->comm_size;
    Rccl *R = rccl();
    if (R && R->CommCount) MK_NCCL(R, R->CommCount((ncclComm_t)m->comm, n_ranks));  // what RCCL itself says
    return MK_OK;
}

int mk_comm_destroy(mk_matcher *m) {
    if (!m) return fail(MK_E_INVALID_ARG, "null matcher");
    if (!m->comm) return MK_OK;
    Rccl *R = rccl();
    if (R) {
        (void)hipSetDevice(m->device);
        (void)R->CommDestroy((ncclComm_t)m->comm);
    }
    m->comm = nullptr;
    return MK_OK;
}

}  // extern "C"
