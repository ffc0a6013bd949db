/* fake_rccl.c -- a stand-in for librccl.so.1 with failure injection, for the unit tests of reduce.cpp's failure
 * modes (tests/test_gpu_multirank.py).  Test infrastructure only: built by the test into a temporary directory that
 * is put in front of LD_LIBRARY_PATH of a child process; libmerkurio_hip.so binds librccl by soname at run time, so
 * the child's mk_reduce_counters / mk_comm_* calls land here.  One rank per communicator only: an all-reduce over one
 * rank leaves the (in-place) vector as it is.
 *
 * FAKE_RCCL_FAIL (read at every call): "initall" -> ncclCommInitAll fails after having written the first comm of the
 * list (a partial initialisation, which the caller must not use or cache); "allreduce" -> ncclAllReduce fails;
 * "groupend" -> ncclGroupEnd fails; "initrank" -> ncclCommInitRank fails.  fake_rccl_live_comms() = communicators
 * created and not destroyed; fake_rccl_calls(i) = number of calls of function i (0 InitAll, 1 AllReduce, 2 Destroy). */
#include <stdlib.h>
#include <string.h>

typedef struct fake_comm { int n, rank; } *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;

static int live = 0, calls[4] = {0, 0, 0, 0};
static int failing(const char *what) {
    const char *f = getenv("FAKE_RCCL_FAIL");
    return f && strcmp(f, what) == 0;
}
static ncclComm_t make(int n, int rank) {
    ncclComm_t c = (ncclComm_t)malloc(sizeof(*c));
    c->n = n;
    c->rank = rank;
    ++live;
    return c;
}
int fake_rccl_live_comms(void) { return live; }
int fake_rccl_calls(int i) { return calls[i & 3]; }

int ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof(*id)); return 0; }
int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    if (failing("initrank") || nranks != 1) return 2;
    *comm = make(nranks, rank);
    return 0;
}
int ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    (void)devlist;
    ++calls[0];
    if (failing("initall")) {
        comm[0] = (ncclComm_t)0x1;  /* garbage the caller must never touch */
        return 2;
    }
    for (int i = 0; i < ndev; ++i) comm[i] = make(ndev, i);
    return 0;
}
int ncclCommDestroy(ncclComm_t c) {
    ++calls[2];
    free(c);
    --live;
    return 0;
}
int ncclCommCount(const ncclComm_t c, int *n) { *n = c->n; return 0; }
int ncclAllReduce(const void *s, void *r, size_t count, int dt, int op, ncclComm_t c, void *stream) {
    (void)count; (void)dt; (void)op; (void)stream;
    ++calls[1];
    if (failing("allreduce") || s != r || c->n != 1) return 5;
    return 0;
}
int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return failing("groupend") ? 3 : 0; }
const char *ncclGetErrorString(int r) { return r == 2 ? "fake: unhandled system error" : r == 5 ? "fake: invalid usage" : r == 3 ? "fake: internal error" : "fake: ?"; }
