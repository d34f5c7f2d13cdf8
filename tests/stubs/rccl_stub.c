/* rccl_stub.c -- a RECORDING stand-in for librccl (test infrastructure; built by tests/test_host_abi.py with gcc, loaded through RDHIP_RCCL_LIB).
 * It moves no data: every entry point the library's exchange uses (csrc/comm.hip) appends one line to the file named by RDHIP_STUB_LOG, so a
 * CPU-only test can check WHICH sends / receives / all-gathers each rank of a communicator issues -- peers, byte counts, offsets into the stack.
 * rdhipStubLocalCopy is the hook comm.hip calls for a rank's own shard when the context has no device (RDHIP_DRY_RUN). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int nranks, rank; } stub_comm;
typedef struct { char internal[128]; } ncclUniqueId;

static void logf_(const char* fmt, long a, long b, long c, long long d, const void* p, const void* q) {
    const char* path = getenv("RDHIP_STUB_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, fmt, a, b, c, d, (unsigned long long)(size_t)p, (unsigned long long)(size_t)q);
    fclose(f);
}

int ncclGetUniqueId(ncclUniqueId* id) { for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 7 + 1); return 0; }
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    stub_comm* c = (stub_comm*)malloc(sizeof *c); c->nranks = nranks; c->rank = rank; *comm = c;
    logf_("init rank=%ld nranks=%ld idok=%ld %lld %llu %llu\n", rank, nranks, id.internal[1] == (char)8, 0, 0, 0);
    return 0;
}
int ncclCommInitAll(void** comms, int ndev, const int* devs) {
    for (int i = 0; i < ndev; ++i) { stub_comm* c = (stub_comm*)malloc(sizeof *c); c->nranks = ndev; c->rank = i; comms[i] = c; }
    logf_("initall rank=%ld nranks=%ld dev0=%ld %lld %llu %llu\n", -1, ndev, devs ? devs[0] : 0, 0, 0, 0);
    return 0;
}
int ncclCommDestroy(void* comm) { free(comm); return 0; }
int ncclGroupStart(void) { logf_("groupstart %ld %ld %ld %lld %llu %llu\n", 0, 0, 0, 0, 0, 0); return 0; }
int ncclGroupEnd(void) { logf_("groupend %ld %ld %ld %lld %llu %llu\n", 0, 0, 0, 0, 0, 0); return 0; }
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
    (void)stream; logf_("send rank=%ld peer=%ld dtype=%ld bytes=%lld buf=%llu %llu\n", ((stub_comm*)comm)->rank, peer, dtype, (long long)count, buf, 0); return 0;
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream) {
    (void)stream; logf_("recv rank=%ld peer=%ld dtype=%ld bytes=%lld buf=%llu %llu\n", ((stub_comm*)comm)->rank, peer, dtype, (long long)count, buf, 0); return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
    (void)stream; logf_("allgather rank=%ld nranks=%ld dtype=%ld bytes=%lld send=%llu recv=%llu\n", ((stub_comm*)comm)->rank, ((stub_comm*)comm)->nranks, dtype, (long long)count, send, recv); return 0;
}
const char* ncclGetErrorString(int r) { (void)r; return "stub"; }
void rdhipStubLocalCopy(void* dst, const void* src, unsigned long long bytes) {
    memcpy(dst, src, (size_t)bytes);
    logf_("localcopy %ld %ld %ld bytes=%lld dst=%llu src=%llu\n", 0, 0, 0, (long long)bytes, dst, src);
}
