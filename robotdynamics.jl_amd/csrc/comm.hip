// comm.hip -- the one exchange the multi-GPU layout of this library has: stacking per-GPU result shards (Jacobians, error-state Jacobians)
// on one GPU or on all of them, over RCCL / xGMI, from plain C (include/rdhip.h, "several GPUs").
//
// Knot points are independent, so evaluation never communicates (contiguous blocks per GPU, rdhip_shard_range); only a caller that needs the
// STACKED result in one place pays this step.  The reference has no counterpart (its trajectory-level loops are comments,
// src/trajectories.jl:425-434): this is the `for k` loop of its callers cut over GPUs plus the concatenation of the pieces.
//
// Communication pattern (8 x MI355X, fully connected xGMI, 7 links per GPU):
//   ROOT: every non-root rank ncclSend()s its shard, the root ncclRecv()s each straight into its place in the stack, all inside ONE
//         ncclGroup -- RCCL has no gather primitive, and the grouped point-to-point form puts every shard on its own link into the root
//         (a ring would push all shards through single links, 7 steps);
//   ALL:  equal shards: ncclAllGather (RCCL picks the topology's algorithm); ragged shards: the grouped send / recv pattern between all pairs.
// librccl is loaded with dlopen() when the first communicator is created: librdhip itself has no link-time dependency on it, single-GPU
// hosts never load it.
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
// The library is dlopen()ed at run time; the header is only needed for the handful of types and prototypes below.  A ROCm install without the RCCL
// development headers still builds librdhip (the declarations follow RCCL's stable C ABI, nccl.h 2.x); -DRDHIP_NO_RCCL_HEADER forces this branch.
#if !defined(RDHIP_NO_RCCL_HEADER) && defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define RDHIP_HAVE_RCCL_HEADER 1
#endif
#endif
#ifndef RDHIP_HAVE_RCCL_HEADER
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;              // (other values are only ever handed back to ncclGetErrorString)
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include "../../include/rdhip.h"

// api.hip
extern "C" void* rdhip_ctx_stream(rdhip_ctx* ctx);
int rdhip_ctx_device_(const rdhip_ctx* ctx);
void rdhip_ctx_set_error_(rdhip_ctx* ctx, const char* msg);

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // optional symbol of a recording stand-in (tests/rccl_stub.c): the local copy of a rank's own shard, reported instead of executed when the
    // context has no device (RDHIP_DRY_RUN)
    void (*StubLocalCopy)(void* dst, const void* src, unsigned long long bytes) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static std::mutex mu;
    static Rccl R;
    static bool tried = false;
    std::lock_guard<std::mutex> g(mu);
    if (!tried) {
        tried = true;
        const char* names[] = {getenv("RDHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (R.so) break;
        }
        if (!R.so) { const char* e = dlerror(); R.err = std::string("librccl could not be loaded (") + (e ? e : "?") + ")"; return &R; }
#define RD_SYM(field, name) R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.so, name)); if (!R.field) { R.err = std::string("librccl lacks ") + name; R.so = nullptr; return &R; }
        RD_SYM(GetUniqueId, "ncclGetUniqueId") RD_SYM(CommInitRank, "ncclCommInitRank") RD_SYM(CommInitAll, "ncclCommInitAll")
        RD_SYM(CommDestroy, "ncclCommDestroy") RD_SYM(GroupStart, "ncclGroupStart") RD_SYM(GroupEnd, "ncclGroupEnd") RD_SYM(Send, "ncclSend")
        RD_SYM(Recv, "ncclRecv") RD_SYM(AllGather, "ncclAllGather") RD_SYM(GetErrorString, "ncclGetErrorString")
#undef RD_SYM
        R.StubLocalCopy = reinterpret_cast<decltype(R.StubLocalCopy)>(dlsym(R.so, "rdhipStubLocalCopy"));
    }
    return &R;
}

thread_local std::string g_comm_err;
int cfail(rdhip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_comm_err = buf;
    rdhip_ctx_set_error_(ctx, buf);            // ctx == NULL: readable through rdhip_last_error(NULL) on this thread
    return code;
}

}  // namespace

struct rdhip_comm {
    int nranks = 0;                     // ranks of the communicator (one per GPU)
    int nlocal = 0;                     // ranks this process drives: all of them (rdhip_comm_create) or one (rdhip_comm_create_rank)
    std::vector<int> rank;              // global rank of local slot l
    std::vector<rdhip_ctx*> ctx;        // its context (device + stream)
    std::vector<ncclComm_t> nccl;       // its RCCL communicator; empty when nranks == 1 (nothing to talk to: librccl is not even loaded)
};

// RDHIP_COMM_FORCE_RCCL=1: a 1-rank communicator goes through librccl as well (ncclCommInit*, a 1-rank ncclAllGather, a grouped ncclSend / ncclRecv to
// itself) instead of degenerating to a local copy: the only way to execute the RCCL calls of this file on a 1-GPU test box
static bool force_rccl() { const char* e = getenv("RDHIP_COMM_FORCE_RCCL"); return e && e[0] == '1'; }

extern "C" {

int rdhip_comm_unique_id(char id[RDHIP_COMM_ID_BYTES]) {
    if (!id) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_comm_unique_id: id is NULL");
    static_assert(RDHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as an opaque 128-byte blob");
    Rccl* R = rccl();
    if (!R->so) return cfail(nullptr, RDHIP_ERR_UNSUPPORTED, "rdhip_comm_unique_id: %s", R->err.c_str());
    ncclUniqueId u;
    const ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return cfail(nullptr, RDHIP_ERR_HIP, "ncclGetUniqueId: %s", R->GetErrorString(r));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return RDHIP_OK;
}

int rdhip_comm_create(rdhip_ctx* const* ctxs, int n, rdhip_comm** out) {
    if (!out) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create: out is NULL");
    *out = nullptr;
    if (!ctxs || n <= 0) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create: no contexts");
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create: context %d is NULL", i);
        devs[i] = rdhip_ctx_device_(ctxs[i]);
        for (int k = 0; k < i; ++k) if (devs[k] == devs[i]) return cfail(ctxs[i], RDHIP_ERR_INVALID_ARG, "rdhip_comm_create: contexts %d and %d sit on the same device %d (one rank per GPU)", k, i, devs[i]);
    }
    rdhip_comm* C = new (std::nothrow) rdhip_comm();
    if (!C) return cfail(nullptr, RDHIP_ERR_HIP, "out of host memory");
    C->nranks = C->nlocal = n;
    C->rank.resize(n); C->ctx.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) C->rank[i] = i;
    if (n > 1 || force_rccl()) {
        Rccl* R = rccl();
        if (!R->so) { delete C; return cfail(ctxs[0], RDHIP_ERR_UNSUPPORTED, "rdhip_comm_create: %s", R->err.c_str()); }
        C->nccl.resize(n);
        const ncclResult_t r = R->CommInitAll(C->nccl.data(), n, devs.data());
        if (r != ncclSuccess) { delete C; return cfail(ctxs[0], RDHIP_ERR_HIP, "ncclCommInitAll over %d devices: %s", n, R->GetErrorString(r)); }
    }
    *out = C;
    return RDHIP_OK;
}

int rdhip_comm_create_rank(rdhip_ctx* ctx, int nranks, int rank, const char id[RDHIP_COMM_ID_BYTES], rdhip_comm** out) {
    if (!out) return cfail(ctx, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create_rank: out is NULL");
    *out = nullptr;
    if (!ctx || nranks <= 0 || rank < 0 || rank >= nranks) return cfail(ctx, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create_rank: bad arguments (nranks %d, rank %d)", nranks, rank);
    const bool with_rccl = nranks > 1 || (force_rccl() && id);
    if (nranks > 1 && !id) return cfail(ctx, RDHIP_ERR_INVALID_ARG, "rdhip_comm_create_rank: id is NULL");
    rdhip_comm* C = new (std::nothrow) rdhip_comm();
    if (!C) return cfail(ctx, RDHIP_ERR_HIP, "out of host memory");
    C->nranks = nranks; C->nlocal = 1; C->rank.assign(1, rank); C->ctx.assign(1, ctx);
    if (with_rccl) {
        Rccl* R = rccl();
        if (!R->so) { delete C; return cfail(ctx, RDHIP_ERR_UNSUPPORTED, "rdhip_comm_create_rank: %s", R->err.c_str()); }
        if (rdhip_ctx_device_(ctx) >= 0 && hipSetDevice(rdhip_ctx_device_(ctx)) != hipSuccess) { delete C; return cfail(ctx, RDHIP_ERR_HIP, "hipSetDevice failed"); }
        ncclUniqueId u; std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        C->nccl.resize(1);
        const ncclResult_t r = R->CommInitRank(&C->nccl[0], nranks, u, rank);
        if (r != ncclSuccess) { delete C; return cfail(ctx, RDHIP_ERR_HIP, "ncclCommInitRank (rank %d of %d): %s", rank, nranks, R->GetErrorString(r)); }
    }
    *out = C;
    return RDHIP_OK;
}

int rdhip_comm_destroy(rdhip_comm* C) {
    if (!C) return RDHIP_OK;
    if (!C->nccl.empty()) { Rccl* R = rccl(); for (ncclComm_t c : C->nccl) if (c) (void)R->CommDestroy(c); }
    delete C;
    return RDHIP_OK;
}

int rdhip_comm_size(const rdhip_comm* C, int* nranks, int* nlocal) {
    if (!C) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "comm is NULL");
    if (nranks) *nranks = C->nranks;
    if (nlocal) *nlocal = C->nlocal;
    return RDHIP_OK;
}

int rdhip_comm_rank(const rdhip_comm* C, int local, int* rank, rdhip_ctx** ctx) {
    if (!C || local < 0 || local >= C->nlocal) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_comm_rank: bad arguments");
    if (rank) *rank = C->rank[local];
    if (ctx) *ctx = C->ctx[local];
    return RDHIP_OK;
}

int rdhip_gather(rdhip_comm* C, int mode, int root, const void* const* shard, const unsigned long long* bytes, void* const* out) {
    if (!C) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_gather: comm is NULL");
    rdhip_ctx* c0 = C->ctx[0];
    if (mode == RDHIP_GATHER_NONE) return RDHIP_OK;                                   // results stay sharded for a sharded consumer
    if (mode != RDHIP_GATHER_ROOT && mode != RDHIP_GATHER_ALL) return cfail(c0, RDHIP_ERR_INVALID_ARG, "rdhip_gather: bad mode %d", mode);
    if (mode == RDHIP_GATHER_ROOT && (root < 0 || root >= C->nranks)) return cfail(c0, RDHIP_ERR_INVALID_ARG, "rdhip_gather: root %d out of range [0,%d)", root, C->nranks);
    if (!shard || !bytes || !out) return cfail(c0, RDHIP_ERR_INVALID_ARG, "rdhip_gather: NULL table");
    const int P = C->nranks;
    const bool use_rccl = !C->nccl.empty();
    std::vector<unsigned long long> off(P + 1, 0);
    for (int r = 0; r < P; ++r) off[r + 1] = off[r] + bytes[r];
    bool equal = true;
    for (int r = 1; r < P; ++r) equal = equal && bytes[r] == bytes[0];
    for (int l = 0; l < C->nlocal; ++l) {
        const int r = C->rank[l];
        const bool receives = mode == RDHIP_GATHER_ALL || r == root;
        if (bytes[r] > 0 && !shard[l]) return cfail(C->ctx[l], RDHIP_ERR_INVALID_ARG, "rdhip_gather: shard of rank %d is NULL", r);
        if (receives && off[P] > 0 && !out[l]) return cfail(C->ctx[l], RDHIP_ERR_INVALID_ARG, "rdhip_gather: output of rank %d is NULL", r);
    }
    // a rank's own shard: a device-to-device copy on its stream
    for (int l = 0; l < C->nlocal; ++l) {
        const int r = C->rank[l];
        const bool receives = mode == RDHIP_GATHER_ALL || r == root;
        if (!receives || bytes[r] == 0) continue;
        if (use_rccl && mode == RDHIP_GATHER_ALL && equal) continue;                    // ncclAllGather places it
        if (use_rccl && P == 1) continue;                                               // (forced 1-rank RCCL: sent to itself below)
        char* dst = static_cast<char*>(out[l]) + off[r];
        if (dst == shard[l]) continue;                                                  // computed in place inside the stack
        if (rdhip_ctx_device_(C->ctx[l]) < 0) {                                         // dry context: the stand-in library records the copy
            Rccl* Rs = rccl();
            if (!Rs->so || !Rs->StubLocalCopy) return cfail(C->ctx[l], RDHIP_ERR_NO_DEVICE, "rdhip_gather: this context has no device (RDHIP_DRY_RUN) and RDHIP_RCCL_LIB is not the recording stand-in");
            Rs->StubLocalCopy(dst, shard[l], bytes[r]);
            continue;
        }
        if (hipSetDevice(rdhip_ctx_device_(C->ctx[l])) != hipSuccess ||
            hipMemcpyAsync(dst, shard[l], bytes[r], hipMemcpyDeviceToDevice, (hipStream_t)rdhip_ctx_stream(C->ctx[l])) != hipSuccess)
            return cfail(C->ctx[l], RDHIP_ERR_HIP, "rdhip_gather: local copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (!use_rccl) return RDHIP_OK;
    Rccl* R = rccl();
    ncclResult_t rc = R->GroupStart();
    if (rc != ncclSuccess) return cfail(c0, RDHIP_ERR_HIP, "ncclGroupStart: %s", R->GetErrorString(rc));
    for (int l = 0; l < C->nlocal && rc == ncclSuccess; ++l) {
        const int r = C->rank[l];
        hipStream_t s = (hipStream_t)rdhip_ctx_stream(C->ctx[l]);
        ncclComm_t comm = C->nccl[l];
        if (mode == RDHIP_GATHER_ALL && equal) {
            if (bytes[0] > 0) rc = R->AllGather(shard[l], out[l], (size_t)bytes[0], ncclChar, comm, s);
            continue;
        }
        if (P == 1) {                                                                   // forced 1-rank RCCL: the shard travels rank 0 -> rank 0
            if (bytes[0] > 0) { rc = R->Send(shard[l], (size_t)bytes[0], ncclChar, 0, comm, s); if (rc == ncclSuccess) rc = R->Recv(out[l], (size_t)bytes[0], ncclChar, 0, comm, s); }
            continue;
        }
        for (int p = 0; p < P && rc == ncclSuccess; ++p) {
            if (p == r) continue;
            const bool p_receives = mode == RDHIP_GATHER_ALL || p == root;
            const bool r_receives = mode == RDHIP_GATHER_ALL || r == root;
            if (p_receives && bytes[r] > 0) rc = R->Send(shard[l], (size_t)bytes[r], ncclChar, p, comm, s);
            if (rc == ncclSuccess && r_receives && bytes[p] > 0) rc = R->Recv(static_cast<char*>(out[l]) + off[p], (size_t)bytes[p], ncclChar, p, comm, s);
        }
    }
    const ncclResult_t re = R->GroupEnd();
    if (rc != ncclSuccess) return cfail(c0, RDHIP_ERR_HIP, "rdhip_gather: RCCL call failed: %s", R->GetErrorString(rc));
    if (re != ncclSuccess) return cfail(c0, RDHIP_ERR_HIP, "ncclGroupEnd: %s", R->GetErrorString(re));
    return RDHIP_OK;
}

// jacobian! on the local ranks' blocks (asynchronous, one stream per rank), then the exchange
int rdhip_jacobian_sharded_gather(rdhip_comm* C, const rdhip_model* model, int rule, int diff, int dtype, long long N,
                                  const void* const* Z, const double* const* t, const double* const* dt, void* const* J, void* const* Y,
                                  int gather, int root, void* const* Jstack) {
    if (!C) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded_gather: comm is NULL");
    if (!model || !Z || !t || !dt || !J) return cfail(C->ctx[0], RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded_gather: NULL argument");
    int n = 0, m = 0;
    int rc = rdhip_model_dims(model, &n, &m, nullptr, nullptr); if (rc) return rc;
    const unsigned long long ts = dtype == RDHIP_F64 ? 8 : 4;
    std::vector<unsigned long long> bytes(C->nranks);
    for (int r = 0; r < C->nranks; ++r) { long long lo, cnt; rdhip_shard_range(N, C->nranks, r, &lo, &cnt); bytes[r] = (unsigned long long)cnt * n * (n + m) * ts; }
    for (int l = 0; l < C->nlocal; ++l) {
        long long lo, cnt; rdhip_shard_range(N, C->nranks, C->rank[l], &lo, &cnt);
        rc = rdhip_jacobian(C->ctx[l], model, rule, diff, dtype, cnt, Z[l], t[l], dt[l], J[l], Y ? Y[l] : nullptr, RDHIP_DEVICE);
        if (rc) { rdhip_ctx_set_error_(nullptr, rdhip_last_error(C->ctx[l])); return rc; }
    }
    if (gather == RDHIP_GATHER_NONE) return RDHIP_OK;
    if (!Jstack) return cfail(C->ctx[0], RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded_gather: Jstack is NULL");
    return rdhip_gather(C, gather, root, J, bytes.data(), Jstack);
}

int rdhip_errstate_jacobian_sharded_gather(rdhip_comm* C, const rdhip_model* model, int dtype, long long N, const void* const* X, int ldx, int clear,
                                           void* const* G, int gather, int root, void* const* Gstack) {
    if (!C) return cfail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_errstate_jacobian_sharded_gather: comm is NULL");
    if (!model || !X || !G) return cfail(C->ctx[0], RDHIP_ERR_INVALID_ARG, "rdhip_errstate_jacobian_sharded_gather: NULL argument");
    int n = 0, e = 0;
    int rc = rdhip_model_dims(model, &n, nullptr, nullptr, &e); if (rc) return rc;
    const unsigned long long ts = dtype == RDHIP_F64 ? 8 : 4;
    std::vector<unsigned long long> bytes(C->nranks);
    for (int r = 0; r < C->nranks; ++r) { long long lo, cnt; rdhip_shard_range(N, C->nranks, r, &lo, &cnt); bytes[r] = (unsigned long long)cnt * n * e * ts; }
    for (int l = 0; l < C->nlocal; ++l) {
        long long lo, cnt; rdhip_shard_range(N, C->nranks, C->rank[l], &lo, &cnt);
        rc = rdhip_errstate_jacobian(C->ctx[l], model, dtype, cnt, X[l], ldx, clear, G[l], RDHIP_DEVICE);
        if (rc) { rdhip_ctx_set_error_(nullptr, rdhip_last_error(C->ctx[l])); return rc; }
    }
    if (gather == RDHIP_GATHER_NONE) return RDHIP_OK;
    if (!Gstack) return cfail(C->ctx[0], RDHIP_ERR_INVALID_ARG, "rdhip_errstate_jacobian_sharded_gather: Gstack is NULL");
    return rdhip_gather(C, gather, root, G, bytes.data(), Gstack);
}

}  // extern "C"
