// api.hip -- the C ABI (include/rdhip.h): contexts, the model registry, integer queries, argument checking,
// host<->device staging for RDHIP_HOST calls, and dispatch into the per-family kernel launchers.
// No compute happens on the CPU here; without a GPU every compute entry point returns RDHIP_ERR_NO_DEVICE.
#include <cmath>
#include <cxxabi.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "jit_api.hpp"

using namespace rd;

struct rdhip_ctx {
    int device;
    unsigned long long serial = 0;        // unique per context of this process (never reused): identifies the issuer of the device's last fused launch
    hipStream_t stream;
    bool own_stream;
    hipEvent_t ev0, ev1;
    bool capturing = false;               // between rdhip_graph_begin_capture and rdhip_graph_end_capture
    int epoch = 0;                        // implicit calls on this context so far (1 .. 2^22, never 0; `epoch << 8` must stay a positive int): LaunchArgs::epoch
    int* d_flag = nullptr;                // device words "a row exchange was needed" of the implicit kernels (allocated with the context: never inside a capture)
    std::string err;
    // grow-only device workspace slots for RDHIP_HOST staging
    void* ws[8];
    size_t ws_bytes[8];
    void* mws = nullptr; size_t mws_bytes = 0;        // rdhip_jacobian_multi's gather / scatter staging (grow-only; see multi_staged)
    // batch tables of rdhip_jacobian_multi calls with more batches than fit the kernel arguments: pinned host image + device copy, a ring of
    // slots so that a call never overwrites a table a kernel of an earlier call may still be reading (the slot's event says when it is free)
    // A solver re-issues the same call on the same arrays every iteration: a slot whose image equals the new table is reused without a copy.
    struct TableSlot { void* host = nullptr; void* dev = nullptr; size_t cap = 0, used = 0; hipEvent_t done = nullptr; bool pending = false, unrecorded = false, hit = false; };
    static constexpr int NTSLOT = 8;
    TableSlot tslot[NTSLOT];
    unsigned tnext = 0;
    int last_multi_launches = 0;          // kernel launches the last rdhip_jacobian_multi call on this context issued (rdhip_ctx_last_multi_launches)
    // fused rollout -> Jacobians (fused.hpp): one 64-bit progress word per 64-trajectory group, zero at allocation; a call's words carry its epoch in
    // the upper half, so nothing is cleared between calls
    unsigned long long* d_progress = nullptr;
    unsigned long long roll_epoch = 0;
    // dynamic tile schedule of the split-role Jacobian kernels (kernels.hpp jac_split_loop): a launch gets a slot of counter words {next, done, eight per-XCD-range counters}, all zero
    // between launches (the launch's last workgroup resets them).  One ring of words per stream of the context: launches of one stream run in order,
    // so a slot comes round again only behind the launch that used it.
    static constexpr int NSCHED = 32, SCHED_WORDS = 16;      // (a slot: next, done, eight per-XCD-range counters; 64 bytes)
    unsigned* d_sched = nullptr;          // 2 streams x NSCHED slots x SCHED_WORDS words, allocated on first use (never inside a capture)
    unsigned sched_next[2] = {0, 0};
    int last_rolljac_launches = 0;        // kernel launches of the last rdhip_rollout_jacobian call (1 = fused)
    bool allow_fused = true;              // rdhip_rollout_jacobian_multi lets ONE model of a call take the fused (spin-waiting) launch
    bool fused_beside = false;            // ... and says that the other models' launches run BESIDE it: the fused launch must not fill every workgroup slot of the chip (LaunchArgs::beside)
    // Launch overlap (rdhip_ctx_set_overlap): consecutive rdhip_jacobian calls on device arrays alternate between the context's stream and one
    // internal stream, with an event edge only where a call touches memory a call still in flight on the OTHER stream touches; everything else
    // joins first (side_join).  A launch's prologue then runs under its predecessor's tail, which one stream cannot do.
    bool overlap = false;
    hipStream_t side = nullptr;           // the internal stream
    hipEvent_t side_done = nullptr, main_mark = nullptr, main_done = nullptr;
    bool side_pending = false;            // work on `side` that the context's stream has not been ordered behind yet
    bool in_overlapped_call = false;      // rdhip_jacobian's overlap path orders itself (check_common must not join)
    unsigned flip = 0;                    // calls issued in overlap mode (parity = stream)
    struct Range { const char* lo; const char* hi; };
    struct Footprint { Range r[5]; int n = 0; };      // Z, t, dt (read), J, Y (written) of a call
    std::vector<Footprint> pend[2];       // calls issued on stream 0 (the context's) / 1 (internal) that work on the OTHER stream has not been ordered behind
    // rdhip_rollout_jacobian_multi: one child context (own stream, own progress words) per additional model, created on first use
    rdhip_ctx* child[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
    // ... in overlap mode a two-model call runs its second model on the INTERNAL stream (a child context that borrows it) and neither forks nor joins
    // when it repeats the previous call argument for argument: `multi_pending` = work on the internal stream that NO later operation may overtake
    // (side_join orders the context's stream behind it), `multi_sig` = the arguments of that call
    rdhip_ctx* side_child = nullptr;
    bool multi_pending = false;
    unsigned long long multi_sig = 0;
    std::vector<unsigned long long> multi_args;      // ... held exactly (the hash alone could collide: ADVICE r05)
    // rdhip_rollout_jacobian, models without the fused launch, time-major stacks: the Jacobians of a finished piece of the rollout run on this stream
    // while the rollout goes on (created on first use)
    hipStream_t pipe = nullptr;
    hipEvent_t pipe_ev = nullptr, pipe_done = nullptr;
};
static constexpr int PROGRESS_WORDS = 4096;
struct rdhip_model {
    ModelDesc d;
    LieDesc lie;
};

static thread_local std::string g_err;

static int fail(rdhip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_err = buf;
    return code;
}
#define HIP_TRY(ctx, call)                                                                             \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, RDHIP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

// for comm.hip (the RCCL exchange lives in its own translation unit)
int rdhip_ctx_device_(const rdhip_ctx* c) { return c->device; }
void rdhip_ctx_set_error_(rdhip_ctx* c, const char* msg) { if (c) c->err = msg; g_err = msg; }

static int rot_params(int rot) { return rot == RDHIP_ROT_QUAT ? 4 : 3; }   // src/liestate.jl:42-46

static void build_lie(rdhip_model* M) {
    LieDesc& L = M->lie; const ModelDesc& d = M->d;
    std::memset(&L, 0, sizeof L);
    L.n = d.n; L.e = d.e; L.nrot = d.lie_nP > 0 ? d.lie_nP - 1 : 0;
    if (d.lie_nP == 0) {
        for (int i = 0; i < d.n && i < 16; ++i) { L.row_kind[i] = 0; L.row_col[i] = (signed char)i; L.err_src[i] = (signed char)i; }
        return;
    }
    L.rot = d.rot;
    const int rp = rot_params(d.rot);
    int r = 0, c = 0;
    for (int v = 0; v < d.lie_nP; ++v) {
        for (int j = 0; j < d.lie_P[v]; ++j) { L.row_kind[r] = 0; L.row_col[r] = (signed char)c; L.err_src[c] = (signed char)r; ++r; ++c; }
        if (v < d.lie_nP - 1) {
            L.rot_row0[v] = (signed char)r; L.rot_col0[v] = (signed char)c;
            // G tables (errstate_jacobian!, nabla)
            // (MRP / RodriguesParam rows -- kind 2 -- are read by errstate3_kernel only)
            for (int i = 0; i < rp; ++i) { L.row_kind[r + i] = rp == 4 ? 1 : 2; L.row_col[r + i] = (signed char)c; L.row_sub[r + i] = (signed char)i; L.row_q0[r + i] = (signed char)r; }
            for (int j = 0; j < 3; ++j) L.err_src[c + j] = (signed char)(-1 - v);
            r += rp; c += 3;
        }
    }
}

// ---- tuning knobs (common.hpp TuneKnobs): RDHIP_TUNE="key=value,..." read once per process; rdhip_tune() merges more pairs at run time ----
static int parse_tune(rd::TuneKnobs& k, const char* s, std::string* bad) {
    std::string str(s ? s : "");
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find_first_of(",; ", pos);
        if (end == std::string::npos) end = str.size();
        const std::string kv = str.substr(pos, end - pos);
        pos = end + 1;
        if (kv.empty()) continue;
        const size_t eq = kv.find('=');
        const std::string key = kv.substr(0, eq), val = eq == std::string::npos ? "" : kv.substr(eq + 1);
        const long long iv = atoll(val.c_str());
        if (key == "static_grid") k.static_grid = iv;
        else if (key == "dyn_grid") k.dyn_grid = iv;
        else if (key == "dynamic_tiles") k.dynamic_tiles = !(val == "0");
        else if (key == "small") k.small = val.empty() ? 0 : val[0];
        else if (key == "small_max") k.small_max = iv;
        else if (key == "rollout") k.rollout = val.empty() ? 0 : val[0];
        else if (key == "rolljac") k.rolljac = val.empty() ? 0 : val[0];
        else if (key == "fused_ratio") k.fused_ratio = atof(val.c_str());
        else if (key == "fused_jac_workgroups") k.fused_jac_workgroups = iv;
        else if (key == "fused_tj_max_knots") k.fused_tj_max_knots = iv;
        else if (key == "rolljac_pipe_min_knots") k.rolljac_pipe_min_knots = iv;
        else if (key == "multi_main") k.multi_main = val.empty() ? -1 : (int)iv;
        else if (key == "ift") k.ift = val.empty() ? 0 : val[0];
        else if (key == "mixed") k.mixed = !(val == "0");
        else if (key == "fused_two_per_cu") k.fused_two_per_cu = !(val == "0");
        else if (key == "mixed_idle") k.mixed_idle = val.empty() ? -1 : (int)iv;
        else { if (bad) *bad = key; return RDHIP_ERR_INVALID_ARG; }
    }
    return RDHIP_OK;
}
static rd::TuneKnobs knobs_from_env() {
    rd::TuneKnobs k;
    std::string bad;
    if (const char* e = getenv("RDHIP_TUNE")) if (parse_tune(k, e, &bad) != RDHIP_OK) fprintf(stderr, "[rdhip] RDHIP_TUNE: unknown key '%s' (include/rdhip.h lists the knobs); the rest of the string is ignored\n", bad.c_str());
    return k;
}
namespace rd { TuneKnobs& tune_knobs_mut() { static TuneKnobs k = knobs_from_env(); return k; } }

extern "C" {

int rdhip_tune(const char* pairs) {
    if (!pairs || !pairs[0] || std::string(pairs) == "reset") { rd::tune_knobs_mut() = knobs_from_env(); return RDHIP_OK; }
    rd::TuneKnobs k = rd::tune_knobs_mut();
    std::string bad;
    if (parse_tune(k, pairs, &bad) != RDHIP_OK) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_tune: unknown key '%s'", bad.c_str());
    rd::tune_knobs_mut() = k;
    return RDHIP_OK;
}

int rdhip_version(void) { return RDHIP_VERSION; }

int rdhip_device_count(int* count) {
    if (!count) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "count is NULL");
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return fail(nullptr, RDHIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n; return RDHIP_OK;
}

int rdhip_ctx_create(int device_id, void* stream, int stream_mode, rdhip_ctx** out) {
    if (!out) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (stream_mode != RDHIP_STREAM_OWN && stream_mode != RDHIP_STREAM_BORROW) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad stream_mode %d", stream_mode);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        // RDHIP_DRY_RUN=1 (test hook, CPU-only hosts): a context WITHOUT a device, good for one thing -- driving the communicator bookkeeping of comm.hip
        // against a recording stand-in for librccl (RDHIP_RCCL_LIB).  Every compute / memory entry point refuses it with RDHIP_ERR_NO_DEVICE.
        const char* dry = getenv("RDHIP_DRY_RUN");
        if (dry && dry[0] == '1' && device_id >= 0) {
            rdhip_ctx* c = new (std::nothrow) rdhip_ctx();
            if (!c) return fail(nullptr, RDHIP_ERR_HIP, "out of host memory");
            c->device = -1 - device_id; c->own_stream = false; c->stream = nullptr; c->ev0 = c->ev1 = nullptr;
            for (int i = 0; i < 8; ++i) { c->ws[i] = nullptr; c->ws_bytes[i] = 0; }
            *out = c; return RDHIP_OK;
        }
        return fail(nullptr, RDHIP_ERR_NO_DEVICE, "no HIP device available (librdhip has no CPU fallback)");
    }
    if (device_id < 0 || device_id >= n) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", device_id, n);
    rdhip_ctx* c = new (std::nothrow) rdhip_ctx();
    if (!c) return fail(nullptr, RDHIP_ERR_HIP, "out of host memory");
    c->device = device_id; c->own_stream = (stream_mode == RDHIP_STREAM_OWN);
    { static std::atomic<unsigned long long> next_serial{1}; c->serial = next_serial.fetch_add(1); }
    for (int i = 0; i < 8; ++i) { c->ws[i] = nullptr; c->ws_bytes[i] = 0; }
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) { if (c->own_stream) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); else c->stream = (hipStream_t)stream; }
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc(&c->d_flag, (size_t)2 * (1 + IMPLICIT_FLAG_WORDS) * sizeof(int));     // (two sets: one per stream of the launch-overlap mode)
    if (e == hipSuccess) e = hipMemset(c->d_flag + (1 + IMPLICIT_FLAG_WORDS), 0, sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, sizeof(int));          // word 0 = epoch of the last call that raised a tile word (0: none yet)
    if (e == hipSuccess) e = hipMalloc(&c->d_progress, (size_t)PROGRESS_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->d_progress, 0, (size_t)PROGRESS_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);          // (the memsets above run on the null stream, which a non-blocking stream's launches are not ordered behind)
    if (e != hipSuccess) { int rc = fail(nullptr, RDHIP_ERR_HIP, "context creation failed: %s", hipGetErrorString(e)); delete c; return rc; }
    *out = c; return RDHIP_OK;
}
int rdhip_ctx_destroy(rdhip_ctx* c) {
    if (!c) return RDHIP_OK;
    if (c->device < 0) { delete c; return RDHIP_OK; }      // dry context: nothing on any device
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 8; ++i) if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->mws) (void)hipFree(c->mws);
    for (auto& sl : c->tslot) { if (sl.host) (void)hipHostFree(sl.host); if (sl.dev) (void)hipFree(sl.dev); if (sl.done) (void)hipEventDestroy(sl.done); }
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_progress) (void)hipFree(c->d_progress);
    if (c->d_sched) (void)hipFree(c->d_sched);
    if (c->side_child) { (void)rdhip_ctx_destroy(c->side_child); c->side_child = nullptr; }      // (borrows c->side: first)
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    for (hipEvent_t e : {c->side_done, c->main_mark, c->main_done}) if (e) (void)hipEventDestroy(e);
    for (auto& ch : c->child) if (ch) { (void)rdhip_ctx_destroy(ch); ch = nullptr; }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    for (auto& e : c->join_ev) if (e) (void)hipEventDestroy(e);
    if (c->pipe) { (void)hipStreamSynchronize(c->pipe); (void)hipStreamDestroy(c->pipe); }
    for (hipEvent_t e : {c->pipe_ev, c->pipe_done}) if (e) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c; return RDHIP_OK;
}
// order the context's stream behind whatever the internal stream still runs (no-op unless overlap mode has put work there)
static int side_join(rdhip_ctx* c) {
    c->multi_sig = 0;                              // whatever comes next is not the repetition of a mixed-model call any more
    if (!c->side_pending) return RDHIP_OK;
    HIP_TRY(c, hipEventRecord(c->side_done, c->side));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_done, 0));
    c->side_pending = false; c->multi_pending = false;
    c->pend[0].clear(); c->pend[1].clear();       // (the context's stream is ordered behind the internal one, and future internal work starts behind a fresh mark)
    return RDHIP_OK;
}
int rdhip_ctx_set_overlap(rdhip_ctx* c, int on) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    if (c->capturing) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_ctx_set_overlap: the context is capturing");
    HIP_TRY(c, hipSetDevice(c->device));
    if (on && !c->own_stream) return fail(c, RDHIP_ERR_UNSUPPORTED, "rdhip_ctx_set_overlap: only for contexts that own their stream (work the caller enqueues on a borrowed stream "
                                                                     "could not be ordered behind the internal stream)");
    if (on && !c->side) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->main_mark, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->main_done, hipEventDisableTiming));
    }
    if (!on) { const int rc = side_join(c); if (rc) return rc; }
    c->overlap = on != 0;
    return RDHIP_OK;
}
int rdhip_ctx_fence(rdhip_ctx* c) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    HIP_TRY(c, hipSetDevice(c->device));
    return side_join(c);
}
int rdhip_ctx_synchronize(rdhip_ctx* c) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); return RDHIP_OK;
}
void* rdhip_ctx_stream(rdhip_ctx* c) { if (c && c->device >= 0) (void)side_join(c); return c ? (void*)c->stream : nullptr; }      // (overlap mode: what has been issued so far is ordered on it)
const char* rdhip_last_error(rdhip_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
int rdhip_last_kernel_name(char* buf, int capacity) {
    if (!buf || capacity < 1) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_last_kernel_name: no buffer");
    buf[0] = 0;
    const LastKernel k = last_kernel();
    const char* raw = k.host_fn ? hipKernelNameRefByPtr(k.host_fn, nullptr) : k.mod_fn ? hipKernelNameRef(k.mod_fn) : nullptr;
    if (!raw) { (void)hipGetLastError(); return RDHIP_OK; }     // nothing launched on this thread yet
    int st = 0;
    char* dem = abi::__cxa_demangle(raw, nullptr, nullptr, &st);
    std::snprintf(buf, (size_t)capacity, "%s%s", (st == 0 && dem) ? dem : raw, k.dyn ? " [dynamic tile schedule]" : "");
    std::free(dem);
    return RDHIP_OK;
}
// ---- HIP graphs: record a sequence of device-memory calls once, replay it with one launch (small batches are launch-bound) ----
struct rdhip_graph { hipGraph_t graph; hipGraphExec_t exec; int device; };
int rdhip_graph_begin_capture(rdhip_ctx* c) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    if (c->capturing) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_graph_begin_capture: the context is already capturing");
    HIP_TRY(c, hipSetDevice(c->device));
    { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    return RDHIP_OK;
}
int rdhip_graph_end_capture(rdhip_ctx* c, rdhip_graph** out) {
    if (!c || !out) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (!c->capturing) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_graph_end_capture: the context is not capturing");
    c->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (e != hipSuccess || !g) { (void)hipGetLastError(); return fail(c, RDHIP_ERR_HIP, "hipStreamEndCapture failed: %s (a call inside the capture was not capturable)", hipGetErrorString(e)); }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return fail(c, RDHIP_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    rdhip_graph* G = new (std::nothrow) rdhip_graph{g, x, c->device};
    if (!G) { (void)hipGraphExecDestroy(x); (void)hipGraphDestroy(g); return fail(c, RDHIP_ERR_HIP, "out of host memory"); }
    *out = G; return RDHIP_OK;
}
int rdhip_graph_launch(rdhip_ctx* c, rdhip_graph* g) {
    if (!c || !g) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    if (c->capturing) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_graph_launch: the context is capturing");
    if (g->device != c->device) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_graph_launch: the graph was captured on device %d, the context is on device %d", g->device, c->device);
    HIP_TRY(c, hipSetDevice(c->device));
    { const int rc = side_join(c); if (rc) return rc; }     // ORDERING RULE (rdhip.h): behind everything issued so far, the internal stream of the overlap mode included
    HIP_TRY(c, hipGraphLaunch(g->exec, c->stream));
    return RDHIP_OK;
}
int rdhip_graph_destroy(rdhip_graph* g) {
    if (!g) return RDHIP_OK;
    (void)hipGraphExecDestroy(g->exec); (void)hipGraphDestroy(g->graph);
    delete g; return RDHIP_OK;
}

int rdhip_timer_start(rdhip_ctx* c) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream)); return RDHIP_OK;
}
int rdhip_timer_stop(rdhip_ctx* c, float* ms) {
    if (!c || !ms) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream)); HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1)); return RDHIP_OK;
}
int rdhip_malloc(rdhip_ctx* c, void** p, unsigned long long bytes) {
    if (!c || !p) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipMalloc(p, bytes ? bytes : 1)); return RDHIP_OK;
}
int rdhip_free(rdhip_ctx* c, void* p) {
    // ctx may be NULL: a garbage-collected host (the Julia module's finalizers run in no particular order) must be able to release device memory
    // after the context it was allocated through is gone; hipFree finds the owning device from the pointer
    if (c && c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    if (c) HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipFree(p)); return RDHIP_OK;
}
int rdhip_memcpy_h2d(rdhip_ctx* c, void* dst, const void* src, unsigned long long bytes) {
    if (!c) return fail(c, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    HIP_TRY(c, hipSetDevice(c->device)); { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); return RDHIP_OK;
}
int rdhip_memcpy_d2h(rdhip_ctx* c, void* dst, const void* src, unsigned long long bytes) {
    if (!c) return fail(c, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    HIP_TRY(c, hipSetDevice(c->device)); { const int rc = side_join(c); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(c, hipStreamSynchronize(c->stream)); return RDHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// models
// ---------------------------------------------------------------------------------------------------------
int rdhip_model_create(int kind, int rot, int frame, const double* params, int nparams, rdhip_model** out) {
    if (!out) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (nparams < 0 || nparams > 64 || (nparams > 0 && !params)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad params (nparams=%d)", nparams);
    if (rot < RDHIP_ROT_QUAT || rot > RDHIP_ROT_RP) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad rotation %d", rot);
    if (frame != RDHIP_FRAME_WORLD && frame != RDHIP_FRAME_BODY) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad velocity frame %d", frame);
    rdhip_model* M = new (std::nothrow) rdhip_model();
    if (!M) return fail(nullptr, RDHIP_ERR_HIP, "out of host memory");
    ModelDesc& d = M->d; std::memset(&d, 0, sizeof d);
    d.kind = kind; d.rot = rot; d.frame = frame; d.nparams = nparams;
    for (int i = 0; i < nparams; ++i) d.params[i] = params[i];
    const int rp = rot_params(rot);
    int need = 0;
    switch (kind) {
        case RDHIP_MODEL_CARTPOLE: d.n = 4; d.m = 1; need = 4; break;
        case RDHIP_MODEL_DOUBLE_INTEGRATOR: need = 1; if (nparams >= 1) { d.m = (int)params[0]; d.n = 2 * d.m; } break;
        case RDHIP_MODEL_LINEAR: need = 2; if (nparams >= 2) { d.n = (int)params[0]; d.m = (int)params[1]; need = 2 + d.n * d.n + d.n * d.m; } break;
        case RDHIP_MODEL_SATELLITE: d.n = 9 + rp; d.m = 6; need = 10; d.lie_nP = 2; d.lie_P[0] = 3; d.lie_P[1] = 6; break;       // src/rigidbody.jl:48
        case RDHIP_MODEL_QUADROTOR: d.n = 9 + rp; d.m = 4; need = 16; d.lie_nP = 2; d.lie_P[0] = 3; d.lie_P[1] = 6; break;
        case RDHIP_MODEL_SATELLITE_PAIR: d.n = 2 * rp + 6; d.m = 6; need = 18; d.lie_nP = 3; d.lie_P[0] = 0; d.lie_P[1] = 3; d.lie_P[2] = 3; break;
        case RDHIP_MODEL_LIE_SATELLITE: d.n = 3 + rp; d.m = 3; need = 3; d.lie_nP = 2; d.lie_P[0] = 3; d.lie_P[1] = 0; break;
        default: delete M; return fail(nullptr, RDHIP_ERR_INVALID_ARG, "unknown model kind %d", kind);
    }
    if (nparams != need) { delete M; return fail(nullptr, RDHIP_ERR_INVALID_ARG, "model kind %d expects %d params, got %d", kind, need, nparams); }
    if (d.n <= 0 || d.m <= 0 || d.n > 16 || d.m > 8) { delete M; return fail(nullptr, RDHIP_ERR_INVALID_ARG, "unsupported dimensions n=%d m=%d", d.n, d.m); }
    // errstate_dim: 3 per rotation + vector parts (src/liestate.jl:124) ; Euclidean: n (src/statevectortype.jl:144)
    if (d.lie_nP) { int sp = 0; for (int i = 0; i < d.lie_nP; ++i) sp += d.lie_P[i]; d.e = 3 * (d.lie_nP - 1) + sp; } else d.e = d.n;
    d.newton_iters = 10; d.newton_tol = 1e-12;
    build_lie(M);
    *out = M; return RDHIP_OK;
}
int rdhip_model_destroy(rdhip_model* M) { if (M && M->d.jit) jit_model_destroy(M->d.jit); delete M; return RDHIP_OK; }

// shared by the two custom-model constructors: argument checks, the ModelDesc, the LieState partition; `discrete` picks the plug-in object
static int create_custom(int n, int m, const char* body, const char* jac_body, const char* preamble, const double* params, int nparams, int rot, const int* P, int nP, bool discrete, rdhip_model** out) {
    if (!out) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!body || !*body) return fail(nullptr, RDHIP_ERR_INVALID_ARG, discrete ? "step_body is empty" : "dynamics_body is empty");
    if (n <= 0 || m <= 0 || n > 16 || m > 8) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "unsupported dimensions n=%d m=%d (n <= 16, m <= 8)", n, m);
    if (nparams < 0 || nparams > 64 || (nparams > 0 && !params)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad params (nparams=%d)", nparams);
    if (nP < 0 || nP > 8 || (nP > 0 && !P)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad LieState partition (nP=%d)", nP);
    if (nP > 0 && (rot < RDHIP_ROT_QUAT || rot > RDHIP_ROT_RP)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad rotation %d", rot);
    if (nP - 1 > 4) return fail(nullptr, RDHIP_ERR_UNSUPPORTED, "a custom LieState may hold at most 4 rotations (got %d)", nP - 1);
    rdhip_model* M = new (std::nothrow) rdhip_model();
    if (!M) return fail(nullptr, RDHIP_ERR_HIP, "out of host memory");
    ModelDesc& d = M->d; std::memset(&d, 0, sizeof d);
    d.kind = RDHIP_MODEL_CUSTOM; d.rot = nP > 0 ? rot : RDHIP_ROT_QUAT; d.frame = RDHIP_FRAME_WORLD; d.n = n; d.m = m; d.nparams = nparams;
    for (int i = 0; i < nparams; ++i) d.params[i] = params[i];
    d.lie_nP = nP;
    int sp = 0; for (int i = 0; i < nP; ++i) { if (P[i] < 0) { delete M; return fail(nullptr, RDHIP_ERR_INVALID_ARG, "negative LieState block"); } d.lie_P[i] = P[i]; sp += P[i]; }
    if (nP > 0) {
        const int len = rot_params(d.rot) * (nP - 1) + sp;                  // length(LieState{R,P}), src/liestate.jl:121-123
        if (len != n) { delete M; return fail(nullptr, RDHIP_ERR_DIM_MISMATCH, "LieState partition has length %d but n = %d", len, n); }
        d.e = 3 * (nP - 1) + sp;
    } else d.e = n;
    d.jit = discrete ? jit_model_create_discrete(n, m, body, jac_body, preamble, d.params, nparams, d.rot == RDHIP_ROT_QUAT, d.lie_P, nP)
                     : jit_model_create(n, m, body, preamble, d.params, nparams, d.rot == RDHIP_ROT_QUAT, d.lie_P, nP);
    d.newton_iters = 10; d.newton_tol = 1e-12;
    build_lie(M);
    *out = M; return RDHIP_OK;
}
int rdhip_model_create_custom(int n, int m, const char* body, const char* preamble, const double* params, int nparams, int rot, const int* P, int nP, rdhip_model** out) {
    return create_custom(n, m, body, nullptr, preamble, params, nparams, rot, P, nP, false, out);
}
int rdhip_model_create_custom_discrete(int n, int m, const char* step_body, const char* jac_body, const char* preamble, const double* params, int nparams, int rot, const int* P, int nP,
                                       rdhip_model** out) {
    return create_custom(n, m, step_body, jac_body, preamble, params, nparams, rot, P, nP, true, out);
}
int rdhip_model_set_user_jacobian(rdhip_model* M, const char* jac_body) {
    if (!M || !M->d.jit) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_model_set_user_jacobian: not a custom model");
    return jit_set_user_jacobian(M->d.jit, jac_body);
}
int rdhip_model_is_discrete(const rdhip_model* M, int* discrete) {
    if (!M || !discrete) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *discrete = (M->d.kind == RDHIP_MODEL_CUSTOM && jit_is_discrete(M->d.jit)) ? 1 : 0; return RDHIP_OK;
}
// the rule argument of a call against the model: a user-defined DiscreteDynamics model IS its own rule (RDHIP_DISCRETE, nothing else); every other model
// takes lo .. hi of the quadrature rules
static int rule_ok(rdhip_ctx* c, const rdhip_model* M, int rule, int lo, int hi, const char* what) {
    const bool discrete = M && M->d.kind == RDHIP_MODEL_CUSTOM && jit_is_discrete(M->d.jit);
    if (discrete) return rule == RDHIP_DISCRETE ? RDHIP_OK : fail(c, RDHIP_ERR_INVALID_ARG, "%s: a DiscreteDynamics model has no quadrature rule and no continuous dynamics: pass RDHIP_DISCRETE (got %d)", what, rule);
    if (rule == RDHIP_DISCRETE) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: RDHIP_DISCRETE is the rule of user-defined discrete models (rdhip_model_create_custom_discrete)", what);
    if (rule < lo || rule > hi) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad rule %d", what, rule);
    return RDHIP_OK;
}
int rdhip_model_set_newton(rdhip_model* M, int iters, double tol) {
    if (!M) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "model is NULL");
    if (iters < 1 || !(tol > 0)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "newton_iters must be >= 1 and newton_tol > 0");
    M->d.newton_iters = iters; M->d.newton_tol = tol; return RDHIP_OK;
}
int rdhip_model_balance_columns(rdhip_model* M) {
    if (!M || !M->d.jit) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_model_balance_columns: not a custom model");
    if (jit_is_discrete(M->d.jit)) return fail(nullptr, RDHIP_ERR_UNSUPPORTED, "rdhip_model_balance_columns: continuous custom models only");
    const int rc = jit_balance(M->d.jit);
    return rc ? fail(nullptr, rc, "%s", jit_last_error(M->d.jit)) : RDHIP_OK;
}
int rdhip_model_set_cascade(rdhip_model* M, int nblocks, const int* state_block, const int* control_block) {
    if (!M || !M->d.jit) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_model_set_cascade: not a custom model");
    if (jit_is_discrete(M->d.jit)) return fail(nullptr, RDHIP_ERR_UNSUPPORTED, "rdhip_model_set_cascade: continuous custom models only (a discrete step has no stage points to share)");
    const int rc = jit_set_cascade(M->d.jit, nblocks, state_block, control_block);
    return rc ? fail(nullptr, rc, "rdhip_model_set_cascade: %s", jit_last_error(M->d.jit)) : RDHIP_OK;
}
int rdhip_model_precompile(rdhip_model* M, int op, int rule, int dtype) {
    if (!M || !M->d.jit) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_model_precompile: not a custom model");
    if (op < 0 || (op > 20 && !(op >= 32 && (op - 32 == 11 || op - 32 == 19 || op - 32 == 20 || (op - 32 >= 2 && op - 32 <= 6)))) || rule < RDHIP_CONTINUOUS || rule > RDHIP_DISCRETE || (dtype != RDHIP_F64 && dtype != RDHIP_F32)) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_model_precompile: bad op / rule / dtype");
    const int rc = jit_precompile(M->d.jit, op, rule, dtype);
    return rc ? fail(nullptr, rc, "%s", jit_last_error(M->d.jit)) : RDHIP_OK;
}
int rdhip_model_dims(const rdhip_model* M, int* n, int* m, int* p, int* e) {
    if (!M) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "model is NULL");
    if (n) *n = M->d.n; if (m) *m = M->d.m; if (p) *p = M->d.n; if (e) *e = M->d.e; return RDHIP_OK;
}
int rdhip_model_statevectortype(const rdhip_model* M, int* is_rot) {
    if (!M || !is_rot) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *is_rot = M->d.lie_nP > 0; return RDHIP_OK;
}
int rdhip_model_liestate(const rdhip_model* M, int* rot, int* nP, int* P) {
    if (!M || !rot || !nP || !P) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *rot = M->d.rot; *nP = M->d.lie_nP; for (int i = 0; i < M->d.lie_nP; ++i) P[i] = M->d.lie_P[i]; return RDHIP_OK;
}
int rdhip_model_gen_inds(const rdhip_model* M, int first[5], int len[5]) {
    if (!M || !first || !len) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "NULL argument");
    if (M->d.kind != RDHIP_MODEL_SATELLITE && M->d.kind != RDHIP_MODEL_QUADROTOR) return fail(nullptr, RDHIP_ERR_UNSUPPORTED, "gen_inds is defined for RigidBody models only");
    const int rp = rot_params(M->d.rot);
    first[0] = 1; len[0] = 3; first[1] = 4; len[1] = rp; first[2] = 4 + rp; len[2] = 3; first[3] = 7 + rp; len[3] = 3; first[4] = 10 + rp; len[4] = M->d.m;
    return RDHIP_OK;
}
int rdhip_liestate_query(int rot, const int* P, int nP, int* length, int* edim, int* rot_first, int* vec_first) {
    if (!P || nP < 1 || nP > 8 || !length || !edim || !rot_first || !vec_first) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad LieState query");
    if (rot < RDHIP_ROT_QUAT || rot > RDHIP_ROT_RP) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "bad rotation %d", rot);
    const int rp = rot_params(rot); int sp = 0;
    for (int i = 0; i < nP; ++i) { if (P[i] < 0) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "negative partition"); sp += P[i]; }
    *length = rp * (nP - 1) + sp; *edim = 3 * (nP - 1) + sp;
    int acc = 0;
    for (int i = 1; i <= nP; ++i) {   // 1-based like Julia
        vec_first[i - 1] = acc + (i - 1) * rp + 1;            // vec_inds(R,P,i)
        acc += P[i - 1];
        if (i <= nP - 1) rot_first[i - 1] = acc + (i - 1) * rp + 1;   // rot_inds(R,P,i)
    }
    return RDHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// staging helper for RDHIP_HOST calls
// ---------------------------------------------------------------------------------------------------------
struct Stage {
    rdhip_ctx* c; int slot = 0; int rc = RDHIP_OK;
    struct Out { void* host; void* dev; size_t bytes; };
    std::vector<Out> outs;
    explicit Stage(rdhip_ctx* ctx) : c(ctx) {}
    void* reserve(size_t bytes) {
        if (rc) return nullptr;
        // the grow-only workspace may hipMalloc / hipFree (illegal on a capturing thread) and a later call may free the slot a captured
        // kernel would write to: a call that needs it cannot be recorded into a graph
        if (c->capturing) { rc = fail(c, RDHIP_ERR_INVALID_ARG, "this call needs the context's internal workspace and cannot be captured into a graph: pass the optional output (Y) / 16-byte aligned buffers so that it does not"); return nullptr; }
        if (slot >= 8) { rc = fail(c, RDHIP_ERR_HIP, "staging slots exhausted"); return nullptr; }
        if (c->ws_bytes[slot] < bytes) {
            if (c->ws[slot]) (void)hipFree(c->ws[slot]);
            c->ws[slot] = nullptr; c->ws_bytes[slot] = 0;
            hipError_t e = hipMalloc(&c->ws[slot], bytes ? bytes : 1);
            if (e != hipSuccess) { rc = fail(c, RDHIP_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
            c->ws_bytes[slot] = bytes;
        }
        return c->ws[slot++];
    }
    const void* in(const void* host, size_t bytes) {
        if (!host) return nullptr;
        void* d = reserve(bytes); if (!d) return nullptr;
        hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = fail(c, RDHIP_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e));
        return d;
    }
    void* out(void* host, size_t bytes, bool preload = false) {
        if (!host) return nullptr;
        void* d = reserve(bytes); if (!d) return nullptr;
        if (preload) { hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream); if (e != hipSuccess) rc = fail(c, RDHIP_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e)); }
        outs.push_back({host, d, bytes}); return d;
    }
    int finish(int krc) {
        if (rc) return rc;
        if (krc) return krc;
        for (auto& o : outs) { hipError_t e = hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream); if (e != hipSuccess) return fail(c, RDHIP_ERR_HIP, "D2H copy failed: %s", hipGetErrorString(e)); }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, RDHIP_ERR_HIP, "stream sync failed: %s", hipGetErrorString(e));
        return RDHIP_OK;
    }
};

static int check_common(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, int memspace) {
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (!M) return fail(c, RDHIP_ERR_INVALID_ARG, "model is NULL");
    if (dtype != RDHIP_F64 && dtype != RDHIP_F32) return fail(c, RDHIP_ERR_INVALID_ARG, "bad dtype %d", dtype);
    if (N < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "negative N");
    if (memspace != RDHIP_DEVICE && memspace != RDHIP_HOST) return fail(c, RDHIP_ERR_INVALID_ARG, "bad memspace %d", memspace);
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN): no compute entry point works on it");
    if (c->capturing && memspace == RDHIP_HOST) return fail(c, RDHIP_ERR_INVALID_ARG, "host-memory calls copy and synchronise: they cannot be captured into a graph");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, RDHIP_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
    if (c->side_pending && !c->in_overlapped_call) return side_join(c);       // any other operation: ordered behind everything issued so far
    return RDHIP_OK;
}
// what a MRP / RodriguesParam model can do: op 0 = dynamics-type calls (the RigidBody{MRP|RodriguesParam} satellites and quadrotors, models.hpp
// RigidBody3, and custom models; the error-state expansion builds on them), op 1 = the error-state maps (any LieState{MRP|RodriguesParam,P}: the 3 x 3
// blocks of errstate.hip)
static int check_rotation(rdhip_ctx* c, const rdhip_model* M, int op) {
    if (M->d.lie_nP == 0 || M->d.rot == RDHIP_ROT_QUAT) return RDHIP_OK;
    if (op == 1) return RDHIP_OK;
    if (M->d.kind == RDHIP_MODEL_SATELLITE || M->d.kind == RDHIP_MODEL_QUADROTOR || M->d.kind == RDHIP_MODEL_CUSTOM) return RDHIP_OK;   // (a custom model's dynamics are the user's source)
    return fail(c, RDHIP_ERR_UNSUPPORTED, "MRP / RodriguesParam dynamics exist for the RigidBody satellite and quadrotor only");
}
static thread_local const ModelDesc* g_jit_model = nullptr;     // set by family_* for custom models so that kernel_status can report the JIT error
static int kernel_status(rdhip_ctx* c, int rc, const char* what) {
    if (rc == RDHIP_OK) return rc;
    if (g_jit_model && g_jit_model->jit) { const ModelDesc* d = g_jit_model; g_jit_model = nullptr; return fail(c, rc, "%s: %s", what, jit_last_error(d->jit)); }
    if (rc == RDHIP_ERR_HIP) return fail(c, rc, "%s: kernel launch failed: %s", what, hipGetErrorString(hipGetLastError()));
    if (rc == RDHIP_ERR_NOT_IMPLEMENTED) return fail(c, rc, "%s: User-defined Jacobian not implemented for this model", what);
    if (rc == RDHIP_ERR_UNSUPPORTED) return fail(c, rc, "%s: model dimensions / kind not compiled in", what);
    return fail(c, rc, "%s: invalid rule / diff / dtype", what);
}

static int family_eval(const ModelDesc& d, int rule, int dtype, const LaunchArgs& a) {
    g_jit_model = nullptr;
    if (d.kind == RDHIP_MODEL_CUSTOM) { const int rc = jit_eval(d.jit, rule, dtype, a); if (rc) g_jit_model = &d; return rc; }
    switch (d.kind) {
        case RDHIP_MODEL_CARTPOLE: return fam_cartpole_eval(d, rule, dtype, a);
        case RDHIP_MODEL_DOUBLE_INTEGRATOR: case RDHIP_MODEL_LINEAR: return fam_simple_eval(d, rule, dtype, a);
        case RDHIP_MODEL_SATELLITE: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_eval(d, rule, dtype, a) : fam_satellite_eval(d, rule, dtype, a);
        case RDHIP_MODEL_QUADROTOR: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_eval(d, rule, dtype, a) : fam_quadrotor_eval(d, rule, dtype, a);
        default: return fam_lie_eval(d, rule, dtype, a);
    }
}
static int family_jacobian(const ModelDesc& d, int rule, int diff, int dtype, const LaunchArgs& a) {
    g_jit_model = nullptr;
    if (d.kind == RDHIP_MODEL_CUSTOM) { const int rc = jit_jacobian(d.jit, rule, diff, dtype, a); if (rc) g_jit_model = &d; return rc; }
    switch (d.kind) {
        case RDHIP_MODEL_CARTPOLE: return fam_cartpole_jacobian(d, rule, diff, dtype, a);
        case RDHIP_MODEL_DOUBLE_INTEGRATOR: case RDHIP_MODEL_LINEAR: return fam_simple_jacobian(d, rule, diff, dtype, a);
        case RDHIP_MODEL_SATELLITE: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_jacobian(d, rule, diff, dtype, a) : fam_satellite_jacobian(d, rule, diff, dtype, a);
        case RDHIP_MODEL_QUADROTOR: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_jacobian(d, rule, diff, dtype, a) : fam_quadrotor_jacobian(d, rule, diff, dtype, a);
        default: return fam_lie_jacobian(d, rule, diff, dtype, a);
    }
}
static int family_rollout(const ModelDesc& d, int rule, int dtype, const LaunchArgs& a) {
    g_jit_model = nullptr;
    if (d.kind == RDHIP_MODEL_CUSTOM) { const int rc = jit_rollout(d.jit, rule, dtype, a); if (rc) g_jit_model = &d; return rc; }
    switch (d.kind) {
        case RDHIP_MODEL_CARTPOLE: return fam_cartpole_rollout(d, rule, dtype, a);
        case RDHIP_MODEL_DOUBLE_INTEGRATOR: case RDHIP_MODEL_LINEAR: return fam_simple_rollout(d, rule, dtype, a);
        case RDHIP_MODEL_SATELLITE: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_rollout(d, rule, dtype, a) : fam_satellite_rollout(d, rule, dtype, a);
        case RDHIP_MODEL_QUADROTOR: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_rollout(d, rule, dtype, a) : fam_quadrotor_rollout(d, rule, dtype, a);
        default: return fam_lie_rollout(d, rule, dtype, a);
    }
}

// rollout + Jacobians in one launch, or RDHIP_ERR_UNSUPPORTED before anything has been launched (models without a cascade structure, custom models)
static int family_rolljac(const ModelDesc& d, int rule, int diff, int dtype, const LaunchArgs& a) {
    g_jit_model = nullptr;
    if (d.rot != RDHIP_ROT_QUAT) return RDHIP_ERR_UNSUPPORTED;
    switch (d.kind) {
        case RDHIP_MODEL_SATELLITE: return fam_satellite_rolljac(d, rule, diff, dtype, a);
        case RDHIP_MODEL_QUADROTOR: return fam_quadrotor_rolljac(d, rule, diff, dtype, a);
        default: return RDHIP_ERR_UNSUPPORTED;
    }
}

// At most ONE spin-waiting launch in flight per device and process.  The Jacobian workgroups of a fused rollout -> Jacobians launch (fused.hpp) hold
// their compute unit while they wait for progress words.  ONE such launch cannot deadlock (its producers have the lowest workgroup indices, every
// XCD places its share of them first, and kernels that do not wait for anybody drain by themselves); TWO in flight can: workgroups are dealt
// round-robin to the 8 XCDs, each XCD arbitrates between hardware queues on its own, so launch A's consumers can fill one XCD and launch B's
// another while each launch's remaining producers are queued behind the other's consumers.  So a fused launch is ordered behind the previous
// fused launch on the device (an event edge when that one went to another stream), and rdhip_rollout_jacobian_multi gives the fused form to one
// model per call (the others run their two launches, whose workgroups wait for nobody, beside it).  Not covered: launches replayed from captured
// graphs and other processes on the same GPU (rdhip.h says so).
static std::mutex g_fused_mu;
// Per device: who issued the last fused launch (context serial + stream handle, compared only, never used) and an event recorded BEHIND it on the
// stream that issued it -- at launch time, when that stream is alive by construction.  The next fused launch from anybody else waits for the event; the
// same issuer's next launch follows in stream order and waits for nothing.  Round 5 kept the raw stream instead and recorded on it LATER, from the next
// launcher's call: a borrowed stream destroyed in between was undefined behaviour that rdhip.h could only forbid.  An event outlives
// the stream it was recorded on.  (Handle reuse -- a new stream at a destroyed one's address -- is why the context's serial is part of the identity.)
// The marker packet costs a stream of back-to-back fused launches 4 us per launch (profiles/r06/c5_whole_timeline.txt: 5.8 us from the end of one 250-us
// launch to the start of the next, 1.4 us without), so an issuer that has had the device to itself for FUSED_EVENT_RUN launches stops recording; if
// somebody else then turns up, that launch finds no event to wait for and waits for the whole device instead, once (hipDeviceSynchronize -- every stream,
// alive or destroyed), and events are recorded again for the next FUSED_EVENT_RUN launches.  A process with one context and one stream, the usual case,
// pays for 8 markers; two issuers that alternate (rdhip_ctx_set_overlap) keep their events.
constexpr int FUSED_EVENT_RUN = 8;
static hipEvent_t g_fused_done[64] = {};
static hipStream_t g_fused_stream[64] = {};
static unsigned long long g_fused_serial[64] = {};
static int g_fused_run[64] = {};              // launches in a row by the current issuer
static bool g_fused_has_event[64] = {};       // the last launch was followed by an event
static int spin_launch(rdhip_ctx* c, hipStream_t stream, bool capturing, const std::function<int()>& launch) {
    std::lock_guard<std::mutex> g(g_fused_mu);
    const int slot = c->device & 63;
    const bool same = g_fused_serial[slot] == c->serial && g_fused_stream[slot] == stream;
    if (!capturing && g_fused_serial[slot] && !same) {
        if (g_fused_has_event[slot]) { if (hipStreamWaitEvent(stream, g_fused_done[slot], 0) != hipSuccess) return RDHIP_ERR_HIP; }
        else if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RDHIP_ERR_HIP;
    }
    const int k = launch();
    if (k == RDHIP_OK && !capturing) {
        g_fused_run[slot] = same ? g_fused_run[slot] + 1 : 0;
        g_fused_has_event[slot] = g_fused_run[slot] < FUSED_EVENT_RUN;
        if (g_fused_has_event[slot]) {
            if (!g_fused_done[slot] && hipEventCreateWithFlags(&g_fused_done[slot], hipEventDisableTiming) != hipSuccess) return RDHIP_ERR_HIP;
            if (hipEventRecord(g_fused_done[slot], stream) != hipSuccess) return RDHIP_ERR_HIP;
        }
        g_fused_stream[slot] = stream; g_fused_serial[slot] = c->serial;
    }
    return k;
}
static int fused_launch(rdhip_ctx* c, const ModelDesc& d, int rule, int diff, int dtype, const LaunchArgs& f, bool capturing) {
    return spin_launch(c, f.stream, capturing, [&] { return family_rolljac(d, rule, diff, dtype, f); });
}

static int family_implicit(rdhip_ctx* c, const ModelDesc& d, int mode, int dtype, LaunchArgs a, int flag_set = 0) {
    g_jit_model = nullptr;
    a.newton_iters = d.newton_iters; a.newton_tol = d.newton_tol;
    if (mode != 2) {     // Newton kernels: register pass + gated LDS pass communicate through a device word
        a.flag = c->d_flag + (size_t)flag_set * (1 + IMPLICIT_FLAG_WORDS);
        c->epoch = c->epoch >= (1 << 22) ? 1 : c->epoch + 1;
        a.epoch = c->epoch;
    }
    switch (d.kind) {
        case RDHIP_MODEL_CARTPOLE: return fam_cartpole_implicit(d, mode, dtype, a);
        case RDHIP_MODEL_DOUBLE_INTEGRATOR: case RDHIP_MODEL_LINEAR: return fam_simple_implicit(d, mode, dtype, a);
        case RDHIP_MODEL_SATELLITE: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_implicit(d, mode, dtype, a) : fam_satellite_implicit(d, mode, dtype, a);
        case RDHIP_MODEL_QUADROTOR: return d.rot != RDHIP_ROT_QUAT ? fam_rigid3_implicit(d, mode, dtype, a) : fam_quadrotor_implicit(d, mode, dtype, a);
        case RDHIP_MODEL_CUSTOM: { const int rc = jit_implicit(d.jit, mode, dtype, a); if (rc) g_jit_model = &d; return rc; }
        default: return fam_lie_implicit(d, mode, dtype, a);
    }
}

static int eval_common(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long N, const void* Z, const double* t, const double* dt,
                       void* Y, int memspace, const char* what) {
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    rc = rule_ok(c, M, rule, RDHIP_CONTINUOUS, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (N > 0 && (!Z || !t || !dt || !Y)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    const bool implicit = rule == RDHIP_IMPLICIT_MIDPOINT;
    LaunchArgs a{}; a.stream = c->stream; a.N = N;
    if (memspace == RDHIP_DEVICE) { a.Z = Z; a.t = t; a.dt = dt; a.Y = Y; return kernel_status(c, implicit ? family_implicit(c, d, 0, dtype, a) : family_eval(d, rule, dtype, a), what); }
    Stage s(c);
    a.Z = s.in(Z, (size_t)N * (d.n + d.m) * ts); a.t = (const double*)s.in(t, (size_t)N * 8); a.dt = (const double*)s.in(dt, (size_t)N * 8);
    a.Y = s.out(Y, (size_t)N * d.n * ts);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, implicit ? family_implicit(c, d, 0, dtype, a) : family_eval(d, rule, dtype, a), what));
}

int rdhip_dynamics(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* Z, const double* t, const double* dt, void* Xdot, int memspace) {
    return eval_common(c, M, RDHIP_CONTINUOUS, dtype, N, Z, t, dt, Xdot, memspace, "rdhip_dynamics");
}
int rdhip_discrete_dynamics(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long N, const void* Z, const double* t, const double* dt, void* Xn, int memspace) {
    if (rule == RDHIP_CONTINUOUS) return fail(c, RDHIP_ERR_INVALID_ARG, "rdhip_discrete_dynamics: a quadrature rule is required");
    return eval_common(c, M, rule, dtype, N, Z, t, dt, Xn, memspace, "rdhip_discrete_dynamics");
}

// wrenches(model, x, u, t) and wrench_jacobian!(F, model, z) for every knot point (src/rigidbody.jl:244-248, :315-320)
static int wrench_common(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* Z, void* Out, int jac, int memspace, const char* what) {
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (N > 0 && (!Z || !Out)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    const ModelDesc& d = M->d;
    if (d.kind != RDHIP_MODEL_SATELLITE && d.kind != RDHIP_MODEL_QUADROTOR)
        return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: defined for the compiled-in RigidBody models (a custom model's forces / moments are part of its dynamics source)", what);
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4, per = jac ? (size_t)6 * (d.n + d.m) : 6;
    LaunchArgs a{}; a.stream = c->stream; a.N = N;
    if (memspace == RDHIP_DEVICE) { a.Z = Z; (jac ? a.J : a.Y) = Out; return kernel_status(c, fam_wrench(d, dtype, a, jac), what); }
    Stage s(c);
    a.Z = s.in(Z, (size_t)N * (d.n + d.m) * ts);
    (jac ? a.J : a.Y) = s.out(Out, (size_t)N * per * ts);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, fam_wrench(d, dtype, a, jac), what));
}
int rdhip_wrenches(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* Z, const double* t, void* W, int memspace) {
    (void)t;                                                      // the compiled-in wrenches do not depend on time
    return wrench_common(c, M, dtype, N, Z, W, 0, memspace, "rdhip_wrenches");
}
int rdhip_wrench_jacobian(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* Z, const double* t, void* F, int memspace) {
    (void)t;
    return wrench_common(c, M, dtype, N, Z, F, 1, memspace, "rdhip_wrench_jacobian");
}
// wrench_sparsity(model) (src/rigidbody.jl:322-347): js[i + 2 j], i = 0 (F) / 1 (M), j = r, q, v, w, u -- 0 where the block is identically zero
int rdhip_model_wrench_sparsity(const rdhip_model* M, int js[10]) {
    if (!M || !js) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "NULL argument");
    const int k = M->d.kind;
    if (k != RDHIP_MODEL_SATELLITE && k != RDHIP_MODEL_QUADROTOR && !(k == RDHIP_MODEL_CUSTOM && M->d.lie_nP == 2 && M->d.lie_P[0] == 3 && M->d.lie_P[1] == 6))
        return fail(nullptr, RDHIP_ERR_UNSUPPORTED, "wrench_sparsity is defined for RigidBody models only");
    for (int i = 0; i < 10; ++i) js[i] = 1;                       // the reference's default: every block may be non-zero
    if (k != RDHIP_MODEL_CUSTOM) {
        // satellite: F = q * u[1:3], M = u[4:6]; quadrotor: F = m g + q * [0, 0, sum kf u], M = M(u): forces see q and u, moments u only
        js[0] = 0; js[1] = 0;  /* d/dr */   js[2] = 1; js[3] = 0;  /* d/dq */   js[4] = 0; js[5] = 0;  /* d/dv */   js[6] = 0; js[7] = 0;  /* d/dw */
    }
    return RDHIP_OK;
}

static bool stream_is_capturing(rdhip_ctx* c);
// the counter words of this launch's dynamic tile schedule on stream `me` (0: the context's, 1: the internal one), or NULL: static schedule (a capturing
// stream -- a graph would bake one slot into every replay --, RDHIP_TUNE=dynamic_tiles=0, or no memory for the words)
// `ntiles`, `dtype`: of the launch -- only launches that can take the schedule at all (fp64, 8 tiles per workgroup of the reduced grid: dispatch.hpp
// dyn_grid) use up a slot.
// A slot is handed out on the promise that its words are zero: the launch that used it before leaves them so (its last workgroup resets them).  A
// launch that never finished -- a fault, an aborted process sharing the context's memory -- would break that promise for good, and every later launch on
// the slot would silently skip tiles.  So each time a stream's ring comes round, the whole ring is zeroed again ON that stream (every launch that used
// it is then complete by stream order): a 2 KB fill per 32 big launches, and a poisoned slot heals within one turn of the ring.
static unsigned* take_sched(rdhip_ctx* c, int me, long long ntiles, int dtype) {
    if (!tune_knobs().dynamic_tiles || dtype != RDHIP_F64 || ntiles < 8LL * (long long)dyn_grid((unsigned)dev_info().cus) || stream_is_capturing(c)) return nullptr;
    if (!c->d_sched) {
        const size_t bytes = (size_t)2 * rdhip_ctx::NSCHED * rdhip_ctx::SCHED_WORDS * sizeof(unsigned);
        if (hipMalloc(&c->d_sched, bytes) != hipSuccess) { (void)hipGetLastError(); c->d_sched = nullptr; return nullptr; }
        // zeroed ON the context's stream and waited for (once per context): a memset of the null stream is not ordered against a non-blocking stream's launches
        if (hipMemsetAsync(c->d_sched, 0, bytes, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(c->d_sched); c->d_sched = nullptr; return nullptr;
        }
    }
    const unsigned turn = c->sched_next[me]++;
    const unsigned slot = turn % rdhip_ctx::NSCHED;
    unsigned* ring = c->d_sched + (size_t)me * rdhip_ctx::NSCHED * rdhip_ctx::SCHED_WORDS;
    if (slot == 0 && turn > 0) {
        hipStream_t st = me == 0 ? c->stream : c->side;
        if (hipMemsetAsync(ring, 0, (size_t)rdhip_ctx::NSCHED * rdhip_ctx::SCHED_WORDS * sizeof(unsigned), st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    return ring + (size_t)slot * rdhip_ctx::SCHED_WORDS;
}
int rdhip_jacobian(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, long long N, const void* Z, const double* t, const double* dt,
                   void* J, void* Y, int memspace) {
    const char* what = "rdhip_jacobian";
    const bool ov = c && c->overlap && memspace == RDHIP_DEVICE && !c->capturing && M && M->d.kind != RDHIP_MODEL_CUSTOM && (rule != RDHIP_IMPLICIT_MIDPOINT || Y != nullptr);
    if (c) c->in_overlapped_call = ov && !c->multi_pending;     // (a mixed-model call left work on the internal stream that is in no footprint list: join first)
    int rc = check_common(c, M, dtype, N, memspace);
    if (c) c->in_overlapped_call = false;
    if (rc) return rc;
    rc = rule_ok(c, M, rule, RDHIP_CONTINUOUS, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (diff < RDHIP_FORWARD_AD || diff > RDHIP_USER_DEFINED) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad diff method %d", what, diff);
    if (N > 0 && (!Z || !t || !dt || !J)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    const bool implicit = rule == RDHIP_IMPLICIT_MIDPOINT;
    if (implicit && diff != RDHIP_FORWARD_AD) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: implicit rules take ImplicitFunctionTheorem(ForwardAD()) only", what);
    LaunchArgs a{}; a.stream = c->stream; a.N = N;
    Stage s(c);
    a.Z = Z; a.t = t; a.dt = dt; a.J = J; a.Y = Y;
    if (memspace == RDHIP_HOST) {
        a.Z = s.in(Z, (size_t)N * (d.n + d.m) * ts); a.t = (const double*)s.in(t, (size_t)N * 8); a.dt = (const double*)s.in(dt, (size_t)N * 8);
        a.J = s.out(J, (size_t)N * d.n * (d.n + d.m) * ts); a.Y = s.out(Y, (size_t)N * d.n * ts);
        if (s.rc) return s.rc;
    }
    if (implicit && !a.Y && N > 0) { a.Y = s.reserve((size_t)N * d.n * ts); if (s.rc) return s.rc; }     // the Newton solution has to land somewhere
    if (ov && N > 0) {
        // overlap mode: this call goes to the stream the previous one did not use.  It is ordered behind (a) everything the context's stream held
        // when overlap work began or was last joined -- both streams start behind `main_mark` -- and (b) the previous call, on the other stream,
        // only if the two footprints intersect (the call before that ran on this very stream).
        rdhip_ctx::Footprint fp;
        auto add = [&](const void* p, size_t bytes) { if (p && bytes) { fp.r[fp.n].lo = (const char*)p; fp.r[fp.n].hi = (const char*)p + bytes; ++fp.n; } };
        add(Z, (size_t)N * (d.n + d.m) * ts); add(t, (size_t)N * 8); add(dt, (size_t)N * 8); add(J, (size_t)N * d.n * (d.n + d.m) * ts); add(Y, (size_t)N * d.n * ts);
        const unsigned me = c->flip & 1, other = me ^ 1;
        const int nread = 3;                               // r[0..2] are read, the rest written
        bool conflict = c->pend[other].size() >= 16;       // (bounded bookkeeping: a full list is cleared by an edge)
        for (const rdhip_ctx::Footprint& q : c->pend[other])
            for (int i = 0; i < fp.n && !conflict; ++i)
                for (int j = 0; j < q.n && !conflict; ++j) {
                    const bool both_read = i < nread && j < nread;
                    if (!both_read && fp.r[i].lo < q.r[j].hi && q.r[j].lo < fp.r[i].hi) conflict = true;
                }
        hipStream_t mine = me == 0 ? c->stream : c->side, theirs = me == 0 ? c->side : c->stream;
        if (me == 1 && !c->side_pending) {                 // the internal stream starts behind what the context's stream holds now
            HIP_TRY(c, hipEventRecord(c->main_mark, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->side, c->main_mark, 0));
            c->pend[0].clear();
            conflict = false;
        }
        if (conflict) {                                    // everything issued on the other stream so far comes first
            hipEvent_t ev = me == 0 ? c->side_done : c->main_done;
            HIP_TRY(c, hipEventRecord(ev, theirs));
            HIP_TRY(c, hipStreamWaitEvent(mine, ev, 0));
            c->pend[other].clear();
        }
        a.stream = mine;
        a.sched = take_sched(c, (int)me, (N + 63) / 64, dtype);
        // (implicit rule: each stream has its own set of flag words, so the two launches in flight do not share any)
        const int k = kernel_status(c, implicit ? family_implicit(c, d, 1, dtype, a, (int)me) : family_jacobian(d, rule, diff, dtype, a), what);
        c->pend[me].push_back(fp);
        if (me == 1) c->side_pending = true;
        ++c->flip;
        return k;
    }
    if (N > 0) a.sched = take_sched(c, 0, (N + 63) / 64, dtype);
    const int k = kernel_status(c, implicit ? family_implicit(c, d, 1, dtype, a) : family_jacobian(d, rule, diff, dtype, a), what);
    return memspace == RDHIP_HOST ? s.finish(k) : k;
}

// jacobian! over several independent packed trajectories in one launch.  TrajectoryOptimization-sized problems hold 101 ... 4 096 knot points
// per trajectory (examples/cartpole.jl:46): one such batch is launch-bound (9 MB of output takes 1.2 us at 8 TB/s, a launch ~8 us); a thousand
// of them enumerated by one persistent tile loop are one bandwidth-regime launch.
static bool stream_is_capturing(rdhip_ctx* c) {
    if (c->capturing) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;          // a borrowed stream may be captured by its owner (torch.cuda.graph)
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
// device table of `nb` batches in the next ring slot; returns the slot (its event is recorded by the caller behind the kernel) or nullptr
static rdhip_ctx::TableSlot* upload_table(rdhip_ctx* c, const std::vector<MultiRecPod>& recs, const std::vector<unsigned>& tile0) {
    const size_t rec_bytes = recs.size() * sizeof(MultiRecPod), t0_bytes = tile0.size() * sizeof(unsigned), bytes = rec_bytes + t0_bytes;
    for (auto& sl : c->tslot)                              // the same table as an earlier call (same arrays, same lengths): its device copy is still valid
        if (sl.used == bytes && sl.host && std::memcmp(sl.host, recs.data(), rec_bytes) == 0 && std::memcmp(static_cast<char*>(sl.host) + rec_bytes, tile0.data(), t0_bytes) == 0) { sl.hit = true; return &sl; }
    rdhip_ctx::TableSlot& sl = c->tslot[c->tnext++ % rdhip_ctx::NTSLOT];
    sl.hit = false;
    // launches that found their table in the slot recorded no event (an event per launch is a barrier packet between back-to-back kernels: the next
    // launch's prologue no longer runs under this one's drain, ~3 us of a 28 us call); before such a slot is overwritten, everything queued so far ends
    if (sl.unrecorded) {
        if (hipEventRecord(sl.done, c->stream) != hipSuccess || hipEventSynchronize(sl.done) != hipSuccess) { fail(c, RDHIP_ERR_HIP, "batch table: waiting for the stream failed"); return nullptr; }
        sl.unrecorded = false; sl.pending = false;
    }
    if (sl.pending) { if (hipEventSynchronize(sl.done) != hipSuccess) { fail(c, RDHIP_ERR_HIP, "batch table: hipEventSynchronize failed"); return nullptr; } sl.pending = false; }
    sl.used = 0;
    if (sl.cap < bytes) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        sl.host = sl.dev = nullptr; sl.cap = 0;
        const size_t cap = bytes + bytes / 2;
        if (hipHostMalloc(&sl.host, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&sl.dev, cap) != hipSuccess) { fail(c, RDHIP_ERR_HIP, "batch table: allocation of %zu bytes failed", cap); return nullptr; }
        sl.cap = cap;
    }
    if (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) { fail(c, RDHIP_ERR_HIP, "batch table: hipEventCreate failed"); return nullptr; }
    std::memcpy(sl.host, recs.data(), rec_bytes);
    std::memcpy(static_cast<char*>(sl.host) + rec_bytes, tile0.data(), t0_bytes);
    if (hipMemcpyAsync(sl.dev, sl.host, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { fail(c, RDHIP_ERR_HIP, "batch table: H2D copy failed"); return nullptr; }
    sl.used = bytes;
    return &sl;
}
// ---- rdhip_jacobian_multi for (model, rule, diff) combinations without a batch-table kernel: gather, ONE rdhip_jacobian, scatter --------------------------
// Round 6 ships the one-launch multi kernels for ForwardAD only (every shipped instantiation is exercised); FiniteDifference, UserDefined,
// the implicit rule and custom models ran one rdhip_jacobian call per batch -- 8.8 ms for 1 024 batches of 101 knots.  Knot points are independent, so the
// batches' inputs are copied into ONE contiguous stack (a model-independent copy kernel over a table of the batches), the single-batch kernel runs over
// it, and a second copy kernel hands the Jacobians out: three launches, the Jacobian stack moved twice more.
struct BatchCopyRec { const void* z; const double* t; const double* dt; void* j; void* y; long long off; long long n; };
// words: 4-byte units per knot point of the array handled (z: nz T / 4, t and dt: 2, J: n nz T / 4, y: n T / 4); 16-byte units where both sides allow it
__device__ inline void copy_words(unsigned* __restrict__ dst, const unsigned* __restrict__ src, long long nwords) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if ((((unsigned long long)dst | (unsigned long long)src) & 15ULL) == 0 && (nwords & 3) == 0) {
        typedef unsigned V4 __attribute__((ext_vector_type(4)));
        for (long long i = i0; i < nwords / 4; i += step) reinterpret_cast<V4*>(dst)[i] = reinterpret_cast<const V4*>(src)[i];
    } else if ((((unsigned long long)dst | (unsigned long long)src) & 7ULL) == 0 && (nwords & 1) == 0) {
        for (long long i = i0; i < nwords / 2; i += step) reinterpret_cast<unsigned long long*>(dst)[i] = reinterpret_cast<const unsigned long long*>(src)[i];
    } else {
        for (long long i = i0; i < nwords; i += step) dst[i] = src[i];
    }
}
__global__ void __launch_bounds__(256) batch_gather_kernel(const BatchCopyRec* __restrict__ recs, int zw, unsigned* Zs, unsigned* Ts, unsigned* Ds) {
    const BatchCopyRec r = recs[blockIdx.y];
    copy_words(Zs + r.off * zw, static_cast<const unsigned*>(r.z), r.n * zw);
    copy_words(Ts + r.off * 2, reinterpret_cast<const unsigned*>(r.t), r.n * 2);
    copy_words(Ds + r.off * 2, reinterpret_cast<const unsigned*>(r.dt), r.n * 2);
}
__global__ void __launch_bounds__(256) batch_scatter_kernel(const BatchCopyRec* __restrict__ recs, int jw, int yw, const unsigned* Js, const unsigned* Ys) {
    const BatchCopyRec r = recs[blockIdx.y];
    copy_words(static_cast<unsigned*>(r.j), Js + r.off * jw, r.n * jw);
    if (r.y && Ys) copy_words(static_cast<unsigned*>(r.y), Ys + r.off * yw, r.n * yw);
}
// RDHIP_ERR_UNSUPPORTED: not this way (the caller runs the batches one by one); otherwise the status of the three launches
static int multi_staged(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, const std::vector<int>& live, const long long* N, const void* const* Z,
                        const double* const* t, const double* const* dt, void* const* J, void* const* Y, const char* what) {
    if (c->overlap || stream_is_capturing(c)) return RDHIP_ERR_UNSUPPORTED;      // (one staging area per context, reused in stream order; growing it allocates)
    const ModelDesc& d = M->d;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4, zb = (size_t)(d.n + d.m) * ts, jb = (size_t)d.n * (d.n + d.m) * ts, yb = (size_t)d.n * ts;
    long long total = 0, nmax = 0; bool any_y = false;
    std::vector<BatchCopyRec> recs; recs.reserve(live.size());
    for (int b : live) { recs.push_back({Z[b], t[b], dt[b], J[b], Y ? Y[b] : nullptr, total, N[b]}); total += N[b]; nmax = N[b] > nmax ? N[b] : nmax; any_y = any_y || (Y && Y[b]); }
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_rec = 0, o_z = up(recs.size() * sizeof(BatchCopyRec)), o_t = o_z + up((size_t)total * zb), o_d = o_t + up((size_t)total * 8), o_j = o_d + up((size_t)total * 8),
                 o_y = o_j + up((size_t)total * jb), need = o_y + (any_y ? up((size_t)total * yb) : 0);
    if (c->mws_bytes < need) {
        if (c->mws) (void)hipFree(c->mws);
        c->mws = nullptr; c->mws_bytes = 0;
        if (hipMalloc(&c->mws, need) != hipSuccess) { (void)hipGetLastError(); return RDHIP_ERR_UNSUPPORTED; }      // (no room for a second copy of the Jacobians: one by one)
        c->mws_bytes = need;
    }
    char* w = static_cast<char*>(c->mws);
    // (pageable source: the runtime stages it before the call returns, so the vector may go)
    if (hipMemcpyAsync(w + o_rec, recs.data(), recs.size() * sizeof(BatchCopyRec), hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(c, RDHIP_ERR_HIP, "%s: batch table copy failed", what);
    const unsigned gx = (unsigned)std::min<long long>(32, std::max<long long>(1, (nmax * (long long)(jb / 4) + 4095) / 4096));
    hipLaunchKernelGGL(batch_gather_kernel, dim3(gx, (unsigned)recs.size()), dim3(256), 0, c->stream, reinterpret_cast<const BatchCopyRec*>(w + o_rec), (int)(zb / 4),
                       reinterpret_cast<unsigned*>(w + o_z), reinterpret_cast<unsigned*>(w + o_t), reinterpret_cast<unsigned*>(w + o_d));
    if (hipGetLastError() != hipSuccess) return fail(c, RDHIP_ERR_HIP, "%s: gather launch failed", what);
    const int rc = rdhip_jacobian(c, M, rule, diff, dtype, total, w + o_z, reinterpret_cast<const double*>(w + o_t), reinterpret_cast<const double*>(w + o_d), w + o_j,
                                  any_y ? w + o_y : nullptr, RDHIP_DEVICE);
    if (rc) return rc;
    hipLaunchKernelGGL(batch_scatter_kernel, dim3(gx, (unsigned)recs.size()), dim3(256), 0, c->stream, reinterpret_cast<const BatchCopyRec*>(w + o_rec), (int)(jb / 4), (int)(yb / 4),
                       reinterpret_cast<const unsigned*>(w + o_j), any_y ? reinterpret_cast<const unsigned*>(w + o_y) : nullptr);
    if (hipGetLastError() != hipSuccess) return fail(c, RDHIP_ERR_HIP, "%s: scatter launch failed", what);
    c->last_multi_launches += 2 + ((rule == RDHIP_IMPLICIT_MIDPOINT) ? 3 : 1);
    return RDHIP_OK;
}
int rdhip_jacobian_multi(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, int nbatch, const long long* N,
                         const void* const* Z, const double* const* t, const double* const* dt, void* const* J, void* const* Y, int memspace) {
    const char* what = "rdhip_jacobian_multi";
    int rc = check_common(c, M, dtype, 0, memspace); if (rc) return rc;
    c->last_multi_launches = 0;
    if (nbatch < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: negative nbatch", what);
    if (nbatch == 0) return RDHIP_OK;
    if (!N || !Z || !t || !dt || !J) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL table", what);
    rc = rule_ok(c, M, rule, RDHIP_CONTINUOUS, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (diff < RDHIP_FORWARD_AD || diff > RDHIP_USER_DEFINED) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad diff method %d", what, diff);
    std::vector<int> live;
    bool aligned = true, equal = true;
    for (int b = 0; b < nbatch; ++b) {
        if (N[b] < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: batch %d has negative N", what, b);
        if (N[b] == 0) continue;
        if (!Z[b] || !t[b] || !dt[b] || !J[b]) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: batch %d has a NULL data pointer", what, b);
        aligned = aligned && !(reinterpret_cast<uintptr_t>(J[b]) & 15) && !(Y && (reinterpret_cast<uintptr_t>(Y[b]) & 15));
        if (!live.empty() && N[b] != N[live[0]]) equal = false;
        live.push_back(b);
    }
    rc = check_rotation(c, M, 0); if (rc) return rc;
    if (live.empty()) return RDHIP_OK;
    const ModelDesc& d = M->d;
    // fused: explicit rules on device pointers -- ForwardAD on aligned outputs for the compiled-in models (dispatch.hpp launch_jacobian_multi says
    // RDHIP_ERR_UNSUPPORTED for the rest before anything is launched), everything for custom models.  Implicit rules (a Newton pass, the Jacobian pass
    // and a gated exact pass per batch) and host arrays run the batches one rdhip_jacobian call at a time.
    bool fused = memspace == RDHIP_DEVICE && rule != RDHIP_IMPLICIT_MIDPOINT;
    if (fused) {
        const int nl = (int)live.size();
        const bool table = nl > MULTI_INLINE && !stream_is_capturing(c);      // (a capturing stream: groups of MULTI_INLINE in the kernel arguments -- no allocation, no event)
        const int group = table ? nl : MULTI_INLINE;
        for (int g0 = 0; fused && g0 < nl; g0 += group) {
            const int ng = nl - g0 < group ? nl - g0 : group;
            MultiDesc md{}; MultiPod& p = md.pod;
            md.aligned16 = aligned; p.nb = ng; p.uniform = 0; p.ext = nullptr; p.ext_tile0 = nullptr;
            rdhip_ctx::TableSlot* slot = nullptr;
            long long acc = 0;
            for (int i = 0; i < ng; ++i) acc += (N[live[g0 + i]] + 63) / 64;
            if (acc >= (1LL << 31)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: more than 2^31 tiles of 64 knot points in one call", what);
            acc = 0;
            p.group = 1; p.pad_ = 0;
            if (table) {
                std::vector<MultiRecPod> recs(ng); std::vector<unsigned> tile0(ng + 1);
                for (int i = 0; i < ng; ++i) {
                    const int b = live[g0 + i];
                    recs[i] = MultiRecPod{Z[b], t[b], dt[b], J[b], Y ? Y[b] : nullptr, N[b], {0, 0}};
                    tile0[i] = (unsigned)acc; acc += (N[b] + 63) / 64;
                }
                tile0[ng] = (unsigned)acc;
                slot = upload_table(c, recs, tile0);
                if (!slot) return RDHIP_ERR_HIP;
                p.ext = static_cast<const MultiRecPod*>(slot->dev);
                p.ext_tile0 = reinterpret_cast<const unsigned*>(static_cast<const char*>(slot->dev) + recs.size() * sizeof(MultiRecPod));
                p.uniform = equal ? (int)((N[live[0]] + 63) / 64) : 0;
                p.group = (ng + MULTI_INLINE - 1) / MULTI_INLINE;
                for (int i = 0; i <= MULTI_INLINE; ++i) p.tile0[i] = (long long)i * p.group < ng ? tile0[(size_t)i * p.group] : (unsigned)acc;     // coarse level
                for (int i = 0; i < MULTI_INLINE; ++i) { p.N[i] = 0; p.Z[i] = nullptr; p.t[i] = nullptr; p.dt[i] = nullptr; p.J[i] = nullptr; p.Y[i] = nullptr; }
            } else {
                for (int i = 0; i < MULTI_INLINE; ++i) {
                    const bool on = i < ng; const int b = on ? live[g0 + i] : 0;
                    p.tile0[i] = (unsigned)acc;
                    p.N[i] = on ? N[b] : 0; p.Z[i] = on ? Z[b] : nullptr; p.t[i] = on ? t[b] : nullptr; p.dt[i] = on ? dt[b] : nullptr;
                    p.J[i] = on ? J[b] : nullptr; p.Y[i] = (on && Y) ? Y[b] : nullptr;
                    if (on) acc += (N[b] + 63) / 64;
                }
                p.tile0[MULTI_INLINE] = (unsigned)acc;
                for (int i = ng; i <= MULTI_INLINE; ++i) p.tile0[i] = (unsigned)acc;
            }
            p.total = acc;
            LaunchArgs a{}; a.stream = c->stream; a.multi = &md;
            // dynamic tile schedule (kernels.hpp jac_split_loop) only for LONG trajectories: it gives up the resolved-tile ring, so every tile's descriptor is
            // looked up in front of the compute waves' input prefetch -- a scalar-cache hit while consecutive tiles share a batch record, a memory round trip
            // per tile when batches are one or two tiles long (1 024 x 101 knots: 40.7 -> 47.8 us with the counter, 4 096 x 101: 151 -> 166; 256 x 4 096:
            // 374-387 -> 353 us)
            a.sched = (acc >= 16LL * ng) ? take_sched(c, 0, acc, dtype) : nullptr;
            const int k = family_jacobian(d, rule, diff, dtype, a);
            if (k == RDHIP_ERR_UNSUPPORTED && g0 == 0) { fused = false; break; }      // no fused kernel for this (model, rule, diff): decided before anything was launched
            if (k) return kernel_status(c, k, what);
            ++c->last_multi_launches;
            if (slot) {
                if (slot->hit) slot->unrecorded = true;            // the caller's arrays of an earlier call again (a solver's loop): no event, see upload_table
                else { if (hipEventRecord(slot->done, c->stream) != hipSuccess) return fail(c, RDHIP_ERR_HIP, "%s: hipEventRecord failed", what); slot->pending = true; }
            }
        }
    }
    if (fused) return RDHIP_OK;
    static const bool verbose = getenv("RDHIP_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[rdhip] %s: no fused kernel for (model kind %d, rule %d, diff %d, %s pointers): %d batches run as separate rdhip_jacobian calls\n",
                         what, d.kind, rule, diff, memspace == RDHIP_DEVICE ? "device" : "host", (int)live.size());
    // Same results, one call per RUN of batches: knot points are independent, so batches that follow each other in memory in every array (the slices of
    // one stack -- what a caller holds who keeps its trajectories in one allocation) are one rdhip_jacobian call over their concatenation.  1 024 x 101
    // satellite knots under FiniteDifference: 8.8 ms as 1 024 calls, one launch when the slices are adjacent (tools/bench_configs.py --sweep).
    const size_t tsz = dtype == RDHIP_F64 ? 8 : 4;
    const size_t zb = (size_t)(d.n + d.m) * tsz, jb = (size_t)d.n * (d.n + d.m) * tsz, yb = (size_t)d.n * tsz;
    auto follows = [&](int a, int b) {          // batch b starts where batch a ends, in every array
        const size_t na = (size_t)N[a];
        return (const char*)Z[b] == (const char*)Z[a] + na * zb && t[b] == t[a] + na && dt[b] == dt[a] + na && (char*)J[b] == (char*)J[a] + na * jb &&
               ((Y == nullptr) || (!Y[a] && !Y[b]) || (Y[a] && Y[b] && (char*)Y[b] == (char*)Y[a] + na * yb));
    };
    size_t nruns = live.empty() ? 0 : 1;
    for (size_t i = 1; i < live.size(); ++i) nruns += follows(live[i - 1], live[i]) ? 0 : 1;
    if (nruns >= 4 && memspace == RDHIP_DEVICE && nbatch <= 65535) {          // (a handful of calls is cheaper than two more passes over the Jacobians)
        rc = multi_staged(c, M, rule, diff, dtype, live, N, Z, t, dt, J, Y, what);
        if (rc != RDHIP_ERR_UNSUPPORTED) return rc;
    }
    for (size_t i = 0; i < live.size();) {
        size_t e = i + 1; long long n = N[live[i]];
        while (e < live.size() && follows(live[e - 1], live[e])) { n += N[live[e]]; ++e; }
        const int b = live[i];
        rc = rdhip_jacobian(c, M, rule, diff, dtype, n, Z[b], t[b], dt[b], J[b], Y ? Y[b] : nullptr, memspace);
        if (rc) return rc;
        c->last_multi_launches += (rule == RDHIP_IMPLICIT_MIDPOINT) ? 3 : 1;
        i = e;
    }
    return RDHIP_OK;
}
int rdhip_ctx_last_multi_launches(rdhip_ctx* c, int* launches) {
    if (!c || !launches) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *launches = c->last_multi_launches; return RDHIP_OK;
}

// dynamics_error / dynamics_error_jacobian!: the discrete dynamics (Jacobian) of the first N-1 knot points, minus the next state
static int dynerr_common(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, long long N, const void* Z, const double* t, const double* dt,
                         void* J2, void* J1, void* E, bool want_jac, int memspace, const char* what) {
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (rule == RDHIP_CONTINUOUS) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: a quadrature rule is required (got %d)", what, rule);
    rc = rule_ok(c, M, rule, RDHIP_EULER, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (want_jac && (diff < RDHIP_FORWARD_AD || diff > RDHIP_USER_DEFINED)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad diff method %d", what, diff);
    const long long P = N - 1;
    if (P <= 0) return RDHIP_OK;
    if (!Z || !t || !dt || (!want_jac && !E)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d; const int nz = d.n + d.m;
    Stage s(c);
    const void* dZ = Z; const double* dtm = t; const double* ddt = dt; void* dJ2 = J2; void* dJ1 = J1; void* dE = E;
    if (memspace == RDHIP_HOST) {
        dZ = s.in(Z, (size_t)N * nz * ts); dtm = (const double*)s.in(t, (size_t)N * 8); ddt = (const double*)s.in(dt, (size_t)N * 8);
        dJ2 = s.out(J2, (size_t)P * d.n * nz * ts); dJ1 = s.out(J1, (size_t)P * d.n * nz * ts); dE = s.out(E, (size_t)P * d.n * ts);
        if (s.rc) return s.rc;
    }
    LaunchArgs a{}; a.stream = c->stream; a.N = P; a.Z = dZ; a.t = dtm; a.dt = ddt;
    int k = RDHIP_OK;
    if (rule == RDHIP_IMPLICIT_MIDPOINT) {          // e = x1 + h f((x1+x2)/2, u1) - x2 ; J1 = [I + (h/2)A, hB] ; J2 = [(h/2)A - I, 0]  (src/integration.jl:631-690)
        if (want_jac && diff != RDHIP_FORWARD_AD) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: ForwardAD only for implicit rules", what);
        a.Y = dE; a.J = want_jac ? dJ1 : nullptr; a.J2 = want_jac ? dJ2 : nullptr;
        k = kernel_status(c, family_implicit(c, d, 2, dtype, a), what);
        return memspace == RDHIP_HOST ? s.finish(k) : k;
    }
    if (want_jac && dJ1) { a.J = dJ1; a.Y = dE; k = family_jacobian(d, rule, diff, dtype, a); }       // f(z_k) lands in E
    else if (dE) { a.Y = dE; k = family_eval(d, rule, dtype, a); }
    if (k == RDHIP_OK && dE) k = launch_sub_next_state(dtype, c->stream, P, d.n, nz, dZ, dE);
    if (k == RDHIP_OK && want_jac && dJ2) k = launch_fill_neg_identity(dtype, c->stream, P, d.n, nz, dJ2);
    k = kernel_status(c, k, what);
    return memspace == RDHIP_HOST ? s.finish(k) : k;
}
int rdhip_dynamics_error(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long N, const void* Z, const double* t, const double* dt, void* E, int memspace) {
    return dynerr_common(c, M, rule, RDHIP_FORWARD_AD, dtype, N, Z, t, dt, nullptr, nullptr, E, false, memspace, "rdhip_dynamics_error");
}
int rdhip_dynamics_error_jacobian(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, long long N, const void* Z, const double* t, const double* dt,
                                  void* J2, void* J1, void* E, int memspace) {
    return dynerr_common(c, M, rule, diff, dtype, N, Z, t, dt, J2, J1, E, true, memspace, "rdhip_dynamics_error_jacobian");
}

int rdhip_error_expansion(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long N, const void* Z, const double* t, const double* dt,
                          void* Ebar, void* Y, int memspace) {
    const char* what = "rdhip_error_expansion";
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (rule == RDHIP_CONTINUOUS || rule == RDHIP_IMPLICIT_MIDPOINT) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: an explicit quadrature rule is required (got %d)", what, rule);
    rc = rule_ok(c, M, rule, RDHIP_EULER, RDHIP_RK4, what); if (rc) return rc;
    const long long P = N - 1;
    if (P <= 0) return RDHIP_OK;
    if (!Z || !t || !dt || !Ebar) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const ModelDesc& d = M->d;
    const bool quat = d.lie_nP == 0 || d.rot == RDHIP_ROT_QUAT;       // MRP / RodriguesParam states: Jacobian stack + the generic projection (3 x 3 blocks)
    const bool rigid = quat && (d.kind == RDHIP_MODEL_SATELLITE || d.kind == RDHIP_MODEL_QUADROTOR);
    const bool lie_fused = quat && (d.kind == RDHIP_MODEL_SATELLITE_PAIR || d.kind == RDHIP_MODEL_LIE_SATELLITE);   // compiled-in LieGroupModels: fused kernel too
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const int nz = d.n + d.m;
    Stage s(c);
    LaunchArgs a{}; a.stream = c->stream; a.N = P; a.Z = Z; a.t = t; a.dt = dt; a.J = Ebar; a.Y = Y;
    if (memspace == RDHIP_HOST) {
        a.Z = s.in(Z, (size_t)N * nz * ts); a.t = (const double*)s.in(t, (size_t)N * 8); a.dt = (const double*)s.in(dt, (size_t)N * 8);
        a.J = s.out(Ebar, (size_t)P * d.e * (d.e + d.m) * ts); a.Y = s.out(Y, (size_t)P * d.n * ts);
        if (s.rc) return s.rc;
    }
    int k;
    if (P > 0) a.sched = take_sched(c, 0, (P + 63) / 64, dtype);
    if (d.lie_nP == 0) k = family_jacobian(d, rule, RDHIP_FORWARD_AD, dtype, a);   // EuclideanState: G = I
    else if (lie_fused && !(reinterpret_cast<uintptr_t>(a.J) & 15) && !(reinterpret_cast<uintptr_t>(a.Y) & 15)) k = fam_lie_errexp(d, rule, dtype, a);
    else if (quat && d.kind == RDHIP_MODEL_CUSTOM && !(reinterpret_cast<uintptr_t>(a.J) & 15) && !(reinterpret_cast<uintptr_t>(a.Y) & 15)) {
        k = jit_errexp(d.jit, rule, dtype, a); if (k) g_jit_model = &d;          // custom LieGroupModels: the same fused kernel, JIT-compiled
    }
    else if (!rigid) {
        // unaligned outputs: Jacobians into a context workspace, then the generic projection kernel
        void* Jw = s.reserve((size_t)P * d.n * nz * ts);
        if (s.rc) return s.rc;
        void* Eout = a.J;
        a.J = Jw;
        k = family_jacobian(d, rule, RDHIP_FORWARD_AD, dtype, a);
        if (k == RDHIP_OK) k = launch_project_errexp(M->lie, dtype, c->stream, P, d.m, a.Z, Jw, Eout);
    } else {
        if ((reinterpret_cast<uintptr_t>(a.J) & 15) || (reinterpret_cast<uintptr_t>(a.Y) & 15))
            return fail(c, RDHIP_ERR_INVALID_ARG, "%s: Ebar and Y must be 16-byte aligned", what);
        k = d.kind == RDHIP_MODEL_SATELLITE ? fam_satellite_errexp(d, rule, dtype, a) : fam_quadrotor_errexp(d, rule, dtype, a);
    }
    k = kernel_status(c, k, what);
    return memspace == RDHIP_HOST ? s.finish(k) : k;
}

int rdhip_errstate_jacobian(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* X, int ldx, int clear, void* G, int memspace) {
    const char* what = "rdhip_errstate_jacobian";
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (N > 0 && (!X || !G)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    if (ldx < M->d.n) return fail(c, RDHIP_ERR_DIM_MISMATCH, "%s: ldx=%d < state_dim=%d", what, ldx, M->d.n);
    rc = check_rotation(c, M, 1); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    if (memspace == RDHIP_DEVICE) return kernel_status(c, launch_errstate_jacobian(M->lie, dtype, c->stream, N, X, ldx, clear, G), what);
    Stage s(c);
    const void* dX = s.in(X, (size_t)N * ldx * ts);
    void* dG = s.out(G, (size_t)N * d.n * d.e * ts, /*preload=*/!clear);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, launch_errstate_jacobian(M->lie, dtype, c->stream, N, dX, ldx, clear, dG), what));
}

int rdhip_state_diff_map(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* X, int ldx, const void* X0, int ldx0, int errmap, void* dX, int memspace) {
    const char* what = "rdhip_state_diff";
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (N > 0 && (!X || !X0 || !dX)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    if (ldx < M->d.n || ldx0 < M->d.n) return fail(c, RDHIP_ERR_DIM_MISMATCH, "%s: leading dimension < state_dim=%d", what, M->d.n);
    if (errmap < RDHIP_ERRMAP_CAYLEY || errmap > RDHIP_ERRMAP_IDENTITY) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad error map %d", what, errmap);
    // IdentityMap of a quaternion is its four parameters; Rotations.rotation_error wraps the result in a three-element RotationError and throws
    if (errmap == RDHIP_ERRMAP_IDENTITY && M->lie.nrot > 0 && M->lie.rot == RDHIP_ROT_QUAT)
        return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: IdentityMap is defined for three-parameter rotations (MRP, RodriguesParam) only", what);
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;      // state_diff serves QuatRotation, MRP and RodriguesParam
    if (memspace == RDHIP_DEVICE) return kernel_status(c, launch_state_diff(M->lie, dtype, c->stream, N, X, ldx, X0, ldx0, errmap, dX), what);
    Stage s(c);
    const void* a = s.in(X, (size_t)N * ldx * ts); const void* b = s.in(X0, (size_t)N * ldx0 * ts);
    void* o = s.out(dX, (size_t)N * d.e * ts);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, launch_state_diff(M->lie, dtype, c->stream, N, a, ldx, b, ldx0, errmap, o), what));
}
int rdhip_state_diff(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* X, int ldx, const void* X0, int ldx0, void* dX, int memspace) {
    return rdhip_state_diff_map(c, M, dtype, N, X, ldx, X0, ldx0, RDHIP_ERRMAP_CAYLEY, dX, memspace);
}

int rdhip_nabla_errstate_jacobian(rdhip_ctx* c, const rdhip_model* M, int dtype, long long N, const void* X, int ldx, const void* Xb, int ldxb,
                                  int clear, void* H, int memspace) {
    const char* what = "rdhip_nabla_errstate_jacobian";
    int rc = check_common(c, M, dtype, N, memspace); if (rc) return rc;
    if (N > 0 && (!X || !Xb || !H)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    if (ldx < M->d.n || ldxb < M->d.n) return fail(c, RDHIP_ERR_DIM_MISMATCH, "%s: leading dimension < state_dim=%d", what, M->d.n);
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    if (memspace == RDHIP_DEVICE) return kernel_status(c, launch_nabla_errstate(M->lie, dtype, c->stream, N, X, ldx, Xb, ldxb, clear, H), what);
    Stage s(c);
    const void* a = s.in(X, (size_t)N * ldx * ts); const void* b = s.in(Xb, (size_t)N * ldxb * ts);
    void* o = s.out(H, (size_t)N * d.e * d.e * ts, /*preload=*/!clear);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, launch_nabla_errstate(M->lie, dtype, c->stream, N, a, ldx, b, ldxb, clear, o), what));
}

int rdhip_rollout_layout(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long K, long long Mt, int layout, void* Z, const double* t, const double* dt,
                         const void* X0, int memspace) {
    const char* what = "rdhip_rollout";
    int rc = check_common(c, M, dtype, Mt, memspace); if (rc) return rc;
    if (K < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: negative K", what);
    rc = rule_ok(c, M, rule, RDHIP_EULER, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (layout != RDHIP_TRAJECTORY_MAJOR && layout != RDHIP_TIME_MAJOR) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad layout %d", what, layout);
    if (K * Mt > 0 && (!Z || !t || !dt)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    LaunchArgs a{}; a.stream = c->stream; a.N = Mt; a.K = K; a.time_major = layout == RDHIP_TIME_MAJOR;
    const bool implicit = rule == RDHIP_IMPLICIT_MIDPOINT;
    if (implicit && a.time_major) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: implicit rules take trajectory-major stacks", what);
    if (memspace == RDHIP_DEVICE) { a.Z = Z; a.t = t; a.dt = dt; a.X0 = X0; return kernel_status(c, implicit ? family_implicit(c, d, 3, dtype, a) : family_rollout(d, rule, dtype, a), what); }
    Stage s(c);
    a.Z = s.out(Z, (size_t)K * Mt * (d.n + d.m) * ts, /*preload=*/true);
    a.t = (const double*)s.in(t, (size_t)K * Mt * 8); a.dt = (const double*)s.in(dt, (size_t)K * Mt * 8);
    a.X0 = s.in(X0, (size_t)Mt * d.n * ts);
    if (s.rc) return s.rc;
    return s.finish(kernel_status(c, implicit ? family_implicit(c, d, 3, dtype, a) : family_rollout(d, rule, dtype, a), what));
}
// rollout! on split arrays (rollout_split.hpp): the controls are read once, the states written once, in whole lines
int rdhip_rollout_split(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long K, long long Mt, int layout, void* X, const void* U,
                        const double* t, const double* dt, const void* X0, int memspace) {
    const char* what = "rdhip_rollout_split";
    int rc = check_common(c, M, dtype, Mt, memspace); if (rc) return rc;
    if (K < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: negative K", what);
    rc = rule_ok(c, M, rule, RDHIP_EULER, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (layout != RDHIP_TRAJECTORY_MAJOR && layout != RDHIP_TIME_MAJOR) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad layout %d", what, layout);
    if (rule == RDHIP_IMPLICIT_MIDPOINT) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: explicit rules only (an implicit rollout is a latency chain of Newton solves: rdhip_rollout)", what);
    if (M->d.kind == RDHIP_MODEL_CUSTOM) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: compiled-in models only (custom models: rdhip_rollout)", what);
    if (K * Mt == 0) return RDHIP_OK;
    if (!X || !t || !dt || (K > 1 && !U)) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d;
    const size_t xb = (size_t)K * Mt * d.n * ts, ub = (size_t)(K - 1) * Mt * d.m * ts;
    LaunchArgs a{}; a.stream = c->stream; a.N = Mt; a.K = K; a.time_major = layout == RDHIP_TIME_MAJOR;
    Stage s(c);
    a.Z = X; a.U = U; a.t = t; a.dt = dt; a.X0 = X0;
    if (memspace == RDHIP_HOST) {
        a.Z = s.out(X, xb, /*preload=*/X0 == nullptr); a.U = s.in(U, ub); a.t = (const double*)s.in(t, (size_t)K * Mt * 8); a.dt = (const double*)s.in(dt, (size_t)K * Mt * 8);
        a.X0 = s.in(X0, (size_t)Mt * d.n * ts);
        if (s.rc) return s.rc;
    }
    int k = RDHIP_OK;
    if (K == 1) {                                                      // nothing to simulate: knot 0 is the initial state (both layouts hold it at the same place)
        if (a.X0 && hipMemcpyAsync(const_cast<void*>(a.Z), a.X0, (size_t)Mt * d.n * ts, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) k = fail(c, RDHIP_ERR_HIP, "%s: copying the initial states failed", what);
    } else k = kernel_status(c, family_rollout(d, rule, dtype, a), what);
    return memspace == RDHIP_HOST ? s.finish(k) : k;
}
int rdhip_rollout(rdhip_ctx* c, const rdhip_model* M, int rule, int dtype, long long K, long long Mt, void* Z, const double* t, const double* dt,
                  const void* X0, int memspace) {
    return rdhip_rollout_layout(c, M, rule, dtype, K, Mt, RDHIP_TRAJECTORY_MAJOR, Z, t, dt, X0, memspace);
}

// rollout!, then the Jacobians along the rollout: what a shooting / iLQR forward pass + linearisation needs, in one call.
// Two launches back to back on the context's stream; the Jacobian kernel finds the simulated states in L2 / Infinity Cache.
// (Issuing the rollout in pieces and computing the Jacobians of finished pieces on a second stream was measured and dropped: the
// rollout's latency chain slows down by as much as it overlaps once the Jacobian kernel saturates the memory system)
int rdhip_rollout_jacobian(rdhip_ctx* c, const rdhip_model* M, int rule, int diff, int dtype, long long K, long long Mt, int layout,
                           void* Z, const double* t, const double* dt, const void* X0, void* J, int memspace) {
    const char* what = "rdhip_rollout_jacobian";
    int rc = check_common(c, M, dtype, Mt, memspace); if (rc) return rc;
    if (K < 0) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: negative K", what);
    rc = rule_ok(c, M, rule, RDHIP_EULER, RDHIP_IMPLICIT_MIDPOINT, what); if (rc) return rc;
    if (diff < RDHIP_FORWARD_AD || diff > RDHIP_USER_DEFINED) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad diff method %d", what, diff);
    if (layout != RDHIP_TRAJECTORY_MAJOR && layout != RDHIP_TIME_MAJOR) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: bad layout %d", what, layout);
    const long long N = K * Mt;
    if (N == 0) return RDHIP_OK;
    if (!Z || !t || !dt || !J) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL data pointer", what);
    rc = check_rotation(c, M, 0); if (rc) return rc;
    const bool implicit = rule == RDHIP_IMPLICIT_MIDPOINT;
    if (implicit && diff != RDHIP_FORWARD_AD) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: implicit rules take ImplicitFunctionTheorem(ForwardAD()) only", what);
    if (implicit && layout == RDHIP_TIME_MAJOR) return fail(c, RDHIP_ERR_UNSUPPORTED, "%s: implicit rules take trajectory-major data only", what);
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const ModelDesc& d = M->d; const int nz = d.n + d.m;
    Stage s(c);
    LaunchArgs r{}; r.stream = c->stream; r.N = Mt; r.K = K; r.Z = Z; r.t = t; r.dt = dt; r.X0 = X0; r.time_major = (layout == RDHIP_TIME_MAJOR);
    void* dJ = J;
    if (memspace == RDHIP_HOST) {
        r.Z = s.out(Z, (size_t)N * nz * ts, /*preload=*/true); r.t = (const double*)s.in(t, (size_t)N * 8); r.dt = (const double*)s.in(dt, (size_t)N * 8);
        r.X0 = s.in(X0, (size_t)Mt * d.n * ts); dJ = s.out(J, (size_t)N * d.n * nz * ts);
        if (s.rc) return s.rc;
    }
    LaunchArgs j{}; j.stream = c->stream; j.N = N; j.Z = r.Z; j.t = r.t; j.dt = r.dt; j.J = dJ;
    if (implicit) { j.Y = s.reserve((size_t)N * d.n * ts); if (s.rc) return s.rc; }      // the Newton solutions have to land somewhere
    c->last_rolljac_launches = 0;
    LaunchArgs f = r; f.J = dJ; f.progress = c->d_progress; f.progress_words = PROGRESS_WORDS; f.probe = 1; f.beside = c->fused_beside ? 1 : 0;
    // A TRAJECTORY-major stack of bandwidth size goes the two-launch way: its fused form gathers tiles of 4 trajectories x 16 knots (ragged when K is not a
    // multiple of 16, short runs of output) and was measured SLOWER than rollout + Jacobians in two launches from a few hundred thousand knot points
    // on (4 096 x 101 fp64: 252 vs 221 us; 16 384 x 64: 365 vs 365; the latency-sized 512 x 256 fp32: 138 vs 148, so small stacks stay fused); the
    // Jacobian launch over the contiguous stack then also gets the dynamic tile schedule.  Time-major stacks are faster fused at every size measured.
    const long long tj_max = tune_knobs().fused_tj_max_knots;
    const bool tj_two_launches = layout != RDHIP_TIME_MAJOR && N >= tj_max && dtype == RDHIP_F64;      // (fp32: the gathered tiles win at every size measured -- 4 096 x 256: 254 against 288 us, 2xx at two workgroups per CU)
    if (!implicit && c->allow_fused && !tj_two_launches && family_rolljac(d, rule, diff, dtype, f) == RDHIP_OK) {      // (probe: nothing is launched)
        // ONE launch: rollout workgroups + Jacobian workgroups that consume finished chunks through progress words (fused.hpp)
        f.probe = 0;
        const bool capturing = stream_is_capturing(c);
        // a captured launch is replayed with the epoch it was recorded with: its words are cleared by a memset node in front of it instead
        if (capturing) HIP_TRY(c, hipMemsetAsync(c->d_progress, 0, (size_t)PROGRESS_WORDS * sizeof(unsigned long long), c->stream));
        c->roll_epoch = (c->roll_epoch + 1) & 0x7fffffffULL; if (c->roll_epoch == 0) c->roll_epoch = 1;
        f.pbase = c->roll_epoch << 32;
        const int kf = fused_launch(c, d, rule, diff, dtype, f, capturing);
        if (kf != RDHIP_ERR_UNSUPPORTED) { c->last_rolljac_launches = 1; const int k = kernel_status(c, kf, what); return memspace == RDHIP_HOST ? s.finish(k) : k; }
    }
    // Models without the fused launch (small models, custom models, the models of a mixed-model call that did not get it): a TIME-major stack of
    // bandwidth size is rolled out in a few pieces of consecutive knots, and the Jacobians of a finished piece -- one contiguous range of the stack --
    // run on a second stream while the next piece is rolled out.  Nothing spins: every launch waits for nobody, so any number of these may be in
    // flight beside a fused launch.  A piece starts from the last knot of the one before it (read back from Z, the same bits the one-piece rollout
    // carries in registers), so states and Jacobians are those of the two plain launches.  BASELINE config 5's cartpole half (4 096 x 256 fp32):
    // rollout 124 us + Jacobians 19 us one after the other; inside the mixed-model call the Jacobians started at 173 us and took 98 us beside
    // the satellite's fused launch -- they were the end of the call.
    const long long pipe_min = tune_knobs().rolljac_pipe_min_knots;     // tuning switch (0 = never)
    constexpr long long PIECES = 4;
    // ... and only while the rollout is a LATENCY chain that leaves most of the chip idle (one workgroup per 64 trajectories on at most half of the
    // compute units): a batch of 16 384 trajectories fills the chip with its rollout alone, its Jacobians have nothing to run on beside it, and the
    // pieces only add launches -- 16 384 x 64 fp64 satellite knots (no fused launch: 256 groups) 489 us in pieces, 390 us as the two plain launches
    const bool chain = 2 * ((Mt + 63) / 64) <= (long long)dev_info().cus;
    if (!implicit && layout == RDHIP_TIME_MAJOR && pipe_min > 0 && N >= pipe_min && K >= 16 * PIECES && chain && !stream_is_capturing(c)) {
        HIP_TRY(c, hipSetDevice(c->device));
        if (!c->pipe) HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe, hipStreamNonBlocking));
        if (!c->pipe_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->pipe_ev, hipEventDisableTiming));
        if (!c->pipe_done) HIP_TRY(c, hipEventCreateWithFlags(&c->pipe_done, hipEventDisableTiming));
        const long long len = (K + PIECES - 1) / PIECES;
        const size_t zb = (size_t)Mt * nz * ts, jb = (size_t)Mt * d.n * nz * ts;      // bytes of one knot of the whole stack
        int k = RDHIP_OK, launches = 0;
        for (long long k0 = 0; k0 < K && !k; k0 += len) {
            const long long nk = K - k0 < len ? K - k0 : len;                  // knots k0 .. k0 + nk - 1 get their Jacobians from this piece
            const bool last = k0 + nk >= K;
            LaunchArgs rc = r;
            rc.Z = (const char*)r.Z + (size_t)k0 * zb; rc.t = r.t + k0 * Mt; rc.dt = r.dt + k0 * Mt;
            rc.K = last ? nk : nk + 1;                                         // ... and the piece simulates up to the first knot of the next one
            if (k0 > 0) rc.X0 = nullptr;
            if (rc.K >= 2 || k0 == 0) { k = kernel_status(c, family_rollout(d, rule, dtype, rc), what); ++launches; }
            if (k) break;
            // what the Jacobians of this piece read is complete once the piece's rollout is -- except its last knot's successor, which they do not need
            // (a failure here must not skip the join below: Jacobian pieces already queued on c->pipe would be left unordered against c->stream)
            hipError_t he = hipEventRecord(c->pipe_ev, c->stream);
            if (he == hipSuccess) he = hipStreamWaitEvent(c->pipe, c->pipe_ev, 0);
            if (he != hipSuccess) { k = fail(c, RDHIP_ERR_HIP, "%s: ordering a Jacobian piece behind its rollout failed: %s", what, hipGetErrorString(he)); break; }
            LaunchArgs jc = j;
            jc.stream = c->pipe; jc.N = nk * Mt; jc.Z = rc.Z; jc.t = rc.t; jc.dt = rc.dt; jc.J = (char*)dJ + (size_t)k0 * jb; jc.sched = nullptr;
            k = kernel_status(c, family_jacobian(d, rule, diff, dtype, jc), what); ++launches;
        }
        // (the context's stream is ordered behind the Jacobians whatever happened above)
        {
            hipError_t he = hipEventRecord(c->pipe_done, c->pipe);
            if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->pipe_done, 0);
            if (he != hipSuccess) { (void)hipStreamSynchronize(c->pipe); if (!k) k = fail(c, RDHIP_ERR_HIP, "%s: joining the Jacobian pieces failed: %s", what, hipGetErrorString(he)); }
        }
        c->last_rolljac_launches = launches;
        return memspace == RDHIP_HOST ? s.finish(k) : k;
    }
    int k = kernel_status(c, implicit ? family_implicit(c, d, 3, dtype, r) : family_rollout(d, rule, dtype, r), what);
    if (!k) { j.sched = take_sched(c, 0, (N + 63) / 64, dtype); k = kernel_status(c, implicit ? family_implicit(c, d, 1, dtype, j) : family_jacobian(d, rule, diff, dtype, j), what); }
    c->last_rolljac_launches = 2;
    return memspace == RDHIP_HOST ? s.finish(k) : k;
}
// The mixed-model Monte-Carlo batch (BASELINE config 5: even trajectories Cartpole, odd Satellite): every model's rollout + Jacobians is issued on its
// own stream between one fork and one join on the context's stream, so that the latency chains of the rollouts run side by side (each occupies a
// handful of compute units) and the Jacobian workgroups of the fused kernels fill the rest.
int rdhip_rollout_jacobian_multi(rdhip_ctx* c, int nmodels, const rdhip_model* const* models, const int* rule, int diff, int dtype, const long long* K, const long long* Mt,
                                 int layout, void* const* Z, const double* const* t, const double* const* dt, const void* const* X0, void* const* J, int memspace) {
    const char* what = "rdhip_rollout_jacobian_multi";
    if (!c) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (c->device < 0) return fail(c, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    if (nmodels < 0 || nmodels > 4) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: 0 .. 4 models per call (got %d)", what, nmodels);
    if (nmodels == 0) return RDHIP_OK;
    if (!models || !rule || !K || !Mt || !Z || !t || !dt || !J) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: NULL table", what);
    for (int p = 0; p < nmodels; ++p) if (!models[p]) return fail(c, RDHIP_ERR_INVALID_ARG, "%s: model %d is NULL", what, p);
    const bool concurrent = memspace == RDHIP_DEVICE && nmodels > 1 && !stream_is_capturing(c);
    if (!concurrent) {                                      // host arrays (each call copies and synchronises) or a capturing stream: one after the other
        for (int p = 0; p < nmodels; ++p) {
            const int rc = rdhip_rollout_jacobian(c, models[p], rule[p], diff, dtype, K[p], Mt[p], layout, Z[p], t[p], dt[p], X0 ? X0[p] : nullptr, J[p], memspace);
            if (rc) return rc;
        }
        return RDHIP_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // ONE launch for the pair the one-launch form exists for (mixed.hpp: a satellite-type rigid body + the cartpole under one rule, either layout,
    // ForwardAD): both rollouts and both models' Jacobians in one grid on the context's stream -- no fork, no join, no second stream
    if (nmodels == 2 && diff == RDHIP_FORWARD_AD && tune_knobs().mixed && c->allow_fused) {
        for (int pa = 0; pa < 2; ++pa) {
            const int pb = 1 - pa;
            bool ok = true;
            for (int p = 0; p < 2; ++p) ok = ok && Z[p] && t[p] && dt[p] && J[p] && K[p] > 0 && Mt[p] > 0;
            if (!ok) break;
            LaunchArgs a{}, b{};
            a.stream = c->stream; a.N = Mt[pa]; a.K = K[pa]; a.Z = Z[pa]; a.t = t[pa]; a.dt = dt[pa]; a.X0 = X0 ? X0[pa] : nullptr; a.J = J[pa]; a.time_major = layout == RDHIP_TIME_MAJOR ? 1 : 0;
            b.stream = c->stream; b.N = Mt[pb]; b.K = K[pb]; b.Z = Z[pb]; b.t = t[pb]; b.dt = dt[pb]; b.X0 = X0 ? X0[pb] : nullptr; b.J = J[pb]; b.time_major = a.time_major;
            a.progress = c->d_progress; a.progress_words = PROGRESS_WORDS; a.probe = 1;
            if (fam_mixed_rolljac(models[pa]->d, rule[pa], models[pb]->d, rule[pb], diff, dtype, a, b) != RDHIP_OK) continue;
            if (dtype != RDHIP_F64 && dtype != RDHIP_F32) break;
            if (c->side_pending) { const int rc = side_join(c); if (rc) return rc; }       // (ORDERING RULE of rdhip.h: behind whatever the overlap mode left on the internal stream)
            c->multi_sig = 0;
            a.probe = 0;
            c->roll_epoch = (c->roll_epoch + 1) & 0x7fffffffULL; if (c->roll_epoch == 0) c->roll_epoch = 1;
            a.pbase = c->roll_epoch << 32;
            const int k = spin_launch(c, c->stream, false, [&] { return fam_mixed_rolljac(models[pa]->d, rule[pa], models[pb]->d, rule[pb], diff, dtype, a, b); });
            c->last_rolljac_launches = 1;
            return kernel_status(c, k, what);
        }
    }
    // Overlap mode (contexts that own their stream, rdhip_ctx_set_overlap), two models: the second model runs on the context's INTERNAL stream and the
    // call neither waits for it at its end -- the next operation of any other kind does (side_join: the ORDERING RULE of rdhip.h) -- nor forks when it
    // REPEATS the previous call argument for argument (a solver's loop: every stream re-issues its own kernels on its own arrays, in order).  The
    // context's stream then carries nothing but the long model's launches back to back; fork record, join wait and the launch behind them are three
    // packets, ~18 us between one call's fused launch and the next one's.
    const bool lazy = c->overlap && c->own_stream && nmodels == 2 && c->side != nullptr;
    unsigned long long sig = 1469598103934665603ULL;
    std::vector<unsigned long long> argv_now;
    if (lazy) {
        auto mix = [&](unsigned long long v) { sig = (sig ^ v) * 1099511628211ULL; argv_now.push_back(v); };
        mix((unsigned long long)diff); mix((unsigned long long)dtype); mix((unsigned long long)layout);
        for (int p = 0; p < nmodels; ++p) {
            mix((unsigned long long)(uintptr_t)models[p]); mix((unsigned long long)rule[p]); mix((unsigned long long)K[p]); mix((unsigned long long)Mt[p]);
            mix((unsigned long long)(uintptr_t)Z[p]); mix((unsigned long long)(uintptr_t)t[p]); mix((unsigned long long)(uintptr_t)dt[p]);
            mix((unsigned long long)(uintptr_t)(X0 ? X0[p] : nullptr)); mix((unsigned long long)(uintptr_t)J[p]);
        }
        if (sig == 0) sig = 1;
    }
    const bool chained = lazy && c->multi_pending && c->multi_sig == sig && c->multi_args == argv_now;
    if (!chained) { const int rc = side_join(c); if (rc) return rc; }     // the fork event below must sit behind the overlap mode's internal stream as well (ORDERING RULE, rdhip.h)
    if (!c->fork_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
    if (lazy) {
        if (!c->side_child) { const int rc = rdhip_ctx_create(c->device, (void*)c->side, RDHIP_STREAM_BORROW, &c->side_child); if (rc) return fail(c, rc, "%s: %s", what, rdhip_last_error(nullptr)); }
    } else for (int p = 1; p < nmodels; ++p) {
        if (!c->child[p - 1]) { const int rc = rdhip_ctx_create(c->device, nullptr, RDHIP_STREAM_OWN, &c->child[p - 1]); if (rc) return fail(c, rc, "%s: %s", what, rdhip_last_error(nullptr)); }
        if (!c->join_ev[p - 1]) HIP_TRY(c, hipEventCreateWithFlags(&c->join_ev[p - 1], hipEventDisableTiming));
    }
    if (!chained) HIP_TRY(c, hipEventRecord(c->fork_ev, c->stream));      // the children start behind the work already queued on this context
    // ONE model of the call takes the fused (spin-waiting) launch -- the one with the most Jacobian bytes among those that have it; the others run
    // rollout and Jacobians as two launches on their streams (see fused_launch)
    int fused_model = -1; double fused_bytes = -1;
    double jbytes[4] = {0, 0, 0, 0};
    for (int p = 0; p < nmodels; ++p) {
        if (!Z[p] || !t[p] || !dt[p] || !J[p] || K[p] <= 0 || Mt[p] <= 0 || rule[p] < RDHIP_EULER || rule[p] > RDHIP_RK4) continue;
        const ModelDesc& d = models[p]->d;
        LaunchArgs f{}; f.N = Mt[p]; f.K = K[p]; f.Z = Z[p]; f.t = t[p]; f.dt = dt[p]; f.J = J[p]; f.time_major = (layout == RDHIP_TIME_MAJOR);
        f.progress = c->d_progress; f.progress_words = PROGRESS_WORDS; f.probe = 1;
        jbytes[p] = (double)K[p] * (double)Mt[p] * d.n * (d.n + d.m) * (dtype == RDHIP_F64 ? 8.0 : 4.0);
        if (family_rolljac(d, rule[p], diff, dtype, f) == RDHIP_OK && jbytes[p] > fused_bytes) { fused_model = p; fused_bytes = jbytes[p]; }
    }
    // The context's OWN stream takes the model whose part of the call is expected to take longest: consecutive calls then follow each other on one
    // stream, kernel behind kernel, and the event round trips of fork and join (~7 us before a child stream starts, ~20 us between the end of a child's
    // last kernel and the next call's first) fall on the models that have the slack.  Estimate per model: a rollout is a chain of K - 1 steps of
    // ~0.5 us, its Jacobians a stream at ~5 TB/s -- one after the other, or overlapped for the model with the fused launch.  Same-box A/B: the whole
    // of BASELINE config 5 on one GPU (the satellite's fused launch is the long part) 343 us with the cartpole on the context's stream, 294 us with
    // the satellite there; one GPU's share of it (both rollouts ~120 us, the cartpole's Jacobians behind its own) 160 us / 174 us.
    const int main_env = tune_knobs().multi_main;                  // tuning switch: a model index (-1: the estimate below)
    int main_model = 0; double longest = -1;
    for (int p = 0; p < nmodels; ++p) {
        const double chain = 0.5 * (double)(K[p] > 1 ? K[p] - 1 : 0), stream = jbytes[p] / 5.0e6;
        const double est = p == fused_model ? (chain > stream ? chain : stream) : chain + stream;
        if (est > longest) { longest = est; main_model = p; }
    }
    if (main_env >= 0 && main_env < nmodels) main_model = main_env;
    int rc0 = RDHIP_OK, slot = 0;
    for (int p = 0; p < nmodels && rc0 == RDHIP_OK; ++p) {  // the side streams first: the context's own stream takes its model last
        if (p == main_model) continue;
        rdhip_ctx* ch = lazy ? c->side_child : c->child[slot];
        if (!chained) HIP_TRY(c, hipStreamWaitEvent(ch->stream, c->fork_ev, 0));
        ch->allow_fused = (p == fused_model); ch->fused_beside = true;
        rc0 = rdhip_rollout_jacobian(ch, models[p], rule[p], diff, dtype, K[p], Mt[p], layout, Z[p], t[p], dt[p], X0 ? X0[p] : nullptr, J[p], memspace);
        if (rc0) fail(c, rc0, "%s (model %d): %s", what, p, rdhip_last_error(ch));
        if (!lazy) HIP_TRY(c, hipEventRecord(c->join_ev[slot], ch->stream));
        ++slot;
    }
    if (lazy) { c->side_pending = true; c->multi_pending = true; }      // (whatever happened above: the internal stream has work nothing may overtake)
    if (rc0 == RDHIP_OK) {
        const int p = main_model;
        c->allow_fused = (fused_model == p); c->fused_beside = nmodels > 1;
        c->in_overlapped_call = lazy;                        // (check_common must not join here: this call orders itself)
        rc0 = rdhip_rollout_jacobian(c, models[p], rule[p], diff, dtype, K[p], Mt[p], layout, Z[p], t[p], dt[p], X0 ? X0[p] : nullptr, J[p], memspace);
        c->in_overlapped_call = false;
        c->allow_fused = true; c->fused_beside = false;
    }
    if (lazy) { c->multi_sig = rc0 == RDHIP_OK ? sig : 0; c->multi_args.swap(argv_now); return rc0; }
    for (int p = 1; p < nmodels; ++p) if (c->join_ev[p - 1]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->join_ev[p - 1], 0));      // join: later work on the context sees every model's results
    return rc0;
}
int rdhip_ctx_last_rollout_jacobian_launches(rdhip_ctx* c, int* launches) {
    if (!c || !launches) return fail(c, RDHIP_ERR_INVALID_ARG, "NULL argument");
    *launches = c->last_rolljac_launches; return RDHIP_OK;
}

// ---- several GPUs in ONE process (hosts without torch.distributed / RCCL) --------------------------------------------------------
int rdhip_shard_range(long long N, int nshards, int i, long long* begin, long long* count) {
    if (N < 0 || nshards <= 0 || i < 0 || i >= nshards || !begin || !count) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_shard_range: bad arguments");
    const long long per = (N + nshards - 1) / nshards;
    const long long lo = (long long)i * per < N ? (long long)i * per : N;
    const long long hi = lo + per < N ? lo + per : N;
    *begin = lo; *count = hi - lo;
    return RDHIP_OK;
}
int rdhip_jacobian_sharded(rdhip_ctx* const* ctxs, int nctx, const rdhip_model* M, int rule, int diff, int dtype, long long N,
                           const void* Z, const double* t, const double* dt, void* J, void* Y) {
    if (!ctxs || nctx <= 0) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded: no contexts");
    for (int i = 0; i < nctx; ++i) {
        if (!ctxs[i]) return fail(nullptr, RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded: context %d is NULL", i);
        for (int k = 0; k < i; ++k) if (ctxs[k] == ctxs[i]) return fail(ctxs[i], RDHIP_ERR_INVALID_ARG, "rdhip_jacobian_sharded: context %d is listed twice (a context serves one host thread)", i);
    }
    if (!M) return fail(ctxs[0], RDHIP_ERR_INVALID_ARG, "model is NULL");
    if (dtype != RDHIP_F64 && dtype != RDHIP_F32) return fail(ctxs[0], RDHIP_ERR_INVALID_ARG, "bad dtype %d", dtype);
    if (N < 0) return fail(ctxs[0], RDHIP_ERR_INVALID_ARG, "negative N");
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const int n = M->d.n, nz = M->d.n + M->d.m;
    std::vector<int> rcs(nctx, RDHIP_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < nctx; ++i) {
        long long lo = 0, cnt = 0; rdhip_shard_range(N, nctx, i, &lo, &cnt);
        th.emplace_back([=, &rcs]() {
            rcs[i] = rdhip_jacobian(ctxs[i], M, rule, diff, dtype, cnt, Z ? (const char*)Z + (size_t)lo * nz * ts : nullptr, t ? t + lo : nullptr, dt ? dt + lo : nullptr,
                                    J ? (char*)J + (size_t)lo * n * nz * ts : nullptr, Y ? (char*)Y + (size_t)lo * n * ts : nullptr, RDHIP_HOST);
        });
    }
    for (auto& x : th) x.join();
    for (int i = 0; i < nctx; ++i) if (rcs[i]) { g_err = ctxs[i]->err; return rcs[i]; }     // also readable through rdhip_last_error(NULL)
    return RDHIP_OK;
}
int rdhip_memcpy_peer(rdhip_ctx* dc, void* dst, rdhip_ctx* sc, const void* src, unsigned long long bytes) {
    if (!dc || !sc) return fail(dc ? dc : sc, RDHIP_ERR_INVALID_ARG, "ctx is NULL");
    if (bytes == 0) return RDHIP_OK;
    if (!dst || !src) return fail(dc, RDHIP_ERR_INVALID_ARG, "rdhip_memcpy_peer: NULL pointer");
    if (dc->device < 0 || sc->device < 0) return fail(dc, RDHIP_ERR_NO_DEVICE, "this context has no device (RDHIP_DRY_RUN)");
    // order after the producer's stream, run on the consumer's stream
    HIP_TRY(sc, hipSetDevice(sc->device));
    { int rc = side_join(sc); if (rc) return rc; rc = side_join(dc); if (rc) return rc; }
    HIP_TRY(sc, hipEventRecord(sc->ev1, sc->stream));
    HIP_TRY(dc, hipSetDevice(dc->device));
    HIP_TRY(dc, hipStreamWaitEvent(dc->stream, sc->ev1, 0));
    HIP_TRY(dc, hipMemcpyPeerAsync(dst, dc->device, src, sc->device, bytes, dc->stream));
    return RDHIP_OK;
}

}  // extern "C"
