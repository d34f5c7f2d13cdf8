// common.hpp -- host-side descriptors shared by the C-ABI layer (api.hip) and the per-family kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include "../../include/rdhip.h"

namespace rd {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch (keyed by the kernel's
// address: kernel function-pointer TYPES are shared between different kernels, so a per-type static would not do)
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t lds) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;     // bit = device ordinal
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = done.find(kernel);
        if (it != done.end() && (it->second & bit)) return hipSuccess;
    }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(mu); done[kernel] |= bit; }
    return e;
}


// the kernel the calling host thread launched last through the library (rdhip_last_kernel_name: a benchmark or a test names the kernel a call was
// dispatched to instead of assuming it); compiled-in kernels by their host stub, hipRTC kernels by their module function
struct LastKernel { const void* host_fn; hipFunction_t mod_fn; bool dyn = false; };      // dyn: the launch was handed the counter words of the dynamic tile schedule
inline LastKernel& last_kernel() { static thread_local LastKernel k{nullptr, nullptr}; return k; }
#define RD_LAUNCH(kern, grid, block, lds, stream, ...)                                                    \
    do {                                                                                                  \
        auto rd_kern_ = (kern);                                                                           \
        rd::last_kernel() = rd::LastKernel{reinterpret_cast<const void*>(rd_kern_), nullptr, false};      \
        hipLaunchKernelGGL(rd_kern_, grid, block, lds, stream, __VA_ARGS__);                              \
    } while (0)

// compute units and LDS per compute unit of the current device, from the runtime (a CPX / NPS-partitioned MI355X reports 32 CUs per
// partition, not 256: the persistent grids are sized from this, never from a constant); cached per device ordinal
struct DevInfo { int cus; size_t lds_per_cu; };
inline DevInfo dev_info() {
    static std::mutex mu;
    static DevInfo cache[64];
    static bool have[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DevInfo{256, 160 * 1024};
    const int slot = dev & 63;
    std::lock_guard<std::mutex> g(mu);
    if (!have[slot]) {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        // (the library is gfx950-only and a gfx950 CU has 160 KiB of LDS; some runtimes report the 64 KiB per-workgroup default here)
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds < 160 * 1024) lds = 160 * 1024;
        (void)hipGetLastError();
        cache[slot] = DevInfo{cus, (size_t)lds};
        have[slot] = true;
    }
    return cache[slot];
}
// ---- tuning knobs -----------------------------------------------------------------------------------------------------------------------------
// Every developer switch of the dispatcher in ONE place: parsed once per process from RDHIP_TUNE="key=value,key=value" (api.hip tune_knobs_mut; the
// exported rdhip_tune() merges more pairs at run time -- tests and tuning scripts).  None of them changes a result; they select between kernels /
// launch shapes that the dispatcher would otherwise choose by size.  include/rdhip.h lists them for users of the library.
struct TuneKnobs {
    long long static_grid = 0;            // workgroups of the one-per-CU persistent launches (0: one per compute unit)         tune/tune_grid.hip
    long long dyn_grid = 0;               // workgroups of launches with the dynamic tile schedule (0: three quarters of the CUs)
    bool dynamic_tiles = true;            // dynamic_tiles=0: no launch takes the dynamic tile schedule
    char small = 0;                       // small=tile: the tile loop for every batch size (no four-workgroups-per-tile form)
    long long small_max = 2;              // workgroups per CU up to which the four-workgroups-per-tile form is used
    char rollout = 0;                     // rollout=plain|cascade|stream: force a rollout kernel where it applies
    char rolljac = 0;                     // rolljac=split: rollout + Jacobians always as two launches
    double fused_ratio = 0;               // the rollout-step / tile-time ratio the fused launch's second part is sized with (0: 5.9 for fp32, none for fp64)
    long long fused_jac_workgroups = 0;   // Jacobian workgroups of a fused launch (0: every CU the rollout groups leave)
    long long fused_tj_max_knots = 262144;      // trajectory-major stacks from this many knot points on take two launches
    long long rolljac_pipe_min_knots = 262144;  // time-major stacks of models without the fused launch are pipelined in pieces from this size on (0: never)
    int multi_main = -1;                  // rdhip_rollout_jacobian_multi: the model that gets the fused launch (-1: the one with most Jacobian bytes)
    int mixed_idle = -1;                  // mixed_idle=0/1: the small model's rollout on workgroups of its own / on the idle waves of the other model's rollout workgroups (-1: by size)
    bool mixed = true;                    // mixed=0: rdhip_rollout_jacobian_multi never takes the one-launch form of a (satellite, cartpole) pair (mixed.hpp)
    bool fused_two_per_cu = true;         // fused_two_per_cu=0: the fp32 one-launch forms never take their two-workgroups-per-CU kernel (fused.hpp fused_min_waves)
    char ift = 0;                         // ift=reg: the one-kernel register path; ift=3: Newton kernel + tile kernel + gated exact kernel (round 2)
};
TuneKnobs& tune_knobs_mut();              // api.hip
inline const TuneKnobs& tune_knobs() { return tune_knobs_mut(); }

// persistent grid for the tile kernels: as many workgroups as fit on the chip at once (LDS- and wave-limited), never more than tiles
inline unsigned tile_grid(size_t lds_bytes, int threads, long long ntiles) {
    const DevInfo di = dev_info();
    const int by_lds = (int)(di.lds_per_cu / (lds_bytes ? lds_bytes : 1));
    const int by_waves = 2048 / threads;
    int per_cu = by_lds < by_waves ? by_lds : by_waves;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    const long long g = (long long)di.cus * per_cu;
    return (unsigned)(ntiles < g ? ntiles : g);
}

// the same with the kernel's REGISTER footprint taken into account (the LDS / thread bound above assumes the registers never limit residency: an fp32
// tile kernel whose 67 KB of LDS would let two workgroups share a CU but whose 146 registers do not must get one workgroup per CU, or half of the
// "persistent" grid queues behind the other half); asked from the runtime once per (kernel, device)
inline unsigned tile_grid_occ(const void* kernel, size_t lds_bytes, int threads, long long ntiles) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> per_cu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return tile_grid(lds_bytes, threads, ntiles);
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = per_cu_of[dev & 63].find(kernel);
        if (it != per_cu_of[dev & 63].end()) per_cu = it->second;
    }
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds_bytes) != hipSuccess || n < 1) { (void)hipGetLastError(); return tile_grid(lds_bytes, threads, ntiles); }
        per_cu = n > 8 ? 8 : n;
        std::lock_guard<std::mutex> g(mu);
        per_cu_of[dev & 63][kernel] = per_cu;
    }
    const unsigned lds_bound = tile_grid(lds_bytes, threads, ntiles);
    const long long grid_env = tune_knobs().static_grid;     // (tuning) workgroups per launch of the one-per-CU persistent kernels instead of one per compute unit
    if (grid_env > 0 && per_cu == 1) { const long long r0 = ntiles < grid_env ? ntiles : grid_env; return (unsigned)(r0 < (long long)lds_bound ? r0 : (long long)lds_bound); }
    const long long g = (long long)dev_info().cus * per_cu;
    const long long r = ntiles < g ? ntiles : g;
    return (unsigned)(r < (long long)lds_bound ? r : (long long)lds_bound);
}

// Persistent grid of the split-role launches WITH the dynamic tile schedule: three quarters of the chip, not all of it.  Round 5 (tune/tune_grid.hip, two
// boxes, satellite RK4 ForwardAD fp64, us per launch on 256 / 224 / 208 / 192 / 176 / 160 workgroups): 131 072 knots 52.6 / 52.0 / 49.6 / 47.7 / 50.0 /
// 53.5, 262 144: 97.9 / 95.6 / 87.7 / 89.7 / 93.0 / 99.2, 1 M: 404 / 396 / 387 / 370 / 370 / 391 -- 192 workgroups move 9-10 % more bytes per second
// than 256.  Every workgroup streams its own 126 KB slab; fewer concurrent streams meet fewer open DRAM rows per bank (a streaming fill shows the same
// preference), and the counter keeps the smaller grid balanced.  Static launches of a few tiles per workgroup (the 65 536-knot headline: 4 per
// workgroup) lose to tile quantisation what they gain (27.15 us on 256, 26.85 on 160, 27.4 on 176, 28.6 on 192) and keep the whole chip; fp32
// tiles are half the size and want all 256 (1 M knots: 205 us on 256, 217-222 on 192).  RDHIP_TUNE=dyn_grid=N overrides (tuning).
// Only where it was measured: kernels with ONE workgroup per compute unit (full == the CU count).  A kernel whose tiles are small enough for two or three
// workgroups per CU (LieSatellite: 45 KB) keeps its full grid -- round 5 cut those to 192 workgroups too, a quarter of what fits.
inline unsigned dyn_grid(unsigned full) {
    const long long env = tune_knobs().dyn_grid;
    const unsigned cus = (unsigned)dev_info().cus;
    if (env > 0) return (unsigned)(env < (long long)full ? env : (long long)full);
    if (full != cus) return full;
    const unsigned g = (cus * 3 / 4) & ~7u;
    return g >= 8 ? g : full;
}

// implicit kernels: flag words of a context (word 0: whole batch; words 1 .. : one per 64-knot tile)
constexpr long long IMPLICIT_FLAG_WORDS = 1 << 20;

struct ModelDesc {
    int kind, rot, frame;
    int n, m, e;
    int nparams;
    double params[64];
    int lie_nP;        // 0 => EuclideanState
    int lie_P[8];
    void* jit;         // RDHIP_MODEL_CUSTOM: the hipRTC plug-in object (jit.hip)
    int newton_iters;  // ImplicitNewtonCache defaults (src/integration.jl:599-600): 10 iterations, tolerance 1e-12
    double newton_tol;
};

// rdhip_jacobian_multi: several independent packed trajectories for ONE launch.  MultiPod is the byte image of kernels.hpp MultiBatch<T> (typed
// pointers there; dispatch.hpp asserts the sizes agree): up to MULTI_INLINE batches inline (they travel in the kernel arguments), any number
// through the device table `ext` / `ext_tile0` (then the inline arrays are unused).
constexpr int MULTI_INLINE = 32;
struct MultiRecPod { const void* Z; const double* t; const double* dt; void* J; void* Y; long long N; long long pad[2]; };   // 64 bytes
struct MultiPod {
    int nb;
    int uniform;                 // > 0: every batch has this many 64-knot tiles
    long long total;             // tiles of all batches (< 2^31)
    const MultiRecPod* ext;      // device table (nb records) or nullptr
    const unsigned* ext_tile0;   // device: nb + 1 entries
    int group;                   // ext: batches per entry of the coarse inline table
    int pad_;
    unsigned tile0[MULTI_INLINE + 1];
    long long N[MULTI_INLINE];
    const void* Z[MULTI_INLINE]; const double* t[MULTI_INLINE]; const double* dt[MULTI_INLINE]; void* J[MULTI_INLINE]; void* Y[MULTI_INLINE];
};
struct MultiDesc {
    MultiPod pod;
    bool aligned16;              // every J / Y of every batch is 16-byte aligned (the split-role kernels copy 16-byte units)
};

struct LaunchArgs {
    hipStream_t stream;
    long long N;        // knot points (rollout: trajectories M)
    long long K;        // rollout: knots per trajectory
    const void* Z;
    const double* t;
    const double* dt;
    void* J;
    void* Y;            // eval: output; jacobian: optional f(z)
    const void* X0;     // rollout
    const void* U;      // rollout on split arrays (rollout_split.hpp): controls m x (K-1) x M; Z then holds the states X n x K x M.  NULL: a packed stack
    void* J2;           // implicit dynamics_error_jacobian: d e / d z2 (J holds d e / d z1)
    int newton_iters;   // implicit rules
    double newton_tol;
    int time_major;     // rollout: knot k of trajectory j at k*M + j instead of j*K + k
    int* flag;          // implicit rules: device word of the context ("a row exchange was needed"); NULL = LDS kernels only
    int epoch;          // implicit rules: 1 .. 2^22, different for consecutive calls on a context (0: none); see implicit_midpoint_kernel.  A captured
                        // graph bakes ONE epoch into its kernel arguments: replays stay correct (the tile words are rewritten by every call) but may lose
                        // the gated pass's one-load early exit when a stale gate[0] happens to equal that epoch (it then scans its tile words)
    unsigned long long* progress;   // fused rollout -> Jacobians: one progress word per 64-trajectory group (context workspace, zero at allocation)
    unsigned long long pbase;       //   (call epoch << 32): words of earlier calls compare below
    int progress_words;             //   capacity of `progress`
    int probe;                      //   != 0: say whether the fused launch exists for these arguments (RDHIP_OK / RDHIP_ERR_UNSUPPORTED) without launching it
    int beside;                     //   != 0: launches of other models of the same call run beside this one on other streams: keep to one workgroup per compute unit
    unsigned* sched;    // split-role Jacobian kernels: the launch's counter words of the dynamic tile schedule (kernels.hpp jac_split_loop), or NULL
    const MultiDesc* multi;   // jacobian: non-NULL = several batches in one launch (N, Z, t, dt, J, Y above are ignored); families that have no
                              // fused kernel for the (model, rule, diff) return RDHIP_ERR_UNSUPPORTED BEFORE launching anything and the caller
                              // loops over the batches
};

// per-family entry points (one translation unit each so they compile in parallel)
#define RD_DECLARE_FAMILY(name)                                                                      \
    int name##_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);                  \
    int name##_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a);    \
    int name##_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);       \
    int name##_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a);   /* mode 0: x_next, 1: x_next + IFT Jacobian, 2: dynamics error (+ Jacobians) of the pairs, 3: rollout */ \
    int name##_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a);   /* rollout + Jacobians in one launch (fused.hpp) or RDHIP_ERR_UNSUPPORTED before anything is launched */
// the mixed-model batch as one launch (fam_mixed.hip, mixed.hpp): model A (args a) with the one-launch rollout -> Jacobians form + small model B (args b)
int fam_mixed_rolljac(const ModelDesc& da, int rule_a, const ModelDesc& db, int rule_b, int diff, int dtype, const LaunchArgs& a, const LaunchArgs& b);
RD_DECLARE_FAMILY(fam_cartpole)
RD_DECLARE_FAMILY(fam_simple)      // double integrator, linear
RD_DECLARE_FAMILY(fam_satellite)
RD_DECLARE_FAMILY(fam_quadrotor)
RD_DECLARE_FAMILY(fam_lie)
RD_DECLARE_FAMILY(fam_rigid3)   // RigidBody{MRP}, RigidBody{RodriguesParam}
// wrenches(model, x, u, t) (jac = 0: a.Y = W, 6 x N) and d wrenches / d z (jac = 1: a.J = F, 6 x (n+m) x N) of the RigidBody models (fam_wrench.hip)
int fam_wrench(const ModelDesc& d, int dtype, const LaunchArgs& a, int jac);
int fam_satellite_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);
int fam_quadrotor_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);
int fam_lie_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a);               // satellite pair, lie satellite

// hipRTC model plug-in (jit.hip)
void* jit_model_create(int n, int m, const char* body, const char* preamble, const double* params, int np, int quat_rotations, const int* P, int nP);
void jit_model_destroy(void* h);
const char* jit_last_error(void* h);
int jit_precompile(void* h, int op, int rule, int dtype);
int jit_balance(void* h);
int jit_set_cascade(void* h, int nb, const int* state_block, const int* control_block);
int jit_eval(void* h, int rule, int dtype, const LaunchArgs& a);
int jit_jacobian(void* h, int rule, int diff, int dtype, const LaunchArgs& a);
int jit_rollout(void* h, int rule, int dtype, const LaunchArgs& a);
int jit_errexp(void* h, int rule, int dtype, const LaunchArgs& a);
int jit_implicit(void* h, int mode, int dtype, const LaunchArgs& a);

// error-state kernels (errstate.hip)
struct LieDesc {
    int n, e, nrot;
    int rot;                    // rdhip_rotation of the rotation blocks (QUAT: 4 parameters, MRP / RP: 3)
    signed char row_kind[16];   // 0: vector row, 1: quaternion row, 2: MRP / RodriguesParam row
    signed char row_col[16];    // vector row: its column in G ; quaternion row: first column of the 3-wide block
    signed char row_sub[16];    // quaternion row: component index 0..3
    signed char row_q0[16];     // quaternion row: state index of the quaternion's scalar part
    signed char rot_row0[4];    // state index of rotation k's scalar part
    signed char rot_col0[4];    // error-state index of rotation k's first component
    signed char err_src[16];    // state_diff: error index -> state index for vector parts, -1-k for rotation k parts
};
int launch_errstate_jacobian(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, int clear, void* G);
int launch_state_diff(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* X0, int ldx0, int errmap, void* dX);
// dynamics-error helpers (errstate.hip): E[:,k] -= Z[0:n, k+1] ; J2[k] = [-I 0]
int launch_sub_next_state(int dtype, hipStream_t s, long long pairs, int n, int nz, const void* Z, void* E);
int launch_fill_neg_identity(int dtype, hipStream_t s, long long pairs, int n, int nz, void* J2);
// generic (any LieState) error-state projection of a stack of Jacobians: E[k] = G(x_{k+1})' [A_k G(x_k) | B_k]
int launch_project_errexp(const LieDesc& d, int dtype, hipStream_t s, long long pairs, int m, const void* Z, const void* J, void* E);
int launch_nabla_errstate(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* Xb, int ldxb, int clear, void* H);

}  // namespace rd
