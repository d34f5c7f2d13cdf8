// jit.hip -- hipRTC model plug-in: user-defined continuous dynamics on the GPU.
//
// The reference is built around user-written `dynamics(model, x, u)` methods (src/dynamics.jl:23-45; RigidBody users write
// `forces` / `moments`, src/rigidbody.jl:244-258).  Julia cannot be compiled for the GPU here, so a custom model is given as the
// BODY of a HIP device function, written once on a generic scalar type S:
//
//     template <class S> __device__ void f(const S* x, const S* u, T t, S* xdot) const   // p[] = the model's parameters (type T),
//     { <body> }                                                                         // t = the time the dynamics are evaluated at
//
// and the library JIT-compiles the same kernel templates the compiled-in models use (kernels.hpp, embedded at build time) for
// that model with hipRTC: `S` becomes T for evaluation and Dual<T,K> for ForwardAD / the chain rule, so every rule (Euler, RK2,
// RK3, RK4), diff method (ForwardAD, FiniteDifference, UserDefined) and dtype is available.  Kernels are compiled lazily per
// (operation, rule, dtype) on first use (a few seconds each) and cached in the model object.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>
#include "common.hpp"
#include "jit_api.hpp"
#include "embedded_headers.inc"

namespace rd {

struct JitKernel { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; size_t lds = 0; int block = 0; int knots = 64; bool persistent = false; bool small = false; };
struct JitModel {
    int n = 0, m = 0, np = 0;
    std::string name, body, preamble;
    // discrete: `body` is the body of step(x, u, t, h, xn) -- a user-defined DiscreteDynamics model (src/discrete_dynamics.jl:80-108); every call takes
    // rule RDHIP_DISCRETE.  jac_body (may be empty): the user's own Jacobian, jac(x, u, t, J) / jac(x, u, t, h, J): what RDHIP_USER_DEFINED evaluates
    // (src/dynamics.jl:115-116, src/discrete_dynamics.jl:111-113; user_model.hpp)
    bool discrete = false;
    std::string jac_body;
    std::vector<double> params;
    std::mutex mu;
    std::map<std::string, std::vector<char>> code;        // key -> code object (device independent)
    std::map<std::string, std::string> lowered;
    std::map<std::string, JitKernel> loaded;              // key@device -> loaded function (entries are never erased while the model lives: launches use them outside the lock)
    int generation = 0;                                   // jit_balance re-keys the entries it replaces ("retired<generation>:..."): their modules stay loaded until the model is destroyed
    std::string err;
    std::vector<int> plan_wave;                           // column -> compute wave (after jit_balance); empty: contiguous ranges
    int plan_value_wave = 0;
    int plan_split = 0;                                   // > 0: the leading plan_split columns are the early part of a first tile (TablePlanS)
    std::vector<int> rot_starts;                         // LieState{QuatRotation,P}: state index of every quaternion (empty: EuclideanState or MRP / RP)
    // declared block structure (jit_set_cascade; models.hpp Cascade<>): state component -> block, control -> first block that reads it
    int cas_nb = 0;
    std::vector<int> cas_state, cas_control;
    // the declaration has been verified against the model's Jacobian pattern (first GPU use).  Read and written under `vmu` only: a second host
    // thread that arrives while the check is running waits for its verdict instead of launching cascade kernels on an unverified declaration.
    bool cas_checked = false;
    std::mutex vmu;
};

enum { JOP_EVAL = 0, JOP_ROLLOUT = 1, JOP_JAC_AD = 2, JOP_JAC_FD = 3, JOP_JAC_CHAIN = 4, JOP_JAC_AD_LOOP = 5, JOP_JAC_CHAIN_LOOP = 6,
       JOP_IMPLICIT_X = 7, JOP_IMPLICIT_J = 8, JOP_IMPLICIT_ERR = 9, JOP_IMPLICIT_ROLLOUT = 10, JOP_JAC_FD_LOOP = 11, JOP_ERREXP = 12,
       JOP_IMPLICIT_X_REG = 13, JOP_IMPLICIT_J_REG = 14, JOP_IMPLICIT_ROLLOUT_REG = 15, JOP_IMPLICIT_IFT = 16, JOP_IMPLICIT_ROLLOUT_CAS = 17,
       JOP_ROLLOUT_PLAIN = 18, JOP_JAC_USER = 19, JOP_JAC_USER_CHAIN = 20,
       JOP_MULTI = 32 };           // + JOP_MULTI: the same Jacobian operation over a MultiBatch (rdhip_jacobian_multi), ops 2 - 6 and 11   // 5, 6, 11: element-wise fallbacks for unaligned outputs; 18: one lane per trajectory whatever the model declares

// CascadeLds<Fn, Cas, 4>::BYTES (cascade.hpp) of a declared block structure for a rule with `ns` stage points: ring of published stage points +
// the IO wavefront's double-buffered control / time / state tiles + flags.  The cascade kernels exist for 1 <= m <= 8, n <= 16 (the IO wavefront's lane
// groups) and only while this fits the 160 KiB of a gfx950 CU; a declaration that does not fit keeps the plain kernels (cascade_fits).
static size_t cascade_lds_bytes(const JitModel& M, int ns, int dtype) {
    const size_t ts = dtype == RDHIP_F64 ? 8 : 4;
    auto a16 = [](size_t b) { return (b + 15) / 16 * 16; };
    const int R = 4, S = 4;
    int npub = 0; for (int i = 0; i < M.n; ++i) npub += M.cas_state[i] < M.cas_nb - 1 ? 1 : 0;
    size_t units = ((size_t)npub * ts + 15) / 16; if (units % 2 == 0) ++units;               // lane-major ring: lane stride = an odd number of 16-byte units ...
    const size_t in_e = (size_t)S * 64 * (M.m | 1), out_e = (size_t)S * 64 * (M.n | 1);
    const size_t padded = (size_t)ns * 64 * (units * 16 / ts);
    const bool lane_major = npub > 0 && a16(R * padded * ts) + a16(2 * in_e * ts) + 2 * (size_t)2 * S * 64 * 8 + a16(2 * out_e * ts) + 64 <= 160 * 1024;   // ... where it fits (CascadeLds::LANE_MAJOR)
    const size_t ring_slot = lane_major ? padded : (size_t)ns * npub * 64, ts_elems = (size_t)2 * S * 64;
    const size_t in_off = a16(R * ring_slot * ts), ts_off = in_off + a16(2 * in_e * ts), out_off = ts_off + 2 * ts_elems * 8, flag_off = out_off + a16(2 * out_e * ts);
    return flag_off + 64;
}
static int rule_stage_count(int rule) { return (rule == RDHIP_EULER || rule == RDHIP_IMPLICIT_MIDPOINT) ? 1 : rule == RDHIP_RK2 ? 2 : rule == RDHIP_RK3 ? 3 : 4; }
static bool cascade_fits(const JitModel& M, int rule, int dtype) {
    return M.cas_nb > 1 && M.m >= 1 && M.m <= 8 && M.n <= 16 && cascade_lds_bytes(M, rule_stage_count(rule), dtype) <= 160 * 1024;
}

static std::string rot_start_expr(const JitModel& M) {
    std::ostringstream e;
    for (size_t r = 0; r + 1 < M.rot_starts.size(); ++r) e << "if (r == " << r << ") return " << M.rot_starts[r] << "; ";
    e << "return " << (M.rot_starts.empty() ? 0 : M.rot_starts.back()) << ";";
    return e.str();
}
static std::string program_source(const JitModel& M) {
    std::ostringstream s;
    const bool user = M.discrete || !M.jac_body.empty();
    s << "#include \"implicit.hpp\"\n#include \"cascade.hpp\"\n" << (user ? "#include \"user_model.hpp\"\n" : "") << "namespace rd {\n" << M.preamble << "\n"
      << "template <class T_> struct UserModel {\n    using T = T_;\n    static constexpr int NX = " << M.n << ", NU = " << M.m << ";\n"
      << "    static constexpr bool HAS_USER_JACOBIAN = " << ((M.discrete && M.jac_body.empty()) ? "false" : "true") << ";\n    static constexpr int NROT = " << M.rot_starts.size() << ";\n"
      << "    static constexpr int rot_start(int r) { " << rot_start_expr(M) << " }\n    T p[" << (M.np > 0 ? M.np : 1) << "];\n";
    // the bodies sit in an inner block: a body written for the time-invariant signature may declare its own `t`
    if (M.discrete) s << "    template <class S> RD_DEV void step(const S* x, const S* u, T t, T h, S* xn) const {\n        (void)t; (void)h;\n        {\n" << M.body << "\n        }\n    }\n";
    else s << "    template <class S> RD_DEV void f(const S* x, const S* u, T t, S* xdot) const {\n        (void)t;\n        {\n" << M.body << "\n        }\n    }\n";
    if (!M.jac_body.empty()) {
        if (M.discrete) s << "    RD_DEV void jac(const T* x, const T* u, T t, T h, T* J) const {\n        (void)t; (void)h;\n        {\n" << M.jac_body << "\n        }\n    }\n";
        else s << "    RD_DEV void jac(const T* x, const T* u, T t, T* J) const {\n        (void)t;\n        {\n" << M.jac_body << "\n        }\n    }\n";
    }
    s << "};\n";
    if (M.cas_nb > 1) {            // the declared block structure: the cascade kernels (FiniteDifference, rollout, Newton) pick it up like a compiled-in model's
        s << "template <class T_> struct Cascade<UserModel<T_> > {\n    static constexpr int NB = " << M.cas_nb << ";\n"
          << "    static constexpr int block_of(int i) { constexpr int b[" << M.n << "] = {";
        for (int i = 0; i < M.n; ++i) s << (i ? ", " : "") << M.cas_state[i];
        s << "}; return b[i]; }\n    static constexpr int control_block(int j) { constexpr int b[" << (M.m > 0 ? M.m : 1) << "] = {";
        for (int j = 0; j < M.m; ++j) s << (j ? ", " : "") << M.cas_control[j];
        if (M.m == 0) s << "0";
        s << "}; return b[j]; }\n};\n";
    }
    s << "}\n";
    return s.str();
}
static std::string kernel_expr(const JitModel& M, int op, int rule, int dtype) {
    const char* T = dtype == RDHIP_F64 ? "double" : "float";
    const int nz = M.n + M.m;
    std::ostringstream fn, s;
    if (op == JOP_IMPLICIT_ROLLOUT_CAS) {    // implicit rollout along the declared cascade (implicit.hpp ImplicitRollFun on cascade.hpp rollout_cascade_kernel)
        s << "rd::rollout_cascade_kernel<rd::ImplicitRollFun<rd::UserModel<" << T << "> >, " << T << ", rd::Cascade<rd::UserModel<" << T << "> >, 4>";
        return s.str();
    }
    if (op == JOP_IMPLICIT_IFT) {            // IFT Jacobian through the split-role tile loop (implicit.hpp implicit_ift_split_kernel)
        s << "rd::implicit_ift_split_kernel<rd::IftFun<rd::UserModel<" << T << "> >, " << T << ", ";
        if (!M.plan_wave.empty()) {
            if (M.plan_split > 0) s << "rd::TablePlanS<" << nz << ", 4, " << M.plan_value_wave << ", " << M.plan_split;
            else s << "rd::TablePlan<" << nz << ", 4, " << M.plan_value_wave;
            for (int c = 0; c < nz; ++c) s << ", " << M.plan_wave[c];
            s << ">";
        } else s << "rd::ContiguousPlan<" << nz << ", 4>";
        s << ", 4>";
        return s.str();
    }
    if (op >= JOP_IMPLICIT_X_REG && op <= JOP_IMPLICIT_ROLLOUT_REG) {          // Newton matrix in registers (implicit.hpp)
        s << "rd::" << (op == JOP_IMPLICIT_ROLLOUT_REG ? "implicit_rollout_reg_kernel" : "implicit_midpoint_reg_kernel") << "<rd::UserModel<" << T << ">, " << T;
        if (op == JOP_IMPLICIT_X_REG) s << ", false"; else if (op == JOP_IMPLICIT_J_REG) s << ", true";
        s << ">";
        return s.str();
    }
    if (op >= JOP_IMPLICIT_X && op <= JOP_IMPLICIT_ROLLOUT) {      // ImplicitMidpoint kernels are templated on the model itself
        const char* k = op == JOP_IMPLICIT_ERR ? "implicit_midpoint_error_kernel" : op == JOP_IMPLICIT_ROLLOUT ? "implicit_rollout_kernel" : "implicit_midpoint_kernel";
        s << "rd::" << k << "<rd::UserModel<" << T << ">, " << T;
        if (op == JOP_IMPLICIT_X) s << ", false"; else if (op == JOP_IMPLICIT_J) s << ", true";
        s << ">";
        return s.str();
    }
    if (op == JOP_ERREXP) {          // the Jacobian kernel with the tangent seeded in error coordinates and the rows projected (kernels.hpp)
        const int ne = M.n - (int)M.rot_starts.size(), nc = ne + M.m;
        std::ostringstream plan;
        if (!M.plan_wave.empty() && nc >= 4) {           // balanced state columns -> error columns: a rotation's 3 error columns inherit the waves of its first 3 quaternion columns
            plan << "rd::TablePlan<" << nc << ", 4, " << M.plan_value_wave;
            size_t r = 0;
            for (int st = 0; st < M.n;) {
                if (r < M.rot_starts.size() && st == M.rot_starts[r]) { for (int j = 0; j < 3; ++j) plan << ", " << M.plan_wave[st + j]; st += 4; ++r; }
                else { plan << ", " << M.plan_wave[st]; ++st; }
            }
            for (int i = 0; i < M.m; ++i) plan << ", " << M.plan_wave[M.n + i];
            plan << ">";
        } else plan << "rd::ContiguousPlan<" << nc << ", " << (nc >= 4 ? 4 : nc) << ">";
        s << "rd::jac_errexp_split_kernel<rd::ErrFun<rd::UserModel<" << T << ">, " << rule << ">, " << T << ", " << plan.str() << ", 4>";
        return s.str();
    }
    fn << "rd::Fun<rd::UserModel<" << T << ">, " << rule << ">";
    const int ng = nz >= 4 ? 4 : nz;
    std::ostringstream plan;
    const int bop = op >= JOP_MULTI ? op - JOP_MULTI : op;
    if (!M.plan_wave.empty() && (bop == JOP_JAC_AD || bop == JOP_JAC_CHAIN || (bop == JOP_JAC_FD && M.cas_nb > 1 && rule != RDHIP_CONTINUOUS)) && nz > 10) {     // cost-balanced column -> wave table
        if (M.plan_split > 0) plan << "rd::TablePlanS<" << nz << ", 4, " << M.plan_value_wave << ", " << M.plan_split;
        else plan << "rd::TablePlan<" << nz << ", 4, " << M.plan_value_wave;
        for (int c = 0; c < nz; ++c) plan << ", " << M.plan_wave[c];
        plan << ">";
    } else plan << "rd::ContiguousPlan<" << nz << ", " << ng << ">";
    if (op >= JOP_MULTI) {           // several packed trajectories, one launch: the generic multi kernels on the body the single-batch kernel uses
        const int base = op - JOP_MULTI;
        const std::string F = fn.str(), P = plan.str(), cont = "rd::ContiguousPlan<" + std::to_string(nz) + ", " + std::to_string(ng) + ">";
        const std::string cas = std::string("rd::Cascade<rd::UserModel<") + T + "> >";
        auto split = [&](const std::string& body) { s << "rd::jac_split_multi_kernel<" << F << ", " << T << ", " << P << ", 4, " << body << ", 1>"; };
        auto tile = [&](const std::string& body_name) { s << "rd::jac_tile_multi_kernel<" << F << ", " << T << ", " << cont << ", rd::" << body_name << "<" << F << ", " << T << ", " << cont << "> >"; };
        switch (base) {
            case JOP_JAC_AD:
                if (nz <= 10) s << "rd::jac_ad_small_multi_kernel<" << F << ", " << T << ">";
                else split("rd::AdUnrolledBody<" + F + ", " + T + ", " + P + ">");
                break;
            case JOP_JAC_FD:
                if (nz <= 10) tile("FdBody");
                else if (M.cas_nb > 1 && rule != RDHIP_CONTINUOUS) split("rd::FdCascadeBody<" + F + ", " + T + ", " + P + ", " + cas + ">");
                else split("rd::FdBody<" + F + ", " + T + ", " + P + ">");
                break;
            case JOP_JAC_CHAIN:
                if (nz <= 10) tile("ChainBody");
                else split("rd::ChainUnrolledBody<" + F + ", " + T + ", " + P + ">");
                break;
            case JOP_JAC_AD_LOOP: tile("AdLoopBody"); break;
            case JOP_JAC_FD_LOOP: tile("FdBody"); break;
            case JOP_JAC_USER: case JOP_JAC_USER_CHAIN: {            // the user's own Jacobian: one wave per tile (user_model.hpp)
                const std::string one = "rd::ContiguousPlan<" + std::to_string(nz) + ", 2>";
                s << "rd::jac_tile_multi_kernel<" << F << ", " << T << ", " << one << ", rd::" << (base == JOP_JAC_USER ? "UserJacBody" : "UserChainBody") << "<" << F << ", " << T << ", " << one << "> >";
                break;
            }
            default: tile("ChainBody"); break;
        }
        return s.str();
    }
    switch (op) {
        case JOP_EVAL: s << "rd::eval_kernel<" << fn.str() << ", " << T << ">"; break;
        case JOP_ROLLOUT_PLAIN: s << "rd::rollout_kernel<" << fn.str() << ", " << T << ">"; break;
        case JOP_ROLLOUT:
            if (cascade_fits(M, rule, dtype)) s << "rd::rollout_cascade_kernel<" << fn.str() << ", " << T << ", rd::Cascade<rd::UserModel<" << T << "> >, 4>";
            else s << "rd::rollout_kernel<" << fn.str() << ", " << T << ">";
            break;
        case JOP_JAC_AD:       // small models: all columns per lane; larger ones: the split-role kernel with compile-time columns
            if (nz <= 10) s << "rd::jac_ad_small_kernel<" << fn.str() << ", " << T << ">";
            else s << "rd::jac_ad_split_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ", 4>";
            break;
        case JOP_JAC_AD_LOOP: s << "rd::jac_ad_loop_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ">"; break;
        case JOP_JAC_FD:
            if (nz <= 10) s << "rd::jac_fd_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ">";
            else if (M.cas_nb > 1 && rule != RDHIP_CONTINUOUS) s << "rd::jac_fd_cascade_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ", rd::Cascade<rd::UserModel<" << T << "> >, 4>";
            else s << "rd::jac_fd_split_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ", 4>";
            break;
        case JOP_JAC_FD_LOOP: s << "rd::jac_fd_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ">"; break;
        case JOP_JAC_CHAIN:
            if (nz <= 10) s << "rd::jac_chain_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ">";
            else s << "rd::jac_chain_split_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ", 4>";
            break;
        case JOP_JAC_USER: s << "rd::jac_user_kernel<" << fn.str() << ", " << T << ", rd::ContiguousPlan<" << nz << ", 2> >"; break;
        case JOP_JAC_USER_CHAIN: s << "rd::jac_user_chain_kernel<" << fn.str() << ", " << T << ", rd::ContiguousPlan<" << nz << ", 2> >"; break;
        default: s << "rd::jac_chain_kernel<" << fn.str() << ", " << T << ", " << plan.str() << ">"; break;
    }
    return s.str();
}

static const char* jit_opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-signed-zeros", "-ffinite-math-only", "-fno-slp-vectorize", "-DRDHIP_RTC=1"};

// ---- optional on-disk cache of compiled kernels (RDHIP_CACHE_DIR): a solver process pays the 2-5 s per kernel once, not per run ----
static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
static std::string cache_path(const std::string& src, const std::string& expr) {
    const char* dir = getenv("RDHIP_CACHE_DIR");
    if (!dir || !*dir) return std::string();
    uint64_t h = fnv1a(src);
    h = fnv1a(expr, h);
    for (int i = 0; i < rdhip_embedded_count; ++i) h = fnv1a(rdhip_embedded_sources[i], h);      // the kernel templates themselves
    h = fnv1a(std::to_string(HIP_VERSION), h);
    for (const char* o : jit_opts) h = fnv1a(o, h);                                                  // the compile options
    char name[40]; snprintf(name, sizeof name, "/rdhip_%016llx.co", (unsigned long long)h);
    return std::string(dir) + name;
}
static bool cache_load(const std::string& path, std::string& lowered, std::vector<char>& code) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    uint32_t magic = 0, nl = 0; uint64_t nc = 0;
    if (fread(&magic, 4, 1, f) == 1 && magic == 0x52444843u && fread(&nl, 4, 1, f) == 1 && fread(&nc, 8, 1, f) == 1 && nl < 65536 && nc < (1ull << 30)) {
        lowered.resize(nl); code.resize(nc);
        ok = fread(&lowered[0], 1, nl, f) == nl && fread(code.data(), 1, nc, f) == nc;
    }
    fclose(f);
    return ok;
}
static void cache_store(const std::string& path, const std::string& lowered, const std::vector<char>& code) {
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                        // an unwritable cache directory is not an error
    const uint32_t magic = 0x52444843u, nl = (uint32_t)lowered.size(); const uint64_t nc = code.size();
    const bool ok = fwrite(&magic, 4, 1, f) == 1 && fwrite(&nl, 4, 1, f) == 1 && fwrite(&nc, 8, 1, f) == 1 && fwrite(lowered.data(), 1, nl, f) == nl && fwrite(code.data(), 1, nc, f) == nc;
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());       // rename: readers never see a partial file
}

// compile (no GPU needed): code object + lowered kernel name for (op, rule, dtype)
static int jit_compile(JitModel& M, int op, int rule, int dtype, const std::string& key) {
    if (M.code.count(key)) return RDHIP_OK;
    const std::string src = program_source(M), expr = kernel_expr(M, op, rule, dtype);
    const std::string cpath = cache_path(src, expr);
    if (!cpath.empty()) {
        std::string low; std::vector<char> code;
        if (cache_load(cpath, low, code)) { M.lowered[key] = low; M.code[key] = std::move(code); return RDHIP_OK; }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "rdhip_user_model.hip", rdhip_embedded_count, rdhip_embedded_sources, rdhip_embedded_names) != HIPRTC_SUCCESS) {
        M.err = "hiprtcCreateProgram failed"; return RDHIP_ERR_HIP;
    }
    hiprtcAddNameExpression(prog, expr.c_str());
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)(sizeof jit_opts / sizeof jit_opts[0]), jit_opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, ' '); if (ls) hiprtcGetProgramLog(prog, &log[0]);
        M.err = "hipRTC compilation of the user model failed:\n" + log.substr(0, 3000);
        hiprtcDestroyProgram(&prog);
        return RDHIP_ERR_INVALID_ARG;
    }
    const char* low = nullptr;
    if (hiprtcGetLoweredName(prog, expr.c_str(), &low) != HIPRTC_SUCCESS || !low) { M.err = "hiprtcGetLoweredName failed for " + expr; hiprtcDestroyProgram(&prog); return RDHIP_ERR_HIP; }
    M.lowered[key] = low;
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (!cpath.empty()) cache_store(cpath, M.lowered[key], code);
    M.code[key] = std::move(code);
    return RDHIP_OK;
}

static int jit_get(JitModel& M, int op_, int rule, int dtype, JitKernel** out) {
    std::lock_guard<std::mutex> g(M.mu);
    std::ostringstream k; k << op_ << ":" << rule << ":" << dtype;
    const std::string key = k.str();
    int rc = jit_compile(M, op_, rule, dtype, key); if (rc) return rc;
    const int op = op_ >= JOP_MULTI ? op_ - JOP_MULTI : op_;        // block size / LDS bytes: those of the single-batch kernel with the same body
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { M.err = "hipGetDevice failed"; return RDHIP_ERR_NO_DEVICE; }
    std::ostringstream kd; kd << key << "@" << dev;
    auto it = M.loaded.find(kd.str());
    if (it == M.loaded.end()) {
        JitKernel jk;
        if (hipModuleLoadData(&jk.mod, M.code[key].data()) != hipSuccess) { M.err = "hipModuleLoadData failed"; return RDHIP_ERR_HIP; }
        if (hipModuleGetFunction(&jk.fn, jk.mod, M.lowered[key].c_str()) != hipSuccess) { M.err = "hipModuleGetFunction failed for " + M.lowered[key]; return RDHIP_ERR_HIP; }
        const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const int nz = M.n + M.m, ny = M.n;
        jk.small = (op == JOP_JAC_AD && nz <= 10);
        const bool split = nz > 10 && (op == JOP_JAC_AD || op == JOP_JAC_CHAIN || op == JOP_JAC_FD || op == JOP_IMPLICIT_IFT);
        auto a16 = [](size_t b) { return (b + 15) / 16 * 16; };
        if (op == JOP_EVAL) jk.block = 256;
        else if ((op == JOP_ROLLOUT || op == JOP_IMPLICIT_ROLLOUT_CAS) && cascade_fits(M, op == JOP_IMPLICIT_ROLLOUT_CAS ? RDHIP_IMPLICIT_MIDPOINT : rule, dtype)) {
            // one wavefront per block + the IO wavefront
            jk.block = (M.cas_nb + 1) * 64; jk.lds = cascade_lds_bytes(M, rule_stage_count(op == JOP_IMPLICIT_ROLLOUT_CAS ? RDHIP_IMPLICIT_MIDPOINT : rule), dtype);
        }
        else if (op == JOP_ROLLOUT || op == JOP_ROLLOUT_PLAIN || op == JOP_IMPLICIT_ERR || (op >= JOP_IMPLICIT_X_REG && op <= JOP_IMPLICIT_ROLLOUT_REG)) { jk.block = 64; if (op == JOP_IMPLICIT_J_REG) jk.lds = (size_t)64 * ((4 * ny) | 1) * ts; }    // ImplicitRegOps::LDS_BYTES
        else if (op >= JOP_IMPLICIT_X && op <= JOP_IMPLICIT_ROLLOUT) {      // ImplicitLds<Mdl,T>: KP knot points per wave, BYTES
            const size_t lane_bytes = (size_t)((ny * ny + ny) | 1) * ts + 16 * 4;
            int kp = 64;
            if (2 * 64 * lane_bytes > 160 * 1024) { kp = (int)((80 * 1024) / lane_bytes); if (kp > 64) kp = 64; if (kp < 1) kp = 1; }
            jk.block = 64; jk.knots = kp; jk.lds = (size_t)kp * lane_bytes;
        }
        else if (op == JOP_ERREXP) {       // SplitLds<ErrFun<...>, T, ContiguousPlan<ne+m, ng>, 4>::BYTES
            const int ne = M.n - (int)M.rot_starts.size(), nc = ne + M.m, ng = nc >= 4 ? 4 : nc;
            jk.block = (ng + 4) * 64; jk.persistent = true;
            jk.lds = a16((size_t)64 * ne * nc * ts) + a16((size_t)64 * ny * ts) + a16((size_t)65 * nz * ts) + 2 * 64 * 8;
        }
        else if (jk.small) {             // SmallLds<Fn,T>: WAVES independent waves per workgroup, as many (4, 2, 1) as fit the LDS
            const size_t wave_bytes = a16((size_t)64 * ny * nz * ts) + a16((size_t)64 * ny * ts) + a16((size_t)64 * nz * ts);
            const int waves = 4 * wave_bytes <= 160 * 1024 ? 4 : 2 * wave_bytes <= 160 * 1024 ? 2 : 1;
            jk.block = 64 * waves; jk.persistent = true; jk.lds = waves * wave_bytes;
        }
        else if (op == JOP_JAC_USER || op == JOP_JAC_USER_CHAIN) {         // JacLds<Fn, T, ContiguousPlan<nz, 2>>::BYTES, two waves
            jk.block = 128; jk.persistent = true; jk.lds = ((size_t)64 * ((nz * ny + ny) | 1) + (size_t)64 * nz) * ts + 2 * 64 * 8;
        }
        else if (split) { jk.block = 512; jk.persistent = true; jk.lds = a16((size_t)64 * ny * nz * ts) + a16((size_t)64 * ny * ts) + a16((size_t)64 * nz * ts) + 2 * 64 * 8; }   // SplitLds<Fn,T,Plan,4>::BYTES
        else { jk.block = 64 * (nz >= 4 ? 4 : nz); jk.persistent = true; jk.lds = ((size_t)64 * ((nz * ny + ny) | 1) + (size_t)64 * nz) * ts + 2 * 64 * 8; }   // JacLds<Fn,T,ContiguousPlan>::BYTES
        if (jk.lds > 64 * 1024) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(jk.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jk.lds) != hipSuccess) {
                (void)hipGetLastError(); M.err = "model too large for the JIT tile kernels (LDS tile > 64 KB and the attribute call failed)"; return RDHIP_ERR_UNSUPPORTED;
            }
        }
        it = M.loaded.emplace(kd.str(), jk).first;
    }
    *out = &it->second;
    return RDHIP_OK;
}

// kernels compiled for an earlier plan / block structure are stale.  They are not unloaded: jit_get hands out JitKernel pointers that are launched
// after the model lock is released (e.g. by the host threads of rdhip_jacobian_sharded); the entries move to a retired key and their modules are
// unloaded with the model.  Caller holds M.mu.
static void retire_compiled(JitModel& M) {
    ++M.generation;
    std::vector<std::string> keys;
    for (auto& kv : M.loaded) if (kv.first.compare(0, 7, "retired") != 0) keys.push_back(kv.first);
    for (const std::string& k : keys) {          // re-key the map NODE (extract / insert keeps the JitKernel at its address)
        auto nh = M.loaded.extract(k);
        nh.key() = "retired" + std::to_string(M.generation) + ":" + k;
        M.loaded.insert(std::move(nh));
    }
    M.code.clear(); M.lowered.clear();
}

// Declare the block structure of a custom model (models.hpp Cascade<>): block b's derivative reads blocks 0..b and the controls whose
// control_block is <= b.  nb <= 1 removes the declaration.  The declaration is VERIFIED against the model on first GPU use (jit_verify_cascade).
int jit_set_cascade(void* h, int nb, const int* state_block, const int* control_block) {
    JitModel& M = *static_cast<JitModel*>(h);
    std::lock_guard<std::mutex> vg(M.vmu);
    std::lock_guard<std::mutex> g(M.mu);
    if (nb <= 1) { M.cas_nb = 0; M.cas_state.clear(); M.cas_control.clear(); M.cas_checked = false; retire_compiled(M); return RDHIP_OK; }
    if (nb > 6 || !state_block || !control_block) { M.err = "cascade: 2..6 blocks, non-NULL tables"; return RDHIP_ERR_INVALID_ARG; }
    std::vector<int> cnt(nb, 0);
    for (int i = 0; i < M.n; ++i) { if (state_block[i] < 0 || state_block[i] >= nb) { M.err = "cascade: state block index out of range"; return RDHIP_ERR_INVALID_ARG; } ++cnt[state_block[i]]; }
    for (int j = 0; j < M.m; ++j) if (control_block[j] < 0 || control_block[j] >= nb) { M.err = "cascade: control block index out of range"; return RDHIP_ERR_INVALID_ARG; }
    for (int b = 0; b < nb; ++b) if (cnt[b] == 0 || cnt[b] > 8) { M.err = "cascade: every block needs 1..8 state components"; return RDHIP_ERR_INVALID_ARG; }
    M.cas_nb = nb; M.cas_state.assign(state_block, state_block + M.n); M.cas_control.assign(control_block, control_block + M.m); M.cas_checked = false;
    retire_compiled(M);
    return RDHIP_OK;
}

// Column costs by compiling each column alone (RK4, fp64) and reading the size of the code object, then longest-processing-time-first
// assignment of the columns to the 4 compute waves of the split-role kernel; f(z) goes to the lightest wave.
int jit_balance(void* h) {
    JitModel& M = *static_cast<JitModel*>(h);
    std::lock_guard<std::mutex> g(M.mu);
    const int nz = M.n + M.m;
    if (nz <= 10) return RDHIP_OK;                       // small models use the all-columns-per-lane kernel
    const std::string src = program_source(M);
    std::vector<size_t> cost(nz, 0);
    for (int c = 0; c < nz; ++c) {
        std::ostringstream e; e << "rd::col_probe_kernel<rd::Fun<rd::UserModel<double>, 3>, double, " << c << ">";
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "rdhip_user_model.hip", rdhip_embedded_count, rdhip_embedded_sources, rdhip_embedded_names) != HIPRTC_SUCCESS) { M.err = "hiprtcCreateProgram failed"; return RDHIP_ERR_HIP; }
        hiprtcAddNameExpression(prog, e.str().c_str());
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-signed-zeros", "-ffinite-math-only", "-fno-slp-vectorize"};
        if (hiprtcCompileProgram(prog, 6, opts) != HIPRTC_SUCCESS) {
            size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, ' '); if (ls) hiprtcGetProgramLog(prog, &log[0]);
            M.err = "hipRTC compilation of the user model failed:\n" + log.substr(0, 3000); hiprtcDestroyProgram(&prog); return RDHIP_ERR_INVALID_ARG;
        }
        hiprtcGetCodeSize(prog, &cost[c]);
        hiprtcDestroyProgram(&prog);
    }
    size_t base = cost[0]; for (size_t v : cost) base = v < base ? v : base;        // fixed code-object overhead ~ the cheapest column
    std::vector<int> order(nz); for (int c = 0; c < nz; ++c) order[c] = c;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    // early part of a workgroup's first tile (TablePlanS, kernels.hpp): the prefix of columns that holds the most bytes for the least arithmetic
    // -- at least a quarter of a row, and only if it is worth a barrier (a satellite: r, q, v = 53 % of the bytes for 32 % of the work)
    size_t total = 0; for (int c = 0; c < nz; ++c) total += cost[c] - base + 64;
    int split = 0; double best = 0.10;
    { size_t acc = 0; for (int sp = 1; sp < nz; ++sp) { acc += cost[sp - 1] - base + 64; const double gain = (double)sp / nz - (double)acc / (double)total; if (4 * sp >= nz && gain > best) { best = gain; split = sp; } } }
    M.plan_split = split;
    size_t load[4] = {0, 0, 0, 0};
    M.plan_wave.assign(nz, 0);
    // longest-processing-time-first, the early columns first (so that the four waves reach the split together), the others on top of those loads
    for (int pass = 0; pass < 2; ++pass)
        for (int c : order) {
            if ((c < split) != (pass == 0)) continue;
            int w = 0; for (int i = 1; i < 4; ++i) if (load[i] < load[w]) w = i;
            M.plan_wave[c] = w; load[w] += cost[c] - base + 64;
        }
    int vw = 0; for (int i = 1; i < 4; ++i) if (load[i] < load[vw]) vw = i;
    M.plan_value_wave = vw;
    retire_compiled(M);
    return RDHIP_OK;
}

// ---- entry points used by api.hip ------------------------------------------------------------------------------------
void* jit_model_create(int n, int m, const char* body, const char* preamble, const double* params, int np, int quat_rotations, const int* P, int nP) {
    JitModel* M = new JitModel();
    if (quat_rotations && nP > 0) { int s0 = 0; for (int r = 0; r + 1 < nP; ++r) { s0 += P[r]; M->rot_starts.push_back(s0); s0 += 4; } }
    M->n = n; M->m = m; M->np = np; M->body = body ? body : ""; M->preamble = preamble ? preamble : "";
    M->params.assign(params, params + np);
    return M;
}
// a user-defined DiscreteDynamics model: `step_body` is the body of step(x, u, t, h, xn) (program_source)
void* jit_model_create_discrete(int n, int m, const char* step_body, const char* jac_body, const char* preamble, const double* params, int np, int quat_rotations, const int* P, int nP) {
    JitModel* M = static_cast<JitModel*>(jit_model_create(n, m, step_body, preamble, params, np, quat_rotations, P, nP));
    M->discrete = true; M->jac_body = jac_body ? jac_body : "";
    return M;
}
int jit_set_user_jacobian(void* h, const char* jac_body) {
    JitModel& M = *static_cast<JitModel*>(h);
    std::lock_guard<std::mutex> g(M.mu);
    M.jac_body = jac_body ? jac_body : "";
    retire_compiled(M);
    return RDHIP_OK;
}
int jit_is_discrete(void* h) { return h && static_cast<JitModel*>(h)->discrete ? 1 : 0; }
int jit_has_user_jacobian(void* h) { return h && !static_cast<JitModel*>(h)->jac_body.empty() ? 1 : 0; }
void jit_model_destroy(void* h) {
    JitModel* M = static_cast<JitModel*>(h);
    if (!M) return;
    for (auto& kv : M->loaded) if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);     // (retired entries included: they stay in `loaded` under a retired key)
    delete M;
}
const char* jit_last_error(void* h) { return static_cast<JitModel*>(h)->err.c_str(); }
static int user_jac_op(const JitModel& M, int rule, int diff);
int jit_precompile(void* h, int op, int rule, int dtype) {
    JitModel& M = *static_cast<JitModel*>(h);
    std::lock_guard<std::mutex> g(M.mu);
    if (M.discrete != (rule == RDHIP_DISCRETE)) { M.err = "precompile: RDHIP_DISCRETE is the rule of discrete models, and their only one"; return RDHIP_ERR_INVALID_ARG; }
    // "UserDefined" (the chain-rule op) of a model with a Jacobian body means the kernel that evaluates the body, where one exists (jit_jacobian_impl)
    if (op == JOP_JAC_CHAIN || op == JOP_MULTI + JOP_JAC_CHAIN) { const int u = user_jac_op(M, rule, RDHIP_USER_DEFINED); if (u >= 0) op = (op >= JOP_MULTI ? JOP_MULTI : 0) + u; }
    if ((op == JOP_JAC_USER || op == JOP_JAC_USER_CHAIN || op == JOP_MULTI + JOP_JAC_USER || op == JOP_MULTI + JOP_JAC_USER_CHAIN) && M.jac_body.empty()) { M.err = "precompile: the model has no Jacobian body"; return RDHIP_ERR_NOT_IMPLEMENTED; }
    std::ostringstream k; k << op << ":" << rule << ":" << dtype;
    return jit_compile(M, op, rule, dtype, k.str());
}

static void pack_params(const JitModel& M, int dtype, std::vector<unsigned char>& buf) {
    const int np = M.np > 0 ? M.np : 1;
    if (dtype == RDHIP_F64) { buf.resize(np * 8); std::vector<double> p(np, 0.0); for (int i = 0; i < M.np; ++i) p[i] = M.params[i]; std::memcpy(buf.data(), p.data(), np * 8); }
    else { buf.resize(np * 4); std::vector<float> p(np, 0.0f); for (int i = 0; i < M.np; ++i) p[i] = (float)M.params[i]; std::memcpy(buf.data(), p.data(), np * 4); }
}
static int launch(JitModel& M, JitKernel* k, unsigned grid, hipStream_t s, void** args) {
    rd::last_kernel() = rd::LastKernel{nullptr, k->fn, false};
    if (hipModuleLaunchKernel(k->fn, grid, 1, 1, k->block, 1, 1, (unsigned)k->lds, s, args, nullptr) != hipSuccess) { M.err = "hipModuleLaunchKernel failed"; return RDHIP_ERR_HIP; }
    return RDHIP_OK;
}
int jit_eval(void* h, int rule, int dtype, const LaunchArgs& a) {
    JitModel& M = *static_cast<JitModel*>(h);
    if (a.N == 0) return RDHIP_OK;
    JitKernel* k; int rc = jit_get(M, JOP_EVAL, rule, dtype, &k); if (rc) return rc;
    std::vector<unsigned char> fn; pack_params(M, dtype, fn);
    long long N = a.N; const void* Z = a.Z; const double* t = a.t; const double* dt = a.dt; void* Y = a.Y;
    void* args[] = {fn.data(), &N, &Z, &t, &dt, &Y};
    return launch(M, k, (unsigned)((N + 255) / 256), a.stream, args);
}
// A declared block structure is only a promise; the cascade kernels produce wrong numbers if it is false.  Before its first use on a GPU the
// continuous ForwardAD Jacobian is evaluated at 32 points and every entry that the declaration says is structurally zero -- a row of block a
// against a state column of a block > a, or against a control whose control_block is > a -- must be exactly zero (tangents propagate exact
// zeros; non-finite entries are skipped).
static int jit_jacobian_impl(JitModel& M, int rule, int diff, int dtype, const LaunchArgs& a);
static int jit_verify_cascade(JitModel& M, hipStream_t stream) {
    if (M.cas_nb <= 1) return RDHIP_OK;
    std::lock_guard<std::mutex> vg(M.vmu);                 // (M.mu is taken inside by jit_get; set_cascade resets cas_checked under both)
    if (M.cas_checked) return RDHIP_OK;
    // the check allocates, copies and synchronises: none of that may happen on a capturing stream (it would invalidate the capture)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        M.err = "the declared cascade has not been verified yet and the stream is capturing: run this call (or any rollout / Jacobian of the model) once before rdhip_graph_begin_capture";
        return RDHIP_ERR_INVALID_ARG;
    }
    (void)hipGetLastError();
    const int n = M.n, m = M.m, nz = n + m, NP = 32;
    std::vector<double> Z((size_t)NP * nz), t(NP, 0.3), dt(NP, 0.1), J((size_t)NP * n * nz);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (double& v : Z) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = 0.1 + 0.8 * (double)(st >> 11) / 9007199254740992.0; }
    void *dZ = nullptr, *dT = nullptr, *dD = nullptr, *dJ = nullptr;
    int rc = RDHIP_ERR_HIP;
    if (hipMalloc(&dZ, Z.size() * 8) == hipSuccess && hipMalloc(&dT, NP * 8) == hipSuccess && hipMalloc(&dD, NP * 8) == hipSuccess && hipMalloc(&dJ, J.size() * 8) == hipSuccess &&
        hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dT, t.data(), NP * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dD, dt.data(), NP * 8, hipMemcpyHostToDevice) == hipSuccess) {
        LaunchArgs a{}; a.stream = stream; a.N = NP; a.Z = dZ; a.t = (const double*)dT; a.dt = (const double*)dD; a.J = dJ;
        rc = jit_jacobian_impl(M, RDHIP_CONTINUOUS, RDHIP_FORWARD_AD, RDHIP_F64, a);
        if (rc == RDHIP_OK && (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(J.data(), dJ, J.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = RDHIP_ERR_HIP;
    } else M.err = "cascade check: device allocation failed";
    (void)hipFree(dZ); (void)hipFree(dT); (void)hipFree(dD); (void)hipFree(dJ);
    if (rc) return rc;
    for (int k = 0; k < NP; ++k)
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < nz; ++c) {
                const int col_block = c < n ? M.cas_state[c] : M.cas_control[c - n];
                const double v = J[((size_t)k * nz + c) * n + i];
                unsigned long long bits; std::memcpy(&bits, &v, 8);
                const bool finite = ((bits >> 52) & 0x7ff) != 0x7ff;          // (tested on the bit pattern: the build uses -ffinite-math-only)
                if (M.cas_state[i] < col_block && finite && (bits << 1) != 0) {
                    char msg[256];
                    snprintf(msg, sizeof msg, "the declared cascade is not one: d xdot[%d] / d %s[%d] = %g, but xdot[%d] is in block %d and that %s in block %d",
                             i, c < n ? "x" : "u", c < n ? c : c - n, v, i, M.cas_state[i], c < n ? "state" : "control enters", col_block);
                    M.err = msg;
                    return RDHIP_ERR_INVALID_ARG;
                }
            }
    M.cas_checked = true;
    return RDHIP_OK;
}
int jit_rollout(void* h, int rule, int dtype, const LaunchArgs& a) {
    JitModel& M = *static_cast<JitModel*>(h);
    if (a.N == 0 || a.K == 0) return RDHIP_OK;
    // the cascade kernel where the declaration fits it (cascade_fits) and the IO wavefront's 32-bit lane offsets reach (launch_rollout applies the same bound)
    const bool cas = cascade_fits(M, rule, dtype) && a.K < (1LL << 26) && a.N < (1LL << 26);
    if (cas) { const int vc = jit_verify_cascade(M, a.stream); if (vc) return vc; }
    JitKernel* k; int rc = jit_get(M, cas ? JOP_ROLLOUT : JOP_ROLLOUT_PLAIN, rule, dtype, &k); if (rc) return rc;
    std::vector<unsigned char> fn; pack_params(M, dtype, fn);
    long long K = a.K, Mt = a.N, sj = a.time_major ? 1 : a.K, sk = a.time_major ? a.N : 1;
    const void* Z = a.Z; const double* t = a.t; const double* dt = a.dt; const void* X0 = a.X0;
    void* args[] = {fn.data(), &K, &Mt, &sj, &sk, &Z, &t, &dt, &X0};
    return launch(M, k, (unsigned)((Mt + 63) / 64), a.stream, args);
}
int jit_jacobian(void* h, int rule, int diff, int dtype, const LaunchArgs& a) {
    JitModel& M = *static_cast<JitModel*>(h);
    if (a.N == 0 && !a.multi) return RDHIP_OK;
    if (M.cas_nb > 1) { const int vc = jit_verify_cascade(M, a.stream); if (vc) return vc; }
    return jit_jacobian_impl(M, rule, diff, dtype, a);
}
// RDHIP_USER_DEFINED with the user's own formulas (user_model.hpp), or -1: a discrete model's own jac; a continuous model's jac for rule CONTINUOUS and,
// through the integrators' chain rule on stage Jacobians held in registers, for the explicit rules of models with n + m <= 10
static int user_jac_op(const JitModel& M, int rule, int diff) {
    if (diff != RDHIP_USER_DEFINED || M.jac_body.empty()) return -1;
    if (M.discrete || rule == RDHIP_CONTINUOUS) return JOP_JAC_USER;
    return (M.n + M.m <= 10 && rule >= RDHIP_EULER && rule <= RDHIP_RK4) ? JOP_JAC_USER_CHAIN : -1;
}
static int jit_jacobian_impl(JitModel& M, int rule, int diff, int dtype, const LaunchArgs& a) {
    if (M.discrete && diff == RDHIP_USER_DEFINED && M.jac_body.empty()) { M.err = "User-defined discrete dynamics Jacobian not defined."; return RDHIP_ERR_NOT_IMPLEMENTED; }      // src/discrete_dynamics.jl:112-113
    const int uop = user_jac_op(M, rule, diff);
    int op = uop >= 0 ? uop : diff == RDHIP_FORWARD_AD ? JOP_JAC_AD : diff == RDHIP_FINITE_DIFF ? JOP_JAC_FD : JOP_JAC_CHAIN;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.J) | reinterpret_cast<uintptr_t>(a.Y)) & 15) == 0;
    if (uop < 0 && M.n + M.m > 10 && !aligned) op = (op == JOP_JAC_AD) ? JOP_JAC_AD_LOOP : (op == JOP_JAC_CHAIN ? JOP_JAC_CHAIN_LOOP : JOP_JAC_FD_LOOP);   // the split-role kernels copy 16-B units
    if (a.multi) {                    // rdhip_jacobian_multi: the same body over a MultiBatch (kernel argument = the MultiPod image)
        if (a.multi->pod.total == 0) return RDHIP_OK;
        int mop = uop >= 0 ? uop : diff == RDHIP_FORWARD_AD ? JOP_JAC_AD : diff == RDHIP_FINITE_DIFF ? JOP_JAC_FD : JOP_JAC_CHAIN;
        if (uop < 0 && M.n + M.m > 10 && !a.multi->aligned16) mop = (mop == JOP_JAC_AD) ? JOP_JAC_AD_LOOP : (mop == JOP_JAC_CHAIN ? JOP_JAC_CHAIN_LOOP : JOP_JAC_FD_LOOP);
        JitKernel* km; int rcm = jit_get(M, JOP_MULTI + mop, rule, dtype, &km); if (rcm) return rcm;
        std::vector<unsigned char> fnm; pack_params(M, dtype, fnm);
        MultiPod pod = a.multi->pod;
        unsigned* msched = a.sched;                    // (the split-role multi kernel's last parameter; the tile multi kernel has none and ignores the entry)
        void* margs[] = {fnm.data(), &pod, &msched};
        const int wm = km->block / 64;
        const unsigned gridm = km->small ? tile_grid(km->lds, km->block, (pod.total + wm - 1) / wm) : tile_grid(km->lds, km->block, pod.total);
        return launch(M, km, gridm, a.stream, margs);
    }
    JitKernel* k; int rc = jit_get(M, op, rule, dtype, &k); if (rc) return rc;
    std::vector<unsigned char> fn; pack_params(M, dtype, fn);
    long long N = a.N; const void* Z = a.Z; const double* t = a.t; const double* dt = a.dt; void* J = a.J; void* Y = a.Y;
    unsigned* sched = a.sched;                           // (read by the split-role kernels only; the others have no such parameter and ignore the entry)
    void* args[] = {fn.data(), &N, &Z, &t, &dt, &J, &Y, &sched};
    const long long ntiles = (N + 63) / 64;
    unsigned grid;
    if (k->small) { const int w = k->block / 64; grid = tile_grid(k->lds, k->block, (ntiles + w - 1) / w); }      // persistent: one tile per wave and pass
    else grid = tile_grid(k->lds, k->block, ntiles);   // persistent tile loop: as many workgroups as fit at once, never more than tiles
    return launch(M, k, grid, a.stream, args);
}

// fused error-state expansion of a custom LieState{QuatRotation,P} model (dispatch.hpp launch_errexp)
int jit_errexp(void* h, int rule, int dtype, const LaunchArgs& a) {
    JitModel& M = *static_cast<JitModel*>(h);
    if (a.N == 0) return RDHIP_OK;
    if (M.rot_starts.empty()) { M.err = "error-state expansion kernel: the model has no quaternion"; return RDHIP_ERR_INVALID_ARG; }
    JitKernel* k; int rc = jit_get(M, JOP_ERREXP, rule, dtype, &k); if (rc) return rc;
    std::vector<unsigned char> fn; pack_params(M, dtype, fn);
    long long N = a.N; const void* Z = a.Z; const double* t = a.t; const double* dt = a.dt; void* E = a.J; void* Y = a.Y;
    unsigned* sched = a.sched;
    void* args[] = {fn.data(), &N, &Z, &t, &dt, &E, &Y, &sched};
    const long long ntiles = (N + 63) / 64;
    return launch(M, k, tile_grid(k->lds, k->block, ntiles), a.stream, args);
}

// DiscretizedDynamics{ImplicitMidpoint}(custom model): mode as in dispatch.hpp launch_implicit
int jit_implicit(void* h, int mode, int dtype, const LaunchArgs& a) {
    JitModel& M = *static_cast<JitModel*>(h);
    if (a.N == 0) return RDHIP_OK;
    if (M.cas_nb > 1) { const int vc = jit_verify_cascade(M, a.stream); if (vc) return vc; }
    const int op = mode == 0 ? JOP_IMPLICIT_X : mode == 1 ? JOP_IMPLICIT_J : mode == 2 ? JOP_IMPLICIT_ERR : JOP_IMPLICIT_ROLLOUT;
    JitKernel* k; int rc = jit_get(M, op, RDHIP_IMPLICIT_MIDPOINT, dtype, &k); if (rc) return rc;
    std::vector<unsigned char> fn; pack_params(M, dtype, fn);
    long long N = a.N, K = a.K; const void* Z = a.Z; const double* t = a.t; const double* dt = a.dt; void* Y = a.Y; void* J = a.J; void* J2 = a.J2; const void* X0 = a.X0;
    int iters = a.newton_iters; double tol = a.newton_tol; void* nul = nullptr;
    const long long nblk = (N + k->knots - 1) / k->knots;
    const unsigned grid = (unsigned)(mode == 2 ? (N + 63) / 64 : nblk);
    if (mode == 2) { void* args[] = {fn.data(), &N, &Z, &t, &dt, &Y, &J2, &J}; return launch(M, k, grid, a.stream, args); }
    // n <= 14: the register kernel first (exact for n <= 6, where it pivots in registers); the LDS kernel behind it gated on the flags
    const int* gate = nullptr;
    const unsigned gr = (unsigned)((N + 63) / 64);
    int tile_flags = (mode != 3 && (long long)gr <= IMPLICIT_FLAG_WORDS) ? 1 : 0;      // one word per 64-knot tile, written unconditionally
    // mode 1, models whose Jacobian tile fits the split-role loop: Newton solve (register kernel) -> IFT Jacobian as whole-line tile stores ->
    // the exact LDS kernel for flagged tiles (dispatch.hpp launch_implicit does the same for the compiled-in models)
    {
        const size_t ts = dtype == RDHIP_F64 ? 8 : 4; const int nz = M.n + M.m;
        auto a16 = [](size_t b) { return (b + 15) / 16 * 16; };
        const size_t lds_ift = a16((size_t)64 * M.n * nz * ts) + a16((size_t)64 * M.n * ts) + a16((size_t)64 * nz * ts) + 2 * 64 * 8;
        if (mode == 1 && a.flag && M.n > 6 && M.n <= 14 && nz > 10 && lds_ift <= 160 * 1024 && (reinterpret_cast<uintptr_t>(a.J) & 15) == 0) {
            JitKernel *kr, *ki;
            rc = jit_get(M, JOP_IMPLICIT_IFT, RDHIP_IMPLICIT_MIDPOINT, dtype, &ki); if (rc) return rc;
            int* flag = a.flag;
            if (tile_flags && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && tune_knobs().ift != '3') {
                // ONE launch (dispatch.hpp launch_implicit, implicit.hpp IftBody): Newton inside the tile kernel, x_next through its f(z) slab, flagged tiles
                // redone by their own workgroup behind the loop.  IftFun<UserModel<T>> = { UserModel m; double tol; int* flag; int iters; int solve; const T* xn; }
                std::vector<unsigned char> one(fn);
                one.resize((one.size() + 7) / 8 * 8);
                const size_t o = one.size();
                one.resize(o + 32, 0);
                const int solve1 = (lds_ift + 2 * (((size_t)64 * M.n * ts + 16 + 15) / 16 * 16) <= 160 * 1024) ? 2 : 1;       // IftBody::HELPER: the helper wave's double buffer fits next to the tile
                std::memcpy(one.data() + o, &tol, 8); std::memcpy(one.data() + o + 8, &flag, 8); std::memcpy(one.data() + o + 16, &iters, 4); std::memcpy(one.data() + o + 20, &solve1, 4);
                unsigned* sched = nullptr;
                void* ar[] = {one.data(), &N, &Z, &t, &dt, &J, &Y, &sched};
                return launch(M, ki, tile_grid(ki->lds, ki->block, (N + 63) / 64), a.stream, ar);
            }
            rc = jit_get(M, JOP_IMPLICIT_X_REG, RDHIP_IMPLICIT_MIDPOINT, dtype, &kr); if (rc) return rc;
            if (!tile_flags && hipMemsetAsync(flag, 0, sizeof(int), a.stream) != hipSuccess) { M.err = "hipMemsetAsync failed"; return RDHIP_ERR_HIP; }
            int tf2 = tile_flags | 2;
            { void* ar[] = {fn.data(), &N, &Z, &t, &dt, &Y, (void*)&nul, &iters, &tol, &flag, &tf2}; rc = launch(M, kr, gr, a.stream, ar); if (rc) return rc; }
            // IftFun<UserModel<T>> = { UserModel m; double tol; int* flag; int iters; int solve; const T* xn; } with the natural alignment of its members
            std::vector<unsigned char> ifn(fn);
            ifn.resize((ifn.size() + 7) / 8 * 8);
            const size_t off = ifn.size();
            ifn.resize(off + 32, 0);
            const void* xn = Y; const int solve0 = 0;
            int* iflag = tile_flags ? flag : nullptr;        // (per-tile words only)
            std::memcpy(ifn.data() + off, &tol, 8); std::memcpy(ifn.data() + off + 8, &iflag, 8); std::memcpy(ifn.data() + off + 16, &iters, 4);
            std::memcpy(ifn.data() + off + 20, &solve0, 4); std::memcpy(ifn.data() + off + 24, &xn, 8);
            { unsigned* sched = nullptr; void* ar[] = {ifn.data(), &N, &Z, &t, &dt, &J, (void*)&nul, &sched}; rc = launch(M, ki, tile_grid(ki->lds, ki->block, (N + 63) / 64), a.stream, ar); if (rc) return rc; }
            const int* gate2 = flag;
            void* args[] = {fn.data(), &N, &Z, &t, &dt, &Y, (void*)&J, &iters, &tol, &gate2, &tile_flags};
            return launch(M, k, tile_grid(k->lds, 64, nblk), a.stream, args);
        }
    }
    // mode 3 with a declared cascade: the rollout along the cascade (dispatch.hpp launch_implicit does the same for the compiled-in rigid bodies), then
    // the exact kernel gated on the flag word
    if (mode == 3 && a.flag && cascade_fits(M, RDHIP_IMPLICIT_MIDPOINT, dtype) && K < (1LL << 26) && N < (1LL << 26)) {
        if (K == 0) return RDHIP_OK;
        JitKernel* kc; rc = jit_get(M, JOP_IMPLICIT_ROLLOUT_CAS, RDHIP_IMPLICIT_MIDPOINT, dtype, &kc); if (rc) return rc;
        int* flag = a.flag;
        if (hipMemsetAsync(flag, 0, sizeof(int), a.stream) != hipSuccess) { M.err = "hipMemsetAsync failed"; return RDHIP_ERR_HIP; }
        // ImplicitRollFun<UserModel<T>> = { UserModel m; int iters; double tol; int* flag; } with the natural alignment of its members
        std::vector<unsigned char> rf(fn);
        rf.resize((rf.size() + 3) / 4 * 4);
        const size_t o_it = rf.size();
        rf.resize((o_it + 4 + 7) / 8 * 8);
        const size_t o_tol = rf.size();
        rf.resize(o_tol + 16, 0);
        std::memcpy(rf.data() + o_it, &iters, 4); std::memcpy(rf.data() + o_tol, &tol, 8); std::memcpy(rf.data() + o_tol + 8, &flag, 8);
        long long sj = K, sk = 1;
        { void* ar[] = {rf.data(), &K, &N, &sj, &sk, &Z, &t, &dt, &X0}; rc = launch(M, kc, (unsigned)((N + 63) / 64), a.stream, ar); if (rc) return rc; }
        const int* gate3 = flag;
        void* args[] = {fn.data(), &K, &N, &Z, &t, &dt, &X0, &iters, &tol, &gate3};
        return launch(M, k, grid, a.stream, args);
    }
    if (a.flag && M.n <= 14) {
        JitKernel* kr; rc = jit_get(M, mode == 0 ? JOP_IMPLICIT_X_REG : mode == 1 ? JOP_IMPLICIT_J_REG : JOP_IMPLICIT_ROLLOUT_REG, RDHIP_IMPLICIT_MIDPOINT, dtype, &kr); if (rc) return rc;
        int* flag = a.flag;
        if (M.n > 6 && !tile_flags && hipMemsetAsync(flag, 0, sizeof(int), a.stream) != hipSuccess) { M.err = "hipMemsetAsync failed"; return RDHIP_ERR_HIP; }
        if (mode == 3) { if (K == 0) return RDHIP_OK; void* ar[] = {fn.data(), &K, &N, &Z, &t, &dt, &X0, &iters, &tol, &flag}; rc = launch(M, kr, gr, a.stream, ar); }
        else { void* ar[] = {fn.data(), &N, &Z, &t, &dt, &Y, mode == 1 ? (void*)&J : (void*)&nul, &iters, &tol, &flag, &tile_flags}; rc = launch(M, kr, gr, a.stream, ar); }
        if (rc || M.n <= 6) return rc;
        gate = flag;
    }
    if (mode == 3) { if (K == 0) return RDHIP_OK; void* args[] = {fn.data(), &K, &N, &Z, &t, &dt, &X0, &iters, &tol, &gate}; return launch(M, k, grid, a.stream, args); }
    void* args[] = {fn.data(), &N, &Z, &t, &dt, &Y, mode == 1 ? (void*)&J : (void*)&nul, &iters, &tol, &gate, &tile_flags};
    return launch(M, k, tile_grid(k->lds, 64, nblk), a.stream, args);
}

}  // namespace rd
