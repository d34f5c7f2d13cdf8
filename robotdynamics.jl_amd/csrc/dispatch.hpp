// dispatch.hpp -- run-time (rule, diff, dtype) -> kernel template instantiation.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace rd {

inline int hip_status(hipError_t e) { return e == hipSuccess ? RDHIP_OK : RDHIP_ERR_HIP; }

// which (model, rule) pairs get the compile-time-column ForwardAD kernel (big code, structural sparsity folded)
template <class Mdl, int RULE> struct UnrollPolicy { static constexpr bool value = false; static constexpr int KW = 1; };

template <class Fn> int launch_eval(const Fn& fn, const LaunchArgs& a) {
    using T = typename Fn::T;
    if (a.N == 0) return RDHIP_OK;
    const unsigned grid = (unsigned)((a.N + 255) / 256);
    hipLaunchKernelGGL((eval_kernel<Fn, T>), dim3(grid), dim3(256), 0, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y);
    return hip_status(hipGetLastError());
}
template <class Fn> int launch_rollout(const Fn& fn, const LaunchArgs& a) {
    using T = typename Fn::T;
    if (a.N == 0 || a.K == 0) return RDHIP_OK;
    const unsigned grid = (unsigned)((a.N + 63) / 64);
    hipLaunchKernelGGL((rollout_kernel<Fn, T>), dim3(grid), dim3(64), 0, a.stream, fn, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
    return hip_status(hipGetLastError());
}
template <class Mdl, int RULE> int launch_jacobian(const Mdl& m, int diff, const LaunchArgs& a) {
    using Fn = Fun<Mdl, RULE>;
    using T = typename Fn::T;
    constexpr int NZ = Fn::NX + Fn::NU;
    constexpr int NG = NZ >= 4 ? 4 : NZ;
    if (a.N == 0) return RDHIP_OK;
    Fn fn{m};
    const unsigned grid = (unsigned)((a.N + TILE - 1) / TILE);
    const dim3 block(NG * TILE);
    if (diff == RDHIP_FORWARD_AD) {
        if constexpr (UnrollPolicy<Mdl, RULE>::value)
            hipLaunchKernelGGL((jac_ad_unrolled_kernel<Fn, T, NG, UnrollPolicy<Mdl, RULE>::KW>), dim3(grid), block, 0, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
        else
            hipLaunchKernelGGL((jac_ad_loop_kernel<Fn, T, NG>), dim3(grid), block, 0, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
    } else if (diff == RDHIP_FINITE_DIFF) {
        hipLaunchKernelGGL((jac_fd_kernel<Fn, T, NG>), dim3(grid), block, 0, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
    } else {
        return RDHIP_ERR_NOT_IMPLEMENTED;
    }
    return hip_status(hipGetLastError());
}

template <class Mdl> int dispatch_eval_rule(const Mdl& m, int rule, const LaunchArgs& a) {
    switch (rule) {
        case RDHIP_CONTINUOUS: return launch_eval(Fun<Mdl, RULE_CONTINUOUS>{m}, a);
        case RDHIP_EULER: return launch_eval(Fun<Mdl, RULE_EULER>{m}, a);
        case RDHIP_RK2: return launch_eval(Fun<Mdl, RULE_RK2>{m}, a);
        case RDHIP_RK3: return launch_eval(Fun<Mdl, RULE_RK3>{m}, a);
        case RDHIP_RK4: return launch_eval(Fun<Mdl, RULE_RK4>{m}, a);
    }
    return RDHIP_ERR_INVALID_ARG;
}
template <class Mdl> int dispatch_rollout_rule(const Mdl& m, int rule, const LaunchArgs& a) {
    switch (rule) {
        case RDHIP_EULER: return launch_rollout(Fun<Mdl, RULE_EULER>{m}, a);
        case RDHIP_RK2: return launch_rollout(Fun<Mdl, RULE_RK2>{m}, a);
        case RDHIP_RK3: return launch_rollout(Fun<Mdl, RULE_RK3>{m}, a);
        case RDHIP_RK4: return launch_rollout(Fun<Mdl, RULE_RK4>{m}, a);
    }
    return RDHIP_ERR_INVALID_ARG;
}
template <class Mdl> int dispatch_jacobian_rule(const Mdl& m, int rule, int diff, const LaunchArgs& a) {
    switch (rule) {
        case RDHIP_CONTINUOUS: return launch_jacobian<Mdl, RULE_CONTINUOUS>(m, diff, a);
        case RDHIP_EULER: return launch_jacobian<Mdl, RULE_EULER>(m, diff, a);
        case RDHIP_RK2: return launch_jacobian<Mdl, RULE_RK2>(m, diff, a);
        case RDHIP_RK3: return launch_jacobian<Mdl, RULE_RK3>(m, diff, a);
        case RDHIP_RK4: return launch_jacobian<Mdl, RULE_RK4>(m, diff, a);
    }
    return RDHIP_ERR_INVALID_ARG;
}

// dtype dispatch for a model class template MT<T>
template <template <class> class MT> int dispatch_eval(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return dispatch_eval_rule(MT<double>::make(md.params), rule, a);
    if (dtype == RDHIP_F32) return dispatch_eval_rule(MT<float>::make(md.params), rule, a);
    return RDHIP_ERR_INVALID_ARG;
}
template <template <class> class MT> int dispatch_rollout(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return dispatch_rollout_rule(MT<double>::make(md.params), rule, a);
    if (dtype == RDHIP_F32) return dispatch_rollout_rule(MT<float>::make(md.params), rule, a);
    return RDHIP_ERR_INVALID_ARG;
}
template <template <class> class MT> int dispatch_jacobian(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return dispatch_jacobian_rule(MT<double>::make(md.params), rule, diff, a);
    if (dtype == RDHIP_F32) return dispatch_jacobian_rule(MT<float>::make(md.params), rule, diff, a);
    return RDHIP_ERR_INVALID_ARG;
}

}  // namespace rd
