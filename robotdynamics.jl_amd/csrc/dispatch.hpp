// dispatch.hpp -- run-time (rule, diff, dtype) -> kernel template instantiation.
#pragma once
#include "common.hpp"
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "kernels.hpp"
#include "cascade.hpp"
#include "implicit.hpp"
#include "fused.hpp"
#include "rollout_stream.hpp"
#include "rollout_split.hpp"

namespace rd {

inline int hip_status(hipError_t e) { return e == hipSuccess ? RDHIP_OK : RDHIP_ERR_HIP; }

// which (model, rule) pairs get the compile-time-column ForwardAD kernel (big code, structural sparsity folded)
template <class Mdl, int RULE> struct UnrollPolicy { static constexpr bool value = false; static constexpr int NC = 0; using Plan = void; };

// LEAN variants of a model family: they keep every operator (evaluation, the three diff methods on the split-role tile loop, error-state expansion,
// implicit rules, rollouts) but not the forms that exist only to shave latency off special shapes -- the one-launch rollout -> Jacobians kernel
// (the caller's two launches instead) and the four-workgroups-per-tile kernel for batches below 8 192 knot points (the tile loop instead).  Round 6:
// the full-inertia rigid bodies (fam_rigid.hpp; the reference's own models all carry Diagonal inertias) -- those two forms were 12 MB of their code.
template <class Mdl> struct LeanVariant { static constexpr bool value = false; };

// does (model, rule) have the one-launch rollout -> Jacobians kernel (fused.hpp)?  A cascade structure, a four-wave compile-time column plan with four copy
// waves, and not a lean variant
template <class Mdl, int RULE, class = void> struct FusedFormOf { static constexpr bool value = false; };
template <class Mdl, int RULE> struct FusedFormOf<Mdl, RULE, typename std::enable_if<(Cascade<Mdl>::NB > 1) && UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC == 4 && !LeanVariant<Mdl>::value>::type> {
    static constexpr bool value = UnrollPolicy<Mdl, RULE>::Plan::NG == 4;
};

// column plan of the implicit-function-theorem tile kernel: the model's ForwardAD plan if it has one (the per-column costs have the same
// ordering: a column costs what it can reach downstream), contiguous ranges otherwise
template <class Mdl, class = void> struct IftPlanOf { using Plan = ContiguousPlan<Mdl::NX + Mdl::NU, 4>; };
template <class Mdl> struct IftPlanOf<Mdl, typename std::enable_if<UnrollPolicy<Mdl, RULE_RK4>::value>::type> { using Plan = typename UnrollPolicy<Mdl, RULE_RK4>::Plan; };

// cost-balanced column plan of the cascade FiniteDifference kernel (defined next to the model's other plans)
template <class Mdl> struct FdPlanOf { static constexpr bool value = false; using Plan = void; };

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class Fn> int launch_eval(const Fn& fn, const LaunchArgs& a) {
    using T = typename Fn::T;
    if (a.N == 0) return RDHIP_OK;
    const unsigned grid = (unsigned)((a.N + 255) / 256);
    RD_LAUNCH((eval_kernel<Fn, T>), dim3(grid), dim3(256), 0, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y);
    return hip_status(hipGetLastError());
}
// rollout! on split arrays (rdhip_rollout_split): one lane per trajectory, both layouts by strides (rollout_split.hpp)
template <class Fn> int launch_rollout_split(const Fn& fn, const LaunchArgs& a) {
    using T = typename Fn::T;
    if (a.N == 0 || a.K < 2) return RDHIP_OK;                           // (K < 2: nothing to simulate; api.hip copies the initial state)
    const long long K = a.K, M = a.N;
    const bool tm = a.time_major != 0;
    RD_LAUNCH((rollout_split_lane_kernel<Fn, T>), dim3((unsigned)((M + 63) / 64)), dim3(64), 0, a.stream, fn, K, M, tm ? 1LL : K, tm ? M : 1LL, tm ? 1LL : K - 1, tm ? M : 1LL,
              tm ? 1LL : K, tm ? M : 1LL, (T*)a.Z, (const T*)a.U, a.t, a.dt, (const T*)a.X0);
    return hip_status(hipGetLastError());
}
template <class Fn> int launch_rollout(const Fn& fn, const LaunchArgs& a) {
    using T = typename Fn::T;
    if (a.N == 0 || a.K == 0) return RDHIP_OK;
    if (a.U) return launch_rollout_split(fn, a);
    const unsigned grid = (unsigned)((a.N + 63) / 64);
    const long long sj = a.time_major ? 1 : a.K, sk = a.time_major ? a.N : 1;
    using Cas = Cascade<decltype(Fn::m)>;
    const char roll_env = tune_knobs().rollout;                            // tuning switch: 'p'lain / 'c'ascade / 's'tream force a kernel where it applies (0: by size)
    if constexpr (Fn::RULE_ID != RULE_IMPLICIT_MIDPOINT) {
        // the throughput regime (rollout_stream.hpp): at least two 64-trajectory groups per CU.  Chunks move as whole 16-byte units, so every run of
        // the stack must start on a 16-byte boundary; other shapes take the one-lane-per-trajectory kernel there (the cascade kernel buys latency, not
        // traffic: 0.30-0.36 of the roofline at 65 536 x 32 / 262 144 x 16 where the plain kernel reaches 0.37-0.44 and the streaming one 0.41-0.49)
        const bool forced = roll_env == 's';
        if ((forced || (!roll_env && (long long)grid >= 2LL * dev_info().cus)) && a.K >= 1) {
            typedef unsigned int V4 __attribute__((ext_vector_type(4)));
            constexpr int ZB = (Fn::NX + Fn::NU) * (int)sizeof(T);
            // (only where a knot point is big enough for the chunk traffic to dominate: with 76-byte fp32 rigid-body or 40-byte cartpole knot points the
            // one compute wave per group and the two barriers per chunk are the bottleneck -- measured 0.32-0.34 against 0.46-0.48 for the one-lane
            // kernel on fp32 satellite stacks, 0.16 against 0.25-0.49 for the cartpole)
            const bool al = ZB >= 128 && aligned16(a.Z) && ((a.time_major ? a.N : a.K) * ZB) % 16 == 0;
            if (al && a.time_major) {
                using L = StreamCfg<Fn, T, true>;
                if constexpr (L::UNITS && ZB >= 128) {
                    auto ks = rollout_stream_kernel<Fn, T, true, V4>;
                    if (ensure_dynamic_lds(reinterpret_cast<const void*>(ks), L::BYTES) != hipSuccess) return RDHIP_ERR_HIP;
                    RD_LAUNCH(ks, dim3(grid), dim3(STREAM_THREADS), L::BYTES, a.stream, fn, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
                    return hip_status(hipGetLastError());
                }
            } else if (al) {
                using L = StreamCfg<Fn, T, false>;
                if constexpr (L::UNITS && ZB >= 128) {
                    auto ks = rollout_stream_kernel<Fn, T, false, V4>;
                    if (ensure_dynamic_lds(reinterpret_cast<const void*>(ks), L::BYTES) != hipSuccess) return RDHIP_ERR_HIP;
                    RD_LAUNCH(ks, dim3(grid), dim3(STREAM_THREADS), L::BYTES, a.stream, fn, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
                    return hip_status(hipGetLastError());
                }
            }
            if (!forced) {
                RD_LAUNCH((rollout_kernel<Fn, T>), dim3(grid), dim3(64), 0, a.stream, fn, a.K, a.N, sj, sk, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
                return hip_status(hipGetLastError());
            }
        }
    }
    if constexpr (Cas::NB > 1) {
        // one wavefront per cascade block + an IO wavefront (cascade.hpp): the latency chain of a step shrinks to the longest block's
        const bool plain = roll_env == 'p';
        if (!plain && a.K < (1LL << 26) && a.N < (1LL << 26)) {            // (the IO wave's per-lane offsets are 32-bit)
            constexpr int R = 4;
            // models whose rollout + Jacobians can run as one launch (fused.hpp) roll out through THAT kernel (rollout workgroups only): the two-call
            // and the one-call path then execute the same machine code and agree to the last bit
            if constexpr (Fn::RULE_ID != RULE_IMPLICIT_MIDPOINT && FusedFormOf<typename std::remove_cv<decltype(Fn::m)>::type, Fn::RULE_ID>::value) {
                using Mdl = typename std::remove_cv<decltype(Fn::m)>::type;
                using Plan = typename UnrollPolicy<Mdl, Fn::RULE_ID>::Plan;
                // ... while the rollout groups leave half of the chip free (the regime the one-launch form exists for: a latency chain on a few CUs).
                // Between that and the streaming regime above (8 192 .. 32 768 trajectories) every SIMD has a trajectory wave of its own and the
                // one-lane kernel below is as fast as the cascade kernel was (rollout_stream.hpp: 0.6 of the roofline at 16 384 x 64) -- round 6 dropped
                // the cascade kernel's instantiations for these models (2 MB of code for that one window).
                if (2LL * grid <= (long long)dev_info().cus) {
                    constexpr size_t ldsf = FusedLds<Fn, T, Cas, R, Plan, 4>::BYTES;
                    auto kf = rollout_jac_fused_kernel<Fn, T, Cas, R, Plan, 4>;
                    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kf), ldsf) != hipSuccess) return RDHIP_ERR_HIP;
                    RD_LAUNCH(kf, dim3(grid), dim3(FUSED_THREADS), ldsf, a.stream, fn, a.K, a.N, (int)grid, sj, sk, (T*)a.Z, a.t, a.dt, (const T*)a.X0,
                                       (T*)nullptr, (unsigned long long*)nullptr, 0ULL, 0LL);
                    return hip_status(hipGetLastError());
                }
                if (roll_env == 'c') return RDHIP_ERR_UNSUPPORTED;          // (tuning switch rollout=cascade: no such kernel for this model)
            } else {
                constexpr size_t lds = CascadeLds<Fn, Cas, R>::BYTES;
                auto kern = rollout_cascade_kernel<Fn, T, Cas, R>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                RD_LAUNCH(kern, dim3(grid), dim3((Cas::NB + 1) * TILE), lds, a.stream, fn, a.K, a.N, sj, sk, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
                return hip_status(hipGetLastError());
            }
        }
    }
    RD_LAUNCH((rollout_kernel<Fn, T>), dim3(grid), dim3(64), 0, a.stream, fn, a.K, a.N, sj, sk, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
    return hip_status(hipGetLastError());
}
// (tile_grid: common.hpp -- persistent grids are sized from the device's reported CU count and LDS size)

template <class Fn, class T, class Plan, class K> int launch_jac_kernel(K kernel, const Fn& fn, const LaunchArgs& a) {
    constexpr size_t lds = JacLds<Fn, T, Plan>::BYTES;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS of a gfx950 CU");
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return RDHIP_ERR_HIP;
    const long long ntiles = (a.N + TILE - 1) / TILE;
    RD_LAUNCH(kernel, dim3(tile_grid_occ(reinterpret_cast<const void*>(kernel), lds, Plan::NG * TILE, ntiles)), dim3(Plan::NG * TILE), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
    return hip_status(hipGetLastError());
}

template <class Fn, class T, class Plan, int NC, class K, class KD> int launch_split_kernel(K kernel, KD kernel_dyn, const Fn& fn, const LaunchArgs& a) {
    constexpr size_t lds = SplitLds<Fn, T, Plan, NC>::BYTES;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS of a gfx950 CU");
    const long long ntiles = (a.N + TILE - 1) / TILE;
    constexpr int threads = (Plan::NG + NC) * TILE;
    // the instantiation with the dynamic tile schedule (kernels.hpp jac_split_loop) when the launch has counter words and 8 or more tiles per workgroup
    {   // (the two are pointers of one type: the same pointer twice means "no such instantiation")
        if (a.sched && reinterpret_cast<const void*>(kernel_dyn) != reinterpret_cast<const void*>(kernel)) {
            const unsigned gd = dyn_grid(tile_grid_occ(reinterpret_cast<const void*>(kernel_dyn), lds, threads, ntiles));
            if (ntiles >= 8LL * gd) {
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel_dyn), lds) != hipSuccess) return RDHIP_ERR_HIP;
                RD_LAUNCH(kernel_dyn, dim3(gd), dim3(threads), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y, a.sched);
                last_kernel().dyn = true;
                return hip_status(hipGetLastError());
            }
        }
    }
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return RDHIP_ERR_HIP;
    RD_LAUNCH(kernel, dim3(tile_grid_occ(reinterpret_cast<const void*>(kernel), lds, threads, ntiles)), dim3(threads), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y, (unsigned*)nullptr);
    return hip_status(hipGetLastError());
}
template <class Fn, class T, class Plan, int NC, class K> int launch_split_kernel(K kernel, const Fn& fn, const LaunchArgs& a) {
    return launch_split_kernel<Fn, T, Plan, NC>(kernel, kernel, fn, a);
}
// Bodies other than the ForwardAD flagship carry the dynamic tile schedule in their ONE fp64 instantiation (DYN = true; launches below 8 tiles per
// workgroup pass no counter words and the loop takes its static order at run time).  Round 5 shipped a second, schedule-free instantiation of each for
// those launches: the dormant schedule costs them 0.3-0.5 us of ~28 (kernels.hpp SingleBatchDyn), the second copies cost 4.5 MB of code and nothing
// ever measured or tested most of them (profiles/r05/kernel_reach.txt).  The flagship keeps both (jac_ad_split_kernel).
template <class Fn, class T, class Plan, int NC, class KD> int launch_split_kernel_rt(KD kernel_dyn, const Fn& fn, const LaunchArgs& a) {
    constexpr size_t lds = SplitLds<Fn, T, Plan, NC>::BYTES;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS of a gfx950 CU");
    const long long ntiles = (a.N + TILE - 1) / TILE;
    constexpr int threads = (Plan::NG + NC) * TILE;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernel_dyn), lds) != hipSuccess) return RDHIP_ERR_HIP;
    const unsigned full = tile_grid_occ(reinterpret_cast<const void*>(kernel_dyn), lds, threads, ntiles);
    if (a.sched) {
        const unsigned gd = dyn_grid(full);
        if (ntiles >= 8LL * gd) {
            RD_LAUNCH(kernel_dyn, dim3(gd), dim3(threads), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y, a.sched);
                last_kernel().dyn = true;
            return hip_status(hipGetLastError());
        }
    }
    RD_LAUNCH(kernel_dyn, dim3(full), dim3(threads), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y, (unsigned*)nullptr);
    return hip_status(hipGetLastError());
}

// the same loop for kernels without a separate f(z) output (signature ..., J)
template <class Fn, class T, class Plan, int NC, class K> int launch_split_kernel_j(K kernel, const Fn& fn, const LaunchArgs& a) {
    constexpr size_t lds = SplitLds<Fn, T, Plan, NC>::BYTES;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS of a gfx950 CU");
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return RDHIP_ERR_HIP;
    const long long ntiles = (a.N + TILE - 1) / TILE;
    constexpr int threads = (Plan::NG + NC) * TILE;
    RD_LAUNCH(kernel, dim3(tile_grid_occ(reinterpret_cast<const void*>(kernel), lds, threads, ntiles)), dim3(threads), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J);
    return hip_status(hipGetLastError());
}


// host MultiDesc -> the kernel argument (same bytes: MultiPod is the image of MultiBatch<T>)
template <class T> MultiBatch<T> to_multi_batch(const MultiDesc& md) {
    static_assert(sizeof(MultiBatch<T>) == sizeof(MultiPod) && sizeof(MultiRec<T>) == sizeof(MultiRecPod) && MULTI_MAXB == MULTI_INLINE, "MultiPod is the byte image of MultiBatch<T>");
    MultiBatch<T> mb;
    std::memcpy(static_cast<void*>(&mb), &md.pod, sizeof mb);
    return mb;
}
template <class Fn, class T, class Plan, int NC, class Body, int MINB = 1> int launch_split_multi(const Fn& fn, const LaunchArgs& a) {
    constexpr size_t lds = SplitLds<Fn, T, Plan, NC>::BYTES;
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS of a gfx950 CU");
    constexpr int threads = (Plan::NG + NC) * TILE;
    // fp64: ONE instantiation that carries the dynamic tile schedule and takes it at run time (counter words + 8 or more tiles per workgroup of the reduced
    // grid), like launch_split_kernel_rt; fp32 tiles never take it (kernels.hpp DYN_OK)
    auto kern = jac_split_multi_kernel<Fn, T, Plan, NC, Body, MINB, sizeof(T) == 8>;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
    const unsigned full = tile_grid_occ(reinterpret_cast<const void*>(kern), lds, threads, a.multi->pod.total);
    if (sizeof(T) == 8 && a.sched) {
        const unsigned gd = dyn_grid(full);
        if (a.multi->pod.total >= 8LL * gd) {
            RD_LAUNCH(kern, dim3(gd), dim3(threads), lds, a.stream, fn, to_multi_batch<T>(*a.multi), a.sched);
                last_kernel().dyn = true;
            return hip_status(hipGetLastError());
        }
    }
    RD_LAUNCH(kern, dim3(full), dim3(threads), lds, a.stream, fn, to_multi_batch<T>(*a.multi), (unsigned*)nullptr);
    return hip_status(hipGetLastError());
}
// several packed trajectories, one launch.  Round 6: the one-launch form exists for ForwardAD -- the reference's default diff method
// of a DiscretizedDynamics (src/discretized_dynamics.jl:190) -- on 16-byte aligned outputs: the small-model kernel, or the split-role tile loop on the
// body the single-batch kernel uses.  Every other case (FiniteDifference, the UserDefined chain rule, outputs that are not 16-byte aligned) returns
// RDHIP_ERR_UNSUPPORTED before anything is launched and rdhip_jacobian_multi runs the batches as separate rdhip_jacobian calls (api.hip): same results.
// The batch-table instantiations of those bodies were 52 % of the library's device code (tools/code_size.py on the round-5 build: 66 of 101 MB of
// symbols) and nothing measured or tested them (profiles/r05/kernel_reach.txt: 10 of 1 021 launched).
template <class Mdl, int RULE> int launch_jacobian_multi(const Mdl& m, int diff, const LaunchArgs& a) {
    using Fn = Fun<Mdl, RULE>;
    using T = typename Fn::T;
    constexpr int NZ = Fn::NX + Fn::NU;
    using DefPlan = ContiguousPlan<NZ, (NZ >= 4 ? 4 : NZ)>;
    constexpr bool split_fits = NZ > 10 && SplitLds<Fn, T, DefPlan, 4>::BYTES <= 160 * 1024;
    Fn fn{m};
    const long long ntiles = a.multi->pod.total;
    if (ntiles == 0) return RDHIP_OK;
    const bool al = a.multi->aligned16;
    if (diff < RDHIP_FORWARD_AD || diff > RDHIP_USER_DEFINED) return RDHIP_ERR_INVALID_ARG;
    if (diff != RDHIP_FORWARD_AD) return RDHIP_ERR_UNSUPPORTED;
    if constexpr (NZ <= 10 && !(UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC > 0)) {       // (a small model with a split-role plan of its own -- LieSatellite -- takes the tile loop)
        constexpr size_t lds = SmallLds<Fn, T>::BYTES;
        auto kern = jac_ad_small_multi_kernel<Fn, T>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
        constexpr int W = SmallLds<Fn, T>::WAVES;
        RD_LAUNCH(kern, dim3(tile_grid_occ(reinterpret_cast<const void*>(kern), lds, W * TILE, (ntiles + W - 1) / W)), dim3(W * TILE), lds, a.stream, fn, to_multi_batch<T>(*a.multi));
        return hip_status(hipGetLastError());
    } else if constexpr (UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC > 0) {
        using Plan = typename UnrollPolicy<Mdl, RULE>::Plan;
        constexpr int NC = UnrollPolicy<Mdl, RULE>::NC;
        if (al) return launch_split_multi<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>, (sizeof(T) == 4 ? RD_F32_MINB : 1)>(fn, a);
        return RDHIP_ERR_UNSUPPORTED;
    } else if constexpr (split_fits) {
        if (al) return launch_split_multi<Fn, T, DefPlan, 4, AdLoopBody<Fn, T, DefPlan>>(fn, a);
        return RDHIP_ERR_UNSUPPORTED;
    } else return RDHIP_ERR_UNSUPPORTED;
}

template <class Mdl, int RULE> int launch_jacobian(const Mdl& m, int diff, const LaunchArgs& a) {
    using Fn = Fun<Mdl, RULE>;
    using T = typename Fn::T;
    constexpr int NZ = Fn::NX + Fn::NU;
    using DefPlan = ContiguousPlan<NZ, (NZ >= 4 ? 4 : NZ)>;     // compact kernels: 4 waves per tile, one chunk
    // run-time-column bodies also go through the split-role loop when the output tile is big enough to be worth dedicated copy waves
    constexpr bool split_fits = NZ > 10 && SplitLds<Fn, T, DefPlan, 4>::BYTES <= 160 * 1024;
    if (a.multi) return launch_jacobian_multi<Mdl, RULE>(m, diff, a);
    if (a.N == 0) return RDHIP_OK;
    Fn fn{m};
    if (diff == RDHIP_FORWARD_AD) {
        if constexpr (NZ <= 10 && !(UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC > 0)) {       // (a small model with a split-role plan of its own -- LieSatellite -- takes the tile loop)
            // small models: one wave = 64 knot points x all columns (no per-wave replication of the primal)
            constexpr size_t lds = SmallLds<Fn, T>::BYTES;
            auto kern = jac_ad_small_kernel<Fn, T>;
            if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
            const long long ntiles = (a.N + TILE - 1) / TILE;
            constexpr int W = SmallLds<Fn, T>::WAVES;
            RD_LAUNCH(kern, dim3(tile_grid_occ(reinterpret_cast<const void*>(kern), lds, W * TILE, (ntiles + W - 1) / W)), dim3(W * TILE), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
            return hip_status(hipGetLastError());
        } else if constexpr (UnrollPolicy<Mdl, RULE>::value) {
            using Plan = typename UnrollPolicy<Mdl, RULE>::Plan;
            constexpr int NC = UnrollPolicy<Mdl, RULE>::NC;
            if constexpr (Plan::NG == 4 && !LeanVariant<Mdl>::value) {
                // small batches (at most two of these workgroups per CU: up to 8 192 knot points on a whole MI355X): the columns of a tile over four
                // workgroups.  rocprofv3 medians against the tile loop: 4.8 vs 7.5 us up to 512 knot points, 5.7 vs 8.7 at 4 096, 8.0 vs 8.7 at 8 192;
                // with four per CU 10.8 vs 8.6 at 12 288 and 12.8 vs 9.3 at 16 384 (tools/small_trace.sh)
                const long long ntiles = (a.N + TILE - 1) / TILE;
                const long long per_cu = tune_knobs().small_max;           // (tuning: small_max / small=tile, common.hpp TuneKnobs)
                if (4 * ntiles <= per_cu * (long long)dev_info().cus && tune_knobs().small != 't') {
                    constexpr size_t lds = ColSplitLds<Fn, T>::BYTES;
                    auto kern = jac_ad_colsplit_kernel<Fn, T, Plan>;
                    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                    RD_LAUNCH(kern, dim3((unsigned)(32 * ((ntiles + 7) / 8))), dim3(4 * TILE), lds, a.stream, fn, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.J, (T*)a.Y);
                    return hip_status(hipGetLastError());
                }
            }
            if constexpr (NC > 0) {
                // compute/copy split-role kernel: needs 16-byte aligned output slabs (a tile is copied as 16-byte units)
                if (aligned16(a.J) && aligned16(a.Y))
                    { if constexpr (sizeof(T) == 8) return launch_split_kernel<Fn, T, Plan, NC>(jac_ad_split_kernel<Fn, T, Plan, NC>, jac_ad_split_kernel<Fn, T, Plan, NC, true>, fn, a);
                      else return launch_split_kernel<Fn, T, Plan, NC>(jac_ad_split_kernel<Fn, T, Plan, NC>, fn, a); }
            }
            return launch_jac_kernel<Fn, T, Plan>(jac_ad_unrolled_kernel<Fn, T, Plan>, fn, a);
        } else {
            if constexpr (split_fits)
                if (aligned16(a.J) && aligned16(a.Y)) { if constexpr (sizeof(T) == 8) return launch_split_kernel_rt<Fn, T, DefPlan, 4>(jac_ad_loop_split_kernel<Fn, T, DefPlan, 4, true>, fn, a);
                  else return launch_split_kernel<Fn, T, DefPlan, 4>(jac_ad_loop_split_kernel<Fn, T, DefPlan, 4>, fn, a); }
            return launch_jac_kernel<Fn, T, DefPlan>(jac_ad_loop_kernel<Fn, T, DefPlan>, fn, a);
        }
    } else if (diff == RDHIP_FINITE_DIFF) {
        // models with a cascade structure and a column plan: perturbed evaluations integrate only the blocks a column can change (cascade.hpp)
        if constexpr (RULE != RULE_CONTINUOUS && Cascade<Mdl>::NB > 1 && FdPlanOf<Mdl>::value) {
            using Plan = typename FdPlanOf<Mdl>::Plan;
            if (aligned16(a.J) && aligned16(a.Y)) { if constexpr (sizeof(T) == 8) return launch_split_kernel_rt<Fn, T, Plan, 4>(jac_fd_cascade_kernel<Fn, T, Plan, Cascade<Mdl>, 4, true>, fn, a);
              else return launch_split_kernel<Fn, T, Plan, 4>(jac_fd_cascade_kernel<Fn, T, Plan, Cascade<Mdl>, 4>, fn, a); }
        }
        else if constexpr (split_fits)                     // (aligned outputs of a model with a cascade plan never get here: no instantiation for them)
            if (aligned16(a.J) && aligned16(a.Y)) { if constexpr (sizeof(T) == 8) return launch_split_kernel_rt<Fn, T, DefPlan, 4>(jac_fd_split_kernel<Fn, T, DefPlan, 4, true>, fn, a);
              else return launch_split_kernel<Fn, T, DefPlan, 4>(jac_fd_split_kernel<Fn, T, DefPlan, 4>, fn, a); }
        return launch_jac_kernel<Fn, T, DefPlan>(jac_fd_kernel<Fn, T, DefPlan>, fn, a);
    } else if (diff == RDHIP_USER_DEFINED) {
        if constexpr (Mdl::HAS_USER_JACOBIAN) {
            if constexpr (UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC > 0) {
                using Plan = typename UnrollPolicy<Mdl, RULE>::Plan;
                constexpr int NC = UnrollPolicy<Mdl, RULE>::NC;
                if (aligned16(a.J) && aligned16(a.Y))
                    { if constexpr (sizeof(T) == 8) return launch_split_kernel_rt<Fn, T, Plan, NC>(jac_chain_split_kernel<Fn, T, Plan, NC, true>, fn, a);
                      else return launch_split_kernel<Fn, T, Plan, NC>(jac_chain_split_kernel<Fn, T, Plan, NC>, fn, a); }
            }
            return launch_jac_kernel<Fn, T, DefPlan>(jac_chain_kernel<Fn, T, DefPlan>, fn, a);
        }
        else return RDHIP_ERR_NOT_IMPLEMENTED;           // jacobian!(fun, J, y, x, u) not defined for this model (src/functionbase.jl:257)
    }
    return RDHIP_ERR_INVALID_ARG;
}

// rollout! + Jacobians along it in one launch (fused.hpp): models with a cascade structure and a compile-time column plan, ForwardAD, time-major
// data whose 64-trajectory tiles start on 16-byte boundaries; anything else -> RDHIP_ERR_UNSUPPORTED (the caller issues the two launches)
template <class Mdl, int RULE> int launch_rollout_jacobian_fused(const Mdl& m, int diff, const LaunchArgs& a) {
    using Fn = Fun<Mdl, RULE>;
    using T = typename Fn::T;
    using Cas = Cascade<Mdl>;
    if constexpr (Cas::NB > 1 && UnrollPolicy<Mdl, RULE>::value && UnrollPolicy<Mdl, RULE>::NC == 4 && !LeanVariant<Mdl>::value) {
        using Plan = typename UnrollPolicy<Mdl, RULE>::Plan;
        constexpr int NC = 4, R = 4;
        if constexpr (Plan::NG == 4) {
            if (tune_knobs().rolljac == 's') return RDHIP_ERR_UNSUPPORTED;      // tuning switch rolljac=split: the two launches
            const long long K = a.K, M = a.N;
            if (diff != RDHIP_FORWARD_AD || !a.progress || K < 1 || M < 1 || K >= (1LL << 26) || M >= (1LL << 26)) return RDHIP_ERR_UNSUPPORTED;
            // time-major tiles are copied as aligned 16-byte units of one slab: every knot's row of tiles must start aligned (trajectory-major
            // tiles are gathered run by run and take any alignment)
            if (a.time_major && (!aligned16(a.J) || (M * Fn::NY * (Fn::NX + Fn::NU) * (long long)sizeof(T)) % 16 != 0)) return RDHIP_ERR_UNSUPPORTED;
            const long long G = (M + TILE - 1) / TILE;
            using FL = FusedLds<Fn, T, Cas, R, Plan, NC>;
            constexpr size_t lds = FL::BYTES;
            static_assert(lds <= 160 * 1024, "fused kernel: LDS");
            const long long cus = dev_info().cus;
            // the rollout workgroups must leave most of the chip to the Jacobians (and all of them must be resident together with consumers)
            if (G * FUSED_PROGRESS_STRIDE > a.progress_words || 2 * G > cus) return RDHIP_ERR_UNSUPPORTED;
            // (FusedWalk counts items and virtual indices in 32 bits: K G items + one grid of slack; a stack that large does not fit a GPU's memory, but say so)
            if (K * G + 4 * cus >= (1LL << 32)) return RDHIP_ERR_UNSUPPORTED;
            if (a.probe) return RDHIP_OK;
            // The walk's second part (fused.hpp FusedWalk): the rollout workgroups produce G tiles per step, the nj Jacobian workgroups consume nj tiles per
            // tile time; while the rollout runs, the consumers get through  A = nj (K - 1) step / tile_time  tiles.  If that is (nearly) everything the
            // consumers keep up and there is no second part; otherwise the tiles behind A are shared by ALL workgroups, the former rollout workgroups
            // included.  fp32, satellite RK4 (tune/tune_c5.hip QUICK mode -- one process per split, 1.5 s of load first, the first measurement behind it:
            // the only numbers that repeat to +-1 us; a sweep inside one process drifts by 10 % with the chip's power state --, 4 096 x 256): 234 / 228 /
            // 221 / 224 / 227 / 231 / 232 us at 0.40 / 0.45 / 0.50 / 0.55 / 0.60 / 0.65 / 0.70 K, i.e. ratio 5.9 (at the minimum the former rollout
            // workgroups and the Jacobian workgroups end together, 205-213 and 208-218 us); calibrated on the satellite only.
            // fp64 takes NO second part: its 126 KB tiles are bandwidth-bound on three quarters of the chip already (the plain kernel is faster on
            // 192 workgroups than on 256, launch_split_kernel above) -- 4 096 x 101 knots 162-166 us without, 173-181 us with a split anywhere
            // between 0.3 and 0.55 K.  RDHIP_TUNE=fused_ratio=x (tuning) forces a ratio for either type.
            const double ratio_env = tune_knobs().fused_ratio;
            const long long nj_env = tune_knobs().fused_jac_workgroups;     // test switch: fewer Jacobian workgroups (tests/test_gpu_round4.py sets it through rdhip_tune)
            auto split_of = [&](long long ntiles, long long nj) -> long long {
                if (sizeof(T) == 8 && !(ratio_env > 0)) return 0;
                const double ratio = ratio_env > 0 ? ratio_env : 5.9;
                const double A = (double)nj * (double)(K - 1) / ratio;
                if (G < 8 || A >= 0.85 * (double)ntiles) return 0;
                long long sp = (long long)A; sp -= sp % G;
                return sp < G ? 0 : sp;
            };
            // fp32, a launch bound by its Jacobian workgroups (what the second part of the walk was built for): TWO workgroups per CU instead -- the kernel
            // under 128 registers with the rollout role's ring in its compact layout (fused.hpp fused_min_waves, cascade.hpp CASCADE_COMPACT), G + (2 cus - G)
            // workgroups, no second part.  4 096 x 256 satellite knots: 221-226 us with the best split -> 199-201 us.  Smaller launches, whose rollout chain
            // is the critical path, keep the 256-register kernel (its chain is 3 % shorter).
            if constexpr (sizeof(T) == 4 && (RULE == RULE_RK3 || RULE == RULE_RK4)) {         // (the rules the reference's users roll out with; Euler and RK2 keep the one form)
                constexpr int RC = R | CASCADE_COMPACT;
                using FL2 = FusedLds<Fn, T, Cas, RC, Plan, NC>;
                static_assert(2 * FL2::BYTES <= 160 * 1024, "two workgroups of the compact fused kernel per CU");
                const long long ntl = a.time_major ? K * G : ((K + GATHER_RUN - 1) / GATHER_RUN) * ((M + GATHER_NRUN - 1) / GATHER_NRUN);
                if (!a.beside && tune_knobs().fused_two_per_cu && !(ratio_env > 0) && nj_env < 1 && split_of(ntl, cus - G) > 0) {
                    const long long nj = 2 * cus - G;
                    if (a.time_major) {
                        auto kern = rollout_jac_fused_kernel<Fn, T, Cas, RC, Plan, NC, true>;
                        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), FL2::BYTES) != hipSuccess) return RDHIP_ERR_HIP;
                        RD_LAUNCH(kern, dim3((unsigned)(G + nj)), dim3(FUSED_THREADS), FL2::BYTES, a.stream, Fn{m}, K, M, (int)G, 1LL, M, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
                                           a.progress, a.pbase, 0LL);
                    } else {
                        auto kern = rollout_jac_fused_kernel<Fn, T, Cas, RC, Plan, NC, false>;
                        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), FL2::BYTES) != hipSuccess) return RDHIP_ERR_HIP;
                        RD_LAUNCH(kern, dim3((unsigned)(G + nj)), dim3(FUSED_THREADS), FL2::BYTES, a.stream, Fn{m}, K, M, (int)G, K, 1LL, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
                                           a.progress, a.pbase, 0LL);
                    }
                    return hip_status(hipGetLastError());
                }
            }
            if (a.time_major) {
                auto kern = rollout_jac_fused_kernel<Fn, T, Cas, R, Plan, NC, true>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                const long long ntiles = K * G;
                long long nj = cus - G; if (nj > ntiles) nj = ntiles;
                if (nj_env >= 1 && nj_env < nj) nj = nj_env;
                RD_LAUNCH(kern, dim3((unsigned)(G + nj)), dim3(FUSED_THREADS), lds, a.stream, Fn{m}, K, M, (int)G, 1LL, M, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
                                   a.progress, a.pbase, split_of(ntiles, nj));
            } else {
                auto kern = rollout_jac_fused_kernel<Fn, T, Cas, R, Plan, NC, false>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                const long long ntiles = ((K + GATHER_RUN - 1) / GATHER_RUN) * ((M + GATHER_NRUN - 1) / GATHER_NRUN);
                long long nj = cus - G; if (nj > ntiles) nj = ntiles;
                if (nj_env >= 1 && nj_env < nj) nj = nj_env;
                RD_LAUNCH(kern, dim3((unsigned)(G + nj)), dim3(FUSED_THREADS), lds, a.stream, Fn{m}, K, M, (int)G, K, 1LL, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
                                   a.progress, a.pbase, split_of(ntiles, nj));
            }
            return hip_status(hipGetLastError());
        }
    }
    return RDHIP_ERR_UNSUPPORTED;
}
template <class Mdl> int dispatch_rolljac_rule(const Mdl& m, int rule, int diff, const LaunchArgs& a) {
    switch (rule) {
        case RDHIP_EULER: return launch_rollout_jacobian_fused<Mdl, RULE_EULER>(m, diff, a);
        case RDHIP_RK2: return launch_rollout_jacobian_fused<Mdl, RULE_RK2>(m, diff, a);
        case RDHIP_RK3: return launch_rollout_jacobian_fused<Mdl, RULE_RK3>(m, diff, a);
        case RDHIP_RK4: return launch_rollout_jacobian_fused<Mdl, RULE_RK4>(m, diff, a);
    }
    return RDHIP_ERR_UNSUPPORTED;
}
template <template <class> class MT> int dispatch_rolljac(const ModelDesc& md, int rule, int diff, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return dispatch_rolljac_rule(MT<double>::make(md.params), rule, diff, a);
    if (dtype == RDHIP_F32) return dispatch_rolljac_rule(MT<float>::make(md.params), rule, diff, a);
    return RDHIP_ERR_INVALID_ARG;
}

// error-state expansion (rigid bodies: fused kernel; needs 16-byte aligned outputs)
template <class Mdl, int RULE> struct ErrPlanOf;      // column plan over the e+m error-state columns, defined next to the model's other plans
template <class Mdl, int RULE> int launch_errexp(const Mdl& m, const LaunchArgs& a) {
    using Fn = ErrFun<Mdl, RULE>;
    using T = typename Fn::T;
    using Plan = typename ErrPlanOf<Mdl, RULE>::Plan;
    constexpr int NC = 4;
    if (a.N == 0) return RDHIP_OK;
    if (!aligned16(a.J) || !aligned16(a.Y)) return RDHIP_ERR_INVALID_ARG;
    Fn fn{{m}};
    { if constexpr (sizeof(T) == 8) return launch_split_kernel_rt<Fn, T, Plan, NC>(jac_errexp_split_kernel<Fn, T, Plan, NC, true>, fn, a);
      else return launch_split_kernel<Fn, T, Plan, NC>(jac_errexp_split_kernel<Fn, T, Plan, NC>, fn, a); }
}
template <class Mdl> int dispatch_errexp_rule(const Mdl& m, int rule, const LaunchArgs& a) {
    switch (rule) {
        case RDHIP_EULER: return launch_errexp<Mdl, RULE_EULER>(m, a);
        case RDHIP_RK2: return launch_errexp<Mdl, RULE_RK2>(m, a);
        case RDHIP_RK3: return launch_errexp<Mdl, RULE_RK3>(m, a);
        case RDHIP_RK4: return launch_errexp<Mdl, RULE_RK4>(m, a);
    }
    return RDHIP_ERR_INVALID_ARG;
}
template <template <class> class MT> int dispatch_errexp(const ModelDesc& md, int rule, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return dispatch_errexp_rule(MT<double>::make(md.params), rule, a);
    if (dtype == RDHIP_F32) return dispatch_errexp_rule(MT<float>::make(md.params), rule, a);
    return RDHIP_ERR_INVALID_ARG;
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

// DiscretizedDynamics{ImplicitMidpoint}: mode 0 = Newton solve, 1 = Newton solve + implicit-function-theorem Jacobian, 2 = dynamics error of pairs
template <class Mdl> int launch_implicit(const Mdl& m, int mode, const LaunchArgs& a) {
    using T = typename Mdl::T;
    if (a.N == 0) return RDHIP_OK;
    const unsigned grid = (unsigned)((a.N + TILE - 1) / TILE);
    if (mode == 2) {
        RD_LAUNCH((implicit_midpoint_error_kernel<Mdl, T>), dim3(grid), dim3(TILE), 0, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)a.J2, (T*)a.J);
        return hip_status(hipGetLastError());
    }
    constexpr size_t lds = ImplicitLds<Mdl, T>::BYTES;
    static_assert(lds <= 160 * 1024, "Newton matrices do not fit the LDS of a CU");
    const long long nblk = (a.N + ImplicitLds<Mdl, T>::KP - 1) / ImplicitLds<Mdl, T>::KP;               // Newton kernels: KP knot points per wave
    const unsigned gridn = tile_grid(lds, TILE, nblk);                                                    // persistent
    // modes 0, 1, 3: the register kernel first; the LDS kernel (real row exchanges) behind it, gated on the flags the first one raises
    const int* gate = nullptr;
    const unsigned gridr = (unsigned)((a.N + TILE - 1) / TILE);
    constexpr bool reg_is_exact = ImplicitRegOps<Mdl, T>::PIVOT;     // small models pivot in registers: no second pass
    // one flag word per 64-knot tile when the context has enough of them (written unconditionally by the register kernel: no memset)
    const int tile_flags = (mode != 3 && (long long)gridr <= IMPLICIT_FLAG_WORDS) ? (1 | (a.epoch << 8)) : 0;
    constexpr size_t lds_reg = ImplicitRegOps<Mdl, T>::LDS_BYTES;
    if (!a.flag) return RDHIP_ERR_INVALID_ARG;                       // (api.hip hands the context's flag words to every Newton call)
    if constexpr (reg_is_exact) {
        // small models: the register kernel exchanges rows for real, so there is no LDS pass behind it (round 6: the 42 instantiations of the LDS
        // kernels these models carried were never launched -- profiles/r06/kernel_reach.txt -- and are no longer built)
        if (mode == 3) {
            if (a.K == 0) return RDHIP_OK;
            RD_LAUNCH((implicit_rollout_reg_kernel<Mdl, T>), dim3(gridr), dim3(TILE), 0, a.stream, m, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0, a.newton_iters, a.newton_tol, a.flag);
        } else if (mode == 1) {
            RD_LAUNCH((implicit_midpoint_reg_kernel<Mdl, T, true>), dim3(gridr), dim3(TILE), lds_reg, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)a.J, a.newton_iters, a.newton_tol, a.flag, tile_flags);
        } else {
            RD_LAUNCH((implicit_midpoint_reg_kernel<Mdl, T, false>), dim3(gridr), dim3(TILE), 0, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)nullptr, a.newton_iters, a.newton_tol, a.flag, tile_flags);
        }
        (void)gate; (void)gridn; (void)nblk;
        return hip_status(hipGetLastError());
    } else {
    if (!tile_flags && hipMemsetAsync(a.flag, 0, sizeof(int), a.stream) != hipSuccess) return RDHIP_ERR_HIP;
    gate = a.flag;
    if (mode == 3) {          // rollout: a.N trajectories of a.K knot points
        if (a.K == 0) return RDHIP_OK;
        if constexpr (Cascade<Mdl>::NB > 1) {
            // one wavefront per cascade block + the IO wavefront, every wave solving its block of the implicit equation per step (ImplicitRollFun)
            if (tune_knobs().rollout != 'p' && a.K < (1LL << 26) && a.N < (1LL << 26)) {      // (tuning switch rollout=plain keeps one lane per trajectory)
                using RFn = ImplicitRollFun<Mdl>;
                using Cas = Cascade<Mdl>;
                constexpr size_t ldsr = CascadeLds<RFn, Cas, 4>::BYTES;
                auto kr = rollout_cascade_kernel<RFn, T, Cas, 4>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kr), ldsr) != hipSuccess) return RDHIP_ERR_HIP;
                const long long sj = a.time_major ? 1 : a.K, sk = a.time_major ? a.N : 1;
                RD_LAUNCH(kr, dim3(gridr), dim3((Cas::NB + 1) * TILE), ldsr, a.stream, RFn{m, a.newton_iters, a.newton_tol, a.flag}, a.K, a.N, sj, sk, (T*)a.Z, a.t, a.dt, (const T*)a.X0);
                auto kern = implicit_rollout_kernel<Mdl, T>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                RD_LAUNCH(kern, dim3((unsigned)nblk), dim3(TILE), lds, a.stream, m, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0, a.newton_iters, a.newton_tol, gate);
                return hip_status(hipGetLastError());
            }
        }
        RD_LAUNCH((implicit_rollout_reg_kernel<Mdl, T>), dim3(gridr), dim3(TILE), 0, a.stream, m, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0, a.newton_iters, a.newton_tol, a.flag);
        auto kern = implicit_rollout_kernel<Mdl, T>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
        RD_LAUNCH(kern, dim3((unsigned)nblk), dim3(TILE), lds, a.stream, m, a.K, a.N, (T*)a.Z, a.t, a.dt, (const T*)a.X0, a.newton_iters, a.newton_tol, gate);
        return hip_status(hipGetLastError());
    }
    if (mode == 1) {
        // models whose Jacobian tile fits the split-role loop: Newton solve (register kernel), then the IFT Jacobian as whole-line tile stores
        // (implicit_ift_split_kernel), then the exact LDS kernel for the tiles either of them flagged
        constexpr int NZ = Mdl::NX + Mdl::NU;
        using IFn = IftFun<Mdl>;
        using IPlan = typename IftPlanOf<Mdl>::Plan;
        if constexpr (NZ > 10 && SplitLds<IFn, T, IPlan, 4>::BYTES <= 160 * 1024) {
            const char ift_mode = tune_knobs().ift;                        // tuning switches: 'r' = the one-kernel register path, '3' = Newton kernel + tile kernel + gated exact kernel (round 2)
            if (aligned16(a.J) && ift_mode != 'r') {
                // ONE launch: the tile kernel's compute waves run the Newton solve themselves, x_next leaves through the tile loop's f(z) slab, and the
                // tiles that need the exact solve are redone by their own workgroup behind the loop (implicit.hpp IftBody)
                if (aligned16(a.Y) && (long long)gridr <= IMPLICIT_FLAG_WORDS && ift_mode != '3')
                    return launch_split_kernel<IFn, T, IPlan, 4>(implicit_ift_split_kernel<IFn, T, IPlan, 4>, IFn{m, a.newton_tol, a.flag, a.newton_iters, IftBody<IFn, T, IPlan>::HELPER ? 2 : 1, nullptr}, a);
                RD_LAUNCH((implicit_midpoint_reg_kernel<Mdl, T, false>), dim3(gridr), dim3(TILE), 0, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)nullptr, a.newton_iters, a.newton_tol, a.flag, tile_flags | 2);
                LaunchArgs b = a;
                b.Y = nullptr;
                // (the tile kernel reads x_next from a.Y; its flag words are per tile, which the gated kernel behind it scans when word 0 carries this call's epoch)
                const int rc = launch_split_kernel<IFn, T, IPlan, 4>(implicit_ift_split_kernel<IFn, T, IPlan, 4>, IFn{m, a.newton_tol, (tile_flags & 1) ? a.flag : nullptr, a.newton_iters, 0, (const T*)a.Y}, b);
                if (rc) return rc;
                auto kern = implicit_midpoint_kernel<Mdl, T, true>;
                if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
                RD_LAUNCH(kern, dim3(gridn), dim3(TILE), lds, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)a.J, a.newton_iters, a.newton_tol, gate, tile_flags & 1);
                return hip_status(hipGetLastError());
            }
        }
        RD_LAUNCH((implicit_midpoint_reg_kernel<Mdl, T, true>), dim3(gridr), dim3(TILE), lds_reg, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)a.J, a.newton_iters, a.newton_tol, a.flag, tile_flags);
        auto kern = implicit_midpoint_kernel<Mdl, T, true>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
        RD_LAUNCH(kern, dim3(gridn), dim3(TILE), lds, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)a.J, a.newton_iters, a.newton_tol, gate, tile_flags);
    } else {
        RD_LAUNCH((implicit_midpoint_reg_kernel<Mdl, T, false>), dim3(gridr), dim3(TILE), 0, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)nullptr, a.newton_iters, a.newton_tol, a.flag, tile_flags);
        auto kern = implicit_midpoint_kernel<Mdl, T, false>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
        RD_LAUNCH(kern, dim3(gridn), dim3(TILE), lds, a.stream, m, a.N, (const T*)a.Z, a.t, a.dt, (T*)a.Y, (T*)nullptr, a.newton_iters, a.newton_tol, gate, tile_flags);
    }
    return hip_status(hipGetLastError());
    }
}
template <template <class> class MT> int dispatch_implicit(const ModelDesc& md, int mode, int dtype, const LaunchArgs& a) {
    if (dtype == RDHIP_F64) return launch_implicit(MT<double>::make(md.params), mode, a);
    if (dtype == RDHIP_F32) return launch_implicit(MT<float>::make(md.params), mode, a);
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
