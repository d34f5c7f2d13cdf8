// fam_mixed.hip -- the mixed-model Monte-Carlo batch as ONE launch (mixed.hpp): RigidBody{QuatRotation} satellite (world-frame velocities, diagonal
// inertia -- examples/single_satellite.jl, test/rigidbody_test.jl `Body`) + Cartpole (examples/cartpole.jl), both under the same rule (RK3 or RK4),
// fp32 and fp64: BASELINE config 5's pair.  Every other pair of models keeps rdhip_rollout_jacobian_multi's streams (api.hip).
#include "fam_rigid.hpp"
#include "mixed.hpp"
namespace rd {
template <class T, int RULE, bool TM> static int launch_mixed(const ModelDesc& da, const ModelDesc& db, const LaunchArgs& a, const LaunchArgs& b) {
    using MA = RigidBody<T, WRENCH_SATELLITE, false, true, false>;
    using MB = Cartpole<T>;
    using FnA = Fun<MA, RULE>;
    using FnB = Fun<MB, RULE>;
    using Cas = Cascade<MA>;
    using Plan = SatPlan;
    constexpr int NC = 4, R = 4;
    using FL = FusedLds<FnA, T, Cas, R, Plan, NC>;
    constexpr size_t lds = FL::BYTES;
    static_assert(lds <= 160 * 1024, "mixed kernel: LDS");
    static_assert((FUSED_THREADS / TILE) * SmallLds<FnB, T>::WAVE_BYTES <= lds, "the small model's wave-private tiles fit the tile loop's LDS");
    const long long KA = a.K, MAt = a.N, KB = b.K, MBt = b.N;
    if (!a.progress || KA < 1 || MAt < 1 || KB < 1 || MBt < 1 || KA >= (1LL << 26) || MAt >= (1LL << 26) || KB >= (1LL << 26) || MBt >= (1LL << 26)) return RDHIP_ERR_UNSUPPORTED;
    // time-major: A's tiles are copied as aligned 16-byte units of one slab (launch_rollout_jacobian_fused); trajectory-major tiles are gathered run by run
    if (TM && (!aligned16(a.J) || (MAt * FnA::NY * (FnA::NX + FnA::NU) * (long long)sizeof(T)) % 16 != 0)) return RDHIP_ERR_UNSUPPORTED;
    const long long GA = (MAt + TILE - 1) / TILE, ngB = (MBt + TILE - 1) / TILE;
    const long long cus = dev_info().cus;
    // Where B's rollout runs: on the idle waves of A's rollout workgroups (FUSED_THREADS / 64 - (NB + 1) each) when the launch is bound by A's Jacobian
    // tiles (32 groups and more: the rollout is not the critical path and every compute unit counts as a Jacobian workgroup), on workgroups of its own
    // otherwise.  tune knob mixed_idle=0/1 forces either.
    constexpr int IDLE = FUSED_THREADS / TILE - (Cas::NB + 1);
    const int idle_env = tune_knobs().mixed_idle;
    const bool on_idle = ngB <= (long long)IDLE * GA && (idle_env < 0 ? GA >= 32 : idle_env != 0);
    const long long GB = on_idle ? 0 : (ngB + mixed_wpb<TM>() - 1) / mixed_wpb<TM>();
    // every workgroup of the grid must be resident together (producers + at least as many Jacobian workgroups); B's words sit behind A's
    if ((GA + 1) * FUSED_PROGRESS_STRIDE > a.progress_words || 2 * (GA + GB) > cus) return RDHIP_ERR_UNSUPPORTED;
    const long long ntiles = TM ? KA * GA : ((KA + GATHER_RUN - 1) / GATHER_RUN) * ((MAt + GATHER_NRUN - 1) / GATHER_NRUN);
    if (ntiles + 4 * cus >= (1LL << 32) || (KB * MBt + TILE - 1) / TILE >= (1LL << 31)) return RDHIP_ERR_UNSUPPORTED;
    if (a.probe) return RDHIP_OK;
    const long long nj = cus - GA - GB;
    // fp32, a launch bound by A's Jacobian tiles: the two-per-CU form of the kernel (dispatch.hpp launch_rollout_jacobian_fused), 2 cus workgroups, no second part
    if constexpr (sizeof(T) == 4) {
        constexpr int RC = R | CASCADE_COMPACT;
        using FL2 = FusedLds<FnA, T, Cas, RC, Plan, NC>;
        static_assert(2 * FL2::BYTES <= 160 * 1024 && (FUSED_THREADS / TILE) * SmallLds<FnB, T>::WAVE_BYTES <= FL2::BYTES, "two workgroups of the compact mixed kernel per CU");
        const double A2 = (double)nj * (double)(KA - 1) / 5.9;
        if (tune_knobs().fused_two_per_cu && !(tune_knobs().fused_ratio > 0) && GA >= 8 && A2 < 0.85 * (double)ntiles) {
            auto k2 = rollout_jac_mixed_kernel<FnA, T, Cas, RC, Plan, NC, FnB, TM>;
            if (ensure_dynamic_lds(reinterpret_cast<const void*>(k2), FL2::BYTES) != hipSuccess) return RDHIP_ERR_HIP;
            RD_LAUNCH(k2, dim3((unsigned)(2 * cus)), dim3(FUSED_THREADS), FL2::BYTES, a.stream, FnA{MA::make(da.params)}, KA, MAt, (int)GA, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
                      FnB{MB::make(db.params)}, KB, MBt, (int)GB, (T*)b.Z, b.t, b.dt, (const T*)b.X0, (T*)b.J, a.progress, a.pbase, 0LL);
            return hip_status(hipGetLastError());
        }
    }
    auto kern = rollout_jac_mixed_kernel<FnA, T, Cas, R, Plan, NC, FnB, TM>;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
    // the second part of the walk, sized as in launch_rollout_jacobian_fused (fp32: ratio 5.9 of rollout step to tile time; fp64: none)
    long long split = 0;
    {
        const double ratio_env = tune_knobs().fused_ratio;
        if (!(sizeof(T) == 8 && !(ratio_env > 0))) {
            const double ratio = ratio_env > 0 ? ratio_env : 5.9;
            const double A = (double)nj * (double)(KA - 1) / ratio;
            if (GA >= 8 && A < 0.85 * (double)ntiles) { long long sp = (long long)A; sp -= sp % GA; split = sp < GA ? 0 : sp; }
        }
    }
    RD_LAUNCH(kern, dim3((unsigned)cus), dim3(FUSED_THREADS), lds, a.stream, FnA{MA::make(da.params)}, KA, MAt, (int)GA, (T*)a.Z, a.t, a.dt, (const T*)a.X0, (T*)a.J,
              FnB{MB::make(db.params)}, KB, MBt, (int)GB, (T*)b.Z, b.t, b.dt, (const T*)b.X0, (T*)b.J, a.progress, a.pbase, split);
    return hip_status(hipGetLastError());
}
// RDHIP_OK (launched, or a.probe: could be launched) / RDHIP_ERR_UNSUPPORTED before anything is launched: not this pair of models / rules / shapes
int fam_mixed_rolljac(const ModelDesc& da, int rule_a, const ModelDesc& db, int rule_b, int diff, int dtype, const LaunchArgs& a, const LaunchArgs& b) {
    if (diff != RDHIP_FORWARD_AD || rule_a != rule_b || (rule_a != RDHIP_RK4 && rule_a != RDHIP_RK3)) return RDHIP_ERR_UNSUPPORTED;
    if (da.kind != RDHIP_MODEL_SATELLITE || da.rot != RDHIP_ROT_QUAT || da.frame != RDHIP_FRAME_WORLD || !diag_inertia(da) || db.kind != RDHIP_MODEL_CARTPOLE) return RDHIP_ERR_UNSUPPORTED;
    if (a.time_major != b.time_major) return RDHIP_ERR_UNSUPPORTED;
    const bool rk4 = rule_a == RDHIP_RK4;
    if (a.time_major) {
        if (dtype == RDHIP_F32) return rk4 ? launch_mixed<float, RULE_RK4, true>(da, db, a, b) : launch_mixed<float, RULE_RK3, true>(da, db, a, b);
        if (dtype == RDHIP_F64) return rk4 ? launch_mixed<double, RULE_RK4, true>(da, db, a, b) : launch_mixed<double, RULE_RK3, true>(da, db, a, b);
    } else {
        // a vector of trajectories, the reference's own layout (round 6, last: config 5 357 -> 2xx us on such stacks).  fp64 stacks of bandwidth size stay
        // with the two launches per model, as in rdhip_rollout_jacobian (api.hip tj_two_launches: the gathered fp64 tiles were measured slower there)
        if (dtype == RDHIP_F32) return rk4 ? launch_mixed<float, RULE_RK4, false>(da, db, a, b) : launch_mixed<float, RULE_RK3, false>(da, db, a, b);
        if (dtype == RDHIP_F64) {
            if (a.K * a.N >= tune_knobs().fused_tj_max_knots) return RDHIP_ERR_UNSUPPORTED;
            return rk4 ? launch_mixed<double, RULE_RK4, false>(da, db, a, b) : launch_mixed<double, RULE_RK3, false>(da, db, a, b);
        }
    }
    return RDHIP_ERR_INVALID_ARG;
}
}  // namespace rd
