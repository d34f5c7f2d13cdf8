// user_model.hpp -- what only the hipRTC plug-in (jit.hip) instantiates: user-defined DISCRETE dynamics and user-written Jacobians.
//
// The reference lets a user write a model directly in discrete time -- `discrete_dynamics(model, x, u, t, dt)` on a `DiscreteDynamics` subtype
// (src/discrete_dynamics.jl:1-108; test/discrete_dynamics_test.jl:5-63 is its double integrator) -- and lets `UserDefined()` mean the user's OWN
// formulas: `jacobian!(model, J, xdot, x, u, t)` for a continuous model (src/dynamics.jl:115-116; test/cartpole_model.jl:57-96 is the cartpole's),
// `jacobian!(model, J, xn, x, u, t, dt)` for a discrete one (src/discrete_dynamics.jl:111-113).  The compiled-in families never include this
// header: it is embedded next to the kernel headers and pulled in by the source jit.hip generates.
//
//   RULE_DISCRETE          Fun<Mdl, RULE_DISCRETE>::eval is the model's own `step` -- every kernel that is templated on a Fun (evaluation, rollout,
//                          ForwardAD, FiniteDifference, dynamics error, error-state expansion) then works on a discrete model unchanged.
//   UserJacBody            UserDefined for models that bring `jac`: one computing wave per 64-knot tile, the user's function writes the n x (n+m) column-major
//                          matrix of its knot point straight into the lane's row of the LDS output tile (zero-filled first, as `J .= 0` opens the
//                          reference's examples), the tile leaves through the generic tile loop.
//   UserChainBody          UserDefined for DiscretizedDynamics{Euler / RK2 / RK3 / RK4} of a continuous model that brings `jac`: the integrators'
//                          chain rule (src/integration.jl:85-93, :149-177, :302-337) on the user's stage Jacobians -- the reference's literal
//                          recipe, column by column.  The stage Jacobians live in registers: offered for n + m <= 10; bigger models keep the
//                          matrix-free chain rule on forward duals (kernels.hpp ChainBody), which computes the same products without the user's
//                          formulas (rdhip.h says so).
#pragma once
#include "kernels.hpp"

namespace rd {

constexpr int RULE_DISCRETE = 5;     // (= RDHIP_DISCRETE: rule codes travel from the C ABI into the templates unchanged)

template <class Mdl> struct Fun<Mdl, RULE_DISCRETE> {
    using T = typename Mdl::T;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = Mdl::NX, RULE_ID = RULE_DISCRETE;
    static constexpr int NCOLS = Mdl::NX + Mdl::NU, NVAL = Mdl::NX, NEXT = 0;
    Mdl m;
    template <class S> RD_DEV void eval(const S* x, const S* u, T t, T h, S* y) const { m.step(x, u, t, h, y); }
};

// the user's Jacobian of `fn` at one knot point into J (column-major NY x NZ, zero on entry)
template <class Fn, class T> RD_DEV void user_jac(const Fn& fn, const T* x, const T* u, T t, T h, T* J) {
    if constexpr (Fn::RULE_ID == RULE_DISCRETE) fn.m.jac(x, u, t, h, J);
    else fn.m.jac(x, u, t, J);
}

template <class Fn, class T, class Plan> struct UserJacBody {
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY;
        // (two waves per workgroup: the tile loop stages t and dt with one wave each and shares the copy-out; the arithmetic is wave 0's)
        static_assert(Plan::NG == 2 && Plan::NCHUNK == 1 && (Fn::RULE_ID == RULE_CONTINUOUS || Fn::RULE_ID == RULE_DISCRETE), "two waves per tile; the model's own function");
        if (fr.g != 0) return;
        T* J = fr.row;                                       // this lane's row of the LDS tile: NZ columns of NY values, then f(z)
#pragma unroll
        for (int e = 0; e < NY * NZ; ++e) J[e] = T(0);
        user_jac<Fn, T>(fn, fr.x, fr.u, fr.t, fr.h, J);
        // (a terminal knot point: the user's function was called with control(z) = zeros -- read_knot -- and what it wrote is handed back as it is, like
        // the reference's jacobian!(fun, J, y, state(z), control(z)), src/functionbase.jl:252-257; rounds 1-5 zeroed its control columns)
        if (fr.want_y) { T y[NY]; fn.eval(fr.x, fr.u, fr.t, fr.h, y); fr.put_value(NZ, y); }
    }
};
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_user_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                  const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, UserJacBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

// A_d, B_d of an explicit rule from the user's continuous Jacobians at the stage points, one column of [A_d B_d] at a time: with w = e_c,
//   d1 = (A1 w + B1 wu) h ; d2 = (A2 (w + d1/2) + B2 wu) h ; d3 = (A3 (w + d2/2) + B3 wu) h ; d4 = (A4 (w + d3) + B4 wu) h ; col = w + (d1 + 2 d2 + 2 d3 + d4)/6
// which is column c of  I + (dA1 + 2 dA2 + 2 dA3 + dA4)/6  |  (dB1 + 2 dB2 + 2 dB3 + dB4)/6  with the dA_i, dB_i of src/integration.jl:302-337
// (RK3: stage points x, x + k1/2, x - k1 + 2 k2, weights 1, 4, 1, :149-177; Euler: I + A1 h | B1 h, :85-93; RK2 = explicit midpoint).
template <class Fn, class T, class Plan> struct UserChainBody {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, RULE = Fn::RULE_ID;
    static constexpr int NS = RULE == RULE_EULER ? 1 : RULE == RULE_RK2 ? 2 : RULE == RULE_RK3 ? 3 : 4;
    // (A w + B wu) from a stored stage Jacobian
    static RD_DEV void mv(const T (&Js)[NY * NZ], const T (&w)[NX], const T (&wu)[NU], T (&o)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) s = s + Js[j * NY + i] * w[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) s = s + Js[(NX + j) * NY + i] * wu[j];
            o[i] = s;
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        static_assert(Plan::NG == 2 && Plan::NCHUNK == 1 && RULE >= RULE_EULER && RULE <= RULE_RK4, "two waves per tile (see UserJacBody); explicit rules");
        if (fr.g != 0) return;
        const T h = fr.h;
        // stage points exactly as the reference forms them (k_i = f(.) h), then the user's Jacobian at each
        T xs[NS][NX], k1[NX], k2[NX], k3[NX];
        T ts[NS];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[0][i] = fr.x[i];
        ts[0] = fr.t;
        if constexpr (NS >= 2) {
            model_f(fn.m, fr.x, fr.u, fr.t, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k1[i] = k1[i] * h; xs[1][i] = fr.x[i] + k1[i] * T(0.5); }
            ts[1] = fr.t + h * T(0.5);
        }
        if constexpr (NS >= 3) {
            model_f(fn.m, xs[1], fr.u, ts[1], k2);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k2[i] = k2[i] * h; xs[2][i] = (RULE == RULE_RK3) ? fr.x[i] - k1[i] + k2[i] * T(2) : fr.x[i] + k2[i] * T(0.5); }
            ts[2] = (RULE == RULE_RK3) ? fr.t + h : fr.t + h * T(0.5);
        }
        if constexpr (NS >= 4) {
            model_f(fn.m, xs[2], fr.u, ts[2], k3);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k3[i] = k3[i] * h; xs[3][i] = fr.x[i] + k3[i]; }
            ts[3] = fr.t + h;
        }
        T Js[NS][NY * NZ];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int e = 0; e < NY * NZ; ++e) Js[s][e] = T(0);
            fn.m.jac(xs[s], fr.u, ts[s], Js[s]);
        }
        if (fr.want_y) { T y[NY]; fn.eval(fr.x, fr.u, fr.t, fr.h, y); fr.put_value(NZ, y); }
#pragma unroll
        for (int c = 0; c < NZ; ++c) {
            T w[NX], wu[NU], d1[NX], d2[NX], d3[NX], d4[NX], wsv[NX], col[NY];
#pragma unroll
            for (int i = 0; i < NX; ++i) w[i] = (i == c) ? T(1) : T(0);
#pragma unroll
            for (int i = 0; i < NU; ++i) wu[i] = (NX + i == c && !fr.term) ? T(1) : T(0);      // terminal knot point: control columns are zero
            mv(Js[0], w, wu, d1);
            if constexpr (RULE == RULE_EULER) {
#pragma unroll
                for (int i = 0; i < NX; ++i) col[i] = w[i] + d1[i] * h;
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) { d1[i] = d1[i] * h; wsv[i] = w[i] + d1[i] * T(0.5); }
                mv(Js[1], wsv, wu, d2);
                if constexpr (RULE == RULE_RK2) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + d2[i] * h;
                } else if constexpr (RULE == RULE_RK3) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; wsv[i] = w[i] - d1[i] + d2[i] * T(2); }
                    mv(Js[2], wsv, wu, d3);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(4) + d3[i] * h) / T(6);
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; wsv[i] = w[i] + d2[i] * T(0.5); }
                    mv(Js[2], wsv, wu, d3);
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d3[i] = d3[i] * h; wsv[i] = w[i] + d3[i]; }
                    mv(Js[3], wsv, wu, d4);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(2) + d3[i] * T(2) + d4[i] * h) / T(6);
                }
            }
            fr.put_at(c, col);
        }
    }
};
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_user_chain_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                        const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, UserChainBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}
// ... and over several packed trajectories (rdhip_jacobian_multi): jac_tile_multi_kernel<Fn, T, Plan, UserJacBody / UserChainBody> (kernels.hpp)

}  // namespace rd
