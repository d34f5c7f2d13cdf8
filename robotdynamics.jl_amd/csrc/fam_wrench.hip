// fam_wrench.hip -- wrenches(model, x, u, t) = [forces; moments] of the compiled-in RigidBody models (src/rigidbody.jl:244-248) for every knot
// point of a batch, and its Jacobian with respect to z = [x; u]: what the reference's default wrench_jacobian!(F, model, z)
// (src/rigidbody.jl:315-320, ForwardDiff.jacobian of `wrenches`) is written to return -- in v0.4.8 that default cannot run, the knot-point
// method of `wrenches` it calls is commented out at :243; test/rigidbody_test.jl:33-46 holds the closed form for the satellite wrench.
// Not a bandwidth path of its own (6 x (n+m) values per knot point): one lane per knot point, one forward dual per column with a run-time
// seed, the 64 x 6 x (n+m) tile assembled in LDS and stored as a lane-contiguous sweep.
#include "common.hpp"
#include "kernels.hpp"

namespace rd {

// the wrench of models.hpp's rigid bodies as a free function of the unit quaternion (RigidBody::wrench; RigidBody3::f holds the same lines)
template <int WRENCH, class T, class S> RD_DEV void rigid_wrench(const S* q, const S* u, T kf, T L, T km, const T* grav, S* F, S* tau) {
    if constexpr (WRENCH == WRENCH_SATELLITE) {
        quat_rotate(q, u, F);
        tau[0] = u[3]; tau[1] = u[4]; tau[2] = u[5];
    } else {
        const S F1 = max0_(u[0] * kf), F2 = max0_(u[1] * kf), F3 = max0_(u[2] * kf), F4 = max0_(u[3] * kf);
        const S fb[3] = {S(T(0)), S(T(0)), F1 + F2 + F3 + F4};
        S qF[3]; quat_rotate(q, fb, qF);
#pragma unroll
        for (int i = 0; i < 3; ++i) F[i] = qF[i] + grav[i];
        tau[0] = (F2 - F4) * L; tau[1] = (F3 - F1) * L;
        tau[2] = (u[0] - u[1] + u[2] - u[3]) * km;
    }
}
// ROT3: the attitude is an MRP / RodriguesParam (n = 12) and acts through its unit quaternion (RigidBody3::to_quat)
template <class Mdl, int WRENCH, bool ROT3> struct WrenchFun {
    using T = typename Mdl::T;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = 6;
    Mdl m;
    template <class S> RD_DEV void eval(const S* x, const S* u, S* y) const {
        if constexpr (ROT3) { S q[4]; m.to_quat(x + 3, q); rigid_wrench<WRENCH, T, S>(q, u, m.kf, m.L, m.km, m.grav, y, y + 3); }
        else rigid_wrench<WRENCH, T, S>(x + 3, u, m.kf, m.L, m.km, m.grav, y, y + 3);
    }
};

// W[:, k] = wrenches(x_k, u_k)   /   F[:, :, k] = d wrenches / d z at z_k      (JAC selects)
template <class Fn, class T, bool JAC>
__global__ void __launch_bounds__(TILE) wrench_kernel(Fn fn, long long N, const T* __restrict__ Z, T* __restrict__ Out) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = 6, OW = JAC ? NY * NZ : NY;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    T* tile = reinterpret_cast<T*>(smem_w);                       // [64][OW | 1]: odd row stride, lane-strided writes hit distinct banks
    constexpr int RS = OW | 1;
    const long long k0 = (long long)blockIdx.x * TILE;
    const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE), lane = threadIdx.x;
    const long long k = k0 + (lane < nk ? lane : nk - 1);
    T x[NX], u[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Z[k * NZ + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = Z[k * NZ + NX + i];
    if constexpr (!JAC) {
        T y[NY]; fn.eval(x, u, y);
#pragma unroll
        for (int i = 0; i < NY; ++i) tile[lane * RS + i] = y[i];
    } else {
        using D = Dual<T, 1>;
        for (int c = 0; c < NZ; ++c) {                            // run-time column: the wrench is a few dozen operations
            D xd[NX], ud[NU], yd[NY];
#pragma unroll
            for (int i = 0; i < NX; ++i) { xd[i].v = x[i]; xd[i].d[0] = (i == c) ? T(1) : T(0); }
#pragma unroll
            for (int i = 0; i < NU; ++i) { ud[i].v = u[i]; ud[i].d[0] = (NX + i == c) ? T(1) : T(0); }
            fn.eval(xd, ud, yd);
#pragma unroll
            for (int i = 0; i < NY; ++i) tile[lane * RS + c * NY + i] = yd[i].d[0];
        }
    }
    __syncthreads();
    T* dst = Out + k0 * OW;
    for (int e = lane; e < nk * OW; e += TILE) { const int r = e / OW; dst[e] = tile[r * RS + (e - r * OW)]; }
}

template <class Fn, class T> static int launch_wrench(const Fn& fn, const LaunchArgs& a, bool jac) {
    if (a.N == 0) return RDHIP_OK;
    constexpr int NZ = Fn::NX + Fn::NU;
    const unsigned grid = (unsigned)((a.N + TILE - 1) / TILE);
    if (jac) {
        constexpr size_t lds = (size_t)TILE * ((6 * NZ) | 1) * sizeof(T);
        auto kern = wrench_kernel<Fn, T, true>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return RDHIP_ERR_HIP;
        RD_LAUNCH(kern, dim3(grid), dim3(TILE), lds, a.stream, fn, a.N, (const T*)a.Z, (T*)a.J);
    } else {
        constexpr size_t lds = (size_t)TILE * 7 * sizeof(T);
        RD_LAUNCH((wrench_kernel<Fn, T, false>), dim3(grid), dim3(TILE), lds, a.stream, fn, a.N, (const T*)a.Z, (T*)a.Y);
    }
    return hipGetLastError() == hipSuccess ? RDHIP_OK : RDHIP_ERR_HIP;
}
template <class T, int WRENCH> static int wrench_T(const ModelDesc& d, const LaunchArgs& a, bool jac) {
    if (d.rot == RDHIP_ROT_QUAT) {
        using Mdl = RigidBody<T, WRENCH, false, false>;          // (the wrench sees neither the velocity frame nor the inertia)
        return launch_wrench<WrenchFun<Mdl, WRENCH, false>, T>(WrenchFun<Mdl, WRENCH, false>{Mdl::make(d.params)}, a, jac);
    }
    using Mdl = RigidBody3<T, WRENCH>;
    return launch_wrench<WrenchFun<Mdl, WRENCH, true>, T>(WrenchFun<Mdl, WRENCH, true>{Mdl::make(d.params, d.rot, d.frame == RDHIP_FRAME_BODY)}, a, jac);
}
// a.Y: W (6 x N) when !jac; a.J: F (6 x (n+m) x N) when jac
int fam_wrench(const ModelDesc& d, int dtype, const LaunchArgs& a, int jac) {
    if (d.kind != RDHIP_MODEL_SATELLITE && d.kind != RDHIP_MODEL_QUADROTOR) return RDHIP_ERR_UNSUPPORTED;
    if (dtype != RDHIP_F64 && dtype != RDHIP_F32) return RDHIP_ERR_INVALID_ARG;
    if (d.kind == RDHIP_MODEL_SATELLITE) return dtype == RDHIP_F64 ? wrench_T<double, WRENCH_SATELLITE>(d, a, jac != 0) : wrench_T<float, WRENCH_SATELLITE>(d, a, jac != 0);
    return dtype == RDHIP_F64 ? wrench_T<double, WRENCH_QUADROTOR>(d, a, jac != 0) : wrench_T<float, WRENCH_QUADROTOR>(d, a, jac != 0);
}

}  // namespace rd
