// kernels.hpp -- the batched kernels.  Data layout in HBM (all column-major, knot point k contiguous):
//   Z  (n+m) x N      t, dt  N (fp64)      Y  n x N      J  n x (n+m) x N
// Mapping (gfx950, wave64): one LANE owns one knot point, one WAVEFRONT owns 64 knot points x a group of
// Jacobian columns, one WORKGROUP (NG waves) owns a tile of 64 knot points.  The z tile is read once, coalesced,
// into LDS and shared by the NG waves; every finished Jacobian column (n values per lane) is transposed through
// a per-wave LDS buffer so that the global stores are lane-contiguous.  Everything between load and store lives
// in registers; there is no inter-workgroup data reuse, so no XCD-aware remap is needed (each 64-knot tile is
// touched by exactly one workgroup).
#pragma once
#include "models.hpp"

namespace rd {

constexpr int TILE = 64;   // knot points per wave = wave width on gfx950

template <class T> RD_DEV T sqrt_eps();
template <> RD_DEV double sqrt_eps<double>() { return 1.4901161193847656e-08; }   // sqrt(eps(Float64))
template <> RD_DEV float sqrt_eps<float>() { return 3.4526698300124393e-04f; }     // sqrt(eps(Float32))

RD_DEV void wave_sync() {
    // LDS hand-off inside ONE wave: DS ops of a wave execute in order, we only need the compiler to keep them so.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// load knot point `lane` of the tile from the LDS copy; apply the terminal-knot rule to the control
template <class Fn, class T>
RD_DEV void read_knot(const T* zs, int lane, bool term, T (&x)[Fn::NX], T (&u)[Fn::NU]) {
    constexpr int NZ = Fn::NX + Fn::NU;
#pragma unroll
    for (int i = 0; i < Fn::NX; ++i) x[i] = zs[lane * NZ + i];
#pragma unroll
    for (int i = 0; i < Fn::NU; ++i) u[i] = term ? T(0) : zs[lane * NZ + Fn::NX + i];
}

// transpose one per-lane column (NY values) through the wave's LDS buffer and store it lane-contiguously:
// element e = kn*NY + i of the tile column goes to dst[kn*stride + i]
template <int NY, class T>
RD_DEV void store_column(T* buf, const T (&col)[NY], T* dst, long long stride, int lane, int nk) {
#pragma unroll
    for (int i = 0; i < NY; ++i) buf[lane * NY + i] = col[i];
    wave_sync();
#pragma unroll
    for (int it = 0; it < NY; ++it) {
        const int e = it * TILE + lane;
        const int kn = e / NY, i = e - kn * NY;
        if (kn < nk) dst[kn * stride + i] = buf[e];
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------------------
// evaluate: Y[:,k] = fun(z_k)      (dynamics / discrete_dynamics)
// ------------------------------------------------------------------------------------------------------------
template <class Fn, class T>
__global__ void __launch_bounds__(256) eval_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                   const double* __restrict__ dtt, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NB = 256;
    __shared__ T lds[NB * (NZ > NY ? NZ : NY)];
    const long long k0 = (long long)blockIdx.x * NB;
    const int nk = (int)((N - k0) < NB ? (N - k0) : NB);
    for (int e = threadIdx.x; e < nk * NZ; e += NB) lds[e] = Z[k0 * NZ + e];
    __syncthreads();
    const int kl = threadIdx.x < nk ? threadIdx.x : nk - 1;
    const T t = (T)tt[k0 + kl], h = (T)dtt[k0 + kl];
    T x[NX], u[NU], y[NY];
    read_knot<Fn, T>(lds, kl, dtt[k0 + kl] == 0.0, x, u);
    fn.eval(x, u, t, h, y);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NY; ++i) lds[threadIdx.x * NY + i] = y[i];
    __syncthreads();
    for (int e = threadIdx.x; e < nk * NY; e += NB) Y[k0 * NY + e] = lds[e];
}

// ------------------------------------------------------------------------------------------------------------
// rollout: thread j simulates trajectory j, sequential in k (src/trajectories.jl:436-441)
// ------------------------------------------------------------------------------------------------------------
template <class Fn, class T>
__global__ void __launch_bounds__(64) rollout_kernel(Fn fn, long long K, long long M, T* __restrict__ Z, const double* __restrict__ tt,
                                                     const double* __restrict__ dtt, const T* __restrict__ X0) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    T* Zj = Z + j * K * NZ;
    T x[NX], u[NU], y[NX];
    if (X0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = X0[j * NX + i]; Zj[i] = x[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = Zj[i];
    }
    for (long long k = 1; k < K; ++k) {
        const double h = dtt[j * K + k - 1];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = (h == 0.0) ? T(0) : Zj[(k - 1) * NZ + NX + i];
        fn.eval(x, u, (T)tt[j * K + k - 1], (T)h, y);
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = y[i]; Zj[k * NZ + i] = y[i]; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Jacobians.  Common frame: workgroup = NG waves, wave g owns columns [g*NZ/NG, (g+1)*NZ/NG).
// ------------------------------------------------------------------------------------------------------------
template <class Fn, class T, int NG> struct JacFrame {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY;
    T x[NX], u[NU];
    T t, h;
    bool term;
    int lane, g, nk;
    T* buf;          // this wave's LDS staging buffer (TILE*NY)
    T* Jt;           // J + k0*NY*NZ
    T* Yt;           // Y + k0*NY or nullptr
    RD_DEV void put_column(int c, const T (&col)[NY]) { store_column<NY, T>(buf, col, Jt + (long long)c * NY, (long long)NY * NZ, lane, nk); }
    RD_DEV void put_value(const T (&y)[NY]) { if (Yt) store_column<NY, T>(buf, y, Yt, (long long)NY, lane, nk); }
};

#define RD_JAC_PROLOGUE(NGv)                                                                                  \
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY;                                        \
    __shared__ T zs[TILE * NZ];                                                                               \
    __shared__ T stage[NGv * TILE * NY];                                                                      \
    JacFrame<Fn, T, NGv> fr;                                                                                  \
    fr.lane = threadIdx.x & 63; fr.g = threadIdx.x >> 6;                                                      \
    const long long k0 = (long long)blockIdx.x * TILE;                                                        \
    fr.nk = (int)((N - k0) < TILE ? (N - k0) : TILE);                                                         \
    for (int e = threadIdx.x; e < fr.nk * NZ; e += NGv * TILE) zs[e] = Z[k0 * NZ + e];                        \
    __syncthreads();                                                                                          \
    const int kl = fr.lane < fr.nk ? fr.lane : fr.nk - 1;                                                     \
    const double hd = dtt[k0 + kl];                                                                           \
    fr.t = (T)tt[k0 + kl]; fr.h = (T)hd; fr.term = (hd == 0.0);                                               \
    read_knot<Fn, T>(zs, kl, fr.term, fr.x, fr.u);                                                            \
    fr.buf = stage + fr.g * TILE * NY;                                                                        \
    fr.Jt = J + k0 * NY * NZ; fr.Yt = Y ? Y + k0 * NY : nullptr;

// ---- ForwardAD, compile-time columns: KW tangent directions per pass, seeds are constants --------------------
template <class Fn, class T, int NG, int KW, int C, int C1, bool FIRST> struct AdCols {   // FIRST: this pass also emits f(z)
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T, NG>& fr) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY;
        constexpr int W = (C1 - C) < KW ? (C1 - C) : KW;
        if constexpr (W > 0) {
            using D = Dual<T, W>;
            D xd[NX], ud[NU], yd[NY];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                xd[i].v = fr.x[i];
#pragma unroll
                for (int c = 0; c < W; ++c) xd[i].d[c] = (i == C + c) ? T(1) : T(0);
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                ud[i].v = fr.u[i];
#pragma unroll
                for (int c = 0; c < W; ++c) ud[i].d[c] = (NX + i == C + c) ? T(1) : T(0);
            }
            fn.eval(xd, ud, fr.t, fr.h, yd);
            if constexpr (FIRST) {
                T yv[NY];
#pragma unroll
                for (int i = 0; i < NY; ++i) yv[i] = yd[i].v;
                fr.put_value(yv);
            }
#pragma unroll
            for (int c = 0; c < W; ++c) {
                T col[NY];
                // control columns of a terminal knot point are zero (`!is_terminal(z) * u`, src/knotpoint.jl:67)
                const T sc = (C + c >= NX && fr.term) ? T(0) : T(1);
#pragma unroll
                for (int i = 0; i < NY; ++i) col[i] = yd[i].d[c] * sc;
                fr.put_column(C + c, col);
            }
            AdCols<Fn, T, NG, KW, C + W, C1, false>::run(fn, fr);
        }
    }
};
template <class Fn, class T, int NG, int KW, int G> struct AdGroups {
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T, NG>& fr) {
        constexpr int NZ = Fn::NX + Fn::NU;
        if constexpr (G < NG) {
            if (fr.g == G) AdCols<Fn, T, NG, KW, G * NZ / NG, (G + 1) * NZ / NG, G == 0>::run(fn, fr);
            else AdGroups<Fn, T, NG, KW, G + 1>::run(fn, fr);
        }
    }
};
template <class Fn, class T, int NG, int KW>
__global__ void __launch_bounds__(NG * TILE) jac_ad_unrolled_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                    const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    RD_JAC_PROLOGUE(NG)
    AdGroups<Fn, T, NG, KW, 0>::run(fn, fr);
}

// ---- ForwardAD, run-time column loop (compact code for the long tail of model x rule combinations) -----------
template <class Fn, class T, int NG>
__global__ void __launch_bounds__(NG * TILE) jac_ad_loop_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    RD_JAC_PROLOGUE(NG)
    using D = Dual<T, 1>;
    const int c0 = fr.g * NZ / NG, c1 = (fr.g + 1) * NZ / NG;
    for (int c = c0; c < c1; ++c) {
        D xd[NX], ud[NU], yd[NY];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xd[i].v = fr.x[i]; xd[i].d[0] = (i == c) ? T(1) : T(0); }
#pragma unroll
        for (int i = 0; i < NU; ++i) { ud[i].v = fr.u[i]; ud[i].d[0] = (NX + i == c) ? T(1) : T(0); }
        fn.eval(xd, ud, fr.t, fr.h, yd);
        T col[NY];
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < NY; ++i) col[i] = yd[i].v;
            fr.put_value(col);
        }
        const T sc = (c >= NX && fr.term) ? T(0) : T(1);
#pragma unroll
        for (int i = 0; i < NY; ++i) col[i] = yd[i].d[0] * sc;
        fr.put_column(c, col);
    }
}

// ---- FiniteDifference: forward differences, eps_j = max(sqrt(eps)|z_j|, sqrt(eps)) (FiniteDiff.jl default) -----
template <class Fn, class T, int NG>
__global__ void __launch_bounds__(NG * TILE) jac_fd_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                           const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    RD_JAC_PROLOGUE(NG)
    T f0[NY];
    fn.eval(fr.x, fr.u, fr.t, fr.h, f0);
    if (fr.g == 0) fr.put_value(f0);
    const int c0 = fr.g * NZ / NG, c1 = (fr.g + 1) * NZ / NG;
    for (int c = c0; c < c1; ++c) {
        T x1[NX], u1[NU], f1[NY];
        T zc = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { x1[i] = fr.x[i]; if (i == c) zc = fr.x[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { u1[i] = fr.u[i]; if (NX + i == c) zc = fr.u[i]; }
        const T se = sqrt_eps<T>();
        const T az = zc < T(0) ? -zc : zc;
        const T eps = (se * az > se) ? se * az : se;
        // a terminal knot point evaluates f with u = 0 for every perturbation (control(z) = false * u)
#pragma unroll
        for (int i = 0; i < NX; ++i) if (i == c) x1[i] = x1[i] + eps;
#pragma unroll
        for (int i = 0; i < NU; ++i) if (NX + i == c && !fr.term) u1[i] = u1[i] + eps;
        fn.eval(x1, u1, fr.t, fr.h, f1);
        T col[NY];
#pragma unroll
        for (int i = 0; i < NY; ++i) col[i] = (f1[i] - f0[i]) / eps;
        fr.put_column(c, col);
    }
}

}  // namespace rd
