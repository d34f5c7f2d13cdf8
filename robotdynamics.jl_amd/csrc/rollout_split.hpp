// rollout_split.hpp -- rollout! (src/trajectories.jl:436-441) on SPLIT arrays: states X n x K x M written, controls U m x (K-1) x M read
// (SURVEY 8b sketches the entry point: rdhip_rollout(x0[n x M], U[m x (K-1) x M], ... -> X)).
//
// Why: the reference's KnotPoint interleaves state and control, z = [x; u], and an in-place rollout of a packed stack either writes whole
// knot points back (rollout_stream_kernel: (n+m) T in and (n+m) T out per knot point, 1.93 x the algorithmic bytes of a 32 768 x 32 fp64 batch,
// profiles/r05/pmc_kernels.md) or leaves 48 of every 152 bytes of a line out (rollout_kernel: partial-line stores, strided control loads).  With the controls
// in an array of their own the rollout reads U, t, dt once and writes X once:  satellite fp64  48 + 16 + 104 = 168 B per knot point instead of 320.
//
// ONE kernel, one lane per trajectory, both layouts by strides:
//   trajectory-major  X[(j K + k) n + i],      U[(j (K-1) + k) m + i]   -- M consecutive trajectories (a Julia Array{T,3} n x K x M)
//   time-major        X[(k M + j) n + i],      U[(k M + j) m + i]       -- knot k of all trajectories contiguous: the lanes' accesses are dense
// rollout_kernel's loop on the split arrays -- controls and times requested ROLLOUT_PREFETCH steps ahead, the new state stored behind the arithmetic --, the
// same step function, the same bits as rdhip_rollout's one-lane kernel.  Time-major 32 768 x 32 fp64 satellite knots: 42 us against 66 us for the packed
// stack (0.52 against 0.33 of the roofline on the same algorithmic bytes); fp32 65 536 x 32: 38 against 63 us; cartpole 262 144 x 32: 115 against 154 us.
// (A chunk-streaming form -- a workgroup per 64 trajectories, four IO wavefronts moving chunks of controls in and states out through LDS, one compute
// wavefront, the organisation of rollout_stream.hpp -- was built first and measured SLOWER than this kernel in four cases of five: 57 / 243 / 59 us
// against 42 / 203 / 48 for the fp64 satellite, 455 against 115 us for the cartpole, whose step starves on one compute wave per workgroup, and it spilled
// in fp32; only trajectory-major 262 144 x 16 went its way, 251 against 292 us.  Not shipped; profiles/r06/split_rollout.jsonl has both.)
#pragma once
#include "kernels.hpp"

namespace rd {

// (sjx, skx) / (sju, sku) / (sj, sk): strides of trajectory and knot in X / U / t, dt (in knot points): (K, 1), (K-1, 1), (K, 1) or (1, M), (1, M), (1, M).
template <class Fn, class T>
__global__ void __launch_bounds__(64) rollout_split_lane_kernel(Fn fn, long long K, long long M, long long sjx, long long skx, long long sju, long long sku, long long sj, long long sk,
                                                                T* __restrict__ X, const T* __restrict__ U, const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0) {
    constexpr int NX = Fn::NX, NU = Fn::NU, D = ROLLOUT_PREFETCH;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    T x[NX], u[NU], y[NX];
    {
        const T* xs = X0 ? X0 + j * NX : X + j * sjx * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xs[i];
    }
    const long long nsteps = K - 1, lastu = nsteps - 1;           // K >= 2 (the host handles K < 2)
    T un[D][NU]; double hn[D], tn[D];
    auto fetch = [&](int d, long long knot) {                     // knot <= K - 2: a step that exists
        const long long it = j * sj + knot * sk, iu = j * sju + knot * sku;
        hn[d] = dtt[it]; tn[d] = tt[it];
#pragma unroll
        for (int i = 0; i < NU; ++i) un[d][i] = U[iu * NU + i];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, d < lastu ? d : lastu);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) X[j * sjx * NX + i] = x[i];
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](long long k, int d) {
        const double h = hn[d]; const T tk = (T)tn[d];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(h) ? T(0) : un[d][i];
        fetch(d, k + D < lastu ? k + D : lastu);
        __builtin_amdgcn_sched_barrier(0);
        fn.eval(x, u, tk, (T)h, y);
        T* Xk = X + (j * sjx + (k + 1) * skx) * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = y[i]; Xk[i] = y[i]; }
    };
    long long k = 0;
    for (; k + D <= nsteps; k += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(k + d, d);
    }
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) if (k + d < nsteps) step(k + d, d);
}

}  // namespace rd
