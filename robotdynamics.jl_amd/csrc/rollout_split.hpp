// rollout_split.hpp -- rollout! (src/trajectories.jl:436-441) on SPLIT arrays: states X n x K x M written, controls U m x (K-1) x M read
// (SURVEY 8b sketches the entry point: rdhip_rollout(x0[n x M], U[m x (K-1) x M], ... -> X)).
//
// Why: the reference's KnotPoint interleaves state and control, z = [x; u], and an in-place rollout of a packed stack has to write whole
// knot points back -- a store pattern that leaves 48 of every 152 bytes out pays a read-modify-write at the memory side -- so rdhip_rollout's streaming kernel
// moves (n+m) T in and (n+m) T out per knot point: 1.93 x the algorithmic m T + 16 in, n T out (profiles/r05/pmc_kernels.md, 32 768 x 32 fp64).
// With the controls in an array of their own the rollout reads U, t, dt once and writes X once, every line whole:
//   satellite fp64: 48 + 16 + 104 = 168 B per knot point instead of 320.
//
// Organisation = rollout_stream.hpp's: a workgroup owns 64 trajectories; four IO wavefronts bring a chunk of C steps' controls (+ t, dt) into LDS and
// take the chunk's C states out, one compute wavefront (lane = trajectory) advances the chunk in between; the next chunk's inputs are requested
// before the finished one is stored.  ONE instantiation per (model, rule, dtype) serves both layouts of the batch:
//   trajectory-major  X[(j K + k) n + i],      U[(j (K-1) + k) m + i]   -- M consecutive trajectories (a Julia Array{T,3} n x K x M)
//   time-major        X[(k M + j) n + i],      U[(k M + j) m + i]       -- knot k of all trajectories contiguous
// A chunk consists of RUNS of contiguous elements (time-major: C runs of 64 n; trajectory-major: 64 runs of C n) which the IO waves move element by
// element, consecutive lanes consecutive addresses (8-byte units in fp64: no alignment requirement on K, M or the array bases), into an LDS image
// with the same run order; the compute lanes address it with the layout's strides.
#pragma once
#include "kernels.hpp"

namespace rd {

constexpr int SPLIT_IO_WAVES = 4, SPLIT_THREADS = (1 + SPLIT_IO_WAVES) * TILE;

template <class Fn, class T> struct SplitRollCfg {
    static constexpr int NX = Fn::NX, NU = Fn::NU, SZ = (int)sizeof(T);
    static constexpr int chunk() { int c = 8; while (c > 1 && TILE * c * (NX + NU) * SZ > 40 * 1024) c /= 2; return c; }
    static constexpr int C = chunk();
    // trajectory-major runs are padded to an odd number of 8-byte words: the compute lanes sit one run apart
    static constexpr int pad_run(int elems) { int w = (elems * SZ + 7) / 8; if (w % 2 == 0) ++w; return w * 8 / SZ; }
    static constexpr int XRUN_TJ = pad_run(C * NX), URUN_TJ = pad_run(C * NU);
    static constexpr int X_ELEMS = (TILE * XRUN_TJ > C * TILE * NX ? TILE * XRUN_TJ : C * TILE * NX);
    static constexpr int U_ELEMS = (TILE * URUN_TJ > C * TILE * NU ? TILE * URUN_TJ : C * TILE * NU);
    static_assert(NU >= 1, "models without controls have nothing to read");
    static constexpr size_t X_BYTES = ((size_t)X_ELEMS * SZ + 15) / 16 * 16, U_BYTES = ((size_t)U_ELEMS * SZ + 15) / 16 * 16;
    static constexpr size_t BYTES = X_BYTES + U_BYTES + (size_t)2 * C * TILE * sizeof(double);
};

template <class Fn, class T>
__global__ void __launch_bounds__(SPLIT_THREADS, 3) rollout_split_kernel(Fn fn, long long K, long long M, int tm, T* __restrict__ X, const T* __restrict__ U,
                                                                         const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0) {
    using L = SplitRollCfg<Fn, T>;
    constexpr int NX = L::NX, NU = L::NU, C = L::C, NIOT = SPLIT_IO_WAVES * TILE;
    constexpr int XE = C * TILE * NX, UE = C * TILE * NU;                                   // elements of a full chunk
    constexpr int XPT = (XE + NIOT - 1) / NIOT, UPT = (UE + NIOT - 1) / NIOT, TPT = (C * TILE + NIOT - 1) / NIOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);
    T* us = reinterpret_cast<T*>(smem + L::X_BYTES);
    double* ts = reinterpret_cast<double*>(smem + L::X_BYTES + L::U_BYTES);              // t[s][j], dt[s][j]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long jb = (long long)blockIdx.x * TILE;
    const int nv = (int)((M - jb) < TILE ? (M - jb) : TILE);
    const long long nsteps = K - 1;
    const int nch = (int)((K + C - 1) / C);                                                 // chunks of knots (the states); chunk c needs the steps c C .. c C + C - 1
    // run geometry of the layout (elements): a chunk = nrun runs of xrun (urun) elements, LDS run stride xrs (urs), global run stride xgs (ugs)
    const int xrun = tm ? TILE * NX : C * NX, urun = tm ? TILE * NU : C * NU;
    const int xrs = tm ? TILE * NX : L::XRUN_TJ, urs = tm ? TILE * NU : L::URUN_TJ;
    const long long xgs = tm ? M * NX : K * NX, ugs = tm ? M * NU : nsteps * NU;
    if (wave > 0) {
        // =========================== IO wavefronts ===========================
        const int tid = threadIdx.x - TILE;
        T rin[UPT]; double rt[TPT], rh[TPT];
        auto request = [&](int c) {                                  // controls and times of the steps of chunk c
            const long long k0 = (long long)c * C;
            const long long left = nsteps - k0;
            const int ns = (int)(left < 0 ? 0 : (left < C ? left : C));                    // steps of this chunk that exist
            const int nr = tm ? ns : nv, valid = tm ? nv * NU : ns * NU;                    // runs that exist, elements of a run that exist
            if (ns > 0) {                                            // (the last chunk of a K = c C + 1 rollout holds the final state and no step: nothing to read)
                const T* base = U + (tm ? (k0 * M + jb) * NU : (jb * nsteps + k0) * NU);
#pragma unroll
                for (int i = 0; i < UPT; ++i) {                     // runs / elements that do not exist re-read element 0 of the chunk: the loads stay unconditional
                    const int u = i * NIOT + tid;
                    const int r = tm ? u / (TILE * NU) : u / (C * NU), o = u - r * urun;
                    const bool ok = u < UE && r < nr && o < valid;
                    rin[i] = base[ok ? r * ugs + o : 0];
                }
            }
            const int nk = (int)((K - k0) < C ? (K - k0) : C);
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                const int e = i * NIOT + tid;
                int j = tm ? e % TILE : e / C, s = tm ? e / TILE : e % C;
                j = j < nv ? j : nv - 1; s = s < nk ? s : nk - 1;
                const long long idx = tm ? (k0 + s) * M + jb + j : (jb + j) * K + k0 + s;
                rt[i] = tt[idx]; rh[i] = dtt[idx];
            }
        };
        request(0);
        for (int c = 0; c <= nch; ++c) {
            T rout[XPT];
            __syncthreads();                                         // barrier A: the states of chunk c - 1 are complete in LDS
            if (c > 0) {
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const int u = i * NIOT + tid;
                    const int r = tm ? u / (TILE * NX) : u / (C * NX), o = u - r * xrun;
                    rout[i] = xs[u < XE ? r * xrs + o : 0];
                }
            }
            if (c < nch) {
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                    const int u = i * NIOT + tid;
                    const int r = tm ? u / (TILE * NU) : u / (C * NU), o = u - r * urun;
                    if (u < UE) us[r * urs + o] = rin[i];
                }
#pragma unroll
                for (int i = 0; i < TPT; ++i) {
                    const int e = i * NIOT + tid;
                    const int j = tm ? e % TILE : e / C, s = tm ? e / TILE : e % C;
                    if (e < C * TILE) { ts[s * TILE + j] = rt[i]; ts[(C + s) * TILE + j] = rh[i]; }
                }
            }
            __syncthreads();                                         // barrier B: the inputs of chunk c are in LDS
            // the next chunk is requested BEFORE the finished one is stored: the loads are then the older operations (one in-order counter)
            if (c + 1 < nch) request(c + 1);
            if (c > 0) {
                const long long k0 = (long long)(c - 1) * C;
                const int nk = (int)((K - k0) < C ? (K - k0) : C);
                const int nr = tm ? nk : nv, valid = tm ? nv * NX : nk * NX;
                T* base = X + (tm ? (k0 * M + jb) * NX : (jb * K + k0) * NX);
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const int u = i * NIOT + tid;
                    const int r = tm ? u / (TILE * NX) : u / (C * NX), o = u - r * xrun;
                    if (u < XE && r < nr && o < valid) __builtin_nontemporal_store(rout[i], base + r * xgs + o);
                }
            }
        }
        return;
    }
    // =========================== compute wavefront: lane = trajectory ===========================
    T x[NX], y[NX];
    const int jl = lane < nv ? lane : nv - 1;
    {
        const T* x0 = X0 ? X0 + (jb + jl) * NX : X + (tm ? (jb + jl) * NX : (jb + jl) * K * NX);     // knot 0 of the trajectory
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = x0[i];
    }
    // element (knot s of the chunk, trajectory j, component i) of the LDS images
    const int xj = tm ? NX : L::XRUN_TJ, xk = tm ? TILE * NX : NX, uj = tm ? NU : L::URUN_TJ, uk = tm ? TILE * NU : NU;
    for (int c = 0; c <= nch; ++c) {
        __syncthreads();                                             // barrier A
        __syncthreads();                                             // barrier B: chunk c's controls are in LDS, chunk c - 1's states are in the IO waves' registers
        if (c == nch) break;
        const long long k0 = (long long)c * C;
        const int nk = (int)((K - k0) < C ? (K - k0) : C);
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[lane * xj + i] = x[i];      // the state this chunk starts from is its first knot point
#pragma unroll
        for (int s = 0; s < C; ++s) {
            if (s < nk && k0 + s < nsteps) {
                const double h = ts[(C + s) * TILE + lane];
                const T tk = (T)ts[s * TILE + lane];
                T u[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) { const T v = us[lane * uj + s * uk + i]; u[i] = is_terminal_dt(h) ? T(0) : v; }
                fn.eval(x, u, tk, (T)h, y);
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = y[i];
                if (s + 1 < C) {
                    if (s + 1 < nk) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) xs[lane * xj + (s + 1) * xk + i] = y[i];
                    }
                }
            }
        }
    }
}

}  // namespace rd
