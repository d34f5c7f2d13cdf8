// rollout_stream.hpp -- rollout! (src/trajectories.jl:436-441) for BIG batches of trajectories: the throughput regime.
//
// With tens of thousands of trajectories a rollout is no latency chain any more: every SIMD has work, and what bounds the launch is how the
// knot points move.  One lane per trajectory (rollout_kernel, and the compute waves' direct path of cascade.hpp) touches 64 different cache
// lines per memory instruction and relies on the L2 to keep the lines a lane walks along; that holds up to ~16 000 trajectories (0.6 of the
// HBM roofline at 16 384 x 64) and collapses beyond (0.30-0.38 at 65 536 x 32 and 262 144 x 16: the lines of 65 536 lanes do not fit the L2s).
// Here a workgroup owns 64 trajectories and streams them through LDS in chunks of C knot points:
//
//  * the chunk's knot points -- states AND controls, exactly the bytes of Z -- arrive in LDS as whole 16-byte units of contiguous runs (a
//    trajectory's C knot points for the reference's layout, 64 trajectories' knot k for time-major stacks), are advanced IN PLACE by one compute
//    wavefront (lane = trajectory: the new state of knot k + 1 overwrites the stale one in LDS) and leave as the same runs, whole 128-byte lines:
//    the controls are written back unchanged, because a store pattern that leaves 48 of every 152 bytes out pays a read-modify-write at the
//    memory side ("whole lines or it pays").  HBM traffic: (n+m) T read + (n+m) T written + 16 per knot point, against the
//    algorithmic (n+m) T + n T + 16;
//  * IO wavefronts move the chunks: chunk c + 1 is requested into their registers before chunk c - 1 is stored, the LDS tile is swapped
//    between two workgroup barriers (results out, next inputs in) -- the split-role organisation of the Jacobian tile loop;
//  * two workgroups per CU overlap one group's arithmetic with the other's traffic.
//
// What the counters say (tools/pmc_stream.sh, 262 144 x 16 satellite knot points, fp64): this kernel fetches 1.04 x and writes 1.04 x the bytes of
// Z + t + dt / Z -- every line once each way, 334 B per knot point -- and so does rollout_kernel (1.27 x fetched, 0.95 x written: its 104-byte
// state stores cost whole sectors, its 48-byte control loads whole lines).  An in-place rollout therefore moves ~1.23 x its algorithmic bytes
// whatever the access pattern, and the memory system sustains 4.6-4.8 TB/s of that half-read, half-write traffic: 0.49 (time-major) / 0.41
// (trajectory-major: a trajectory's 4-knot run starts and ends inside a line) of the roofline in algorithmic bytes, against 0.44 / 0.37 for
// rollout_kernel and 0.30-0.36 for the cascade kernel, whose strength is the latency of a step, not the traffic.
#pragma once
#include "kernels.hpp"

namespace rd {

// four IO wavefronts (ten 16-byte units each way per thread) and a register budget of three waves per SIMD: measured against two IO waves (the
// compiler keeps every unit's address live: 900 B of scratch per lane, 3x slower) and against four waves per SIMD (the fp64 RK4 step of a rigid
// body spills at 128 registers: 2x slower).  Two workgroups of five waves are resident per CU.
#ifndef STREAM_IO_WAVES_N
#define STREAM_IO_WAVES_N 4
#endif
#ifndef STREAM_WAVES_PER_SIMD
#define STREAM_WAVES_PER_SIMD 3
#endif
constexpr int STREAM_IO_WAVES = STREAM_IO_WAVES_N, STREAM_THREADS = (1 + STREAM_IO_WAVES) * TILE;

template <class Fn, class T, bool TM> struct StreamCfg {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, SZ = (int)sizeof(T);
    static constexpr int chunk_knots() { int c = 8; while (c > 1 && TILE * c * NZ * SZ > 40 * 1024) c /= 2; return c; }
    static constexpr int C = chunk_knots();                                       // knot points per chunk
    static constexpr int NRUN = TM ? C : TILE;                                    // contiguous runs of a chunk
    static constexpr int RUN_ELEMS = TM ? TILE * NZ : C * NZ, RUN_BYTES = RUN_ELEMS * SZ;
    static constexpr int RS = (RUN_BYTES + 15) / 16 * 16 + (TM ? 0 : 16);         // LDS run stride (trajectory runs padded: the lanes of the compute wave sit a run apart)
    static constexpr bool UNITS = RUN_BYTES % 16 == 0;                            // runs are whole 16-byte units (given aligned run starts: checked by the host)
    static constexpr size_t Z_BYTES = (size_t)NRUN * RS, TS_OFF = Z_BYTES, BYTES = TS_OFF + (size_t)2 * C * TILE * sizeof(double);
    // byte offset of component i of (trajectory j, knot s of the chunk) in the LDS tile
    static RD_DEV int at(int j, int s, int i) { return TM ? s * RS + (j * NZ + i) * SZ : j * RS + (s * NZ + i) * SZ; }
};

// U: the unit the IO wavefronts move (16 bytes when every run starts and ends on a 16-byte boundary, else one element)
template <class Fn, class T, bool TM, class U>
__global__ void __launch_bounds__(STREAM_THREADS, STREAM_WAVES_PER_SIMD) rollout_stream_kernel(Fn fn, long long K, long long M, T* __restrict__ Z, const double* __restrict__ tt,
                                                                        const double* __restrict__ dtt, const T* __restrict__ X0) {
    using L = StreamCfg<Fn, T, TM>;
    constexpr int NX = L::NX, NU = L::NU, NZ = L::NZ, SZ = L::SZ, C = L::C, USZ = (int)sizeof(U);
    constexpr int NIOT = STREAM_IO_WAVES * TILE;
    constexpr int UPR = L::RUN_BYTES / USZ, NUNIT = L::NRUN * UPR, UPT = (NUNIT + NIOT - 1) / NIOT;      // units per run / chunk / IO thread
    constexpr int TPT = (C * TILE + NIOT - 1) / NIOT;                                                   // time stamps per IO thread (each of t, dt)
    static_assert(L::RUN_BYTES % USZ == 0, "runs are whole units");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ts = reinterpret_cast<double*>(smem + L::TS_OFF);                     // t[s][j], dt[s][j]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long jb = (long long)blockIdx.x * TILE;
    const int nv = (int)((M - jb) < TILE ? (M - jb) : TILE);                      // trajectories of this group
    const int nch = (int)((K + C - 1) / C);
    if (wave > 0) {
        // =========================== IO wavefronts ===========================
        const int tid = threadIdx.x - TILE;
        // unit u of a chunk: run r = u / UPR, unit o of the run; runs / units that do not exist re-read run 0 / unit 0 (loads stay unconditional)
        auto run_ptr = [&](long long k0, int r) -> unsigned char* {
            const long long knot = TM ? (k0 + r) * M + jb : (jb + r) * K + k0;
            return reinterpret_cast<unsigned char*>(Z + knot * NZ);
        };
        U rin[UPT]; double rt[TPT], rh[TPT];
        auto request = [&](int c) {
            const long long k0 = (long long)c * C;
            const int nk = (int)((K - k0) < C ? (K - k0) : C);
            const int nr = TM ? nk : nv, rb = (TM ? nv : nk) * NZ * SZ;
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int u = i * NIOT + tid, r = u / UPR, o = u - r * UPR;
                const bool ok = r < nr && o * USZ + USZ <= rb;
                const U* src = reinterpret_cast<const U*>(run_ptr(k0, ok ? r : 0) + (ok ? o : 0) * USZ);
                if constexpr (TM) rin[i] = __builtin_nontemporal_load(src); else rin[i] = *src;
            }
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                const int e = i * NIOT + tid;
                int j = TM ? e % TILE : e / C, s = TM ? e / TILE : e % C;
                j = j < nv ? j : nv - 1; s = s < nk ? s : nk - 1;
                const long long idx = TM ? (k0 + s) * M + jb + j : (jb + j) * K + k0 + s;
                rt[i] = tt[idx]; rh[i] = dtt[idx];
            }
        };
        request(0);
        for (int c = 0; c <= nch; ++c) {
            // chunk c - 1 is finished in LDS, chunk c sits in registers
            U rout[UPT];
            __syncthreads();                                   // barrier A
            if (c > 0) {
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                    const int u = i * NIOT + tid, r = u / UPR, o = u - r * UPR;
                    rout[i] = *reinterpret_cast<const U*>(smem + (u < NUNIT ? r * L::RS + o * USZ : 0));
                }
            }
            if (c < nch) {
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                    const int u = i * NIOT + tid, r = u / UPR, o = u - r * UPR;
                    if (u < NUNIT) *reinterpret_cast<U*>(smem + r * L::RS + o * USZ) = rin[i];
                }
#pragma unroll
                for (int i = 0; i < TPT; ++i) {
                    const int e = i * NIOT + tid;
                    const int j = TM ? e % TILE : e / C, s = TM ? e / TILE : e % C;
                    if (e < C * TILE) { ts[s * TILE + j] = rt[i]; ts[(C + s) * TILE + j] = rh[i]; }
                }
            }
            __syncthreads();                                   // barrier B
            // the next chunk is requested BEFORE the finished one is stored: the loads are then the older operations, and waiting for them does
            // not wait for the stores
            if (c + 1 < nch) request(c + 1);
            if (c > 0) {
                const long long k0 = (long long)(c - 1) * C;
                const int nk = (int)((K - k0) < C ? (K - k0) : C);
                const int nr = TM ? nk : nv, rb = (TM ? nv : nk) * NZ * SZ;
#pragma unroll
                for (int i = 0; i < UPT; ++i) {
                    const int u = i * NIOT + tid, r = u / UPR, o = u - r * UPR;
                    // (non-temporal: written once, not re-read by this launch; time-major stacks also LOAD non-temporally -- 0.50 -> 0.53-0.56 --, the
                    // reference's layout does not: the runs of one trajectory's consecutive chunks share their first / last line)
                    if (r < nr && o * USZ + USZ <= rb) __builtin_nontemporal_store(rout[i], reinterpret_cast<U*>(run_ptr(k0, r) + o * USZ));
                }
            }
        }
        return;
    }
    // =========================== compute wavefront: lane = trajectory ===========================
    T x[NX], y[NX];
    const int jl = lane < nv ? lane : nv - 1;
    if (X0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X0[(jb + jl) * NX + i];
    }
    for (int c = 0; c <= nch; ++c) {
        __syncthreads();                                       // barrier A
        __syncthreads();                                       // barrier B: chunk c is in LDS
        if (c == nch) break;
        const long long k0 = (long long)c * C;
        const int nk = (int)((K - k0) < C ? (K - k0) : C);
        if (c == 0 && !X0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = *reinterpret_cast<const T*>(smem + L::at(jl, 0, i));
        }
        // the state this chunk starts from goes into its first knot point (knot 0: the initial state)
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<T*>(smem + L::at(lane, 0, i)) = x[i];
#pragma unroll
        for (int s = 0; s < C; ++s) {
            if (s < nk && k0 + s < K - 1) {
                const double h = ts[(C + s) * TILE + lane];
                const T tk = (T)ts[s * TILE + lane];
                T u[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) { const T v = *reinterpret_cast<const T*>(smem + L::at(lane, s, NX + i)); u[i] = is_terminal_dt(h) ? T(0) : v; }
                fn.eval(x, u, tk, (T)h, y);
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = y[i];
                if (s + 1 < C) {
                    if (s + 1 < nk) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) *reinterpret_cast<T*>(smem + L::at(lane, s + 1, i)) = y[i];
                    }
                }
            }
        }
    }
}

}  // namespace rd
