// mixed.hpp -- the mixed-model Monte-Carlo batch (BASELINE config 5: even trajectories Cartpole, odd Satellite) as ONE launch:
//   rollout!(sig, dmodel_A, Z_A, x0_A); rollout!(sig, dmodel_B, Z_B, x0_B); jacobian! at every simulated knot point of both
// (src/trajectories.jl:436-441 + src/functionbase.jl:242-257), A a model with the one-launch rollout -> Jacobians form (fused.hpp: a rigid body with a
// cascade structure and a compile-time column plan), B a SMALL model (n + m <= 10: the cartpole), time-major stacks.
//
// Round 5 ran this call as the fused launch of A on the context's stream and B's rollout + Jacobians in four pipelined pieces on a second stream,
// between a fork and a join: 257-262 us for 4 096 + 4 096 trajectories x 256 knots in fp32, of which 18 us are the three packets of fork and join
// between consecutive calls and ~20 us are B's eight kernels competing with A's launch (profiles/r05/c5_whole_timeline.txt).  Here the call
// is one grid with four roles, no fork, no join, no second stream:
//
//   workgroups 0 .. GA-1          A's rollout, one 64-trajectory group each (cascade waves + IO wave, progress words), and on the workgroup's IDLE waves
//                                 B's rollout: a wave simulates 64 of B's trajectories, one lane each (plain stores, one agent-scope release fence at
//                                 the end; the last wave to finish raises B's flag word)
//   every workgroup               A's Jacobian tiles, in the order of production (jac_split_loop over RolloutTilesTM): the Jacobian workgroups from the
//                                 start, the former rollout workgroups of EITHER model on the part of the tile list behind `split` (FusedWalk: both
//                                 kinds are "producers" to it)
//   every workgroup, at its end   B's Jacobians: 64-knot tiles of B's flat stack, one per WAVE (jac_ad_small_walk: all n + m tangents at once, the
//                                 wave's 64 x n x (n+m) results transposed through a wave-private LDS region), dealt statically over all waves of the
//                                 grid.  B's rollout finished two hundred microseconds earlier; the flag is looked at once per wave.
//
// B's Jacobians at the END, not interleaved: they are 8-10 % of the call's bytes, tiles of 5 KB that no bandwidth-sized item list wants in between A's
// 72 KB tiles, and the end of A's walk is where the launch has slack -- workgroups finish A's tiles several microseconds apart (the former rollout
// workgroups and the Jacobian workgroups: 205-213 and 208-218 us), and every one of them goes on to B's tiles the moment it is done.
// Results: A's and B's states are those of rdhip_rollout bit for bit for A (same waves, same code) and to rounding for B (the one-lane rollout compiled
// a second time); the Jacobians agree with rdhip_jacobian's to rounding, as in the fused launch.
#pragma once
#include "fused.hpp"

namespace rd {


// B's rollout of trajectory j (time-major stack: knot k of trajectory j at flat index k M + j): rollout_kernel's loop, statement for statement -- inputs
// requested D steps ahead, the new state stored behind the arithmetic as PLAIN stores, the queue shaped so that waiting for a load never waits for the
// stores behind it (kernels.hpp) -- with a deeper prefetch: beside a saturated store stream a load takes 2-3 us, and at D = 4 the cartpole's 0.17 us step
// became a 0.7 us step (what its stand-alone pieces showed beside the fused launch in round 5).  The states sit in this XCD's write-back L2 (which
// merges the 16-byte pieces of neighbouring trajectories into lines); ONE agent-scope release fence at the end of the wave's rollout writes them back
// before the wave counts itself finished, and the Jacobian waves read them with agent-scope loads at the very end of the launch.
// (Two other forms were built and measured first -- profiles/r06/c5_mixed_attempts.txt: agent-scope write-through stores per lane, 291 us for config 5:
// 4.2 million 4-byte write-through pieces; whole knot blocks through LDS as write-through lines, 337 us: the compiler put a full vmcnt drain -- the
// acknowledgement of the previous step's write-through stores -- in front of every step.)
constexpr int MIXED_PREFETCH = 8;
// rollout waves of B per workgroup of its own: one per SIMD for time-major stacks; ONE for a vector of trajectories, where every access of a wave touches 64
// cache lines and four such waves on one CU queue up on its address path (config 5's share, trajectory-major: 188 us with four, against 155 for the streams)
template <bool TM> constexpr int mixed_wpb() { return TM ? 4 : 1; }
template <class Fn, class T>
RD_DEV void mixed_small_rollout(const Fn& fn, long long j, long long K, long long sj, long long sk, T* __restrict__ Z, const double* __restrict__ tt,
                                const double* __restrict__ dtt, const T* __restrict__ X0) {          // (sj, sk) = (1, M): time-major; (K, 1): trajectory-major
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, D = MIXED_PREFETCH;
    __builtin_amdgcn_s_setprio(3);
    T x[NX], u[NU], y[NX];
    {
        const T* xs = X0 ? X0 + j * NX : Z + j * sj * NZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xs[i];
    }
    if (K < 2) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Z[j * sj * NZ + i] = x[i];
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    const long long last = K - 1, nsteps = K - 1;
    T un[D][NU]; double hn[D], tn[D];
    auto fetch = [&](int d, long long knot) {
        const long long idx = j * sj + knot * sk;
        hn[d] = dtt[idx]; tn[d] = tt[idx];
#pragma unroll
        for (int i = 0; i < NU; ++i) un[d][i] = Z[idx * NZ + NX + i];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, d < last ? d : last);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) Z[j * sj * NZ + i] = x[i];
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](long long k, int d) {
        const double h = hn[d]; const T tk = (T)tn[d];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(h) ? T(0) : un[d][i];
        fetch(d, k + D < last ? k + D : last);
        __builtin_amdgcn_sched_barrier(0);
        fn.eval(x, u, tk, (T)h, y);
        T* Zk = Z + (j * sj + (k + 1) * sk) * NZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = y[i]; Zk[i] = y[i]; }
    };
    long long k = 0;
    for (; k + D <= nsteps; k += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(k + d, d);
    }
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) if (k + d < nsteps) step(k + d, d);
    __builtin_amdgcn_s_setprio(0);
}

// words of B in the context's progress array, behind A's GA blocks: [0] the flag (call epoch << 32 | 1 once every trajectory of B is simulated), [1] the
// count of finished rollout waves (zero between launches: the wave that counts last resets it before it raises the flag)
RD_DEV unsigned long long* mixed_flag_word(unsigned long long* progress, int GA) { return progress + (size_t)GA * FUSED_PROGRESS_STRIDE; }

// A's rollout role with B's rollout on the workgroup's IDLE waves: a rollout workgroup is 512 threads (the tile loop's shape) of which A's cascade uses
// NB + 1 waves (three state blocks + the IO wave); wave NB + 1 + i simulates B's group b + i GA.  First version: B's rollout on workgroups of its own
// (16 for config 5) -- they were missing from the Jacobian workgroups for the whole first part of the walk (176 instead of 192: 232 instead of 225 us for A's
// tiles, profiles/r06/c5_mixed_attempts.txt).  B's one-lane chain beside A's cascade waves costs A's rollout a few percent (one more wave on one SIMD).
template <class FnA, class T, class Cas, int R, class Plan, int NC, class FnB, bool TM>
RD_DEV void mixed_rollout_role(const FnA& fa, long long KA, long long MA, T* __restrict__ ZA, const double* __restrict__ tA, const double* __restrict__ dtA, const T* __restrict__ X0A,
                               const FnB& fb, long long KB, long long MB, int GA, T* __restrict__ ZB, const double* __restrict__ tB, const double* __restrict__ dtB, const T* __restrict__ X0B,
                               unsigned long long* __restrict__ progress, unsigned long long pbase) {
    using L = CascadeLds<FnA, Cas, R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x < 16) reinterpret_cast<int*>(smem + L::FLAG_OFF)[threadIdx.x] = 0;
    __syncthreads();
    const bool whole = ((long long)blockIdx.x + 1) * TILE <= MA;
    const int nfull = (whole && KA >= 2) ? (int)((KA - 1) / L::S) : 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long* pword = progress + (size_t)blockIdx.x * FUSED_PROGRESS_STRIDE;
    if (wave <= Cas::NB) CascadeWaves<FnA, T, Cas, R, 0, true>::run(wave, fa, KA, MA, TM ? 1LL : KA, TM ? MA : 1LL, ZA, tA, dtA, X0A, nfull, pword, pbase);
    else {
        const long long ngB = (MB + TILE - 1) / TILE;
        const long long g = (long long)blockIdx.x + (long long)(wave - Cas::NB - 1) * GA;
        if (g < ngB) {
            long long j = g * TILE + lane;
            j = j < MB ? j : MB - 1;                            // lanes past the batch redo the last trajectory: same values to the same addresses
            mixed_small_rollout<FnB, T>(fb, j, KB, TM ? 1LL : KB, TM ? MB : 1LL, ZB, tB, dtB, X0B);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the wave's state stores are written back (and acknowledged) before it counts itself finished
            if (lane == 0) {
                unsigned long long* bflag = mixed_flag_word(progress, GA);
                unsigned* cnt = reinterpret_cast<unsigned*>(bflag + 1);
                if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ngB - 1)) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(bflag, pbase | 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    __syncthreads();                                   // every wave has seen its own stores acknowledged
    if (threadIdx.x == 0) pub_progress(pword, pbase, KA);
}

template <class FnA, class T, class Cas, int R, class Plan, int NC, class FnB, bool TM = true>
__global__ void __launch_bounds__(FUSED_THREADS, (fused_min_waves<T, R>()))
rollout_jac_mixed_kernel(FnA fa, long long KA, long long MA, int GA, T* __restrict__ ZA, const double* __restrict__ tA, const double* __restrict__ dtA, const T* __restrict__ X0A,
                         T* __restrict__ JA, FnB fb, long long KB, long long MB, int GB, T* __restrict__ ZB, const double* __restrict__ tB, const double* __restrict__ dtB,
                         const T* __restrict__ X0B, T* __restrict__ JB, unsigned long long* __restrict__ progress, unsigned long long pbase, long long split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long* bflag = mixed_flag_word(progress, GA);
    if ((int)blockIdx.x < GA) {
        // GB = 0: B's rollout rides on the idle waves of A's rollout workgroups (MB passed); GB > 0: B has workgroups of its own (MB = 0 here)
        mixed_rollout_role<FnA, T, Cas, R, Plan, NC, FnB, TM>(fa, KA, MA, ZA, tA, dtA, X0A, fb, KB, GB == 0 ? MB : 0, GA, ZB, tB, dtB, X0B, progress, pbase);
        __syncthreads();
    } else if ((int)blockIdx.x < GA + GB) {
        // B's rollout on workgroups of its own, mixed_wpb() waves each: for SMALL batches, where A's rollout is the launch's critical path and
        // a second chain on its SIMDs would stretch it (config 5's 8-way share, 8 + 8 groups: 135 us this way, 151 us on A's idle waves; the whole config
        // the other way round: 261 against 255 us, profiles/r06/c5_mixed_attempts.txt)
        const long long ngB = (MB + TILE - 1) / TILE;
        constexpr int WPB = mixed_wpb<TM>();
        const long long g = (long long)((int)blockIdx.x - GA) * WPB + wave;
        if (wave < WPB && g < ngB) {
            const int lane = threadIdx.x & 63;
            long long j = g * TILE + lane;
            j = j < MB ? j : MB - 1;
            mixed_small_rollout<FnB, T>(fb, j, KB, TM ? 1LL : KB, TM ? MB : 1LL, ZB, tB, dtB, X0B);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) {
                unsigned* cnt = reinterpret_cast<unsigned*>(bflag + 1);
                if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ngB - 1)) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(bflag, pbase | 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }
    // ---- A's Jacobian tiles: ONE call site of the tile loop, the walk covers both parts (fused.hpp) ----
    if constexpr (TM) {
        const RolloutTilesTM<T> rt{KA, MA, GA, ZA, tA, dtA, JA, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, (unsigned)(GA + GB), split, KA * GA)};
        jac_split_loop<FnA, T, Plan, NC, AdUnrolledBody<FnA, T, Plan>>(fa, rt);
    } else {                                                    // a vector of trajectories (the reference's layout): gathered tiles of 4 trajectories x 16 knots (fused.hpp)
        const int Q = (int)((MA + GATHER_NRUN - 1) / GATHER_NRUN);
        const RolloutTilesTJ<T> rt{KA, MA, Q, ZA, tA, dtA, JA, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, (unsigned)(GA + GB), split, ((KA + GATHER_RUN - 1) / GATHER_RUN) * Q)};
        jac_split_loop<FnA, T, Plan, NC, AdUnrolledBody<FnA, T, Plan>>(fa, rt);
    }
    // ---- B's Jacobians: one 64-knot tile of the flat stack per wave and trip ----
    __syncthreads();                                            // every wave of the workgroup is out of the tile loop: its LDS is free
    {
        const unsigned long long need = pbase | 1ULL;
        while ((unsigned long long)uniform64((long long)__hip_atomic_load(bflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) __builtin_amdgcn_s_sleep(32);
        asm volatile("" ::: "memory");
        using LB = SmallLds<FnB, T>;
        constexpr int NW = FUSED_THREADS / TILE;
        const SingleBatch<T> sb{KB * MB, ZB, tB, dtB, JB, nullptr};
        jac_ad_small_walk<FnB, T, SingleBatch<T>, true>(fb, sb, (long long)blockIdx.x * NW + wave, (long long)gridDim.x * NW, smem + (size_t)wave * LB::WAVE_BYTES);
    }
}

}  // namespace rd
