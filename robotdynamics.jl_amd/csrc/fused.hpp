// fused.hpp -- rollout! and the Jacobians along the rollout in ONE launch (src/trajectories.jl:436-441 followed by
// src/functionbase.jl:242-257 at every simulated knot point: the forward pass + linearisation of a shooting method).
//
// A rollout is a latency chain that occupies ceil(M / 64) workgroups (one per 64 trajectories: cascade.hpp) for K - 1 dependent steps while the rest
// of the chip idles, and the Jacobians of the simulated knot points are a bandwidth-bound stream that needs the whole chip.  Launched one after the
// other they add up (C5's share of one GPU: 117 + 26 us; 4 096 x 101 fp64: 62 + 156 us).  Here they overlap inside one grid:
//
//   workgroups 0 .. G-1      the cascade rollout of one 64-trajectory group each (compute wavefronts per state block + the IO wavefront); the IO
//                            wavefront stores every finished chunk of 4 steps as agent-scope (write-through) atomics and then announces it in the
//                            group's 64-bit progress word, (call epoch << 32) | knots available;
//   workgroups G ..          the flagship's split-role Jacobian tile loop (kernels.hpp jac_split_loop) over tiles enumerated in the order in
//                            which they become available -- with time-major data tile (k, g) is knot k of group g's 64 trajectories, one
//                            contiguous slab -- whose compute waves poll the producer's progress word before they request a tile's inputs.
//
// Producers never wait for consumers and lower workgroup indices are dispatched first, so ONE such launch cannot deadlock however many workgroups
// are resident; two of them in flight on one device could (each one's consumers holding the compute units the other's producers are queued for),
// which is why the library keeps at most one in flight per device (api.hip fused_launch).  The Jacobian store stream then runs UNDER the rollout chain, a few steps behind it, and what is left after the last step is the tail
// of one chunk.  Results are bit-identical to rdhip_rollout followed by rdhip_jacobian: the same rollout waves, the same column code.
#pragma once
#include "cascade.hpp"

namespace rd {

// tiles of the knot points of M trajectories x K knots in TIME-MAJOR order (knot k of trajectory j at flat index k M + j): item i = (knot i / G,
// group i % G), G = ceil(M / 64) -- the order in which the rollout workgroups produce them
template <class T> struct RolloutTilesTM {
    static constexpr bool RING = false, WAITS = true;
    long long K, M; int G;
    const T* Z; const double* t; const double* dt; T* J;
    const unsigned long long* progress; unsigned long long pbase;
    RD_DEV long long ntiles() const { return K * G; }
    RD_DEV TileRef<T> at(long long item) const {
        const long long k = item / G; const long long g = item - k * G;
        const long long k0 = k * M + g * TILE, end = k * M + ((g + 1) * TILE < M ? (g + 1) * TILE : M);
        return TileRef<T>{Z, t, dt, J, nullptr, end, k0};       // (nk = end - k0; slabs are addressed by the flat knot index k0)
    }
    // every wave that is about to request the tile's inputs polls the word itself (a scalar-uniform load; words of earlier calls carry a smaller
    // epoch and compare below).  The producer's stores were written through before the word was.  On this side the tile's STATES are then read
    // with agent-scope loads that bypass the non-coherent caches (jac_split_loop: descriptors with WAITS): when tile slabs are not multiples of
    // 128 bytes (M not a multiple of 64; K (n+m) sizeof(T) not a multiple of 128 for trajectory-major stacks) this CU may already hold the line
    // that two neighbouring tiles share, read for the earlier tile BEFORE the later one's part was produced.  (An agent-scope acquire FENCE
    // here instead -- buffer_inv sc1 per tile and wave -- invalidates far more than that line: measured 575 instead of 294 us for 4 096 x 256
    // fp32 knots, the rollout workgroups themselves twice as slow.)  The compiler barrier keeps the tile's loads behind the poll; the hardware
    // issues a wave's loads in order and the poll's value has arrived before the branch on it is taken.
    RD_DEV void wait(long long item) const {
        const long long k = item / G; const long long g = item - k * G;
        const unsigned long long need = pbase | (unsigned long long)(k + 1);
        while (__hip_atomic_load(progress + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(4);
        asm volatile("" ::: "memory");
    }
};

// the same for a TRAJECTORY-MAJOR stack (knot k of trajectory j at flat index j K + k -- how the reference holds a batch of trajectories): a tile
// is 4 trajectories x 16 consecutive knots (GATHER tiles of the split-role loop: GATHER_NRUN runs of GATHER_RUN slabs -- long runs: with 4-knot
// runs the chip wrote 3 072 scattered 8 KB pieces at a time and the one-launch call was no faster than two launches), item i = (chunk i / Q of GATHER_RUN knots, trajectory
// group i % Q of GATHER_NRUN trajectories, Q = ceil(M / GATHER_NRUN)), again in the order of production: group g's chunk c is complete when its progress word
// reaches knot GATHER_RUN (c + 1) - 1
template <class T> struct RolloutTilesTJ {
    static constexpr bool RING = false, WAITS = true, GATHER = true;
    long long K, M; int Q;
    const T* Z; const double* t; const double* dt; T* J;
    const unsigned long long* progress; unsigned long long pbase;
    RD_DEV long long ntiles() const { return ((K + GATHER_RUN - 1) / GATHER_RUN) * Q; }
    RD_DEV TileRef<T> at(long long item) const {
        const long long c = item / Q, q4 = item - c * Q, j0 = q4 * GATHER_NRUN, k = c * GATHER_RUN;
        TileRef<T> r{Z, t, dt, J, nullptr, 0, j0 * K + k};
        r.run_stride = K;
        r.run_knots = (int)((K - k) < GATHER_RUN ? (K - k) : GATHER_RUN);
        r.nruns = (int)((M - j0) < GATHER_NRUN ? (M - j0) : GATHER_NRUN);
        return r;
    }
    RD_DEV void wait(long long item) const {
        const long long c = item / Q, q4 = item - c * Q, g = q4 / (TILE / GATHER_NRUN);
        const long long last = (c + 1) * GATHER_RUN < K ? (c + 1) * GATHER_RUN : K;            // knots 0 .. last - 1 are needed
        const unsigned long long need = pbase | (unsigned long long)last;
        while (__hip_atomic_load(progress + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(4);
        asm volatile("" ::: "memory");
    }
};

constexpr int FUSED_THREADS = 8 * TILE;
// register budget of the fp32 fused kernel, in waves per SIMD (tuning switch).  Its two roles need 129 (rollout) and 146 (Jacobian tile loop) registers
// on their own; left alone (1: one workgroup per CU, up to 256) the allocator takes 218 for the combined kernel; capped at 168 (3) it spills 180 bytes
// per lane and the whole-C5 satellite half runs 3 % slower (264 -> 272 us, tune/tune_c5.hip) -- the other model's kernels of a mixed-model call find
// room beside a fused workgroup either way (the cartpole's kernels take 58-72 registers)
#ifndef RD_FUSED_F32_MINW
#define RD_FUSED_F32_MINW 1
#endif
template <class Fn, class T, class Cas, int R, class Plan, int NC> struct FusedLds {
    static constexpr size_t ROLL = CascadeLds<Fn, Cas, R>::BYTES, JAC = SplitLds<Fn, T, Plan, NC>::BYTES + SPLIT_GATHER_EXTRA;
    static constexpr size_t BYTES = ROLL > JAC ? ROLL : JAC;
    static_assert((Cas::NB + 1) * TILE <= FUSED_THREADS && (Plan::NG + NC) * TILE == FUSED_THREADS, "both roles live in one 512-thread workgroup");
};
// (sj, sk): strides of the rollout's data, (K, 1) = the knots of a trajectory contiguous, (1, M) = time-major.  The SAME kernel serves rdhip_rollout
// for these models (grid = G, no Jacobian workgroups, any strides): "rollout, then Jacobians" in two calls and in one are the same machine code
// and give the same bits.  TM: the Jacobian workgroups' tiles are time-major slabs (RolloutTilesTM) or gathered from a trajectory-major stack (RolloutTilesTJ).
// `split` (0 = none): the Jacobian workgroups first share the tiles [0, split) -- what the rollout produces while its workgroups are busy -- and then
// EVERY workgroup of the grid, the former rollout workgroups included, shares [split, ntiles).  A big Monte-Carlo batch (BASELINE config 5 on one
// GPU: 64 rollout groups on a quarter of the chip) produces tiles twice as fast as the other three quarters consume them; without the second
// phase the quarter that rolled out idles for the second half of the launch (measured: 295 us for 4 096 x 256 fp32 satellite knots, the rollout
// workgroups done at 120-135 us).  The host picks `split` from the ratio of the two rates (dispatch.hpp).
template <class Fn, class T, class Cas, int R, class Plan, int NC, bool TM = true>
__global__ void __launch_bounds__(FUSED_THREADS, (sizeof(T) == 4 ? RD_FUSED_F32_MINW : 1)) rollout_jac_fused_kernel(Fn fn, long long K, long long M, int G, long long sj, long long sk, T* __restrict__ Z,
                                                                            const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0,
                                                                            T* __restrict__ J, unsigned long long* __restrict__ progress, unsigned long long pbase, long long split) {
    if ((int)blockIdx.x < G) {
        // ---- rollout of trajectories 64 blockIdx.x ..: the waves of rollout_cascade_kernel, publishing (progress == nullptr: nobody listens) ----
        using L = CascadeLds<Fn, Cas, R>;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        if (threadIdx.x < 16) reinterpret_cast<int*>(smem + L::FLAG_OFF)[threadIdx.x] = 0;
        __syncthreads();
        const bool whole = ((long long)blockIdx.x + 1) * TILE <= M;
        const int nfull = (whole && K >= 2) ? (int)((K - 1) / L::S) : 0;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        unsigned long long* pword = progress ? progress + blockIdx.x : nullptr;
#if defined(RD_FUSED_STAMP)   // tuning only (tune/tune_c5.hip): start / end of every workgroup's role, 100 MHz wall clock, in the unused upper progress words
        if (threadIdx.x == 0 && progress) progress[2048 + blockIdx.x] = wall_clock64();
#endif
        if (wave <= Cas::NB) CascadeWaves<Fn, T, Cas, R, 0, true>::run(wave, fn, K, M, sj, sk, Z, tt, dtt, X0, nfull, pword, pbase);
        __syncthreads();                                   // every wave has seen its own stores acknowledged
        if (threadIdx.x == 0) pub_progress(pword, pbase, K);
#if defined(RD_FUSED_STAMP)
        if (threadIdx.x == 0 && progress) progress[1024 + blockIdx.x] = wall_clock64();
#endif
        if (split <= 0 || !J) return;
        // second phase: this workgroup joins the Jacobian workgroups on the tiles behind `split` (every trajectory is complete by the time ANY rollout
        // workgroup gets here only for its own group: the tiles' progress words are still polled)
        __syncthreads();
    }
#if defined(RD_FUSED_STAMP)
    if (threadIdx.x == 0 && progress) progress[2048 + blockIdx.x] = wall_clock64();
#endif
    const bool producer = (int)blockIdx.x < G;
    // ONE call site for the tile loop (its unrolled column code is ~30 KB of instructions: a second inlined copy would not fit the instruction cache
    // next to the first -- measured: the whole launch 2x slower), walked once or twice
    const int first_phase = (split > 0 && producer) ? 1 : 0, last_phase = split > 0 ? 1 : 0;
#pragma clang loop unroll(disable)
    for (int phase = first_phase; phase <= last_phase; ++phase) {
        // phase 0: the Jacobian workgroups on [0, split) (or on everything); phase 1: every workgroup on [split, ntiles), the former rollout
        // workgroups with the lowest indices (they arrive first)
        const unsigned wg = phase == 0 ? blockIdx.x - G : blockIdx.x, nwg = phase == 0 ? gridDim.x - G : gridDim.x;
        const long long tb = phase == 0 ? 0 : split, te = (phase == 0 && split > 0) ? split : -1;
        if (phase > first_phase) __syncthreads();          // (the loop's waves leave it at different times: the LDS tile is reused)
        if constexpr (TM) {
            const RolloutTilesTM<T> rt{K, M, G, Z, tt, dtt, J, progress, pbase};
            jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, rt, wg, nwg, tb, te);
        } else {
            const RolloutTilesTJ<T> rt{K, M, (int)((M + GATHER_NRUN - 1) / GATHER_NRUN), Z, tt, dtt, J, progress, pbase};
            jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, rt, wg, nwg, tb, te);
        }
    }
#if defined(RD_FUSED_STAMP)
    if (threadIdx.x == 0 && progress) progress[(producer ? 3072 : 1024) + blockIdx.x] = wall_clock64();
#endif
}

}  // namespace rd
