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
// of one chunk.  The simulated states are bit-identical to rdhip_rollout's (the same rollout waves, the same machine code: rdhip_rollout runs THIS kernel
// with grid = G); the Jacobians agree with rdhip_jacobian's to rounding (the same column code, compiled a second time inside this kernel: contraction into
// fused multiply-adds may differ) -- tests hold them to 1e-14 (fp64) / 2e-6 (fp32).
#pragma once
#include "cascade.hpp"

namespace rd {

// A workgroup's WALK through the items of a fused launch (items = tiles in the order of production).  The nj Jacobian workgroups share the items
// [0, split) -- what the rollout produces while its workgroups are busy -- and then EVERY workgroup of the grid, the former rollout workgroups
// included, shares [split, total): a big Monte-Carlo batch (BASELINE config 5 on one GPU: 64 rollout groups) produces tiles faster than the rest
// of the chip consumes them, and without the second part the compute units that rolled out idle for the second half of the launch (measured:
// 295 us for 4 096 x 256 fp32 satellite knots, the rollout workgroups done at 120-135 us).  split = total: no second part.  The tile loop sees
// neither part: it walks the VIRTUAL indices w, w + NW, w + 2 NW, ... < vend() (w = the workgroup's index in the grid of NW), and the descriptor
// turns virtual index w + v NW into this workgroup's v-th item.  (Rounds 3-4 ran the tile loop once per part inside a two-trip loop; the values
// the compiler hoisted out of that loop cost the 384-thread fp32 kernel 46 spilled registers at its 168-register budget.)
struct FusedSeen { unsigned long long word = 0; long long g = -1; };     // (see fused_poll)
// The progress word of group g is word g * FUSED_PROGRESS_STRIDE of the context's array: every word in a 256-byte block of its own, so that the waves
// polling for different groups (and each group's producer, whose announcement is a store to the same line) do not meet in one memory channel.
constexpr int FUSED_PROGRESS_STRIDE = 32;
// Wait until group g's word says `need` (same epoch in the upper half, knots available in the lower).
//  * `seen`: the word this wave read last and its group.  A workgroup's consecutive items mostly belong to ONE group (the grid is a multiple of G
//    workgroups more often than not) and a big batch's rollout runs far ahead of its Jacobians, so the word read for one tile usually covers the next
//    ones as well: no poll then.  A poll is a dependent memory round trip IN FRONT of the next tile's input requests, and under a saturated store
//    stream that round trip plus the requests' own outlasts a tile's arithmetic: BASELINE config 5's satellite half 260 -> 232 us with the cache
//    (tune/tune_c5.hip, same box).
//  * back-off in proportion to what is missing: a small batch's Jacobian workgroups keep up with the rollout and spend most of their time waiting.
//    Polling every 0.1 us, a thousand waves on sixteen words, they slowed the PRODUCERS down -- the IO wave's announcement goes to the line everybody
//    polls, and its next wait for inputs (one in-order counter) sits behind it: rollout workgroups of 1 024 x 256 fp32 trajectories ended at 128 ..
//    157 us inside the fused launch, 118 us alone.  A knot takes the rollout ~0.46 us: sleep ~0.2 us per missing knot, at most 3.5 us.
RD_DEV void fused_poll(const unsigned long long* word, unsigned long long need, long long g, FusedSeen& seen) {
    if (seen.g == g && seen.word >= need) return;
    unsigned long long p;
    while ((p = (unsigned long long)uniform64((long long)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) < need) {
        const unsigned long long miss = need - p;                     // (a word of an earlier call: huge)
        const int n = miss > 16 ? 16 : (int)miss;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
    seen.g = g; seen.word = p;
    // the tile's loads stay behind the poll (compiler); the hardware issues a wave's loads in order and the poll's value has arrived before the branch on
    // it is taken.  The tile's STATES are then read with agent-scope loads that bypass the non-coherent caches (jac_split_loop: descriptors with WAITS).
    // (Only on this path: a tile that `seen` already covers lies behind the barrier of the poll that read `seen`.  In front of EVERY tile the barrier cost
    // the consumer role 250 instead of 234 us at 4 096 x 256 fp32 knots on 192 workgroups: it pins the next tile's input requests in the instruction stream.)
    asm volatile("" ::: "memory");
}
struct FusedWalk {
    unsigned w, NW, G, nj;       // this workgroup, workgroups in the grid, rollout workgroups (indices 0 .. G - 1), Jacobian workgroups (NW - G)
    unsigned split, total;
    unsigned n0, n;              // items this workgroup takes from [0, split) / in all
    static RD_DEV FusedWalk make(unsigned w, unsigned NW, unsigned G, long long split_, long long total_) {
        FusedWalk k;
        k.w = w; k.NW = NW; k.G = G; k.nj = NW - G; k.total = (unsigned)total_;
        k.split = (split_ > 0 && split_ < total_) ? (unsigned)split_ : k.total;
        const unsigned c = w - G;                                                      // (Jacobian workgroups only)
        k.n0 = (w >= G && k.split > c) ? (k.split - c + k.nj - 1) / k.nj : 0u;
        const unsigned first1 = k.split + w;
        k.n = k.n0 + (first1 < k.total ? (k.total - first1 + NW - 1) / NW : 0u);
        return k;
    }
    RD_DEV long long vend() const { return (long long)w + (long long)n * NW; }
    RD_DEV unsigned item(long long V) const {
        const unsigned v = (unsigned)V / NW;
        return v < n0 ? (w - G) + v * nj : split + (v - n0) * NW + w;
    }
};

// tiles of the knot points of M trajectories x K knots in TIME-MAJOR order (knot k of trajectory j at flat index k M + j): item i = (knot i / G,
// group i % G), G = ceil(M / 64) -- the order in which the rollout workgroups produce them
template <class T> struct RolloutTilesTM {
    static constexpr bool RING = false, WAITS = true;
    using Seen = FusedSeen;
    long long K, M; int G;
    const T* Z; const double* t; const double* dt; T* J;
    const unsigned long long* progress; unsigned long long pbase;
    FusedWalk walk;
    RD_DEV long long ntiles() const { return walk.vend(); }
    RD_DEV TileRef<T> at(long long V) const {
        const unsigned item = walk.item(V);
        const long long k = item / (unsigned)G; const long long g = item - k * G;
        const long long k0 = k * M + g * TILE, end = k * M + ((g + 1) * TILE < M ? (g + 1) * TILE : M);
        TileRef<T> r{Z, t, dt, J, nullptr, end, k0};            // (nk = end - k0; slabs are addressed by the flat knot index k0)
        r.wgroup = g; r.wneed = pbase | (unsigned long long)(k + 1);
        return r;
    }
    // every wave that is about to request the tile's inputs checks the producer's word itself (fused_poll; words of earlier calls carry a smaller epoch
    // and compare below).  The producer's stores were written through before the word was.  On this side the tile's STATES are then read with
    // agent-scope loads that bypass the non-coherent caches (jac_split_loop: descriptors with WAITS): when tile slabs are not multiples of 128 bytes
    // (M not a multiple of 64; K (n+m) sizeof(T) not a multiple of 128 for trajectory-major stacks) this CU may already hold the line that two
    // neighbouring tiles share, read for the earlier tile BEFORE the later one's part was produced.  (An agent-scope acquire FENCE here instead --
    // buffer_inv sc1 per tile and wave -- invalidates far more than that line: measured 575 instead of 294 us for 4 096 x 256 fp32 knots, the
    // rollout workgroups themselves twice as slow.)
    RD_DEV void wait(const TileRef<T>& r, FusedSeen& seen) const { fused_poll(progress + r.wgroup * FUSED_PROGRESS_STRIDE, r.wneed, r.wgroup, seen); }
};

// the same for a TRAJECTORY-MAJOR stack (knot k of trajectory j at flat index j K + k -- how the reference holds a batch of trajectories): a tile
// is 4 trajectories x 16 consecutive knots (GATHER tiles of the split-role loop: GATHER_NRUN runs of GATHER_RUN slabs -- long runs: with 4-knot
// runs the chip wrote 3 072 scattered 8 KB pieces at a time and the one-launch call was no faster than two launches), item i = (chunk i / Q of GATHER_RUN knots, trajectory
// group i % Q of GATHER_NRUN trajectories, Q = ceil(M / GATHER_NRUN)), again in the order of production: group g's chunk c is complete when its progress word
// reaches knot GATHER_RUN (c + 1) - 1
template <class T> struct RolloutTilesTJ {
    static constexpr bool RING = false, WAITS = true, GATHER = true;
    using Seen = FusedSeen;
    long long K, M; int Q;
    const T* Z; const double* t; const double* dt; T* J;
    const unsigned long long* progress; unsigned long long pbase;
    FusedWalk walk;
    RD_DEV long long ntiles() const { return walk.vend(); }
    RD_DEV TileRef<T> at(long long V) const {
        const unsigned item = walk.item(V);
        const long long c = item / (unsigned)Q, q4 = item - c * Q, j0 = q4 * GATHER_NRUN, k = c * GATHER_RUN;
        TileRef<T> r{Z, t, dt, J, nullptr, 0, j0 * K + k};
        r.run_stride = K;
        r.run_knots = (int)((K - k) < GATHER_RUN ? (K - k) : GATHER_RUN);
        r.nruns = (int)((M - j0) < GATHER_NRUN ? (M - j0) : GATHER_NRUN);
        const long long last = (c + 1) * GATHER_RUN < K ? (c + 1) * GATHER_RUN : K;            // knots 0 .. last - 1 of the tile's group are needed
        r.wgroup = q4 / (TILE / GATHER_NRUN); r.wneed = pbase | (unsigned long long)last;
        return r;
    }
    RD_DEV void wait(const TileRef<T>& r, FusedSeen& seen) const { fused_poll(progress + r.wgroup * FUSED_PROGRESS_STRIDE, r.wneed, r.wgroup, seen); }
};

// Workgroup shape of the fused kernel: the flagship's NG compute + NC copy wavefronts, 512 threads, one workgroup per CU.  (Round 5 tried NG + 2 = 384
// threads under the 168 registers of three waves per SIMD for fp32, to get TWO workgroups per CU: the occupancy API promises two, the dispatcher never
// places the second -- tune/occ_place.hip: a six-wave workgroup above 128 registers, whatever its LDS, has the CU to itself, because its waves go to the
// SIMDs 2 + 2 + 1 + 1 and a second workgroup's would make 4 on one SIMD; four-wave workgroups do pair up.)
constexpr int FUSED_THREADS = 8 * TILE;
// register budget of the fp32 fused kernel, in waves per SIMD (Tune::FUSED_F32_MINW): the allocator takes 171 registers whether it is allowed 256 (1, 2)
// or 168 (3) -- it was 218 while the tile loop sat inside a two-trip loop over the two parts of the walk; its two roles need 130 (rollout) and 146
// (Jacobian tile loop) on their own.  2 rather than 1: the hint also steers the scheduler, and the consumer role alone measured 4-5 us of 233 faster
// with it (tune/tune_c5.hip shell_plain<1> / <2>).
// (R carries CASCADE_COMPACT: the TWO-PER-CU form of the fp32 kernel -- 128 registers, the rollout role's ring in its compact layout, 78 KB of LDS.  Round 6:
// a launch bound by its Jacobian workgroups (4 096 x 256: 64 rollout groups produce tiles twice as fast as 192 workgroups consume them) runs 64 + 448
// workgroups instead of 64 + 192 -- every CU holds two tile loops, or one next to a rollout group whose chain leaves the CU idle most of the time -- and
// needs no second part of the walk: 221 -> 199-201 us for the satellite half of BASELINE config 5, tune/tune_c5.hip -DRD_FUSED_F32_MINW=4.)
template <class T, int R> constexpr int fused_min_waves() { return sizeof(T) == 4 ? ((R & CASCADE_COMPACT) ? 4 : Tune::FUSED_F32_MINW) : 1; }
template <class Fn, class T, class Cas, int R, class Plan, int NC> struct FusedLds {
    static constexpr size_t ROLL = CascadeLds<Fn, Cas, R>::BYTES, JAC = SplitLds<Fn, T, Plan, NC>::BYTES + SPLIT_GATHER_EXTRA;
    static constexpr size_t BYTES = ROLL > JAC ? ROLL : JAC;
    static_assert((Cas::NB + 1) * TILE <= FUSED_THREADS && (Plan::NG + NC) * TILE == FUSED_THREADS, "both roles live in one 512-thread workgroup");
};
// (sj, sk): strides of the rollout's data, (K, 1) = the knots of a trajectory contiguous, (1, M) = time-major.  The SAME kernel serves rdhip_rollout
// for these models (grid = G, no Jacobian workgroups, any strides): "rollout, then Jacobians" in two calls and in one are the same machine code
// and give the same bits.  TM: the Jacobian workgroups' tiles are time-major slabs (RolloutTilesTM) or gathered from a trajectory-major stack (RolloutTilesTJ).
// `split` (0 = none): see FusedWalk; the host picks it from the ratio of the two rates (dispatch.hpp).
// the rollout role of a workgroup: the waves of rollout_cascade_kernel, publishing (progress == nullptr: nobody listens)
template <class Fn, class T, class Cas, int R, class Plan, int NC>
RD_DEV void fused_rollout_role(const Fn& fn, long long K, long long M, long long sj, long long sk, T* __restrict__ Z, const double* __restrict__ tt,
                                           const double* __restrict__ dtt, const T* __restrict__ X0, unsigned long long* __restrict__ progress, unsigned long long pbase) {
    using L = CascadeLds<Fn, Cas, R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x < 16) reinterpret_cast<int*>(smem + L::FLAG_OFF)[threadIdx.x] = 0;
    __syncthreads();
    const bool whole = ((long long)blockIdx.x + 1) * TILE <= M;
    const int nfull = (whole && K >= 2) ? (int)((K - 1) / L::S) : 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long* pword = progress ? progress + (size_t)blockIdx.x * FUSED_PROGRESS_STRIDE : nullptr;
    Tune::fused_stamp(2048 + blockIdx.x);              // (tuning, tune/tune_c5.hip: start / end of every workgroup's role, 100 MHz wall clock; empty in the library)
    if (wave <= Cas::NB) CascadeWaves<Fn, T, Cas, R, 0, true>::run(wave, fn, K, M, sj, sk, Z, tt, dtt, X0, nfull, pword, pbase);
    __syncthreads();                                   // every wave has seen its own stores acknowledged
    if (threadIdx.x == 0) pub_progress(pword, pbase, K);
    Tune::fused_stamp(1024 + blockIdx.x);
}
template <class Fn, class T, class Cas, int R, class Plan, int NC, bool TM = true>
__global__ void __launch_bounds__(FUSED_THREADS, (fused_min_waves<T, R>())) rollout_jac_fused_kernel(Fn fn, long long K, long long M, int G, long long sj, long long sk, T* __restrict__ Z,
                                                                            const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0,
                                                                            T* __restrict__ J, unsigned long long* __restrict__ progress, unsigned long long pbase, long long split) {
    if ((int)blockIdx.x < G) {
        // ---- rollout of trajectories 64 blockIdx.x .. ----
        fused_rollout_role<Fn, T, Cas, R, Plan, NC>(fn, K, M, sj, sk, Z, tt, dtt, X0, progress, pbase);
        if (split <= 0 || !J) return;
        // second phase: this workgroup joins the Jacobian workgroups on the tiles behind `split` (every trajectory is complete by the time ANY rollout
        // workgroup gets here only for its own group: the tiles' progress words are still polled)
        __syncthreads();
    }
    Tune::fused_stamp(2048 + blockIdx.x);
    const bool producer = (int)blockIdx.x < G;
    // ONE call site for the tile loop (its unrolled column code is ~30 KB of instructions: a second inlined copy would not fit the instruction cache
    // next to the first -- measured: the whole launch 2x slower), and one trip through it: the workgroup's walk covers both parts (FusedWalk)
    if constexpr (TM) {
        const RolloutTilesTM<T> rt{K, M, G, Z, tt, dtt, J, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, (unsigned)G, split, K * G)};
        jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, rt);
    } else {
        const int Q = (int)((M + GATHER_NRUN - 1) / GATHER_NRUN);
        const RolloutTilesTJ<T> rt{K, M, Q, Z, tt, dtt, J, progress, pbase, FusedWalk::make(blockIdx.x, gridDim.x, (unsigned)G, split, ((K + GATHER_RUN - 1) / GATHER_RUN) * Q)};
        jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, rt);
    }
    Tune::fused_stamp((producer ? 3072 : 1024) + blockIdx.x);
}

}  // namespace rd
