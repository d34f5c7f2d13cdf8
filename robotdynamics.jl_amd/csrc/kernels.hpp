// kernels.hpp -- the batched kernels.  Data layout in HBM (all column-major, knot point k contiguous):
//   Z  (n+m) x N      t, dt  N (fp64)      Y  n x N      J  n x (n+m) x N
// Mapping (gfx950, wave64): one LANE owns one knot point, one WAVEFRONT owns 64 knot points x a group of
// Jacobian columns, one WORKGROUP (NG waves) owns a tile of 64 knot points.  Inputs and outputs of a tile are
// contiguous slabs of global memory; they are staged through LDS so that every global access is lane-contiguous.
// Everything between load and store lives in registers.
#pragma once
#include "models.hpp"
#include "tune_default.hpp"
// the tuning policy (tune_default.hpp): the library never defines RD_TUNE; a harness under csrc/tune/ includes tune/harness_tune.hpp first
#ifndef RD_TUNE
#define RD_TUNE rd::DefaultTune
#endif

namespace rd {

using Tune = RD_TUNE;
// register budget of the fp32 ForwardAD tile kernels, in waves per SIMD: 2 = up to 256 registers (the kernels take 146 and do not spill; one
// workgroup per CU).  Rounds 1-3 asked for 4 (128 registers, two workgroups per CU by LDS): 68 bytes of scratch per lane and 30.3 instead of
// 26.9 us for 131 072 fp32 satellite knot points (tune/tune_plan.hip -DTT=float, same-box A/B)
constexpr int RD_F32_MINB = Tune::F32_MINB;
constexpr int TILE = 64;   // knot points per wave = wave width on gfx950
constexpr int ROLLOUT_PREFETCH = 4;   // rollout kernels: inputs are requested this many steps ahead

template <class T> RD_DEV T sqrt_eps();
template <> RD_DEV double sqrt_eps<double>() { return 1.4901161193847656e-08; }   // sqrt(eps(Float64))
template <> RD_DEV float sqrt_eps<float>() { return 3.4526698300124393e-04f; }     // sqrt(eps(Float32))

// dt === 0 (terminal knot point) tested on the BIT PATTERN: the build uses -ffinite-math-only, under which a
// floating-point compare against a NaN/Inf time step (the reference stores dt = Inf on a last knot point that
// carries a control, src/trajectories.jl:104) would be undefined.
RD_DEV bool is_terminal_dt(double h) { return (__double_as_longlong(h) << 1) == 0; }

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

// global -> LDS staging of `total` (<= MAXE) consecutive elements by NT threads: every thread issues ALL of its loads before it waits
// for any (a run-time `load, wait, ds_write` loop costs one memory round trip per pass -- measured as several microseconds in front of
// the first arithmetic of a kernel); out-of-range lanes re-read element 0 so that the loads stay unconditional
template <class E, int MAXE, int NT> RD_DEV void stage_in(const E* __restrict__ src, E* dst, int total, int tid) {
    constexpr int PT = (MAXE + NT - 1) / NT;
    E r[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) { const int e = i * NT + tid; r[i] = src[e < total ? e : 0]; }
#pragma unroll
    for (int i = 0; i < PT; ++i) { const int e = i * NT + tid; if (e < total) dst[e] = r[i]; }
}

// ------------------------------------------------------------------------------------------------------------
// evaluate: Y[:,k] = fun(z_k)      (dynamics / discrete_dynamics)
// ------------------------------------------------------------------------------------------------------------
template <class Fn, class T>
__global__ void __launch_bounds__(256) eval_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                   const double* __restrict__ dtt, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NB = 256;
    constexpr int VN = 16 / (int)sizeof(T);                  // elements per 16-byte unit
    typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
    __shared__ __attribute__((aligned(16))) T lds[NB * (NZ > NY ? NZ : NY)];
    const long long k0 = (long long)blockIdx.x * NB;
    const int nk = (int)((N - k0) < NB ? (N - k0) : NB);
    // tiles start at multiples of 256 knot points, i.e. at multiples of 16 bytes from the array base: 16-byte units when the base is aligned
    const bool zal = (reinterpret_cast<unsigned long long>(Z) & 15) == 0, yal = (reinterpret_cast<unsigned long long>(Y) & 15) == 0;
    {
        const T* src = Z + k0 * NZ;
        const int total = nk * NZ;
        if (zal) {
            const int nv = total / VN;
            stage_in<V, (NB * NZ + VN - 1) / VN, NB>(reinterpret_cast<const V*>(src), reinterpret_cast<V*>(lds), nv, threadIdx.x);
            for (int e = nv * VN + threadIdx.x; e < total; e += NB) lds[e] = src[e];     // < VN elements
        } else stage_in<T, NB * NZ, NB>(src, lds, total, threadIdx.x);
    }
    __syncthreads();
    const int kl = threadIdx.x < nk ? threadIdx.x : nk - 1;
    const T t = (T)tt[k0 + kl], h = (T)dtt[k0 + kl];
    T x[NX], u[NU], y[NY];
    read_knot<Fn, T>(lds, kl, is_terminal_dt(dtt[k0 + kl]), x, u);
    fn.eval(x, u, t, h, y);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NY; ++i) lds[threadIdx.x * NY + i] = y[i];
    __syncthreads();
    {
        T* dst = Y + k0 * NY;
        const int total = nk * NY, nv = yal ? total / VN : 0;
        for (int e = threadIdx.x; e < nv; e += NB) reinterpret_cast<V*>(dst)[e] = reinterpret_cast<const V*>(lds)[e];
        for (int e = nv * VN + threadIdx.x; e < total; e += NB) dst[e] = lds[e];
    }
}

// ------------------------------------------------------------------------------------------------------------
// rollout: thread j simulates trajectory j, sequential in k (src/trajectories.jl:436-441)
// Knot point k of trajectory j sits at index j*sj + k*sk of Z / t / dt: (sj, sk) = (K, 1) for consecutive trajectories, (1, M) for the
// time-major layout (all trajectories' knot k contiguous).
// ------------------------------------------------------------------------------------------------------------
template <class Fn, class T>
__global__ void __launch_bounds__(64) rollout_kernel(Fn fn, long long K, long long M, long long sj, long long sk, T* __restrict__ Z,
                                                     const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU;
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    // a latency chain: when the wave shares its SIMD with waves of another kernel (the mixed-model call runs this rollout next to the persistent
    // Jacobian workgroups of the other model's fused launch, which poll and stream), it should win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    T x[NX], u[NU], y[NX];
    const long long ip = j * sj;                              // knot 0 of this trajectory
    {
        const T* xs = X0 ? X0 + j * NX : Z + ip * NZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xs[i];
    }
    if (K < 2) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Z[ip * NZ + i] = x[i];
        return;
    }
    // The controls / times of knot k + D are requested while step k is computed, and the new state is stored after the arithmetic:
    //  * a rollout streams through memory it has never touched, so every request is an HBM miss; with a distance of ONE step the chain ran at
    //    the memory latency (0.7-1.0 us per step for every model, the cartpole as slow as the satellite), D = ROLLOUT_PREFETCH steps hide it;
    //  * the requests are older than the stores that follow them, so waiting for one (one vmcnt counter for both on gfx950) does not wait
    //    for the stores just issued, and the stores of a step drain under the following steps' arithmetic;
    //  * the initial state is written AFTER the first requests (also when it was read from Z itself) so that the loop is entered with the
    //    same "loads, then NX stores" queue its back edge has: the compiler's waitcnt at the loop header then lets the stores stay in flight.
    // The D slots live in registers (the loop is unrolled D-fold, the slot index is a constant).
    constexpr int D = ROLLOUT_PREFETCH;
    const long long last = K - 1, nsteps = K - 1;
    T un[D][NU]; double hn[D], tn[D];
    auto fetch = [&](int d, long long knot) {                  // knot <= K - 1: inside the trajectory, read unconditionally
        const long long idx = ip + knot * sk;
        hn[d] = dtt[idx]; tn[d] = tt[idx];
#pragma unroll
        for (int i = 0; i < NU; ++i) un[d][i] = Z[idx * NZ + NX + i];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, d < last ? d : last);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) Z[ip * NZ + i] = x[i];
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](long long k, int d) {                      // x_k -> x_{k+1}, inputs in slot d
        const double h = hn[d]; const T tk = (T)tn[d];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(h) ? T(0) : un[d][i];
        fetch(d, k + D < last ? k + D : last);
        __builtin_amdgcn_sched_barrier(0);                    // keep the loads above the arithmetic
        fn.eval(x, u, tk, (T)h, y);
        T* Zk = Z + (ip + (k + 1) * sk) * NZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = y[i]; Zk[i] = y[i]; }
    };
    long long k = 0;
    for (; k + D <= nsteps; k += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(k + d, d);
    }
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) if (k + d < nsteps) step(k + d, d);      // at most D - 1 steps are left
}

// ------------------------------------------------------------------------------------------------------------
// Jacobians -- the generic tile loop (jac_tile_loop: FiniteDifference, run-time-column ForwardAD / chain rule for models without
// a column plan).  One WORKGROUP (NG waves) owns a tile of 64 knot points at a time and walks tiles persistently; one LANE owns one
// knot point; the NZ = n+m Jacobian columns are split over the waves (Plan), optionally in NCHUNK column chunks processed in turn.
//
// LDS per workgroup:  out[64][RS]  one row per knot point holding the chunk's columns (and, in one chunk, f(z)),
//                                  RS odd so that the lane stride hits distinct banks;
//                     z tile 64 x NZ, t[64], dt[64].
// After a chunk is computed the whole workgroup copies it out row by row (row_copy): a row is a (ncol*NY*sizeof(T))-byte
// contiguous run of global memory.  Measured on MI355X: storing single columns from single waves (104-B runs)
// ran 2x slower than a streaming fill.  The next tile's z/t/dt are prefetched into registers before the last copy-out.
// The flagship path (ForwardAD / chain rule with compile-time columns) uses the split-role loop further down instead.
// No inter-workgroup data reuse exists, so no XCD-aware remap is needed: each tile is touched by exactly one workgroup.
// ------------------------------------------------------------------------------------------------------------

// ---- where a tile lives ---------------------------------------------------------------------------------------------
// The tile loops walk GLOBAL tile indices; a batch descriptor maps one to the arrays it belongs to.  SingleBatch: one packed trajectory
// (rdhip_jacobian).  MultiBatch: any number of independent packed trajectories -- different arrays, e.g. 1 024 trajectories of 101 knot points
// -- enumerated back to back by the same persistent loop, so that many small batches cost one bandwidth-regime launch instead of one
// launch-bound kernel each (rdhip_jacobian_multi).  Up to MULTI_MAXB batches travel in the kernel arguments (no device-side table to keep
// alive, and a captured graph node carries it); more than that sit in a device table (`ext`: one 64-byte record per batch, `ext_tile0`: the
// first global tile of every batch, compact so that the binary search stays inside the scalar cache).  Batches of equal size (the common
// case: M trajectories of K knot points each) need no search at all: batch = tile / tiles-per-batch.  Everything here is wave-uniform
// (scalar loads / SALU).
template <class T> struct TileRef {
    const T* Z; const double* t; const double* dt; T* J; T* Y;
    long long N;      // knot points of the batch the tile belongs to
    long long k0;     // first knot point of the tile inside that batch
    // descriptors with GATHER: the tile's 64 slots are GATHER_NRUN runs of GATHER_RUN consecutive knot points, run r starting at knot k0 + r run_stride
    // (4 trajectories x 16 knots of a trajectory-major stack); the first nruns runs and the first run_knots knots of every run exist
    long long run_stride; int run_knots; int nruns;
    // descriptors with WAITS: what the tile waits for -- its producer group and the progress word that makes it available (fused.hpp); at() fills them
    // in, so that the wait in front of every tile does not repeat at()'s index arithmetic: a compute wave issues ~one instruction per 5 cycles whatever
    // its kind, and ~40 scalar instructions per tile in front of the input requests measured 0.18 us of a 3.2 us iteration (tune/tune_c5.hip stamps)
    long long wgroup; unsigned long long wneed;
};
constexpr int GATHER_RUN = 16, GATHER_NRUN = TILE / GATHER_RUN;      // (measured: runs of 4 / 8 / 16 knot points -> the one-launch call takes 0.97 / 0.955 / 0.905 of the two-launch time at 4 096 x 101)
constexpr size_t SPLIT_GATHER_EXTRA = (size_t)GATHER_NRUN * 16;
template <class BT, class = void> struct GatherOf { static constexpr bool value = false; };
template <class BT> struct GatherOf<BT, decltype((void)BT::GATHER)> { static constexpr bool value = BT::GATHER; };
template <class T> struct SingleBatch {
    static constexpr bool RING = false;    // at() is arithmetic
    long long N; const T* Z; const double* t; const double* dt; T* J; T* Y;
    RD_DEV long long ntiles() const { return (N + TILE - 1) / TILE; }
    RD_DEV TileRef<T> at(long long tile) const { return TileRef<T>{Z, t, dt, J, Y, N, tile * TILE}; }
};
// the same with the launch's counter words of the dynamic tile schedule (next, done, eight per-XCD-range counters; zero between launches).  A type of its
// own, and the kernels that take it instantiations of their own (template parameter DYN): compiled into the one kernel, the dormant schedule cost the
// 65 536-knot launch -- four tiles per workgroup, static -- 0.3-0.5 us of 27 (five registers, an LDS word read per iteration and role)
template <class T> struct SingleBatchDyn : SingleBatch<T> { unsigned* sched; };
template <class BT, class = void> struct SchedOf { static constexpr bool HAS = false; static RD_DEV unsigned* get(const BT&) { return nullptr; } };
template <class BT> struct SchedOf<BT, decltype((void)&BT::sched)> { static constexpr bool HAS = true; static RD_DEV unsigned* get(const BT& bt) { return bt.sched; } };
constexpr int MULTI_MAXB = 32;
template <class T> struct MultiRec { const T* Z; const double* t; const double* dt; T* J; T* Y; long long N; long long pad[2]; };   // 64 bytes: one scalar load
template <class T> struct MultiBatch {
    // at() is two dependent memory round trips: the split-role loop resolves the tiles of a workgroup's first iterations IN PARALLEL before its
    // first barrier (one copy-wave lane per iteration) and hands them to everybody through an LDS table (jac_split_loop), so that no lookup sits
    // in front of a compute wave's input prefetch or between a copy wave's stores
    static constexpr bool RING = !Tune::MULTI_NO_RING;
    int nb;
    int uniform;                           // > 0: every batch has this many tiles (no search)
    long long total;                       // tiles of all batches (< 2^31)
    const MultiRec<T>* ext;                // device table of nb records, or nullptr: the inline table below
    const unsigned* ext_tile0;             // device: first global tile of batch b, nb + 1 entries
    int group;                             // ext: batches per entry of the coarse inline table (ceil(nb / MULTI_MAXB))
    int pad_;
    unsigned tile0[MULTI_MAXB + 1];        // inline: first global tile of batch b; ext: of batch b * group; entries past the end = total
    long long N[MULTI_MAXB];
    const T* Z[MULTI_MAXB]; const double* t[MULTI_MAXB]; const double* dt[MULTI_MAXB]; T* J[MULTI_MAXB]; T* Y[MULTI_MAXB];
    RD_DEV long long ntiles() const { return total; }
    // Two memory round trips whatever the number of batches: all entries of a level are requested at once and counted branch-free (a
    // compare-and-advance loop would be one DEPENDENT scalar load per batch passed -- microseconds on a cold scalar cache, in front of the
    // first input load of a workgroup)
    RD_DEV TileRef<T> at(long long tile) const {
        const unsigned tl = (unsigned)tile;
        int c = 0;
#pragma unroll
        for (int i = 1; i <= MULTI_MAXB; ++i) c += (tl >= tile0[i]) ? 1 : 0;
        if (!ext) return TileRef<T>{Z[c], t[c], dt[c], J[c], Y[c], N[c], (long long)(tl - tile0[c]) * TILE};
        // The device table is read through CONSTANT-address-space pointers: it does not change while the kernel runs, and only loads the
        // compiler knows to be invariant become scalar loads.  As plain global loads (the kernel stores to global memory, so nothing proves
        // them unclobbered) every field of the TileRef -- and, through the merge with the inline path, of every TileRef -- lived in VECTOR
        // registers: 14 more VGPRs across the column code, which is at the 256-register limit and spilled 92 bytes per lane.
        typedef const unsigned __attribute__((address_space(4))) * CU32;
        typedef const MultiRec<T> __attribute__((address_space(4))) * CRec;
        CU32 et0 = (CU32)(unsigned long long)ext_tile0;
        CRec er = (CRec)(unsigned long long)ext;
        int b; unsigned first;
        if (uniform > 0) { b = (int)(tl / (unsigned)uniform); first = (unsigned)b * (unsigned)uniform; }
        else {
            const int lo = c * group;
            if (group <= MULTI_MAXB) {                                 // nb <= 1 024: the group's entries in one go
                int k = 0;
#pragma unroll
                for (int i = 1; i < MULTI_MAXB; ++i) {                 // (fixed trip count: independent loads, requested together)
                    const int j = lo + i; const bool in = i < group && j < nb;
                    const unsigned v = et0[in ? j : lo];
                    k += (in && tl >= v) ? 1 : 0;
                }
                b = lo + k;
            } else {
                int l = lo, h = lo + group < nb ? lo + group : nb;     // tile0[l] <= tile < tile0[h]
                while (h - l > 1) { const int mid = (l + h) >> 1; if (tl >= et0[mid]) l = mid; else h = mid; }
                b = l;
            }
            first = et0[b];
        }
        const T* rZ = er[b].Z; const double* rt = er[b].t; const double* rdt = er[b].dt; T* rJ = er[b].J; T* rY = er[b].Y; const long long rN = er[b].N;
        return TileRef<T>{rZ, rt, rdt, rJ, rY, rN, (long long)(tl - first) * TILE};
    }
};

// wave-uniform 64-bit value held in vector registers -> scalar registers
RD_DEV long long uniform64(long long v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
// a batch table together with the launch's counter words of the dynamic tile schedule (jac_split_loop); the table stays where the kernel argument is
template <class T> struct MultiBatchSched {
    static constexpr bool RING = MultiBatch<T>::RING;
    const MultiBatch<T>& m;
    unsigned* sched;
    RD_DEV long long ntiles() const { return m.ntiles(); }
    RD_DEV TileRef<T> at(long long tile) const { return m.at(tile); }
};
template <class BT> struct LooksUp { static constexpr bool value = false; };
template <class T> struct LooksUp<MultiBatch<T>> { static constexpr bool value = true; };
template <class T> struct LooksUp<MultiBatchSched<T>> { static constexpr bool value = true; };
// bt.at(tile) for a tile number that is the same in every lane of the wavefront: a descriptor with a table counts and subtracts in vector registers
// (the compiler vectorises the branch-free search), so without this the tile's pointers and first knot point stay in VGPRs -- 14 registers across the
// column code of a kernel at the 256-register limit, and a waterfall loop around every buffer store of the copy waves, whose resource must be scalar
template <class T, class BT> RD_DEV TileRef<T> at_uniform(const BT& bt, long long tile) {
    const TileRef<T> r = bt.at(tile);
    if constexpr (!LooksUp<BT>::value) return r;
    else return TileRef<T>{(const T*)uniform64((long long)r.Z), (const double*)uniform64((long long)r.t), (const double*)uniform64((long long)r.dt),
                           (T*)uniform64((long long)r.J), (T*)uniform64((long long)r.Y), uniform64(r.N), uniform64(r.k0)};
}


// tuning (tune/tune_stamp.hip, tune/tune_c5.hip): wall-clock stamps (100 MHz) of the split-role loop's phases, [workgroup][iteration < 8][slot < 16]:
// 0 compute wave 0 enters the iteration, 4 + g compute wave g is done with its columns, 2 / 3 compute wave 0 behind barrier A / B, 8 / 9 / 10 copy
// wave 0 behind barrier A / has the tile in registers / has issued its stores, 11 / 12 helper wave starts / ends the next tile's chain, 13 / 15 (iteration
// 0) the workgroup's first / last instruction.  Tune::split_stamp is empty in the library.
#define RD_STAMP(it, slot) Tune::split_stamp((it), (slot))

// Default plan: one chunk, contiguous column ranges per wave, f(z) emitted by wave 0.
template <int NZ, int NG_> struct ContiguousPlan {
    static constexpr int NG = NG_, NCHUNK = 1;
    static constexpr int chunk_begin(int) { return 0; }
    static constexpr int chunk_end(int) { return NZ; }
    static constexpr int wave_of(int c) { int g = 0; while (!(c >= g * NZ / NG && c < (g + 1) * NZ / NG)) ++g; return g; }
    static constexpr int value_chunk = 0, value_wave = 0;
};

// Cost-balanced single-chunk plan from a column -> wave table (columns of one wave need not be contiguous: the tile is
// assembled in LDS and leaves as one linear sweep).
template <int NZ, int NG_, int VW, int... W> struct TablePlan {
    static_assert(sizeof...(W) == NZ, "one wave index per Jacobian column");
    static constexpr int NG = NG_, NCHUNK = 1;
    static constexpr int chunk_begin(int) { return 0; }
    static constexpr int chunk_end(int) { return NZ; }
    static constexpr int wave_of(int c) { constexpr int w[NZ] = {W...}; return w[c]; }
    static constexpr int value_chunk = 0, value_wave = VW;
};
// TablePlan whose first SPLIT columns form the EARLY PART of a workgroup's first tile (split-role loop, bodies that support it): every compute
// wave handles its columns in ascending order, so once all waves are past column SPLIT the leading SPLIT*NY values of every row are final and
// the copy waves start storing them while the remaining columns are still being computed -- the first tile of a launch is the only one whose
// arithmetic has no store stream to hide under.  The table should give every wave a similar amount of work on either side of SPLIT.
template <int NZ, int NG_, int VW, int SPLIT, int... W> struct TablePlanS : TablePlan<NZ, NG_, VW, W...> {
    static_assert(SPLIT > 0 && SPLIT < NZ, "the early part is a proper prefix of the columns");
    static constexpr int FIRST_SPLIT = SPLIT;
};
template <class Plan, class = void> struct FirstSplitOf { static constexpr int N = 0; };
template <class Plan> struct FirstSplitOf<Plan, decltype((void)Plan::FIRST_SPLIT)> { static constexpr int N = Plan::FIRST_SPLIT; };
// Body::HELPER: the LAST copy wave does not copy; it runs Body::helper_tile one tile ahead of the compute waves (results through an LDS double
// buffer of Body::HELPER_BYTES each, handed to the body as fr.helper): a latency chain that must precede a tile's columns (the Newton solve of
// the implicit-function-theorem kernel) then runs beside the previous tile's arithmetic instead of in front of this tile's
template <class Body, class = void> struct HelperOf { static constexpr bool value = false; static constexpr size_t BYTES = 0; };
template <class Body> struct HelperOf<Body, decltype((void)Body::HELPER)> { static constexpr bool value = Body::HELPER; static constexpr size_t BYTES = Body::HELPER_BYTES; };
// Body::COPY_PREFETCH: the next tile's inputs are requested by the COPY waves (at the top of an iteration, behind the previous tile's stores) and written to
// the LDS input tile by them between barriers A and B, instead of riding in the compute waves' registers across the whole tile.  For bodies whose compute
// role is out of registers: there the compiler SPILLED the prefetched values -- and to spill a value it must have arrived, so the compute waves sat in
// s_waitcnt vmcnt(0) for a load round trip under a saturated store stream at the top of every tile (body-frame quadrotor: parked 57 % of their cycles,
// profiles/r06/valu_roofline.md).  Body::COPY_PACE: s_sleep units behind every tile store of a copy wave -- the compute waves' remaining scratch accesses
// queue behind the copy waves' stores on the CU's one memory path, and a burst of 32 stores per lane holds them up longer than a paced one.
template <class Body, class = void> struct CopyPrefetchOf { static constexpr bool value = false; static constexpr int PACE = 0; };
template <class Body> struct CopyPrefetchOf<Body, decltype((void)Body::COPY_PREFETCH)> { static constexpr bool value = Body::COPY_PREFETCH; static constexpr int PACE = Body::COPY_PACE; };
// Body::HELPER_STAGES (bodies with HELPER): the helper wave, which reads the next tile's knot points for its own chain anyway, also puts them into the LDS
// input tile between barriers A and B (lane = knot point: its row of NX + NU values, t, dt) -- the compute waves request nothing.  Same reason as
// CopyPrefetchOf: the implicit-function-theorem kernel's compute role spilled its seven prefetch registers and waited for them at the top of every tile.
template <class Body, class = void> struct HelperStagesOf { static constexpr bool value = false; };
template <class Body> struct HelperStagesOf<Body, decltype((void)Body::HELPER_STAGES)> { static constexpr bool value = Body::HELPER_STAGES; };
template <class Body, class = void> struct EpilogueOf { static constexpr bool value = false; };
template <class Body> struct EpilogueOf<Body, decltype((void)Body::EPILOGUE)> { static constexpr bool value = Body::EPILOGUE; };
template <class Body, class = void> struct EarlyOf { static constexpr bool value = false; };
template <class Body> struct EarlyOf<Body, decltype((void)Body::EARLY)> { static constexpr bool value = Body::EARLY; };

template <class Fn, class T, class Plan> struct JacLds {
    static constexpr int NZ = Fn::NX + Fn::NU, NY = Fn::NY;
    static constexpr int run(int ch) { return (Plan::chunk_end(ch) - Plan::chunk_begin(ch)) * NY + (ch == Plan::value_chunk ? NY : 0); }
    static constexpr int max_run() { int m = 0; for (int ch = 0; ch < Plan::NCHUNK; ++ch) m = run(ch) > m ? run(ch) : m; return m; }
    static constexpr int RS = max_run() | 1;     // odd row stride -> conflict-free lane-strided LDS writes
    static constexpr size_t OUT_ELEMS = (size_t)TILE * RS, Z_ELEMS = (size_t)TILE * NZ;
    static constexpr size_t BYTES = (OUT_ELEMS + Z_ELEMS) * sizeof(T) + 2 * TILE * sizeof(double);
};

// a function object may ask for NAUX extra lane-private inputs per knot point (read from fn.aux_ptr()[k * NAUX + i] one tile ahead by the
// split-role loop and handed to the body in fr.aux): the implicit-function-theorem kernel reads the Newton solution this way (implicit.hpp)
template <class Fn, class = void> struct AuxOf { static constexpr int N = 0; };
template <class Fn> struct AuxOf<Fn, decltype((void)Fn::NAUX)> { static constexpr int N = Fn::NAUX; };

template <class Fn, class T> struct JacFrame {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NVAL = Fn::NVAL, NA = AuxOf<Fn>::N;
    const T* zs;     // the staged input tile (bodies with Fn::NEXT read the following knot point from it)
    int kl;          // this lane's knot point inside the tile
    long long tile;  // global tile index (split-role loop)
    bool first;      // split-role loop: this is the workgroup's first tile (bodies with EARLY pass a workgroup barrier at column Plan::FIRST_SPLIT)
    const unsigned char* helper;   // split-role loop, bodies with HELPER: what the helper wave left for this tile (LDS)
    T aux[NA > 0 ? NA : 1];
    T x[NX], u[NU];
    T t, h;
    bool term;
    int g;
    T* row;          // this lane's row of the LDS output tile
    T* yrow;         // this lane's row of a separate LDS f(z) tile, or nullptr: f(z) goes behind the chunk's columns
    bool want_y;
    bool epi = false;   // split-role loop, bodies with EPILOGUE: set (wave-uniformly, by compute wave 0) when some tile of the workgroup needs the epilogue
    // position inside the row: columns of the current chunk first
    RD_DEV void put_at(int pos, const T (&col)[NY]) {
#pragma unroll
        for (int i = 0; i < NY; ++i) row[pos * NY + i] = col[i];
    }
    RD_DEV void put_value(int pos_after_columns, const T (&y)[NVAL]) {
        if (yrow) {
#pragma unroll
            for (int i = 0; i < NVAL; ++i) yrow[i] = y[i];
        } else {
#pragma unroll
            for (int i = 0; i < NVAL; ++i) row[pos_after_columns * NY + i] = y[i];
        }
    }
};

// 16-byte lane-contiguous copy of one row (LDS -> global) by one wave; rows are only sizeof(T)-aligned on both sides
template <class T> struct Vec16;
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); static constexpr int V = 2; };
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); static constexpr int V = 4; };
template <class T> RD_DEV void row_copy(const T* __restrict__ src, T* __restrict__ dst, int len, int lane) {
    using V = typename Vec16<T>::type;
    constexpr int NV = Vec16<T>::V;
    const int nu = len / NV;
    for (int u = lane; u < nu; u += TILE) reinterpret_cast<V*>(dst)[u] = reinterpret_cast<const V*>(src)[u];
    const int e = nu * NV + lane;
    if (e < len) dst[e] = src[e];
}

// Body::template run<CH>(fn, fr) computes the columns of chunk CH that belong to wave fr.g and `put`s them.
//
// VMEM ordering rule that shapes the loop (gfx950 has ONE counter, vmcnt, for loads and stores, and they may retire
// out of order relative to each other, so waiting for any load also waits for every older store): the only global
// loads of an iteration -- the prefetch of the NEXT tile's z/t/dt -- are issued at the top of the iteration and
// consumed right before the copy-out; the copy-out stores are the last VMEM operations of the iteration and nobody
// waits for them until the next iteration's prefetch is consumed, i.e. they drain behind a whole compute phase.
template <class Fn, class T, class Plan, class Body, class BT>
RD_DEV void jac_tile_loop(const Fn& fn, const BT& bt) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG, NT = NG * TILE;
    using L = JacLds<Fn, T, Plan>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* out = reinterpret_cast<T*>(smem);
    T* zs = out + L::OUT_ELEMS;                                          // z tile
    double* ts = reinterpret_cast<double*>(zs + L::Z_ELEMS);             // t[64], dt[64]
    const int lane = threadIdx.x & 63;
    const long long ntiles = bt.ntiles();
    constexpr int ZPT = (TILE * NZ + NT - 1) / NT;   // z elements each thread stages per tile

    JacFrame<Fn, T> fr;
    fr.g = threadIdx.x >> 6;
    fr.row = out + lane * L::RS;
    fr.yrow = nullptr;
    fr.first = false;                                // (no early part in this loop: every wave computes and copies)

    long long tile = blockIdx.x;
    if (tile >= ntiles) return;
    TileRef<T> cur = at_uniform<T>(bt, tile);
    {   // first tile: stage z, t, dt
        const long long k0 = cur.k0;
        const int nk = (int)((cur.N - k0) < TILE ? (cur.N - k0) : TILE);
        const double* tp = (threadIdx.x < TILE) ? cur.t : cur.dt;
        const int idx = threadIdx.x & (TILE - 1);
        const double t0 = tp[(threadIdx.x < 2 * TILE && idx < nk) ? k0 + idx : 0];
        stage_in<T, TILE * NZ, NT>(cur.Z + k0 * NZ, zs, nk * NZ, threadIdx.x);
        if (threadIdx.x < 2 * TILE) ts[threadIdx.x] = t0;
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const long long k0 = cur.k0;
        const int nk = (int)((cur.N - k0) < TILE ? (cur.N - k0) : TILE);
        fr.want_y = (cur.Y != nullptr);
        // ---- issue the prefetch of the next tile's inputs (registers; consumed after the compute phase)
        const long long nxt = tile + gridDim.x;
        const bool more = nxt < ntiles;
        const TileRef<T> nx = at_uniform<T>(bt, more ? nxt : tile);
        const long long k1 = nx.k0;
        const T* __restrict__ Zn = nx.Z;
        const int nk1 = more ? (int)((nx.N - k1) < TILE ? (nx.N - k1) : TILE) : 0;
        // The loads are UNCONDITIONAL (out-of-range lanes re-read element 0): a load inside an exec-masked branch leaves a
        // "maybe pending" load at the loop back-edge, and the compiler then drains vmcnt(0) -- i.e. every store of the
        // previous copy-out -- at the top of the loop.
        T zpre[ZPT]; double tpre;
#pragma unroll
        for (int r = 0; r < ZPT; ++r) { const int e = r * NT + threadIdx.x; zpre[r] = Zn[(e < nk1 * NZ) ? k1 * NZ + e : 0]; }
        {
            const double* tp = (threadIdx.x < TILE) ? nx.t : nx.dt;
            const int idx = threadIdx.x & (TILE - 1);
            tpre = tp[(threadIdx.x < 2 * TILE && idx < nk1) ? k1 + idx : 0];
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the loads up here
        // ---- this tile
        const int kl = lane < nk ? lane : nk - 1;        // lanes past a ragged end recompute the last knot point (never copied out)
        const double hd = ts[TILE + kl];
        fr.t = (T)ts[kl]; fr.h = (T)hd; fr.term = is_terminal_dt(hd);
        read_knot<Fn, T>(zs, kl, fr.term, fr.x, fr.u);
        T* Jt = cur.J + k0 * (NY * NZ);
        T* Yt = cur.Y ? cur.Y + k0 * NY : nullptr;
#pragma unroll
        for (int ch = 0; ch < Plan::NCHUNK; ++ch) {
            switch (ch) {   // compile-time chunk index for the body
                case 0: Body::template run<0>(fn, fr); break;
                case 1: if constexpr (Plan::NCHUNK > 1) Body::template run<1>(fn, fr); break;
                case 2: if constexpr (Plan::NCHUNK > 2) Body::template run<2>(fn, fr); break;
                default: if constexpr (Plan::NCHUNK > 3) Body::template run<3>(fn, fr); break;
            }
            __syncthreads();                              // chunk complete in LDS
            if (ch == 0) {                                // consume the prefetch BEFORE any store of this iteration is issued
                __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the prefetch and the previous tile's stores (which had a whole compute phase)
                // (every wave finished reading zs/ts before the barrier above, so the single buffer can be refilled)
#pragma unroll
                for (int r = 0; r < ZPT; ++r) { const int e = r * NT + threadIdx.x; if (e < TILE * NZ) zs[e] = zpre[r]; }
                if (threadIdx.x < 2 * TILE) ts[threadIdx.x] = tpre;
                __builtin_amdgcn_sched_barrier(0);
            }
            // copy-out: wave g takes rows g, g+NG, ...; a row is one (two with f(z)) contiguous run of global memory
            const int c0 = Plan::chunk_begin(ch), jr = (Plan::chunk_end(ch) - c0) * NY;
            const bool has_y = (ch == Plan::value_chunk) && Yt;
            for (int kn = fr.g; kn < nk; kn += NG) {
                const T* src = out + kn * L::RS;
                row_copy<T>(src, Jt + kn * (NY * NZ) + c0 * NY, jr, lane);
                if (has_y) row_copy<T>(src + jr, Yt + kn * NY, NY, lane);
            }
            __syncthreads();                              // LDS output tile free again; next inputs visible
        }
        cur = nx;
    }
}

// ---- ForwardAD, compile-time columns: one tangent direction per pass, the seed is a constant -----------------------
// emits f(z) together with the first column it handles when this (chunk, wave) is the plan's value slot
template <class Fn, class T, class Plan, int CH, int G, int C, bool VALUE_PENDING> struct AdCols {
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY;
        constexpr int C0 = Plan::chunk_begin(CH), C1 = Plan::chunk_end(CH);
        if constexpr (C < C1) {
            if constexpr (FirstSplitOf<Plan>::N > 0 && C == FirstSplitOf<Plan>::N) {
                if (fr.first) __syncthreads();              // barrier E of the split-role loop: the early part of the first tile is complete
            }
            if constexpr (Plan::wave_of(C) == G) {
                using D = Dual<T, 1>;
                D xd[NX], ud[NU], yd[NY];
#pragma unroll
                for (int i = 0; i < NX; ++i) { xd[i].v = fr.x[i]; xd[i].d[0] = (i == C) ? T(1) : T(0); }
#pragma unroll
                for (int i = 0; i < NU; ++i) { ud[i].v = fr.u[i]; ud[i].d[0] = (NX + i == C) ? T(1) : T(0); }
                if constexpr (Tune::ABLATE == 2) {            // tuning only: no arithmetic, same store pattern
#pragma unroll
                    for (int i = 0; i < NY; ++i) { yd[i].v = fr.x[i]; yd[i].d[0] = fr.x[i] + T(C); }
                } else fn.eval(xd, ud, fr.t, fr.h, yd);
                T col[NY];
                if constexpr (VALUE_PENDING) {
                    if (fr.want_y) {
#pragma unroll
                        for (int i = 0; i < NY; ++i) col[i] = yd[i].v;
                        fr.put_value(C1 - C0, col);
                    }
                }
                // control columns of a terminal knot point are zero (`!is_terminal(z) * u`, src/knotpoint.jl:67)
                const T sc = (C >= NX && fr.term) ? T(0) : T(1);
#pragma unroll
                for (int i = 0; i < NY; ++i) col[i] = yd[i].d[0] * sc;
                fr.put_at(C - C0, col);
                if constexpr (Tune::COL_BARRIER) __builtin_amdgcn_sched_barrier(0);      // (tuning) finish this column before the scheduler starts interleaving the next one
                AdCols<Fn, T, Plan, CH, G, C + 1, false>::run(fn, fr);
            } else {
                AdCols<Fn, T, Plan, CH, G, C + 1, VALUE_PENDING>::run(fn, fr);
            }
        }
    }
};
// models whose ForwardAD tile loop prefetches through the copy waves and paces their stores (CopyPrefetchOf above; specialised next to the model)
template <class Mdl, class T> struct CopyPrefetchModel { static constexpr bool value = false; static constexpr int PACE = 0; };
template <class Mdl, class T> struct CopyPrefetchFd { static constexpr bool value = false; };       // the same switch for the FiniteDifference body (cascade.hpp)
template <class Mdl, class T> struct CopyPrefetchChain { static constexpr bool value = false; static constexpr int PACE = 0; };       // ... and for the UserDefined chain-rule body
template <class Fn, class T, class Plan> struct AdUnrolledBody {
    static constexpr bool EARLY = true;                     // AdCols passes barrier E at column Plan::FIRST_SPLIT of a first tile
    static constexpr bool COPY_PREFETCH = CopyPrefetchModel<decltype(Fn::m), T>::value;
    static constexpr int COPY_PACE = CopyPrefetchModel<decltype(Fn::m), T>::PACE;
    template <int CH, int G> static RD_DEV void waves(const Fn& fn, JacFrame<Fn, T>& fr) {
        if constexpr (G < Plan::NG) {
            if (fr.g == G) AdCols<Fn, T, Plan, CH, G, Plan::chunk_begin(CH), (CH == Plan::value_chunk && G == Plan::value_wave)>::run(fn, fr);
            else waves<CH, G + 1>(fn, fr);
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) { waves<CH, 0>(fn, fr); }
};
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_ad_unrolled_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                         const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, AdUnrolledBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

// one Jacobian column of one knot point per lane, nothing else: the hipRTC plug-in compiles this per column and uses the code size as
// the column's cost when it balances a custom model's columns over the compute waves (jit.hip)
template <class Fn, class T, int C>
__global__ void col_probe_kernel(Fn fn, const T* __restrict__ Z, T* __restrict__ O) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY;
    using D = Dual<T, 1>;
    D xd[NX], ud[NU], yd[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xd[i].v = Z[threadIdx.x * (NX + NU) + i]; xd[i].d[0] = (i == C) ? T(1) : T(0); }
#pragma unroll
    for (int i = 0; i < NU; ++i) { ud[i].v = Z[threadIdx.x * (NX + NU) + NX + i]; ud[i].d[0] = (NX + i == C) ? T(1) : T(0); }
    fn.eval(xd, ud, T(0), Z[4096], yd);
#pragma unroll
    for (int i = 0; i < NY; ++i) O[threadIdx.x * NY + i] = yd[i].d[0];
}

// ---- split-role variant of the tile loop --------------------------------------------------------------------------
// A copy-out that is issued by the same waves that compute cannot overlap with compute: when HBM is the bottleneck the
// store instructions block AT ISSUE, so a workgroup whose LDS tile fills a whole CU alternates "all compute" / "all
// store" phases (measured: total = store time + compute time).  Here a workgroup has NG COMPUTE waves and NC COPY
// waves.  Per tile:   compute waves fill the LDS tile            | copy waves issue the previous tile's stores from VGPRs
//                     -------- barrier A (tile complete) --------
//                     compute waves refill z/t/dt (prefetched)   | copy waves pull the whole tile LDS -> VGPRs (16 B/lane)
//                     -------- barrier B (LDS free) -------------
// so the HBM write stream of tile t runs under the arithmetic of tile t+1.  Requires 16-byte aligned J and Y (the
// copy is a linear 16-byte sweep over the tile, which is one contiguous slab of global memory).
// tile stores are streaming (written once, never re-read by this kernel): the non-temporal hint measured 3-5 % faster
// than plain stores on MI355X (less dirty data left in L2 for the end-of-kernel write-back); Tune::PLAIN_STORES reverts, Tune::ABLATE == 3 drops the stores
template <class V> RD_DEV void tile_store(V* p, V v, int lane) {
    if constexpr (Tune::ABLATE == 3) { if (lane == 99) *p = v; }
    else if constexpr (Tune::PLAIN_STORES) *p = v;
    else __builtin_nontemporal_store(v, p);
}
#define RD_TILE_STORE(p, v) tile_store((p), (v), lane)
// Tile stores of the copy waves as BUFFER stores: the tile slab's base in a scalar resource descriptor, ONE vector register with the lane's byte offset
// (16 ct) for the whole sweep and the unit's offset (r x 4 096 bytes for four copy waves) as the instruction's scalar offset.  As global stores every
// unit carried its own 64-bit per-lane address: the compiler hoisted those 2 x 31 registers out of the tile loop, and in the batch-table kernels (256
// registers, rdhip_jacobian_multi) spilled some of them and reloaded them INSIDE the store sweep, each reload behind a wait for every store issued so far.
template <class V> RD_DEV void tile_store_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned unit_off, V v, int lane) {
    if constexpr (Tune::ABLATE == 3) { if (lane == 99) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane_off, unit_off, 2); }
    else if constexpr (Tune::PLAIN_STORES) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane_off, unit_off, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane_off, unit_off, 2);      // aux 2 = nt (streaming: written once, never re-read here)
}
#define RD_TILE_STORE_BUF(rsrc, lane_off, unit_off, v) tile_store_buf((rsrc), (lane_off), (unit_off), (v), lane)
RD_DEV __amdgpu_buffer_rsrc_t tile_rsrc(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, (short)0, 0x7fffffff, 0x00020000); }
template <class Fn, class T, class Plan, int NC> struct SplitLds {
    static constexpr int NZ = Fn::NX + Fn::NU, NY = Fn::NY, OW = Fn::NY * Fn::NCOLS, NVAL = Fn::NVAL, TIN = TILE + Fn::NEXT;
    static constexpr size_t J_ELEMS = (size_t)TILE * OW, Y_ELEMS = (size_t)TILE * NVAL, Z_ELEMS = (size_t)TIN * NZ;
    static constexpr size_t J_BYTES = (J_ELEMS * sizeof(T) + 15) / 16 * 16, Y_BYTES = (Y_ELEMS * sizeof(T) + 15) / 16 * 16;
    static constexpr size_t BYTES = J_BYTES + Y_BYTES + (Z_ELEMS * sizeof(T) + 15) / 16 * 16 + 2 * TILE * sizeof(double);
    static constexpr int JU = (int)(J_ELEMS * sizeof(T) / 16), YU = (int)(Y_ELEMS * sizeof(T) / 16);   // whole 16-B units per full tile
    static constexpr int JPL = (JU + NC * TILE - 1) / (NC * TILE), YPL = (YU + NC * TILE - 1) / (NC * TILE);   // units per copy lane
    static_assert(J_ELEMS * sizeof(T) % 16 == 0 && Y_ELEMS * sizeof(T) % 16 == 0, "tile slabs must be multiples of 16 bytes");
};

// LDS table of resolved tiles (descriptors with RING): one 64-byte entry per iteration of the workgroup.  Values are wave-uniform for the
// reader, who turns them back into scalars (v_readfirstlane) so that addresses stay "scalar base + lane offset".
template <class T> RD_DEV void ring_put(long long* slot, const TileRef<T>& r) {
    slot[0] = (long long)r.Z; slot[1] = (long long)r.t; slot[2] = (long long)r.dt; slot[3] = (long long)r.J; slot[4] = (long long)r.Y; slot[5] = r.N; slot[6] = r.k0;
}
template <class T> RD_DEV TileRef<T> ring_get(const long long* slot) {
    long long v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = slot[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = uniform64(v[i]);
    return TileRef<T>{(const T*)v[0], (const double*)v[1], (const double*)v[2], (T*)v[3], (T*)v[4], v[5], v[6]};
}

// a descriptor with WAITS = true makes the compute waves call bt.wait(ref, seen) before they request a tile's inputs (tiles produced by other
// workgroups of the same launch: the fused rollout -> Jacobians kernel, fused.hpp)
template <class BT, class = void> struct WaitsOf { static constexpr bool value = false; struct Seen {}; };
template <class BT> struct WaitsOf<BT, decltype((void)BT::WAITS)> { static constexpr bool value = BT::WAITS; using Seen = typename BT::Seen; };
// tile inputs of a descriptor with WAITS were written by other workgroups of the SAME launch (possibly on another XCD, i.e. behind another L2): they
// are read with agent-scope loads, which bypass the non-coherent caches -- a plain load could be served from a line this compute unit cached
// while it read a neighbouring tile, before this tile's part of the line was produced (tiles that do not start on 128-byte lines)
template <class BT, class = void> struct CoherentOf { static constexpr bool value = WaitsOf<BT>::value; };
template <class BT> struct CoherentOf<BT, decltype((void)BT::COHERENT)> { static constexpr bool value = BT::COHERENT; };      // (tuning: a waiting descriptor with plain loads)
template <bool COHERENT, class T> RD_DEV T tile_load(const T* p) {
    if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
// wg / nwg: this workgroup's index among the nwg workgroups that share the tiles (the whole grid unless the launch has other roles as well)
// [t_begin, t_end): the part of the descriptor's tiles this call walks (t_end < 0: all of them) -- the fused rollout -> Jacobians kernel runs the loop
// twice, over the tiles produced while the rollout workgroups were busy and over the rest, which those workgroups then share (fused.hpp)
template <class Fn, class T, class Plan, int NC, class Body, class BT>
RD_DEV void jac_split_loop(const Fn& fn, const BT& bt, unsigned wg, unsigned nwg, long long t_begin = 0, long long t_end = -1) {
    // per batch: N = knot points that produce output; inputs of N + Fn::NEXT knot points are read (a body with NEXT = 1 also sees x_{k+1})
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, OW = Fn::NY * Fn::NCOLS, NVAL = Fn::NVAL, TIN = TILE + Fn::NEXT, NG = Plan::NG, NTC = NG * TILE;
    constexpr int NA = AuxOf<Fn>::N;
    // early part of a workgroup's first tile: the leading ES columns (ES * NY values of every row), see TablePlanS
    constexpr int ES = (EarlyOf<Body>::value && !GatherOf<BT>::value && !Tune::NO_EARLY) ? FirstSplitOf<Plan>::N : 0;
    static_assert(Plan::NCHUNK == 1, "the split-role kernel stages whole tiles");
    using L = SplitLds<Fn, T, Plan, NC>;
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    // GATHER tiles (runs of GATHER_RUN knot points from GATHER_NRUN places): every run of the LDS output tile is followed by 16 bytes of slack, so
    // that a run can sit at the same offset modulo 16 as its slab in global memory (slabs of n (n+m) values are only sizeof(T)-aligned) and the
    // copy waves still move whole aligned 16-byte units; the dynamic LDS is SPLIT_GATHER_EXTRA bytes larger
    constexpr bool GATHER = GatherOf<BT>::value;
    constexpr int RB = OW * (int)sizeof(T);                                        // bytes of one knot point's slab
    constexpr int GRUN_BYTES = GATHER_RUN * RB + 16;
    constexpr size_t JB = L::J_BYTES + (GATHER ? SPLIT_GATHER_EXTRA : 0);
    static_assert(!GATHER || (Fn::NEXT == 0 && NC == 4 && (GATHER_RUN * RB) % 16 == 0), "gather tiles: four copy waves of GATHER_NRUN / 4 runs each");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* jt = reinterpret_cast<T*>(smem);
    T* yt = reinterpret_cast<T*>(smem + JB);
    T* zs = reinterpret_cast<T*>(smem + JB + L::Y_BYTES);
    double* ts = reinterpret_cast<double*>(smem + JB + L::Y_BYTES + (L::Z_ELEMS * sizeof(T) + 15) / 16 * 16);
    // resolved tiles of this workgroup's iterations 0 .. RING_N - 1 (static LDS next to the dynamic tile, as many entries as fit the CU's 160 KiB)
    constexpr int RING_FREE = (int)(((long long)160 * 1024 - (long long)L::BYTES - 256) / 64);      // (256 bytes stay free for the loop's other static words: sched_ring, hbuf, epi_word -- a tile that
                                                                                                     // left the ring EXACTLY the rest of the CU's LDS could not be launched with the schedule compiled in: SatellitePair fp64, found by the round-6 sweep)
    constexpr int RING_N = !BT::RING ? 0 : RING_FREE >= 64 ? 64 : RING_FREE >= 32 ? 32 : RING_FREE >= 16 ? 16 : RING_FREE >= 8 ? 8 : 0;
    __shared__ long long ring[RING_N > 0 ? RING_N : 1][8];
    constexpr bool HELPER = HelperOf<Body>::value;
    constexpr int NCC = HELPER ? NC - 1 : NC;                // copy waves that copy
    constexpr int JPLc = (L::JU + NCC * TILE - 1) / (NCC * TILE), YPLc = (L::YU + NCC * TILE - 1) / (NCC * TILE);
    // NPASS = 2 (two copy waves, fp32): the copy waves take the tile out of LDS in two halves -- the first half's stores are issued in front of
    // barrier B, the second half's behind it -- and hold half a tile in registers at a time: 64 data registers instead of 128, which is what lets
    // a six-wave workgroup stay under the 168 registers of three waves per SIMD, i.e. TWO workgroups per compute unit (see UnrollPolicy / F32_NC)
    constexpr int NPASS = ((sizeof(T) == 4 || Tune::NPASS_ANY) && NCC == 2 && !GATHER) ? 2 : 1;      // (Tune::NPASS_ANY: fp64 tiles too, tune/tune_ng.hip)
    constexpr int JPP = (JPLc + NPASS - 1) / NPASS;          // units per copy lane and pass
    constexpr bool CPF = ((CopyPrefetchOf<Body>::value && !Tune::COPY_PREFETCH_NONE) || Tune::COPY_PREFETCH_ALL) && !GATHER && NA == 0 && !HELPER && NPASS == 1 && !WaitsOf<BT>::value && NCC * TILE >= 2 * TILE;
    constexpr bool HSTAGE = HELPER && HelperStagesOf<Body>::value && !Tune::NO_HELPER_STAGE && !GATHER && NA == 0 && Fn::NEXT == 0 && !WaitsOf<BT>::value;
    constexpr int PACE = Tune::COPY_PACE > 0 ? Tune::COPY_PACE : (CPF ? CopyPrefetchOf<Body>::PACE : 0);
    constexpr int ZPTC = (TIN * NZ + NCC * TILE - 1) / (NCC * TILE);      // input elements per copy thread
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[HELPER ? 2 * HelperOf<Body>::BYTES : 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ntiles = (t_end >= 0 && t_end < bt.ntiles()) ? t_end : bt.ntiles();
    // tile schedule of this workgroup: tile, tile + tstep, ... < tend.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  For long
    // launches (>= 64 tiles per workgroup, i.e. from ~1 M knot points on a whole chip) every XCD walks its OWN contiguous eighth of the tiles,
    // strided over its workgroups: each XCD's L2 and address-translation caches then see one sequential stream instead of every 8th slab of a
    // multi-GB output (measured against the plain `tile = blockIdx, step = gridDim` order: 0.67 -> 0.72-0.77 of the HBM peak at 2-4 M knot
    // points, +2 % at 1 M).  Short launches keep the plain order: with a handful of tiles per workgroup the XCD ranges cost 1-4 % (and 15 % for
    // 64 fused batches of 1 024 knot points); a contiguous range per WORKGROUP costs 5-9 % at every size.
    long long tile, tend, tstep;
    // (tuning, tune/tune_bign.hip: Tune::tile_map_runtime() = 100 R + v, R = 1, 2, 4 ranges per XCD: every XCD walks R contiguous ranges of ntiles / (8 R)
    // tiles, its workgroups split into R groups; v = 0: an XCD's ranges are adjacent (range xcd R + sub), v = 1: interleaved (range sub 8 + xcd).
    // Tune::TILE_MAP = 0 / 2: the plain order / the XCD ranges at every size.  The library's policy returns 0 and 1: both fold away.)
    const int mp_ = Tune::tile_map_runtime();
    if (mp_ >= 100 && (nwg & 7) == 0) {
        const int R = mp_ / 100, v = mp_ % 100;
        const long long xcd = wg & 7, j = wg >> 3, sub = j % R, jj = j / R, nr = 8LL * R;
        const long long rid = v == 0 ? xcd * R + sub : sub * 8 + xcd, ntr = (ntiles + nr - 1) / nr;
        tile = rid * ntr + jj; tend = ((rid + 1) * ntr < ntiles) ? (rid + 1) * ntr : ntiles; tstep = (nwg >> 3) / R;
    } else if (Tune::TILE_MAP != 0 && (nwg & 7) == 0 && (Tune::TILE_MAP == 2 || ntiles >= 64 * (long long)nwg) && !WaitsOf<BT>::value) {     // (tiles that are waited for keep the order in which they become available)
        const long long ntx = (ntiles + 7) / 8, xcd = wg & 7;
        tile = xcd * ntx + (wg >> 3);
        tend = ((xcd + 1) * ntx < ntiles) ? (xcd + 1) * ntx : ntiles; tstep = nwg >> 3;
    } else { tile = t_begin + wg; tend = ntiles; tstep = nwg; }
    // DYNAMIC schedule (descriptors that carry the launch's counter words; launches of >= 8 tiles per workgroup): a workgroup's first three tiles are wg,
    // wg + nwg, wg + 2 nwg, every further one comes from a device-wide counter, fetched by one lane THREE iterations ahead (the next tile's inputs are
    // requested a whole iteration early, so its number must be known two iterations early) in the quiet window between barriers A and B, where the
    // atomic does not queue behind a tile's stores.  Why: the XCDs do not run at the same rate (stamps, tune/tune_stamp.hip: last store issued 37 ..
    // 47 us after launch, by XCD, at 131 072 satellite knots) and with equal shares a launch ends with its slowest workgroup -- the longer the launch
    // the further apart: 262 144 knots 100.8 -> 94.8 us, 524 288 202.8 -> 181.1, 1 M 398 (XCD ranges) -> 367, 4 M 1 857 -> 1 584 (tune/tune_dyn.hip,
    // same binary, alternating passes, identical outputs).  Below 8 tiles per workgroup the counter costs more than it returns (65 536 knots: +0.6 us),
    // and handing out ALL tiles from it puts 256 atomics on one address in front of every workgroup's first tile (+1.5 us).  The plain order plus the
    // counter also beats the static XCD ranges above, which it replaces when the words are there.  sched = {next, done, eight range counters (XR below)}:
    // all zero between launches -- the last workgroup to leave resets them.
    // (fp64 tiles only: the fp32 kernels run two workgroups per CU on half-size tiles -- twice the fetch rate on the one counter word -- and came out 8 %
    // SLOWER with it, 181 -> 197 us for 1 M satellite knots, same box, RDHIP_TUNE=dynamic_tiles=0/1)
    // (Tune::DYN_F32, tune/tune_dyn.hip -DTT=float: the counter for fp32 tiles too, Tune::DYN_CHUNK tiles per atomic)
    constexpr bool DYN_OK = SchedOf<BT>::HAS && !GATHER && !WaitsOf<BT>::value && !HELPER && !EpilogueOf<Body>::value && (sizeof(T) == 8 || Tune::DYN_F32);
    constexpr int RD_DYN_CHUNK = Tune::DYN_CHUNK;
    constexpr int DYN_STATIC = 3;
    __shared__ long long sched_ring[4];
    unsigned* sched = nullptr;
    if constexpr (DYN_OK) sched = SchedOf<BT>::get(bt);
    const bool dyn = DYN_OK && sched != nullptr && t_begin == 0 && t_end < 0 && ntiles >= 8 * (long long)nwg;
    // XR: the counter per XCD RANGE.  Every XCD walks its own contiguous eighth of the tiles (what the static ranges did: one sequential stream per L2
    // and translation cache), its 32 workgroups taking the range's tiles from the range's counter (words 2 .. 9 of the slot); a workgroup whose range
    // is used up moves on to the next XCD's range, so a fast XCD finishes a slow one's work.
    // Measured against the one counter over the plain order (tune/tune_dyn.hip -DRD_DYN_XR=..., two boxes): 2 M knots 792 -> 766 and 791 -> 716 us, 4 M
    // 1 581 -> 1 518 and 1 567 -> 1 426 us (0.78-0.83 of the roofline), but 1 M 317 -> 328 and 322 -> 336 us: from 128 tiles per workgroup on.
    constexpr int RD_DYN_XR = Tune::DYN_XR;
    const bool xr = dyn && (nwg & 7) == 0 && ntiles >= RD_DYN_XR * (long long)nwg;
    const long long xr_ntx = (ntiles + 7) / 8;
    const int xr_nxw = (int)(nwg >> 3), xr_own = (int)(wg & 7);
    if (dyn) { tile = wg; tend = ntiles; tstep = nwg; }        // (plain order, whatever the static map would have been)
    if (xr) { tile = xr_own * xr_ntx + (wg >> 3); tstep = xr_nxw; }   // the first DYN_STATIC tiles: the head of the own range, strided over the XCD's workgroups
    // (one thread per workgroup, thread 0.)  Its counter fetch-adds must be PERFORMED before its done count is issued -- the workgroup that counts last
    // stores zeros, and a fetch-add that arrives behind them would leave a counter non-zero for the slot's next launch, which would then silently skip
    // tiles: the last three fetches of a workgroup return values nobody waits for, so the wait is explicit (vmcnt(0): this wave has no stores in flight,
    // the copy waves do the storing).  All of these are agent-scope read-modify-writes of one 64-byte line, executed where the line lives; "performed,
    // then issued" orders them there whatever the route.  (The host also re-zeroes a stream's ring every time it comes round: api.hip take_sched.)
    auto leave = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (__hip_atomic_fetch_add(sched + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sched + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (xr) for (int x = 0; x < 8; ++x) __hip_atomic_store(sched + 2 + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // XR, thread 0: the counter value `a` drawn from range `v` as a tile number; if that range is used up, draw from the following ranges (waiting for each
    // answer: at most seven times per workgroup and launch, at its end); ntiles = nothing left anywhere
    int xr_victim = xr_own;
    auto xr_resolve = [&](unsigned a) -> long long {
        for (;;) {
            const long long rend = (xr_victim + 1) * xr_ntx < ntiles ? (xr_victim + 1) * xr_ntx : ntiles;
            const long long t = xr_victim * xr_ntx + (long long)DYN_STATIC * xr_nxw + a;
            if (t < rend) return t;
            xr_victim = (xr_victim + 1) & 7;
            if (xr_victim == xr_own) { xr_victim = -1; return ntiles; }
            a = __hip_atomic_fetch_add(sched + 2 + xr_victim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (tile >= tend) { if (dyn && threadIdx.x == 0) leave(); return; }
    const long long tile0 = tile;
#define RD_NEXT_TILE(it) (dyn ? sched_ring[((it) + 1) & 3] : tile + tstep)
    bool epi_any = false;
    __shared__ int epi_word;
    if (wave == 0) RD_STAMP(0, 13);                        // (slots 13 / 15 of iteration 0: the workgroup's first and last instruction)

    if (wave < NG) {
        // =========================== compute waves ===========================
        constexpr int ZPT = (TIN * NZ + NTC - 1) / NTC;
        // gather tiles: slot s of the tile = knot (s % GATHER_RUN) of run (s / GATHER_RUN); slots that do not exist repeat the last one that does
        auto gather_knot = [](const TileRef<T>& r, int slot) -> long long {
            int run = slot / GATHER_RUN, kk = slot - run * GATHER_RUN;
            run = run < r.nruns ? run : r.nruns - 1; kk = kk < r.run_knots ? kk : r.run_knots - 1;
            return r.k0 + run * r.run_stride + kk;
        };
        auto gather_z = [&](const TileRef<T>& r, int e) -> long long { const int slot = e / NZ; return gather_knot(r, slot) * NZ + (e - slot * NZ); };
        (void)gather_knot; (void)gather_z;
        JacFrame<Fn, T> fr;
        fr.g = wave;
        fr.first = ES > 0;
        fr.row = jt + lane * OW;
        fr.yrow = yt + lane * NVAL;
        fr.zs = zs;
        TileRef<T> cur = at_uniform<T>(bt, tile);
        typename WaitsOf<BT>::Seen wseen; (void)wseen;       // (what this wave knows about the producers' progress: see the descriptor's wait)
        if constexpr (WaitsOf<BT>::value) bt.wait(cur, wseen);
        {   // first tile: all of a thread's loads are issued before any is waited for (a load -> wait -> LDS-write loop costs one memory
            // round trip per pass: five of them in front of the first arithmetic of the kernel)
            const long long k0 = cur.k0, N = cur.N, NIN = N + Fn::NEXT;
            const T* __restrict__ Z = cur.Z;
            const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
            const int nin = (int)((NIN - k0) < TIN ? (NIN - k0) : TIN);
            T z0[ZPT]; double t0;
            if constexpr (GATHER) {
#pragma unroll
                for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; z0[r] = tile_load<CoherentOf<BT>::value>(Z + gather_z(cur, e < TILE * NZ ? e : 0)); }
                const double* tp = (threadIdx.x < TILE) ? cur.t : cur.dt;
                t0 = tp[gather_knot(cur, threadIdx.x & (TILE - 1))];
            } else {
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; z0[r] = tile_load<CoherentOf<BT>::value>(Z + ((e < nin * NZ) ? k0 * NZ + e : 0)); }
            {
                const double* tp = (threadIdx.x < TILE) ? cur.t : cur.dt;
                const int idx = threadIdx.x & (TILE - 1);
                t0 = tp[(threadIdx.x < 2 * TILE && idx < nk) ? k0 + idx : 0];
            }
            }
            if constexpr (NA > 0) {                        // lane-private extra inputs of the first tile (requested behind the tile's own loads)
                const T* __restrict__ A = fn.aux_ptr();
                const long long ka = k0 + (lane < nk ? lane : nk - 1);
#pragma unroll
                for (int i = 0; i < NA; ++i) fr.aux[i] = A[ka * NA + i];
            }
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; if (e < TIN * NZ) zs[e] = z0[r]; }
            if (threadIdx.x < 2 * TILE) ts[threadIdx.x] = t0;
        }
        long long sched_pend = tile + 2 * tstep;           // thread 0: the tile fetched last (tile it + 2 while iteration it runs)
        int sched_have = 1;                                // (RD_DYN_CHUNK > 1: how many consecutive tiles from sched_pend on are this workgroup's)
        unsigned xr_raw = 0; bool xr_pending = false;      // XR: a counter value that has not been turned into a tile number yet
        if (dyn && threadIdx.x == 0) sched_ring[1] = tile + tstep;
        __syncthreads();                                   // barrier 0
        for (int it = 0; tile < tend; tile = RD_NEXT_TILE(it), ++it) {
            const int nk = (int)((cur.N - cur.k0) < TILE ? (cur.N - cur.k0) : TILE);
            if (wave == 0) RD_STAMP(it, 0);
            fr.want_y = (cur.Y != nullptr);
            const long long nxt = RD_NEXT_TILE(it);
            const bool more = nxt < tend;
            // (the last iteration re-reads element 0 of its own batch: the loads stay unconditional)
            TileRef<T> nx;
            if (RING_N > 0 && !dyn && it + 1 < RING_N) nx = more ? ring_get<T>(ring[it + 1]) : cur;
            else nx = at_uniform<T>(bt, more ? nxt : tile);
            if constexpr (WaitsOf<BT>::value) { if (more) bt.wait(nx, wseen); }
            const long long k1 = nx.k0;
            const T* __restrict__ Z = nx.Z;
            const int nk1 = more ? (int)((nx.N - k1) < TILE ? (nx.N - k1) : TILE) : 0;
            const int nin1 = more ? (int)((nx.N + Fn::NEXT - k1) < TIN ? (nx.N + Fn::NEXT - k1) : TIN) : 0;
            T zpre[ZPT]; double tpre;                       // unconditional prefetch of the next tile's inputs
            if constexpr (GATHER) {
#pragma unroll
                for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; zpre[r] = tile_load<CoherentOf<BT>::value>(Z + gather_z(nx, e < TILE * NZ ? e : 0)); }
                const double* tp = (threadIdx.x < TILE) ? nx.t : nx.dt;
                tpre = tp[gather_knot(nx, threadIdx.x & (TILE - 1))];
            } else if constexpr (!CPF && !HSTAGE) {
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; zpre[r] = tile_load<CoherentOf<BT>::value>(Z + ((e < nin1 * NZ) ? k1 * NZ + e : 0)); }
            {
                const double* tp = (threadIdx.x < TILE) ? nx.t : nx.dt;
                const int idx = threadIdx.x & (TILE - 1);
                tpre = tp[(threadIdx.x < 2 * TILE && idx < nk1) ? k1 + idx : 0];
            }
            }
            T apre[NA > 0 ? NA : 1];
            if constexpr (NA > 0) {
                const T* __restrict__ A = fn.aux_ptr();
                const long long ka = more ? k1 + (lane < nk1 ? lane : nk1 - 1) : 0;
#pragma unroll
                for (int i = 0; i < NA; ++i) apre[i] = A[ka * NA + i];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kl = GATHER ? lane : (lane < nk ? lane : nk - 1);
            if constexpr (GATHER) {                          // this lane's row: run (lane / 4) of the tile, shifted like the run's slab in global memory
                const int run = lane / GATHER_RUN, kk = lane - run * GATHER_RUN;
                const unsigned padb = (unsigned)(reinterpret_cast<unsigned long long>(cur.J + (cur.k0 + run * cur.run_stride) * OW) & 15);
                fr.row = reinterpret_cast<T*>(smem + (size_t)run * GRUN_BYTES + padb) + kk * OW;
                fr.want_y = false;
            }
            const double hd = ts[TILE + kl];
            fr.t = (T)ts[kl]; fr.h = (T)hd; fr.term = is_terminal_dt(hd);
            read_knot<Fn, T>(zs, kl, fr.term, fr.x, fr.u);
            fr.kl = kl;
            fr.tile = tile;
            fr.helper = hbuf + (size_t)(it & 1) * HelperOf<Body>::BYTES;
            Body::template run<0>(fn, fr);
            RD_STAMP(it, 4 + wave);
            fr.first = false;
            if constexpr (NA > 0) {
#pragma unroll
                for (int i = 0; i < NA; ++i) fr.aux[i] = apre[i];
            }
            __syncthreads();                               // barrier A: tile complete; all compute waves are done with zs/ts
            if (wave == 0) RD_STAMP(it, 2);
            if (dyn && threadIdx.x == 0) {                 // tile it + 2 becomes visible behind barrier B; tile it + 3 is requested
                if (xr) {
                    if (xr_pending) sched_pend = xr_resolve(xr_raw);
                    sched_ring[(it + 2) & 3] = sched_pend;
                    if (xr_victim >= 0) { xr_raw = __hip_atomic_fetch_add(sched + 2 + xr_victim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); xr_pending = true; }
                    else { sched_pend = ntiles; xr_pending = false; }
                } else {
                sched_ring[(it + 2) & 3] = sched_pend;
                if (RD_DYN_CHUNK > 1 && sched_have > 1) { ++sched_pend; --sched_have; }
                else {
                    sched_pend = (long long)DYN_STATIC * nwg + __hip_atomic_fetch_add(sched, (unsigned)RD_DYN_CHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sched_have = RD_DYN_CHUNK;
                }
                }
            }
            if constexpr (!CPF && !HSTAGE) {
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTC + threadIdx.x; if (e < TIN * NZ) zs[e] = zpre[r]; }
            if (threadIdx.x < 2 * TILE) ts[threadIdx.x] = tpre;
            }
            __syncthreads();                               // barrier B: LDS tile drained into the copy waves' registers
            if (wave == 0) RD_STAMP(it, 3);
            cur = nx;
        }
        epi_any = fr.epi;
        if (dyn && threadIdx.x == 0) leave();
    } else if (HELPER && wave == NG + NC - 1) {
        // =========================== helper wave =============================
        if constexpr (HELPER) {
            Body::helper_tile(fn, at_uniform<T>(bt, tile), hbuf);
            __syncthreads();                                   // barrier 0
            bool first = ES > 0;
            for (int it = 0; tile < tend; tile = RD_NEXT_TILE(it), ++it) {
                if (!Tune::HELPER_CHAIN_FIRST && first) { __syncthreads(); first = false; } // barrier E first: the early part of the first tile must not wait for the next tile's chain
                const long long nxt = RD_NEXT_TILE(it);
                RD_STAMP(it, 11);
                T hz[HSTAGE ? NZ : 1]; double ht = 0, hh = 0;      // HSTAGE: this lane's knot point of the next tile (lanes past a ragged tile's end repeat its last one)
                if constexpr (HSTAGE) {
                    if (nxt < tend) Body::helper_tile(fn, at_uniform<T>(bt, nxt), hbuf + (size_t)((it + 1) & 1) * HelperOf<Body>::BYTES, 0, hz, &ht, &hh);
                } else {
                if (nxt < tend) Body::helper_tile(fn, at_uniform<T>(bt, nxt), hbuf + (size_t)((it + 1) & 1) * HelperOf<Body>::BYTES);
                }
                RD_STAMP(it, 12);
                if (Tune::HELPER_CHAIN_FIRST && first) { __syncthreads(); first = false; }  // (tuning: the next tile's chain in front of barrier E)
                __syncthreads();                               // barrier A
                if constexpr (HSTAGE) {                        // the compute waves are done with the input tile
                    if (nxt < tend) {
                        const int hl = threadIdx.x & 63;
#pragma unroll
                        for (int i = 0; i < NZ; ++i) zs[hl * NZ + i] = hz[i];
                        ts[hl] = ht; ts[TILE + hl] = hh;
                    }
                }
                __syncthreads();                               // barrier B
            }
        }
    } else {
        // =========================== copy waves ==============================
        const int ct = threadIdx.x - NG * TILE;            // 0 .. NCC*64-1
        const V4* jsrc = reinterpret_cast<const V4*>(jt);
        const V4* ysrc = reinterpret_cast<const V4*>(yt);
        __builtin_amdgcn_s_setprio(Tune::COPY_PRIO);       // the copy waves' few instructions are latency-critical for the store stream
        if constexpr (RING_N > 0) {                        // one lane per iteration, all lookups in flight together, while the compute waves fetch the first inputs
            const long long ti = tile + (long long)ct * tstep;
            if (ct < RING_N && ti < tend) ring_put<T>(ring[ct], bt.at(ti));
        }
        __syncthreads();                                   // barrier 0
        bool first = ES > 0;
        // 16-byte unit u of the tile slab lies in a 128-byte line that is covered by the early columns [0, ES) of ONE row
        auto early_line = [](int u) {
            constexpr int RB = OW * (int)sizeof(T), EB = ES * Fn::NY * (int)sizeof(T);
            const int l0 = (u >> 3) << 7, k = l0 / RB;
            return l0 + 128 <= k * RB + EB;
        };
        TileRef<T> cur = at_uniform<T>(bt, tile);
        if constexpr (GATHER) {
            // gather tiles: copy wave w moves runs RPW w .. RPW w + RPW - 1 (RPW = GATHER_NRUN / 4 = 1).  A run is GATHER_RUN consecutive slabs = one contiguous piece of global memory that
            // starts at byte A (only sizeof(T)-aligned); in LDS it sits at the same offset modulo 16 (padb = A & 15), so the wave moves the ALIGNED
            // 16-byte units that cover it -- whole 128-byte lines except at the two ends of the run (3 % of its lines) -- and handles the at most two
            // partly covered units element by element.
            constexpr int UPR = (GRUN_BYTES + 15) / 16, PASSES = (UPR + TILE - 1) / TILE, RPW = GATHER_NRUN / 4;      // units per run, sweeps per run, runs per wave
            const int cw = ct >> 6, cl = ct & 63;
            for (; tile < tend; tile += tstep) {
                V4 gv[RPW * PASSES];
                __syncthreads();                           // barrier A
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const unsigned char* lrun = smem + (size_t)(cw * RPW + q) * GRUN_BYTES;
#pragma unroll
                    for (int p2 = 0; p2 < PASSES; ++p2) { const int u = p2 * TILE + cl; gv[q * PASSES + p2] = *reinterpret_cast<const V4*>(lrun + 16 * (u < UPR ? u : 0)); }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): the tile is in registers
                __syncthreads();                           // barrier B
                const long long nxt_tile = tile + tstep;
                const TileRef<T> nxt = at_uniform<T>(bt, nxt_tile < tend ? nxt_tile : tile);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const int run = cw * RPW + q;
                    unsigned char* A = reinterpret_cast<unsigned char*>(cur.J + (cur.k0 + run * cur.run_stride) * OW);
                    const int padb = (int)(reinterpret_cast<unsigned long long>(A) & 15);
                    const int lo = padb, hi = padb + (run < cur.nruns ? cur.run_knots * RB : 0);       // valid bytes of the unit grid that starts at A - padb
                    unsigned char* G0 = A - padb;
#pragma unroll
                    for (int p2 = 0; p2 < PASSES; ++p2) {
                        const int u = p2 * TILE + cl, b0 = 16 * u;
                        if (b0 >= lo && b0 + 16 <= hi) RD_TILE_STORE(reinterpret_cast<V4*>(G0 + b0), gv[q * PASSES + p2]);
                        else if (b0 + 16 > lo && b0 < hi) {       // a unit at an end of the run: the elements inside [lo, hi)
                            const T* pv = reinterpret_cast<const T*>(&gv[q * PASSES + p2]);
                            T* pd = reinterpret_cast<T*>(G0 + b0);
                            for (int e = 0; e < (int)(16 / sizeof(T)); ++e) { const int be = b0 + e * (int)sizeof(T); if (be >= lo && be + (int)sizeof(T) <= hi) pd[e] = pv[e]; }
                        }
                    }
                }
                cur = nxt;
            }
        } else
        for (int it = 0; tile < tend; tile = RD_NEXT_TILE(it), ++it) {
            const long long k0 = cur.k0;
            const int nk = (int)((cur.N - k0) < TILE ? (cur.N - k0) : TILE);
            T* __restrict__ J = cur.J; T* __restrict__ Y = cur.Y;
            V4 jv[JPP], yv[YPLc];
            T zin[CPF ? ZPTC : 1]; double tin = 0;             // CPF: the next tile's inputs, requested now (behind the previous tile's stores in this wave's queue)
            if constexpr (CPF) {
                const long long nt_ = RD_NEXT_TILE(it);
                const bool more_ = nt_ < tend;
                TileRef<T> nx_;
                if (RING_N > 0 && !dyn && it + 1 < RING_N) nx_ = more_ ? ring_get<T>(ring[it + 1]) : cur;
                else nx_ = at_uniform<T>(bt, more_ ? nt_ : tile);
                const long long k1_ = nx_.k0;
                const int nk1_ = more_ ? (int)((nx_.N - k1_) < TILE ? (nx_.N - k1_) : TILE) : 0;
                const int nin1_ = more_ ? (int)((nx_.N + Fn::NEXT - k1_) < TIN ? (nx_.N + Fn::NEXT - k1_) : TIN) : 0;
                const T* __restrict__ Zn_ = nx_.Z;
#pragma unroll
                for (int r = 0; r < ZPTC; ++r) { const int e = r * (NCC * TILE) + ct; zin[r] = Zn_[(e < nin1_ * NZ) ? k1_ * NZ + e : 0]; }
                const double* tp_ = (ct < TILE) ? nx_.t : nx_.dt;
                const int idx_ = ct & (TILE - 1);
                tin = tp_[(ct < 2 * TILE && idx_ < nk1_) ? k1_ + idx_ : 0];
            }
            bool skip_early = false;
            if constexpr (ES > 0) {
                if (first) {
                    // first tile: the compute waves pass barrier E at column ES; the 128-byte lines that lie entirely inside the early columns
                    // of one row leave now, under the arithmetic of the remaining columns (whole lines only: partially written lines are
                    // what made every chunked store pattern slow), and the sweep behind barrier B skips them
                    __syncthreads();                       // barrier E
                    if (nk == TILE) {
                        const __amdgpu_buffer_rsrc_t jr = tile_rsrc(cur.J + k0 * OW);
#pragma unroll
                        for (int r = 0; r < JPLc; ++r) {
                            const int u = r * (NCC * TILE) + ct;
                            if (u < L::JU && early_line(u)) RD_TILE_STORE_BUF(jr, (unsigned)ct * 16u, (unsigned)(r * (NCC * TILE) * 16), jsrc[u]);
                        }
                        skip_early = true;
                    }
                    first = false;
                }
            }
            __syncthreads();                               // barrier A
            if (ct < TILE) RD_STAMP(it, 8);
            // number of valid bytes of a ragged last tile is nk*OW*sizeof(T), a multiple of sizeof(T) only
            const int jbytes = nk * OW * (int)sizeof(T), ybytes = nk * NVAL * (int)sizeof(T);
            V4* jdst = reinterpret_cast<V4*>(J + k0 * OW);
            const __amdgpu_buffer_rsrc_t jrs = tile_rsrc(J + k0 * OW);
            TileRef<T> nxt;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
                for (int r = 0; r < JPP; ++r) { const int u = (ps * JPP + r) * (NCC * TILE) + ct; jv[r] = jsrc[(ps * JPP + r < JPLc && u < L::JU) ? u : 0]; }
                if (ps == NPASS - 1) {
#pragma unroll
                    for (int r = 0; r < YPLc; ++r) { const int u = r * (NCC * TILE) + ct; yv[r] = ysrc[u < L::YU ? u : 0]; }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this part of the tile is in registers
                if (ps == NPASS - 1) {
                    if (ct < TILE) RD_STAMP(it, 9);
                    if constexpr (CPF) {                   // the compute waves are past barrier A: the input tile is free
#pragma unroll
                        for (int r = 0; r < ZPTC; ++r) { const int e = r * (NCC * TILE) + ct; if (e < TIN * NZ) zs[e] = zin[r]; }
                        if (ct < 2 * TILE) ts[ct] = tin;
                    }
                    __syncthreads();                       // barrier B
                    // where the NEXT tile lives, looked up now: with a batch table this is two dependent scalar-memory round trips, which must pass under
                    // this tile's store issue and not between the last store and barrier A of the next tile (where the compute waves are already waiting)
                    const long long nxt_tile = RD_NEXT_TILE(it);
                    if (RING_N > 0 && !dyn && it + 1 < RING_N) nxt = nxt_tile < tend ? ring_get<T>(ring[it + 1]) : cur;
                    else nxt = at_uniform<T>(bt, nxt_tile < tend ? nxt_tile : tile);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < JPP; ++r) {
                    if (ps * JPP + r >= JPLc) continue;
                    const int u = (ps * JPP + r) * (NCC * TILE) + ct;
                    if (ES > 0 && skip_early && early_line(u)) continue;
                    if (u * 16 + 16 <= jbytes) {
                        RD_TILE_STORE_BUF(jrs, (unsigned)ct * 16u, (unsigned)((ps * JPP + r) * (NCC * TILE) * 16), jv[r]);
                        if constexpr (PACE > 0) __builtin_amdgcn_s_sleep(PACE);
                    }
                    else if (u * 16 < jbytes) {            // partial last unit of a ragged tile: element-wise
                        const T* pv = reinterpret_cast<const T*>(&jv[r]);
                        T* pd = reinterpret_cast<T*>(jdst + u);
                        for (int q = 0; q < (int)(16 / sizeof(T)); ++q) if (u * 16 + (q + 1) * (int)sizeof(T) <= jbytes) pd[q] = pv[q];
                    }
                }
            }
            if (Y) {
                V4* ydst = reinterpret_cast<V4*>(Y + k0 * NVAL);
                const __amdgpu_buffer_rsrc_t yrs = tile_rsrc(Y + k0 * NVAL);
#pragma unroll
                for (int r = 0; r < YPLc; ++r) {
                    const int u = r * (NCC * TILE) + ct;
                    if (u * 16 + 16 <= ybytes) RD_TILE_STORE_BUF(yrs, (unsigned)ct * 16u, (unsigned)(r * (NCC * TILE) * 16), yv[r]);
                    else if (u * 16 < ybytes) {
                        const T* pv = reinterpret_cast<const T*>(&yv[r]);
                        T* pd = reinterpret_cast<T*>(ydst + u);
                        for (int q = 0; q < (int)(16 / sizeof(T)); ++q) if (u * 16 + (q + 1) * (int)sizeof(T) <= ybytes) pd[q] = pv[q];
                    }
                }
            }
            if (ct < TILE) RD_STAMP(it, 10);
            cur = nxt;
        }
    }
    // a body may have work behind the tile loop (the implicit-function-theorem kernel redoes the tiles that asked for the exact solve): every wave of
    // the workgroup has passed the same barriers, the copy waves' stores are acknowledged, and the whole dynamic LDS is free again
    // -- but only a workgroup that has such a tile waits for its stores (in front of the epilogue's own loads that wait is the launch's drain plus one
    // dependent round trip per tile for the flag words, 2 us at the end of EVERY workgroup of a 30 us launch): compute wave 0 saw every tile's verdict
    if constexpr (EpilogueOf<Body>::value) {
        if (threadIdx.x == 0) epi_word = epi_any ? 1 : 0;
        __syncthreads();
        if (epi_word != 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
            __syncthreads();
            Body::epilogue(fn, bt, tile0, tend, tstep, smem, (size_t)L::BYTES);
        }
    }
    if (wave == 0) RD_STAMP(0, 15);
}
#undef RD_NEXT_TILE
template <class Fn, class T, class Plan, int NC, class Body, class BT>
RD_DEV void jac_split_loop(const Fn& fn, const BT& bt) { jac_split_loop<Fn, T, Plan, NC, Body, BT>(fn, bt, blockIdx.x, gridDim.x); }
template <class Fn, class T, class Plan, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE, (sizeof(T) == 4 ? RD_F32_MINB : 1)) jac_ad_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                               const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, J, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}
// the same loops over several independent packed trajectories (rdhip_jacobian_multi), for ANY body: ForwardAD with compile-time or run-time columns,
// FiniteDifference (plain or along the cascade), the UserDefined chain rule, the implicit-function-theorem columns.  MINB: minimum workgroups per
// CU the register allocation must allow (the ForwardAD flagship keeps the occupancy of its single-batch kernel)
template <class Fn, class T, class Plan, int NC, class Body, int MINB, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE, MINB) jac_split_multi_kernel(Fn fn, MultiBatch<T> mb, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, Body>(fn, MultiBatchSched<T>{mb, sched});
    else jac_split_loop<Fn, T, Plan, NC, Body>(fn, mb);
}
template <class Fn, class T, class Plan, class Body>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_tile_multi_kernel(Fn fn, MultiBatch<T> mb) {
    jac_tile_loop<Fn, T, Plan, Body>(fn, mb);
}

// ---- small batches: one tile's columns over FOUR workgroups -----------------------------------------------------------------------------
// A batch of a few thousand knot points (the reference's unit: ONE trajectory of 101 - 4 096 knot points) occupies at most 64 of the 256 CUs with
// the tile loop above, and what it costs is a latency: launch, input round trip, then ~900 dependent fp64 instructions per compute wave (3.5 us),
// then the stores.  Here workgroup (tile, g) computes only the columns the plan gives to compute wave g, spread over its own four waves (a wave
// holds one or two columns: ~350 instead of ~900 instructions behind the inputs), so a 4 096-knot batch runs on all 256 CUs.  Every wave moves
// its finished columns through a wave-private LDS buffer and stores them as 104-byte runs (8 B per lane, consecutive lanes consecutive
// addresses); the partially written lines this leaves are harmless at these sizes -- the whole output is a few MB and stays in the caches until
// it is complete (at 65 536 knot points the same pattern would cost what it cost the first ImplicitMidpoint kernel, implicit.hpp).
// Same arithmetic per column as AdCols (same seeds, same fn.eval): results are bit-identical to the tile loop's.
template <class Plan, int G, int C> struct GroupIndex {          // position of column C among the columns of plan wave G (compile time)
    static constexpr int value() { int k = 0; for (int c = 0; c < C; ++c) k += Plan::wave_of(c) == G ? 1 : 0; return k; }
};
template <class Fn, class T, class Plan, int G, int W, int C> struct ColSplitCols {
    static RD_DEV void run(const Fn& fn, const T (&x)[Fn::NX], const T (&u)[Fn::NU], T t, T h, bool term, T* buf, T* __restrict__ Jt, int nk, int lane) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY, NZ = NX + NU;
        if constexpr (C < NZ) {
            if constexpr (Plan::wave_of(C) == G && (GroupIndex<Plan, G, C>::value() & 3) == W) {
                using D = Dual<T, 1>;
                D xd[NX], ud[NU], yd[NY];
#pragma unroll
                for (int i = 0; i < NX; ++i) { xd[i].v = x[i]; xd[i].d[0] = (i == C) ? T(1) : T(0); }
#pragma unroll
                for (int i = 0; i < NU; ++i) { ud[i].v = u[i]; ud[i].d[0] = (NX + i == C) ? T(1) : T(0); }
                fn.eval(xd, ud, t, h, yd);
                const T sc = (C >= NX && term) ? T(0) : T(1);      // control columns of a terminal knot point are zero
#pragma unroll
                for (int i = 0; i < NY; ++i) buf[lane * NY + i] = yd[i].d[0] * sc;
                wave_sync();
                // 64 rows x NY values: element e = row * NY + i of the buffer goes to row's slab at column C
#pragma unroll
                for (int it = 0; it < NY; ++it) {
                    const int e = it * TILE + lane, row = e / NY, i = e - row * NY;
                    if (row < nk) Jt[(size_t)row * NY * NZ + C * NY + i] = buf[e];
                }
                wave_sync();
            }
            ColSplitCols<Fn, T, Plan, G, W, C + 1>::run(fn, x, u, t, h, term, buf, Jt, nk, lane);
        }
    }
};
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(4 * TILE) jac_ad_colsplit_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                  const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY;
    static_assert(Plan::NG == 4 && Fn::NEXT == 0, "four column groups; bodies that read the next knot point use the tile loop");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* zs = reinterpret_cast<T*>(smem);                                                    // 64 x NZ inputs
    double* ts = reinterpret_cast<double*>(smem + ((size_t)TILE * NZ * sizeof(T) + 15) / 16 * 16);        // t[64], dt[64]
    T* bufs = reinterpret_cast<T*>(smem + ((size_t)TILE * NZ * sizeof(T) + 15) / 16 * 16 + 2 * TILE * sizeof(double));   // 4 x (64 x NY)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the four workgroups of a tile write into the same 128-byte lines: keep them on ONE XCD (workgroups are dealt to the 8 XCDs round-robin), so
    // that their pieces of a line meet in one L2 instead of leaving four XCDs as partial lines.  Grid = 32 x ceil(ntiles / 8).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long tile = (long long)(slot >> 2) * 8 + xcd;
    const int grp = slot & 3;
    if (tile * TILE >= N) return;
    const long long k0 = tile * TILE;
    const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
    {
        const double* tp = (threadIdx.x < TILE) ? tt : dtt;
        const int idx = threadIdx.x & (TILE - 1);
        const double t0 = tp[(threadIdx.x < 2 * TILE && idx < nk) ? k0 + idx : 0];
        stage_in<T, TILE * NZ, 4 * TILE>(Z + k0 * NZ, zs, nk * NZ, threadIdx.x);
        if (threadIdx.x < 2 * TILE) ts[threadIdx.x] = t0;
    }
    __syncthreads();
    const int kl = lane < nk ? lane : nk - 1;
    const double hd = ts[TILE + kl];
    const T t = (T)ts[kl], h = (T)hd;
    const bool term = is_terminal_dt(hd);
    T x[NX], u[NU];
    read_knot<Fn, T>(zs, kl, term, x, u);
    T* buf = bufs + (size_t)wave * TILE * NY;
    T* __restrict__ Jt = J + k0 * NY * NZ;
    if (Y && grp == Plan::value_wave && wave == 3) {              // f(z): the last wave of the value group (it carries the fewest columns)
        T y[NY];
        fn.eval(x, u, t, h, y);
#pragma unroll
        for (int i = 0; i < NY; ++i) buf[lane * NY + i] = y[i];
        wave_sync();
#pragma unroll
        for (int it = 0; it < NY; ++it) { const int e = it * TILE + lane; if (e < nk * NY) Y[k0 * NY + e] = buf[e]; }
        wave_sync();
    }
#define RD_COLSPLIT_CASE(G_, W_) if (grp == G_ && wave == W_) ColSplitCols<Fn, T, Plan, G_, W_, 0>::run(fn, x, u, t, h, term, buf, Jt, nk, lane)
    RD_COLSPLIT_CASE(0, 0); else RD_COLSPLIT_CASE(0, 1); else RD_COLSPLIT_CASE(0, 2); else RD_COLSPLIT_CASE(0, 3);
    else RD_COLSPLIT_CASE(1, 0); else RD_COLSPLIT_CASE(1, 1); else RD_COLSPLIT_CASE(1, 2); else RD_COLSPLIT_CASE(1, 3);
    else RD_COLSPLIT_CASE(2, 0); else RD_COLSPLIT_CASE(2, 1); else RD_COLSPLIT_CASE(2, 2); else RD_COLSPLIT_CASE(2, 3);
    else RD_COLSPLIT_CASE(3, 0); else RD_COLSPLIT_CASE(3, 1); else RD_COLSPLIT_CASE(3, 2); else RD_COLSPLIT_CASE(3, 3);
#undef RD_COLSPLIT_CASE
}
template <class Fn, class T> struct ColSplitLds {
    static constexpr size_t BYTES = ((size_t)TILE * (Fn::NX + Fn::NU) * sizeof(T) + 15) / 16 * 16 + 2 * TILE * sizeof(double) + (size_t)4 * TILE * Fn::NY * sizeof(T);
};

// ---- error-state expansion: [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k] fused into the Jacobian kernel ----------------------
// (error_expansion! of TrajectoryOptimization.jl; G = errstate_jacobian!, src/liestate.jl:262-298).  G is block diagonal with
// identities except for the 4x3 blocks L(q)H of the normalised quaternions, so the projection costs no extra pass:
//   columns: the tangent is seeded in ERROR coordinates -- a unit vector for a vector part (compile-time constant, folds), the
//            column j of L(q)H on the quaternion entries for an attitude part (3 passes per rotation instead of 4);
//   rows:    the 4 quaternion rows of every finished column are contracted with L(q+)H of the NEXT knot point's quaternion.
// Output per knot point: e x (e+m) instead of n x (n+m) (12x18 vs 13x19 for a rigid body: 12.5 % fewer bytes).
template <class T> RD_DEV void lh_column(const T (&q)[4], int j, T (&c)[4]) {     // column j of L(q) H, q = [w x y z]
    if (j == 0) { c[0] = -q[1]; c[1] = q[0]; c[2] = q[3]; c[3] = -q[2]; }
    else if (j == 1) { c[0] = -q[2]; c[1] = -q[3]; c[2] = q[0]; c[3] = q[1]; }
    else { c[0] = -q[3]; c[1] = q[2]; c[2] = -q[1]; c[3] = q[0]; }
}
// rows of one finished column: vector rows are copied, the 4 quaternion rows of rotation r contract with L(q+)H (compile-time row index)
template <class Mdl, class T, int R, int NE> struct ErrRows {
    template <class D> static RD_DEV void apply(const D* yd, const T (&qn)[Mdl::NROT > 0 ? Mdl::NROT : 1][4], T sc, T (&out)[NE]) {
        if constexpr (R < NE) {
            constexpr ErrSlot sl = err_slot<Mdl>(R);
            if constexpr (sl.rot < 0) out[R] = yd[sl.s].d[0] * sc;
            else {
                T col[4]; lh_column<T>(qn[sl.rot], sl.j, col);
                out[R] = (col[0] * yd[sl.s].d[0] + col[1] * yd[sl.s + 1].d[0] + col[2] * yd[sl.s + 2].d[0] + col[3] * yd[sl.s + 3].d[0]) * sc;
            }
            ErrRows<Mdl, T, R + 1, NE>::apply(yd, qn, sc, out);
        }
    }
};
template <class Fn, class T, class Plan, int CH, int G, int C, bool VALUE_PENDING> struct ErrCols {
    using Mdl = decltype(Fn::m);
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr, const T (&qc)[Mdl::NROT > 0 ? Mdl::NROT : 1][4], const T (&qn)[Mdl::NROT > 0 ? Mdl::NROT : 1][4]) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NE = Fn::NE;
        constexpr int C0 = Plan::chunk_begin(CH), C1 = Plan::chunk_end(CH);
        if constexpr (C < C1) {
            if constexpr (FirstSplitOf<Plan>::N > 0 && C == FirstSplitOf<Plan>::N) {
                if (fr.first) __syncthreads();              // barrier E of the split-role loop: the early part of the first tile is complete
            }
            if constexpr (Plan::wave_of(C) == G) {
                using D = Dual<T, 1>;
                D xd[NX], ud[NU], yd[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { xd[i].v = fr.x[i]; xd[i].d[0] = T(0); }
#pragma unroll
                for (int i = 0; i < NU; ++i) { ud[i].v = fr.u[i]; ud[i].d[0] = (NE + i == C) ? T(1) : T(0); }
                if constexpr (C < NE) {
                    constexpr ErrSlot sl = err_slot<Mdl>(C);
                    if constexpr (sl.rot < 0) xd[sl.s].d[0] = T(1);
                    else {
                        T col[4]; lh_column<T>(qc[sl.rot], sl.j, col);
#pragma unroll
                        for (int i = 0; i < 4; ++i) xd[sl.s + i].d[0] = col[i];
                    }
                }
                fn.eval(xd, ud, fr.t, fr.h, yd);
                if constexpr (VALUE_PENDING) {
                    if (fr.want_y) {
                        T val[NX];
#pragma unroll
                        for (int i = 0; i < NX; ++i) val[i] = yd[i].v;
                        fr.put_value(C1 - C0, val);
                    }
                }
                const T sc = (C >= NE && fr.term) ? T(0) : T(1);       // control columns of a terminal knot point are zero
                T out[NE];
                ErrRows<Mdl, T, 0, NE>::apply(yd, qn, sc, out);
                fr.put_at(C - C0, out);
                ErrCols<Fn, T, Plan, CH, G, C + 1, false>::run(fn, fr, qc, qn);
            } else {
                ErrCols<Fn, T, Plan, CH, G, C + 1, VALUE_PENDING>::run(fn, fr, qc, qn);
            }
        }
    }
};
template <class Fn, class T, class Plan> struct ErrUnrolledBody {
    static constexpr bool EARLY = true;
    using Mdl = decltype(Fn::m);
    static constexpr int NR = Mdl::NROT > 0 ? Mdl::NROT : 1;
    template <int CH, int G> static RD_DEV void waves(const Fn& fn, JacFrame<Fn, T>& fr, const T (&qc)[NR][4], const T (&qn)[NR][4]) {
        if constexpr (G < Plan::NG) {
            if (fr.g == G) ErrCols<Fn, T, Plan, CH, G, Plan::chunk_begin(CH), (CH == Plan::value_chunk && G == Plan::value_wave)>::run(fn, fr, qc, qn);
            else waves<CH, G + 1>(fn, fr, qc, qn);
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NZ = Fn::NX + Fn::NU;
        T qc[NR][4], qn[NR][4];                              // normalised quaternions of this and of the next knot point
#pragma unroll
        for (int r = 0; r < Mdl::NROT; ++r) {
            const int s0 = Mdl::rot_start(r);
            const T* xn = fr.zs + (fr.kl + 1) * NZ + s0;
            const T a0 = fr.x[s0], a1 = fr.x[s0 + 1], a2 = fr.x[s0 + 2], a3 = fr.x[s0 + 3];
            const T b0 = xn[0], b1 = xn[1], b2 = xn[2], b3 = xn[3];
            const T ia = T(1) / sqrt_(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3), ib = T(1) / sqrt_(b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3);
            qc[r][0] = a0 * ia; qc[r][1] = a1 * ia; qc[r][2] = a2 * ia; qc[r][3] = a3 * ia;
            qn[r][0] = b0 * ib; qn[r][1] = b1 * ib; qn[r][2] = b2 * ib; qn[r][3] = b3 * ib;
        }
        waves<CH, 0>(fn, fr, qc, qn);
    }
};
template <class Fn, class T, class Plan, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_errexp_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                   const double* __restrict__ dtt, T* __restrict__ E, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, ErrUnrolledBody<Fn, T, Plan>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, E, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, ErrUnrolledBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, E, Y});
}

// ---- small models (n + m <= 10): ONE wave owns 64 knot points and ALL Jacobian columns ---------------------------------
// For a small model the per-wave column split above replicates the primal (and, for the cartpole, its sin/cos and divisions)
// once per wave for a few hundred bytes of output.  Here a lane propagates all n+m tangents at once (Dual<T, n+m>, seeds are
// compile-time constants so structural zeros fold), the wave's 64 x (n x (n+m) + n) results are transposed through a
// wave-private LDS region and leave as linear 16-B-per-lane sweeps (a wave's J and f(z) tiles are contiguous slabs).
// Four independent waves per workgroup (no workgroup barrier), one-shot grid.
// Persistent: wave w of workgroup b walks tiles 4 b + w, + 4 gridDim, ...; the next tile's inputs are requested before the current tile is
// computed.  The wave's LDS region holds the J tile and the f(z) tile in their GLOBAL layout (a tile is a contiguous slab of each array), so the
// copy-out is a linear sweep of 16-byte units -- ds_read_b128 + non-temporal global store, whole 128-byte lines, the LDS offsets fixed per lane --
// instead of one 8-byte store with a division per element (which was 15 % of the kernel's instructions, and the kernel is VALU-bound: the cartpole's
// RK4 Jacobian is ~1 450 fp64 instructions per 64 knot points against 12 KB of output).
template <class Fn, class T> struct SmallLds {
    static constexpr int NZ = Fn::NX + Fn::NU, NY = Fn::NY, JN = NY * NZ;
    static constexpr size_t a16(size_t b) { return (b + 15) / 16 * 16; }
    static constexpr size_t J_BYTES = a16((size_t)TILE * JN * sizeof(T)), Y_BYTES = a16((size_t)TILE * NY * sizeof(T)), Z_BYTES = a16((size_t)TILE * NZ * sizeof(T));
    static constexpr size_t WAVE_BYTES = J_BYTES + Y_BYTES + Z_BYTES;
    static constexpr int WAVES = 4 * WAVE_BYTES <= 160 * 1024 ? 4 : 2 * WAVE_BYTES <= 160 * 1024 ? 2 : 1;     // independent waves per workgroup (block size = 64 WAVES)
    static constexpr size_t BYTES = WAVES * WAVE_BYTES;
    static constexpr int JU = (int)((size_t)TILE * JN * sizeof(T) / 16), YU = (int)((size_t)TILE * NY * sizeof(T) / 16);     // whole 16-byte units of a full tile
    static_assert(((size_t)TILE * JN * sizeof(T)) % 16 == 0 && ((size_t)TILE * NY * sizeof(T)) % 16 == 0, "64-knot slabs are multiples of 16 bytes");
};
// the walk of ONE wave: tiles tile, tile + tstep, ... of the descriptor, `base` = the wave's private LDS region (SmallLds::WAVE_BYTES).  COH: the tile
// inputs were written by other workgroups of the same launch (mixed.hpp): agent-scope loads, as in jac_split_loop
template <class Fn, class T, class BT, bool COH = false>
RD_DEV void jac_ad_small_walk(const Fn& fn, const BT& bt, long long tile, const long long tstep, unsigned char* base) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, JN = NY * NZ;
    using L = SmallLds<Fn, T>;
    using D = Dual<T, NZ>;
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    const int lane = threadIdx.x & 63;
    T* jt = reinterpret_cast<T*>(base);
    T* yt = reinterpret_cast<T*>(base + L::J_BYTES);
    T* zs = reinterpret_cast<T*>(base + L::J_BYTES + L::Y_BYTES);
    const long long ntiles = bt.ntiles();
    if (tile >= ntiles) return;
    constexpr int ZPT = (TILE * NZ + TILE - 1) / TILE;        // = NZ
    TileRef<T> cur = at_uniform<T>(bt, tile);
    T zpre[ZPT]; double tpre, hpre;
    auto request = [&](const TileRef<T>& r, bool live) {       // the tile's inputs -> registers (lane-contiguous; lanes past the end re-read element 0)
        const int nkr = live ? (int)((r.N - r.k0) < TILE ? (r.N - r.k0) : TILE) : 0;
        const T* __restrict__ Zr = r.Z + r.k0 * NZ;
#pragma unroll
        for (int i = 0; i < ZPT; ++i) { const int e = i * TILE + lane; zpre[i] = tile_load<COH>(Zr + (e < nkr * NZ ? e : 0)); }
        const int kl = lane < nkr ? lane : (nkr > 0 ? nkr - 1 : 0);
        tpre = r.t[r.k0 + kl]; hpre = r.dt[r.k0 + kl];
    };
    request(cur, true);
    for (; tile < ntiles; tile += tstep) {
        const long long k0 = cur.k0;
        const int nk = (int)((cur.N - k0) < TILE ? (cur.N - k0) : TILE);
        const int kl = lane < nk ? lane : nk - 1;
        // this tile's inputs: registers -> LDS -> one knot point per lane
#pragma unroll
        for (int i = 0; i < ZPT; ++i) { const int e = i * TILE + lane; if (e < TILE * NZ) zs[e] = zpre[i]; }
        const T t = (T)tpre, h = (T)hpre;
        const bool term = is_terminal_dt(hpre);
        wave_sync();
        D xd[NX], ud[NU], yd[NY];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            xd[i].v = zs[kl * NZ + i];
#pragma unroll
            for (int j = 0; j < NZ; ++j) xd[i].d[j] = (i == j) ? T(1) : T(0);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            ud[i].v = term ? T(0) : zs[kl * NZ + NX + i];
#pragma unroll
            for (int j = 0; j < NZ; ++j) ud[i].d[j] = (NX + i == j) ? T(1) : T(0);
        }
        wave_sync();
        // the next tile's inputs are requested now and consumed at the top of the next iteration
        const long long nxt = tile + tstep;
        const bool more = nxt < ntiles;
        const TileRef<T> nx = at_uniform<T>(bt, more ? nxt : tile);
        request(nx, more);
        __builtin_amdgcn_sched_barrier(0);
        fn.eval(xd, ud, t, h, yd);
        T* row = jt + lane * JN;
#pragma unroll
        for (int c = 0; c < NZ; ++c) {
            const T sc = (c >= NX && term) ? T(0) : T(1);       // control columns of a terminal knot point are zero
#pragma unroll
            for (int i = 0; i < NY; ++i) row[c * NY + i] = yd[i].d[c] * sc;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) yt[lane * NY + i] = yd[i].v;
        wave_sync();
        // copy-out: the tile slabs J[k0 ..] and Y[k0 ..] are contiguous
        T* Jt = cur.J + k0 * JN;
        T* Yt = cur.Y ? cur.Y + k0 * NY : nullptr;
        const bool al16 = ((reinterpret_cast<unsigned long long>(Jt) | reinterpret_cast<unsigned long long>(Yt)) & 15) == 0;
        if (al16) {
            const int jbytes = nk * JN * (int)sizeof(T), ybytes = nk * NY * (int)sizeof(T);
            const V4* jsrc = reinterpret_cast<const V4*>(jt);
            V4* jdst = reinterpret_cast<V4*>(Jt);
#pragma unroll
            for (int r = 0; r < (L::JU + TILE - 1) / TILE; ++r) {
                const int u = r * TILE + lane;
                if (u * 16 + 16 <= jbytes) RD_TILE_STORE(jdst + u, jsrc[u]);
                else if (u * 16 < jbytes) { for (int q = 0; q < (int)(16 / sizeof(T)); ++q) if (u * 16 + (q + 1) * (int)sizeof(T) <= jbytes) Jt[u * (16 / sizeof(T)) + q] = jt[u * (16 / sizeof(T)) + q]; }
            }
            if (Yt) {
                const V4* ysrc = reinterpret_cast<const V4*>(yt);
                V4* ydst = reinterpret_cast<V4*>(Yt);
#pragma unroll
                for (int r = 0; r < (L::YU + TILE - 1) / TILE; ++r) {
                    const int u = r * TILE + lane;
                    if (u * 16 + 16 <= ybytes) RD_TILE_STORE(ydst + u, ysrc[u]);
                    else if (u * 16 < ybytes) { for (int q = 0; q < (int)(16 / sizeof(T)); ++q) if (u * 16 + (q + 1) * (int)sizeof(T) <= ybytes) Yt[u * (16 / sizeof(T)) + q] = yt[u * (16 / sizeof(T)) + q]; }
                }
            }
        } else {                                                // slabs that are only sizeof(T)-aligned: element by element
            for (int e = lane; e < nk * JN; e += TILE) Jt[e] = jt[e];
            if (Yt) for (int e = lane; e < nk * NY; e += TILE) Yt[e] = yt[e];
        }
        wave_sync();
        cur = nx;
    }
}
template <class Fn, class T, class BT>
RD_DEV void jac_ad_small_body(const Fn& fn, const BT& bt) {
    using L = SmallLds<Fn, T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    jac_ad_small_walk<Fn, T, BT>(fn, bt, (long long)blockIdx.x * L::WAVES + wave, (long long)gridDim.x * L::WAVES, smem + wave * L::WAVE_BYTES);
}
template <class Fn, class T>
__global__ void __launch_bounds__(4 * TILE) jac_ad_small_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_ad_small_body<Fn, T>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}
template <class Fn, class T>
__global__ void __launch_bounds__(4 * TILE) jac_ad_small_multi_kernel(Fn fn, MultiBatch<T> mb) {     // rdhip_jacobian_multi
    jac_ad_small_body<Fn, T>(fn, mb);
}

// ---- ForwardAD, run-time column loop (compact code for the long tail of model x rule combinations) -----------
template <class Fn, class T, class Plan> struct AdLoopBody {
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG;
        static_assert(Plan::NCHUNK == 1, "run-time column loops use the single-chunk plan");
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
            if (c == 0 && fr.want_y) {
#pragma unroll
                for (int i = 0; i < NY; ++i) col[i] = yd[i].v;
                fr.put_value(NZ, col);
            }
            const T sc = (c >= NX && fr.term) ? T(0) : T(1);
#pragma unroll
            for (int i = 0; i < NY; ++i) col[i] = yd[i].d[0] * sc;
            fr.put_at(c, col);
        }
    }
};
template <class Fn, class T, class Plan, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_ad_loop_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                    const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, AdLoopBody<Fn, T, Plan>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, J, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, AdLoopBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_ad_loop_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                     const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, AdLoopBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

// ---- UserDefined: the integrators' chain rule over continuous stage Jacobians -----------------------------------------
// src/integration.jl:85-93 (Euler), :149-177 (RK3), :302-337 (RK4); RK2 = explicit midpoint, A_d = I + A2 (I + A1 h/2) h.
//   dA1 = A1 h ; dA2 = A2 (I + dA1/2) h ; dA3 = A3 (I + dA2/2) h ; dA4 = A4 (I + dA3) h ; A_d = I + (dA1 + 2dA2 + 2dA3 + dA4)/6
//   dB1 = B1 h ; dB2 = B2 h + A2 dB1 h/2 ; ...                                       B_d =     (dB1 + 2dB2 + 2dB3 + dB4)/6
// evaluated one column at a time: every product A_i w + B_i w_u is a Jacobian-vector product of the CONTINUOUS model at the
// stage point x_i (the k_i are computed once per knot point exactly as the reference does), taken matrix-free with a forward
// dual.  The reference's analytic RigidBody Jacobian (src/rigidbody.jl:260-313) is not executable as written;
// its blocks equal these products.  Models without a user-defined Jacobian in the reference return NOT_IMPLEMENTED.
template <class Fn, class T, class Plan> struct ChainBody {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG, RULE = Fn::RULE_ID;
    using D = Dual<T, 1>;
    // (A w + B wu) at stage point xs
    static RD_DEV void jvp(const Fn& fn, const T (&xs)[NX], const T (&u)[NU], T ts, const T (&w)[NX], const T (&wu)[NU], T (&o)[NX]) {
        D xd[NX], ud[NU], fd[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xd[i].v = xs[i]; xd[i].d[0] = w[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { ud[i].v = u[i]; ud[i].d[0] = wu[i]; }
        model_f(fn.m, xd, ud, ts, fd);
#pragma unroll
        for (int i = 0; i < NX; ++i) o[i] = fd[i].d[0];
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        static_assert(Plan::NCHUNK == 1, "run-time column loops use the single-chunk plan");
        const T h = fr.h;
        // stage points (StaticReturn form of the reference: k_i = f(.) h)
        T x2[NX], x3[NX], x4[NX], k1[NX], k2[NX], k3[NX];
        if constexpr (RULE != RULE_CONTINUOUS && RULE != RULE_EULER) {
            model_f(fn.m, fr.x, fr.u, fr.t, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k1[i] = k1[i] * h; x2[i] = fr.x[i] + k1[i] * T(0.5); }
            if constexpr (RULE == RULE_RK3 || RULE == RULE_RK4) {
                model_f(fn.m, x2, fr.u, fr.t + h * T(0.5), k2);
#pragma unroll
                for (int i = 0; i < NX; ++i) { k2[i] = k2[i] * h; x3[i] = (RULE == RULE_RK3) ? fr.x[i] - k1[i] + k2[i] * T(2) : fr.x[i] + k2[i] * T(0.5); }
            }
            if constexpr (RULE == RULE_RK4) {
                model_f(fn.m, x3, fr.u, fr.t + h * T(0.5), k3);
#pragma unroll
                for (int i = 0; i < NX; ++i) x4[i] = fr.x[i] + k3[i] * h;
            }
        }
        if (fr.g == 0 && fr.want_y) {
            T y[NY]; fn.eval(fr.x, fr.u, fr.t, fr.h, y);
            fr.put_value(NZ, y);
        }
        const int c0 = fr.g * NZ / NG, c1 = (fr.g + 1) * NZ / NG;
        for (int c = c0; c < c1; ++c) {
            T w[NX], wu[NU], d1[NX], d2[NX], d3[NX], d4[NX], ws[NX], col[NY];
#pragma unroll
            for (int i = 0; i < NX; ++i) w[i] = (i == c) ? T(1) : T(0);
#pragma unroll
            // terminal knot point: the control columns of a DISCRETE rule are zero (h = 0).  The continuous rule hands the user's own Jacobian back as the
            // reference does -- jacobian!(fun, J, y, state(z), control(z)) with control(z) = zeros (src/functionbase.jl:252-257, src/knotpoint.jl:67):
            // d f / d u at u = 0, NOT zeroed (only ForwardAD / FiniteDifference differentiate through `!is_terminal(z) * u`)
            for (int i = 0; i < NU; ++i) wu[i] = (NX + i == c && (!fr.term || RULE == RULE_CONTINUOUS)) ? T(1) : T(0);
            jvp(fn, fr.x, fr.u, fr.t, w, wu, d1);
            if constexpr (RULE == RULE_CONTINUOUS) {
#pragma unroll
                for (int i = 0; i < NX; ++i) col[i] = d1[i];
            } else if constexpr (RULE == RULE_EULER) {
#pragma unroll
                for (int i = 0; i < NX; ++i) col[i] = w[i] + d1[i] * h;
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) { d1[i] = d1[i] * h; ws[i] = w[i] + d1[i] * T(0.5); }
                jvp(fn, x2, fr.u, fr.t + h * T(0.5), ws, wu, d2);
                if constexpr (RULE == RULE_RK2) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + d2[i] * h;
                } else if constexpr (RULE == RULE_RK3) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; ws[i] = w[i] - d1[i] + d2[i] * T(2); }
                    jvp(fn, x3, fr.u, fr.t + h, ws, wu, d3);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(4) + d3[i] * h) / T(6);
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; ws[i] = w[i] + d2[i] * T(0.5); }
                    jvp(fn, x3, fr.u, fr.t + h * T(0.5), ws, wu, d3);
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d3[i] = d3[i] * h; ws[i] = w[i] + d3[i]; }
                    jvp(fn, x4, fr.u, fr.t + h, ws, wu, d4);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(2) + d3[i] * T(2) + d4[i] * h) / T(6);
                }
            }
            fr.put_at(c, col);
        }
    }
};
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_chain_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                   const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, ChainBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

// compile-time-column form of the chain rule for the split-role kernel: the direction w = e_C is a constant, so the stage
// Jacobian-vector products fold exactly like the ForwardAD columns do
template <class Fn, class T, class Plan, int CH, int G, int C, bool VALUE_PENDING> struct ChainCols {
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr, const T (&x2)[Fn::NX], const T (&x3)[Fn::NX], const T (&x4)[Fn::NX]) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY, RULE = Fn::RULE_ID;
        constexpr int C0 = Plan::chunk_begin(CH), C1 = Plan::chunk_end(CH);
        using CB = ChainBody<Fn, T, Plan>;
        if constexpr (C < C1) {
            if constexpr (FirstSplitOf<Plan>::N > 0 && C == FirstSplitOf<Plan>::N) {
                if (fr.first) __syncthreads();              // barrier E of the split-role loop: the early part of the first tile is complete
            }
            if constexpr (Plan::wave_of(C) == G) {
                if constexpr (VALUE_PENDING) {
                    if (fr.want_y) { T y[NY]; fn.eval(fr.x, fr.u, fr.t, fr.h, y); fr.put_value(C1 - C0, y); }
                }
                const T h = fr.h;
                T w[NX], wu[NU], d1[NX], d2[NX], d3[NX], d4[NX], ws[NX], col[NY];
#pragma unroll
                for (int i = 0; i < NX; ++i) w[i] = (i == C) ? T(1) : T(0);
#pragma unroll
                for (int i = 0; i < NU; ++i) wu[i] = (NX + i == C) ? ((fr.term && RULE != RULE_CONTINUOUS) ? T(0) : T(1)) : T(0);      // (see ChainBody: the continuous rule keeps d f / d u at a terminal knot)
                CB::jvp(fn, fr.x, fr.u, fr.t, w, wu, d1);
                if constexpr (RULE == RULE_CONTINUOUS) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = d1[i];
                } else if constexpr (RULE == RULE_EULER) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = w[i] + d1[i] * h;
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { d1[i] = d1[i] * h; ws[i] = w[i] + d1[i] * T(0.5); }
                    CB::jvp(fn, x2, fr.u, fr.t + h * T(0.5), ws, wu, d2);
                    if constexpr (RULE == RULE_RK2) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) col[i] = w[i] + d2[i] * h;
                    } else if constexpr (RULE == RULE_RK3) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; ws[i] = w[i] - d1[i] + d2[i] * T(2); }
                        CB::jvp(fn, x3, fr.u, fr.t + h, ws, wu, d3);
#pragma unroll
                        for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(4) + d3[i] * h) / T(6);
                    } else {
#pragma unroll
                        for (int i = 0; i < NX; ++i) { d2[i] = d2[i] * h; ws[i] = w[i] + d2[i] * T(0.5); }
                        CB::jvp(fn, x3, fr.u, fr.t + h * T(0.5), ws, wu, d3);
#pragma unroll
                        for (int i = 0; i < NX; ++i) { d3[i] = d3[i] * h; ws[i] = w[i] + d3[i]; }
                        CB::jvp(fn, x4, fr.u, fr.t + h, ws, wu, d4);
#pragma unroll
                        for (int i = 0; i < NX; ++i) col[i] = w[i] + (d1[i] + d2[i] * T(2) + d3[i] * T(2) + d4[i] * h) / T(6);
                    }
                }
                fr.put_at(C - C0, col);
                ChainCols<Fn, T, Plan, CH, G, C + 1, false>::run(fn, fr, x2, x3, x4);
            } else {
                ChainCols<Fn, T, Plan, CH, G, C + 1, VALUE_PENDING>::run(fn, fr, x2, x3, x4);
            }
        }
    }
};
template <class Fn, class T, class Plan> struct ChainUnrolledBody {
    static constexpr bool EARLY = true;
    static constexpr bool COPY_PREFETCH = CopyPrefetchChain<decltype(Fn::m), T>::value;       // (CopyPrefetchOf; the models: fam_rigid.hpp)
    static constexpr int COPY_PACE = CopyPrefetchChain<decltype(Fn::m), T>::PACE;
    template <int CH, int G> static RD_DEV void waves(const Fn& fn, JacFrame<Fn, T>& fr, const T (&x2)[Fn::NX], const T (&x3)[Fn::NX], const T (&x4)[Fn::NX]) {
        if constexpr (G < Plan::NG) {
            if (fr.g == G) ChainCols<Fn, T, Plan, CH, G, Plan::chunk_begin(CH), (CH == Plan::value_chunk && G == Plan::value_wave)>::run(fn, fr, x2, x3, x4);
            else waves<CH, G + 1>(fn, fr, x2, x3, x4);
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NX = Fn::NX, RULE = Fn::RULE_ID;
        const T h = fr.h;
        T x2[NX], x3[NX], x4[NX], k1[NX], k2[NX], k3[NX];     // stage points, k_i = f(.) h as in the reference
#pragma unroll
        for (int i = 0; i < NX; ++i) x2[i] = x3[i] = x4[i] = fr.x[i];
        if constexpr (RULE != RULE_CONTINUOUS && RULE != RULE_EULER) {
            model_f(fn.m, fr.x, fr.u, fr.t, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k1[i] = k1[i] * h; x2[i] = fr.x[i] + k1[i] * T(0.5); }
            if constexpr (RULE == RULE_RK3 || RULE == RULE_RK4) {
                model_f(fn.m, x2, fr.u, fr.t + h * T(0.5), k2);
#pragma unroll
                for (int i = 0; i < NX; ++i) { k2[i] = k2[i] * h; x3[i] = (RULE == RULE_RK3) ? fr.x[i] - k1[i] + k2[i] * T(2) : fr.x[i] + k2[i] * T(0.5); }
            }
            if constexpr (RULE == RULE_RK4) {
                model_f(fn.m, x3, fr.u, fr.t + h * T(0.5), k3);
#pragma unroll
                for (int i = 0; i < NX; ++i) x4[i] = fr.x[i] + k3[i] * h;
            }
        }
        waves<CH, 0>(fn, fr, x2, x3, x4);
    }
};
template <class Fn, class T, class Plan, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_chain_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                  const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, ChainUnrolledBody<Fn, T, Plan>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, J, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, ChainUnrolledBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

// ---- FiniteDifference: forward differences, eps_j = max(sqrt(eps)|z_j|, sqrt(eps)) (FiniteDiff.jl default) -----
template <class T> RD_DEV T fd_step(T zc) {             // eps_j = max(sqrt(eps) |z_j|, sqrt(eps))
    const T se = sqrt_eps<T>();
    const T az = zc < T(0) ? -zc : zc;
    return (se * az > se) ? se * az : se;
}
template <class Fn, class T, class Plan> struct FdBody {
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG;
        static_assert(Plan::NCHUNK == 1, "run-time column loops use the single-chunk plan");
        T f0[NY];
        fn.eval(fr.x, fr.u, fr.t, fr.h, f0);
        if (fr.g == 0 && fr.want_y) fr.put_value(NZ, f0);
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
            const T inv = T(1) / eps;                         // one division per column (the quotient differs from (f1-f0)/eps by <= 1 ulp)
#pragma unroll
            for (int i = 0; i < NY; ++i) col[i] = (f1[i] - f0[i]) * inv;
            fr.put_at(c, col);
        }
    }
};
template <class Fn, class T, class Plan, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_fd_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                               const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, FdBody<Fn, T, Plan>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, J, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, FdBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__(Plan::NG * TILE) jac_fd_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_tile_loop<Fn, T, Plan, FdBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

}  // namespace rd
