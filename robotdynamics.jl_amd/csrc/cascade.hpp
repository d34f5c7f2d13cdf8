// cascade.hpp -- rollout! (src/trajectories.jl:436-441) for models that declare a Cascade<> block structure (models.hpp: the rigid bodies).
//
// What bounds a rollout on this machine (measured with the RD_ROLL_ABLATE switches, tune/tune_rollout.hip): NOT the arithmetic.  One lane per
// trajectory means every global access of a wavefront touches 64 different cache lines; such an instruction occupies the CU's address
// path for ~60-100 cycles, a step has 11-16 of them, and that alone is 0.6-0.9 us per step (the kernel with the arithmetic removed runs as
// slowly as with it).  Behind that sits the latency chain: step k+1 needs step k, one fp64 wavefront issues a dependent instruction every
// ~7 cycles, 300 instructions per satellite RK4 step.  Both are attacked:
//
//  * IO wavefront.  The workgroup that owns 64 trajectories has one extra wavefront that does ALL global traffic, in chunks of S steps:
//    it copies the chunk's controls / times into LDS and the chunk's new states out of LDS with group-contiguous accesses (8 or 16 lanes
//    per knot point, the knot points of an instruction adjacent in memory: ~6-10 cache lines per instruction instead of 64; every address
//    is "uniform base + per-lane constant", no per-access address arithmetic), double-buffered, one chunk ahead of / behind the compute
//    wavefronts.  The compute wavefronts never issue a global instruction inside the loop.
//  * Cascade.  The chain is shortened by cutting the STATE, not the batch: one compute wavefront per cascade block (w | q | r, v), each on
//    its own SIMD, each integrating only its block -- the model's dynamics function is instantiated per block and everything the block's
//    derivative does not need folds away (the w wave runs Euler's equation alone: 81 fp64 instructions per RK4 step instead of 307).
//    Block b needs the STAGE POINTS of blocks < b: a wave publishes the stage points of its step k in an LDS ring (slot k mod R,
//    [stage][component][lane], one flag word per wave counting finished steps) and the downstream waves pick them up one step later --
//    dependencies only run downstream, so the waves form a pipeline skewed by one step per block and nobody waits on a round trip.
//
// LDS operations of one wave execute in order, so "data, then flag" on the producer and "flag, then data" on the consumer need no fence
// that drains the queue; a consumer issues next step's flag + data reads BEFORE this step's arithmetic and only checks the flag afterwards
// (re-reading in the rare case it was early), so the LDS latency is off the chain.  A producer may run at most R steps ahead of the last
// wave (back-pressure word `cons`).  Steps that do not fill a chunk (K - 1 mod S) and a workgroup whose 64 trajectories reach past the
// batch take the direct path: the compute waves load and store their own rows (the one-lane-per-trajectory access pattern).
#pragma once
#include "kernels.hpp"

namespace rd {

constexpr int CASCADE_COMPACT = 64;   // flag in the ring-depth template argument, see CascadeLds
constexpr int CASCADE_CHUNK = 4;      // S: steps per IO chunk (even: the two upstream-stage buffers rotate inside a chunk)

template <class Fn, class Cas, int R_> struct CascadeLds {
    using T = typename Fn::T;
    // R_ = ring depth (+ CASCADE_COMPACT: the ring always [stage][component][lane] -- 28 instead of 48 KB for an fp32 rigid body, which is what lets
    // TWO workgroups of the fused rollout -> Jacobians kernel share a compute unit, fused.hpp; the hand-off costs 3 % of the chain: 123 against 119.5 us
    // for 4 096 x 256 satellite steps)
    static constexpr int NX = Fn::NX, NU = Fn::NU, NB = Cas::NB, NS = rule_stages<Fn::RULE_ID>(), R = R_ & 63, S = CASCADE_CHUNK;
    static constexpr bool COMPACT = (R_ & 64) != 0;
    static constexpr bool published(int i) { return Cas::block_of(i) < NB - 1; }
    static constexpr int pub_index(int i) { int p = 0; for (int j = 0; j < i; ++j) p += published(j) ? 1 : 0; return p; }
    static constexpr int NPUB = pub_index(NX);
    static constexpr size_t a16(size_t b) { return (b + 15) / 16 * 16; }
    static constexpr int NUP = NU | 1, NXP = NX | 1;                         // odd row strides: lane-strided reads / writes hit distinct banks
    // ring slot layout [stage][lane][NPP]: the published components of one lane are ADJACENT, so a block's stage point moves as one or two 16-byte LDS
    // instructions instead of one instruction per component (the hand-off is LDS-instruction-issue bound: 14 / 17 / 28 cycles per b32 / b64 / b128
    // instruction, measured); the lane stride is an odd number of 16-byte units, which keeps 16 consecutive lanes of a b128 access on distinct banks
    // (where the padded layout does not fit the CU's LDS next to the IO tiles -- a rigid body in fp64 -- the ring stays [stage][component][lane])
    static constexpr int npp() { int u = (int)((NPUB * sizeof(T) + 15) / 16); if (u % 2 == 0) ++u; return u * 16 / (int)sizeof(T); }
    static constexpr size_t a16_(size_t b) { return (b + 15) / 16 * 16; }
    static constexpr size_t bytes_with(size_t ring_slot) {
        return a16_(R * ring_slot * sizeof(T)) + a16_(2 * (size_t)S * TILE * (NU | 1) * sizeof(T)) + 2 * (size_t)2 * S * TILE * sizeof(double) + a16_(2 * (size_t)S * TILE * (NX | 1) * sizeof(T)) + 64;
    }
    static constexpr bool LANE_MAJOR = NPUB > 0 && !COMPACT && bytes_with((size_t)NS * TILE * npp()) <= 160 * 1024;
    static constexpr int NPP = LANE_MAJOR ? npp() : NPUB;
    static constexpr size_t RING_SLOT = (size_t)NS * TILE * NPP;             // elements per ring slot
    // element index of (stage s, published component p, lane l) inside a slot
    static constexpr size_t ring_index(int s, int p, int l) { return LANE_MAJOR ? ((size_t)s * TILE + l) * NPP + p : ((size_t)s * NPUB + p) * TILE + l; }
    static constexpr size_t lane_offset(int l) { return LANE_MAJOR ? (size_t)l * NPP : (size_t)l; }
    static constexpr size_t IN_ELEMS = (size_t)S * TILE * NUP, OUT_ELEMS = (size_t)S * TILE * NXP, TS_ELEMS = (size_t)2 * S * TILE;
    static constexpr size_t IN_OFF = a16(R * RING_SLOT * sizeof(T));         // IN[2]: controls of a chunk, [s][trajectory][NUP]
    static constexpr size_t TS_OFF = IN_OFF + a16(2 * IN_ELEMS * sizeof(T)); // TS[2]: t[s][trajectory], dt[s][trajectory] (fp64)
    static constexpr size_t OUT_OFF = TS_OFF + 2 * TS_ELEMS * sizeof(double);// OUT[2]: new states of a chunk, [s][trajectory][NXP]
    static constexpr size_t FLAG_OFF = OUT_OFF + a16(2 * OUT_ELEMS * sizeof(T));
    static constexpr size_t BYTES = FLAG_OFF + 64;                           // prod[NB], cons, in_ready, out_free
    static_assert((R & (R - 1)) == 0, "ring depth must be a power of two");
    static_assert(S == 4 && NU <= 8 && NX <= 16, "IO lane groups: 8 lanes per control row, 16 per state row, 4 steps per chunk");
};
template <class Fn, class Cas, int B, int RC = 4> struct CascadeHook {
    using T = typename Fn::T;
    static constexpr int NX = Fn::NX, NS = rule_stages<Fn::RULE_ID>();
    static constexpr int NB = Cas::NB;
    static constexpr bool published(int i) { return Cas::block_of(i) < NB - 1; }
    static constexpr int pub_index(int i) { int p = 0; for (int j = 0; j < i; ++j) p += published(j) ? 1 : 0; return p; }
    static constexpr int NPUB = pub_index(NX);
    const T (&U)[NS][NX];                             // stage points of the upstream blocks (entries of other blocks are never touched)
    using LL = CascadeLds<Fn, Cas, RC>;
    T __attribute__((address_space(3)))* slot;        // ring slot of this step, already offset by the lane (CascadeLds::ring_index / lane_offset)
    // this block's stage points go to the ring AS THEY ARE FORMED: the LDS writes (~29 cycles of LDS pipe per 16 B/lane instruction, measured)
    // then drain under the remaining arithmetic instead of queueing up behind it
    template <int S> RD_DEV void at(T (&xs)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (Cas::block_of(i) < B) xs[i] = U[S][i];
            else if (Cas::block_of(i) == B) { if constexpr (B < NB - 1) slot[LL::ring_index(S, pub_index(i), 0)] = xs[i]; }
            else xs[i] = T(0);                              // downstream components: this block's derivative does not read them
        }
    }
};
// explicit LDS address space (volatile accesses through a generic pointer would stay FLAT instructions).  Flags are volatile; the data are
// plain LDS accesses fenced by compiler barriers (RD_LDS_ORDER) -- the hardware already serves a wave's LDS operations in order --, so reads of
// components a block does not use disappear and neighbours pair into ds_read2 / ds_write2
#define RD_LDS_ORDER() asm volatile("" ::: "memory")
// lgkmcnt(0) inside the slow path of a poll: the loop re-issues reads into registers the fast path loaded long ago; if they could still be
// pending at the join, every later use would have to wait as if they had just been issued (the compiler's counters merge conservatively)
#define RD_LDS_DRAIN() __builtin_amdgcn_s_waitcnt(0xC07F)
#define RD_LDS_T(TYPE) TYPE __attribute__((address_space(3)))*
RD_DEV int lds_flag(volatile int __attribute__((address_space(3)))* f) { return __builtin_amdgcn_readfirstlane(*f); }
// PUB (the fused rollout -> Jacobians kernel, fused.hpp): the simulated states are read by OTHER workgroups of the same launch, possibly on another XCD
// (each XCD has its own L2).  They are stored as agent-scope atomics -- written through to memory -- and announced through a 64-bit progress word per
// 64-trajectory group, (call epoch << 32) | knots available, stored once the data stores it covers have been acknowledged (vmcnt).
template <bool PUB, class T> RD_DEV void pub_store(T* p, T v) {
    if constexpr (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
RD_DEV void pub_progress(unsigned long long* word, unsigned long long base, long long knots) {
    if (word) __hip_atomic_store(word, base | (unsigned long long)knots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- compute wavefront of block B -------------------------------------------------------------------------------------------------------
template <class Fn, class T, class Cas, int R, int B, bool PUB>
RD_DEV void cascade_compute_wave_body(const Fn& fn, long long K, long long M, long long sj, long long sk, T* __restrict__ Z, const double* __restrict__ tt,
                                      const double* __restrict__ dtt, const T* __restrict__ X0, int nfull) {
    using L = CascadeLds<Fn, Cas, R>;
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NS = L::NS, NB = Cas::NB, S = L::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // (declared here, not passed in: the accesses must stay LDS instructions)
    typedef RD_LDS_T(T) LdsT;
    typedef RD_LDS_T(double) LdsD;
    typedef RD_LDS_T(volatile int) LdsI;
    LdsT ring = (LdsT)smem;
    LdsI prod = (LdsI)(smem + L::FLAG_OFF);          // prod[b]: steps finished (stage points published, new state in OUT) by wave b
    LdsI cons = prod + NB;                           // steps whose upstream data the LAST wave holds in registers
    LdsI in_ready = prod + NB + 1;                   // chunks whose inputs the IO wave has put into IN / TS
    LdsI out_free = prod + NB + 2;                   // chunks whose states the IO wave has taken out of OUT
    const int lane = threadIdx.x & 63;
    // lanes past the batch redo the last trajectory and store the same values to the same addresses: no exec-masked stores (a conditional
    // store makes the compiler's vmcnt bookkeeping conservative: every step would wait for the previous step's stores)
    const long long j0 = (long long)blockIdx.x * TILE + lane;
    const long long j = j0 < M ? j0 : M - 1;
    T x[NX], y[NX];
    const long long ip = j * sj;
    {
        const T* xs = X0 ? X0 + j * NX : Z + ip * NZ;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = (Cas::block_of(i) == B) ? xs[i] : T(0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) if (Cas::block_of(i) == B) pub_store<PUB>(Z + ip * NZ + i, x[i]);
    if constexpr (PUB) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): knot 0 is announced together with the first chunk (cascade_io_wave)
    if (K < 2) return;
    auto read_slot = [&](int step, T (&dst)[NS][NX]) {
        LdsT slot = ring + (size_t)(step & (L::R - 1)) * L::RING_SLOT;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (Cas::block_of(i) < B) dst[s][i] = slot[L::ring_index(s, L::pub_index(i), 0) + L::lane_offset(lane)];
        RD_LDS_ORDER();
    };
    T UA[NS][NX], UB[NS][NX];                                 // upstream stage points of this / the next step (roles alternate)
    int cfc = 0;                                              // `cons` as last read (one step old: it only grows, an old value errs on the safe side)
    if constexpr (B > 0) {                                    // upstream stage points of step 0
        while (lds_flag(prod + (B - 1)) < 1) __builtin_amdgcn_s_sleep(1);
        RD_LDS_ORDER();
        read_slot(0, UA);
        if constexpr (B == NB - 1) *cons = 1;
    }
    const int nsteps = (int)(K - 1);
    // step k: x_k -> x_{k+1} with inputs (u, t, h); upstream stage points of step k in Uc, those of step k+1 are fetched into Un.
    // `emit(i, value)` disposes of the new state's component i (LDS OUT tile or global memory).
    auto step = [&](int k, const T (&uk)[NU], double h, T tk, const T (&Uc)[NS][NX], T (&Un)[NS][NX], auto&& emit) {
        T u[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(h) ? T(0) : uk[i];
        // next step's upstream data, requested now and checked after the arithmetic (flag first: the LDS serves a wave in order)
        const bool more = k + 1 < nsteps;
        int fl = 0;
        if constexpr (B < NB - 1) {
            // ring slot k mod R is free once the last wave holds step k - R: checked with the value of `cons` requested a step ago
            if (__builtin_amdgcn_readfirstlane(cfc) < k - L::R + 1) {
                do { __builtin_amdgcn_s_sleep(1); cfc = *cons; RD_LDS_DRAIN(); } while (__builtin_amdgcn_readfirstlane(cfc) < k - L::R + 1);
            }
            RD_LDS_ORDER();
        }
        if constexpr (B > 0) { if (more) { fl = prod[B - 1]; RD_LDS_ORDER(); read_slot(k + 1, Un); } }
        if constexpr (B < NB - 1) cfc = *cons;
        __builtin_amdgcn_sched_barrier(0);                    // keep the requests above the arithmetic
        CascadeHook<Fn, Cas, B, R> hk{Uc, ring + (size_t)(k & (L::R - 1)) * L::RING_SLOT + L::lane_offset(lane)};
        if constexpr (Fn::RULE_ID == RULE_IMPLICIT_MIDPOINT) fn.template block_step<Cas, B>(x, u, tk, (T)h, y, hk);      // implicit.hpp ImplicitRollFun
        else rk_hooked<Fn::RULE_ID>(fn.m, x, u, tk, (T)h, y, hk);
        // next step's upstream flag is LOOKED AT here, before the last writes -- lgkmcnt is a 4-bit counter: behind ~20 LDS instructions the
        // wait for an old read degenerates into waiting for the writes to drain -- but a wave that has to wait for its upstream publishes
        // its own step first: the spin loop sits behind the writes
        bool early = false;
        if constexpr (B > 0) { if (more) early = __builtin_amdgcn_readfirstlane(fl) < k + 2; }
#pragma unroll
        for (int i = 0; i < NX; ++i) if (Cas::block_of(i) == B) { x[i] = y[i]; emit(i, y[i]); }
        RD_LDS_ORDER();
        prod[B] = k + 1;
        if constexpr (B > 0) {
            if (more) {
                if (early) {
                    do { __builtin_amdgcn_s_sleep(1); fl = prod[B - 1]; RD_LDS_ORDER(); read_slot(k + 1, Un); RD_LDS_DRAIN(); } while (__builtin_amdgcn_readfirstlane(fl) < k + 2);
                }
                if constexpr (B == NB - 1) *cons = k + 2;
            }
        }
    };
    // ---- full chunks: inputs from the IN / TS tiles, new states into the OUT tile (the IO wave moves them) ----
    int k = 0;
    for (int c = 0; c < nfull; ++c, k += S) {
        {
            int f_in = *in_ready, f_out = *out_free;            // both flags in one round trip
            if (__builtin_amdgcn_readfirstlane(f_in) < c + 1 || __builtin_amdgcn_readfirstlane(f_out) < c - 1) {
                do { __builtin_amdgcn_s_sleep(1); f_in = *in_ready; f_out = *out_free; RD_LDS_DRAIN(); }
                while (__builtin_amdgcn_readfirstlane(f_in) < c + 1 || __builtin_amdgcn_readfirstlane(f_out) < c - 1);
            }
        }
        RD_LDS_ORDER();
        LdsT in = (LdsT)(smem + L::IN_OFF) + (size_t)(c & 1) * L::IN_ELEMS;
        LdsD ts = (LdsD)(smem + L::TS_OFF) + (size_t)(c & 1) * L::TS_ELEMS;
        LdsT out = (LdsT)(smem + L::OUT_OFF) + (size_t)(c & 1) * L::OUT_ELEMS;
        T uin[S][NU]; double hh[S], tk[S];
#pragma unroll
        for (int d = 0; d < S; ++d) {
            tk[d] = ts[d * TILE + lane]; hh[d] = ts[(S + d) * TILE + lane];
#pragma unroll
            for (int i = 0; i < NU; ++i) uin[d][i] = in[(d * TILE + lane) * L::NUP + i];
        }
        RD_LDS_ORDER();
#pragma unroll
        for (int d = 0; d < S; d += 2) {
            step(k + d, uin[d], hh[d], (T)tk[d], UA, UB, [&](int i, T v) { out[(d * TILE + lane) * L::NXP + i] = v; });
            step(k + d + 1, uin[d + 1], hh[d + 1], (T)tk[d + 1], UB, UA, [&](int i, T v) { out[((d + 1) * TILE + lane) * L::NXP + i] = v; });
        }
    }
    // ---- remaining steps (fewer than a chunk; or every step of a workgroup that reaches past the batch): own rows, directly ----
    const int rest = nsteps - k;
    if (rest <= 0) return;
    if (rest < S) {
        T uin[S][NU]; double hh[S], tk[S];
#pragma unroll
        for (int d = 0; d + 1 < S; ++d) {
            const long long idx = ip + (long long)(k + d < nsteps ? k + d : nsteps) * sk;
            hh[d] = dtt[idx]; tk[d] = tt[idx];
#pragma unroll
            for (int i = 0; i < NU; ++i) uin[d][i] = Z[idx * NZ + NX + i];
        }
#pragma unroll
        for (int d = 0; d + 1 < S; d += 2) {
            if (k + d < nsteps) { T* Zk = Z + (ip + (long long)(k + d + 1) * sk) * NZ; step(k + d, uin[d], hh[d], (T)tk[d], UA, UB, [&](int i, T v) { pub_store<PUB>(Zk + i, v); }); }
            if (d + 2 < S) { if (k + d + 1 < nsteps) { T* Zk = Z + (ip + (long long)(k + d + 2) * sk) * NZ; step(k + d + 1, uin[d + 1], hh[d + 1], (T)tk[d + 1], UB, UA, [&](int i, T v) { pub_store<PUB>(Zk + i, v); }); } }
        }
        return;
    }
    // whole trajectory on the direct path: inputs requested D steps ahead (every request is an HBM miss; requests are older than the stores
    // that follow them, so waiting for one never drains a store -- rollout_kernel)
    constexpr int D = ROLLOUT_PREFETCH;
    static_assert(D % 2 == 0, "the prefetch slots and the two upstream buffers rotate together");
    const long long last = K - 1;
    T un[D][NU]; double hn[D], tn[D];
    auto fetch = [&](int d, long long knot) {                  // knot is wave-uniform and <= K - 1
        const long long idx = ip + knot * sk;
        hn[d] = dtt[idx]; tn[d] = tt[idx];
#pragma unroll
        for (int i = 0; i < NU; ++i) un[d][i] = Z[idx * NZ + NX + i];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, k + d < last ? k + d : last);
    auto dstep = [&](int kk, int d, const T (&Uc)[NS][NX], T (&Un)[NS][NX]) {
        T uk[NU]; const double h = hn[d]; const T tk = (T)tn[d];
#pragma unroll
        for (int i = 0; i < NU; ++i) uk[i] = un[d][i];
        { const long long kn = (long long)kk + D; fetch(d, kn < last ? kn : last); }
        T* Zk = Z + (ip + ((long long)kk + 1) * sk) * NZ;
        step(kk, uk, h, tk, Uc, Un, [&](int i, T v) { pub_store<PUB>(Zk + i, v); });
    };
    for (; k + D <= nsteps; k += D) {
#pragma unroll
        for (int d = 0; d < D; d += 2) { dstep(k + d, d, UA, UB); dstep(k + d + 1, d + 1, UB, UA); }
    }
#pragma unroll
    for (int d = 0; d < D; d += 2) {                          // at most D - 1 steps are left
        if (k + d < nsteps) dstep(k + d, d, UA, UB);
        if (d + 2 < D) { if (k + d + 1 < nsteps) dstep(k + d + 1, d + 1, UB, UA); }
    }
}

template <class Fn, class T, class Cas, int R, int B, bool PUB = false>
RD_DEV void cascade_compute_wave(const Fn& fn, long long K, long long M, long long sj, long long sk, T* __restrict__ Z, const double* __restrict__ tt,
                                 const double* __restrict__ dtt, const T* __restrict__ X0, int nfull) {
    cascade_compute_wave_body<Fn, T, Cas, R, B, PUB>(fn, K, M, sj, sk, Z, tt, dtt, X0, nfull);
    if constexpr (PUB) __builtin_amdgcn_s_waitcnt(0x0F70);     // the wave's direct stores are acknowledged before the workgroup announces the whole trajectory
}

// ---- IO wavefront ---------------------------------------------------------------------------------------------------------------------
// Lane groups: 8 lanes per control row (NU <= 8 of them active), 16 lanes per state row (NX <= 16 active), 64 lanes over 64 time stamps.
// The rows one instruction touches are adjacent in memory for the layout at hand (TM = time-major: knot k of all trajectories contiguous;
// otherwise the knots of one trajectory are contiguous), and instruction `it` of a chunk differs from instruction 0 by a wave-uniform
// stride, so an address is (uniform base + it * uniform stride) + a per-lane constant computed once.
template <class Fn, class T, class Cas, int R, bool TM, bool PUB = false>
RD_DEV void cascade_io_wave(long long K, long long M, T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt, int nfull,
                            unsigned long long* pword = nullptr, unsigned long long pbase = 0) {
    using L = CascadeLds<Fn, Cas, R>;
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NB = Cas::NB, S = L::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef RD_LDS_T(T) LdsT;
    typedef RD_LDS_T(double) LdsD;
    typedef RD_LDS_T(volatile int) LdsI;
    LdsI prod = (LdsI)(smem + L::FLAG_OFF);
    LdsI in_ready = prod + NB + 1;
    LdsI out_free = prod + NB + 2;
    if (nfull <= 0) return;
    const unsigned lane = threadIdx.x & 63;
    const long long jb = (long long)blockIdx.x * TILE;        // first trajectory of the workgroup (all 64 exist: the kernel checked)
    // element (trajectory jb + j, knot k, component i) of Z sits at ((jb + j) * sj + k * sk) * NZ + i  with (sj, sk) = (K, 1) or (1, M)
    const long long sj = TM ? 1 : K, sk = TM ? M : 1;
    // -- controls: NUI instructions per chunk, 8 rows (8 lanes each) per instruction
    //    knot-contiguous: rows (j = 2 it + (l>>5), s = (l>>3)&3)           time-major: rows (s = it / 8, j = (it % 8) * 8 + (l>>3))
    constexpr int NUI = S * TILE / 8;
    const unsigned ui = lane & 7;
    const unsigned uic = ui < (unsigned)NU ? ui : NU - 1;     // idle lanes of a group duplicate the last component's access
    unsigned u_goff, u_loff; long long u_gstride; int u_lstride;
    if (TM) {
        const unsigned jl = lane >> 3;
        u_goff = jl * NZ + NX + uic;                          u_loff = jl * L::NUP + uic;
        // it -> s = it >> 3, jg = it & 7:  global (s * M + jg * 8) * NZ, LDS (s * 64 + jg * 8) * NUP : not one stride, handled per `it` below
        u_gstride = 0; u_lstride = 0;
    } else {
        const unsigned j0 = lane >> 5, s = (lane >> 3) & 3;
        u_goff = (unsigned)((j0 * K + s) * NZ + NX + uic);    u_loff = (s * TILE + j0) * L::NUP + uic;
        u_gstride = 2 * K * NZ;                               u_lstride = 2 * L::NUP;
    }
    // -- states: NXI instructions per chunk, 4 rows (16 lanes each) per instruction
    //    knot-contiguous: rows (j = it, s = l>>4)                          time-major: rows (s = it / 16, j = (it % 16) * 4 + (l>>4))
    constexpr int NXI = S * TILE / 4;
    const unsigned xi = (lane & 15) < (unsigned)NX ? (lane & 15) : NX - 1;     // idle lanes duplicate the last component's access
    unsigned x_goff, x_loff;
    if (TM) { const unsigned jl = lane >> 4; x_goff = jl * NZ + xi; x_loff = jl * L::NXP + xi; }
    else { const unsigned s = lane >> 4; x_goff = (unsigned)(s * NZ + xi); x_loff = s * TILE * L::NXP + xi; }
    // -- time stamps: S instructions per chunk for t and for dt
    //    knot-contiguous: rows (j = it * 16 + (l>>2), s = l&3)             time-major: rows (s = it, j = l)
    unsigned t_goff, t_loff;
    if (TM) { t_goff = lane; t_loff = lane; }
    else { const unsigned jl = lane >> 2, s = lane & 3; t_goff = (unsigned)(jl * K + s); t_loff = s * TILE + jl; }

    // No exec-masked memory instruction and no branch around one in this loop: idle lanes of a group duplicate a neighbour's access (same
    // address, same value), chunk indices past the end are clamped.  With masks or branches the compiler's vmcnt bookkeeping turns
    // conservative and every wait becomes "everything outstanding", i.e. a store round trip per chunk.
    T ru[NUI]; double rt[S], rh[S];
    auto load_chunk = [&](int n) {                             // controls / times of knots n S .. n S + S - 1 -> registers
        const long long k0 = (long long)n * S;
        const T* zb = Z + (jb * sj + k0 * sk) * NZ;
        const double* tb = tt + jb * sj + k0 * sk;
        const double* hb = dtt + jb * sj + k0 * sk;
#pragma unroll
        for (int it = 0; it < NUI; ++it) {
            const long long g = TM ? ((long long)(it >> 3) * M + (it & 7) * 8) * NZ : (long long)it * u_gstride;
            ru[it] = zb[g + u_goff];
        }
#pragma unroll
        for (int it = 0; it < S; ++it) {
            const long long g = TM ? (long long)it * M : (long long)it * 16 * K;
            rt[it] = tb[g + t_goff]; rh[it] = hb[g + t_goff];
        }
    };
    auto write_in = [&](int buf, int ready) {                  // registers -> IN / TS tile `buf`, then announce `ready` chunks
        LdsT in = (LdsT)(smem + L::IN_OFF) + (size_t)buf * L::IN_ELEMS;
        LdsD ts = (LdsD)(smem + L::TS_OFF) + (size_t)buf * L::TS_ELEMS;
#pragma unroll
        for (int it = 0; it < NUI; ++it) {
            const int l = TM ? ((it >> 3) * TILE + (it & 7) * 8) * L::NUP : it * u_lstride;
            in[l + u_loff] = ru[it];
        }
#pragma unroll
        for (int it = 0; it < S; ++it) {
            const int l = TM ? it * TILE : it * 16;
            ts[l + t_loff] = rt[it]; ts[S * TILE + l + t_loff] = rh[it];
        }
        RD_LDS_ORDER();
        *in_ready = ready;
    };
    load_chunk(0); write_in(0, 1);
    load_chunk(nfull > 1 ? 1 : 0); write_in(1, nfull > 1 ? 2 : 1);
    for (int n = 0; n < nfull; ++n) {
        const int n2 = n + 2 < nfull ? n + 2 : nfull - 1;     // (past the end: the last chunk again, into a tile nobody reads any more)
        load_chunk(n2);                                        // requested before the wait below: the HBM latency passes meanwhile
        const int need = (n + 1) * S;
#pragma unroll
        for (int b = 0; b < NB; ++b) while (lds_flag(prod + b) < need) __builtin_amdgcn_s_sleep(1);
        RD_LDS_ORDER();
        LdsT out = (LdsT)(smem + L::OUT_OFF) + (size_t)(n & 1) * L::OUT_ELEMS;
        T* zb = Z + (jb * sj + ((long long)n * S + 1) * sk) * NZ;          // states go to knots n S + 1 .. n S + S
        T rx[NXI];
#pragma unroll
        for (int it = 0; it < NXI; ++it) {
            const int l = TM ? ((it >> 4) * TILE + (it & 15) * 4) * L::NXP : it * L::NXP;
            rx[it] = out[l + x_loff];
        }
        RD_LDS_ORDER();
        *out_free = n + 1;                                     // (ordered behind the reads above: the tile may be overwritten now)
        // the inputs go into LDS BEFORE the stores are issued: the loads are then the youngest memory operations in flight and waiting for
        // them waits only for the previous chunk's stores (a chunk old), not for stores just issued
        write_in(n & 1, n + 3 < nfull ? n + 3 : nfull);
        // (the inputs just written were requested at the top of this iteration: their arrival says that the stores of chunk n - 1, which are
        // older, have been acknowledged -- knots 0 .. 4 n are complete)
        if constexpr (PUB) pub_progress(pword, pbase, (long long)n * S + 1);
#pragma unroll
        for (int it = 0; it < NXI; ++it) {
            const long long g = TM ? ((long long)(it >> 4) * M + (it & 15) * 4) * NZ : (long long)it * K * NZ;
            pub_store<PUB>(zb + g + x_goff, rx[it]);
        }
    }
    if constexpr (PUB) { __builtin_amdgcn_s_waitcnt(0x0F70); pub_progress(pword, pbase, (long long)nfull * S + 1); }
}

// ---- FiniteDifference along the cascade ---------------------------------------------------------------------------------------------
// Forward differences evaluate the discrete dynamics 1 + (n+m) times per knot point (src/jacobian_gen.jl:643-685, FiniteDiff's forward
// default).  Perturbing component c leaves every block UPSTREAM of c's block bit-for-bit unchanged -- those rows of the column are exact
// zeros in the reference as well -- so the perturbed evaluation only integrates the blocks from c's block downwards, reading the upstream
// blocks' stage points from the base evaluation (kept in registers: 4 stages x 7 components for a rigid body).  Columns are compile-time
// constants, the model's dynamics function is instantiated per column class and the untouched rows fold away: 4 400 instead of 6 200 fp64
// instructions per satellite knot point, the base evaluation included (a position or velocity column costs the r,v block alone).
template <class Fn, class Cas> struct FdBaseHook {             // base evaluation: remember the stage points of the blocks that have a downstream
    using T = typename Fn::T;
    static constexpr int NX = Fn::NX, NS = rule_stages<Fn::RULE_ID>();
    T (&S)[NS][NX];
    template <int St> RD_DEV void at(T (&xs)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) if (Cas::block_of(i) < Cas::NB - 1) S[St][i] = xs[i];
    }
};
template <class Fn, class Cas, int B> struct FdColHook {        // perturbed evaluation from block B downwards
    using T = typename Fn::T;
    static constexpr int NX = Fn::NX, NS = rule_stages<Fn::RULE_ID>();
    const T (&S)[NS][NX];
    template <int St> RD_DEV void at(T (&xs)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) if (Cas::block_of(i) < B) xs[i] = S[St][i];
    }
};
template <class Fn, class T, class Plan, class Cas, int G, int C> struct FdCascadeCols {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NS = rule_stages<Fn::RULE_ID>();
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr, const T (&f0)[NX], const T (&S)[NS][NX]) {
        if constexpr (C < NZ) {
            if constexpr (FirstSplitOf<Plan>::N > 0 && C == FirstSplitOf<Plan>::N) {
                if (fr.first) __syncthreads();              // barrier E of the split-role loop: the early part of the first tile is complete
            }
            if constexpr (Plan::wave_of(C) == G) {
                constexpr int B = C < NX ? Cas::block_of(C) : Cas::control_block(C - NX);
                T x1[NX], u1[NU], y1[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) x1[i] = fr.x[i];
#pragma unroll
                for (int i = 0; i < NU; ++i) u1[i] = fr.u[i];
                T eps;
                if constexpr (C < NX) { eps = fd_step<T>(fr.x[C]); x1[C] = x1[C] + eps; }
                else { eps = fd_step<T>(fr.u[C - NX]); if (!fr.term) u1[C - NX] = u1[C - NX] + eps; }     // a terminal knot point evaluates f with u = 0 throughout
                // (opaque copies: without them the compiler merges sub-expressions ACROSS columns -- every column starts from the same point --
                // and the merged values stay live through the whole column sequence: 217 spilled registers)
#pragma unroll
                for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(x1[i]));
#pragma unroll
                for (int i = 0; i < NU; ++i) asm volatile("" : "+v"(u1[i]));
                FdColHook<Fn, Cas, B> hk{S};
                rk_hooked<Fn::RULE_ID>(fn.m, x1, u1, fr.t, fr.h, y1, hk);
                const T inv = T(1) / eps;                     // one division per column (the quotient differs from (f1-f0)/eps by <= 1 ulp)
                T col[NY];
#pragma unroll
                for (int i = 0; i < NY; ++i) col[i] = (Cas::block_of(i) < B) ? T(0) : (y1[i] - f0[i]) * inv;
                if constexpr (C >= NX) {                     // control columns of a terminal knot point: f does not see u, the column is exactly zero
                    if (fr.term) {
#pragma unroll
                        for (int i = 0; i < NY; ++i) col[i] = T(0);
                    }
                }
                fr.put_at(C, col);
                __builtin_amdgcn_sched_barrier(0);          // finish this column before the next one starts: interleaved columns spill
            }
            FdCascadeCols<Fn, T, Plan, Cas, G, C + 1>::run(fn, fr, f0, S);
        }
    }
};
template <class Fn, class T, class Plan, class Cas> struct FdCascadeBody {
    static constexpr bool EARLY = true;
    static constexpr bool COPY_PREFETCH = CopyPrefetchFd<decltype(Fn::m), T>::value;      // (kernels.hpp CopyPrefetchOf; the models: fam_rigid.hpp)
    static constexpr int COPY_PACE = 0;
    static constexpr int NX = Fn::NX, NS = rule_stages<Fn::RULE_ID>();
    template <int G> static RD_DEV void waves(const Fn& fn, JacFrame<Fn, T>& fr, const T (&f0)[NX], const T (&S)[NS][NX]) {
        if constexpr (G < Plan::NG) {
            if (fr.g == G) FdCascadeCols<Fn, T, Plan, Cas, G, 0>::run(fn, fr, f0, S);
            else waves<G + 1>(fn, fr, f0, S);
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        static_assert(Plan::NCHUNK == 1, "the split-role kernel stages whole tiles");
        T f0[NX], S[NS][NX];
        FdBaseHook<Fn, Cas> hb{S};
        rk_hooked<Fn::RULE_ID>(fn.m, fr.x, fr.u, fr.t, fr.h, f0, hb);
        if (fr.g == Plan::value_wave && fr.want_y) fr.put_value(Fn::NX + Fn::NU, f0);
        waves<0>(fn, fr, f0, S);
    }
};
template <class Fn, class T, class Plan, class Cas, int NC, bool DYN = false>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_fd_cascade_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                 const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y, unsigned* sched) {
    if constexpr (DYN) jac_split_loop<Fn, T, Plan, NC, FdCascadeBody<Fn, T, Plan, Cas>>(fn, SingleBatchDyn<T>{{N, Z, tt, dtt, J, Y}, sched});
    else jac_split_loop<Fn, T, Plan, NC, FdCascadeBody<Fn, T, Plan, Cas>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Y});
}

template <class Fn, class T, class Cas, int R, int B, bool PUB = false> struct CascadeWaves {
    static RD_DEV void run(int wave, const Fn& fn, long long K, long long M, long long sj, long long sk, T* Z, const double* tt, const double* dtt,
                           const T* X0, int nfull, unsigned long long* pword = nullptr, unsigned long long pbase = 0) {
        if constexpr (B < Cas::NB) {
            if (wave == B) cascade_compute_wave<Fn, T, Cas, R, B, PUB>(fn, K, M, sj, sk, Z, tt, dtt, X0, nfull);
            else CascadeWaves<Fn, T, Cas, R, B + 1, PUB>::run(wave, fn, K, M, sj, sk, Z, tt, dtt, X0, nfull, pword, pbase);
        } else {
            if (sk == 1) cascade_io_wave<Fn, T, Cas, R, false, PUB>(K, M, Z, tt, dtt, nfull, pword, pbase);
            else cascade_io_wave<Fn, T, Cas, R, true, PUB>(K, M, Z, tt, dtt, nfull, pword, pbase);
        }
    }
};
// (sj, sk) = (K, 1) [the knots of a trajectory are contiguous] or (1, M) [time-major]
template <class Fn, class T, class Cas, int R>
__global__ void __launch_bounds__((Cas::NB + 1) * TILE) rollout_cascade_kernel(Fn fn, long long K, long long M, long long sj, long long sk, T* __restrict__ Z,
                                                                               const double* __restrict__ tt, const double* __restrict__ dtt, const T* __restrict__ X0) {
    using L = CascadeLds<Fn, Cas, R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x < 16) reinterpret_cast<int*>(smem + L::FLAG_OFF)[threadIdx.x] = 0;     // prod[0..NB), cons, in_ready, out_free
    __syncthreads();
    // chunks served by the IO wave: all full ones, unless the workgroup's 64 trajectories reach past the batch (then everything is direct)
    const bool whole = ((long long)blockIdx.x + 1) * TILE <= M;
    const int nfull = (whole && K >= 2) ? (int)((K - 1) / L::S) : 0;
    CascadeWaves<Fn, T, Cas, R, 0>::run(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fn, K, M, sj, sk, Z, tt, dtt, X0, nfull);
}
#undef RD_LDS_T
#undef RD_LDS_DRAIN

}  // namespace rd
