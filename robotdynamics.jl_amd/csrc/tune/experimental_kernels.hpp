// experimental_kernels.hpp -- organisations of the flagship ForwardAD Jacobian kernel that were built, verified bit-compatible
// with the shipped kernel and MEASURED SLOWER on MI355X.  Included by the tuning harness only
// (tune_jac.hip); the product library does not compile them.
//   jac_ad_wave_kernel    one wave per (tile, column group), one-shot grid                      43-46 us  (shipped: ~30)
//   jac_ad_split2_kernel  two column chunks double-buffered in LDS, streaming copy waves         37-42 us
//   jac_ad_split3_kernel  as split2, LDS flags instead of workgroup barriers, two copy groups    ~50 us
#pragma once
#include "../kernels.hpp"

namespace rd {

// Two column chunks [0, SPLIT) and [SPLIT, NZ) with a column -> wave table; f(z) travels with the second chunk.
template <int NZ, int NG_, int SPLIT, int VW, int... W> struct TablePlan2 {
    static_assert(sizeof...(W) == NZ, "one wave index per Jacobian column");
    static constexpr int NG = NG_, NCHUNK = 2;
    static constexpr int chunk_begin(int ch) { return ch == 0 ? 0 : SPLIT; }
    static constexpr int chunk_end(int ch) { return ch == 0 ? SPLIT : NZ; }
    static constexpr int wave_of(int c) { constexpr int w[NZ] = {W...}; return w[c]; }
    static constexpr int value_chunk = 1, value_wave = VW;
};

// ---- double-buffered two-chunk split-role kernel ----------------------------------------------------------------------
// Measured on the single-buffer split kernel above (phase time stamps): per tile the store chain is
// [barrier A] -> LDS->registers (0.5 us) -> [barrier B] -> store issue (3.5-4 us, HBM-bound: all CUs are in lock step),
// the compute chain is 3-3.5 us, the first tile's compute (3.7 us) has no stores under it.  Here the tile is produced in
// TWO column chunks into TWO LDS buffers: the compute waves fill buffer c%2 with chunk c while the copy waves stream
// chunk c-1 from the other buffer straight to global memory (ds_read -> global_store, a few units in flight, no register
// staging of a whole tile).  ONE workgroup barrier per chunk: barrier k = "chunk k complete AND chunk k-1 drained".
// The un-overlapped prologue shrinks to half a tile of arithmetic and the copy waves never wait for an LDS hand-off.
// A chunk's rows are runs of (ncols*NY) values per knot point at a pitch of NY*NZ: stored as lane-contiguous elements.
template <class T> constexpr int lds_row_stride(int r) {
    // rows must start 16-B aligned in LDS (ds_read_b128) -> stride multiple of V = 16/sizeof(T); among those pick one whose
    // dword stride is not a multiple of 16 banks (fp64: stride = 2 mod 4 doubles -> 2-way conflicts on the lane-strided writes)
    constexpr int V = 16 / (int)sizeof(T);
    int rs = (r + V - 1) / V * V;
    while ((rs / V) % 2 == 0) rs += V;
    return rs;
}
template <class Fn, class T, class Plan> struct Split2Lds {
    static constexpr int NZ = Fn::NX + Fn::NU, NY = Fn::NY;
    static_assert(Plan::NCHUNK == 2, "two column chunks");
    static constexpr int R0 = (Plan::chunk_end(0) - Plan::chunk_begin(0)) * NY;          // values per knot point, chunk 0
    static constexpr int R1 = (Plan::chunk_end(1) - Plan::chunk_begin(1)) * NY;          // chunk 1
    static constexpr int RS0 = lds_row_stride<T>(R0), RS1 = lds_row_stride<T>(R1);
    static constexpr size_t B0_BYTES = (size_t)TILE * RS0 * sizeof(T);
    static constexpr size_t B1_BYTES = (size_t)TILE * RS1 * sizeof(T);
    static constexpr size_t Y_BYTES = ((size_t)TILE * NY * sizeof(T) + 15) / 16 * 16;    // f(z) tile, dense (a contiguous slab of Y)
    static constexpr size_t Z_ELEMS = (size_t)TILE * NZ;
    static constexpr size_t IN_BYTES = (Z_ELEMS * sizeof(T) + 2 * TILE * sizeof(double) + 15) / 16 * 16;
    static constexpr size_t BYTES = B0_BYTES + B1_BYTES + Y_BYTES + 2 * IN_BYTES;
};

// copy R values per row from LDS rows (stride RS, 16-B aligned) to global rows (stride pitch, element aligned), ROWS rows,
// by NTH copy threads (`ct` = my index): 16-B units, then the odd tail of each row element-wise.  Full tiles only.
template <class T, int R, int RS, int ROWS, int NTH>
RD_DEV void drain_rows(const T* __restrict__ lds, T* __restrict__ dst, int pitch, int ct) {
    using VA = typename Vec16<T>::type;                                        // element-aligned 16-B vector (global side)
    typedef T VL __attribute__((ext_vector_type(16 / sizeof(T))));           // 16-B aligned vector (LDS side)
    constexpr int V = 16 / (int)sizeof(T), UPR = R / V, TAILN = R - UPR * V, UNITS = ROWS * UPR;
#pragma unroll
    for (int e0 = 0; e0 < UNITS; e0 += NTH) {
        const int e = e0 + ct;
        if (e0 + NTH <= UNITS || e < UNITS) {
            const int row = e / UPR, u = e - row * UPR;
            const VL v = *reinterpret_cast<const VL*>(lds + row * RS + u * V);
            VA o;
#pragma unroll
            for (int q = 0; q < V; ++q) o[q] = v[q];
            *reinterpret_cast<VA*>(dst + row * pitch + u * V) = o;
        }
    }
    if constexpr (TAILN > 0) {
#pragma unroll
        for (int e0 = 0; e0 < ROWS * TAILN; e0 += NTH) {
            const int e = e0 + ct;
            if (e < ROWS * TAILN) { const int row = e / TAILN, j = e - row * TAILN; dst[row * pitch + UPR * V + j] = lds[row * RS + UPR * V + j]; }
        }
    }
}
// ragged last tile: element-wise with a row predicate
template <class T, int R, int RS, int NTH>
RD_DEV void drain_rows_ragged(const T* __restrict__ lds, T* __restrict__ dst, int pitch, int nk, int ct) {
    for (int e = ct; e < TILE * R; e += NTH) {
        const int row = e / R, col = e - row * R;
        if (row < nk) dst[row * pitch + col] = lds[row * RS + col];
    }
}

template <class Fn, class T, class Plan, int NC, class Body>
RD_DEV void jac_split2_loop(const Fn& fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                            T* __restrict__ J, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG;
    using L = Split2Lds<Fn, T, Plan>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* b0 = reinterpret_cast<T*>(smem);
    T* b1 = reinterpret_cast<T*>(smem + L::B0_BYTES);
    T* yb = reinterpret_cast<T*>(smem + L::B0_BYTES + L::B1_BYTES);
    unsigned char* in0 = smem + L::B0_BYTES + L::B1_BYTES + L::Y_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: role and column branches are scalar branches
    const long long ntiles = (N + TILE - 1) / TILE;
    long long tile = blockIdx.x;
    if (tile >= ntiles) return;

    // The compute waves never touch global memory: a VMEM instruction of a wave whose CU is saturated with stores blocks AT
    // ISSUE (measured: 1.8 us per tile at the input prefetch), so the copy waves -- which live in the memory pipeline anyway --
    // also fetch the next tile's z/t/dt (issued ahead of a drain) and stage them into the other LDS input buffer.
    if (wave < NG) {
        // =========================== compute waves ===========================
        JacFrame<Fn, T> fr;
        fr.g = wave;
        fr.yrow = yb + lane * NY;
        fr.want_y = (Y != nullptr);
        __syncthreads();                                   // barrier: first inputs staged
        int buf = 0;
        for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
            const T* zs = reinterpret_cast<const T*>(in0 + buf * L::IN_BYTES);
            const double* ts = reinterpret_cast<const double*>(zs + L::Z_ELEMS);
            const long long k0 = tile * TILE;
            const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
            const int kl = lane < nk ? lane : nk - 1;
            const double hd = ts[TILE + kl];
            fr.t = (T)ts[kl]; fr.h = (T)hd; fr.term = is_terminal_dt(hd);
            read_knot<Fn, T>(zs, kl, fr.term, fr.x, fr.u);
            [[maybe_unused]] const int it_ = (int)((tile - blockIdx.x) / gridDim.x);
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 0);
            fr.row = b0 + lane * L::RS0;
            Body::template run<0>(fn, fr);
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 1);
            __syncthreads();                               // chunk 0 complete (and the previous tile's chunk 1 drained)
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 2);
            fr.row = b1 + lane * L::RS1;
            Body::template run<1>(fn, fr);
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 3);
            __syncthreads();                               // chunk 1 complete (and chunk 0 drained), next inputs visible
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 4);
        }
    } else {
        // =========================== copy waves ==============================
        constexpr int NTK = NC * TILE;
        constexpr int ZPT = (TILE * NZ + NTK - 1) / NTK;
        const int ct = threadIdx.x - NG * TILE;            // 0 .. NC*64-1
        {
            T* zs = reinterpret_cast<T*>(in0);
            double* ts = reinterpret_cast<double*>(zs + L::Z_ELEMS);
            const long long k0 = tile * TILE;
            const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
            for (int e = ct; e < nk * NZ; e += NTK) zs[e] = Z[k0 * NZ + e];
            if (ct < nk) ts[ct] = tt[k0 + ct];
            else if (ct >= TILE && ct < TILE + nk) ts[ct] = dtt[k0 + ct - TILE];
        }
        // inputs of tile i+1 are LOADED (into registers) right after the barrier "chunk 1 of tile i-1 complete" and WRITTEN to
        // the other LDS input buffer right after the barrier "chunk 0 of tile i complete": the wait for those loads then
        // sits behind a compute-bound barrier wait, not behind this wave's own freshly issued stores.
        T zpre[ZPT]; double tpre;
        auto load_inputs = [&](long long tl) {
            const long long k1 = tl * TILE;
            const int nk1 = (tl < ntiles) ? (int)((N - k1) < TILE ? (N - k1) : TILE) : 0;
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTK + ct; zpre[r] = Z[(e < nk1 * NZ) ? k1 * NZ + e : 0]; }   // unconditional loads
            const double* tp = (ct < TILE) ? tt : dtt;
            const int idx = ct & (TILE - 1);
            tpre = tp[(ct < 2 * TILE && idx < nk1) ? k1 + idx : 0];
        };
        load_inputs(tile + gridDim.x);
        __syncthreads();                                   // barrier: first inputs staged
        int buf = 0;
        for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
            const long long k0 = tile * TILE;
            const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
            T* Jt = J + k0 * (NY * NZ);
            T* zn = reinterpret_cast<T*>(in0 + (buf ^ 1) * L::IN_BYTES);
            double* tn = reinterpret_cast<double*>(zn + L::Z_ELEMS);
            [[maybe_unused]] const int it_ = (int)((tile - blockIdx.x) / gridDim.x);
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 0);
            __syncthreads();                               // chunk 0 complete
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 1);
#pragma unroll
            for (int r = 0; r < ZPT; ++r) { const int e = r * NTK + ct; if (e < TILE * NZ) zn[e] = zpre[r]; }      // inputs of the next tile
            if (ct < 2 * TILE) tn[ct] = tpre;
            __builtin_amdgcn_sched_barrier(0);
            if (nk == TILE) drain_rows<T, L::R0, L::RS0, TILE, NTK>(b0, Jt + Plan::chunk_begin(0) * NY, NY * NZ, ct);
            else drain_rows_ragged<T, L::R0, L::RS0, NTK>(b0, Jt + Plan::chunk_begin(0) * NY, NY * NZ, nk, ct);
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 2);
            __syncthreads();                               // chunk 1 complete; next inputs visible
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 3);
            load_inputs(tile + 2 * (long long)gridDim.x);  // ahead of the drain in the memory pipeline
            __builtin_amdgcn_sched_barrier(0);
            if (nk == TILE) {
                drain_rows<T, L::R1, L::RS1, TILE, NTK>(b1, Jt + Plan::chunk_begin(1) * NY, NY * NZ, ct);
                if (Y) drain_rows<T, TILE * NY, TILE * NY, 1, NTK>(yb, Y + k0 * NY, 0, ct);               // the f(z) tile is one dense row
            } else {
                drain_rows_ragged<T, L::R1, L::RS1, NTK>(b1, Jt + Plan::chunk_begin(1) * NY, NY * NZ, nk, ct);
                if (Y) for (int e = ct; e < nk * NY; e += NTK) Y[k0 * NY + e] = yb[e];
            }
            RD_STAMP_AT(blockIdx.x * 8 + wave, it_ * 5 + 4);
        }
#ifdef RD_STAMP
        __builtin_amdgcn_s_waitcnt(0x0F70); RD_STAMP_AT(blockIdx.x * 8 + wave, 60);   // all stores complete
#endif
    }
}
template <class Fn, class T, class Plan, int NC>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) jac_ad_split2_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_split2_loop<Fn, T, Plan, NC, AdUnrolledBody<Fn, T, Plan>>(fn, N, Z, tt, dtt, J, Y);
}

// ---- flag-synchronised two-chunk kernel: the store stream never pauses ------------------------------------------------
// In both kernels above the copy waves alternate "LDS -> registers" and "issue stores" and meet the compute waves at
// workgroup barriers, so per tile the chain is [store issue 3.5-4 us] + [LDS read 0.5 us] + barrier skew, and a wave that is
// blocked issuing stores cannot arrive at a barrier.  Here the s_barrier is replaced by monotonic LDS flags (workgroup-scope
// release/acquire): two copy GROUPS of two waves each own one column chunk (one LDS buffer) of every tile and run
// independently of each other --
//   compute waves, tile i:  wait inputs(i) ; wait empty[0] >= 2i ; chunk 0 -> buffer 0 ; full[0] += 1 each ;
//                                            wait empty[1] >= 2i ; chunk 1 -> buffer 1 ; full[1] += 1 each
//   copy group g, tile i:   wait full[g] >= 4(i+1) ; buffer g -> registers ; empty[g] += 1 each ; issue the stores
// so while group 0 is blocked issuing chunk 0's stores, group 1 picks up chunk 1 the moment it is complete, and the compute
// waves only ever wait for an LDS read-out (0.3-0.5 us, done a whole chunk earlier), never for a store.  Group 0 also
// fetches the next tile's z/t/dt (compute waves never touch global memory).  Every spin is bounded.
// The flags and everything they order live in LDS, so the hand-off needs LDS ordering only: "s_waitcnt lgkmcnt(0)" (this
// wave's LDS accesses are complete; LDS executes a CU's requests in order) plus a compiler barrier.  A workgroup-scope
// release FENCE would also drain vmcnt, i.e. make a copy wave wait for every store it has in flight (measured: 2.8 us).
RD_DEV void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
RD_DEV void flag_wait(unsigned* flag, unsigned target) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) __builtin_trap();          // never reached in a correct schedule; turns a would-be hang into a fault
    }
    lds_order();
}
RD_DEV void flag_signal(unsigned* flag, int lane) {
    lds_order();                                            // all lanes' LDS reads/writes of this wave are done
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS rows (stride RS, 16-B aligned) -> registers -> global rows (stride pitch, element aligned); ROWS rows of R values, NTH threads
template <class T, int R, int RS, int ROWS, int NTH> struct RowMover {
    using VA = typename Vec16<T>::type;
    typedef T VL __attribute__((ext_vector_type(16 / sizeof(T))));
    static constexpr int V = 16 / (int)sizeof(T), UPR = R / V, TAILN = R - UPR * V, UNITS = ROWS * UPR, NIT = (UNITS + NTH - 1) / NTH;
    static constexpr int TUNITS = ROWS * TAILN, TIT = (TUNITS + NTH - 1) / NTH;
    VL v[NIT];
    T tl[TIT > 0 ? TIT : 1];
    // Unit e = i*NTH + ct is (row, u) = (e / UPR, e % UPR).  Only the first pair is divided out; the rest advance incrementally
    // (u += NTH % UPR, row += NTH / UPR, carry) with running LDS / global offsets: ~5 cheap VALU per unit instead of three
    // quarter-rate integer multiplies.  `ct` is laundered through an empty asm so that this arithmetic is redone per call: hoisted
    // out of the persistent tile loop it would pin 2 x NIT index registers on top of the 4 x NIT data registers.
    static constexpr int DR = NTH / UPR, DU = NTH % UPR;
    RD_DEV void load(const T* __restrict__ lds, int ct) {
        asm volatile("" : "+v"(ct));
        int row = ct / UPR, u = ct - row * UPR;
        int lo = row * RS + u * V;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = (i + 1) * NTH <= UNITS || i * NTH + ct < UNITS;
            v[i] = *reinterpret_cast<const VL*>(lds + (ok ? lo : 0));
            u += DU; lo += DR * RS + DU * V;
            if (u >= UPR) { u -= UPR; lo += RS - UPR * V; }
        }
#pragma unroll
        for (int i = 0; i < TIT; ++i) {
            const int e = i * NTH + ct, ec = e < TUNITS ? e : 0;
            const int r = ec / (TAILN > 0 ? TAILN : 1), j = ec - r * (TAILN > 0 ? TAILN : 1);
            tl[i] = lds[r * RS + UPR * V + j];
        }
    }
    RD_DEV void store(T* __restrict__ dst, int pitch, int nrows, int ct) const {
        asm volatile("" : "+v"(ct));
        int row = ct / UPR, u = ct - row * UPR;
        unsigned go = (unsigned)row * (unsigned)pitch + (unsigned)(u * V);    // uniform base + 32-bit lane offset (saddr form)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = ((i + 1) * NTH <= UNITS || i * NTH + ct < UNITS) && row < nrows;
            if (ok) {
                VA o;
#pragma unroll
                for (int q = 0; q < V; ++q) o[q] = v[i][q];
                __builtin_nontemporal_store(o, reinterpret_cast<VA*>(dst + go));
            }
            u += DU; row += DR; go += (unsigned)(DR * pitch + DU * V);
            if (u >= UPR) { u -= UPR; row += 1; go += (unsigned)(pitch - UPR * V); }
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TIT; ++i) {
            const int e = i * NTH + ct;
            const int r = e / (TAILN > 0 ? TAILN : 1), j = e - r * (TAILN > 0 ? TAILN : 1);
            if (e < TUNITS && r < nrows) { const unsigned off = (unsigned)r * (unsigned)pitch + (unsigned)(UPR * V + j); __builtin_nontemporal_store(tl[i], dst + off); }
        }
    }
};

template <class Fn, class T, class Plan> struct Split3Lds {
    using B = Split2Lds<Fn, T, Plan>;
    static constexpr size_t FLAG_OFF = B::BYTES;
#ifdef RD_STAMP
    static constexpr size_t BYTES = B::BYTES + 64 + 8 * 64 * 8;
#else
    static constexpr size_t BYTES = B::BYTES + 64;           // flags: full[2], empty[2], in_ready[2]
#endif
};

template <class Fn, class T, class Plan, class Body>
RD_DEV void jac_split3_loop(const Fn& fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                            T* __restrict__ J, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NG = Plan::NG, NCG = 2, NTG = NCG * TILE;
    using L = Split2Lds<Fn, T, Plan>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* b0 = reinterpret_cast<T*>(smem);
    T* b1 = reinterpret_cast<T*>(smem + L::B0_BYTES);
    T* yb = reinterpret_cast<T*>(smem + L::B0_BYTES + L::B1_BYTES);
    unsigned char* in0 = smem + L::B0_BYTES + L::B1_BYTES + L::Y_BYTES;
    unsigned* flags = reinterpret_cast<unsigned*>(smem + Split3Lds<Fn, T, Plan>::FLAG_OFF);
    unsigned* full = flags;          // [2]
    unsigned* empty = flags + 2;     // [2]
    unsigned* in_ready = flags + 4;  // [2]
#ifdef RD_STAMP   // time stamps are collected in LDS and written out at the end: a global store from a compute wave would block at issue
    long long* stl = reinterpret_cast<long long*>(smem + Split3Lds<Fn, T, Plan>::FLAG_OFF + 64);
#define RD_STAMP_LDS(ev) do { if ((threadIdx.x & 63) == 0) stl[(threadIdx.x >> 6) * 64 + (ev)] = wall_clock64(); } while (0)
#define RD_STAMP_FLUSH() do { lds_order(); if (g_stamps) g_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)] = stl[(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)]; } while (0)
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) stl[i] = 0;
#else
#define RD_STAMP_LDS(ev) do {} while (0)
#define RD_STAMP_FLUSH() do {} while (0)
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long ntiles = (N + TILE - 1) / TILE;
    long long tile = blockIdx.x;
    if (tile >= ntiles) return;
    if (threadIdx.x < 8) flags[threadIdx.x] = 0;
    __syncthreads();                                        // the only workgroup barrier: flags initialised

    if (wave < NG) {
        // =========================== compute waves ===========================
        JacFrame<Fn, T> fr;
        fr.g = wave;
        fr.yrow = yb + lane * NY;
        fr.want_y = (Y != nullptr);
        for (unsigned it = 0; tile < ntiles; tile += gridDim.x, ++it) {
            const int buf = it & 1;
            const T* zs = reinterpret_cast<const T*>(in0 + buf * L::IN_BYTES);
            const double* ts = reinterpret_cast<const double*>(zs + L::Z_ELEMS);
            const long long k0 = tile * TILE;
            const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
            const int kl = lane < nk ? lane : nk - 1;
            RD_STAMP_LDS(it * 5 + 0);
            flag_wait(&in_ready[buf], NCG * (it / 2 + 1));
            const double hd = ts[TILE + kl];
            fr.t = (T)ts[kl]; fr.h = (T)hd; fr.term = is_terminal_dt(hd);
            read_knot<Fn, T>(zs, kl, fr.term, fr.x, fr.u);
            flag_wait(&empty[0], NCG * it);
            RD_STAMP_LDS(it * 5 + 1);
            fr.row = b0 + lane * L::RS0;
            Body::template run<0>(fn, fr);
            flag_signal(&full[0], lane);
            RD_STAMP_LDS(it * 5 + 2);
            flag_wait(&empty[1], NCG * it);
            RD_STAMP_LDS(it * 5 + 3);
            fr.row = b1 + lane * L::RS1;
            Body::template run<1>(fn, fr);
            flag_signal(&full[1], lane);
            RD_STAMP_LDS(it * 5 + 4);
        }
        RD_STAMP_FLUSH();
    } else {
        // =========================== copy groups =============================
        const int grp = (wave - NG) / NCG;                  // 0: chunk 0 + input staging, 1: chunk 1 + f(z)
        const int ct = threadIdx.x - (NG + grp * NCG) * TILE;   // 0 .. 127 inside the group
        if (grp == 0) {
            constexpr int ZPT = (TILE * NZ + NTG - 1) / NTG;
            auto stage_inputs = [&](long long tl, int buf, const T (&zr)[ZPT], double tr) {
                T* zn = reinterpret_cast<T*>(in0 + buf * L::IN_BYTES);
                double* tn = reinterpret_cast<double*>(zn + L::Z_ELEMS);
#pragma unroll
                for (int r = 0; r < ZPT; ++r) { const int e = r * NTG + ct; if (e < TILE * NZ) zn[e] = zr[r]; }
                tn[ct] = tr;                                   // 128 threads: t[64], dt[64]
                flag_signal(&in_ready[buf], lane);
            };
            T zpre[ZPT]; double tpre;
            auto load_inputs = [&](long long tl) {
                const long long k1 = tl * TILE;
                const int nk1 = (tl < ntiles) ? (int)((N - k1) < TILE ? (N - k1) : TILE) : 0;
#pragma unroll
                for (int r = 0; r < ZPT; ++r) { const int e = r * NTG + ct; zpre[r] = Z[(e < nk1 * NZ) ? k1 * NZ + e : 0]; }   // unconditional loads
                const double* tp = (ct < TILE) ? tt : dtt;
                const int idx = ct & (TILE - 1);
                tpre = tp[(idx < nk1) ? k1 + idx : 0];
            };
            load_inputs(tile);
            stage_inputs(tile, 0, zpre, tpre);
            load_inputs(tile + gridDim.x);
            RowMover<T, L::R0, L::RS0, TILE, NTG> mv;
            for (unsigned it = 0; tile < ntiles; tile += gridDim.x, ++it) {
                const long long k0 = tile * TILE;
                const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
                RD_STAMP_LDS(it * 5 + 0);
                flag_wait(&full[0], NG * (it + 1));
                RD_STAMP_LDS(it * 5 + 1);
                // all compute waves are past chunk 0 of tile `it`, hence done reading the inputs of tile it-1: its buffer is free
                stage_inputs(tile + gridDim.x, (it + 1) & 1, zpre, tpre);
                __builtin_amdgcn_sched_barrier(0);
                mv.load(b0, ct);
                flag_signal(&empty[0], lane);
                RD_STAMP_LDS(it * 5 + 2);
                mv.store(J + k0 * (NY * NZ) + Plan::chunk_begin(0) * NY, NY * NZ, nk, ct);
                RD_STAMP_LDS(it * 5 + 3);
                __builtin_amdgcn_sched_barrier(0);
                load_inputs(tile + 2 * (long long)gridDim.x);   // consumed one tile later, after a compute-bound wait
            }
            RD_STAMP_FLUSH();
        } else {
            RowMover<T, L::R1, L::RS1, TILE, NTG> mv;
            RowMover<T, TILE * NY, TILE * NY, 1, NTG> my;
            for (unsigned it = 0; tile < ntiles; tile += gridDim.x, ++it) {
                const long long k0 = tile * TILE;
                const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
                RD_STAMP_LDS(it * 5 + 0);
                flag_wait(&full[1], NG * (it + 1));
                RD_STAMP_LDS(it * 5 + 1);
                mv.load(b1, ct);
                if (Y) my.load(yb, ct);
                flag_signal(&empty[1], lane);
                RD_STAMP_LDS(it * 5 + 2);
                mv.store(J + k0 * (NY * NZ) + Plan::chunk_begin(1) * NY, NY * NZ, nk, ct);
                RD_STAMP_LDS(it * 5 + 3);
                if (Y) {
                    if (nk == TILE) my.store(Y + k0 * NY, 0, 1, ct);
                    else for (int e = ct; e < nk * NY; e += NTG) Y[k0 * NY + e] = yb[e];   // ragged last tile (yb is not overwritten before the next full[1])
                }
            }
            RD_STAMP_FLUSH();
        }
    }
}
template <class Fn, class T, class Plan>
__global__ void __launch_bounds__((Plan::NG + 4) * TILE) jac_ad_split3_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                              const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    jac_split3_loop<Fn, T, Plan, AdUnrolledBody<Fn, T, Plan>>(fn, N, Z, tt, dtt, J, Y);
}

// ---- one WAVE per (tile, column group): the flagship ForwardAD organisation -----------------------------------------
// Measured on MI355X (tools/fp64_ubench.hip): a one-shot grid of small workgroups streams stores ~25 % faster
// than a persistent grid with a static tile partition (hardware workgroup dispatch balances the per-CU differences in
// achieved HBM bandwidth), and with only 4 tiles per CU at N = 65 536 a persistent pipeline pays its un-overlapped
// first compute phase four times over.  So: workgroup = ONE wave = 64 knot points x one group of Jacobian columns
// (compile-time column list, structural zeros folded), ~10 KB of LDS and ~128 VGPRs, so that ~16 such waves live on a
// CU in different phases: some computing, some with their stores draining.  A wave
//   1. stages its tile's z (coalesced global loads -> LDS slice -> one knot point per lane),
//   2. computes its columns in registers (AdRun below),
//   3. transposes each run of columns through an LDS slice of SL knot points and stores it as lane-contiguous
//      8-byte elements: a run is (ncols * NY) consecutive values per knot point, rows NY*NZ apart.
// Blocks are numbered so that all column groups of a tile land on the same XCD (blockIdx % 8 is the XCD): the
// tile's z is fetched into ONE L2, and the partially written 128-B lines at run boundaries merge in that L2.
template <int NZ, int NG_> struct GroupPlan {          // default: NG contiguous column ranges, f(z) emitted by group 0
    static constexpr int NGROUPS = NG_, NRUNS = 1;
    static constexpr int run_begin(int g, int) { return g * NZ / NG_; }
    static constexpr int run_end(int g, int) { return (g + 1) * NZ / NG_; }
    static constexpr int value_group = 0;
};

template <class Fn, class T, class Plan, int SL> struct WaveLds {
    static constexpr int NZ = Fn::NX + Fn::NU, NY = Fn::NY;
    static constexpr int max_cols() {
        int m = 1;
        for (int g = 0; g < Plan::NGROUPS; ++g) for (int r = 0; r < Plan::NRUNS; ++r) { const int c = Plan::run_end(g, r) - Plan::run_begin(g, r); m = c > m ? c : m; }
        return m;
    }
    static constexpr int RS = (max_cols() * NY) | 1;          // odd row stride: conflict-free lane-strided ds_write_b64
    static constexpr int ZS = NZ | 1;
    static constexpr size_t ELEMS = (size_t)SL * (RS > ZS ? RS : ZS);
    static constexpr size_t BYTES = ELEMS * sizeof(T);
};

// columns [C, C1) of one run, one tangent direction per pass with a compile-time seed; results land in out[(c-C0)*NY + i]
template <class Fn, class T, int C0, int C, int C1> struct AdRun {
    static RD_DEV void run(const Fn& fn, const T (&x)[Fn::NX], const T (&u)[Fn::NU], T t, T h, bool term, T* out) {
        constexpr int NX = Fn::NX, NU = Fn::NU, NY = Fn::NY;
        if constexpr (C < C1) {
            using D = Dual<T, 1>;
            D xd[NX], ud[NU], yd[NY];
#pragma unroll
            for (int i = 0; i < NX; ++i) { xd[i].v = x[i]; xd[i].d[0] = (i == C) ? T(1) : T(0); }
#pragma unroll
            for (int i = 0; i < NU; ++i) { ud[i].v = u[i]; ud[i].d[0] = (NX + i == C) ? T(1) : T(0); }
            fn.eval(xd, ud, t, h, yd);
            const T sc = (C >= NX && term) ? T(0) : T(1);      // control columns of a terminal knot point are zero
#pragma unroll
            for (int i = 0; i < NY; ++i) out[(C - C0) * NY + i] = yd[i].d[0] * sc;
            AdRun<Fn, T, C0, C + 1, C1>::run(fn, x, u, t, h, term, out);
        }
    }
};

// transpose R values per lane (lane = knot point) through LDS slices of SL knot points and store them to
// dst[row * pitch + j], j < R, rows < nk: consecutive lanes write consecutive 8-byte (4-byte) elements.
template <class T, int R, int RS, int SL>
RD_DEV void wave_flush(T* lds, const T (&vals)[R], T* __restrict__ dst, int pitch, int nk, int lane) {
#pragma unroll
    for (int s = 0; s < TILE / SL; ++s) {
        wave_sync();
        if ((lane / SL) == s) {
#pragma unroll
            for (int j = 0; j < R; ++j) lds[(lane % SL) * RS + j] = vals[j];
        }
        wave_sync();
        const int rows = nk - s * SL;                      // valid rows of this slice (may be <= 0 or > SL)
        T* d = dst + (long long)s * SL * pitch;
#pragma unroll
        for (int e0 = 0; e0 < SL * R; e0 += TILE) {
            const int e = e0 + lane;
            const int row = e / R, col = e - row * R;
            if (e < SL * R && row < rows) d[row * pitch + col] = lds[row * RS + col];
        }
    }
}

template <class Fn, class T, class Plan, int SL, int G> struct WaveGroup {
    static constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY;
    using L = WaveLds<Fn, T, Plan, SL>;
    template <int RUN> static RD_DEV void runs(const Fn& fn, const T (&x)[NX], const T (&u)[NU], T t, T h, bool term, T* lds, T* __restrict__ Jt, int nk, int lane) {
        if constexpr (RUN < Plan::NRUNS) {
            constexpr int C0 = Plan::run_begin(G, RUN), C1 = Plan::run_end(G, RUN);
            if constexpr (C1 > C0) {
                T out[(C1 - C0) * NY];
                AdRun<Fn, T, C0, C0, C1>::run(fn, x, u, t, h, term, out);
                wave_flush<T, (C1 - C0) * NY, L::RS, SL>(lds, out, Jt + C0 * NY, NY * NZ, nk, lane);
            }
            runs<RUN + 1>(fn, x, u, t, h, term, lds, Jt, nk, lane);
        }
    }
    static RD_DEV void run(const Fn& fn, const T (&x)[NX], const T (&u)[NU], T t, T h, bool term, T* lds, T* __restrict__ Jt, T* __restrict__ Yt, int nk, int lane) {
        if constexpr (G == Plan::value_group) {
            if (Yt) {
                T y[NY];
                fn.eval(x, u, t, h, y);
                wave_flush<T, NY, L::RS, SL>(lds, y, Yt, NY, nk, lane);
            }
        }
        runs<0>(fn, x, u, t, h, term, lds, Jt, nk, lane);
    }
};

template <class Fn, class T, class Plan, int SL, int G> struct WaveDispatch {
    static RD_DEV void go(int g, const Fn& fn, const T (&x)[Fn::NX], const T (&u)[Fn::NU], T t, T h, bool term, T* lds, T* __restrict__ Jt, T* __restrict__ Yt, int nk, int lane) {
        if constexpr (G < Plan::NGROUPS) {
            if (g == G) WaveGroup<Fn, T, Plan, SL, G>::run(fn, x, u, t, h, term, lds, Jt, Yt, nk, lane);
            else WaveDispatch<Fn, T, Plan, SL, G + 1>::go(g, fn, x, u, t, h, term, lds, Jt, Yt, nk, lane);
        }
    }
};

template <class Fn, class T, class Plan, int SL, int OCC>
__global__ void __launch_bounds__(TILE, OCC) jac_ad_wave_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                           const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Y) {
    constexpr int NX = Fn::NX, NU = Fn::NU, NZ = NX + NU, NY = Fn::NY, NGR = Plan::NGROUPS;
    using L = WaveLds<Fn, T, Plan, SL>;
    static_assert(TILE % SL == 0, "slice must divide the wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // dynamic: L::BYTES (the compiler's static-LDS occupancy guess assumes 64 KB per CU)
    T* lds = reinterpret_cast<T*>(smem);
    const int lane = threadIdx.x;
    // bundle of 8*NGR consecutive blocks = 8 tiles x NGR groups; block j of a bundle: XCD x = j % 8 owns tile x of the bundle
    const long long b = blockIdx.x;
    const int j = (int)(b % (8 * NGR));
    const int g = j / 8;
    const long long tile = (b / (8 * NGR)) * 8 + (j % 8);
    const long long k0 = tile * TILE;
    if (k0 >= N) return;
    const int nk = (int)((N - k0) < TILE ? (N - k0) : TILE);
    const int kl = lane < nk ? lane : nk - 1;              // lanes past a ragged end recompute the last knot point (never stored)
    const double hd = dtt[k0 + kl];
    const T t = (T)tt[k0 + kl], h = (T)hd;
    const bool term = is_terminal_dt(hd);
    // ---- stage z: all global loads first, then slice by slice through LDS
    constexpr int ZPS = (SL * NZ + TILE - 1) / TILE;        // elements per lane per slice
    T zr[TILE / SL][ZPS];
#pragma unroll
    for (int s = 0; s < TILE / SL; ++s)
#pragma unroll
        for (int r = 0; r < ZPS; ++r) {
            const int e = r * TILE + lane;
            const long long ge = (k0 + s * SL) * NZ + e;
            zr[s][r] = (e < SL * NZ && ge < N * NZ) ? Z[ge] : T(0);
        }
    T x[NX], u[NU];
#pragma unroll
    for (int s = 0; s < TILE / SL; ++s) {
        wave_sync();
#pragma unroll
        for (int r = 0; r < ZPS; ++r) {
            const int e = r * TILE + lane;
            if (e < SL * NZ) { const int row = e / NZ, col = e - row * NZ; lds[row * L::ZS + col] = zr[s][r]; }
        }
        wave_sync();
        const int ks = kl - s * SL;                         // this lane's row in slice s, if it belongs to it
        if (ks >= 0 && ks < SL) {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = lds[ks * L::ZS + i];
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = term ? T(0) : lds[ks * L::ZS + NX + i];
        }
    }
    T* Jt = J + k0 * (NY * NZ);
    T* Yt = Y ? Y + k0 * NY : nullptr;
    WaveDispatch<Fn, T, Plan, SL, 0>::go(g, fn, x, u, t, h, term, lds, Jt, Yt, nk, lane);
}

}  // namespace rd
