// implicit.hpp -- DiscretizedDynamics{ImplicitMidpoint}: batched Newton solve and implicit-function-theorem Jacobian.
//   reference: src/integration.jl:620-729 (the scheme, its dynamics_error and Jacobians), :422-463 (Newton solve, StaticReturn flavour),
//              :517-538 (jacobian! by the implicit function theorem), :585-605 (defaults: 10 iterations, tolerance 1e-12)
//   e(x2; z1) = x1 + h f((x1 + x2)/2, u1, t + h/2) - x2        J2 = [(h/2) A - I, 0]        J1 = [I + (h/2) A, h B]        x_next: e = 0
//   discrete Jacobian = -J2[:, 1:n] \ J1,  [A B] = continuous Jacobian at (xmid, u1, t)
// Time arguments exactly as the reference passes them: the residual evaluates the dynamics at t + h/2 (src/integration.jl:640,659), its
// Jacobians evaluate the continuous Jacobian at t (src/integration.jl:684,716).  For time-invariant models (all of the reference's) the two
// evaluations are the same expression and the compiler merges them.
// One lane owns one knot point; its n x n Newton matrix lives in LDS (lane-private, odd stride -> conflict-free), because n*n
// doubles do not fit a lane's registers.  LU with partial pivoting like `lu`; right-hand sides stay in registers (compile-time
// indices, row swaps by select).  Continuous Jacobian columns are matrix-free forward-dual JVPs of the model.  This kernel is
// compute- and LDS-latency-bound (one wave per CU for n = 13 in fp64), not a streaming kernel.
#pragma once
#include "kernels.hpp"

namespace rd {

template <class Mdl, class T> struct ImplicitLds {
    static constexpr int NX = Mdl::NX;
    static constexpr int RSA = (NX * NX + NX) | 1;                  // per-lane stride (elements): the matrix + one right-hand side
    static constexpr size_t LANE_BYTES = (size_t)RSA * sizeof(T) + 16 * sizeof(int);   // + pivots
    // knot points per wave: a wave runs on one SIMD and is latency-bound, so what counts is how many knot points are resident per
    // CU.  When 64 lanes' matrices allow only one wave per CU (n = 13 in fp64: 93 KB), waves of KP < 64 knot points that fit TWO per CU
    // (on two SIMDs) hold 1.6x more knot points in flight; lanes >= KP idle.
    static constexpr int knots_per_wave() {
        if (2 * TILE * LANE_BYTES <= 160 * 1024) return TILE;
        const int kp = (int)((80 * 1024) / LANE_BYTES);
        return kp < TILE ? (kp < 1 ? 1 : kp) : TILE;
    }
    static constexpr int KP = knots_per_wave();
    static constexpr size_t BYTES = (size_t)KP * LANE_BYTES;
};

template <class Mdl, class T> struct ImplicitOps {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU;
    using D = Dual<T, 1>;
    // one column of the continuous Jacobian at time tj (direction e_c of z = [x; u])
    template <int C> static RD_DEV void jcol(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T (&col)[NX]) {
        D xd[NX], ud[NU], fd[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xd[i].v = xm[i]; xd[i].d[0] = (i == C) ? T(1) : T(0); }
#pragma unroll
        for (int i = 0; i < NU; ++i) { ud[i].v = u[i]; ud[i].d[0] = (NX + i == C) ? T(1) : T(0); }
        model_f(m, xd, ud, tj, fd);
#pragma unroll
        for (int i = 0; i < NX; ++i) col[i] = fd[i].d[0];
    }
    // A2 = (h/2) A - I into the lane's LDS matrix (column-major); f = the dynamics at (xm, u, tm) for the residual e = x1 + h f - x2
    template <int C> static RD_DEV void fill_cols(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T tm, T h, T* A, T (&f)[NX]) {
        if constexpr (C == 0) model_f(m, xm, u, tm, f);
        if constexpr (C < NX) {
            T col[NX];
            jcol<C>(m, xm, u, tj, col);
#pragma unroll
            for (int i = 0; i < NX; ++i) A[i + C * NX] = col[i] * (h * T(0.5)) - ((i == C) ? T(1) : T(0));
            fill_cols<C + 1>(m, xm, u, tj, tm, h, A, f);
        }
    }
    // in-place LU with partial pivoting of the lane's matrix; pivots into piv[0..NX).  Fully unrolled over the elimination step K: only
    // the pivot row index p is a run-time (per-lane) LDS address; the pivot row and column live in registers during the step, so the
    // update costs two LDS operations per FMA, all independent of each other (the run-time-indexed triple loop was LDS-latency-bound).
    template <int K> static RD_DEV void lu_step(T* A, int* piv) {
        if constexpr (K < NX) {
            T ck[NX];                                            // column K from row K down
#pragma unroll
            for (int i = K; i < NX; ++i) ck[i] = A[i + K * NX];
            int p = K; T best = ck[K] < T(0) ? -ck[K] : ck[K];
#pragma unroll
            for (int i = K + 1; i < NX; ++i) { const T v = ck[i] < T(0) ? -ck[i] : ck[i]; if (v > best) { best = v; p = i; } }
            { const int tmp = piv[K]; piv[K] = piv[p]; piv[p] = tmp; }   // piv[] holds the row permutation: row i of P*A is row piv[i] of A
            // swap rows K and p (all columns: the stored multipliers move with their rows); the new row K stays in registers
            T rk[NX];
            if (p != K) {
#pragma unroll
                for (int jj = 0; jj < NX; ++jj) { const T a = A[K + jj * NX], b = A[p + jj * NX]; A[K + jj * NX] = b; A[p + jj * NX] = a; rk[jj] = b; }
            } else {                                             // no exchange (the usual case for I - (h/2) A): only the part of row K that the step uses
                rk[K] = ck[K];
#pragma unroll
                for (int jj = K + 1; jj < NX; ++jj) rk[jj] = A[K + jj * NX];
            }
            const T inv = T(1) / rk[K];
            A[K + K * NX] = inv;                                  // the diagonal of U is kept as its reciprocal: the solves multiply instead of dividing
#pragma unroll
            for (int i = K + 1; i < NX; ++i) {
                const T l = A[i + K * NX] * inv;               // (re-read: row p's entry moved to row K's old place)
                A[i + K * NX] = l;
#pragma unroll
                for (int jj = K + 1; jj < NX; ++jj) A[i + jj * NX] -= l * rk[jj];
            }
            lu_step<K + 1>(A, piv);
        }
    }
    static RD_DEV void lu(T* A, int* piv) {
#pragma unroll
        for (int i = 0; i < NX; ++i) piv[i] = i;
        lu_step<0>(A, piv);
    }
    // b <- A^{-1} b with the factors above (unit-lower L, U with its diagonal stored as reciprocals); b in registers (compile-time indices).  The row permutation is applied through the
    // lane's LDS staging vector: a select chain on registers is turned into a dynamically indexed private array (scratch memory).
    static RD_DEV void solve(T* A, const int* piv, T (&b)[NX]) {
        T* sb = A + NX * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) sb[i] = b[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) b[i] = sb[piv[i]];
#pragma unroll
        for (int i = 1; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < i; ++j) b[i] -= A[i + j * NX] * b[j];
        }
#pragma unroll
        for (int i = NX - 1; i >= 0; --i) {
#pragma unroll
            for (int j = i + 1; j < NX; ++j) b[i] -= A[i + j * NX] * b[j];
            b[i] = b[i] * A[i + i * NX];                         // reciprocal of the pivot (stored by lu_step)
        }
    }
    // several right-hand sides at once: every factor entry is read from LDS once per group instead of once per column, and the G
    // substitution chains are independent of each other (the single-column solve is a chain of LDS round trips)
    template <int G> static RD_DEV void solve_multi(T* A, const int* piv, T (&b)[G][NX]) {
        T* sb = A + NX * NX;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < NX; ++i) sb[i] = b[g][i];
#pragma unroll
            for (int i = 0; i < NX; ++i) b[g][i] = sb[piv[i]];
        }
#pragma unroll
        for (int i = 1; i < NX; ++i) {
#pragma unroll
            for (int jj = 0; jj < i; ++jj) {
                const T a = A[i + jj * NX];
#pragma unroll
                for (int g = 0; g < G; ++g) b[g][i] -= a * b[g][jj];
            }
        }
#pragma unroll
        for (int i = NX - 1; i >= 0; --i) {
#pragma unroll
            for (int jj = i + 1; jj < NX; ++jj) {
                const T a = A[i + jj * NX];
#pragma unroll
                for (int g = 0; g < G; ++g) b[g][i] -= a * b[g][jj];
            }
            const T inv = A[i + i * NX];                         // reciprocal of the pivot (stored by lu_step)
#pragma unroll
            for (int g = 0; g < G; ++g) b[g][i] = b[g][i] * inv;
        }
    }
    // IFT columns: J[:, C] = -A2^{-1} J1[:, C],  J1 = [I + (h/2) A, h B]; groups of up to 4 columns per solve
    template <int C, int G> static RD_DEV void ift_rhs(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T h, T (&b)[G][NX]) {
        if constexpr (C % 4 < G) {      // (C - C0) < G, groups start at multiples of 4
            jcol<C>(m, xm, u, tj, b[C % 4]);
#pragma unroll
            for (int i = 0; i < NX; ++i) b[C % 4][i] = (C < NX) ? b[C % 4][i] * (h * T(0.5)) + ((i == C) ? T(1) : T(0)) : b[C % 4][i] * h;
            if constexpr (C % 4 + 1 < G) ift_rhs<C + 1, G>(m, xm, u, tj, h, b);
        }
    }
    template <int C> static RD_DEV void ift_cols(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T h, bool term, T* A, const int* piv, T* Jk, bool store) {
        if constexpr (C < NX + NU) {
            constexpr int G = (NX + NU - C) < 4 ? (NX + NU - C) : 4;
            T b[G][NX];
            ift_rhs<C, G>(m, xm, u, tj, h, b);
            solve_multi<G>(A, piv, b);
            if (store) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const T sc = (C + g >= NX && term) ? T(0) : T(-1);       // control columns of a terminal knot point are zero
#pragma unroll
                    for (int i = 0; i < NX; ++i) Jk[(C + g) * NX + i] = b[g][i] * sc;
                }
            }
            ift_cols<C + 4>(m, xm, u, tj, h, term, A, piv, Jk, store);
        }
    }
};

// The exact solve of knot points first .. first + count - 1 (count <= KPL <= 64 lanes of one wavefront; lane-private matrices at lds, pivots behind
// KPL of them): Newton's method with real row exchanges, and with WANT_J the implicit-function-theorem Jacobian at the last TESTED iterate.
template <class Mdl, class T, bool WANT_J>
RD_DEV void implicit_exact_knots(const Mdl& m, long long N, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt, T* __restrict__ Xn,
                                 T* __restrict__ J, int newton_iters, double tol, long long first, int count, unsigned char* lds, int KPL) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NZ = NX + NU;
    using L = ImplicitLds<Mdl, T>;
    using Ops = ImplicitOps<Mdl, T>;
    const int lane = threadIdx.x & 63;
    if (lane >= count) return;                                           // lanes without a matrix slot stay idle
    T* A = reinterpret_cast<T*>(lds) + lane * L::RSA;
    int* piv = reinterpret_cast<int*>(lds + (size_t)KPL * L::RSA * sizeof(T)) + lane * 16;
    const long long k = first + lane;
    const bool active = k < N;
    const long long kc = active ? k : N - 1;
    const double hd = dtt[kc];
    const T h = (T)hd, tj = (T)tt[kc], tm = tj + h * T(0.5);
    const bool term = is_terminal_dt(hd);
    T x1[NX], u[NU], xn[NX], xl[NX], xm[NX], f[NX], r[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { x1[i] = Z[kc * NZ + i]; xn[i] = x1[i]; xl[i] = x1[i]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = term ? T(0) : Z[kc * NZ + NX + i];
    bool done = false;
    for (int it = 0; it < newton_iters; ++it) {
        if (!done) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { xl[i] = xn[i]; xm[i] = (x1[i] + xn[i]) * T(0.5); }
        }
        // residual and Newton matrix at the last tested iterate xl (recomputed, unchanged, for lanes that are already done)
#pragma unroll
        for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xl[i]) * T(0.5);
        Ops::template fill_cols<0>(m, xm, u, tj, tm, h, A, f);
        T nr = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { r[i] = x1[i] + h * f[i] - xl[i]; nr += r[i] * r[i]; }
        if ((double)sqrt_(nr) < tol) done = true;
        if (__all(done)) break;
        Ops::lu(A, piv);
        Ops::solve(A, piv, r);
        if (!done) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = xl[i] - r[i];
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Xn[k * NX + i] = xn[i];
    }
    if constexpr (WANT_J) {
        // Jacobians at the last TESTED iterate (what the reference's cache holds when the loop exits)
#pragma unroll
        for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xl[i]) * T(0.5);
        Ops::template fill_cols<0>(m, xm, u, tj, tm, h, A, f);
        Ops::lu(A, piv);
        Ops::template ift_cols<0>(m, xm, u, tj, h, term, A, piv, J + kc * NX * NZ, active);
    }
}

// gate == NULL: every block of KP knot points is solved.  gate != NULL (second pass behind the register kernel): tile_flags = 0 -> the whole
// batch iff gate[0] is set; tile_flags = 1 -> only the blocks that overlap a 64-knot tile whose word gate[1 + tile] the register kernel raised
// (a row exchange somewhere no longer sends the whole batch through this 12x slower kernel).  Persistent grid: with nothing raised the pass
// costs a launch of <= one workgroup per CU, not a sweep of N / KP workgroups with 93 KB of LDS each.
template <class Mdl, class T, bool WANT_J>
__global__ void __launch_bounds__(TILE) implicit_midpoint_kernel(Mdl m, long long N, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                                 T* __restrict__ Xn, T* __restrict__ J, int newton_iters, double tol, const int* __restrict__ gate, int tile_flags) {
    using L = ImplicitLds<Mdl, T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // tile_flags: bit 0 = one flag word per 64-knot tile (gate[1 + tile]) instead of the single word gate[0]; bits 8.. = the call's epoch (0: none):
    // with tile words, gate[0] holds the epoch of the last call in which ANY tile was raised, so a pass with nothing to do ends after one load
    // instead of scanning its tile words (the register kernels store the epoch next to the tile word they raise; no clearing, no memset)
    const int epoch = tile_flags >> 8;
    tile_flags &= 1;
    if (gate && !tile_flags && *gate == 0) return;                       // second pass behind the register kernel: only if a row exchange was needed
    if (gate && tile_flags && epoch && *gate != epoch) return;
    const int lane = threadIdx.x;
    const long long nblk = (N + L::KP - 1) / L::KP;
    for (long long base = blockIdx.x; base < nblk; base += (long long)TILE * gridDim.x) {
    // the workgroup's next 64 blocks are base + l * gridDim.x: lane l looks at the tile words of block l, ONE memory round trip decides all of them
    unsigned long long todo = ~0ull;
    if (gate && tile_flags) {
        const long long b = base + (long long)lane * gridDim.x;
        int raised = 0;
        if (b < nblk) {
            const long long first = b * L::KP, lastk = (first + L::KP < N ? first + L::KP : N) - 1;
            raised = gate[1 + first / TILE] | gate[1 + lastk / TILE];
        }
        todo = __ballot(raised != 0);
    }
    for (int l = 0; l < TILE; ++l) {
    const long long blk = base + (long long)l * gridDim.x;
    if (blk >= nblk) break;
    if (!((todo >> l) & 1)) continue;
    implicit_exact_knots<Mdl, T, WANT_J>(m, N, Z, tt, dtt, Xn, J, newton_iters, tol, blk * L::KP, L::KP, smem, L::KP);
    }
    }
}

// ---- the Newton matrix in REGISTERS --------------------------------------------------------------------------------------------
// With the continuous Jacobian columns produced by compile-time tangent seeds, the compiler knows the sparsity pattern of (h/2) A - I
// and of its LU fill-in: a fully unrolled factorisation WITHOUT row exchanges keeps only the structurally non-zero entries (the
// satellite's 13 x 13 matrix fits 238 VGPRs, no LDS, two waves per SIMD) and skips the arithmetic on the zeros -- an order of magnitude
// faster than the LDS version above.  For n > 6 partial pivoting is reproduced exactly as long as it would not have exchanged rows (|pivot| >=
// every entry below it, the usual case for I - (h/2) A); every lane checks that, and if any lane of the launch fails the check a flag
// is raised and the LDS kernels above (launched behind this one, gated on the flag) recompute the whole batch with real row exchanges.
// ---- Newton's method along the cascade ---------------------------------------------------------------------------------------------------
// For a model with a block structure (Cascade<>: block b's derivative reads blocks 0..b only) the Newton matrix (h/2) A - I is block lower
// triangular, and so is the implicit equation itself: block 0 of x_next solves its own small system, block 1 follows from it, and so on.  The
// full Newton iteration of the reference (src/integration.jl:422-463) converges to the same root -- its block-0 iterates ARE the iterates of
// block 0 alone, the later blocks trail by the linearisation error of the earlier ones -- but solving block by block costs a 3 x 3 iteration for
// a rigid body's angular velocity, ONE 4 x 4 solve for its attitude (the kinematics are linear in q) and one 6 x 6 solve for r, v instead of
// 3-4 factorisations of a 13 x 13 matrix.  What has to stay exact is the reference's behaviour when its loop ends at the iteration limit: it then returns an
// iterate that is NOT the root (and takes the Jacobian one update earlier), so a lane is handed to the exact kernel (`hand_over`) whenever the
// reference's full iteration could have run into the limit: its convergence test passes at the test at which block 0 passes or up to NB - 1
// tests later, and block 0's own test count is known here.
template <class Cas, int NX> struct BlockIndex {
    static constexpr int count(int b) { int c = 0; for (int i = 0; i < NX; ++i) c += Cas::block_of(i) == b ? 1 : 0; return c; }
    static constexpr int nth(int b, int j) { int c = 0; for (int i = 0; i < NX; ++i) if (Cas::block_of(i) == b) { if (c == j) return i; ++c; } return 0; }
};
// LU of a block WITHOUT row exchanges (the compile-time sparsity of the block survives: a world-frame r, v block is [[-I, (h/2) I], [0, -I]] and costs
// nothing), pivots stored as reciprocals; returns true if partial pivoting would have exchanged rows somewhere (the lane then goes to the exact kernel,
// like the lanes of the 13 x 13 factorisation below)
template <class T, int N> RD_DEV bool lu_small(T (&a)[N][N]) {
    bool need = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const T pk = a[k][k] < T(0) ? -a[k][k] : a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) { const T v = a[i][k] < T(0) ? -a[i][k] : a[i][k]; need = need || (v > pk); }
        const T inv = T(1) / a[k][k];
        a[k][k] = inv;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = a[i][k] * inv; a[i][k] = l;
#pragma unroll
            for (int j = k + 1; j < N; ++j) a[i][j] -= l * a[k][j];
        }
    }
    return need;
}
template <class T, int N> RD_DEV void solve_small(const T (&a)[N][N], T (&b)[N]) {
#pragma unroll
    for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) b[i] -= a[i][j] * b[j];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
#pragma unroll
        for (int j = i + 1; j < N; ++j) b[i] -= a[i][j] * b[j];
        b[i] = b[i] * a[i][i];
    }
}
// The reference's tolerance (1e-12 by default) is below what fp32 can resolve: its loop then simply runs to the iteration limit and returns the
// root to rounding.  The block iterations test against max(tol, 32 eps_T max(1, |x|)) instead -- unchanged in fp64 (7e-15 |x|), a rounding-level
// floor in fp32 -- so that an fp32 batch converges in 2-3 iterations and is not handed to the exact kernel for a tolerance nobody can meet.
template <class T> RD_DEV double block_tolerance(double tol, T xmax) {
    const double floor_ = (sizeof(T) == 4 ? 32.0 * 1.1920928955078125e-07 : 32.0 * 2.220446049250313e-16) * (double)(xmax > T(1) ? xmax : T(1));
    return tol > floor_ ? tol : floor_;
}
template <class Mdl, class T, class Cas> struct CascadeNewton {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NB = Cas::NB;
    using BI = BlockIndex<Cas, NX>;
    using Ops = ImplicitOps<Mdl, T>;
    // column J of block B of (h/2) A - I, rows of block B only (compile-time recursion over J; the other rows of the tangent fold away)
    template <int B, int J, int NBK> static RD_DEV void block_cols(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T h, T (&a)[NBK][NBK]) {
        if constexpr (J < NBK) {
            constexpr int C = BI::nth(B, J);
            T col[NX];
            Ops::template jcol<C>(m, xm, u, tj, col);
#pragma unroll
            for (int i = 0; i < NBK; ++i) a[i][J] = col[BI::nth(B, i)] * (h * T(0.5)) - ((i == J) ? T(1) : T(0));
            block_cols<B, J + 1, NBK>(m, xm, u, tj, h, a);
        }
    }
    // block B of x_next (upstream blocks of xn final, this and the downstream ones still x1); tests = residual evaluations until ||r_B|| < tol
    template <int B> static RD_DEV void solve_from(const Mdl& m, const T (&x1)[NX], const T (&u)[NU], T tj, T h, int iters, double tol, T (&xn)[NX], int& tests0, bool& open) {
        if constexpr (B < NB) {
            constexpr int NBK = BI::count(B);
            const T tm = tj + h * T(0.5);
            bool done = false;
            int tests = 0;
            T xmax = T(0);
#pragma unroll
            for (int i = 0; i < NBK; ++i) { const T v = x1[BI::nth(B, i)]; const T av = v < T(0) ? -v : v; xmax = av > xmax ? av : xmax; }
            const double tolb = block_tolerance<T>(tol, xmax);
            for (int it = 0; it < iters; ++it) {
                T xm[NX], f[NX], r[NBK];
#pragma unroll
                for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xn[i]) * T(0.5);
                model_f(m, xm, u, tm, f);
                T nr = T(0);
#pragma unroll
                for (int i = 0; i < NBK; ++i) { const int gi = BI::nth(B, i); r[i] = x1[gi] + h * f[gi] - xn[gi]; nr += r[i] * r[i]; }
                if (!done) ++tests;
                if ((double)sqrt_(nr) < tolb) done = true;
                if (__all(done)) break;
                T a[NBK][NBK];
                block_cols<B, 0, NBK>(m, xm, u, tj, h, a);
                const bool need = lu_small<T, NBK>(a);
                if (!done) open = open || need;
                solve_small<T, NBK>(a, r);
                if (!done) {
#pragma unroll
                    for (int i = 0; i < NBK; ++i) xn[BI::nth(B, i)] -= r[i];
                }
            }
            if (B == 0) tests0 = tests;
            open = open || !done;
            solve_from<B + 1>(m, x1, u, tj, h, iters, tol, xn, tests0, open);
        }
    }
    // returns true if the lane has to be redone by the exact kernel (see above)
    static RD_DEV bool newton(const Mdl& m, const T (&x1)[NX], const T (&u)[NU], T tj, T h, int iters, double tol, T (&xn)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = x1[i];
        int tests0 = 0; bool open = false;
        // every block meets the same norm as the reference's full residual would: ||r||^2 = sum of the blocks' < NB (tol / sqrt(NB))^2
        solve_from<0>(m, x1, u, tj, h, iters, tol * 0.5773502691896258, xn, tests0, open);
        return open || (tests0 + NB - 1 > iters);
    }
};

// rollout!(DiscretizedDynamics{ImplicitMidpoint}) along the cascade: the function object rollout_cascade_kernel (cascade.hpp) steps with.  The
// wavefront of block B solves ITS block of the implicit equation per step -- CascadeNewton's block iteration with the upstream blocks' midpoints
// spliced in by the hook and its own midpoint published for the blocks downstream (the midpoint is the scheme's one "stage point") -- so the
// three blocks of a rigid body run as a pipeline skewed by one step, exactly like the explicit rules: the chain per step is the longest block's
// solve, not the sum.  A lane the block iteration cannot vouch for (CascadeNewton) raises the context's flag word and the exact kernel redoes
// the batch behind this one.
template <class Mdl> struct ImplicitRollFun {
    using T = typename Mdl::T;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, RULE_ID = RULE_IMPLICIT_MIDPOINT;
    Mdl m;
    int iters; double tol;
    int* flag;
    template <class Cas, int B, class Hook> RD_DEV void block_step(const T (&x)[NX], const T* uu, T t, T h, T (&y)[NX], Hook& hk) const {
        using BI = BlockIndex<Cas, NX>;
        using CN = CascadeNewton<Mdl, T, Cas>;
        constexpr int NBK = BI::count(B), NB = Cas::NB;
        T xmax = T(0);
#pragma unroll
        for (int i = 0; i < NBK; ++i) { const T v = x[BI::nth(B, i)]; const T av = v < T(0) ? -v : v; xmax = av > xmax ? av : xmax; }
        const double tolb = block_tolerance<T>(tol * (NB == 2 ? 0.7071067811865476 : NB == 3 ? 0.5773502691896258 : NB == 4 ? 0.5 : 0.4082482904638631), xmax);   // tol / sqrt(NB), NB <= 6
        const T tm = t + h * T(0.5);
        T u[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = uu[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = x[i];
        bool done = false, open = false;
        int tests = 0;
        T xs[NX];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = (x[i] + y[i]) * T(0.5);
            hk.template at<0>(xs);                        // upstream midpoints in, this block's midpoint out
            T f[NX], r[NBK];
            model_f(m, xs, u, tm, f);
            T nr = T(0);
#pragma unroll
            for (int i = 0; i < NBK; ++i) { const int gi = BI::nth(B, i); r[i] = x[gi] + h * f[gi] - y[gi]; nr += r[i] * r[i]; }
            if (!done) ++tests;
            if ((double)sqrt_(nr) < tolb) done = true;
            if (__all(done)) break;
            T a[NBK][NBK];
            CN::template block_cols<B, 0, NBK>(m, xs, u, t, h, a);
            const bool need = lu_small<T, NBK>(a);
            if (!done) open = open || need;
            solve_small<T, NBK>(a, r);
            if (!done) {
#pragma unroll
                for (int i = 0; i < NBK; ++i) y[BI::nth(B, i)] -= r[i];
            }
        }
        if (!__all(done)) {                               // the limit ended the loop behind an update: the published midpoint is one update old
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = (x[i] + y[i]) * T(0.5);
            hk.template at<0>(xs);
        }
        open = open || !done || (B == 0 && tests + NB - 1 > iters);
        if (open && flag) *flag = 1;
    }
};

template <class Mdl, class T> struct ImplicitRegOps {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU;
    using Ops = ImplicitOps<Mdl, T>;
    template <int C> static RD_DEV void fill(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T tm, T h, T (&a)[NX][NX], T (&f)[NX]) {
        if constexpr (C == 0) model_f(m, xm, u, tm, f);
        if constexpr (C < NX) {
            T col[NX];
            Ops::template jcol<C>(m, xm, u, tj, col);
#pragma unroll
            for (int i = 0; i < NX; ++i) a[i][C] = col[i] * (h * T(0.5)) - ((i == C) ? T(1) : T(0));
            fill<C + 1>(m, xm, u, tj, tm, h, a, f);
        }
    }
    // Small matrices (n <= 6: cartpole, double integrators, the linear test systems) carry partial pivoting in registers: the row exchange
    // is a select per element and candidate row (~n^3 selects), the compile-time sparsity is lost but 36 entries fit anyway, and no second
    // pass is ever needed.  (The cartpole's Newton matrix does exchange rows for ordinary inputs.)
    static constexpr bool PIVOT = NX <= 6;
    // LU, pivots stored as reciprocals.  PIVOT: with row exchanges recorded in piv.  Otherwise: without exchanges; returns true if partial
    // pivoting would have exchanged rows somewhere.
    static RD_DEV bool lu(T (&a)[NX][NX], int (&piv)[NX]) {
        bool need = false;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            T pk = a[k][k] < T(0) ? -a[k][k] : a[k][k];
            if constexpr (PIVOT) {
                int p = k;
#pragma unroll
                for (int i = k + 1; i < NX; ++i) { const T v = a[i][k] < T(0) ? -a[i][k] : a[i][k]; if (v > pk) { pk = v; p = i; } }
                piv[k] = p;
#pragma unroll
                for (int i = k + 1; i < NX; ++i) {
                    const bool sw = (p == i);
#pragma unroll
                    for (int j = 0; j < NX; ++j) { const T x = a[k][j], y = a[i][j]; a[k][j] = sw ? y : x; a[i][j] = sw ? x : y; }
                }
            } else {
#pragma unroll
                for (int i = k + 1; i < NX; ++i) { const T v = a[i][k] < T(0) ? -a[i][k] : a[i][k]; need = need || (v > pk); }
            }
            const T inv = T(1) / a[k][k];
            a[k][k] = inv;
#pragma unroll
            for (int i = k + 1; i < NX; ++i) {
                const T l = a[i][k] * inv; a[i][k] = l;
#pragma unroll
                for (int j = k + 1; j < NX; ++j) a[i][j] -= l * a[k][j];
            }
        }
        return need;
    }
    static RD_DEV void solve(const T (&a)[NX][NX], const int (&piv)[NX], T (&b)[NX]) {
        if constexpr (PIVOT) {                                    // the recorded exchanges, in order, on the right-hand side
#pragma unroll
            for (int k = 0; k < NX; ++k) {
#pragma unroll
                for (int i = k + 1; i < NX; ++i) { const bool sw = (piv[k] == i); const T x = b[k], y = b[i]; b[k] = sw ? y : x; b[i] = sw ? x : y; }
            }
        }
#pragma unroll
        for (int i = 1; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < i; ++j) b[i] -= a[i][j] * b[j];
        }
#pragma unroll
        for (int i = NX - 1; i >= 0; --i) {
#pragma unroll
            for (int j = i + 1; j < NX; ++j) b[i] -= a[i][j] * b[j];
            b[i] = b[i] * a[i][i];
        }
    }
    // one Newton solve from x1 (src/integration.jl:422-463): xn = solution, xl = last tested iterate; true if a row exchange was needed
    // (*unconv = the lane left the loop at the iteration limit, i.e. xn is one update past the last tested iterate xl)
    static RD_DEV bool newton(const Mdl& m, const T (&x1)[NX], const T (&u)[NU], T tj, T h, int iters, double tol, T (&xn)[NX], T (&xl)[NX], bool* unconv = nullptr) {
        if constexpr (Cascade<Mdl>::NB > 1 && !Tune::NO_CASCADE_NEWTON) {          // block by block (CascadeNewton above); a lane it cannot vouch for goes to the exact kernel
            const bool hand_over = CascadeNewton<Mdl, T, Cascade<Mdl>>::newton(m, x1, u, tj, h, iters, tol, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) xl[i] = xn[i];
            if (unconv) *unconv = false;
            return hand_over;
        }
        const T tm = tj + h * T(0.5);
        T xm[NX], f[NX], r[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xn[i] = x1[i]; xl[i] = x1[i]; }
        bool done = false, need = false;
        for (int it = 0; it < iters; ++it) {
            if (!done) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xl[i] = xn[i];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xl[i]) * T(0.5);
            T a[NX][NX];
            fill<0>(m, xm, u, tj, tm, h, a, f);
            T nr = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) { r[i] = x1[i] + h * f[i] - xl[i]; nr += r[i] * r[i]; }
            if ((double)sqrt_(nr) < tol) done = true;
            if (__all(done)) break;
            int piv[NX];
            need = lu(a, piv) || need;
            solve(a, piv, r);
            if (!done) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xn[i] = xl[i] - r[i];
            }
        }
        if (unconv) *unconv = !done;
        return need;
    }
    // IFT columns with the register factors: J[:, C] = -A2^{-1} J1[:, C]
    // Finished columns go through a wave-private LDS buffer, CG = 4 columns at a time, so that the global stores are coalesced: a lane owns
    // a knot point, i.e. its column entries sit n(n+m) values apart from its neighbour's -- written directly, every store instruction touches
    // 64 different cache lines.  Transposed, a knot point's CG columns are one run of CG n consecutive values (416 B for a rigid body) and a
    // store instruction writes 16 B per lane over 2-3 such runs (with one column per sweep and 8 B per lane the kernel issued 304 store
    // instructions per tile over 104-B runs, half of its time; now 130).  Rows are only sizeof(T)-aligned: the 16-B stores may be misaligned
    // by 8 B, which gfx950 global memory accepts.  J0 = the wave's first Jacobian, nk = its number of valid knot points.
    // columns per LDS group: 4 (a quarter of a CU's LDS for four waves of a 13-state model) -- or ALL of them when the whole n x (n+m) slab of
    // 64 knot points is small (cartpole: 10 KB per wave): the wave's Jacobians are then one contiguous piece of memory written as whole lines
    // (a 4-of-5-column group leaves every 160-byte slab of the cartpole with partially written 128-byte lines, see implicit_ift_split_kernel)
    static constexpr int CG = (NX * (NX + NU) <= 96) ? (NX + NU) : 4;
    static constexpr int RS = (CG * NX) | 1;                          // lane stride of the group buffer (odd: conflict-free lane-strided writes)
    static constexpr size_t LDS_BYTES = (size_t)TILE * RS * sizeof(T);
    template <int C> static RD_DEV void ift_cols(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T h, bool term, const T (&a)[NX][NX], const int (&piv)[NX],
                                                 T* sb, T* J0, int nk) {
        if constexpr (C < NX + NU) {
            constexpr int G = (NX + NU - C) < CG ? (NX + NU - C) : CG;       // columns of this group
            const int lane = threadIdx.x;
            group_col<C, 0, G>(m, xm, u, tj, h, term, a, piv, sb + lane * RS);
            wave_sync();
            // sweep: knot point kk's run of G*NX values starts at J0 + kk * NX * (NX + NU) + C * NX; V values (16 bytes) per lane
            constexpr int V = 16 / (int)sizeof(T), RUN = G * NX, UNITS = RUN / V, TAIL = RUN - UNITS * V;
            typedef T VT __attribute__((ext_vector_type(16 / sizeof(T)), aligned(sizeof(T))));
            constexpr int NIT = (TILE * UNITS + TILE - 1) / TILE;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = it * TILE + lane, kk = e / UNITS, un = e - kk * UNITS;
                if (kk < nk) {
                    VT v;
#pragma unroll
                    for (int q = 0; q < V; ++q) v[q] = sb[kk * RS + un * V + q];
                    *reinterpret_cast<VT*>(J0 + (size_t)kk * NX * (NX + NU) + C * NX + un * V) = v;
                }
            }
            if constexpr (TAIL > 0) {                                 // the values behind the last whole 16-byte unit of every run
                for (int e = lane; e < nk * TAIL; e += TILE) {
                    const int kk = e / TAIL, q = e - kk * TAIL;
                    J0[(size_t)kk * NX * (NX + NU) + C * NX + UNITS * V + q] = sb[kk * RS + UNITS * V + q];
                }
            }
            wave_sync();
            ift_cols<C + G>(m, xm, u, tj, h, term, a, piv, sb, J0, nk);
        }
    }
    // column C + g of the group into the lane's row of the buffer (compile-time recursion over g)
    template <int C, int g, int G> static RD_DEV void group_col(const Mdl& m, const T (&xm)[NX], const T (&u)[NU], T tj, T h, bool term, const T (&a)[NX][NX],
                                                                const int (&piv)[NX], T* row) {
        if constexpr (g < G) {
            T col[NX];
            Ops::template jcol<C + g>(m, xm, u, tj, col);
#pragma unroll
            for (int i = 0; i < NX; ++i) col[i] = (C + g < NX) ? col[i] * (h * T(0.5)) + ((i == C + g) ? T(1) : T(0)) : col[i] * h;
            solve(a, piv, col);
            const T sc = (C + g >= NX && term) ? T(0) : T(-1);       // control columns of a terminal knot point are zero
#pragma unroll
            for (int i = 0; i < NX; ++i) row[g * NX + i] = col[i] * sc;
            __builtin_amdgcn_sched_barrier(0);                    // one column at a time: interleaved columns cost the second wave per SIMD
            group_col<C, g + 1, G>(m, xm, u, tj, h, term, a, piv, row);
        }
    }
};

template <class Mdl, class T, bool WANT_J>
__global__ void __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu(2))) implicit_midpoint_reg_kernel(Mdl m, long long N, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                                     T* __restrict__ Xn, T* __restrict__ J, int newton_iters, double tol, int* __restrict__ flag, int tile_flags) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NZ = NX + NU;
    using R = ImplicitRegOps<Mdl, T>;
    const long long k = (long long)blockIdx.x * TILE + threadIdx.x;
    const bool active = k < N;
    const long long kc = active ? k : N - 1;
    const double hd = dtt[kc];
    const T h = (T)hd, tj = (T)tt[kc];
    const bool term = is_terminal_dt(hd);
    T x1[NX], u[NU], xn[NX], xl[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x1[i] = Z[kc * NZ + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = term ? T(0) : Z[kc * NZ + NX + i];
    bool unconv = false;
    bool need = R::newton(m, x1, u, tj, h, newton_iters, tol, xn, xl, &unconv);
    // tile_flags bit 1 (set when implicit_ift_split_kernel follows, which linearises about Xn): a lane that stopped at the iteration limit has
    // xn != its last tested iterate -- where the reference's Jacobian is taken -- so its tile goes to the exact kernel as well
    if (!WANT_J && (tile_flags & 2)) need = need || unconv;
    if (active) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Xn[k * NX + i] = xn[i];
    }
    if constexpr (WANT_J) {
        // Jacobians at the last TESTED iterate (what the reference's cache holds when the loop exits)
        T xm[NX], f[NX], a[NX][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xl[i]) * T(0.5);
        int piv[NX];
        R::template fill<0>(m, xm, u, tj, tj + h * T(0.5), h, a, f);
        need = R::lu(a, piv) || need;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem_reg[];
        const long long k0 = (long long)blockIdx.x * TILE;
        R::template ift_cols<0>(m, xm, u, tj, h, term, a, piv, reinterpret_cast<T*>(smem_reg), J + k0 * NX * NZ, (int)((N - k0) < TILE ? (N - k0) : TILE));
    }
    // "partial pivoting would have exchanged rows": per 64-knot tile (word 1 + tile, written unconditionally: no memset in front of the
    // launch) or, for batches with more tiles than the context holds words for, the single word 0 (cleared by the host)
    if (flag) {
        if (tile_flags & 1) {
            const bool any = __any(need);
            if (threadIdx.x == 0) { flag[1 + blockIdx.x] = any ? 1 : 0; if (any && (tile_flags >> 8)) flag[0] = tile_flags >> 8; }      // (+ the call's epoch, see implicit_midpoint_kernel)
        }
        else if (need) *flag = 1;
    }
}

// ---- implicit-function-theorem Jacobian as a split-role tile kernel ------------------------------------------------------------------
// The register kernel above stores a lane's columns in 416-byte pieces of a 1 976-byte slab, five visits per slab spread over the kernel:
// every visit leaves partially written 128-byte lines behind.  While the output still sits in the Infinity Cache from an earlier call that
// goes unnoticed (36 us for 65 536 satellite knot points); written to memory that is not cached -- outputs rotated over four 136 MB buffers --
// the same kernel takes 59 us (L2 -> fabric traffic is 1.11x the algorithmic bytes either way; the partial lines pay at the memory side).
// So for models whose tile fits, the Jacobian goes through the flagship's loop instead (kernels.hpp jac_split_loop): the Newton solve runs
// first as implicit_midpoint_reg_kernel<.., false> (x_next to Y), then this kernel -- four compute waves per 64-knot tile, every one
// factorising (h/2) A - I at the midpoint itself (compile-time sparse, without row exchanges: cheap next to the solves) and solving its own
// share of the columns; copy waves write the finished tile as whole lines.  State columns use  -A2^{-1} (I + (h/2) A) = -I - 2 A2^{-1}
// (A2 = (h/2) A - I): the right-hand side is a unit vector, no second Jacobian-vector product.  x_next reaches the lanes as the loop's
// lane-private auxiliary input, one tile ahead.  A lane whose factorisation would have exchanged rows raises its tile's flag word like the
// register kernel does and the LDS kernel behind recomputes that tile.
template <class Mdl> struct IftFun {
    using T = typename Mdl::T;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = Mdl::NX, NCOLS = Mdl::NX + Mdl::NU, NVAL = Mdl::NX, NEXT = 0;
    Mdl m;
    double tol;          // Newton: stop when ||residual||_2 < tol ...
    int* flag;           // word 1 + tile: some lane of the tile needs the exact solve (row exchange / iteration limit); nullptr: never recomputed
    int iters;           // ... or after this many iterations
    int solve;           // 2: the helper wave runs the Newton solve one tile ahead (one launch for x_next + Jacobian); 1: the compute waves do; 0: x_next is read from xn
    const T* xn;         // solve == 0: the Newton solutions (n x N) of an earlier kernel
};
template <class Fn, class T, class Plan, int G, int C> struct IftCols {
    using Mdl = decltype(Fn::m);
    using R = ImplicitRegOps<Mdl, T>;
    static constexpr int NX = Fn::NX, NU = Fn::NU;
    static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr, const T (&xm)[NX], const T (&a)[NX][NX], const int (&piv)[NX]) {
        if constexpr (C < NX + NU) {
            if constexpr (FirstSplitOf<Plan>::N > 0 && C == FirstSplitOf<Plan>::N) {
                if (fr.first) __syncthreads();              // barrier E of the split-role loop: the early part of the first tile is complete
            }
            if constexpr (Plan::wave_of(C) == G) {
                T col[NX];
                if constexpr (C < NX) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = (i == C) ? T(1) : T(0);
                    R::solve(a, piv, col);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = col[i] * T(-2) - ((i == C) ? T(1) : T(0));
                } else {
                    ImplicitOps<Mdl, T>::template jcol<C>(fn.m, xm, fr.u, fr.t, col);
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = col[i] * fr.h;
                    R::solve(a, piv, col);
                    const T sc = fr.term ? T(0) : T(-1);             // control columns of a terminal knot point are zero
#pragma unroll
                    for (int i = 0; i < NX; ++i) col[i] = col[i] * sc;
                }
                fr.put_at(C, col);
                __builtin_amdgcn_sched_barrier(0);                    // one column at a time (register pressure next to the factors)
            }
            IftCols<Fn, T, Plan, G, C + 1>::run(fn, fr, xm, a, piv);
        }
    }
};
template <class Fn, class T, class Plan> struct IftBody {
    static constexpr bool EARLY = true;
    using Mdl = decltype(Fn::m);
    using R = ImplicitRegOps<Mdl, T>;
    static constexpr int NX = Fn::NX, NU = Fn::NU;
    template <int G> static RD_DEV void waves(const Fn& fn, JacFrame<Fn, T>& fr, const T (&xm)[NX], const T (&a)[NX][NX], const int (&piv)[NX]) {
        if constexpr (G < Plan::NG) {
            if (fr.g == G) IftCols<Fn, T, Plan, G, 0>::run(fn, fr, xm, a, piv);
            else waves<G + 1>(fn, fr, xm, a, piv);
        }
    }
    template <int CH> static RD_DEV void run(const Fn& fn, JacFrame<Fn, T>& fr) {
        T xm[NX], f[NX], a[NX][NX];
        int piv[NX];
        // x_next: every compute wave solves its 64 knot points itself (block by block along the cascade where the model has one: ~1 000 dependent
        // fp64 instructions, 3 us -- under the copy waves' store stream of the previous tile from the second tile on).  A separate Newton kernel in
        // front cost its launch, 7 us with nothing to overlap, and a round trip of x_next through memory.
        T xn[NX], xl[NX];
        bool bad = false;
        if (Tune::IFT_ABL == 2) {                                    // tuning only: no Newton solve at all (x_next = x): what bounds the kernel besides the helper wave's chain?
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = fr.x[i];
        } else if (fn.solve == 2) {                                         // the helper wave solved this tile while the previous one was being computed
            const T* hx = reinterpret_cast<const T*>(fr.helper);
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = hx[fr.kl * NX + i];
            bad = *reinterpret_cast<const int*>(fr.helper + (size_t)TILE * NX * sizeof(T)) != 0;
        } else if (fn.solve) {
            bool unconv = false;
            bad = R::newton(fn.m, fr.x, fr.u, fr.t, fr.h, fn.iters, fn.tol, xn, xl, &unconv);
            bad = bad || unconv;                                     // stopped at the iteration limit: the Jacobian belongs at the last tested iterate (exact solve)
        } else {
            const long long k = fr.tile * TILE + fr.kl;
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = fn.xn[k * NX + i];
        }
        if (fr.g == Plan::value_wave && fr.want_y) fr.put_value(NX + NU, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i) xm[i] = (fr.x[i] + xn[i]) * T(0.5);
        if constexpr (Tune::IFT_ABL == 1) {                          // tuning only: what would the kernel cost if (h/2) A - I came for free?  (a = -I: fill and factorisation fold away)
#pragma unroll
            for (int i = 0; i < NX; ++i) { f[i] = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) a[i][j] = (i == j) ? T(-1) : T(0); }
        } else R::template fill<0>(fn.m, xm, fr.u, fr.t, fr.t + fr.h * T(0.5), fr.h, a, f);
        const bool need = R::lu(a, piv) || bad;
        if (fr.g == 0 && fn.flag) {                                  // every compute wave sees the same knot points: one of them reports (unconditionally: nothing is ever cleared)
            const bool any = __any(need);
            fr.epi = fr.epi || any;
            if ((threadIdx.x & 63) == 0 && (fn.solve || any)) fn.flag[1 + fr.tile] = any ? 1 : 0;      // (solve == 0: the Newton kernel wrote the word, this one only raises it)
        }
        waves<0>(fn, fr, xm, a, piv);
    }
    // The Newton solve is ~1 000 dependent fp64 instructions (3 us at the single-wave issue rate): in front of every tile's columns it made the compute
    // waves the bottleneck (6.5 us per tile against 4.3 us of store issue: 33.7 us per launch, no faster than three launches).  One of the four
    // copy waves takes it over, one tile AHEAD of the compute waves, and leaves x_next in an LDS double buffer; three waves issue the stores
    // just as fast (the store stream is bound by the memory system, not by the waves that feed it).
    static constexpr size_t HELPER_BYTES = ((size_t)TILE * NX * sizeof(T) + 16 + 15) / 16 * 16;
    static constexpr bool HELPER = SplitLds<Fn, T, Plan, 4>::BYTES + 2 * HELPER_BYTES <= 160 * 1024;
    static constexpr bool HELPER_STAGES = true;            // (kernels.hpp HelperStagesOf: zrow / tv / hv below)
    // zrow, tv, hv (HELPER_STAGES): this lane's knot point as it stands in memory, for the loop to put into the LDS input tile
    template <class TR> static RD_DEV void helper_tile(const Fn& fn, const TR& r, unsigned char* dst, int = 0, T* zrow = nullptr, double* tv = nullptr, double* hv = nullptr) {
        constexpr int NZ = NX + NU;
        const int lane = threadIdx.x & 63;
        const int nk = (int)((r.N - r.k0) < TILE ? (r.N - r.k0) : TILE);
        const long long k = r.k0 + (lane < nk ? lane : nk - 1);
        if (fn.solve != 2 || Tune::IFT_ABL == 2) {
            if (zrow) {
#pragma unroll
                for (int i = 0; i < NZ; ++i) zrow[i] = r.Z[k * NZ + i];
                *tv = r.t[k]; *hv = r.dt[k];
            }
            return;
        }
        const double hd = r.dt[k];
        const T h = (T)hd, tj = (T)r.t[k];
        const bool term = is_terminal_dt(hd);
        T x1[NX], u[NU], xn[NX], xl[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x1[i] = r.Z[k * NZ + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) { const T uv = r.Z[k * NZ + NX + i]; if (zrow) zrow[NX + i] = uv; u[i] = term ? T(0) : uv; }
        if (zrow) {
#pragma unroll
            for (int i = 0; i < NX; ++i) zrow[i] = x1[i];
            *tv = r.t[k]; *hv = hd;
        }
        bool unconv = false;
        bool bad = R::newton(fn.m, x1, u, tj, h, fn.iters, fn.tol, xn, xl, &unconv);
        bad = bad || unconv;
        T* hx = reinterpret_cast<T*>(dst);
#pragma unroll
        for (int i = 0; i < NX; ++i) hx[lane * NX + i] = xn[i];
        const bool any = __any(bad);
        if (lane == 0) *reinterpret_cast<int*>(dst + (size_t)TILE * NX * sizeof(T)) = any ? 1 : 0;
    }
    // behind the tile loop: the tiles of THIS workgroup that asked for the exact solve (real row exchanges; Jacobian at the last tested iterate) are
    // redone in place by its first wavefront, with the tile's LDS as the lanes' matrices -- no second launch, and nothing to do in the usual case
    static constexpr bool EPILOGUE = true;
    template <class BT> static RD_DEV void epilogue(const Fn& fn, const BT& bt, long long tile, long long tend, long long tstep, unsigned char* smem, size_t smem_bytes) {
        if (!fn.flag || !fn.solve || threadIdx.x >= TILE) return;
        using L = ImplicitLds<Mdl, T>;
        int kpl = (int)(smem_bytes / L::LANE_BYTES);
        if (kpl > TILE) kpl = TILE;
        for (; tile < tend; tile += tstep) {
            if (__builtin_amdgcn_readfirstlane(fn.flag[1 + tile]) == 0) continue;
            const TileRef<T> r = at_uniform<T>(bt, tile);
            const int nk = (int)((r.N - r.k0) < TILE ? (r.N - r.k0) : TILE);
            for (int c0 = 0; c0 < nk; c0 += kpl)
                implicit_exact_knots<Mdl, T, true>(fn.m, r.N, r.Z, r.t, r.dt, r.Y, r.J, fn.iters, fn.tol, r.k0 + c0, (nk - c0) < kpl ? (nk - c0) : kpl, smem, kpl);
        }
    }
};
template <class Fn, class T, class Plan, int NC>
__global__ void __launch_bounds__((Plan::NG + NC) * TILE) implicit_ift_split_kernel(Fn fn, long long N, const T* __restrict__ Z, const double* __restrict__ tt,
                                                                                     const double* __restrict__ dtt, T* __restrict__ J, T* __restrict__ Xn, unsigned* sched) {
    (void)sched;                                           // (helper wave + epilogue: static schedule)
    jac_split_loop<Fn, T, Plan, NC, IftBody<Fn, T, Plan>>(fn, SingleBatch<T>{N, Z, tt, dtt, J, Xn});
}

template <class Mdl, class T>
__global__ void __launch_bounds__(TILE) implicit_rollout_reg_kernel(Mdl m, long long K, long long M, T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                                    const T* __restrict__ X0, int newton_iters, double tol, int* __restrict__ flag) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NZ = NX + NU;
    using R = ImplicitRegOps<Mdl, T>;
    const long long j = (long long)blockIdx.x * TILE + threadIdx.x;
    const bool active = j < M;
    const long long jc = active ? j : M - 1;
    T* Zj = Z + jc * K * NZ;
    T x1[NX], u[NU], xn[NX], xl[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x1[i] = X0 ? X0[jc * NX + i] : Zj[i];
    if (X0 && active) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Zj[i] = x1[i];
    }
    bool need = false;
    for (long long k = 1; k < K; ++k) {
        const double hd = dtt[jc * K + k - 1];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(hd) ? T(0) : Zj[(k - 1) * NZ + NX + i];
        need = R::newton(m, x1, u, (T)tt[jc * K + k - 1], (T)hd, newton_iters, tol, xn, xl) || need;
#pragma unroll
        for (int i = 0; i < NX; ++i) { x1[i] = xn[i]; if (active) Zj[k * NZ + i] = xn[i]; }
    }
    if (need && flag) *flag = 1;
}

template <class Mdl, class T, int C>
RD_DEV void implicit_error_cols(const Mdl& m, const T (&xm)[Mdl::NX], const T (&u)[Mdl::NU], const T (&x1)[Mdl::NX], const T (&x2)[Mdl::NX], T tj, T h, T* e, T* j2, T* j1) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    if constexpr (C < NX + NU) {
        T col[NX];
        ImplicitOps<Mdl, T>::template jcol<C>(m, xm, u, tj, col);
        if (C == 0 && e) {
            T f[NX]; model_f(m, xm, u, tj + h * T(0.5), f);
#pragma unroll
            for (int i = 0; i < NX; ++i) e[i] = x1[i] + h * f[i] - x2[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const T a = col[i] * h;
            if (C < NX) { if (j1) j1[C * NX + i] = a * T(0.5) + ((i == C) ? T(1) : T(0)); if (j2) j2[C * NX + i] = a * T(0.5) - ((i == C) ? T(1) : T(0)); }
            else { if (j1) j1[C * NX + i] = a; if (j2) j2[C * NX + i] = T(0); }
        }
        implicit_error_cols<Mdl, T, C + 1>(m, xm, u, x1, x2, tj, h, e, j2, j1);
    }
}

// rollout!(sig, DiscretizedDynamics{ImplicitMidpoint}(model), Z, x0): one lane per trajectory, a Newton solve per step
// (src/trajectories.jl:436-441 over src/integration.jl:422-463).  Z: (n+m) x K x M, states of knots 2..K overwritten in place.
template <class Mdl, class T>
__global__ void __launch_bounds__(TILE) implicit_rollout_kernel(Mdl m, long long K, long long M, T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                                const T* __restrict__ X0, int newton_iters, double tol, const int* __restrict__ gate) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NZ = NX + NU;
    using L = ImplicitLds<Mdl, T>;
    using Ops = ImplicitOps<Mdl, T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate && *gate == 0) return;
    if (threadIdx.x >= L::KP) return;
    const int lane = threadIdx.x;
    T* A = reinterpret_cast<T*>(smem) + lane * L::RSA;
    int* piv = reinterpret_cast<int*>(smem + (size_t)L::KP * L::RSA * sizeof(T)) + lane * 16;
    const long long j = (long long)blockIdx.x * L::KP + lane;
    const bool active = j < M;
    const long long jc = active ? j : M - 1;
    T* Zj = Z + jc * K * NZ;
    T x1[NX], u[NU], xn[NX], xl[NX], xm[NX], f[NX], r[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x1[i] = X0 ? X0[jc * NX + i] : Zj[i];
    if (X0 && active) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Zj[i] = x1[i];
    }
    for (long long k = 1; k < K; ++k) {
        const double hd = dtt[jc * K + k - 1];
        const T h = (T)hd, tj = (T)tt[jc * K + k - 1], tm = tj + h * T(0.5);
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = is_terminal_dt(hd) ? T(0) : Zj[(k - 1) * NZ + NX + i];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xn[i] = x1[i]; xl[i] = x1[i]; }
        bool done = false;
        for (int it = 0; it < newton_iters; ++it) {
            if (!done) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xl[i] = xn[i];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) xm[i] = (x1[i] + xl[i]) * T(0.5);
            Ops::template fill_cols<0>(m, xm, u, tj, tm, h, A, f);
            T nr = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) { r[i] = x1[i] + h * f[i] - xl[i]; nr += r[i] * r[i]; }
            if ((double)sqrt_(nr) < tol) done = true;
            if (__all(done)) break;
            Ops::lu(A, piv);
            Ops::solve(A, piv, r);
            if (!done) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xn[i] = xl[i] - r[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) { x1[i] = xn[i]; if (active) Zj[k * NZ + i] = xn[i]; }
    }
}

// dynamics_error / dynamics_error_jacobian! of ImplicitMidpoint for the pairs (Z[k], Z[k+1]) (src/integration.jl:631-690)
template <class Mdl, class T>
__global__ void __launch_bounds__(TILE) implicit_midpoint_error_kernel(Mdl m, long long P, const T* __restrict__ Z, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                                       T* __restrict__ E, T* __restrict__ J2, T* __restrict__ J1) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NZ = NX + NU;
    const long long k = (long long)blockIdx.x * TILE + threadIdx.x;
    if (k >= P) return;
    const double hd = dtt[k];
    const T h = (T)hd, tj = (T)tt[k];
    const bool term = is_terminal_dt(hd);
    T x1[NX], x2[NX], u[NU], xm[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { x1[i] = Z[k * NZ + i]; x2[i] = Z[(k + 1) * NZ + i]; xm[i] = (x1[i] + x2[i]) * T(0.5); }
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = term ? T(0) : Z[k * NZ + NX + i];
    if (E && !J1 && !J2) {
        T f[NX]; model_f(m, xm, u, tj + h * T(0.5), f);
#pragma unroll
        for (int i = 0; i < NX; ++i) E[k * NX + i] = x1[i] + h * f[i] - x2[i];
        return;
    }
    implicit_error_cols<Mdl, T, 0>(m, xm, u, x1, x2, tj, h, E ? E + k * NX : nullptr, J2 ? J2 + k * NX * NZ : nullptr, J1 ? J1 + k * NX * NZ : nullptr);
}
}  // namespace rd
