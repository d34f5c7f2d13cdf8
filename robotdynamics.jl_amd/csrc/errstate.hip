// errstate.hip -- attitude error-state kernels for LieState{QuatRotation,P} models and the Euclidean fall-backs.
//   errstate_jacobian!   src/statevectortype.jl:119-122,149-155 ; src/liestate.jl:262-298
//   state_diff           src/statevectortype.jl:102-104,145-146 ; src/liestate.jl:210-260 (CayleyMap default)
//   nabla-errstate_jac.  src/statevectortype.jl:137-147         ; src/liestate.jl:300-320
// Rotations.jl facts used: QuatRotation(w,x,y,z) normalises by default;
// grad-differential(q) = L(q) H ; grad^2-differential(q,b) = -(q.b) I3 ; rotation_error(q,q0,Cayley) = vec(q0^-1 q)/scalar.
// These kernels are pure bandwidth (12 non-trivial values per 13x12 G): one thread per OUTPUT element, so every
// store instruction writes 64 consecutive elements; the quaternion is re-read through L1/L2 (104 B per knot).
#include "common.hpp"
#include <cstdint>
#include <cstdlib>

namespace rd {

// L(q)H: entry (i, j) = SGN[i][j] * q[IDX[i][j]]
__device__ const signed char LH_IDX[4][3] = {{1, 2, 3}, {0, 3, 2}, {3, 0, 1}, {2, 1, 0}};
__device__ const signed char LH_SGN[4][3] = {{-1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {-1, 1, 1}};

template <class T> __device__ __forceinline__ T rsqrt_(T x);
template <> __device__ __forceinline__ double rsqrt_<double>(double x) { return 1.0 / sqrt(x); }
template <> __device__ __forceinline__ float rsqrt_<float>(float x) { return 1.0f / sqrtf(x); }
template <class T> __device__ __forceinline__ T sqrtT(T x);
template <> __device__ __forceinline__ double sqrtT<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float sqrtT<float>(float x) { return sqrtf(x); }
template <class T> __device__ __forceinline__ T atan2T(T y, T x);
template <> __device__ __forceinline__ double atan2T<double>(double y, double x) { return atan2(y, x); }
template <> __device__ __forceinline__ float atan2T<float>(float y, float x) { return atan2f(y, x); }

template <class T>
__global__ void __launch_bounds__(256) errstate_clear_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, T* __restrict__ G) {
    const long long per = (long long)d.n * d.e;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * per) return;
    const long long k = idx / per;
    const int r_c = (int)(idx - k * per);
    const int c = r_c / d.n, r = r_c - c * d.n;
    T val = T(0);
    if (d.row_kind[r] == 0) {
        val = (c == d.row_col[r]) ? T(1) : T(0);
    } else {
        const int j = c - d.row_col[r];
        if (j >= 0 && j < 3) {
            const T* q = X + k * ldx + d.row_q0[r];
            const T q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const T inv = rsqrt_<T>(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            const int i = d.row_sub[r];
            const int s = LH_IDX[i][j];
            const T qs = s == 0 ? q0 : s == 1 ? q1 : s == 2 ? q2 : q3;
            val = T(LH_SGN[i][j]) * (qs * inv);
        }
    }
    G[idx] = val;
}

// Tile version of the two "clear" kernels (dense G or dense nablaG, everything overwritten): pure streaming stores, so the
// per-element index arithmetic of the kernel above (an integer division by a run-time n*e for every 8 bytes) is the cost,
// not the memory system (measured 1.8 TB/s).  Here a workgroup owns TK knot points: the (TK x per) output tile -- one
// contiguous slab of G -- is zero-filled in LDS with 16-B writes, ONE lane per knot point scatters its few non-zeros
// (identity ones + 12 quaternion entries per rotation, or 3 diagonal entries per rotation for nablaG), and the tile leaves
// as a linear sweep of 16-B-per-lane stores.  One-shot grid (hardware workgroup dispatch balances the CUs).
// MODE 0: errstate Jacobian (n x e per knot)   MODE 1: nabla errstate Jacobian (e x e per knot)
template <class T, int MODE, int TK>
__global__ void __launch_bounds__(256) errstate_tile_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, const T* __restrict__ Xb, int ldxb, T* __restrict__ G) {
    typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr int VN = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    T* tile = reinterpret_cast<T*>(smem_);
    const int rows = MODE == 0 ? d.n : d.e;
    const int per = rows * d.e;
    const long long k0 = (long long)blockIdx.x * TK;
    const int nk = (int)((N - k0) < TK ? (N - k0) : TK);
    const int total = nk * per;                              // elements of this tile (TK*per is a multiple of VN: TK = 32 or 64)
    // the lane's quaternions (and nablaG's xbar parts) are requested first: their latency hides under the zero fill
    T qv[4][4], bv[4][4];
    const bool owner = threadIdx.x < nk;
    {
        const T* x = X + (k0 + (owner ? threadIdx.x : 0)) * ldx;
        const T* xb = MODE == 1 ? Xb + (k0 + (owner ? threadIdx.x : 0)) * ldxb : nullptr;
#pragma unroll
        for (int rot = 0; rot < 4; ++rot) {
            const int r0 = rot < d.nrot ? d.rot_row0[rot] : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { qv[rot][i] = (owner && rot < d.nrot) ? x[r0 + i] : T(1); bv[rot][i] = (MODE == 1 && owner && rot < d.nrot) ? xb[r0 + i] : T(0); }
        }
    }
    V zero;
#pragma unroll
    for (int q = 0; q < VN; ++q) zero[q] = T(0);
    for (int i = threadIdx.x * VN; i < TK * per; i += 256 * VN) *reinterpret_cast<V*>(tile + i) = zero;
    __syncthreads();
    if (owner) {
        T* g = tile + threadIdx.x * per;
        if (MODE == 0) {
            // constant trip count: the descriptor bytes are then read with (batched) scalar loads from the kernel arguments; a loop with a
            // run-time bound indexes the descriptor dynamically, which turns into one dependent global load per row (13 serial round trips)
#pragma unroll
            for (int r = 0; r < 16; ++r) if (r < d.n && d.row_kind[r] == 0) g[r + d.row_col[r] * d.n] = T(1);
        }
#pragma unroll
        for (int rot = 0; rot < 4; ++rot) {
            if (rot >= d.nrot) break;
            const int r0 = d.rot_row0[rot], c0 = d.rot_col0[rot];
            const T q0 = qv[rot][0], q1 = qv[rot][1], q2 = qv[rot][2], q3 = qv[rot][3];
            const T inv = rsqrt_<T>(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            if (MODE == 0) {
                const T a = q0 * inv, b = q1 * inv, c = q2 * inv, e = q3 * inv;      // L(q) H, q normalised (QuatRotation constructor)
                T* g0 = g + r0 + c0 * d.n;
                g0[0] = -b; g0[1] = a;  g0[2] = e;  g0[3] = -c;  g0 += d.n;
                g0[0] = -c; g0[1] = -e; g0[2] = a;  g0[3] = b;   g0 += d.n;
                g0[0] = -e; g0[1] = c;  g0[2] = -b; g0[3] = a;
            } else {
                const T val = -((q0 * bv[rot][0] + q1 * bv[rot][1] + q2 * bv[rot][2] + q3 * bv[rot][3]) * inv);
                for (int j = 0; j < 3; ++j) g[(c0 + j) * (d.e + 1)] = val;
            }
        }
    }
    __syncthreads();
    T* out = G + k0 * per;
    const int nv = total / VN;
#pragma unroll 4
    for (int i = threadIdx.x; i < nv; i += 256) reinterpret_cast<V*>(out)[i] = reinterpret_cast<const V*>(tile)[i];
    for (int i = nv * VN + threadIdx.x; i < total; i += 256) out[i] = tile[i];
}

// reference-exact variant: touch only the pattern entries (ones + rotation blocks), leave the rest of G alone
template <class T>
__global__ void __launch_bounds__(256) errstate_pattern_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, T* __restrict__ G) {
    const int nvec = d.n - 4 * d.nrot;
    const int per = nvec + 12 * d.nrot;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * per) return;
    const long long k = idx / per;
    int p = (int)(idx - k * per);
    T* Gk = G + k * d.n * d.e;
    if (p < nvec) {   // p-th vector row
        int r = 0;
        for (int cnt = -1; r < d.n; ++r) { if (d.row_kind[r] == 0) ++cnt; if (cnt == p) break; }
        Gk[r + d.row_col[r] * d.n] = T(1);
    } else {
        p -= nvec;
        const int rot = p / 12, ij = p - rot * 12, j = ij / 4, i = ij - j * 4;
        const int r0 = d.rot_row0[rot], c0 = d.rot_col0[rot];
        const T* q = X + k * ldx + r0;
        const T q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        const T inv = rsqrt_<T>(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        const int s = LH_IDX[i][j];
        const T qs = s == 0 ? q0 : s == 1 ? q1 : s == 2 ? q2 : q3;
        Gk[(r0 + i) + (c0 + j) * d.n] = T(LH_SGN[i][j]) * (qs * inv);
    }
}

// LieState{MRP,P} / LieState{RodriguesParam,P}: the 3 x 3 rotation blocks grad-differential(p) of Rotations.jl (the Jacobian of params(p * R(delta)) at
// delta = 0, delta in the rotation's own parameters):   MRP  (1 - p.p) I + 2 (skew(p) + p p')      RodriguesParam  I + skew(g) + g g'
// One thread per element of the n x e slab; clear = 0 writes only the pattern (vector-part ones and the blocks), like the reference.
template <class T> __device__ __forceinline__ T rot3_block(int rot, const T* p, int i, int j) {
    const T p0 = p[0], p1 = p[1], p2 = p[2];
    const T pi = i == 0 ? p0 : i == 1 ? p1 : p2, pj = j == 0 ? p0 : j == 1 ? p1 : p2;
    T sk = T(0);
    if (i != j) { const int k = 3 - i - j; const T pk = k == 0 ? p0 : k == 1 ? p1 : p2; sk = ((j - i + 3) % 3 == 2) ? pk : -pk; }
    if (rot == RDHIP_ROT_MRP) return ((i == j) ? T(1) - (p0 * p0 + p1 * p1 + p2 * p2) : T(0)) + T(2) * (sk + pi * pj);
    return ((i == j) ? T(1) : T(0)) + sk + pi * pj;
}
// grad^2-differential(p, b) of the three-parameter rotations, entry (i, j): the contraction with b of the second derivative of delta -> params(p * R(delta))
// at delta = 0 (src/liestate.jl:314 calls Rotations.jl's grad^2-differential; restated from that definition, derivation in DESIGN.md section 4.6):
//   MRP  (b.p) (8 p p' - 2 (1 + p.p) I) + 2 (p c' + c p'),  c = (1 - p.p) b - 2 p x b        RodriguesParam  2 (b.g) g g' + g c' + c g',  c = b - g x b
template <class T> __device__ __forceinline__ T rot3_hessian_entry(int rot, const T* p, const T* b, int i, int j) {
    const T a = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], bp = b[0] * p[0] + b[1] * p[1] + b[2] * p[2];
    const T pxb[3] = {p[1] * b[2] - p[2] * b[1], p[2] * b[0] - p[0] * b[2], p[0] * b[1] - p[1] * b[0]};
    const T pi = p[i], pj = p[j];
    if (rot == RDHIP_ROT_MRP) {
        const T ci = (T(1) - a) * b[i] - T(2) * pxb[i], cj = (T(1) - a) * b[j] - T(2) * pxb[j];
        return bp * (T(8) * pi * pj - ((i == j) ? T(2) * (T(1) + a) : T(0))) + T(2) * (pi * cj + ci * pj);
    }
    const T ci = b[i] - pxb[i], cj = b[j] - pxb[j];
    return T(2) * bp * pi * pj + pi * cj + ci * pj;
}
template <class T>
__global__ void __launch_bounds__(256) errstate3_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, int clear, T* __restrict__ G) {
    const long long per = (long long)d.n * d.e;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * per) return;
    const long long k = idx / per;
    const int r_c = (int)(idx - k * per);
    const int c = r_c / d.n, r = r_c - c * d.n;
    if (d.row_kind[r] == 0) {
        if (c == d.row_col[r]) G[idx] = T(1); else if (clear) G[idx] = T(0);
    } else {
        const int j = c - d.row_col[r];
        if (j >= 0 && j < 3) G[idx] = rot3_block<T>(d.rot, X + k * ldx + d.row_q0[r], d.row_sub[r], j);
        else if (clear) G[idx] = T(0);
    }
}


// inv(errmap)(dq) for dq = [w; v], the quaternion of R0^-1 R (src/liestate.jl:216-217: Rotations.rotation_error(R, R0, errmap); rdhip.h lists the maps).
// `rot`: the state's parameterisation -- IdentityMap returns the relative rotation's own parameters (MRP v / (1 + w), RodriguesParam v / w).
template <class T> __device__ __forceinline__ void errmap_apply(int errmap, int rot, T w, const T (&v)[3], T (&o)[3]) {
    T M;
    if (errmap == RDHIP_ERRMAP_EXPONENTIAL) {
        const T th = sqrtT<T>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        M = th > T(1e-6) ? atan2T<T>(th, w) / th : (T(1) - th * th / (T(3) * w * w)) / w;
    } else if (errmap == RDHIP_ERRMAP_MRP) M = T(2) / (T(1) + w);
    else if (errmap == RDHIP_ERRMAP_QUATVEC) M = w > T(0) ? T(1) : (w < T(0) ? T(-1) : T(0));
    else if (errmap == RDHIP_ERRMAP_IDENTITY && rot == RDHIP_ROT_MRP) M = T(1) / (T(1) + w);
    else M = T(1) / w;                                   // CayleyMap; IdentityMap of a RodriguesParam state
    o[0] = errmap == RDHIP_ERRMAP_CAYLEY ? v[0] / w : v[0] * M;      // (the Cayley error keeps the division it has always had: bit-identical to rounds 1-5)
    o[1] = errmap == RDHIP_ERRMAP_CAYLEY ? v[1] / w : v[1] * M;
    o[2] = errmap == RDHIP_ERRMAP_CAYLEY ? v[2] / w : v[2] * M;
}
template <class T>
__global__ void __launch_bounds__(256) state_diff_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, const T* __restrict__ X0, int ldx0, int errmap, T* __restrict__ dX) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * d.e) return;
    const long long k = idx / d.e;
    const int c = (int)(idx - k * d.e);
    const int src = d.err_src[c];
    const T* x = X + k * ldx; const T* x0 = X0 + k * ldx0;
    if (src >= 0) { dX[idx] = x[src] - x0[src]; return; }
    const int rot = -1 - src;
    const int j = c - d.rot_col0[rot];   // 0..2
    // both rotations as unit quaternions [w x y z]: QuatRotation(w,x,y,z) normalises; RodriguesParam g = v/w -> [1; g]/|.|;
    // MRP p = v/(1+w) -> [1 - p.p; 2p]/(1 + p.p)   (Rotations.jl conversions)
    T qa[4], qb[4];
    const T* ra = x + d.rot_row0[rot]; const T* rb = x0 + d.rot_row0[rot];
    if (d.rot == RDHIP_ROT_QUAT) {
        const T iq = rsqrt_<T>(ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2] + ra[3] * ra[3]);
        const T ip = rsqrt_<T>(rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2] + rb[3] * rb[3]);
        for (int i = 0; i < 4; ++i) { qa[i] = ra[i] * iq; qb[i] = rb[i] * ip; }
    } else if (d.rot == RDHIP_ROT_RP) {
        const T ia = rsqrt_<T>(T(1) + ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2]), ib = rsqrt_<T>(T(1) + rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2]);
        qa[0] = ia; qb[0] = ib;
        for (int i = 0; i < 3; ++i) { qa[1 + i] = ra[i] * ia; qb[1 + i] = rb[i] * ib; }
    } else {
        const T na = ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2], nb = rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2];
        const T ia = T(1) / (T(1) + na), ib = T(1) / (T(1) + nb);
        qa[0] = (T(1) - na) * ia; qb[0] = (T(1) - nb) * ib;
        for (int i = 0; i < 3; ++i) { qa[1 + i] = T(2) * ra[i] * ia; qb[1 + i] = T(2) * rb[i] * ib; }
    }
    const T a0 = qa[0], a1 = qa[1], a2 = qa[2], a3 = qa[3];                      // q
    const T b0 = qb[0], b1 = -qb[1], b2 = -qb[2], b3 = -qb[3];                   // conj(q0)
    // dq = conj(q0) (x) q ; Cayley error = vec(dq) / scalar(dq)  (rotation_error(q, q0, CayleyMap), test/rbstate.jl:157)
    const T w = b0 * a0 - b1 * a1 - b2 * a2 - b3 * a3;
    const T v[3] = {b0 * a1 + b1 * a0 + b2 * a3 - b3 * a2, b0 * a2 - b1 * a3 + b2 * a0 + b3 * a1, b0 * a3 + b1 * a2 - b2 * a1 + b3 * a0};
    T o[3];
    errmap_apply<T>(errmap, d.rot, w, v, o);
    dX[idx] = o[j];
}

template <class T>
__global__ void __launch_bounds__(256) nabla_errstate_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, const T* __restrict__ Xb, int ldxb,
                                                             int clear, T* __restrict__ H) {
    const long long per = clear ? (long long)d.e * d.e : 9LL * d.nrot;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * per) return;
    const long long k = idx / per;
    const int p = (int)(idx - k * per);
    int r, c;
    if (clear) { c = p / d.e; r = p - c * d.e; }
    else { const int rot = p / 9, ij = p - rot * 9; c = d.rot_col0[rot] + ij / 3; r = d.rot_col0[rot] + ij % 3; }
    T val = T(0);
    if (d.rot != RDHIP_ROT_QUAT) {
        for (int rot = 0; rot < d.nrot; ++rot) {
            const int c0 = d.rot_col0[rot];
            if (r >= c0 && r < c0 + 3 && c >= c0 && c < c0 + 3)
                val = rot3_hessian_entry<T>(d.rot, X + k * ldx + d.rot_row0[rot], Xb + k * ldxb + d.rot_row0[rot], r - c0, c - c0);
        }
    } else if (r == c) {
        for (int rot = 0; rot < d.nrot; ++rot) {
            if (r >= d.rot_col0[rot] && r < d.rot_col0[rot] + 3) {
                const T* q = X + k * ldx + d.rot_row0[rot]; const T* b = Xb + k * ldxb + d.rot_row0[rot];
                const T iq = rsqrt_<T>(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                val = -((q[0] * b[0] + q[1] * b[1] + q[2] * b[2] + q[3] * b[3]) * iq);
            }
        }
    }
    H[k * d.e * d.e + r + (long long)c * d.e] = val;
}

template <class T>
__global__ void __launch_bounds__(256) fill_identity_kernel(int n, int e, long long N, T* __restrict__ G, int identity) {
    const long long per = (long long)n * e;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * per) return;
    const int rc = (int)(idx % per);
    G[idx] = (identity && (rc / n) == (rc % n)) ? T(1) : T(0);
}

// E[i, k] -= Z[i, k+1]  (i < n): one thread per element, consecutive threads -> consecutive addresses of E
template <class T>
__global__ void __launch_bounds__(256) sub_next_state_kernel(long long pairs, int n, int nz, const T* __restrict__ Z, T* __restrict__ E) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pairs * n) return;
    const long long k = idx / n;
    const int i = (int)(idx - k * n);
    E[idx] -= Z[(k + 1) * nz + i];
}
// J2[:, :, k] = [-I 0], n x nz column-major
template <class T>
__global__ void __launch_bounds__(256) fill_neg_identity_kernel(long long pairs, int n, int nz, T* __restrict__ J2) {
    const long long per = (long long)n * nz;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pairs * per) return;
    const int rc = (int)(idx % per);
    J2[idx] = ((rc / n) == (rc % n)) ? T(-1) : T(0);
}

// E[k] = G(x_{k+1})' [A_k G(x_k) | B_k] from a stack of n x (n+m) Jacobians, any LieState{R,P} (quaternion 4 x 3 or MRP / RodriguesParam 3 x 3 blocks): one thread per output element.
// (RigidBody models use the fused kernel of kernels.hpp instead; this is the general composite for the other LieGroupModels.)
template <class T> __device__ __forceinline__ T lh_entry(const T* q, T inv, int i, int j) {
    const int s = LH_IDX[i][j];
    return T(LH_SGN[i][j]) * (q[s] * inv);
}
template <class T>
__global__ void __launch_bounds__(256) project_errexp_kernel(LieDesc d, long long pairs, int m, const T* __restrict__ Z, const T* __restrict__ J, T* __restrict__ E) {
    const int nz = d.n + m, ec = d.e + m;
    const long long per = (long long)d.e * ec;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pairs * per) return;
    const long long k = idx / per;
    const int rc = (int)(idx - k * per);
    const int c = rc / d.e, r = rc - c * d.e;             // output (row r, column c), column-major e x (e+m)
    const T* Jk = J + k * d.n * nz;
    const T* x1 = Z + k * nz; const T* x2 = Z + (k + 1) * nz;
    // row side: error row r <- state rows
    int rs[4]; T rw[4]; int nr = 1;
    if (d.err_src[r] >= 0) { rs[0] = d.err_src[r]; rw[0] = T(1); }
    else if (d.rot != RDHIP_ROT_QUAT) {                   // LieState{MRP | RodriguesParam, P}: the 3 x 3 blocks of errstate3_kernel
        const int rot = -1 - d.err_src[r], s0 = d.rot_row0[rot], j = r - d.rot_col0[rot];
        nr = 3; for (int i = 0; i < 3; ++i) { rs[i] = s0 + i; rw[i] = rot3_block<T>(d.rot, x2 + s0, i, j); }
    } else {
        const int rot = -1 - d.err_src[r], s0 = d.rot_row0[rot], j = r - d.rot_col0[rot];
        const T* q = x2 + s0; const T inv = rsqrt_<T>(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        nr = 4; for (int i = 0; i < 4; ++i) { rs[i] = s0 + i; rw[i] = lh_entry<T>(q, inv, i, j); }
    }
    // column side: error column c (< e) <- state columns ; control columns map one to one
    int cs[4]; T cw[4]; int nc = 1;
    if (c >= d.e) { cs[0] = d.n + (c - d.e); cw[0] = T(1); }
    else if (d.err_src[c] >= 0) { cs[0] = d.err_src[c]; cw[0] = T(1); }
    else if (d.rot != RDHIP_ROT_QUAT) {
        const int rot = -1 - d.err_src[c], s0 = d.rot_row0[rot], j = c - d.rot_col0[rot];
        nc = 3; for (int i = 0; i < 3; ++i) { cs[i] = s0 + i; cw[i] = rot3_block<T>(d.rot, x1 + s0, i, j); }
    } else {
        const int rot = -1 - d.err_src[c], s0 = d.rot_row0[rot], j = c - d.rot_col0[rot];
        const T* q = x1 + s0; const T inv = rsqrt_<T>(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        nc = 4; for (int i = 0; i < 4; ++i) { cs[i] = s0 + i; cw[i] = lh_entry<T>(q, inv, i, j); }
    }
    T acc = T(0);
    for (int a = 0; a < nr; ++a) for (int b = 0; b < nc; ++b) acc += rw[a] * cw[b] * Jk[rs[a] + cs[b] * d.n];
    E[idx] = acc;
}

static inline int st(hipError_t e) { return e == hipSuccess ? RDHIP_OK : RDHIP_ERR_HIP; }
static inline unsigned nblk(long long work) { return (unsigned)((work + 255) / 256); }

template <class T> static int errstate_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, int clear, void* G) {
    if (N == 0) return RDHIP_OK;
    if (d.nrot == 0) {   // EuclideanState: G = I (always cleared, src/statevectortype.jl:149-155)
        RD_LAUNCH((fill_identity_kernel<T>), dim3(nblk(N * d.n * d.e)), dim3(256), 0, s, d.n, d.e, N, (T*)G, 1);
    } else if (d.rot != RDHIP_ROT_QUAT) {
        RD_LAUNCH((errstate3_kernel<T>), dim3(nblk(N * d.n * d.e)), dim3(256), 0, s, d, N, (const T*)X, ldx, clear, (T*)G);
    } else if (clear) {
        // 32 knot points per tile (40 KB for a rigid body: four workgroups per CU; 64-knot tiles, two per CU, ran 3 % slower; 16-knot tiles 1.5 %)
        constexpr int TK = 32;
        const size_t lds = (size_t)TK * d.n * d.e * sizeof(T);
        if (lds <= 80 * 1024 && (reinterpret_cast<uintptr_t>(G) & 15) == 0) {
            auto kern = errstate_tile_kernel<T, 0, TK>;
            if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 80 * 1024) != hipSuccess) return RDHIP_ERR_HIP;
            RD_LAUNCH(kern, dim3((unsigned)((N + TK - 1) / TK)), dim3(256), lds, s, d, N, (const T*)X, ldx, (const T*)nullptr, 0, (T*)G);
        } else {
            RD_LAUNCH((errstate_clear_kernel<T>), dim3(nblk(N * d.n * d.e)), dim3(256), 0, s, d, N, (const T*)X, ldx, (T*)G);
        }
    } else {
        const long long per = (d.n - 4 * d.nrot) + 12 * d.nrot;
        RD_LAUNCH((errstate_pattern_kernel<T>), dim3(nblk(N * per)), dim3(256), 0, s, d, N, (const T*)X, ldx, (T*)G);
    }
    return st(hipGetLastError());
}
int launch_errstate_jacobian(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, int clear, void* G) {
    return dtype == RDHIP_F64 ? errstate_T<double>(d, s, N, X, ldx, clear, G) : errstate_T<float>(d, s, N, X, ldx, clear, G);
}
// Tiled state_diff: 128 knot points per workgroup; the x and x0 tiles (contiguous spans of the inputs, any leading dimension <= MAXLD) are
// staged into LDS with lane-contiguous loads, ALL of a thread's loads of both tiles requested before the first is waited for (the loop used to
// run in batches of four: eight dependent memory round trips per workgroup, 20 us for 262 144 quadrotor states = 0.49 of the roofline), ONE
// lane then computes all e error coordinates of its knot point (each relative rotation once, not once per component), and the e x 128 output
// tile leaves as a lane-contiguous sweep.
template <class T, int MAXLD>
__global__ void __launch_bounds__(128) state_diff_tile_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, const T* __restrict__ X0, int ldx0, int errmap, T* __restrict__ dX) {
    constexpr int NB = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sd[];
    T* xa = reinterpret_cast<T*>(smem_sd);
    T* xb = xa + NB * ldx;
    const int RS = d.e | 1;
    T* out = xb + NB * ldx0;
    const long long k0 = (long long)blockIdx.x * NB;
    const int nk = (int)((N - k0) < NB ? (N - k0) : NB);
    const int tid = threadIdx.x;
    {
        // the header only promises ld >= n: the last state of the array may be followed by fewer than ld - n valid elements, so the
        // staged span ends with the last knot point's n-th entry, not with its full ld-wide column
        const T* sa = X + k0 * ldx; const T* sb = X0 + k0 * ldx0;
        const int ta = (nk - 1) * ldx + d.n, tb = (nk - 1) * ldx0 + d.n;
        T ra[MAXLD], rb[MAXLD];
#pragma unroll
        for (int q = 0; q < MAXLD; ++q) { const int e = q * NB + tid; ra[q] = sa[e < ta ? e : 0]; }
#pragma unroll
        for (int q = 0; q < MAXLD; ++q) { const int e = q * NB + tid; rb[q] = sb[e < tb ? e : 0]; }
#pragma unroll
        for (int q = 0; q < MAXLD; ++q) { const int e = q * NB + tid; if (e < ta) xa[e] = ra[q]; }
#pragma unroll
        for (int q = 0; q < MAXLD; ++q) { const int e = q * NB + tid; if (e < tb) xb[e] = rb[q]; }
    }
    __syncthreads();
    if (tid < nk) {
        const T* x = xa + tid * ldx; const T* x0 = xb + tid * ldx0;
        T* o = out + tid * RS;
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < d.e && d.err_src[c] >= 0) o[c] = x[d.err_src[c]] - x0[d.err_src[c]];
#pragma unroll
        for (int rot = 0; rot < 4; ++rot) {
            if (rot >= d.nrot) break;
            const T* ra = x + d.rot_row0[rot]; const T* rb = x0 + d.rot_row0[rot];
            T qa[4], qb[4];
            if (d.rot == RDHIP_ROT_QUAT) {
                const T iq = rsqrt_<T>(ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2] + ra[3] * ra[3]);
                const T ip = rsqrt_<T>(rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2] + rb[3] * rb[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { qa[i] = ra[i] * iq; qb[i] = rb[i] * ip; }
            } else if (d.rot == RDHIP_ROT_RP) {
                const T ia = rsqrt_<T>(T(1) + ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2]), ib = rsqrt_<T>(T(1) + rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2]);
                qa[0] = ia; qb[0] = ib;
#pragma unroll
                for (int i = 0; i < 3; ++i) { qa[1 + i] = ra[i] * ia; qb[1 + i] = rb[i] * ib; }
            } else {
                const T na = ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2], nb = rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2];
                const T ia = T(1) / (T(1) + na), ib = T(1) / (T(1) + nb);
                qa[0] = (T(1) - na) * ia; qb[0] = (T(1) - nb) * ib;
#pragma unroll
                for (int i = 0; i < 3; ++i) { qa[1 + i] = T(2) * ra[i] * ia; qb[1 + i] = T(2) * rb[i] * ib; }
            }
            const T a0 = qa[0], a1 = qa[1], a2 = qa[2], a3 = qa[3];
            const T b0 = qb[0], b1 = -qb[1], b2 = -qb[2], b3 = -qb[3];
            const T w = b0 * a0 - b1 * a1 - b2 * a2 - b3 * a3;                   // same expressions as state_diff_kernel below
            T* oc = o + d.rot_col0[rot];
            const T v[3] = {b0 * a1 + b1 * a0 + b2 * a3 - b3 * a2, b0 * a2 - b1 * a3 + b2 * a0 + b3 * a1, b0 * a3 + b1 * a2 - b2 * a1 + b3 * a0};
            T er[3];
            errmap_apply<T>(errmap, d.rot, w, v, er);
            oc[0] = er[0]; oc[1] = er[1]; oc[2] = er[2];
        }
    }
    __syncthreads();
    T* dst = dX + k0 * d.e;
    const int total = nk * d.e;
    for (int i = tid; i < total; i += NB) { const int r = i / d.e; dst[i] = out[r * RS + (i - r * d.e)]; }
}

template <class T> static int sdiff_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, const void* X0, int ldx0, int errmap, void* dX) {
    if (N == 0) return RDHIP_OK;
    const size_t lds = (size_t)128 * (ldx + ldx0 + (d.e | 1)) * sizeof(T);
    if (ldx <= 16 && ldx0 <= 16 && lds <= 64 * 1024) {
        RD_LAUNCH((state_diff_tile_kernel<T, 16>), dim3((unsigned)((N + 127) / 128)), dim3(128), lds, s, d, N, (const T*)X, ldx, (const T*)X0, ldx0, errmap, (T*)dX);
        return st(hipGetLastError());
    }
    if (ldx <= 32 && ldx0 <= 32 && lds <= 64 * 1024) {
        RD_LAUNCH((state_diff_tile_kernel<T, 32>), dim3((unsigned)((N + 127) / 128)), dim3(128), lds, s, d, N, (const T*)X, ldx, (const T*)X0, ldx0, errmap, (T*)dX);
        return st(hipGetLastError());
    }
    RD_LAUNCH((state_diff_kernel<T>), dim3(nblk(N * d.e)), dim3(256), 0, s, d, N, (const T*)X, ldx, (const T*)X0, ldx0, errmap, (T*)dX);
    return st(hipGetLastError());
}
int launch_state_diff(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* X0, int ldx0, int errmap, void* dX) {
    return dtype == RDHIP_F64 ? sdiff_T<double>(d, s, N, X, ldx, X0, ldx0, errmap, dX) : sdiff_T<float>(d, s, N, X, ldx, X0, ldx0, errmap, dX);
}
template <class T> static int nabla_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, const void* Xb, int ldxb, int clear, void* H) {
    if (N == 0) return RDHIP_OK;
    if (d.nrot == 0) {   // EuclideanState: nablaG .= 0 (src/statevectortype.jl:147)
        RD_LAUNCH((fill_identity_kernel<T>), dim3(nblk(N * d.e * d.e)), dim3(256), 0, s, d.e, d.e, N, (T*)H, 0);
        return st(hipGetLastError());
    }
    constexpr int TK = 64;
    const size_t lds = (size_t)TK * d.e * d.e * sizeof(T);
    if (clear && d.rot == RDHIP_ROT_QUAT && lds <= 80 * 1024 && (reinterpret_cast<uintptr_t>(H) & 15) == 0) {
        auto kern = errstate_tile_kernel<T, 1, TK>;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 80 * 1024) != hipSuccess) return RDHIP_ERR_HIP;
        RD_LAUNCH(kern, dim3((unsigned)((N + TK - 1) / TK)), dim3(256), lds, s, d, N, (const T*)X, ldx, (const T*)Xb, ldxb, (T*)H);
        return st(hipGetLastError());
    }
    const long long per = clear ? (long long)d.e * d.e : 9LL * d.nrot;
    RD_LAUNCH((nabla_errstate_kernel<T>), dim3(nblk(N * per)), dim3(256), 0, s, d, N, (const T*)X, ldx, (const T*)Xb, ldxb, clear, (T*)H);
    return st(hipGetLastError());
}
int launch_nabla_errstate(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* Xb, int ldxb, int clear, void* H) {
    return dtype == RDHIP_F64 ? nabla_T<double>(d, s, N, X, ldx, Xb, ldxb, clear, H) : nabla_T<float>(d, s, N, X, ldx, Xb, ldxb, clear, H);
}

int launch_project_errexp(const LieDesc& d, int dtype, hipStream_t s, long long pairs, int m, const void* Z, const void* J, void* E) {
    if (pairs <= 0) return RDHIP_OK;
    const long long work = pairs * d.e * (d.e + m);
    if (dtype == RDHIP_F64) RD_LAUNCH((project_errexp_kernel<double>), dim3(nblk(work)), dim3(256), 0, s, d, pairs, m, (const double*)Z, (const double*)J, (double*)E);
    else RD_LAUNCH((project_errexp_kernel<float>), dim3(nblk(work)), dim3(256), 0, s, d, pairs, m, (const float*)Z, (const float*)J, (float*)E);
    return st(hipGetLastError());
}
int launch_sub_next_state(int dtype, hipStream_t s, long long pairs, int n, int nz, const void* Z, void* E) {
    if (pairs <= 0) return RDHIP_OK;
    if (dtype == RDHIP_F64) RD_LAUNCH((sub_next_state_kernel<double>), dim3(nblk(pairs * n)), dim3(256), 0, s, pairs, n, nz, (const double*)Z, (double*)E);
    else RD_LAUNCH((sub_next_state_kernel<float>), dim3(nblk(pairs * n)), dim3(256), 0, s, pairs, n, nz, (const float*)Z, (float*)E);
    return st(hipGetLastError());
}
int launch_fill_neg_identity(int dtype, hipStream_t s, long long pairs, int n, int nz, void* J2) {
    if (pairs <= 0) return RDHIP_OK;
    if (dtype == RDHIP_F64) RD_LAUNCH((fill_neg_identity_kernel<double>), dim3(nblk(pairs * n * nz)), dim3(256), 0, s, pairs, n, nz, (double*)J2);
    else RD_LAUNCH((fill_neg_identity_kernel<float>), dim3(nblk(pairs * n * nz)), dim3(256), 0, s, pairs, n, nz, (float*)J2);
    return st(hipGetLastError());
}

}  // namespace rd
