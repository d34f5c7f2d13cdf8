// errstate.hip -- attitude error-state kernels for LieState{QuatRotation,P} models and the Euclidean fall-backs.
//   errstate_jacobian!   src/statevectortype.jl:119-122,149-155 ; src/liestate.jl:262-298
//   state_diff           src/statevectortype.jl:102-104,145-146 ; src/liestate.jl:210-260 (CayleyMap default)
//   nabla-errstate_jac.  src/statevectortype.jl:137-147         ; src/liestate.jl:300-320
// Rotations.jl facts used: QuatRotation(w,x,y,z) normalises by default;
// grad-differential(q) = L(q) H ; grad^2-differential(q,b) = -(q.b) I3 ; rotation_error(q,q0,Cayley) = vec(q0^-1 q)/scalar.
// These kernels are pure bandwidth (12 non-trivial values per 13x12 G): one thread per OUTPUT element, so every
// store instruction writes 64 consecutive elements; the quaternion is re-read through L1/L2 (104 B per knot).
#include "common.hpp"

namespace rd {

// L(q)H: entry (i, j) = SGN[i][j] * q[IDX[i][j]]
__device__ const signed char LH_IDX[4][3] = {{1, 2, 3}, {0, 3, 2}, {3, 0, 1}, {2, 1, 0}};
__device__ const signed char LH_SGN[4][3] = {{-1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {-1, 1, 1}};

template <class T> __device__ __forceinline__ T rsqrt_(T x);
template <> __device__ __forceinline__ double rsqrt_<double>(double x) { return 1.0 / sqrt(x); }
template <> __device__ __forceinline__ float rsqrt_<float>(float x) { return 1.0f / sqrtf(x); }

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

template <class T>
__global__ void __launch_bounds__(256) state_diff_kernel(LieDesc d, long long N, const T* __restrict__ X, int ldx, const T* __restrict__ X0, int ldx0, T* __restrict__ dX) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * d.e) return;
    const long long k = idx / d.e;
    const int c = (int)(idx - k * d.e);
    const int src = d.err_src[c];
    const T* x = X + k * ldx; const T* x0 = X0 + k * ldx0;
    if (src >= 0) { dX[idx] = x[src] - x0[src]; return; }
    const int rot = -1 - src;
    const int j = c - d.rot_col0[rot];   // 0..2
    const T* q = x + d.rot_row0[rot]; const T* p = x0 + d.rot_row0[rot];
    const T iq = rsqrt_<T>(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T ip = rsqrt_<T>(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    const T a0 = q[0] * iq, a1 = q[1] * iq, a2 = q[2] * iq, a3 = q[3] * iq;     // q
    const T b0 = p[0] * ip, b1 = -p[1] * ip, b2 = -p[2] * ip, b3 = -p[3] * ip;  // conj(q0)
    // dq = conj(q0) (x) q
    const T w = b0 * a0 - b1 * a1 - b2 * a2 - b3 * a3;
    const T v = (j == 0) ? (b0 * a1 + b1 * a0 + b2 * a3 - b3 * a2)
              : (j == 1) ? (b0 * a2 - b1 * a3 + b2 * a0 + b3 * a1)
                         : (b0 * a3 + b1 * a2 - b2 * a1 + b3 * a0);
    dX[idx] = v / w;
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
    if (r == c) {
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

static inline int st(hipError_t e) { return e == hipSuccess ? RDHIP_OK : RDHIP_ERR_HIP; }
static inline unsigned nblk(long long work) { return (unsigned)((work + 255) / 256); }

template <class T> static int errstate_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, int clear, void* G) {
    if (N == 0) return RDHIP_OK;
    if (d.nrot == 0) {   // EuclideanState: G = I (always cleared, src/statevectortype.jl:149-155)
        hipLaunchKernelGGL((fill_identity_kernel<T>), dim3(nblk(N * d.n * d.e)), dim3(256), 0, s, d.n, d.e, N, (T*)G, 1);
    } else if (clear) {
        hipLaunchKernelGGL((errstate_clear_kernel<T>), dim3(nblk(N * d.n * d.e)), dim3(256), 0, s, d, N, (const T*)X, ldx, (T*)G);
    } else {
        const long long per = (d.n - 4 * d.nrot) + 12 * d.nrot;
        hipLaunchKernelGGL((errstate_pattern_kernel<T>), dim3(nblk(N * per)), dim3(256), 0, s, d, N, (const T*)X, ldx, (T*)G);
    }
    return st(hipGetLastError());
}
int launch_errstate_jacobian(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, int clear, void* G) {
    return dtype == RDHIP_F64 ? errstate_T<double>(d, s, N, X, ldx, clear, G) : errstate_T<float>(d, s, N, X, ldx, clear, G);
}
template <class T> static int sdiff_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, const void* X0, int ldx0, void* dX) {
    if (N == 0) return RDHIP_OK;
    hipLaunchKernelGGL((state_diff_kernel<T>), dim3(nblk(N * d.e)), dim3(256), 0, s, d, N, (const T*)X, ldx, (const T*)X0, ldx0, (T*)dX);
    return st(hipGetLastError());
}
int launch_state_diff(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* X0, int ldx0, void* dX) {
    return dtype == RDHIP_F64 ? sdiff_T<double>(d, s, N, X, ldx, X0, ldx0, dX) : sdiff_T<float>(d, s, N, X, ldx, X0, ldx0, dX);
}
template <class T> static int nabla_T(const LieDesc& d, hipStream_t s, long long N, const void* X, int ldx, const void* Xb, int ldxb, int clear, void* H) {
    if (N == 0) return RDHIP_OK;
    if (d.nrot == 0) {   // EuclideanState: nablaG .= 0 (src/statevectortype.jl:147)
        hipLaunchKernelGGL((fill_identity_kernel<T>), dim3(nblk(N * d.e * d.e)), dim3(256), 0, s, d.e, d.e, N, (T*)H, 0);
        return st(hipGetLastError());
    }
    const long long per = clear ? (long long)d.e * d.e : 9LL * d.nrot;
    hipLaunchKernelGGL((nabla_errstate_kernel<T>), dim3(nblk(N * per)), dim3(256), 0, s, d, N, (const T*)X, ldx, (const T*)Xb, ldxb, clear, (T*)H);
    return st(hipGetLastError());
}
int launch_nabla_errstate(const LieDesc& d, int dtype, hipStream_t s, long long N, const void* X, int ldx, const void* Xb, int ldxb, int clear, void* H) {
    return dtype == RDHIP_F64 ? nabla_T<double>(d, s, N, X, ldx, Xb, ldxb, clear, H) : nabla_T<float>(d, s, N, X, ldx, Xb, ldxb, clear, H);
}

}  // namespace rd
