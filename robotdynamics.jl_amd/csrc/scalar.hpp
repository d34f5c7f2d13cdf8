// scalar.hpp -- scalar types the model code is instantiated on: T (float/double) and Dual<T,K>,
// a forward-mode dual number carrying K tangent directions.  This is the device-side meaning of the
// reference's DiffMethod ForwardAD (src/jacobian_gen.jl:485-529: ForwardDiff.jacobian of
// `evaluate(fun, StaticKnotPoint(z, _x))`), restructured for the GPU: instead of one Dual carrying all
// n+m partials, each wavefront propagates a few tangent columns whose seeds are COMPILE-TIME constants, so
// that (built with -fno-signed-zeros -ffinite-math-only) the structural zeros of the stage Jacobians fold
// away at compile time and only the ~145/247 structurally non-zero entries cost arithmetic.
#pragma once
#ifndef __HIPCC_RTC__      // hipRTC pre-includes the HIP runtime header
#include <hip/hip_runtime.h>
#endif

#define RD_DEV __device__ __forceinline__
#define RD_HD __host__ __device__ __forceinline__

namespace rd {

template <class T, int K>
struct Dual {
    T v;
    T d[K];
    RD_DEV Dual() {}
    RD_DEV Dual(T x) : v(x) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = T(0);
    }
};

template <class S> struct scalar_of { using type = S; };
template <class T, int K> struct scalar_of<Dual<T, K>> { using type = T; };

#define RD_BIN(op, VEXPR, DEXPR)                                                          \
    template <class T, int K> RD_DEV Dual<T, K> operator op(const Dual<T, K>& a, const Dual<T, K>& b) { \
        Dual<T, K> r; r.v = VEXPR;                                                          \
        _Pragma("unroll") for (int i = 0; i < K; ++i) r.d[i] = DEXPR;                     \
        return r; }
RD_BIN(+, a.v + b.v, a.d[i] + b.d[i])
RD_BIN(-, a.v - b.v, a.d[i] - b.d[i])
RD_BIN(*, a.v * b.v, a.d[i] * b.v + a.v * b.d[i])
#undef RD_BIN
template <class T, int K> RD_DEV Dual<T, K> operator/(const Dual<T, K>& a, const Dual<T, K>& b) {
    Dual<T, K> r; r.v = a.v / b.v; const T ib = T(1) / b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> operator-(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -a.d[i];
    return r;
}
// Dual (op) plain scalar
template <class T, int K> RD_DEV Dual<T, K> operator+(const Dual<T, K>& a, T b) { Dual<T, K> r = a; r.v = a.v + b; return r; }
template <class T, int K> RD_DEV Dual<T, K> operator+(T b, const Dual<T, K>& a) { Dual<T, K> r = a; r.v = b + a.v; return r; }
template <class T, int K> RD_DEV Dual<T, K> operator-(const Dual<T, K>& a, T b) { Dual<T, K> r = a; r.v = a.v - b; return r; }
template <class T, int K> RD_DEV Dual<T, K> operator-(T b, const Dual<T, K>& a) { Dual<T, K> r = -a; r.v = b - a.v; return r; }
template <class T, int K> RD_DEV Dual<T, K> operator*(const Dual<T, K>& a, T b) {
    Dual<T, K> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b;
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> operator*(T b, const Dual<T, K>& a) { return a * b; }
template <class T, int K> RD_DEV Dual<T, K> operator/(const Dual<T, K>& a, T b) {
    Dual<T, K> r; r.v = a.v / b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] / b;
    return r;
}

RD_DEV float sin_(float a) { return sinf(a); }
RD_DEV float cos_(float a) { return cosf(a); }
RD_DEV float sqrt_(float a) { return sqrtf(a); }
RD_DEV double sin_(double a) { return sin(a); }
RD_DEV double cos_(double a) { return cos(a); }
RD_DEV double sqrt_(double a) { return sqrt(a); }
// max(0, a) with ForwardDiff's tie rule: ifelse(0 > a, 0, a)
RD_DEV float max0_(float a) { return (0.0f > a) ? 0.0f : a; }
RD_DEV double max0_(double a) { return (0.0 > a) ? 0.0 : a; }

template <class T, int K> RD_DEV Dual<T, K> sin_(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = sin_(a.v); const T c = cos_(a.v);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> cos_(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = cos_(a.v); const T s = -sin_(a.v);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = s * a.d[i];
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> sqrt_(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = sqrt_(a.v); const T c = T(0.5) / r.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> max0_(const Dual<T, K>& a) {
    if (T(0) > a.v) return Dual<T, K>(T(0));
    return a;
}

}  // namespace rd
