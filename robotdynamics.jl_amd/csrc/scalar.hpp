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

// sin and cos together, < 1 ulp each for |x| < 8e5 (checked against long-double sinl / cosl on 10^7 arguments): three-stage Cody-Waite
// reduction by pi/2 (the medium-argument path of fdlibm's e_rem_pio2, all stages unconditionally so that there is no divergence) and
// the fdlibm kernel polynomials with the tail correction.  ~55 fp64 instructions for both values; the device library's separate
// sin and cos (each with its own extended-precision reduction) cost ~300, which made the cartpole kernels transcendental-bound.
// Larger arguments fall back to the library.
RD_DEV void sincos_(double x, double& sn, double& cs) {
    // the fast path runs unconditionally and the library fixes up the lanes it does not cover BEHIND it, under a wave-uniform test: a per-lane branch
    // in front (two code paths, a merge with register copies, four times per RK4 step of the cartpole) cost the small-model kernels ~10 % of their
    // instructions, and they are bound by fp64 issue
    {
        const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
        double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x), t = r;
        double w = fn * 6.07710050630396597660e-11;
        r = t - w; w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
        t = r;
        w = fn * 2.02226624871116645580e-21;
        r = t - w; w = fn * 8.47842766036889956997e-32 - ((t - r) - w);
        const double y0 = r - w, y1 = (r - y0) - w;
        const double z = y0 * y0, v = z * y0;
        const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
        const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
        const double ww = z * z;
        const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                          ww * ww * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
        const double hz = 0.5 * z, w1 = 1.0 - hz;
        const double kc = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
        const int n = (int)fn;
        const double a = (n & 1) ? kc : ks, b = (n & 1) ? ks : kc;
        sn = (n & 2) ? -a : a;
        cs = ((n + 1) & 2) ? -b : b;
    }
    const bool wide = !(__builtin_fabs(x) < 8.0e5);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(wide) != 0, 0)) {
        if (wide) { sn = sin(x); cs = cos(x); }
    }
}
// fp32: one shared three-constant Cody-Waite reduction by pi/2 (exact products inside the FMAs) and the Cephes single-precision polynomials on
// [-pi/4, pi/4]: <= 1.6 ulp for |x| < 1e5 (checked against float64 sin / cos on 3 10^6 arguments per decade), ~22 instructions for both values.
// The device library's sinf + cosf (each with its own reduction and large-argument branch) cost ~3x that, and a cartpole rollout step is four
// dependent stages of them: 0.78 us per fp32 RK4 step, slower than the 13-state satellite's.  Larger arguments fall back to the library.
RD_DEV void sincos_fast_(float x, float& sn, float& cs) {
    const float fn = __builtin_rintf(x * 6.3661977237e-01f);
    float r = __builtin_fmaf(-fn, 1.5707963705062866f, x);
    r = __builtin_fmaf(-fn, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-fn, -1.7151245100058819e-15f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float ks = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float kc = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const int n = (int)fn;
    const float a = (n & 1) ? kc : ks, b = (n & 1) ? ks : kc;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}
// the same polynomials behind a reduction carried out in DOUBLE precision (two-constant Cody-Waite with exact products inside the FMAs): no range
// limit worth a branch -- <= 1.6 ulp up to 1e7, absolute error below 1e-7 as long as the quadrant count fits a double (|x| < 9e15, where
// consecutive floats are 10^9 apart) -- at the price of three conversions
RD_DEV void sincos_wide_(float x, float& sn, float& cs) {
    const double xd = (double)x;
    const double fn = __builtin_rint(xd * 6.36619772367581382433e-01);
    double rd = __builtin_fma(-fn, 1.5707963267948966, xd);
    rd = __builtin_fma(-fn, 6.123233995736766e-17, rd);
    const float r = (float)rd;
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float ks = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float kc = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const int n = (int)(fn - 4.0 * __builtin_rint(fn * 0.25));      // quadrant, -2 .. 2
    const float a = (n & 1) ? kc : ks, b = (n & 1) ? ks : kc;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}
#ifndef RD_SINCOSF
#define RD_SINCOSF 4
#endif
RD_DEV void sincos_(float x, float& sn, float& cs) {
#if RD_SINCOSF == 0          // tuning: the device library
    sn = sinf(x); cs = cosf(x);
#elif RD_SINCOSF == 1        // tuning: per-lane branch
    if (__builtin_fabsf(x) < 1.0e5f) sincos_fast_(x, sn, cs); else { sn = sinf(x); cs = cosf(x); }
#elif RD_SINCOSF == 2        // tuning: no fallback at all
    sincos_fast_(x, sn, cs);
#elif RD_SINCOSF == 4
    // No branch: a branch to a library fallback for large arguments -- per lane or per wavefront -- sits on the latency chain of every rollout step
    // (four dependent stages of it) and measured 0.87 / 0.73 us per fp32 RK4 cartpole step against 0.48 without one (library: 0.77).
    sincos_wide_(x, sn, cs);
#else
    // the fast path unconditionally (straight-line code on the latency chain of a rollout step); the library only if SOME lane of the wavefront
    // holds an argument beyond the reduction's range -- one wave-uniform scalar branch, never taken for physical angles
    sincos_fast_(x, sn, cs);
    const bool big = !(__builtin_fabsf(x) < 1.0e5f);
    if (__builtin_amdgcn_ballot_w64(big) != 0) { if (big) { sn = sinf(x); cs = cosf(x); } }
#endif
}
RD_DEV float sin_(float a) { float s, c; sincos_(a, s, c); return s; }
RD_DEV float cos_(float a) { float s, c; sincos_(a, s, c); return c; }
RD_DEV float sqrt_(float a) { return sqrtf(a); }
RD_DEV double sin_(double a) { double s, c; sincos_(a, s, c); return s; }
RD_DEV double cos_(double a) { double s, c; sincos_(a, s, c); return c; }
RD_DEV double sqrt_(double a) { return sqrt(a); }
// 1 / a: one division shared by several quotients with the same denominator (each quotient then differs from a true division by <= 1 ulp)
// fp32: hardware reciprocal (1 ulp) + one Newton step instead of the ten-instruction IEEE division sequence -- within an ulp of 1 / a, and a third
// of the depth on a rollout step's latency chain (the cartpole evaluates one per stage)
RD_DEV float recip_(float a) { const float r = __builtin_amdgcn_rcpf(a); return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r); }
RD_DEV double recip_(double a) { return 1.0 / a; }
// max(0, a) with ForwardDiff's tie rule: ifelse(0 > a, 0, a)
RD_DEV float max0_(float a) { return (0.0f > a) ? 0.0f : a; }
RD_DEV double max0_(double a) { return (0.0 > a) ? 0.0 : a; }

template <class T, int K> RD_DEV Dual<T, K> sin_(const Dual<T, K>& a) {
    Dual<T, K> r; T c; sincos_(a.v, r.v, c);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> cos_(const Dual<T, K>& a) {
    Dual<T, K> r; T sp; sincos_(a.v, sp, r.v); const T s = -sp;
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
template <class T, int K> RD_DEV Dual<T, K> recip_(const Dual<T, K>& a) {
    Dual<T, K> r; r.v = recip_(a.v); const T c = -(r.v * r.v);
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class T, int K> RD_DEV Dual<T, K> max0_(const Dual<T, K>& a) {
    if (T(0) > a.v) return Dual<T, K>(T(0));
    return a;
}

}  // namespace rd
